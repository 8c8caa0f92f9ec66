#!/usr/bin/env python3
"""bench.py -- throughput of the per-field NTSC composite / VHS path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "Config 2"): a 720x486, 30 fps, 10-second
synthetic colour-bars clip (300 frames -> 600 output fields; frame k = the 8-bar table rotated by
k pixels), full `-vhs` preset (head switching, luma noise 4, chroma noise 16, chroma phase noise
4, chroma dropout 4, SP tape speed).  One "step" = one pass of the hot path over one such clip
per GPU, frames resident in HBM.  With N GPUs the clip is N x 300 frames, dealt frame-round-robin
to the ranks (weak scaling; no data-path collective -- fields are independent once their rand()
stream positions are fixed).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Other ways to deal the work (same kernels, same JSON line):
  --scaling strong     ONE 300-frame clip dealt frame-round-robin over the ranks (total work fixed)
  --streams S          BASELINE configs[3]: S independent 300-frame streams, stream s on rank s % N

Prints ONE JSON line on rank 0.  Beside the contract keys it carries (N = 1 only, --no-extras to
skip): `value_sustained` (the same step repeated for >= 0.5 s), `end_to_end` (PCIe-inclusive
ntscsim_frames_host rates), `variant422` (the 8-bit YUV422P tool), `sizes` (1920x1080, 3840x2160)
and `presets` (the default preset = BASELINE configs[0]'s workload on the GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "composite-video-simulator_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md: 256 CU x 4 SIMD-32 at 2.4 GHz
N_SIMD = 256 * 4


from bench_side import (device_rate, device_stream_rate, emit, extras, make_bars_clip, time_steps,  # noqa: E402,F401
                        variant_contexts)
from bench_variant import main_to_composite  # noqa: E402


def _cpu_engine(kind, params):
    """The single-threaded CPU engines bench.py times beside the GPU: 'reference' = the
    reference's own composite_layer() text compiled into oracle/_ref/libntsc_ref.so by
    oracle/build_ref.sh (process-wide libc rand(), like the tool); 'port' = oracle/ntsc_oracle.c."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs as L
    return L, (L.RefStream(params) if kind == "reference" else L.OracleStream(params))


def cpu_baseline(kind, params, w, h, n_fields, check_against=None):
    """Engine `kind`, 1 thread like the reference, on the first n_fields of the same clip.
    Returns (fields_per_s, n_checked_ok)."""
    import numpy as np
    L, o = _cpu_engine(kind, params)
    dst = np.zeros((h, w, 4), np.uint8)
    frames = {}
    t = 0.0
    ok = 0
    for cur in range(n_fields):
        fr = cur // 2
        if fr not in frames:
            frames = {fr: L.bars(w, h, fr)}
        t0 = time.perf_counter()
        o.field(dst, frames[fr], (cur & 1) ^ 1, cur)
        t += time.perf_counter() - t0
        if check_against is not None and cur in check_against:
            field = (cur & 1) ^ 1
            if np.array_equal(dst[field::2], check_against[cur]):
                ok += 1
            else:
                raise AssertionError("bench: HIP output of field %d differs from the %s" % (cur, kind))
    return n_fields / t, ok


def cpu_worker(args):
    """`bench.py --cpu-worker A B`: one process of the all-cores CPU leg.  Prepares fields [A, B)
    of the clip for the port (rand() stream positioned by jump-ahead, as a multi-threaded CPU
    implementation would), prints "ready", waits for a line on stdin, runs, prints its end time."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "composite-video-simulator_amd"))
    from ntscsim import _capi, shard
    a, b = int(args.cpu_worker[0]), int(args.cpu_worker[1])
    params = _capi.make_params(args.preset.split())
    L, o = _cpu_engine("port", params)
    o.skip(shard.rng_pos_of_field(params, args.width, args.height, a))
    w, h = args.width, args.height
    dst = np.zeros((h, w, 4), np.uint8)
    src = {}
    for cur in range(a, b):
        src[cur // 2] = L.bars(w, h, cur // 2)
    print("ready", flush=True)
    sys.stdin.readline()
    for cur in range(a, b):
        o.field(dst, src[cur // 2], (cur & 1) ^ 1, cur)
    print("%.6f %d" % (time.time(), int(dst.sum() & 0xFFFF)), flush=True)


def usable_cpus():
    """Logical CPUs this process may actually use: affinity mask, capped by a cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_all_cores(args, n_workers, fields_each):
    """All host CPUs: n_workers processes x fields_each fields of the port, released together once
    every process is ready.  Returns fields_per_s."""
    import subprocess
    procs = []
    for i in range(n_workers):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i * fields_each),
               str((i + 1) * fields_each), "--width", str(args.width), "--height",
               str(args.height), "--preset=" + args.preset]
        procs.append(subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL,
                                      env=dict(os.environ, OMP_NUM_THREADS="1")))
    try:
        for p in procs:
            if p.stdout.readline().strip() != b"ready":
                raise RuntimeError("bench: CPU worker failed to start")
        t0 = time.time()
        for p in procs:
            p.stdin.write(b"go\n")
            p.stdin.flush()
        ends = [float(p.stdout.readline().split()[0]) for p in procs]
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
            p.wait(timeout=60)
    return n_workers * fields_each / (max(ends) - t0)


def valu_roofline(w, h, preset, fields_per_step, tm_ms, ms_per_step):
    """Cycle-weighted VALU roofline from the committed census: wave-instructions per launch
    (SQ_INSTS_VALU, PMC) x mean issue cost of the kernel's instruction mix (tools/isa_cost.py on the
    shipped ISA, priced with tools/valu_rate_probe.hip) = SIMD pipe cycles the step needs."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))["%dx%d %s" % (w, h, preset)]
        kern = tj["valu"]
    except Exception:
        return None, None
    scale = fields_per_step / float(tj["fields_per_launch"])
    peak = N_SIMD * VALU_CLOCK_HZ
    need = {k: v["wave_insts_per_launch"] * scale * v["mean_cycles_per_inst"] for k, v in kern.items()}
    # the same wave-instructions priced at the guide's nominal issue costs (4 cycles per wave64 instruction
    # for fp64 and the other half-rate opcodes, 2 for the full-rate ones): the pipe's real capacity; the
    # measured slowest-wave figures above (4.3 / 2.7) include what a probe loses to arbitration
    need_nom = {k: v["wave_insts_per_launch"] * scale * v.get("mean_cycles_per_inst_nominal", v["mean_cycles_per_inst"])
                for k, v in kern.items()}
    out = {
        "bound": "valu-issue (cycle-weighted)",
        "source": "needed cycles: REPLAYED from profiles/traffic.json (SQ_INSTS_VALU per launch of a PMC pass x the ISA "
                  "census' mean issue cost); the fractions divide them by THIS run's live times (ms_per_step, kernel_ms)",
        "unit": "SIMD pipe cycles/s",
        "peak": peak,
        "pipe_cycles_per_step": need,
        "pipe_cycles_per_step_nominal": need_nom,
        "mean_cycles_per_inst": {k: v["mean_cycles_per_inst"] for k, v in kern.items()},
        "path_frac": sum(need.values()) / (ms_per_step * 1e-3) / peak,
        "path_frac_nominal": sum(need_nom.values()) / (ms_per_step * 1e-3) / peak,
        "hbm_frac_ceiling_exact_mode": (8.0 * w * ((h + 1) // 2 + h // 2) / 2.0 * fields_per_step / (HBM_PEAK_GBS * 1e9)) /
                                       (sum(need_nom.values()) / peak),
        "note": "needed = SQ_INSTS_VALU per launch (profiles/*_pmc_summary.txt) x the mean issue cost "
                "of each kernel's instruction mix (profiles/*_isa_cost.json).  path_frac prices it with the "
                "slowest-wave figures of tools/valu_rate_probe.hip (4.3 cycles for fp64 and the other "
                "half-rate opcodes, 2.7 for the full-rate ones: profiles/*_valu_rates.txt); "
                "path_frac_nominal with the pipe's nominal 4 / 2 cycles (MI355X_MICROARCH.md) -- the "
                "stricter figure, and the one to close on.  peak = 1024 SIMDs x 2.4 GHz; both are needed / "
                "(ms_per_step x peak) with the steps in flight.  hbm_frac_ceiling_exact_mode = the HBM "
                "roofline fraction a kernel chain with exactly this arithmetic would reach at 100 % "
                "nominal VALU issue: the reference's fp64 op count per pixel, not memory, bounds "
                "roofline.frac in exact mode",
    }
    if tm_ms.get("decode"):
        out["k_decode_frac"] = need.get("k_decode", 0.0) / (tm_ms["decode"] * 1e-3) / peak
    return out, tj.get("k_decode_hbm_bytes_per_launch", None) and tj["k_decode_hbm_bytes_per_launch"] * scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--width", type=int, default=720)
    ap.add_argument("--height", type=int, default=486)
    ap.add_argument("--frames", type=int, default=300, help="frames per clip (per GPU per step when scaling is weak)")
    ap.add_argument("--preset", default="-vhs", help="reference CLI switches, space separated")
    ap.add_argument("--inflight", type=int, default=4,
                    help="steps in flight: contexts (own HIP stream, scratch and destination "
                         "clip each) the steps rotate over")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns a 300-frame slice of an N x 300-frame clip; strong: "
                         "one 300-frame clip is dealt over the ranks")
    ap.add_argument("--streams", type=int, default=0,
                    help="BASELINE configs[3]: this many independent clips, stream s on rank s %% N")
    ap.add_argument("--tool", default="ntsc", choices=["ntsc", "to_composite"],
                    help="ntsc = ffmpeg_ntsc's composite_layer on BGRA (BASELINE's metric, default); "
                         "to_composite = the 8-bit YUV422P sibling (ffmpeg_to_composite)")
    ap.add_argument("--mode", default="exact", choices=["exact", "fast32", "float"],
                    help="exact = bit-identical to the reference (fp64, default); fast32 = the exact kernels with fp32 "
                         "filter states; float = the all-float pipeline (csrc/ntsc_float.hip); both within the "
                         "tolerance of tests/test_gpu_fast_mode.py")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only for dry runs of the "
                         "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (and take the barrier / MAX / all-gather branches) even "
                         "with one rank: exercises the RCCL code path on a single GPU")
    ap.add_argument("--cpu-fields", type=int, default=300,
                    help="fields of the clip timed on the single-threaded CPU engines (0 = skip)")
    ap.add_argument("--cpu-mt-fields", type=int, default=8,
                    help="fields per process of the all-cores CPU leg (0 = skip that leg)")
    ap.add_argument("--sustain-seconds", type=float, default=0.5,
                    help="before the W warm-up and K timed steps, repeat the same step for at least this long -> value_sustained")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip end_to_end / variant422 / sizes / presets (they run at N = 1 only)")
    ap.add_argument("--cpu-worker", nargs=2, metavar=("A", "B"), default=None,
                    help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)
    if args.tool == "to_composite":
        return main_to_composite(args)

    import torch
    import ntscsim
    from ntscsim import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    red_dev = dev                       # device of the tiny tensors the ranks exchange
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            red_dev = torch.device("cpu")

    w, h = args.width, args.height
    flags = args.preset.split()
    params = ntscsim.make_params(flags)

    # ---- this rank's share: a list of clips, each (first_frame, frame_stride, n_frames, jobs)
    def shard_of(r):
        """[(first source frame, stride, local frames, jobs)] of rank r; jobs = (cur, field, fieldno, rng_pos)."""
        if args.streams > 0:          # independent clips, whole clips per rank
            full = shard.jobs_for_rank(params, w, h, 2 * args.frames, 0, 1)
            return [(37 * sidx, 1, args.frames, full) for sidx in range(r, args.streams, world)]
        n_global = 2 * args.frames * (world if args.scaling == "weak" else 1)
        jobs = shard.jobs_for_rank(params, w, h, n_global, r, world)
        return [(r, world, (len(jobs) + 1) // 2, jobs)] if jobs else []

    def build(r, nq):
        """Resident inputs + prepared batches of rank r's share, replicated over nq contexts."""
        clips = shard_of(r)
        ctxs = []
        for q in range(nq):
            sm = ntscsim.FieldSimulator(params=params, device=local_rank)
            if args.mode != "exact":
                sm.set_mode(ntscsim._capi.MODE_FAST32 if args.mode == "fast32" else ntscsim._capi.MODE_FLOAT)
            plans, dsts = [], []
            for (first, stride, nloc, jobs) in clips:
                src = build.src.setdefault((first, stride, nloc), make_bars_clip(torch, nloc, w, h, first, stride, dev))
                dst = torch.zeros((nloc, h, w, 4), dtype=torch.uint8, device=dev)
                # local index of global frame g: streams number their own frames 0.., shards own the
                # frames first, first + stride, ...
                def lidx(g):
                    return g if args.streams > 0 else (g - first) // stride
                loc = [(lidx(cur // 2), lidx(cur // 2), field, fieldno) for (cur, field, fieldno, _) in jobs]
                plans.append(sm.prepare(sm.build_descs(src, dst, loc, rng_pos=[j[3] for j in jobs]), w, h))
                dsts.append(dst)
            ctxs.append((sm, plans, dsts, torch.cuda.Stream(dev)))
        return clips, ctxs
    build.src = {}

    # Steps are independent passes over the clip(s), so consecutive steps are software-pipelined over
    # `inflight` contexts, each with its own HIP stream, scratch and destination clip(s) (the ~2,300
    # long-running wavefronts of one 600-field step cannot load 1,024 SIMDs evenly on their own).
    nq = max(1, args.inflight)
    clips, ctxs = build(rank, nq)
    fields_per_step_local = sum(len(c[3]) for c in clips)
    torch.cuda.synchronize(dev)

    def step(i):
        sm, plans, _, st = ctxs[i % nq]
        for pl in plans:
            sm.run_prepared(pl, stream=st.cuda_stream)

    # every context runs once before anything is counted (first-call allocations of its scratch), then
    # the W warm-up steps
    for i in range(nq):
        step(i)
    torch.cuda.synchronize(dev)
    # ---- the same step, repeated for >= sustain-seconds (no other change of configuration).  It runs BEFORE the
    # W warm-up and K timed steps: a GPU that has just left idle needs ~30 ms of load before its clocks are where
    # they stay (measured: K=20 after W=5 712k fields/s, after W=40 759k, after W=80 759k), and the driver's
    # W is 4 ms of work -- so the K timed steps follow half a second of the same work instead of an idle GPU.
    sustained = None
    if args.sustain_seconds > 0 and fields_per_step_local:
        n_s, t1 = 0, time.perf_counter()
        while True:
            for i in range(4 * nq):
                step(n_s + i)
            n_s += 4 * nq
            torch.cuda.synchronize(dev)
            if time.perf_counter() - t1 >= args.sustain_seconds:
                break
        sustained = (n_s, time.perf_counter() - t1)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    # Kernel durations: hipEvents recorded on the launch stream around every kernel of the same
    # step, run right after the timed region on one context.  (Event records between kernels
    # serialise the in-flight streams, so they are kept out of the throughput measurement.)
    tm = {"calls": 0, "setup": 0.0, "encode": 0.0, "decode": 0.0}
    if fields_per_step_local:
        nprof = max(3, min(args.steps, 10))
        sm0, plans0, _, st0 = ctxs[0]
        sm0.set_profiling(True)
        for _ in range(nprof):
            sm0.run_prepared(plans0[0], stream=st0.cuda_stream)
        torch.cuda.synchronize(dev)
        tm = sm0.timings_ms()
        sm0.set_profiling(False)

    def checksum(dsts):
        return int(sum(int(d.to(torch.int64).sum().item()) for d in dsts))

    my_cs = checksum(ctxs[0][2]) if ctxs[0][2] else 0
    allcs, verified = None, None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # gather a checksum and the field count per rank (the only exchange the path needs)
        cs = torch.tensor([my_cs, fields_per_step_local], dtype=torch.int64, device=red_dev)
        allg = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allg, cs)
        allcs = [int(c_[0].item()) for c_ in allg]
        fields_all = [int(c_[1].item()) for c_ in allg]
        if sustained is not None:
            ts = torch.tensor([sustained[1]], dtype=torch.float64, device=red_dev)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            sustained = (sustained[0], float(ts.item()))
    else:
        fields_all = [fields_per_step_local]

    total_fields_per_step = sum(fields_all)
    value = total_fields_per_step * args.steps / elapsed
    alg_bytes_field = 8 * w * ((ntscsim.field_rows(h, 0) + ntscsim.field_rows(h, 1)) / 2.0)

    out = None
    if rank == 0:
        # every rank's checksum, recomputed here on ONE GPU from that rank's share of the work
        if dist is not None:
            exp = []
            for r in range(world):
                if r == 0:
                    exp.append(my_cs)
                    continue
                _, cx = build(r, 1)
                smr, plr, dsr, str_ = cx[0]
                for pl in plr:
                    smr.run_prepared(pl, stream=str_.cuda_stream)
                torch.cuda.synchronize(dev)
                exp.append(checksum(dsr))
                for pl in plr:
                    smr.free_prepared(pl)
                smr.close()
            verified = exp == allcs
        calls = max(1, tm["calls"])
        dec_ms = tm["decode"] / calls
        enc_ms = tm["encode"] / calls
        set_ms = tm["setup"] / calls
        chain_ms = dec_ms + enc_ms + set_ms
        launch_fields = len(clips[0][3]) if clips else 0
        alg_bytes_launch = alg_bytes_field * launch_fields
        achieved = alg_bytes_launch / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        valu, traffic = (None, None)
        if args.mode == "exact":
            valu, traffic = valu_roofline(w, h, args.preset, launch_fields,
                                          {"decode": dec_ms, "encode": enc_ms, "setup": set_ms},
                                          ms_per_step * launch_fields / max(1, fields_per_step_local))
            if valu and dec_ms > 0:
                kd = valu["pipe_cycles_per_step_nominal"].get("k_decode", 0.0)
                valu["k_decode_frac_nominal"] = kd / (dec_ms * 1e-3) / valu["peak"]
        if args.streams > 0:
            deal = "%d independent %d-frame streams, stream s on rank s %% %d" % (args.streams, args.frames, world)
        elif args.scaling == "weak":
            deal = "%d frames -> %d fields per GPU per step, frame-round-robin over %d GPU(s)" % (
                args.frames, 2 * args.frames, world)
        else:
            deal = "one %d-frame clip (%d fields per step) dealt frame-round-robin over %d GPU(s)" % (
                args.frames, 2 * args.frames, world)
        out = {
            "metric": "frames/sec (output frames = fields; %dx%d %s, preset '%s'; steady-state "
                      "pipelined throughput, %d steps in flight)" % (w, h, "PAL" if params.tv_standard else "NTSC",
                                                                      args.preset if args.preset.strip() else "default", nq),
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling if args.streams == 0 else "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.mode == "exact" else "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d 30fps 10s colour-bars clip, preset '%s': %s" % (w, h, args.preset, deal),
                "fields_per_step_per_gpu": fields_all if world > 1 else fields_per_step_local,
                "input_frames_per_sec": value / 2.0,
                "steps_in_flight": nq,
                "pre_roll": None if sustained is None else
                    {"steps": sustained[0], "seconds": sustained[1],
                     "note": "untimed steps of the same work BEFORE the W warm-up steps (= the value_sustained leg; "
                             "--sustain-seconds 0 removes it): a GPU that has just left idle needs ~30 ms of load "
                             "before its clocks settle, the driver's W is ~4 ms of work.  Numbers of earlier rounds "
                             "measured without it (r01, r02) are ~4-6 % lower for that reason alone"},
                "rank_checksums": allcs,
                "rank_checksums_verified": verified,
                "rank_checksums_note": None if dist is None else
                    "sum of all bytes of each rank's destination clip(s); verified = rank 0 re-ran every "
                    "rank's share on its own GPU after the timed region and got the same sums",
                "mode": "exact (bit-identical to the reference: fp64, no FMA contraction)"
                        if args.mode == "exact" else
                        ("fast32 (fp32 filters; <= 1 LSB per 8-bit channel vs the reference)" if args.mode == "fast32" else
                         "float (all-float pipeline; <= 1 LSB per 8-bit channel vs the reference)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_decode" if args.mode != "float" else "k_decode_fp",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # which of these fields were measured in THIS run and which are replayed from a tracked profile
                "live_fields": ["achieved", "frac", "kernel_ms", "kernel_ms_all", "path_achieved",
                                "valu.path_frac", "valu.path_frac_nominal", "valu.k_decode_frac", "valu.k_decode_frac_nominal"],
                "traffic_source": None if traffic is None else
                    "REPLAYED, not measured in this run: profiles/traffic.json (PMC passes of tools/pmc.sh, calibrated with "
                    "tools/fetch_probe.hip; the profile's round is in its 'round' key), scaled to this run's fields per launch",
                "algorithmic_bytes_per_launch": alg_bytes_launch,
                "kernel_ms": dec_ms,
                "note": "the exact path is bound by VALU issue, not by HBM (DESIGN.md, `valu` below): "
                        "frac is reported as the contract asks and is capped near 0.25 in exact mode by the "
                        "reference's fp64 arithmetic (valu.hbm_frac_ceiling_exact_mode); path_achieved uses "
                        "encode+decode+setup time",
                "path_achieved": alg_bytes_launch / (chain_ms * 1e-3) / 1e9 if chain_ms > 0 else 0.0,
                "kernel_ms_all": {"setup": set_ms, "encode": enc_ms, "decode": dec_ms},
                "valu": valu,
                "kernel_timing": "hipEvents on the launch stream, %d steps on one context right "
                                 "after the timed region (un-shared launches; rocprofv3 "
                                 "--kernel-trace --stats of `--inflight 1` agrees, profiles/README.md)" % calls,
            },
        }
        if sustained is not None:
            out["value_sustained"] = total_fields_per_step * sustained[0] / sustained[1]
            out["sustained"] = {"steps": sustained[0], "seconds": sustained[1], "order": "sustained leg, then W warm-up steps, then the K timed steps"}
        if world == 1 and not args.no_extras and args.mode == "exact" and args.streams == 0:
            try:
                out.update(extras(torch, ntscsim, dev, local_rank, args))
            except Exception as e:      # never lose the headline line to an extra
                out["extras_error"] = repr(e)
        if world == 1 and args.cpu_fields > 0 and args.mode == "exact" and fields_per_step_local:
            import numpy as np
            dst = ctxs[0][2][0]
            ncpu = min(args.cpu_fields, len(clips[0][3]))
            # parity spot check on the fields the oracle produces anyway
            host = dst.cpu().numpy()
            chk = {}
            for cur in (0, 1, 2, 3, ncpu - 2, ncpu - 1):
                if 0 <= cur < ncpu:
                    field = (cur & 1) ^ 1
                    chk[cur] = host[cur // 2][field::2].copy()
            # note: field pairs share a dst frame, so each field's rows are intact
            port_fps, ok = cpu_baseline("port", params, w, h, ncpu, chk)
            have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libntsc_ref.so"))
            if have_ref:
                cpu_fps, ok_ref = cpu_baseline("reference", params, w, h, ncpu, chk)
                kind = "reference"
                what = ("composite_layer() of the reference itself (its ffmpeg_ntsc.cpp text "
                        "compiled by oracle/build_ref.sh into oracle/_ref/libntsc_ref.so, g++ -O2 "
                        "-ffp-contract=off, libc rand()), single-threaded like the tool")
            else:
                cpu_fps, ok_ref, kind = port_fps, ok, "port"
                what = ("oracle/ntsc_oracle.c (bit-exact restatement of the single-threaded "
                        "reference; oracle/_ref not present on this box), gcc -O2 -ffp-contract=off")
            out["cpu_baseline"] = {
                "value": cpu_fps,
                "unit": "frames/s",
                "cores": 1,
                "kind": kind,
                "sample": "first %d fields of the same clip, 1 thread; %d fields byte-compared with the HIP output" % (ncpu, ok_ref),
                "engine": what,
                "host_cpus": os.cpu_count(),
                "port_1core": port_fps,
            }
            out["speedup_vs_cpu_1core"] = value / cpu_fps
            if args.cpu_mt_fields > 0:
                nw, quota = usable_cpus()
                mt_fps = cpu_all_cores(args, nw, args.cpu_mt_fields)
                out["cpu_baseline"]["port_all_cores"] = {
                    "value": mt_fps, "cores": nw, "cgroup_cpu_quota": quota,
                    "sample": "%d processes (one per usable logical CPU: affinity mask capped by "
                              "the cgroup CPU quota) x %d fields of the port, rand() positions by "
                              "jump-ahead, released together" % (nw, args.cpu_mt_fields)}
                out["speedup_vs_cpu_all_cores"] = value / mt_fps
    for sm, plans, _, _ in ctxs:
        for pl in plans:
            sm.free_prepared(pl)
        sm.close()
    if dist is not None:
        dist.destroy_process_group()
    # (after the teardown: the contract object is the last thing this process writes to stdout)
    if out is not None:
        emit(out)


if __name__ == "__main__":
    main()
