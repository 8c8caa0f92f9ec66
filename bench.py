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

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "composite-video-simulator_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md: 256 CU x 4 SIMD at 2.4 GHz; one wave64 VALU instruction per 4 cycles
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64 (FMA-counted); the exact path cannot use FMA


def make_bars_clip(torch, n_frames, w, h, first_frame, stride, device):
    """Frames first_frame, first_frame+stride, ...: BGRA 8-bar 75% bars rotated by the frame
    index; alpha 0.  uint8 [n, h, w, 4] in HBM."""
    table = torch.tensor([0xC0C0C0, 0xC0C000, 0x00C0C0, 0x00C000,
                          0xC000C0, 0xC00000, 0x0000C0, 0x000000], dtype=torch.int64, device=device)
    x = torch.arange(w, device=device, dtype=torch.int64)
    rot = (first_frame + stride * torch.arange(n_frames, device=device, dtype=torch.int64))
    sx = (x[None, :] + rot[:, None]) % w
    px = table[(8 * sx) // w]                                   # [n, w] 0xRRGGBB
    row = torch.stack([px & 0xFF, (px >> 8) & 0xFF, (px >> 16) & 0xFF, torch.zeros_like(px)],
                      dim=-1).to(torch.uint8)                   # B, G, R, A
    return row[:, None, :, :].expand(n_frames, h, w, 4).contiguous()


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=720)
    ap.add_argument("--height", type=int, default=486)
    ap.add_argument("--frames", type=int, default=300, help="frames per GPU per step")
    ap.add_argument("--preset", default="-vhs", help="reference CLI switches, space separated")
    ap.add_argument("--inflight", type=int, default=3,
                    help="steps in flight: contexts (own HIP stream, scratch and destination "
                         "clip each) the steps rotate over")
    ap.add_argument("--mode", default="exact", choices=["exact", "fast32"],
                    help="exact = bit-identical to the reference (fp64, default); fast32 = fp32 "
                         "filters within the tolerance of tests/test_gpu_fast_mode.py")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only for dry runs of the "
                         "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-fields", type=int, default=300,
                    help="fields of the clip timed on the single-threaded CPU engines (0 = skip)")
    ap.add_argument("--cpu-mt-fields", type=int, default=8,
                    help="fields per process of the all-cores CPU leg (0 = skip that leg)")
    ap.add_argument("--cpu-worker", nargs=2, metavar=("A", "B"), default=None,
                    help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        return cpu_worker(args)

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
            red_dev = torch.device("cpu")

    w, h = args.width, args.height
    flags = args.preset.split()
    params = ntscsim.make_params(flags)
    n_frames_local = args.frames
    n_fields_global = 2 * n_frames_local * world
    jobs = shard.jobs_for_rank(params, w, h, n_fields_global, rank, world)
    assert len(jobs) == 2 * n_frames_local

    # inputs resident in HBM: this rank's frames rank, rank+world, ...
    src = make_bars_clip(torch, n_frames_local, w, h, rank, world, dev)
    # job -> (local src frame, local dst frame, field, fieldno), explicit rand() positions
    loc = [((cur // 2 - rank) // world, (cur // 2 - rank) // world, field, fieldno)
           for (cur, field, fieldno, _) in jobs]
    # Steps are independent passes over the clip, so consecutive steps are software-pipelined over
    # `inflight` contexts, each with its own HIP stream, scratch and destination clip (the 2,300
    # long-running wavefronts of one 600-field step cannot load 1,024 SIMDs evenly on their own).
    nq = max(1, args.inflight)
    streams = [torch.cuda.Stream(dev) for _ in range(nq)]
    sims = [ntscsim.FieldSimulator(params=params, device=local_rank) for _ in range(nq)]
    if args.mode == "fast32":
        for sm in sims:
            sm.set_mode(ntscsim._capi.MODE_FAST32)
    dsts = [torch.zeros((n_frames_local, h, w, 4), dtype=torch.uint8, device=dev) for _ in range(nq)]
    descs = [sm.build_descs(src, d, loc, rng_pos=[j[3] for j in jobs]) for sm, d in zip(sims, dsts)]
    # prepared batches: descriptor validation, each field's rand() window (a pure function of its
    # stream position) and the record upload happen once; a step is the kernel chain over the
    # resident frames + records (ntscsim_batch_run)
    plans = [sm.prepare(d, w, h) for sm, d in zip(sims, descs)]
    torch.cuda.synchronize(dev)

    def step(i):
        q = i % nq
        sims[q].run_prepared(plans[q], stream=streams[q].cuda_stream)

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
    # serialise the two in-flight streams, so they are kept out of the throughput measurement.)
    nprof = max(3, min(args.steps, 10))
    sims[0].set_profiling(True)
    for _ in range(nprof):
        sims[0].run_prepared(plans[0], stream=streams[0].cuda_stream)
    torch.cuda.synchronize(dev)
    tm = sims[0].timings_ms()
    sims[0].set_profiling(False)
    dst = dsts[0]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # gather a checksum per rank (the only exchange the path needs)
        cs = torch.tensor([int(dst.to(torch.int64).sum().item())], dtype=torch.int64, device=red_dev)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)

    fields_per_step_local = len(jobs)
    total_fields = fields_per_step_local * world * args.steps
    value = total_fields / elapsed
    L_rows = (h + 1) // 2  # both parities have ceil/floor; 486 -> 243
    alg_bytes_field = 8 * w * ((ntscsim.field_rows(h, 0) + ntscsim.field_rows(h, 1)) / 2.0)
    alg_bytes_launch = alg_bytes_field * fields_per_step_local

    out = None
    if rank == 0:
        calls = max(1, tm["calls"])
        dec_ms = tm["decode"] / calls
        enc_ms = tm["encode"] / calls
        set_ms = tm["setup"] / calls
        chain_ms = dec_ms + enc_ms + set_ms
        achieved = alg_bytes_launch / (dec_ms * 1e-3) / 1e9
        traffic = None
        valu = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%dx%d %s" % (w, h, args.preset)
                if key in tj:
                    scale = fields_per_step_local / float(tj[key]["fields_per_launch"])
                    traffic = tj[key]["k_decode_hbm_bytes_per_launch"] * scale
                    if args.mode == "exact" and "valu_wave_insts_per_launch" in tj[key]:
                        vi = {k_: v_ * scale for k_, v_ in tj[key]["valu_wave_insts_per_launch"].items()}
                        # every wave64 VALU instruction (fp64 or int32) occupies its SIMD for one
                        # quad-cycle (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU, profiles/README.md)
                        vpeak = 256 * 4 * VALU_CLOCK_HZ / 4.0
                        valu = {
                            "bound": "valu-issue",
                            "unit": "wave64 VALU instructions/s",
                            "peak": vpeak,
                            "insts_per_step": vi,
                            "k_decode_achieved": vi["k_decode"] / (dec_ms * 1e-3),
                            "k_decode_frac": vi["k_decode"] / (dec_ms * 1e-3) / vpeak,
                            "path_achieved": sum(vi.values()) / (elapsed / args.steps),
                            "path_frac": sum(vi.values()) / (elapsed / args.steps) / vpeak,
                            "note": "instruction counts from the SQ_INSTS_VALU PMC pass "
                                    "(profiles/r01_pmc_summary.txt); peak = 1024 SIMDs x 2.4 GHz / 4 "
                                    "cycles; k_decode alone = one launch (2,315 waves cannot balance "
                                    "1,024 SIMDs), path = whole chain with the steps in flight",
                        }
            except Exception:
                traffic = None
        out = {
            "metric": "frames/sec (output frames = fields; 720x486 NTSC, full VHS preset)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.mode == "exact" else "f32",
            "data": "synthetic",
            "config": {
                "workload": "%dx%d 30fps 10s colour-bars clip (%d frames -> %d fields per GPU per "
                            "step), preset '%s', frame-round-robin over %d GPU(s)" % (
                                w, h, n_frames_local, fields_per_step_local, args.preset, world),
                "fields_per_step_per_gpu": fields_per_step_local,
                "input_frames_per_sec": value / 2.0,
                "steps_in_flight": nq,
                "rank_checksums": [int(c_.item()) for c_ in allcs] if dist is not None else None,
                "mode": "exact (bit-identical to the reference: fp64, no FMA contraction)"
                        if args.mode == "exact" else
                        "fast32 (fp32 filters; <= 1 LSB per 8-bit channel vs the reference)",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_decode",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes_launch,
                "kernel_ms": dec_ms,
                "note": "exact mode is fp64-VALU/latency bound, not HBM bound (DESIGN.md); "
                        "path_achieved uses encode+decode+setup time",
                "path_achieved": alg_bytes_launch / (chain_ms * 1e-3) / 1e9,
                "kernel_ms_all": {"setup": set_ms, "encode": enc_ms, "decode": dec_ms},
                "valu": valu,
                "kernel_timing": "hipEvents on the launch stream, %d steps on one context right "
                                 "after the timed region (un-shared launches; rocprofv3 of this "
                                 "command shows them in its Min column, --inflight 1 in its "
                                 "Average -- profiles/README.md)" % calls,
            },
        }
        if world == 1 and args.cpu_fields > 0 and args.mode == "exact":
            import numpy as np
            ncpu = min(args.cpu_fields, fields_per_step_local)
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
                "sample": "first %d fields of the same clip, %s; %d fields compared byte-for-byte "
                          "with the HIP output" % (ncpu, what, ok_ref),
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
        print(json.dumps(out), flush=True)
    for sm, pl in zip(sims, plans):
        sm.free_prepared(pl)
        sm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
