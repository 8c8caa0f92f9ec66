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


def time_steps(torch, dev, fn, reps):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(reps):
        fn(i)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps


def device_rate(torch, ntscsim, dev, local_rank, flags, w, h, n_frames, reps, inflight=3, params=None, kernels=None):
    """fields/s of the BGRA path on a resident bars clip of n_frames frames (both fields each).
    kernels: a list that receives the kernel forms the step enqueued (ntscsim_debug_last_kernels)."""
    from ntscsim import shard
    if params is None:
        params = ntscsim.make_params(flags)
    jobs = shard.jobs_for_rank(params, w, h, 2 * n_frames, 0, 1)
    src = make_bars_clip(torch, n_frames, w, h, 0, 1, dev)
    loc = [(cur // 2, cur // 2, field, fieldno) for (cur, field, fieldno, _) in jobs]
    sims, plans, dsts, streams = [], [], [], []
    for _ in range(inflight):
        sm = ntscsim.FieldSimulator(params=params, device=local_rank)
        d = torch.zeros((n_frames, h, w, 4), dtype=torch.uint8, device=dev)
        plans.append(sm.prepare(sm.build_descs(src, d, loc, rng_pos=[j[3] for j in jobs]), w, h))
        sims.append(sm); dsts.append(d); streams.append(torch.cuda.Stream(dev))
    def step(i):
        q = i % inflight
        sims[q].run_prepared(plans[q], stream=streams[q].cuda_stream)
    for i in range(inflight):
        step(i)
    dt = time_steps(torch, dev, step, reps)
    if kernels is not None:
        kernels.extend(sims[0].last_kernels())
    for sm, pl in zip(sims, plans):
        sm.free_prepared(pl); sm.close()
    return len(jobs) / dt


def device_stream_rate(torch, ntscsim, dev, local_rank, params, w, h, n_frames, steps, inflight, threads=2):
    """A device-resident STREAM of fresh batches (not a replay of a prepared one): step s is the NEXT 2 * n_frames
    fields of one long stream -- fieldno = s * nf + k, rand() position continuing where step s - 1 ended -- sent
    through ntscsim_fields_device(), i.e. descriptor validation, rand() window derivation per field, record
    upload and the kernel chain are all inside the clock.  `inflight` contexts (own stream, scratch, destination
    clip) take the steps round-robin; `threads` host threads drive them (ctypes releases the GIL, a ctx is only
    ever used by one thread), so the preparation of one step overlaps the GPU work of the others.
    Returns (fields/s, verified): verified = the last step's output equals the same fields run as one
    ordinary batch with explicit rand() positions on a fresh context."""
    import threading
    import numpy as np
    from ntscsim import _capi
    nf = 2 * n_frames
    src = make_bars_clip(torch, n_frames, w, h, 0, 1, dev)
    calls = [ntscsim.calls_per_field(params, w, h, 0), ntscsim.calls_per_field(params, w, h, 1)]
    draws_per_step = sum(calls[(k & 1) ^ 1] for k in range(nf))
    dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("sls", "<i4"), ("dls", "<i4"), ("field", "<u4"),
                   ("flags", "<u4"), ("fieldno", "<u8"), ("rng_pos", "<u8")])
    assert dt.itemsize == C_sizeof_field_desc()
    loc = [(k // 2, k // 2, (k & 1) ^ 1, k) for k in range(nf)]
    ctxs = []
    for q in range(inflight):
        sm = ntscsim.FieldSimulator(params=params, device=local_rank)
        d = torch.zeros((n_frames, h, w, 4), dtype=torch.uint8, device=dev)
        arr = sm.build_descs(src, d, loc)                 # rng_pos: AUTO (continue after the previous descriptor)
        view = np.frombuffer(arr, dtype=dt)
        st = torch.cuda.Stream(dev)
        ctxs.append((sm, d, arr, view, st, [None, None]))
    k_idx = np.arange(nf, dtype=np.uint64)

    def run_step(s):
        sm, d, arr, view, st, evs = ctxs[s % inflight]
        ev = evs[(s // inflight) & 1]
        if ev is not None:
            ev.synchronize()                              # at most two steps queued per context
        view["fieldno"] = np.uint64(s * nf) + k_idx
        view["rng_pos"][0] = s * draws_per_step           # explicit for the first field, the rest follow it
        sm.run_descs(arr, w, h, stream=st.cuda_stream)
        e = torch.cuda.Event()
        e.record(st)
        evs[(s // inflight) & 1] = e

    def worker(j, first, count, bar):
        torch.cuda.set_device(local_rank)
        bar.wait()
        for s in range(first, first + count):
            if (s % inflight) % threads == j:
                run_step(s)

    def timed(first, count):
        bar = threading.Barrier(threads + 1)
        th = [threading.Thread(target=worker, args=(j, first, count, bar)) for j in range(threads)]
        for t in th:
            t.start()
        torch.cuda.synchronize(dev)
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    timed(0, 2 * inflight)                                # first-call allocations
    el = timed(2 * inflight, steps)
    last = 2 * inflight + steps - 1
    got = ctxs[last % inflight][1]
    chk = ntscsim.FieldSimulator(params=params, device=local_rank)
    d2 = torch.zeros_like(got)
    pos, rp = last * draws_per_step, []
    for k in range(nf):
        rp.append(pos)
        pos += calls[(k & 1) ^ 1]
    chk.fields(src, d2, [(k // 2, k // 2, (k & 1) ^ 1, last * nf + k) for k in range(nf)], rng_pos=rp)
    chk.sync()
    ok = bool(torch.equal(got, d2))
    chk.close()
    for sm, *_ in ctxs:
        sm.close()
    return steps * nf / el, ok


def C_sizeof_field_desc():
    import ctypes
    import ntscsim
    return ctypes.sizeof(ntscsim.FieldDesc)


def variant_contexts(torch, ntscsim, dev, local_rank, args, nq):
    """The 8-bit YUV422P tool (ffmpeg_to_composite): nq contexts, each with 2 x frames colour-bars
    YUV422P frames resident in HBM (every field its own frame, processed in place), its descriptor
    array and its stream.  Returns (simulators, step(i), frames of context 0)."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs as L
    w, h = args.width, args.height
    p422 = ntscsim.make_params_to_composite(args.preset.split())
    lib = ntscsim.lib()
    base = L.yuv_bars(w, h, 0, pad=16)
    nf = 2 * args.frames
    sims, arrs, streams, frames = [], [], [], []
    for q in range(nq):
        sm = ntscsim.FieldSimulator(params=p422, device=local_rank)
        fr = [[torch.from_numpy(base.plane(i).copy()).to(dev) for i in range(3)] for _ in range(nf)]
        jobs, pos = [], 0
        for k in range(nf):
            field = (k & 1) ^ 1
            jobs.append({"dst": fr[k], "field": field, "fieldno": k, "rng_pos": pos})
            pos += lib.ntscsim_rng_calls_per_field_422(C.byref(p422), w, h, field)
        sims.append(sm); arrs.append(sm.build_descs422(jobs)); frames.append(fr)
        streams.append(torch.cuda.Stream(dev))
    plans = [sm.prepare422(a, w, h) for sm, a in zip(sims, arrs)]       # prepared batches: a step is only the launches

    def vstep(i):
        q = i % nq
        sims[q].run_prepared422(plans[q], stream=streams[q].cuda_stream)
    vstep.keep = (arrs, frames, streams, plans)
    return sims, vstep, frames[0]


def main_to_composite(args):
    """bench.py --tool to_composite: the same contract for the YUV422P sibling tool
    (ffmpeg_to_composite.cpp:629-952 composite_video_process, one call per field)."""
    import numpy as np
    import torch
    import ntscsim
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    w, h = args.width, args.height
    nf = 2 * args.frames
    nq = max(1, args.inflight)
    sims, vstep, frames0 = variant_contexts(torch, ntscsim, dev, local_rank, args, nq)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for i in range(nq):          # first-call allocations of every context
        vstep(i)
    torch.cuda.synchronize(dev)
    # the sustained leg first (see the primary tool's loop below: clocks of a busy GPU, not of one leaving idle)
    sustained = None
    if args.sustain_seconds > 0:
        n_s, t1 = 0, time.perf_counter()
        while True:
            for i in range(4 * nq):
                vstep(n_s + i)
            n_s += 4 * nq
            torch.cuda.synchronize(dev)
            if time.perf_counter() - t1 >= args.sustain_seconds:
                break
        sustained = (n_s, time.perf_counter() - t1)
    for i in range(args.warmup):
        vstep(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        vstep(i)
    fence()
    elapsed = time.perf_counter() - t0
    # kernel time: hipEvents around the kernels of un-shared launches on one context
    sims[0].set_profiling(True)
    for _ in range(5):
        vstep(0)
    torch.cuda.synchronize(dev)
    tm = sims[0].timings_ms()
    sims[0].set_profiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        import ctypes as C
        import _libs as L
        calls = max(1, tm["calls"])
        k_ms, set_ms = tm["decode"] / calls, tm["setup"] / calls
        rows = (ntscsim.field_rows(h, 0) + ntscsim.field_rows(h, 1)) / 2.0
        alg = 4.0 * w * rows * nf            # 2 B/pixel read + 2 B/pixel written, rows of the field
        value = world * nf * args.steps / elapsed
        traffic, valu = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            te = tj.get("%dx%d %s to_composite" % (w, h, args.preset), {})
            scale = nf / float(te.get("fields_per_launch", nf))
            traffic = te.get("k422_hbm_bytes_per_launch") and te["k422_hbm_bytes_per_launch"] * scale
            if te.get("k422_mean_cycles_per_inst"):
                need = te["k422_wave_insts_per_launch"] * scale * te["k422_mean_cycles_per_inst"]
                peak = 1024 * 2.4e9
                valu = {"bound": "valu-issue (cycle-weighted)", "unit": "SIMD pipe cycles/s", "peak": peak,
                        "pipe_cycles_per_step": need, "mean_cycles_per_inst": te["k422_mean_cycles_per_inst"],
                        "path_frac": need / (elapsed / args.steps) / peak,
                        "path_frac_nominal": (te["k422_wave_insts_per_launch"] * scale * te["k422_mean_cycles_per_inst_nominal"] /
                                              (elapsed / args.steps) / peak) if te.get("k422_mean_cycles_per_inst_nominal") else None,
                        "kernel_frac": need / (k_ms * 1e-3) / peak if k_ms else None,
                        "note": "SQ_INSTS_VALU of k422_fused per launch (profiles/*_pmc_summary_to_composite.txt) x the "
                                "mean issue cost of its loops' instruction mix (tools/loop_census.py --mean; path_frac at "
                                "the probe's slowest-wave costs 4.3 / 2.7 cycles, path_frac_nominal at the pipe's "
                                "nominal 4 / 2); the setup kernels are left out of `need`"}
        except Exception:
            pass
        out = {
            "metric": "frames/sec (ffmpeg_to_composite: output frames = fields; %dx%d YUV422P, preset '%s'; "
                      "steady-state pipelined throughput, %d steps in flight)" % (w, h, args.preset if args.preset.strip() else "default", nq),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d YUV422P colour-bars frames, preset '%s': %d fields per GPU per step, every "
                                   "field its own frame, processed in place (composite_video_process per field)"
                                   % (w, h, args.preset, nf),
                       "tool": "to_composite", "steps_in_flight": nq,
                       "pre_roll": None if sustained is None else
                           {"steps": sustained[0], "seconds": sustained[1],
                            "note": "untimed steps of the same work before the W warm-up steps (= the value_sustained leg)"}},
            "roofline": {"bound": "hbm", "kernel": "k422_fused", "achieved": alg / (k_ms * 1e-3) / 1e9 if k_ms else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms else 0.0,
                         "traffic": traffic, "algorithmic_bytes_per_launch": alg, "kernel_ms": k_ms,
                         "kernel_ms_all": {"setup": set_ms, "process": k_ms}, "valu": valu,
                         "kernel_forms": sims[0].last_kernels(),
                         "note": "4*W*L algorithmic bytes per field; like the BGRA tool the kernel is bound by "
                                 "dependent fp64 filter chains, not by HBM (DESIGN.md section 7)"},
        }
        if sustained is not None:
            out["value_sustained"] = world * nf * sustained[0] / sustained[1]
        if args.cpu_fields > 0:
            p422 = ntscsim.make_params_to_composite(args.preset.split())
            ncpu = min(args.cpu_fields, nf)
            have_ref = L.have_tocomp_ref()
            # parity spot check: 4 fresh fields through HIP and through the CPU engine
            o = L.TocompOracleStream(p422, L.OOB_MEMORY)
            sm = ntscsim.FieldSimulator(params=p422, device=local_rank)
            ok = 0
            for k in range(4):
                fr = L.yuv_bars(w, h, k, pad=16)
                d = [torch.from_numpy(fr.plane(i).copy()).to(dev) for i in range(3)]
                sm.fields422([{"dst": d, "field": (k & 1) ^ 1, "fieldno": k}], w, h)
                sm.sync()
                o.process(fr, (k & 1) ^ 1, k)
                for i in range(3):
                    # (the frame's last row reads past the plane in the reference: excluded)
                    if not np.array_equal(d[i].cpu().numpy()[:h - 1, :fr.pix(i).shape[1]], fr.pix(i)[:h - 1]):
                        raise AssertionError("bench: HIP output of field %d differs from the oracle" % k)
                ok += 1
            sm.close()
            eng = L.TocompRefStream(p422) if have_ref else L.TocompOracleStream(p422, L.OOB_MEMORY)
            fr = L.yuv_bars(w, h, 0, pad=16)
            t0 = time.perf_counter()
            for k in range(ncpu):
                eng.process(fr, (k & 1) ^ 1, k)
            cpu_fps = ncpu / (time.perf_counter() - t0)
            out["cpu_baseline"] = {"value": cpu_fps, "unit": "frames/s", "cores": 1,
                                   "kind": "reference" if have_ref else "port",
                                   "sample": "%d fields of one 720x486 frame processed in place, single thread; %s; "
                                             "%d fresh fields compared byte-for-byte with the HIP output first"
                                             % (ncpu, "composite_video_process() of the reference (oracle/_ref)"
                                                if have_ref else "oracle/tocomp_oracle.c", ok)}
            out["speedup_vs_cpu_1core"] = value / cpu_fps
        print(json.dumps(out), flush=True)
    for sm in sims:
        sm.close()
    if dist is not None:
        dist.destroy_process_group()


def extras(torch, ntscsim, dev, local_rank, args):
    """The numbers README / DESIGN quote beside the headline value (N = 1, rank 0)."""
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _libs as L
    out = {}
    w, h = args.width, args.height
    # ---- PCIe-inclusive: ntscsim_frames_host, 300 host frames in -> 600 bob frames out
    n = args.frames
    params = ntscsim.make_params(args.preset.split())
    one = L.bars(w, h, 0)
    src_pin = torch.empty((n, h, w, 4), dtype=torch.uint8).pin_memory()
    for j in range(n):
        src_pin[j] = torch.from_numpy(np.roll(one, -j, axis=1))
    dst_pin = torch.empty((2 * n, h, w, 4), dtype=torch.uint8).pin_memory()
    fb = w * h + 2 * (w // 2) * ((h + 1) // 2)
    yuv_pin = torch.empty((2 * n, fb), dtype=torch.uint8).pin_memory()
    src_pg, dst_pg = src_pin.numpy().copy(), np.zeros((2 * n, h, w, 4), np.uint8)
    sim = ntscsim.FieldSimulator(params=params, device=local_rank)
    sim.frames_host(dst_pin.numpy()[:8], src_pin.numpy()[:4])
    e2e = {}
    for name, d, s_, kw in (("bgra_pinned", dst_pin.numpy(), src_pin.numpy(), {}),
                            ("bgra_pageable", dst_pg, src_pg, {}),
                            ("yuv420p_pinned", yuv_pin.numpy(), src_pin.numpy(), {"yuv": "420"})):
        best = 0.0
        for _ in range(2):
            sim.rng_pos = 0
            t0 = time.perf_counter()
            sim.frames_host(d, s_, first_fieldno=0, chunk_frames=32, **kw)
            best = max(best, 2 * n / (time.perf_counter() - t0))
        e2e[name] = best
    # ---- YUV420P in -> YUV420P out: 1.5 bytes per pixel each way over the link (the decoder's and the
    # encoder's pixel format; both conversions on the GPU)
    try:
        from ntscsim import _capi as _c
        hs = _c.HostSource()
        cw, chh = w // 2, (h + 1) // 2
        hs.format, hs.width, hs.height, hs.frame_bytes = _c.SRC_YUV420P, w, h, fb
        for k_, (ls_, off_) in enumerate(((w, 0), (cw, w * h), (cw, w * h + cw * chh))):
            hs.linesize[k_], hs.plane_offset[k_] = ls_, off_
        yin = torch.empty((n, (fb + 15) // 16 * 16), dtype=torch.uint8).pin_memory()
        ybars = L.yuv_bars(w, h, 0)
        yin_np = yin.numpy()
        for j in range(n):          # Y | U(4:2:0) | V(4:2:0) of the colour-bars frame rotated by j
            yin_np[j, :w * h] = np.roll(ybars.pix(0), -j, axis=1).reshape(-1)
            yin_np[j, w * h:w * h + cw * chh] = np.roll(ybars.pix(1)[::2], -(j // 2), axis=1).reshape(-1)
            yin_np[j, w * h + cw * chh:fb] = np.roll(ybars.pix(2)[::2], -(j // 2), axis=1).reshape(-1)
        best = 0.0
        for _ in range(2):
            sim.rng_pos = 0
            t0 = time.perf_counter()
            sim.frames_host_scaled(yuv_pin.numpy(), yin_np[:, :fb], hs, w, h, first_fieldno=0, chunk_frames=32, yuv="420")
            best = max(best, 2 * n / (time.perf_counter() - t0))
        e2e["yuv420p_in_yuv420p_out_pinned"] = best
        del yin
    except Exception as e:
        e2e["yuv420p_in_error"] = repr(e)
    # ---- the 1:1 drop-in: one composite_layer() call per ntscsim_field() call, host frames in and out
    one_dst = np.zeros((h, w, 4), np.uint8)
    sim.rng_pos = 0
    for k in range(4):
        sim.field_host(one_dst, one, (k & 1) ^ 1, k)
    t0 = time.perf_counter()
    nfc = 200
    for k in range(nfc):
        sim.field_host(one_dst, one, (k & 1) ^ 1, k)
    e2e["field_call"] = nfc / (time.perf_counter() - t0)
    sim.close()
    e2e["unit"] = "frames/s"
    e2e["note"] = ("ntscsim_frames_host: %d host frames in, %d bob frames out through H2D | kernels | D2H "
                   "on three streams, chunks of 32 frames; pageable = the call pins the caller's buffers "
                   "in place first; yuv420p = the encoder's pixel format made on the GPU (1.5 B/pixel "
                   "back instead of 4); field_call = ntscsim_field(), the synchronous one-field-per-call drop-in (its asynchronous form: field_submit) "
                   "for composite_layer() on pageable host frames (upload, three kernels on 4 wavefronts, download)" % (n, 2 * n))
    # ---- the ffmpeg_ntsc-compatible command line host (synthetic source, discarded output)
    cli = os.path.join(ROOT, "composite-video-simulator_amd", "ntsc_cli")
    if os.path.exists(cli) and (w, h) == (720, 486):
        import re
        import subprocess
        best = 0.0
        for _ in range(2):
            pr = subprocess.run([cli] + args.preset.split() + ["-i", "bars:3000", "-o", "null:"],
                                stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=120)
            m = re.search(r"\(([0-9.]+) fields/s incl", pr.stderr.decode(errors="replace"))
            if m:
                best = max(best, float(m.group(1)))
        e2e["cli"] = best
        e2e["cli_note"] = ("ntsc_cli %s -i bars:3000 -o null: (6000 fields; the tool's own figure for its field loop: host frame "
                           "synthesis, upload, kernels, download; one-off initialisation is outside its clock; best "
                           "of 2 runs)" % args.preset)
    # ---- the asynchronous 1:1 drop-in: ntscsim_submit() / ntscsim_wait() from the reference-shaped loop of
    # host/field_loop.cpp (AVFrame-shaped pageable frames; the call at ffmpeg_ntsc.cpp:2229 replaced, the frame
    # consumed 4 * depth fields later)
    floop = os.path.join(ROOT, "composite-video-simulator_amd", "field_loop")
    if os.path.exists(floop):
        import json as _json
        import subprocess
        def run_loop(*extra):
            best = None
            for _ in range(2):
                pr = subprocess.run([floop] + args.preset.split() + ["--height", str(h), "-width", str(w)] + list(extra),
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=180)
                try:
                    r = _json.loads(pr.stdout.decode().strip().splitlines()[-1])
                except Exception:
                    return {"error": pr.stderr.decode(errors="replace")[-300:]}
                if best is None or r["fields_per_s"] > best["fields_per_s"]:
                    best = r
            return best
        big = ["--fields", "20000", "--warmup", "2000"]
        sync = run_loop("--mode", "sync", "--fields", "1500", "--warmup", "100")
        sub = run_loop("--mode", "submit", "--depth", "32", "--rewrite-src", "1", *big)
        e2e["field_submit"] = sub.get("fields_per_s", 0.0)
        e2e["field_submit_detail"] = {
            "loop_sync_fields_per_s": sync.get("fields_per_s"),
            "depth32_in_rgb_rewritten": sub,
            "depth32_decoder_frames": run_loop("--mode", "submit", "--depth", "32", *big),
            "depth32_bob": run_loop("--mode", "submit", "--depth", "32", "--bob", "1", *big),
            "depth128": run_loop("--mode", "submit", "--depth", "128", *big),
            "depth32_src_stable": run_loop("--mode", "submit", "--depth", "32", "--src-stable", "1", *big),
            "depth32_staging_ring": run_loop("--mode", "submit", "--depth", "32", "--pin", "0", "--rewrite-src", "1", *big),
            "note": "host/field_loop.cpp: the loop of ffmpeg_ntsc.cpp:2202-2282 on AVFrame-shaped pageable frames "
                    "(posix_memalign, linesize rounded to 64) with composite_layer() :2229 replaced by "
                    "ntscsim_submit_avframe() and the frame consumed behind ntscsim_wait() 4 * depth fields later; "
                    "field_submit = depth 32, ONE source frame (in.rgb) rewritten by a memcpy for every new frame "
                    "(the stand-in for sws_scale :603), snapshot semantics (submit returns after the DMA read "
                    "it); decoder_frames = the source is re-pointed at one of 8 frames instead (no host copy); "
                    "src_stable = the caller promises not to touch the source until the wait; staging_ring = "
                    "pin_caller_buffers 0 (one host memcpy each way); loop_sync = the same loop with "
                    "ntscsim_field_avframe() (= field_call from C++)"}
    out["end_to_end"] = e2e
    del src_pin, dst_pin, yuv_pin, src_pg, dst_pg
    # ---- the 8-bit YUV422P tool (ffmpeg_to_composite), 600 fields, every field its own frame
    nf = 2 * args.frames
    nq = max(1, args.inflight)
    sims, vstep, _ = variant_contexts(torch, ntscsim, dev, local_rank, args, nq)
    for i in range(4 * nq):
        vstep(i)
    dt = time_steps(torch, dev, vstep, 12 * nq)
    for sm in sims:
        sm.close()
    out["variant422"] = {"value": nf / dt, "unit": "frames/s", "ms_per_step": dt * 1e3,
                         "workload": "%dx%d YUV422P, preset '%s', %d fields per step (every field its own frame, "
                                     "processed in place), %d steps in flight" % (w, h, args.preset, nf, nq)}
    # ---- the raw-composite decoder (ffmpeg_raw28ntsc): a synthetic 8 x fsc capture resident in HBM
    try:
        # 600 fields (a 10 s capture) = a 30-field synthetic capture repeated 20 times
        nfr = 600
        base = L.raw28_capture(30, 5, 3, 0)
        capture = np.ascontiguousarray(np.tile(base[:30 * 477750], 20)[250000:])
        dec = ntscsim.Raw28Decoder([], device=local_rank)
        cap_dev = torch.from_numpy(capture).to(dev)
        fr = torch.empty((nfr + 2, dec.height, dec.width * 4), dtype=torch.uint8, device=dev)
        nout = dec.decode(cap_dev, fr)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            dec.decode(cap_dev, fr)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps
        st = dec.stats()
        # CPU beside it: the reference text (oracle/_ref) or the port, first 12 fields of the same capture
        sub = np.ascontiguousarray(capture[:14 * 477750])
        t0 = time.perf_counter()
        if L.have_raw28_ref():
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                ref_frames, _ = L.raw28_ref_run(L.raw28_oracle_opts(), sub, os.path.join(td, "cap.u8"))
            kind = "reference"
        else:
            ref_frames, _ = L.raw28_oracle_run(L.raw28_oracle_opts(), sub)
            kind = "port"
        cpu_dt = time.perf_counter() - t0
        ncmp = min(8, ref_frames.shape[0])
        same = bool(np.array_equal(fr[:ncmp].cpu().numpy().reshape(ncmp, dec.height, -1), ref_frames[:ncmp]))
        out["raw28"] = {"value": nout / dt, "unit": "fields/s", "ms_per_capture": dt * 1e3, "fields": nout,
                        "workload": "ffmpeg_raw28ntsc decoder: synthetic %d-field capture at 8 x fsc (%.0f MB, 8 bit) "
                                    "resident in HBM -> %d grey BGRA frames of %dx%d; whole call incl. the host's "
                                    "sync walk" % (nfr, capture.size / 1e6, nout, dec.width, dec.height),
                        "stats": st,
                        "cpu_1core": {"value": ref_frames.shape[0] / cpu_dt, "kind": kind,
                                      "sample": "%d fields of the same capture incl. the tool's start-up filter run" % ref_frames.shape[0]},
                        "first_%d_fields_equal_cpu" % ncmp: same}
        dec.close()
        del cap_dev, fr
    except Exception as e:
        out["raw28_error"] = repr(e)
    # ---- a device-resident stream of FRESH batches (the headline replays prepared ones)
    try:
        best, ok_all = 0.0, True
        for _ in range(2):
            v_, ok_ = device_stream_rate(torch, ntscsim, dev, local_rank, params, w, h, args.frames, 48, args.inflight, threads=2)
            best, ok_all = max(best, v_), ok_all and ok_
        out["device_stream"] = {
            "value": best, "unit": "frames/s", "verified_last_step": ok_all,
            "workload": "%dx%d, preset '%s': every step is the NEXT %d fields of one long stream (new fieldno and "
                        "rand() position per field) through ntscsim_fields_device() -- descriptor validation, rand() "
                        "window derivation, record upload and the kernel chain inside the clock; %d contexts, 2 host "
                        "threads; 48 steps, best of 2" % (w, h, args.preset, 2 * args.frames, args.inflight)}
    except Exception as e:
        out["device_stream"] = {"error": repr(e)}
    # ---- other sizes / presets on the BGRA path
    out["sizes"] = {
        "1920x1080": {"value": device_rate(torch, ntscsim, dev, local_rank, args.preset.split(), 1920, 1080, 136, 8, args.inflight),
                      "unit": "frames/s", "workload": "preset '%s', 272 fields (146,880 scanlines) per step, %d steps in flight" % (args.preset, args.inflight)},
        "3840x2160": {"value": device_rate(torch, ntscsim, dev, local_rank, args.preset.split(), 3840, 2160, 68, 8, args.inflight),
                      "unit": "frames/s", "workload": "preset '%s', 136 fields (146,880 scanlines) per step, %d steps in flight" % (args.preset, args.inflight)},
    }
    out["presets"] = {
        "default": {"value": device_rate(torch, ntscsim, dev, local_rank, [], w, h, args.frames, 24, args.inflight),
                    "unit": "frames/s", "workload": "%dx%d, default preset (BASELINE configs[0] on the GPU), %d fields per step" % (w, h, 2 * args.frames)},
    }
    # the headline preset measured the way the legs below are (24 steps after one per context, no pre-roll): the
    # reference point of their `frac_of_preset`
    ref_rate = device_rate(torch, ntscsim, dev, local_rank, args.preset.split(), w, h, args.frames, 24, args.inflight)
    out["presets"]["preset_same_method"] = {"value": ref_rate, "unit": "frames/s",
                                            "workload": "%dx%d, preset '%s', %d fields per step, 24 steps" % (w, h, args.preset, 2 * args.frames)}
    # switch sets that fall off the hand-tuned kernels' preconditions (the GENERIC / template forms run):
    # which decoder form each one took is recorded beside its rate
    for name, fl in (("vhs_catv2", ["-vhs", "-comp-catv2"]), ("vhs_phase90", ["-vhs", "-comp-phase", "90"]),
                     ("vhs_svideo", ["-vhs", "-vhs-svideo", "1"]),
                     ("vhs_full_outlp", ["-vhs", "-out-composite-lowpass-lite", "0"]),
                     ("vhs_ghost2", None)):
        try:
            kn = []
            if fl is None:          # extension (absent from the reference, parity unpinned): two echo taps
                prm = ntscsim.make_params(["-vhs"])
                prm.ghost_taps = 2
                prm.ghost_delay[0], prm.ghost_delay[1] = 12, 31
                prm.ghost_gain[0], prm.ghost_gain[1] = 64, -32
                v_ = device_rate(torch, ntscsim, dev, local_rank, None, w, h, args.frames, 24, args.inflight, params=prm, kernels=kn)
                what = "-vhs + ghosting extension (2 taps: 12 samples x 64/256, 31 samples x -32/256; absent from the reference, parity unpinned)"
            else:
                v_ = device_rate(torch, ntscsim, dev, local_rank, fl, w, h, args.frames, 24, args.inflight, kernels=kn)
                what = "preset '%s'" % " ".join(fl)
            out["presets"][name] = {"value": v_, "unit": "frames/s", "frac_of_preset": v_ / ref_rate if ref_rate else None,
                                    "kernels": [k_ for k_ in kn if not k_.startswith(("k_field", "k_row"))],
                                    "workload": "%dx%d, %s, %d fields per step" % (w, h, what, 2 * args.frames)}
        except Exception as e:
            out["presets"][name] = {"error": repr(e)}
    return out


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
    ap.add_argument("--mode", default="exact", choices=["exact", "fast32"],
                    help="exact = bit-identical to the reference (fp64, default); fast32 = fp32 "
                         "filters within the tolerance of tests/test_gpu_fast_mode.py")
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
            if args.mode == "fast32":
                sm.set_mode(ntscsim._capi.MODE_FAST32)
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
    for sm, plans, _, _ in ctxs:
        for pl in plans:
            sm.free_prepared(pl)
        sm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
