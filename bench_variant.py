"""bench_variant.py -- `bench.py --tool to_composite`: the bench contract for the 8-bit YUV422P sibling tool
(ffmpeg_to_composite.cpp:629-952 composite_video_process, one call per field), with its own roofline object, CPU
baseline and `presets` legs.  Split out of bench.py in round 5 (bench hygiene)."""
import json
import os
import sys
import time

from bench_side import HBM_PEAK_GBS, ROOT, emit, variant_contexts


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
    out = None
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
        if world == 1 and not args.no_extras:
            # the switch-set families beside the -vhs one, measured the same way: 24 steps after one
            # per context, no pre-roll; which kernel form each took is recorded beside its rate
            import copy
            legs = {}
            for name, fl in (("preset_same_method", args.preset.split()), ("default", []),
                             ("vhs_svideo", ["-vhs", "-vhs-svideo", "1"]), ("vhs_ep", ["-vhs", "-vhs-speed", "ep"]),
                             ("yc_recomb1", ["-yc-recomb", "1"])):
                try:
                    a2 = copy.copy(args)
                    a2.preset = " ".join(fl)
                    sims2, vstep2, _ = variant_contexts(torch, ntscsim, dev, local_rank, a2, nq)
                    for i in range(nq):
                        vstep2(i)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    for i in range(24):
                        vstep2(i)
                    torch.cuda.synchronize(dev)
                    dt = time.perf_counter() - t1
                    legs[name] = {"value": nf * 24 / dt, "unit": "frames/s",
                                  "kernels": [k_ for k_ in sims2[0].last_kernels() if k_.startswith("k422")],
                                  "workload": "%dx%d YUV422P, preset '%s', %d fields per step" % (w, h, a2.preset or "default", nf)}
                    for sm2, pl2 in zip(sims2, vstep2.keep[3]):
                        sm2.free_prepared422(pl2)
                        sm2.close()
                    del vstep2, sims2
                    torch.cuda.empty_cache()
                except Exception as e:
                    legs[name] = {"error": repr(e)}
            ref = legs.get("preset_same_method", {}).get("value")
            for v_ in legs.values():
                if ref and "value" in v_:
                    v_["frac_of_preset"] = v_["value"] / ref
            out["presets"] = legs
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
    for sm in sims:
        sm.close()
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and out is not None:
        emit(out, "bench_extras_to_composite.json")
