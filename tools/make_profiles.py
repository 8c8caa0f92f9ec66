#!/usr/bin/env python3
"""Assembles profiles/ for one measurement round from what the GPU runs left in gpurun_out/ and
from the ISA census of the shipped kernels, and regenerates profiles/traffic.json (read by
bench.py for `roofline.traffic` / `roofline.valu`) and profiles/README.md.

  python tools/make_profiles.py r02 <bench.json> <kstats default dir> <kstats inflight1 dir> \
         <pmc dir> <valu_rates.txt> <chain_probe.txt> <census dir> [bench_fast32.json]

census dir = JSON files written by tools/isa_cost.py --json for k_decode_fast<true,double>,
k_encode_fast<double>, k_row_states, k_field_setup (see tools/refresh_profiles.sh).
"""
import csv
import json
import os
import shutil
import sys

tag, bench, kdef, kif1, pmc, rates, chain, census = sys.argv[1:9]
fast = sys.argv[9] if len(sys.argv) > 9 else None
tocomp = sys.argv[10:13] if len(sys.argv) > 12 else None      # bench json, kstats dir, pmc dir of the YUV422P tool
P = "profiles"
os.makedirs(P, exist_ok=True)
shutil.copy(os.path.join(kdef, "kernel_stats.csv"), "%s/%s_kernel_stats_default_cmd.csv" % (P, tag))
shutil.copy(os.path.join(kif1, "kernel_stats.csv"), "%s/%s_kernel_stats_inflight1.csv" % (P, tag))
shutil.copy(os.path.join(pmc, "summary.txt"), "%s/%s_pmc_summary.txt" % (P, tag))
shutil.copy(rates, "%s/%s_valu_rates.txt" % (P, tag))
shutil.copy(chain, "%s/%s_chain_probe.txt" % (P, tag))
shutil.copy(bench, "%s/%s_bench.json" % (P, tag))
if fast:
    shutil.copy(fast, "%s/%s_bench_fast32.json" % (P, tag))

KERN = {"k_decode": "k_decode_fastILb1EdLb0EE", "k_encode": "k_encode_fastIdE",
        "k_row_states": "k_row_states", "k_field_setup": "k_field_setup"}
cen = {}
for k, f in KERN.items():
    j = json.load(open(os.path.join(census, f + ".json")))
    cen[k] = {x: j[x] for x in ("kernel", "steps_per_pass", "valu_per_step", "fp64_per_step",
                                "half_rate_int_per_step", "full_rate_per_step", "valu_pipe_cycles_per_step",
                                "mean_cycles_per_valu", "mean_cycles_per_valu_nominal", "salu_per_step", "lds_per_step",
                                "vmem_per_step")}
json.dump({"tool": "tools/isa_cost.py on hipcc -S of csrc/ntscsim_hip.hip (hottest loop of each kernel)",
           "issue_cost_cycles": {"half_rate (fp64, cvt, v_cndmask, v_lshl*, v_add3, v_med3, v_mul_*, DPP)": 4.3,
                                 "full_rate (v_add/sub_u32, and/or/xor, shift right, v_mov)": 2.7,
                                 "nominal (mean_cycles_per_valu_nominal)": "4 / 2 cycles, MI355X_MICROARCH.md",
                                 "source": "%s/%s_valu_rates.txt, slowest wave at 3 waves per SIMD" % (P, tag)},
           "kernels": cen}, open("%s/%s_isa_cost.json" % (P, tag), "w"), indent=1)


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        if "ntscsim" in r["Name"]:
            out[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
    return out


def pick(d, name):
    for k, v in d.items():
        if name in k:
            return v
    raise KeyError(name)


a, b = stats("%s/%s_kernel_stats_default_cmd.csv" % (P, tag)), stats("%s/%s_kernel_stats_inflight1.csv" % (P, tag))
d = json.load(open("%s/%s_bench.json" % (P, tag)))
pm, cur = {}, None
for l in open("%s/%s_pmc_summary.txt" % (P, tag)):
    if not l.startswith(" "):
        cur = l.strip()
    else:
        pm.setdefault(cur, {})[l.split()[0]] = float(l.split("mean=")[1])
dec = pick(pm, "k_decode_fast<true")
enc = pick(pm, "k_encode_fast")
rs, fs = pick(pm, "k_row_states"), pick(pm, "k_field_setup")
# FETCH_SIZE / WRITE_SIZE calibrated on this library's access shapes (tools/fetch_probe.hip, tools/fetch_calibrate.sh ->
# profiles/r04_fetch_calibration.txt): bytes actually moved = counter x factor
cal = {}
for l in open("%s/r04_fetch_calibration.txt" % P):
    if "known / counter" in l:
        cal[l.split()[0]] = float(l.split("=")[-1])
f_dec_rd, f_enc_rd = cal["k_read4_stream"], cal["k_read16_rows"]      # comp[x][row] loads | CoopLoader frame rows
f_dec_wr, f_enc_wr = cal["k_write16_rows"], cal["k_write4_stream"]    # cooperative pixel bursts | comp[x][row] stores
dec_b = (dec["FETCH_SIZE"] * f_dec_rd + dec["WRITE_SIZE"] * f_dec_wr) * 1024
enc_b = (enc["FETCH_SIZE"] * f_enc_rd + enc["WRITE_SIZE"] * f_enc_wr) * 1024
traffic = {"720x486 -vhs": {
    "fields_per_launch": 600,
    "k_decode_hbm_bytes_per_launch": dec_b,
    "k_decode_fetch_KiB_raw": dec["FETCH_SIZE"], "k_decode_write_KiB_raw": dec["WRITE_SIZE"],
    "k_encode_fetch_KiB_raw": enc["FETCH_SIZE"], "k_encode_write_KiB_raw": enc["WRITE_SIZE"],
    "calibration": {"k_decode_fetch": f_dec_rd, "k_decode_write": f_dec_wr, "k_encode_fetch": f_enc_rd, "k_encode_write": f_enc_wr,
                    "source": "profiles/r04_fetch_calibration.txt (tools/fetch_probe.hip: known bytes / counter per access shape)"},
    "k_encode_hbm_bytes_per_launch": enc_b,
    "path_hbm_bytes_per_launch": dec_b + enc_b,
    "path_over_algorithmic": (dec_b + enc_b) / (8.0 * 720 * 243 * 600),
    "valu": {
        "k_decode": {"wave_insts_per_launch": dec["SQ_INSTS_VALU"], "mean_cycles_per_inst": cen["k_decode"]["mean_cycles_per_valu"],
                     "mean_cycles_per_inst_nominal": cen["k_decode"]["mean_cycles_per_valu_nominal"]},
        "k_encode": {"wave_insts_per_launch": enc["SQ_INSTS_VALU"], "mean_cycles_per_inst": cen["k_encode"]["mean_cycles_per_valu"],
                     "mean_cycles_per_inst_nominal": cen["k_encode"]["mean_cycles_per_valu_nominal"]},
        "k_row_states": {"wave_insts_per_launch": rs["SQ_INSTS_VALU"], "mean_cycles_per_inst": cen["k_row_states"]["mean_cycles_per_valu"],
                     "mean_cycles_per_inst_nominal": cen["k_row_states"]["mean_cycles_per_valu_nominal"]},
        "k_field_setup": {"wave_insts_per_launch": fs["SQ_INSTS_VALU"], "mean_cycles_per_inst": cen["k_field_setup"]["mean_cycles_per_valu"],
                     "mean_cycles_per_inst_nominal": cen["k_field_setup"]["mean_cycles_per_valu_nominal"]},
    },
    "note": "rocprofv3 --pmc, separate passes (tools/pmc.sh), bench.py --inflight 1; raw FETCH_SIZE / WRITE_SIZE in "
            "KiB, bytes = counter x the factor measured for the kernel's access shape (a 4 B/lane streaming read is "
            "reported at half its bytes like the guide's 16 B/lane one; four-lanes-per-row 64-byte pieces at 1 / %.3f; "
            "writes at face value).  k_decode's fetch is more than its 420 MB composite plane because the VCR's luma path "
            "re-reads every sample 5 + d positions behind the chroma path and part of those reads miss the L2: all of "
            "them with plain stores of the output pixels (2.0 planes, round 3), about two thirds since the output and "
            "the composite plane are written with streaming stores and the re-read is marked nt "
            "(profiles/r04_nt_probe.txt; an LDS ring that removes the re-read was built and measured slower, "
            "csrc/ntsc_decode_fast.hip steady())" % f_enc_rd,
}}
tc = None
if tocomp and os.path.exists(tocomp[0]) and os.path.getsize(tocomp[0]) > 10:
    shutil.copy(tocomp[0], "%s/%s_bench_to_composite.json" % (P, tag))
    shutil.copy(os.path.join(tocomp[1], "kernel_stats.csv"), "%s/%s_kernel_stats_to_composite.csv" % (P, tag))
    shutil.copy(os.path.join(tocomp[2], "summary.txt"), "%s/%s_pmc_summary_to_composite.txt" % (P, tag))
    pm2, cur = {}, None
    for l in open(os.path.join(tocomp[2], "summary.txt")):
        if not l.startswith(" "):
            cur = l.strip()
        else:
            pm2.setdefault(cur, {})[l.split()[0]] = float(l.split("mean=")[1])
    k4 = pick(pm2, "k422_fused")
    tc = {"bench": json.load(open(tocomp[0])), "pmc": k4,
          "stats": stats("%s/%s_kernel_stats_to_composite.csv" % (P, tag))}
    traffic["720x486 -vhs to_composite"] = {
        "fields_per_launch": 600,
        "k422_hbm_bytes_per_launch": (k4["FETCH_SIZE"] + k4["WRITE_SIZE"]) * 1024,
        "k422_fetch_KiB": k4["FETCH_SIZE"], "k422_write_KiB": k4["WRITE_SIZE"],
        "k422_wave_insts_per_launch": k4["SQ_INSTS_VALU"],
        "k422_mean_cycles_per_inst": (json.load(open(os.path.join(census, "k422_fused.json")))["mean_cycles_per_valu"]
                                      if os.path.exists(os.path.join(census, "k422_fused.json")) else None),
        "k422_mean_cycles_per_inst_nominal": (json.load(open(os.path.join(census, "k422_fused.json"))).get("mean_cycles_per_valu_nominal")
                                              if os.path.exists(os.path.join(census, "k422_fused.json")) else None),
        "k422_kernel": None,
        "note": "tools/pmc422.sh (tools/variant_probe.py: one 600-field launch per call); FETCH/WRITE include the "
                "packed composite-byte plane sweep A hands to the streamed pass (1 B/pixel each way)"}
    traffic["720x486 -vhs to_composite"]["k422_kernel"] = [k for k in pm2 if "k422_fused" in k][0]
for k_ in traffic:
    if isinstance(traffic[k_], dict):
        traffic[k_]["round"] = tag        # which measurement round these counters come from (bench.py: roofline.traffic_source)
json.dump(traffic, open("%s/traffic.json" % P, "w"), indent=1)

v = d["roofline"].get("valu") or {}
e2e = d.get("end_to_end", {})
rd = "# profiles/ -- round %s (MI355X, gfx950, ROCm 7.2)\n\n" % tag[1:]
rd += "Everything here comes from `python bench.py` (BASELINE configs[1]: 720x486, 600 fields per step, `-vhs`) and the probes in `tools/`; `tools/refresh_profiles.sh` shows the commands, `tools/make_profiles.py` assembles this directory.  Files of earlier rounds (`r01_*` … `r03_*`) are kept for comparison.\n\n"
rd += "| file | command | what |\n|---|---|---|\n"
rd += "| `%s_bench_driver_cmd.json`, `%s_bench_line_default_cmd.json` | `python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command, nothing skipped); `python bench.py` | what bench.py PRINTS since round 6: the contract object alone (< 4 KB: contract keys, `roofline`, `cpu_baseline`, `side` = one number per side leg); everything else goes to `bench_extras.json` = the next row |\n" % (tag, tag)
rd += "| `%s_bench.json` | the same driver-command run | the FULL dictionary of that run (`bench_extras.json`): value, value_sustained, roofline (+ cycle-weighted `valu`, calibrated `traffic`), cpu_baseline, device_stream, end_to_end (incl. `field_submit`, `field_submit422`), multi_gpu_cpp_host, variant422, raw28, sizes, presets |\n" % tag
rd += "| `%s_kernel_stats_default_cmd.csv` | `rocprofv3 --kernel-trace --stats -- python bench.py --cpu-fields 0 --no-extras` | 4 steps in flight: kernels of different steps share the GPU, wall durations stretch; **Min** = un-shared duration |\n" % tag
rd += "| `%s_kernel_stats_inflight1.csv` | `... --inflight 1` | one step at a time: per-kernel durations without overlap |\n" % tag
rd += "| `%s_pmc_summary.txt` | `tools/pmc.sh` (4 separate `--pmc` passes, `--inflight 1`) | FETCH_SIZE, WRITE_SIZE, SQ counters per kernel (mean per launch) |\n" % tag
rd += "| `%s_valu_rates.txt` | `tools/valu_rate_probe.hip` | issue cost of every opcode class the kernels use, 1-4 waves per SIMD with forced placement |\n" % tag
rd += "| `%s_chain_probe.txt` | `tools/chain_probe.hip` | cost of dependent fp64 / int chains vs instruction-level parallelism |\n" % tag
rd += "| `%s_isa_cost.json` | `tools/isa_cost.py` | cycle-weighted instruction census of each kernel's steady loop |\n" % tag
rd += "| `%s_bench_to_composite.json`, `%s_kernel_stats_to_composite.csv`, `%s_pmc_summary_to_composite.txt` | `python bench.py --tool to_composite`, `tools/kstats.sh ... --tool to_composite --inflight 1`, `tools/pmc422.sh` | the same three for the YUV422P tool |\n" % (tag, tag, tag)
rd += "| `%s_kernel_stats_raw28.csv` | `rocprofv3 --kernel-trace --stats -- python tools/raw28_probe.py` | kernels of the raw-composite decoder on a 600-field capture (4 calls) |\n" % tag
rd += "| `%s_bench_float.json`, `%s_kernel_stats_float.csv`, `%s_float_err.txt`, `%s_float_pmc.txt` | `python bench.py --mode float`, `tools/kstats.sh ... --mode float --inflight 1`, `tools/float_err.py`, `tools/fp_probe.sh` + FETCH_SIZE / WRITE_SIZE passes | (round 6) NTSCSIM_MODE_FLOAT, the all-float pipeline: bench line, kernel durations, error against the oracle per input class (max, share of pixels / channels that differ, mean signed difference, histogram), stall counters of its decoder forms (one wave: variant 10; two-role workgroup: variant 0) and its HBM bytes |\n" % (tag, tag, tag, tag)
rd += "| `%s_sync_pipe.txt`, `%s_fuzz_pipe.txt` | `tools/sync_trace.sh`, `tools/fuzz_pipe.py` | (round 6) the synchronous call as a pipeline of wavefront roles (`k_field_pipe`, `k_field_pipe_tv`: DESIGN 1c): rate per preset and frame memory kind against the one-launch chain, rocprofv3 kernel / copy averages, the roles' clocks inside one call, `ntscsim_submit()` at depths 1-64 with and without the roles; the random sweep against the oracle |\n" % (tag, tag)
rd += "| `%s_pipe_census.txt` | `tools/pipe_census.py` (no GPU) | instruction census of the role kernels' loops by class (VALU f64 / int, SALU, LDS, VMEM, waits) from the built object: a lone wavefront's time per iteration is its total count x ~5.3 cycles |\n" % tag
rd += "| `%s_fuzz_long.txt` | the fuzz tools with larger counts | the last build's longer sweeps (role kernels 30,000 cases, host422 12,000 loops, submit 5,000 loops, the batch-size / concurrency probes) |\n" % tag
rd += "| `%s_role_probe.txt` | `tools/role_probe.py` | (round 6) the synchronous one-field call: shipped chain, two-launch decoder, encoder + VCR half + TV half launched side by side with no dependency (`NTSCSIM_ROLE_PROBE=1`: the ceiling of a stage-pipelined form), and the float pipeline |\n" % tag
rd += "| `r04_decode_census.txt`, `%s_loop_histograms.txt` | `tools/loop_census.py --hist` on `hipcc -S`; source accounting | where the VALU instructions of `k_decode_fast` go, stage by stage; what the round-4 diet removed; what was measured and not done (noise pre-pass, LDS luma ring, packed fp32) -- the kernel is unchanged since; opcode histograms of the steady loops of the hand-tuned decoder forms on this round's build |\n" % tag
rd += "| `r04_fetch_calibration.txt` | `tools/fetch_calibrate.sh` (`tools/fetch_probe.hip`) | FETCH_SIZE / WRITE_SIZE against known byte counts for this library's access shapes: the factors `traffic.json` applies |\n"
rd += "| `%s_submit_probe.txt` | `tools/submit_probe.sh` (`host/field_loop.cpp`, `tools/link_probe.hip`) | `ntscsim_submit()` / `ntscsim_wait()` against the synchronous call: byte identity (FNV-1a of every consumed frame), fields/s by depth / lanes / line doubling / source handling / delivery path / lag, GPU spans of consecutive launches, the host link's rates |\n" % tag
rd += "| `%s_dryrun_two_ranks.txt` | `tools/dryrun_two_ranks_one_gpu.sh` | `bench.py` with two ranks on the one GPU: RCCL refuses two ranks on one device, the same run with gloo verifies every rank's checksum |\n" % tag
rd += "| `r04_nt_probe.txt` | `tools/nt_probe.sh`, `tools/nt422_probe.sh` | A/B builds on one box: streaming (nt) stores / loads on the path's planes -- fields/s, kernel times, raw FETCH_SIZE / WRITE_SIZE per launch (shipped: output pixels, composite plane, the luma path's re-read; not shipped: the encoder's source loads, the YUV422P burst writer) |\n"
rd += "| `r04_raw28_sweep.txt`, `r04_raw28_noise.txt` | `tools/raw28_sweep_r04.sh`, `tools/raw28_noise_probe.py` (`tools/follow_guess_probe.c` for the CPU side) | the raw-composite decoder against the switches of its second sweep (exact scanlines behind the closed-form warm-up, chunks per wavefront, the fall-back when lanes are out of step, the run-based form that was dropped) and against the capture's noise level (links left to the repair rounds, warm-up lengths that close them) |\n"
rd += "| `%s_fuzz_sweep.txt` | `tools/fuzz_%s.sh` | one-off parity sweeps of this round's new code on the final build (round 6: 3,000 random loops through `ntscsim_field422()` / `ntscsim_submit422()` incl. tight rows chained on the device and staged delivery on the copy threads, 2,000 through `ntscsim_submit()`, 1,500 random switch sets / geometries in `NTSCSIM_MODE_FLOAT` with the census of decoder forms, 1,000 + 300 full-size switch sets of both tools, 200 raw captures): 0 failures; `r05_fuzz_sweep.txt`: round 5's |\n" % (tag, tag)
rd += "| `%s_ghost_probe.txt` | `tools/ghost_probe.py` (and with `NTSCSIM_DEBUG_DECODE=8`) | (round 5) the ghosting extension: rates of `-vhs` with 0 / 2 / 4 taps folded into the encoder (`k_encode_fast_gh`) against the same taps as a pass of `k_ghost`, and a delay too long to fold, one run on one box |\n" % tag
rd += "| `r04_fuzz_sweep.txt` | `tools/fuzz_r04.sh` | one-off parity sweeps on the final build (random switch sets of both tools, full size, the any-phase / full-output-low-pass / pre-emphasis / S-Video families at full size with the kernel forms listed; the raw-composite decoder on random captures / streams / speculation settings); the kernels those sweeps cover are unchanged since |\n"
rd += "| `r03_decode_experiments.txt`, `r03_clock_under_load.txt`, `r03_composite_range.txt`, `r03_variant_sweeps.txt` | (round 3) | A/B experiments on the dominant kernel; shader clock under load (2.31-2.32 GHz); value range of the composite plane; wave-clock share of the YUV422P kernel's sweeps |\n"
rd += "| `%s_raw28_front_pmc.txt` | `tools/pmc_raw28.sh`, `tools/follow_probe.hip` | counters of the raw-composite decoder's two front-end sweeps and the cost of one follower step for a lone wavefront |\n" % tag
rd += "| `%s_host422_loop_probe.txt` | `tools/host422_loop_probe.sh` (`host/field_loop422.cpp`) | (round 5) the YUV422P tool's loop on host frames, `ntscsim_field422()` / `ntscsim_submit422()`: byte identity (FNV-1a of every encoder frame) across sync / submit / staging rings / page-owned planes for six switch sets, fields/s by switch set, depth and frame allocation, host time inside the calls |\n" % tag
rd += "| `%s_sync_call_stats.csv`, `r05_sync_call_notes.txt` | `rocprofv3 --kernel-trace --memory-copy-trace --stats -- field_loop -vhs --mode sync` | where the ~0.5 ms of one synchronous `ntscsim_field()` call go, kernel by kernel; (round 5) the same loop alone and beside a process that keeps the GPU busy (same rate: not a clock effect) |\n" % tag
rd += "| `%s_rank_bench.json` | `rank_bench -vhs --spawn 1 --frames 300 --steps 40 --warmup 8` | (round 5) the C++ rank-per-GPU harness over `rccl.h` with the one rank this box has |\n" % tag
rd += "| `traffic.json` | derived (`tools/make_profiles.py`) | HBM bytes and VALU work per launch that `bench.py` turns into `roofline.traffic` / `roofline.valu` |\n\n"
rd += "## Bench line\n\n"
rd += "`value` = %.0f frames/s (fields/s; %d steps, %.3f ms per 600-field step), `value_sustained` = %.0f (the same step for %.2f s).  " % (
    d["value"], d["steps"], d["ms_per_step"], d.get("value_sustained", 0), d.get("sustained", {}).get("seconds", 0))
drv = "gpurun_out/bench_%s_driver_cmd.json" % tag
if os.path.exists(drv) and os.path.getsize(drv) > 10:
    dj = json.load(open(drv))
    rd += "With the driver's own window (`python bench.py --gpus 1 --steps 20 --warmup 5`, `%s_bench_driver_cmd.json`): %.0f frames/s -- 20 steps with four in flight include the pipeline's fill and drain.  " % (tag, dj["value"])
rd += ("`roofline.frac` = %.3f (k_decode, algorithmic HBM bytes / 8 TB/s; at most %.2f with the reference's fp64 arithmetic: "
       "`roofline.valu.hbm_frac_ceiling_exact_mode`); `roofline.valu.path_frac_nominal` = %.2f of the VALU issue capacity at the "
       "pipe's nominal 4 / 2 cycles per instruction, `path_frac` = %.2f at the probe's slowest-wave costs (cycle-weighted VALU "
       "issue is the bound that applies).  " % (d["roofline"]["frac"], v.get("hbm_frac_ceiling_exact_mode") or 0,
                                               v.get("path_frac_nominal") or 0, v.get("path_frac", 0)))
cb = d.get("cpu_baseline", {})
if cb:
    rd += "CPU beside it on the GPU box's host: the reference's own `composite_layer()` (`oracle/_ref`, 1 thread like the tool) %.1f fields/s => %.0fx; our port 1 core %.1f" % (
        cb["value"], d.get("speedup_vs_cpu_1core", 0), cb.get("port_1core", 0))
    if "port_all_cores" in cb:
        rd += "; our port on all %d usable CPUs %.0f fields/s => %.0fx" % (cb["port_all_cores"]["cores"], cb["port_all_cores"]["value"], d.get("speedup_vs_cpu_all_cores", 0))
    rd += ".\n\n"
if e2e:
    rd += "PCIe-inclusive (`end_to_end`, never `value`): BGRA out %.0f frames/s (pinned caller buffers), %.0f (pageable, pinned in place by the call), YUV420P out %.0f.  " % (
        e2e.get("bgra_pinned", 0), e2e.get("bgra_pageable", 0), e2e.get("yuv420p_pinned", 0))
    if e2e.get("field_call"):
        rd += "One field per `ntscsim_field()` call (the 1:1 drop-in, synchronous): %.0f fields/s from the C++ loop on `posix_memalign` frames, %.0f on pinned frames, %.0f through ctypes.  " % (e2e.get("field_call_cpp", 0), e2e.get("field_call_cpp_pinned", 0), e2e["field_call"])
    if e2e.get("field_submit"):
        fsd = e2e.get("field_submit_detail", {})
        rd += ("The same loop with `ntscsim_submit()` / `ntscsim_wait()` (`host/field_loop.cpp`, depth 32): %.0f fields/s with one source frame "
               "rewritten per decoded frame (`end_to_end.field_submit`: frames from ntscsim_host_frame_alloc), %.0f with the source re-pointed at decoded frames, %.0f with the line doubling delivered as well, %.0f at depth 128, "
               "%.0f with the frames in a pool declared by ntscsim_host_pin, %.0f with plain posix_memalign frames (staged: pinned rings + the engine's copy threads).  " % (
                   e2e["field_submit"], (fsd.get("depth32_decoder_frames") or {}).get("fields_per_s", 0),
                   (fsd.get("depth32_bob") or {}).get("fields_per_s", 0), (fsd.get("depth128") or {}).get("fields_per_s", 0),
                   (fsd.get("depth32_declared_pool") or {}).get("fields_per_s", 0), (fsd.get("depth32_malloc_frames_staged") or {}).get("fields_per_s", 0)))
    if e2e.get("cli"):
        rd += "`ntsc_cli -vhs -i bars:3000 -o null:` %.0f fields/s (`end_to_end.cli`).  " % e2e["cli"]
if e2e.get("field_submit422"):
    f4 = e2e.get("field_submit422_detail", {})
    g = lambda k_: (f4.get(k_) or {}).get("fields_per_s", 0) if isinstance(f4.get(k_), dict) else (f4.get(k_) or 0)
    rd += ("\n\nThe YUV422P tool's loop on host frames (`host/field_loop422.cpp`, 720x480, depth 32; `end_to_end.field_submit422*`): %.0f fields/s with `-vhs` "
           "(planes from ntscsim_host_frame_alloc: pinned by construction), %.0f with a declared pool, %.0f with plain posix_memalign planes and no mallopt (staging rings + copy threads), "
           "%.0f with `-vhs -422`, %.0f with the default preset, %.0f with tight rows (704 wide: batched, pad bytes chained on the device), %.0f for the synchronous `ntscsim_field422()`.  " % (
               e2e["field_submit422"], g("depth32_vhs_declared_pool"), g("depth32_vhs_heap_planes"), g("depth32_vhs_422_pinned"), g("depth32_default_preset"),
               g("tight_rows_704"), g("loop_sync_fields_per_s")))
mg = d.get("multi_gpu_cpp_host") or {}
if mg.get("value"):
    rd += "`rank_bench --spawn 1` (C++ host, `rccl.h`): %.0f fields/s, checksums verified: %s.  " % (mg["value"], mg.get("rank_checksums_verified"))
if "device_stream" in d and "value" in d["device_stream"]:
    rd += "\n\nA device-resident stream of fresh batches (`device_stream`: every step the next 600 fields through `ntscsim_fields_device()`, preparation inside the clock): %.0f frames/s = %.2f x `value`, last step verified: %s.  " % (
        d["device_stream"]["value"], d["device_stream"]["value"] / d["value"], d["device_stream"].get("verified_last_step"))
if "variant422" in d:
    rd += "YUV422P tool: %.0f frames/s.  " % d["variant422"]["value"]
if "raw28" in d:
    r28 = d["raw28"]
    rd += "Raw-composite decoder (`raw28`): %d fields in %.1f ms = %.0f fields/s (phases in `raw28.stats`, us); reference text on one host core %.0f fields/s.  " % (
        r28["fields"], r28["ms_per_capture"], r28["value"], r28["cpu_1core"]["value"])
if "sizes" in d:
    rd += "1920x1080: %.0f, 3840x2160: %.0f frames/s.  " % (d["sizes"]["1920x1080"]["value"], d["sizes"]["3840x2160"]["value"])
if "presets" in d:
    rd += "Default preset at 720x486: %.0f frames/s.  " % d["presets"]["default"]["value"]
if isinstance(d.get("fast"), dict) and d["fast"].get("value"):
    rd += ("Tolerance modes (never the default; `fast`, `fast32`): NTSCSIM_MODE_FLOAT %.0f frames/s (decoder `roofline_frac` %.3f, kernels %s), NTSCSIM_MODE_FAST32 %.0f.  " % (
        d["fast"]["value"], d["fast"].get("roofline_frac") or 0, ", ".join(d["fast"].get("kernels", [])), (d.get("fast32") or {}).get("value", 0)))
rd += "\n\n"
rd += "## Kernel durations (us): hipEvents in bench.py vs rocprofv3\n\n"
rd += "| kernel | bench.py hipEvents (isolated pass) | rocprofv3 inflight 1 avg | rocprofv3 default cmd min / avg / max |\n|---|---|---|---|\n"
k = d["roofline"]["kernel_ms_all"]
for nm, key, ev in (("`k_decode_fast<true,double>`", "k_decode_fast<true", k["decode"]), ("`k_encode_fast<double>`", "k_encode_fast", k["encode"])):
    x, y = pick(b, key), pick(a, key)
    rd += "| %s | %.1f | %.1f | %.1f / %.1f / %.1f |\n" % (nm, ev * 1e3, x[1], y[2], y[1], y[3])
rd += "| `k_row_states` + `k_field_setup` | %.1f (\"setup\") | %.1f + %.1f | |\n\n" % (
    k["setup"] * 1e3, pick(b, "k_row_states")[1], pick(b, "k_field_setup")[1])
rd += "## Where the time goes\n\n"
cd = cen["k_decode"]
ce = cen["k_encode"]
rd += ("* `k_decode_fast<true,double>`: 2315 waves x 744 pipeline steps; steady step = %.0f VALU instructions (%.0f fp64, %.0f other half-rate, %.0f full-rate) = %.0f SIMD pipe cycles "
       "(`%s_isa_cost.json`); %.3g wave-instructions per launch (SQ_INSTS_VALU).  248 VGPRs, 2 waves per SIMD: 2048 wave slots for 2315 waves, so an isolated launch pays a second, "
       "one-wave-per-SIMD round (a lone wave needs ~1150 cycles per step, a pair ~1750): that is why `kernel_ms` (%.3f ms) is far above the launch's share of a saturated step.\n" % (
           cd["valu_per_step"], cd["fp64_per_step"], cd["half_rate_int_per_step"], cd["full_rate_per_step"], cd["valu_pipe_cycles_per_step"], tag, dec["SQ_INSTS_VALU"], k["decode"]))
rd += ("* `k_encode_fast<double>`: 2279 waves x 724 steps; steady step = %.0f VALU instructions = %.0f pipe cycles; %.3g wave-instructions per launch.  110 VGPRs; with 2.2 waves per SIMD in one launch it "
       "runs latency-bound (each wave ~410 cycles per step), with more waves resident (steps in flight) it approaches its pipe cost.\n" % (ce["valu_per_step"], ce["valu_pipe_cycles_per_step"], enc["SQ_INSTS_VALU"]))
rd += ("* HBM (PMC, calibrated counters: `r04_fetch_calibration.txt`): k_decode %.0f MB fetched + %.0f MB written, k_encode %.0f MB fetched + %.0f MB written per 600 fields = %.2f x the algorithmic 839.8 MB.  "
       "At %.2f ms per step that is ~%.1f TB/s of traffic on the L2's memory side against 6.3 TB/s achievable: the path is VALU-bound.\n" % (
           dec["FETCH_SIZE"] * f_dec_rd * 1024 / 1e6, dec["WRITE_SIZE"] * f_dec_wr * 1024 / 1e6, enc["FETCH_SIZE"] * f_enc_rd * 1024 / 1e6, enc["WRITE_SIZE"] * f_enc_wr * 1024 / 1e6,
           traffic["720x486 -vhs"]["path_over_algorithmic"],
           d["ms_per_step"], traffic["720x486 -vhs"]["path_hbm_bytes_per_launch"] / (d["ms_per_step"] * 1e-3) / 1e12))
if tc:
    b4 = tc["bench"]
    ks = pick(tc["stats"], "k422_fused")
    kname = [k for k in tc["stats"] if "k422_fused" in k][0].replace("ntscsim::", "")
    rd += ("\n## The YUV422P tool (`bench.py --tool to_composite`)\n\n"
           "`%s_bench_to_composite.json`: value = %.0f frames/s (%.3f ms per 600-field step, %d steps in flight), value_sustained = %.0f; "
           "`%s` (the name in `%s_kernel_stats_to_composite.csv`): %.1f us by hipEvents, %.1f us rocprofv3 avg with one step at a time; "
           "roofline.frac = %.3f of 8 TB/s on 4*W*L algorithmic bytes (%.0f MB per launch) against %.0f MB fetched + %.0f MB written "
           "(`%s_pmc_summary_to_composite.txt`: frame rows in and out plus the one composite-byte plane between sweep A and the streamed pass); %.3g wave-instructions per launch.  "
           "CPU beside it: %.1f frames/s (%s, 1 thread) => %.0fx.\n" % (
               tag, b4["value"], b4["ms_per_step"], b4["config"]["steps_in_flight"], b4.get("value_sustained", 0),
               kname, tag, b4["roofline"]["kernel_ms"] * 1e3, ks[1], b4["roofline"]["frac"], b4["roofline"]["algorithmic_bytes_per_launch"] / 1e6,
               tc["pmc"]["FETCH_SIZE"] * 1024 / 1e6, tc["pmc"]["WRITE_SIZE"] * 1024 / 1e6, tag, tc["pmc"]["SQ_INSTS_VALU"],
               b4.get("cpu_baseline", {}).get("value", 0), b4.get("cpu_baseline", {}).get("kind", "-"), b4.get("speedup_vs_cpu_1core", 0)))
    pr4 = b4.get("presets") or {}
    if pr4:
        rd += "\nSwitch-set families (`presets`, 24 steps after one per context, no pre-roll): " + "; ".join(
            "`%s` %.0f frames/s on `%s`" % (k_, v_.get("value", 0), ", ".join(v_.get("kernels", []))) for k_, v_ in pr4.items() if "value" in v_) + ".\n"
open("%s/README.md" % P, "w").write(rd)
print(rd)
