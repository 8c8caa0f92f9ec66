#!/usr/bin/env python3
"""Copies the rocprofv3 / PMC summaries of one measurement round from gpurun_out/ into profiles/
and regenerates profiles/traffic.json and profiles/README.md.
Usage: python tools/make_profiles.py <bench.json> <prof default dir> <prof inflight1 dir> <pmc dir> [fast32 bench.json]"""
import csv, json, shutil, sys

bench = sys.argv[1]
fast = None
if len(sys.argv) >= 5:      # full refresh; with only <bench.json> the committed summaries are kept
    pdef, psingle, pmc = sys.argv[2:5]
    fast = sys.argv[5] if len(sys.argv) > 5 else None
    shutil.copy(pdef + "/r01_kernel_stats.csv", "profiles/r01_kernel_stats_default_cmd.csv")
    shutil.copy(psingle + "/r01_kernel_stats.csv", "profiles/r01_kernel_stats_inflight1.csv")
    shutil.copy(pmc + "/summary.txt", "profiles/r01_pmc_summary.txt")
if bench != "profiles/r01_bench.json":
    shutil.copy(bench, "profiles/r01_bench.json")
if fast:
    shutil.copy(fast, "profiles/r01_bench_fast32.json")
else:
    fast = "profiles/r01_bench_fast32.json"


def top(path):
    out = []
    for r in csv.DictReader(open(path)):
        if "ntscsim" in r["Name"]:
            out.append((r["Name"].split("(")[0], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    return out


def find(lst, name):
    for n, c, avg, mn, mx in lst:
        if name in n:
            return avg, mn, mx
    raise KeyError(name)


a, b = top("profiles/r01_kernel_stats_default_cmd.csv"), top("profiles/r01_kernel_stats_inflight1.csv")
d = json.load(open("profiles/r01_bench.json"))
pm, cur = {}, None
for l in open("profiles/r01_pmc_summary.txt"):
    if not l.startswith(" "):
        cur = l.strip()
    else:
        pm.setdefault(cur, {})[l.split()[0]] = float(l.split("mean=")[1])
dec = [v for k, v in pm.items() if "k_decode<true, true, 6u" in k][0]
enc = [v for k, v in pm.items() if "k_encode<8u" in k][0]
traffic = {"720x486 -vhs": {
    "fields_per_launch": 600,
    "k_decode_hbm_bytes_per_launch": (dec["FETCH_SIZE"] + dec["WRITE_SIZE"]) * 1024,
    "k_decode_fetch_KiB": dec["FETCH_SIZE"], "k_decode_write_KiB": dec["WRITE_SIZE"],
    "k_encode_fetch_KiB_raw": enc["FETCH_SIZE"],
    "k_encode_fetch_KiB_x2_gfx950_wide_load_correction": 2 * enc["FETCH_SIZE"],
    "k_encode_write_KiB": enc["WRITE_SIZE"],
    "valu_wave_insts_per_launch": {"k_decode": dec["SQ_INSTS_VALU"], "k_encode": enc["SQ_INSTS_VALU"],
                                   "setup": sum(v["SQ_INSTS_VALU"] for k, v in pm.items()
                                                if "k_row_states" in k or "k_field_setup" in k)},
    "note": "rocprofv3 --pmc, one counter per pass (tools/pmc.sh), bench.py --inflight 1; FETCH_SIZE/"
            "WRITE_SIZE are in KiB; WRITE_SIZE is calibrated by k_encode, whose only stores are the "
            "composite plane: 410062 KiB == 720*145800*4 B exactly; k_decode loads are 4 B/lane (no x2 "
            "correction applies), k_encode loads are 16 B/lane (the guide's x2 correction applies)"}}
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
ev = d["roofline"]["kernel_ms_all"]
md = ["# profiles/ -- round 1 (MI355X, gfx950, ROCm 7.2)\n",
      "All files come from `python bench.py` (BASELINE configs[1]: 720x486, 600 fields per step, `-vhs`); "
      "regenerate with `tools/make_profiles.py`.\n",
      "| file | command | what |", "|---|---|---|",
      "| `r01_bench.json` | `python bench.py` | the bench line (value, roofline, cpu_baseline) |",
      "| `r01_bench_fast32.json` | `python bench.py --mode fast32 --cpu-fields 0` | the optional fp32 mode (stated tolerance) |",
      "| `r01_kernel_stats_default_cmd.csv` | `rocprofv3 --kernel-trace --stats -- python bench.py --cpu-fields 0` | same command as the bench line: 3 steps in flight, so kernels of different steps share the GPU and their wall durations stretch; the **Min** column is the un-shared duration |",
      "| `r01_kernel_stats_inflight1.csv` | `... bench.py --cpu-fields 0 --inflight 1` | one step at a time: per-kernel durations without overlap |",
      "| `r01_pmc_summary.txt` | `tools/pmc.sh` (4 separate `--pmc` passes, `--inflight 1`) | FETCH_SIZE, WRITE_SIZE, SQ instruction counters per kernel (mean per launch) |",
      "| `traffic.json` | derived from the PMC summary | HBM bytes per launch that `bench.py` reports as `roofline.traffic` |\n",
      "## Bench line\n",
      "`value` = %.0f frames/s (fields/s), %.3f ms per 600-field step; `roofline.frac` = %.3f (k_decode, HBM "
      "algorithmic bytes) and `roofline.valu.path_frac` = %.2f (VALU issue slots, the bound that applies).  CPU beside "
      "it on the GPU box's host (2x EPYC 9575F, cgroup quota 16 CPUs): the reference's own `composite_layer()` "
      "(`oracle/_ref`, single-threaded like the tool) %.1f fields/s => %.0fx; our port 1 core %.1f fields/s; our port on "
      "all %d usable CPUs %.0f fields/s => %.0fx.%s\n" % (
          d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["valu"]["path_frac"],
          d["cpu_baseline"]["value"], d["speedup_vs_cpu_1core"], d["cpu_baseline"]["port_1core"],
          d["cpu_baseline"]["port_all_cores"]["cores"], d["cpu_baseline"]["port_all_cores"]["value"],
          d["speedup_vs_cpu_all_cores"],
          (" FAST32 mode: %.0f frames/s." % json.load(open(fast))["value"]) if fast else ""),
      "## Kernel durations (us): hipEvents in bench.py vs rocprofv3\n",
      "| kernel | bench.py hipEvents (isolated pass) | rocprofv3 inflight 1 avg | rocprofv3 default cmd min / avg / max |",
      "|---|---|---|---|"]
for kn, key in (("k_decode", "decode"), ("k_encode", "encode")):
    ia, pa = find(b, kn), find(a, kn)
    md.append("| `%s` | %.1f | %.1f | %.1f / %.1f / %.1f |" % (kn, ev[key] * 1e3, ia[0], pa[1], pa[0], pa[2]))
rs, fs = find(b, "k_row_states"), find(b, "k_field_setup")
md.append("| `k_row_states` + `k_field_setup` (+ memset) | %.1f (\"setup\") | %.1f + %.1f | |\n" % (ev["setup"] * 1e3, rs[0], fs[0]))
DSTEPS = 720 + 24      # W + pipeline depth SKT = 7 + cdelay(9, SP) + 7 + 1 (k_decode, -vhs preset)
vd = dec["SQ_INSTS_VALU"] / (dec["SQ_WAVES"] * DSTEPS)
ve = enc["SQ_INSTS_VALU"] / (enc["SQ_WAVES"] * (720 + 4))
ghz = dec["GRBM_GUI_ACTIVE"] / 8 / (find(b, "k_decode")[0] * 1e-6) / 1e9
md += ["## Where the time goes\n",
       "* `k_decode<VHS,COMPOUT,preset>`: %d waves x 744 pipeline steps (W + depth 24), %.0f VALU instructions per step per wave "
       "(SQ_INSTS_VALU / waves / steps, averaged over steady and guarded steps).  DERIVED, assuming every VALU "
       "instruction occupies its SIMD for 4 cycles (SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU in quad-cycles; "
       "`tools/valu_rate_probe.hip` shows simple int/fp32 ops can issue faster, so this is an upper bound on the "
       "busy time): all-VALU-busy time for the busiest SIMDs (3 resident waves) = 3 x 744 x %.0f x 4 cycles = "
       "%.2f ms at the observed %.2f GHz vs %.2f ms measured (~%.0f%%).  The PMC-based fractions over ALL SIMDs are in "
       "the bench line: `roofline.valu.k_decode_frac` (one launch) and `path_frac` (three steps in flight).  With "
       "perfect load balance (2,315 waves over 1,024 SIMDs = 2.26 waves/SIMD) the same instruction "
       "stream would need %.2f ms; three steps in flight recover most of that (%.2f ms per step for the whole chain; a "
       "single 2,400-field batch reaches the same rate on one stream, `tools/bigbatch_probe.py`)." % (
           int(dec["SQ_WAVES"]), vd, vd, 3 * DSTEPS * vd * 4 / (ghz * 1e9) * 1e3, ghz, ev["decode"],
           100 * 3 * DSTEPS * vd * 4 / (ghz * 1e9) * 1e3 / ev["decode"], 2.26 * DSTEPS * vd * 4 / (ghz * 1e9) * 1e3, d["ms_per_step"]),
       "* `k_encode<preset>`: %d waves x 724 steps, %.0f VALU instructions per step." % (int(enc["SQ_WAVES"]), ve),
       "* HBM: k_decode %.0f MB fetched + %.0f MB written per launch, k_encode %.0f MB written (= the composite plane, "
       "exact) -- algorithmic 839.8 MB for the whole path; at the measured %.2f ms per step that is %.1f TB/s of "
       "physical traffic, far from the 6.3 TB/s achievable: the path is VALU-bound." % (
           dec["FETCH_SIZE"] * 1024 / 1e6, dec["WRITE_SIZE"] * 1024 / 1e6, enc["WRITE_SIZE"] * 1024 / 1e6, d["ms_per_step"],
           ((dec["FETCH_SIZE"] + dec["WRITE_SIZE"] + 2 * enc["FETCH_SIZE"] + enc["WRITE_SIZE"]) * 1024 / 1e12) / (d["ms_per_step"] * 1e-3)),
       ""]
open("profiles/README.md", "w").write("\n".join(md))
print("\n".join(md[-12:]))
