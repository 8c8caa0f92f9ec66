#!/usr/bin/env python3
"""Rewrites the "Round-N numbers" paragraph of README.md from the tracked bench lines under
profiles/ (run after tools/refresh_profiles.sh local <tag>):  python tools/readme_numbers.py r03"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
d = json.load(open("profiles/%s_bench.json" % tag))
t = json.load(open("profiles/%s_bench_to_composite.json" % tag))
f = json.load(open("profiles/%s_bench_fast32.json" % tag))
s = open("README.md").read()
import re
a, b = re.search(r"Round-\d numbers", s).start(), s.index("Build: `python -c")
v = d["roofline"]["valu"]
p_ = d.get("presets", {})
cb, e = d["cpu_baseline"], d["end_to_end"]
r28 = d.get("raw28")
raw28 = ("  The raw-composite decoder (`ffmpeg_raw28ntsc`, `raw28`): a 600-field capture resident in HBM decodes at "
         "%.1fk fields/s (the reference text on one host core: %.0f)." % (r28["value"] / 1e3, r28["cpu_1core"]["value"])) if r28 else ""
fcall = ("  One field per synchronous `ntscsim_field()` call (the 1:1 drop-in on host frames): %.0f fields/s." % e["field_call"]) if e.get("field_call") else ""
cli = ("  The raw-file CLI `ntsc_cli -vhs -i bars:3000 -o null:` runs at %.0fk fields/s (`end_to_end.cli`)." % (e["cli"] / 1e3)) if e.get("cli") else ""
ds = d.get("device_stream", {})
fs = e.get("field_submit_detail", {})
def fps(k):
    x = fs.get(k) or {}
    return (x.get("fields_per_s") or 0) / 1e3
pre = (d["config"].get("pre_roll") or {})
f4 = e.get("field_submit422_detail") or {}
def f4ps(k):
    x = f4.get(k)
    return ((x.get("fields_per_s") or 0) if isinstance(x, dict) else (x or 0)) / 1e3
tp = t.get("presets") or {}
v422 = ""
if e.get("field_submit422"):
    v422 = ("  The YUV422P tool on HOST frames (`ntscsim_field422()` / `ntscsim_submit422()`, `host/field_loop422.cpp`: the loop of "
            "ffmpeg_to_composite.cpp:1783-1800 with its four calls replaced by one, 720×480, depth 32): %.0fk fields/s with `-vhs` "
            "(frame planes pinned in place), %.0fk through the staging rings (heap-block planes: no `mallopt`), %.1fk one iteration at a time "
            "(tight rows), %.1fk synchronous (`end_to_end.field_submit422*`)." % (
                e["field_submit422"] / 1e3, f4ps("depth32_vhs_heap_planes"), f4ps("tight_rows_704_one_at_a_time"), f4ps("loop_sync_fields_per_s")))
    if tp.get("default", {}).get("value"):
        v422 += ("  Its switch-set families on the device (`bench.py --tool to_composite` → `presets`): default preset %dk frames/s (`%s`), "
                 "`-vhs -vhs-svideo 1` %dk (`%s`)." % (round(tp["default"]["value"] / 1e3), ", ".join(tp["default"].get("kernels", [])),
                                                     round(tp.get("vhs_svideo", {}).get("value", 0) / 1e3), ", ".join(tp.get("vhs_svideo", {}).get("kernels", []))))
mg = d.get("multi_gpu_cpp_host") or {}
mgpu = ("  One process per GPU with the C++ host over `rccl.h` (`host/rank_bench.cpp`, here with the one rank this box has): %dk fields/s, checksums verified."
        % round(mg["value"] / 1e3)) if mg.get("value") else ""
new = ("Round-%s numbers (1× MI355X, 720×486, 600-field clip, full `-vhs` preset, exact mode; every figure is\n"
       "a key of `profiles/%s_bench.json`, the line `python bench.py` prints; `profiles/README.md` maps the\n"
       "rest; box-to-box spread ≈ ±4 %%: `value` 764-813k over the boxes this round saw, the isolated kernels take the same time on all of them):\n"
       "`value_sustained` **%dk fields/s** (the same 600-field step repeated for %.2f s, four steps in flight) and `value` **%dk** over the %d\n"
       "timed steps that follow it (`config.pre_roll`: the sustained leg runs BEFORE the W warm-up steps, so the timed steps see a GPU at its\n"
       "working clocks; rounds 1 and 2 were measured without it and read 4-6 %% lower for that reason alone).  With the driver's own\n"
       "window (`--steps 20 --warmup 5`): %dk (`profiles/%s_bench_driver_cmd.json`).  A device-resident STREAM of fresh batches -- every step the next\n"
       "600 fields, descriptor validation, `rand()` windows and record upload inside the clock -- runs at %dk (`device_stream`, %.2f × `value`).\n"
       "CPU beside it on the GPU box's host: the reference's own `composite_layer()`\n"
       "(`oracle/_ref`, single-threaded like the tool) %.0f fields/s, our C port %.0f fields/s on one core and\n"
       "%d fields/s on the %d CPUs the box's cgroup allows.  `sizes`: 1920×1080 %.1fk, 3840×2160 %.1fk\n"
       "fields/s; `presets.default`: %dk fields/s; other switch sets, each on a hand-tuned form of its own, as a fraction of the `-vhs` preset measured the same way (`presets.*.frac_of_preset`): %s;\n"
       "the YUV422P tool (`python bench.py --tool to_composite`,\n"
       "`profiles/%s_bench_to_composite.json`): **%dk frames/s** sustained, %dk over the timed steps.  The path is VALU-issue\n"
       "bound, not HBM bound: `roofline.frac` (HBM, algorithmic bytes) = %.3f, and %.2f is the most a kernel chain with the\n"
       "reference's fp64 arithmetic could reach (`roofline.valu.hbm_frac_ceiling_exact_mode`); `roofline.valu.path_frac_nominal` =\n"
       "%.2f of the VALU issue capacity at the pipe's nominal 4 / 2 cycles per instruction (%.2f at the measured slowest-wave\n"
       "costs; PMC instruction counts × each kernel's instruction mix ÷ measured time) — see `profiles/README.md`, `profiles/r04_decode_census.txt` and DESIGN.md §5 for what\n"
       "was measured and what is derived.  The drop-in on HOST frames: one synchronous `ntscsim_field()` per `composite_layer()` call %.1fk fields/s; the\n"
       "same loop with `ntscsim_submit()` / `ntscsim_wait()` at depth 32 (`host/field_loop.cpp`, pageable AVFrame-shaped buffers pinned in place) **%.0fk** with ONE\n"
       "source frame rewritten per decoded frame, %.0fk with the source re-pointed at decoded frames, %.0fk with the line doubling delivered too, %.0fk at depth 128\n"
       "(`end_to_end.field_submit*`; %.0f × the reference on one core; these are the host link's rates, not the GPU's -- 0.7 MB up and 0.7 MB down per field, "
       "1.4 MB down with the line doubling: DESIGN.md §1b).  Whole clips from host memory (`ntscsim_frames_host`): %.0fk\n"
       "fields/s BGRA out, %.0fk with YUV420P made on the GPU, %.0fk with YUV420P in as well.%s%s%s%s  Not a headline: the optional\n"
       "`NTSCSIM_MODE_FAST32` (the same kernels with fp32 filter states, ≤1 LSB, not bit-exact) runs %dk fields/s (`profiles/%s_bench_fast32.json`).\n\n" % (
           tag[1:].lstrip("0"), tag, round(d["value_sustained"] / 1e3), (d.get("sustained") or {}).get("seconds", 0.5), round(d["value"] / 1e3), d["steps"],
           round(json.load(open("profiles/%s_bench_driver_cmd.json" % tag))["value"] / 1e3), tag,
           round(ds.get("value", 0) / 1e3), ds.get("value", 0) / d["value"],
           cb["value"], cb["port_1core"],
           round(cb["port_all_cores"]["value"]), cb["port_all_cores"]["cores"],
           d["sizes"]["1920x1080"]["value"] / 1e3, d["sizes"]["3840x2160"]["value"] / 1e3,
           round(d["presets"]["default"]["value"] / 1e3),
           ", ".join("%s %dk (%.2f)" % (k[4:], round(x["value"] / 1e3), x.get("frac_of_preset") or 0) for k, x in p_.items() if k.startswith("vhs_") and "value" in x),
           tag, round(t["value_sustained"] / 1e3), round(t["value"] / 1e3), d["roofline"]["frac"], v["hbm_frac_ceiling_exact_mode"],
           v["path_frac_nominal"], v["path_frac"], e["field_call"] / 1e3,
           e.get("field_submit", 0) / 1e3, fps("depth32_decoder_frames"), fps("depth32_bob"), fps("depth128"),
           e.get("field_submit", 0) / cb["value"],
           e["bgra_pinned"] / 1e3, e["yuv420p_pinned"] / 1e3,
           e.get("yuv420p_in_yuv420p_out_pinned", 0) / 1e3, cli, raw28, v422, mgpu,
           round(f["value"] / 1e3), tag))
open("README.md", "w").write(s[:a] + new + s[b:])
print(new)
