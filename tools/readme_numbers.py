#!/usr/bin/env python3
"""Rewrites the "Round-N numbers" paragraph of README.md from the tracked bench lines under
profiles/ (run after tools/refresh_profiles.sh local <tag>):  python tools/readme_numbers.py r03"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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
new = ("Round-%s numbers (1× MI355X, 720×486, 600-field clip, full `-vhs` preset, exact mode; every figure is\n"
       "a key of `profiles/%s_bench.json`, the line `python bench.py` prints; `profiles/README.md` maps the\n"
       "rest; box-to-box spread ≈ ±4 %%: 733k-790k for `value` on ten boxes):\n"
       "`value` **%dk fields/s** over %d steps with four steps in flight, `value_sustained` **%dk** over\n"
       "0.5 s (round 2: 729k, round 1: 542k).  CPU beside it on the GPU box's host: the reference's own `composite_layer()`\n"
       "(`oracle/_ref`, single-threaded like the tool) %.0f fields/s, our C port %.0f fields/s on one core and\n"
       "%d fields/s on the %d CPUs the box's cgroup allows.  `sizes`: 1920×1080 %.1fk, 3840×2160 %.1fk\n"
       "fields/s; `presets.default`: %dk fields/s; other switch sets (`presets.vhs_*`; catv2 and svideo run hand-tuned forms of their own, the rest the generic kernels): %s;\n"
       "the YUV422P tool (`python bench.py --tool to_composite`,\n"
       "`profiles/%s_bench_to_composite.json`): **%dk frames/s** (round 2: 654k, round 1: 262k).  The path is VALU-issue\n"
       "bound, not HBM bound: `roofline.frac` (HBM, algorithmic bytes) = %.2f, and %.2f is the most a kernel chain with the\n"
       "reference's fp64 arithmetic could reach (`roofline.valu.hbm_frac_ceiling_exact_mode`); `roofline.valu.path_frac_nominal` =\n"
       "%.2f of the VALU issue capacity at the pipe's nominal 4 / 2 cycles per instruction (%.2f at the measured slowest-wave\n"
       "costs; PMC instruction counts × each kernel's instruction mix ÷ measured time) — see `profiles/README.md` and DESIGN.md §5 for what\n"
       "was measured and what is derived.  PCIe-inclusive (`end_to_end`, `ntscsim_frames_host`): %.0fk\n"
       "fields/s BGRA out, %.0fk with YUV420P made on the GPU, %.0fk with YUV420P in as well.%s%s%s  Optional `NTSCSIM_MODE_FAST32` (fp32 filters,\n"
       "≤1 LSB, not bit-exact): %dk fields/s (`profiles/%s_bench_fast32.json`).\n\n" % (
           tag[1:].lstrip("0"), tag, round(d["value"] / 1e3), d["steps"], round(d["value_sustained"] / 1e3), cb["value"], cb["port_1core"],
           round(cb["port_all_cores"]["value"]), cb["port_all_cores"]["cores"],
           d["sizes"]["1920x1080"]["value"] / 1e3, d["sizes"]["3840x2160"]["value"] / 1e3,
           round(d["presets"]["default"]["value"] / 1e3),
           ", ".join("%s %dk" % (k[4:], round(x["value"] / 1e3)) for k, x in p_.items() if k.startswith("vhs_") and "value" in x),
           tag, round(t["value"] / 1e3), d["roofline"]["frac"], v["hbm_frac_ceiling_exact_mode"],
           v["path_frac_nominal"], v["path_frac"], e["bgra_pinned"] / 1e3, e["yuv420p_pinned"] / 1e3,
           e.get("yuv420p_in_yuv420p_out_pinned", 0) / 1e3, fcall, cli, raw28,
           round(f["value"] / 1e3), tag))
open("README.md", "w").write(s[:a] + new + s[b:])
print(new)
