#!/usr/bin/env python3
"""Rewrites the numbers table of README.md (between the markers `<!-- numbers:begin -->` / `<!-- numbers:end -->`) from
the tracked bench files under profiles/ (run after tools/refresh_profiles.sh local <tag>):
    python tools/readme_numbers.py r06"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = "profiles/%s_" % tag
line = json.load(open(P + "bench_driver_cmd.json"))          # what the driver's command printed
full = json.load(open(P + "bench.json"))                     # bench_extras.json of the same run
flt = json.load(open(P + "bench_float.json"))
f32 = json.load(open(P + "bench_fast32.json"))
toc = json.load(open(P + "bench_to_composite.json"))
try:
    dflt = json.load(open(P + "bench_line_default_cmd.json"))
except Exception:
    dflt = None
e = full.get("end_to_end", {})
fs, f4 = e.get("field_submit_detail", {}), e.get("field_submit422_detail", {})


def rate(d, k):
    x = d.get(k)
    return (x.get("fields_per_s") if isinstance(x, dict) else x) or 0


def k(v):
    return "%.0fk" % (v / 1e3) if v >= 10000 else "%.2fk" % (v / 1e3)


cb = line["cpu_baseline"]
rows = [
    ("**headline, driver's window** (`python bench.py --gpus 1 --steps 20 --warmup 5`): 600-field `-vhs` clip resident in HBM, exact mode, four steps in flight",
     "**%s fields/s** (`value`), %s sustained over 0.5 s" % (k(line["value"]), k(line["value_sustained"])), "`%sbench_driver_cmd.json`" % P),
    ("the same with `bench.py`'s own defaults (100 timed steps after 40 warm-up steps)",
     "%s fields/s" % k(dflt["value"]) if dflt else "-", "`%sbench_line_default_cmd.json`" % P),
    ("`roofline` of the dominant kernel (`k_decode_fast<true,double>`, algorithmic 8·W·L bytes per field ÷ its duration ÷ 8 TB/s)",
     "`frac` **%.3f** (%.0f GB/s); the kernel moves %.2f × those bytes over HBM (encoder + decoder: %.2f ×, the composite plane between them); the bound that applies is VALU issue: %.2f of its nominal capacity for the whole path" % (
         line["roofline"]["frac"], line["roofline"]["achieved"], (full["roofline"].get("traffic") or 0) / full["roofline"]["algorithmic_bytes_per_launch"],
         json.load(open("profiles/traffic.json"))["720x486 -vhs"]["path_over_algorithmic"], line["roofline"]["valu"]["path_frac_nominal"]), "DESIGN.md §5, `profiles/traffic.json`"),
    ("CPU beside it on the GPU box's host: the reference's own `composite_layer()` text, 1 thread like the tool",
     "%.1f fields/s (our C port: %.0f on 1 core, %.0f on %d cores)" % (cb["value"], cb["port_1core"], cb["port_all_cores"]["value"], cb["port_all_cores"]["cores"]),
     "`cpu_baseline`"),
    ("device-resident STREAM of fresh batches (descriptor validation, `rand()` windows, record upload inside the clock)", "%s fields/s" % k(line["side"]["device_stream"]), "`side.device_stream`"),
    ("other sizes, `-vhs`: 1920×1080 / 3840×2160", "%s / %s fields/s" % (k(line["side"]["sizes"]["1920x1080"]), k(line["side"]["sizes"]["3840x2160"])), "`side.sizes`"),
    ("default preset (BASELINE configs[0] on the GPU)", "%s fields/s" % k(line["side"]["presets"]["default"]), "`side.presets`"),
    ("the YUV422P tool (`ffmpeg_to_composite`), `-vhs`, device resident", "%s frames/s (sustained %s)" % (k(toc["value"]), k(toc.get("value_sustained", 0))), "`%sbench_to_composite.json`" % P),
    ("the raw-composite decoder (`ffmpeg_raw28ntsc`), 600-field capture resident in HBM", "%s fields/s" % k(line["side"]["raw28"]), "`side.raw28`"),
    ("**tolerance mode `NTSCSIM_MODE_FLOAT`** (all-float pipeline, ≤ 1 LSB, never the default)",
     "**%s fields/s**, decoder `roofline.frac` %.3f (FAST32, the exact kernels with float states: %s)" % (k(flt["value"]), flt["roofline"]["frac"], k(f32["value"])),
     "`%sbench_float.json`, DESIGN.md §3b" % P),
    ("drop-in on HOST frames, synchronous: one `ntscsim_field()` per `composite_layer()` call (`host/field_loop.cpp --mode sync`; a pipeline of wavefront roles, DESIGN §1c): `posix_memalign` frames / pinned frames / from Python",
     "%s / %s / %s fields/s (%.0f / %.0f × the reference on one core)" % (k(e.get("field_call_cpp", 0)), k(e.get("field_call_cpp_pinned", 0)), k(e.get("field_call", 0)),
                                                                       e.get("field_call_cpp", 0) / cb["value"], e.get("field_call_cpp_pinned", 0) / cb["value"]), "`side.field_call`, `side.field_call_pinned`, `side.field_call_python`"),
    ("... asynchronous, `ntscsim_submit()` / `ntscsim_wait()` at depth 32 (`host/field_loop.cpp`): frames from `ntscsim_host_frame_alloc()` / a pool declared with `ntscsim_host_pin()` / plain `posix_memalign` frames (staged)",
     "%s / %s / %s fields/s" % (k(e.get("field_submit", 0)), k(rate(fs, "depth32_declared_pool")), k(rate(fs, "depth32_malloc_frames_staged"))), "`end_to_end.field_submit*`"),
    ("the YUV422P tool's loop on host frames, depth 32 (`host/field_loop422.cpp`): pinned planes / plain heap planes, no `mallopt` / tight rows (704 wide)",
     "%s / %s / %s fields/s" % (k(e.get("field_submit422", 0)), k(rate(f4, "depth32_vhs_heap_planes")), k(rate(f4, "tight_rows_704"))),
     "`end_to_end.field_submit422*`"),
    ("... synchronous: one `ntscsim_field422()` per loop iteration (four wavefront roles, `k422_pipe`, DESIGN §7c): heap planes / pinned planes",
     "%s / %s fields/s" % (k(e.get("field_call422", 0) or rate(f4, "loop_sync_fields_per_s")), k(e.get("field_call422_pinned", 0))), "`side.field_call422`, `side.field_call422_pinned`"),
    ("whole clips from host memory (`ntscsim_frames_host`): BGRA out / YUV420P out / YUV420P in and out", "%s / %s / %s fields/s" % (
        k(e.get("bgra_pinned", 0)), k(e.get("yuv420p_pinned", 0)), k(e.get("yuv420p_in_yuv420p_out_pinned", 0))), "`end_to_end.*_pinned`"),
    ("one process per GPU, C++ host over `rccl.h` (`host/rank_bench.cpp`), with the one rank this box has", "%s fields/s, checksums verified" % k(line["side"]["multi_gpu_cpp_host"]), "`side.multi_gpu_cpp_host`"),
]
tbl = "| what (1× MI355X, 720×486 unless said otherwise) | measured | where |\n|---|---|---|\n" + "\n".join("| %s | %s | %s |" % r for r in rows)
s = open("README.md").read()
a, b = s.index("<!-- numbers:begin -->"), s.index("<!-- numbers:end -->")
s = s[:a] + "<!-- numbers:begin -->\n" + tbl + "\n" + s[b:]
open("README.md", "w").write(s)
print(tbl)
