#!/usr/bin/env python3
"""Generates tests/golden/ntsc_golden.npz from the REFERENCE's own hot-path code.

Runs only in the build container, where /root/reference exists: oracle/build_ref.sh compiles the
line ranges of ffmpeg_ntsc.cpp holding composite_layer() and its helpers (never copied into this
repo) into oracle/_ref/libntsc_ref.so; this script drives it over tests/cases.py and stores
inputs + expected outputs (data only).  Each case calls composite_layer() for fields 0..n-1 with
field=(k&1)^1, fieldno=k, all into ONE dst frame that starts zeroed -- exactly what the
reference's field loop does with a 1-frame delay ring (ffmpeg_ntsc.cpp:2070-2092, :2229).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _libs as L  # noqa: E402
import cases  # noqa: E402


def main():
    if not L.have_ref():
        raise SystemExit("oracle/_ref/libntsc_ref.so missing: run `make -C oracle ref`")
    out = {}
    manifest = []
    for i, (name, flags, w, h, n, kind, il, tff) in enumerate(cases.CASES):
        p = L.make_params(flags)
        srcs = [cases.make_source(kind, w, h, j) for j in range((n + 1) // 2)]
        r = L.RefStream(p)
        dst = np.zeros((h, w, 4), np.uint8)
        per_field = []
        for (si, field, fieldno) in cases.case_jobs(n):
            r.field(dst, srcs[si], field, fieldno, il, tff)
            per_field.append(dst[field::2].copy())
        out["%s__src" % name] = np.stack(srcs)
        for k, a in enumerate(per_field):
            out["%s__field%d" % (name, k)] = a
        out["%s__final" % name] = dst
        manifest.append({"name": name, "flags": flags, "w": w, "h": h, "n": n, "src": kind,
                         "interlaced": il, "tff": tff, "fnv1a_final": "%016x" % L.fnv1a(dst)})
    np.savez_compressed(os.path.join(HERE, "ntsc_golden.npz"), **out)
    with open(os.path.join(HERE, "ntsc_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py",
                   "source": "reference ffmpeg_ntsc.cpp:72-106,205-214,756-809,1375-1921 via "
                             "oracle/build_ref.sh (g++ -O2 -ffp-contract=off, glibc rand(), seed 1)",
                   "cases": manifest}, f, indent=1)
    # full-size hashes (inputs are procedural: 8-bar clip, frame k//2 rotated by k//2 pixels)
    big = []
    for (w, h, flags, n) in [(720, 480, [], 8), (720, 480, ["-vhs"], 8), (720, 486, [], 8),
                             (720, 486, ["-vhs"], 8), (1920, 1080, [], 4), (1920, 1080, ["-vhs"], 4),
                             (3840, 2160, [], 4), (3840, 2160, ["-vhs"], 8)]:
        p = L.make_params(flags)
        r = L.RefStream(p)
        dst = np.zeros((h, w, 4), np.uint8)
        hashes = []
        for k in range(n):
            r.field(dst, L.bars(w, h, k // 2), (k & 1) ^ 1, k)
            hashes.append("%016x" % L.fnv1a(dst))
        big.append({"w": w, "h": h, "flags": flags, "n": n, "fnv1a_after_each_field": hashes})
    with open(os.path.join(HERE, "ntsc_fullsize_hashes.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "cases": big}, f, indent=1)
    print("wrote %d cases" % len(manifest))


if __name__ == "__main__":
    main()
