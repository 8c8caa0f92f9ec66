#!/usr/bin/env python3
"""Generates tests/golden/tocomp_golden.npz from the REFERENCE's own 8-bit hot-path code
(oracle/_ref/libtocomp_ref.so = line ranges of /root/reference/ffmpeg_to_composite.cpp compiled by
oracle/build_ref.sh).  Build-container only.  Each case: one YUV422P buffer (Y|U|V in one
allocation, linesize == width, so the reference's two-byte read past each luma row lands on the
next row / next plane, all of it recorded), fields 0..n-1 processed in place exactly like the
tool's loop; the WHOLE buffer after every field is stored (data only)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _libs as L  # noqa: E402
import cases422  # noqa: E402

SUBSET = ["default", "vhs", "vhs_bars", "vhs_oddh", "vhs_ep", "vhs_svideo", "vhs_pal", "out_lite_only",
          "catv3_vhs", "phase90", "after_yc_sep", "yc_recomb2", "amp30", "dropout_often", "hs_inframe"]


def main():
    if not L.have_tocomp_ref():
        raise SystemExit("oracle/_ref/libtocomp_ref.so missing: run `make -C oracle ref`")
    out, manifest = {}, []
    for (name, flags, w, h, n, kind) in cases422.CASES422:
        if name not in SUBSET:
            continue
        p = L.make_params_tocomp(flags)
        srcs = [cases422.make_source422(kind, w, h, j) for j in range((n + 1) // 2)]
        fr = srcs[0].copy()
        r = L.TocompRefStream(p)
        out["%s__init" % name] = fr.buf.copy()
        for j, s in enumerate(srcs):
            out["%s__src%d" % (name, j)] = s.buf.copy()
        for k in range(n):
            field = (k & 1) ^ 1
            for i in range(3):
                fr.plane(i)[field::2] = srcs[k // 2].plane(i)[field::2]
            r.process(fr, field, k)
            out["%s__after%d" % (name, k)] = fr.buf.copy()
        manifest.append({"name": name, "flags": flags, "w": w, "h": h, "n": n, "src": kind})
    np.savez_compressed(os.path.join(HERE, "tocomp_golden.npz"), **out)
    with open(os.path.join(HERE, "tocomp_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden422.py",
                   "source": "reference ffmpeg_to_composite.cpp:97-131,261,267-333,335-351,353-553,"
                             "629-1129 via oracle/build_ref.sh", "cases": manifest}, f, indent=1)
    print("wrote %d cases" % len(manifest))


if __name__ == "__main__":
    main()
