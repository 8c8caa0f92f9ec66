#!/usr/bin/env python3
"""Generates tests/golden/ntsc_bob_golden.npz from the REFERENCE's own field loop text.

Runs only in the build container (needs oracle/_ref/libntsc_ref.so, built by oracle/build_ref.sh from
/root/reference): for every case the reference's composite_layer() (ffmpeg_ntsc.cpp:1570) is called for
fields 0..n-1 into ONE frame, each call followed by the loop's "field deinterlace" block (:2233-2257,
extracted verbatim into a function) -- what main()'s field loop does between :2229 and the encoder.
Stores the sources and the frame after every field (data only).  Even and odd heights: the field-0
branch stops at y + 1 < height (:2248), so an even height leaves its last odd row alone.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _libs as L  # noqa: E402

CASES = [("bob_default_even", [], 96, 32, 4), ("bob_default_odd", [], 96, 33, 4),
         ("bob_vhs_even", ["-vhs"], 96, 32, 4), ("bob_vhs_odd", ["-vhs"], 100, 35, 4),
         ("bob_vhs_h2", ["-vhs"], 64, 2, 2), ("bob_vhs_h3", ["-vhs"], 64, 3, 2)]


def main():
    if not L.have_ref():
        raise SystemExit("oracle/_ref/libntsc_ref.so missing: run `sh oracle/build_ref.sh`")
    out = {}
    for (name, flags, w, h, n) in CASES:
        p = L.make_params(flags)
        srcs = [L.noise_frame(w, h, 40 + j) for j in range((n + 1) // 2)]
        r = L.RefStream(p)
        dst = np.full((h, w, 4), 0x5A, np.uint8)
        out["%s__src" % name] = np.stack(srcs)
        for k in range(n):
            r.field(dst, srcs[k // 2], (k & 1) ^ 1, k)
            r.bob(dst, k)
            out["%s__after%d" % (name, k)] = dst.copy()
    np.savez_compressed(os.path.join(HERE, "ntsc_bob_golden.npz"), **out)
    print("wrote %d cases" % len(CASES))


if __name__ == "__main__":
    main()
