"""Generates tests/golden/pure_pins.npz from oracle/_ref/libref_pure.so -- the reference's own text for
LowpassFilter, RGB_to_YIQ, YIQ_to_RGB, clampu8, black_key and hsync_dc_proc compiled by
oracle/build_ref_pure.sh with libc / STL headers alone (no stand-in declarations).  The file holds inputs'
seeds and the reference's outputs (hashes + samples): data, not source.  Run in the container that has
/root/reference:   sh oracle/build_ref_pure.sh && python tests/golden/make_pure_pins.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pure as P  # noqa: E402


def main():
    lib = P.pure_ref()
    out = {}
    out["cube_fnv"] = np.array([P.cube_hash(lib, "ref")], np.uint64)
    idx = P.cube_sample_index()
    out["cube_sample"] = P.rgb_to_yiq(lib, "ref", P.cube_triples(idx))
    for k, (tool, rate, hz, reset, hp, seed, n) in enumerate(P.FILTER_CASES):
        y, alpha = P.run_filter(lib, "ref", tool, rate, hz, reset, hp, P.filter_input(seed, n))
        out["filter%02d" % k] = y.view(np.uint64)
        out["alpha%02d" % k] = np.array([alpha], np.float64).view(np.uint64)
    yiq = P.yiq_input()
    rgb = P.yiq_to_rgb(lib, "ref", yiq)
    out["yiq_rgb_fnv"] = np.array([P.fnv(rgb)], np.uint64)
    out["yiq_rgb_sample"] = rgb[:4096].astype(np.uint8)
    x = P.clamp_input()
    out["clamp_fnv"] = np.array([P.fnv(P.clampu8(lib, "ref", x))], np.uint64)
    for level in P.BKEY_LEVELS:
        for wch in (0, 1):
            d, f = P.bkey_input(level)
            P.black_key(lib, "ref", level, wch, d, f)
            out["bkey_%d_%d" % (level, wch)] = np.array([P.fnv(d), P.fnv(f)], np.uint64)
    for k, (rate, mark, fields, seed, noise, cut) in enumerate(P.FRONT_CASES):
        cap = P.front_capture(fields, seed, noise, cut)
        h, r = P.raw28_front(lib, "ref", rate, mark, cap)
        out["front%d" % k] = np.array([P.fnv(h), P.fnv(r)], np.uint64)
        out["front%d_head" % k] = np.stack([h[:8192], r[:8192]])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pure_pins.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
