#!/usr/bin/env python3
"""tools/float_err.py -- error of a tolerance mode (FAST32 / FLOAT) against the oracle: per case the largest channel
difference, the share of pixels that differ at all, the mean signed difference per channel (B, G, R: what a luma bias
constant has to cancel) and the histogram of differences.  GPU box only; prints one line per case and a JSON summary.
  python tools/float_err.py [--mode float|fast32] [--quick]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "composite-video-simulator_amd"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="float")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    import torch
    import _libs as L
    import ntscsim
    from ntscsim import _capi
    mode = _capi.MODE_FLOAT if a.mode == "float" else _capi.MODE_FAST32
    cases = [([], 720, 486, 4, "bars"), ([], 720, 486, 2, "noise"), (["-vhs"], 720, 486, 4, "bars"),
             (["-vhs"], 720, 486, 4, "noise"), (["-vhs"], 720, 486, 2, "ramp")]
    if not a.quick:
        cases += [(["-vhs", "-vhs-speed", "ep"], 720, 480, 2, "noise"), (["-vhs", "-vhs-speed", "lp"], 704, 480, 2, "noise"),
                  (["-vhs"], 1920, 1080, 2, "noise"), (["-vhs"], 3840, 2160, 2, "bars"), (["-vhs"], 3840, 2160, 2, "noise")]
    out = []
    for flags, w, h, n, kind in cases:
        p = L.make_params(flags)
        if kind == "noise":
            srcs = [L.noise_frame(w, h, 5 + j) for j in range((n + 1) // 2)]
        elif kind == "bars":
            srcs = [L.bars(w, h, j) for j in range((n + 1) // 2)]
        else:       # smooth ramps: every level of every channel, slowly varying (natural-image-like truncation statistics)
            x = np.arange(w)[None, :]
            y = np.arange(h)[:, None]
            fr = np.zeros((h, w, 4), np.uint8)
            fr[..., 0] = (x * 255 // max(1, w - 1)); fr[..., 1] = (y * 255 // max(1, h - 1)); fr[..., 2] = ((x + y) * 255 // (w + h - 2))
            srcs = [fr for _ in range((n + 1) // 2)]
        o = L.OracleStream(p)
        exp = np.zeros((n, h, w, 4), np.uint8)
        for k in range(n):
            o.field(exp[k], srcs[k // 2], (k & 1) ^ 1, k)
        sim = ntscsim.FieldSimulator(params=p)
        sim.set_mode(mode)
        src = torch.from_numpy(np.stack(srcs)).cuda()
        dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
        sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)])
        sim.sync()
        kern = sim.last_kernels()
        got = dst.cpu().numpy()
        sim.close()
        rows = np.concatenate([(got[k].astype(np.int16) - exp[k].astype(np.int16))[((k & 1) ^ 1)::2] for k in range(n)])
        d = rows[..., :3]
        hist = {int(v): int((d == v).sum()) for v in np.unique(d)}
        frac_px = float((np.abs(d).max(axis=-1) > 0).mean())
        edge = d[:, -24:, :]
        rec = {"flags": " ".join(flags) or "default", "size": "%dx%d" % (w, h), "kind": kind, "fields": n,
               "max_abs": int(np.abs(d).max()), "frac_pixels_differ": frac_px,
               "frac_channels_differ": float((d != 0).mean()), "mean_signed_bgr": [float(d[..., c].mean()) for c in range(3)],
               "hist": hist, "max_abs_last24_columns": int(np.abs(edge).max()), "max_abs_first24_columns": int(np.abs(d[:, :24]).max()),
               "kernels": [k_ for k_ in kern if k_.startswith(("k_enc", "k_dec"))]}
        out.append(rec)
        print("%-26s %-10s %-5s max %d  px differ %.4f  ch differ %.4f  mean bgr %+.4f %+.4f %+.4f  %s" % (
            rec["flags"], rec["size"], kind, rec["max_abs"], frac_px, rec["frac_channels_differ"], *rec["mean_signed_bgr"],
            ",".join(rec["kernels"])), flush=True)
    print(json.dumps({"mode": a.mode, "cases": out}))


if __name__ == "__main__":
    main()
