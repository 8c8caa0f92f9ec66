"""Developer probe (not a test): run HIP path vs oracle on a few configurations and print
mismatch statistics.  Usage on the GPU box:  python tools/gpu_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import _libs as L
import ntscsim
import torch

def run(flags, w, h, nfields, src_kind="noise", stage=False, **ov):
    p = L.make_params(flags, **ov)
    frames = [L.noise_frame(w, h, 0x1234567 + i) if src_kind == "noise" else L.bars(w, h, i)
              for i in range((nfields + 1) // 2)]
    jobs = [(k // 2, k, (k & 1) ^ 1, k) for k in range(nfields)]
    # oracle: each field into its own zeroed dst
    o = L.OracleStream(p)
    exp = np.zeros((nfields, h, w, 4), np.uint8)
    taps = []
    for (si, di, field, fieldno) in jobs:
        t = o.field(exp[di], frames[si], field, fieldno, taps=["composite_y"] if stage else None)
        taps.append(t)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack(frames)).cuda()
    dst = torch.zeros((nfields, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, jobs)
    sim.sync()
    got = dst.cpu().numpy()
    bad = (got != exp).any(axis=-1)
    msg = "flags=%s %dx%d n=%d src=%s ov=%s: mismatching px %d / %d" % (
        flags, w, h, nfields, src_kind, ov, bad.sum(), bad.size)
    if stage:
        comp = sim.debug_composite(nfields, w, h)
        for i, (si, di, field, fieldno) in enumerate(jobs):
            Lr = L.field_rows(h, field)
            cb = (comp[i, :Lr] != taps[i]["composite_y"])
            if cb.any():
                ys, xs = np.nonzero(cb)
                msg += "\n   composite mismatch field %d: %d px, first (k=%d,x=%d) got %d exp %d" % (
                    i, cb.sum(), ys[0], xs[0], comp[i, ys[0], xs[0]], taps[i]["composite_y"][ys[0], xs[0]])
    if bad.any():
        fi, ys, xs = np.nonzero(bad)
        msg += "\n   first: field %d y %d x %d got %s exp %s; rows with errors: %s ; x hist: %s" % (
            fi[0], ys[0], xs[0], got[fi[0], ys[0], xs[0]], exp[fi[0], ys[0], xs[0]],
            np.unique(ys)[:12], np.unique(xs)[:16])
    print(msg, flush=True)
    sim.close()
    return int(bad.sum())

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    tot = 0
    tot += run([], 64, 16, 2, stage=True, video_noise=0)
    tot += run([], 64, 16, 2, stage=True)
    tot += run([], 96, 32, 4, stage=True)
    tot += run(["-vhs", "-vhs-svideo", "1"], 96, 32, 4, stage=True)
    tot += run(["-vhs"], 96, 32, 4, stage=True)
    tot += run(["-vhs"], 100, 34, 3, src_kind="bars")
    tot += run(["-vhs"], 97, 33, 3)
    tot += run(["-vhs", "-vhs-speed", "ep"], 96, 32, 4)
    tot += run(["-vhs", "-out-composite-lowpass-lite", "0"], 96, 32, 4)
    tot += run(["-out-composite-lowpass", "0"], 96, 32, 4)
    tot += run(["-comp-catv3"], 96, 32, 4)
    tot += run(["-vhs"], 720, 480, 4, src_kind="bars")
    tot += run(["-vhs"], 720, 486, 4)
    print("TOTAL mismatches", tot)
