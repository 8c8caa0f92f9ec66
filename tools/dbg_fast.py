"""Developer probe (GPU): k_decode_fast vs the template PRESET decoder, first mismatching columns."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "composite-video-simulator_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, ntscsim, _libs as L
flags = sys.argv[1].split() if len(sys.argv) > 1 else ["-vhs"]
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (720, 480)
p = L.make_params(flags)
src = torch.from_numpy(np.stack([L.noise_frame(w, h, 5), L.noise_frame(w, h, 6)])).cuda()
jobs = [(k // 2, k, (k & 1) ^ 1, k) for k in range(4)]
outs = []
for nofast in (0, 1):
    sim = ntscsim.FieldSimulator(params=p)
    sim.debug_no_fast_decode(bool(nofast))
    dst = torch.zeros((4, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, jobs); sim.sync()
    outs.append(dst.cpu().numpy())
    sim.close()
a, b = outs
bad = (a != b).any(axis=-1)
print("mismatching pixels:", int(bad.sum()), "of", bad.size)
if bad.any():
    cols = np.where(bad.any(axis=(0, 1)))[0]
    rows = np.where(bad.any(axis=(0, 2)))[0]
    print("columns:", cols[:40], "...", cols[-10:])
    print("rows:", rows[:20], "... n=", len(rows))
    f, y, x = np.argwhere(bad)[0]
    print("first:", f, y, x, a[f, y, max(0, x - 2):x + 4], b[f, y, max(0, x - 2):x + 4])
if bad.any():
    for f in range(a.shape[0]):
        for y in range(h):
            if bad[f, y].any():
                xs = np.where(bad[f, y])[0]
                print("field", f, "row", y, "first col", xs[0], "n", len(xs))
