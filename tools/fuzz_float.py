"""Developer tool (GPU box): NTSCSIM_MODE_FLOAT on random switch sets / geometries / sources against the oracle: every channel of
every pixel within +-1 LSB, rows of the other field and alpha untouched, rand() position equal; census of the decoder forms.
    python tools/fuzz_float.py 90000 1500"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np
import torch
import _libs as L
import ntscsim
from ntscsim import _capi

s0, n = int(sys.argv[1]), int(sys.argv[2])
bad, forms, worst, t0 = [], {}, 0.0, time.time()
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    f = list(r.choice([["-vhs"], [], ["-vhs", "-vhs-speed", "ep"], ["-vhs", "-vhs-speed", "lp"], ["-vhs", "-tvstd", "pal"],
                       ["-vhs", "-vhs-svideo", "1"], ["-vhs", "-comp-phase", "90"], ["-vhs", "-comp-catv"], ["-noise", "9"],
                       ["-vhs", "-vhs-chroma-vblend", "0"], ["-vhs", "-chroma-dropout", "20000"], ["-vhs", "-noise", "30", "-chroma-noise", "60"]]))
    if r.random() < 0.2:
        f += ["-vhs-head-switching-point", "%.3f" % r.uniform(0.3, 0.97)]
    w = r.choice([16, 20, 36, 64, 100, 256, 360, 720]) if r.random() < 0.8 else r.choice([33, 97, 130])
    h = r.choice([2, 5, 9, 17, 34, 63, 130])
    nf = r.choice([1, 2, 3, 5])
    try:
        p = L.make_params(f)
    except Exception:
        continue
    srcs = [L.noise_frame(w, h, seed * 7 + j) for j in range((nf + 1) // 2)]
    o = L.OracleStream(p)
    exp = np.full((nf, h, w, 4), 0x55, np.uint8)
    for k in range(nf):
        o.field(exp[k], srcs[k // 2], (k & 1) ^ 1, k)
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_mode(_capi.MODE_FLOAT)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.full((nf, h, w, 4), 0x55, dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(nf)])
    sim.sync()
    got = dst.cpu().numpy()
    kern = [k_ for k_ in sim.last_kernels() if k_.startswith("k_dec")]
    forms[kern[0] if kern else "?"] = forms.get(kern[0] if kern else "?", 0) + 1
    ok = sim.rng_pos == o.rng_pos
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    ok = ok and d.max() <= 1
    for k in range(nf):
        fld = (k & 1) ^ 1
        ok = ok and (got[k][1 - fld::2] == 0x55).all() and not got[k][fld::2, :, 3].any()
        if w >= 256 and h >= 34:
            worst = max(worst, float((d[k][fld::2].max(axis=-1) > 0).mean()))
    sim.close()
    if not ok:
        bad.append((seed, f, w, h, nf, int(d.max())))
        print("FAIL", bad[-1], flush=True)
print("%d cases, %d failures, decoder forms %s, largest share of differing pixels (noise frames >= 256 x 34) %.4f, %.0f s" % (
    n, len(bad), forms, worst, time.time() - t0))
