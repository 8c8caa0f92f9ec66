"""Developer tool (GPU box): HIP against the oracles at 720x486 / 720x480 with seeded random switch sets,
both tools, two fields each:  python tools/fuzz_fullsize.py 0 300"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, torch, ntscsim
import test_fuzz_params as T, cases, cases422, test_variant422 as V
import _libs as L
s0, n = int(sys.argv[1]), int(sys.argv[2])
t0, bad = time.time(), []
for seed in range(s0, s0 + n):
    w, h = (720, 486) if seed % 3 else (720, 480)
    f, _, _, _, kind, il, tff = T.draw(40000 + seed)
    p = L.make_params(f)
    src = cases.make_source(kind, w, h, seed)
    o = L.OracleStream(p)
    want = np.full((h, w, 4), 9, np.uint8)
    sim = ntscsim.FieldSimulator(params=p)
    got = np.full((h, w, 4), 9, np.uint8)
    for (si, field, fieldno) in cases.case_jobs(2):
        o.field(want, src, field, fieldno, il, tff)
        sim.field_host(got, src, field, fieldno, il, tff)
    if not np.array_equal(got, want):
        bad.append(("ntsc", seed, f))
    sim.close()
    f, _, _, _, kind = T.draw422(50000 + seed)
    p = L.make_params_tocomp(f)
    pad = 32 if seed & 1 else 0
    fr = cases422.make_source422(kind, w, h, seed, pad)
    mask = V.last_row_margin_mask(fr, pad)
    oo = L.TocompOracleStream(p, L.OOB_MEMORY)
    sim = ntscsim.FieldSimulator(params=p)
    whole, dev = V.to_dev_onebuf(torch, fr)
    for k in range(2):
        oo.process(fr, (k & 1) ^ 1, k)
        sim.fields422([{"dst": dev, "field": (k & 1) ^ 1, "fieldno": k}], w, h)
        sim.sync()
        g = whole.cpu().numpy()
        if ((g != fr.buf) & mask).any():
            bad.append(("tocomp", seed, f, k)); break
        if pad < 2:
            fr.buf[~mask] = g[~mask]
    sim.close()
print("full size: %d random switch sets x 2 tools x 2 fields in %.0f s, %d failures" % (n, time.time() - t0, len(bad)))
for b in bad[:8]:
    print(b)
