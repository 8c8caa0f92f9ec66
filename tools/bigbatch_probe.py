"""Developer probe: one very large batch (2400 fields) -- index-width sanity + balance."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntscsim, _libs as L
from ntscsim import shard
from bench import make_bars_clip
w, h, nfr = 720, 486, 1200
dev = torch.device("cuda", 0)
p = ntscsim.make_params(["-vhs"])
jobs = shard.jobs_for_rank(p, w, h, 2 * nfr, 0, 1)
src = make_bars_clip(torch, nfr, w, h, 0, 1, dev)
dst = torch.zeros((nfr, h, w, 4), dtype=torch.uint8, device=dev)
sim = ntscsim.FieldSimulator(params=p)
loc = [(cur // 2, cur // 2, f, fn) for (cur, f, fn, _) in jobs]
d = sim.build_descs(src, dst, loc, rng_pos=[j[3] for j in jobs])
plan = sim.prepare(d, w, h)
sim.run_prepared(plan); sim.sync()
t0 = time.perf_counter()
for _ in range(5): sim.run_prepared(plan)
sim.sync(); dt = (time.perf_counter() - t0) / 5
print("2400 fields in one batch: %.2f ms, %.0f fields/s" % (dt * 1e3, 2 * nfr / dt))
# compare with four 600-field batches and with the oracle on scattered fields
dst2 = torch.zeros_like(dst)
for q in range(4):
    sub = loc[q * 600:(q + 1) * 600]
    sim.fields(src, dst2, sub, rng_pos=[j[3] for j in jobs[q * 600:(q + 1) * 600]])
sim.sync()
print("equal to 4 x 600:", bool(torch.equal(dst, dst2)))
got = dst.cpu().numpy()
for k in (0, 1199, 1200, 2399):
    o = L.OracleStream(p); o.skip(jobs[k][3])
    e = np.zeros((h, w, 4), np.uint8)
    field = (k & 1) ^ 1
    o.field(e, src[k // 2].cpu().numpy(), field, k)
    print("field", k, "oracle match:", bool(np.array_equal(got[k // 2][field::2], e[field::2])))
