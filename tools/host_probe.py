import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntscsim
from ntscsim import shard
from bench import make_bars_clip
w, h, nfr = 720, 486, 300
dev = torch.device("cuda", 0)
p = ntscsim.make_params(["-vhs"])
jobs = shard.jobs_for_rank(p, w, h, 2 * nfr, 0, 1)
src = make_bars_clip(torch, nfr, w, h, 0, 1, dev)
loc = [(cur // 2, cur // 2, f, fn) for (cur, f, fn, _) in jobs]
sim = ntscsim.FieldSimulator(params=p)
dst = torch.zeros((nfr, h, w, 4), dtype=torch.uint8, device=dev)
descs = sim.build_descs(src, dst, loc, rng_pos=[j[3] for j in jobs])
st = torch.cuda.Stream(dev)
for _ in range(3): sim.run_descs(descs, w, h, stream=st.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): sim.run_descs(descs, w, h, stream=st.cuda_stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue ms/call", (t1 - t0) / 20 * 1e3, "total ms/call", (t2 - t0) / 20 * 1e3)
