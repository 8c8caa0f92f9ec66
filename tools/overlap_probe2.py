"""Developer probe: step pipelining over N contexts, torch streams vs the contexts' own streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ntscsim, ctypes as C
from ntscsim import shard
from bench import make_bars_clip
w, h, nfr = 720, 486, 300
dev = torch.device("cuda", 0)
p = ntscsim.make_params(["-vhs"])
jobs = shard.jobs_for_rank(p, w, h, 2 * nfr, 0, 1)
src = make_bars_clip(torch, nfr, w, h, 0, 1, dev)
loc = [(cur // 2, cur // 2, f, fn) for (cur, f, fn, _) in jobs]
mode = sys.argv[1]
for ns in [int(a) for a in sys.argv[2:]] or (1, 2, 3, 4):
    sims = [ntscsim.FieldSimulator(params=p) for _ in range(ns)]
    dsts = [torch.zeros((nfr, h, w, 4), dtype=torch.uint8, device=dev) for _ in range(ns)]
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    descs = [s.build_descs(src, d, loc, rng_pos=[j[3] for j in jobs]) for s, d in zip(sims, dsts)]
    torch.cuda.synchronize()
    def run(K):
        for i in range(K):
            q = i % ns
            if mode == "torch":
                sims[q].run_descs(descs[q], w, h, stream=streams[q].cuda_stream)
            else:
                rc = sims[q]._lib.ntscsim_fields_device(sims[q]._h, descs[q], len(descs[q]), w, h, None)
                assert rc == 0
    run(4); torch.cuda.synchronize()
    K = 24
    t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "contexts", ns, "fields/s", 2 * nfr * K / dt, "ms/step", dt / K * 1e3, flush=True)
    for s in sims: s.close()
