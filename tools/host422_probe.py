import sys, os, time, argparse
sys.path.insert(0,'composite-video-simulator_amd'); sys.path.insert(0,'.')
import torch, ntscsim, bench
args = argparse.Namespace(width=720, height=486, preset="-vhs", frames=300)
dev = torch.device("cuda", 0)
sims, vstep, _ = bench.variant_contexts(torch, ntscsim, dev, 0, args, 4)
for i in range(8): vstep(i)
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(40): vstep(i)
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print("host enqueue per step %.3f ms; total per step %.3f ms" % ((t1-t0)/40*1e3, (t2-t0)/40*1e3))
