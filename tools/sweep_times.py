"""Developer tool (GPU box): wave-clock time per sweep of k422_fused, from an A/B build with
-DF422_AB_TIMES (tools/build_variants.sh times "-DF422_AB_TIMES"):
   NTSCSIM_LIB=tools/bin/variants/lib_times.so python tools/sweep_times.py [inflight]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import argparse, torch, ntscsim, bench
args = argparse.Namespace(width=720, height=486, preset="-vhs", frames=300)
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
sims, vstep, _ = bench.variant_contexts(torch, ntscsim, dev, 0, args, nq)
lib = ntscsim.lib()
out = (C.c_ulonglong * 8)()
for i in range(2 * nq): vstep(i)
torch.cuda.synchronize()
lib.ntscsim_debug_422_times(out, 1)
n = 5 * nq
for i in range(n): vstep(i)
torch.cuda.synchronize()
lib.ntscsim_debug_422_times(out, 0)
waves = 2315 * n
names = ["A", "head switch", "B1", "B2", "B3"]
tot = sum(out[k] for k in range(5))
for k in range(5):
    print("%-12s %9.0f clock ticks per wave  %5.1f %%" % (names[k], out[k] / waves, 100.0 * out[k] / tot))
print("total %.0f ticks per wave, %d steps in flight" % (tot / waves, nq))
