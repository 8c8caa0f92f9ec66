"""Developer probe: the raw-composite decoder on a 600-field synthetic capture resident in HBM (the
bench's `raw28` leg on its own; run under rocprofv3 by tools/refresh_profiles.sh)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, ntscsim
import _libs as L
base = L.raw28_capture(30, 5, 3, 0)
capture = np.ascontiguousarray(np.tile(base[:30 * 477750], 20)[250000:])
dec = ntscsim.Raw28Decoder([])
if os.environ.get("RAW28_PROBE_WARM"):  # warm-up scanlines only (the chunk length stays the decoder's choice)
    dec.set_speculation(int(os.environ["RAW28_PROBE_WARM"]), 0)
if len(sys.argv) > 2:      # raw28_probe.py <warm-up scanlines> <chunk samples>: speculation settings (results must not change)
    dec.set_speculation(int(sys.argv[1]), int(sys.argv[2]))
cap = torch.from_numpy(capture).cuda()
fr = torch.empty((602, dec.height, dec.width * 4), dtype=torch.uint8, device="cuda")
n = dec.decode(cap, fr)
t0 = time.perf_counter()
for _ in range(3):
    dec.decode(cap, fr)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print("%d fields in %.2f ms = %.0f fields/s; %s" % (n, dt * 1e3, n / dt, dec.stats()))
