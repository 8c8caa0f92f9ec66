"""Developer probe (GPU box): the raw-composite decoder on 600-field synthetic captures of different noise levels --
how many links the closed-form warm-up leaves to the repair rounds, and what that costs (tools/raw28_probe.py is the
noise-3 case; results never depend on any of it: tests/test_raw28.py)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch, ntscsim
import _libs as L
for noise in [int(v) for v in os.environ.get("RAW28_PROBE_NOISE", "0,3,6,12,24").split(",")]:
    base = L.raw28_capture(30, 5, noise, 0)
    capture = np.ascontiguousarray(np.tile(base[:30 * 477750], 20)[250000:])
    dec = ntscsim.Raw28Decoder([])
    if os.environ.get("RAW28_PROBE_WARM"):
        dec.set_speculation(int(os.environ["RAW28_PROBE_WARM"]), 0)
    cap = torch.from_numpy(capture).cuda()
    fr = torch.empty((602, dec.height, dec.width * 4), dtype=torch.uint8, device="cuda")
    n = dec.decode(cap, fr)
    t0 = time.perf_counter()
    for _ in range(3):
        dec.decode(cap, fr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = dec.stats()
    print("noise %2d: %d fields in %.2f ms = %.0f fields/s; repair rounds %d, chunks repaired %d, front end %d us; %d sync runs, walk %d us, tail rounds %d, redo %d us" %
          (noise, n, dt * 1e3, n / dt, st["front_rounds"], st["chunks_repaired"], st["us_front"], st["sync_runs"], st["us_walk"], st["tail_rounds"], st["us_redo"]))
    dec.close()
