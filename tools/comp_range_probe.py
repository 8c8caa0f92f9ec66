import sys
sys.path.insert(0, "composite-video-simulator_amd"); sys.path.insert(0, "tests")
import numpy as np, torch, ntscsim
import _libs as L
w, h = 720, 486
for name, fl in (("-vhs", ["-vhs"]), ("default", []), ("-vhs -comp-catv3", ["-vhs", "-comp-catv3"])):
    p = L.make_params(fl)
    for kind in ("bars", "noise"):
        src = np.stack([L.bars(w, h, 0) if kind == "bars" else L.noise_frame(w, h, 5)])
        sim = ntscsim.FieldSimulator(params=p)
        s = torch.from_numpy(src).cuda(); d = torch.zeros((1, h, w, 4), dtype=torch.uint8, device="cuda")
        sim.fields(s, d, [(0, 0, 1, 0)]); sim.sync()
        c = sim.debug_composite(1, w, h)
        print("%-18s %-6s composite plane min %d max %d  (int16: -32768 .. 32767)" % (name, kind, int(c.min()), int(c.max())))
        sim.close()
