"""Developer probe (GPU box): the BGRA tool -- N fields in ONE launch == the same fields in launches of 8, per mode?
(the companion of tools/halo_race_probe.py, which found the in-place race of the YUV422P tool)
    python tools/batch_invariance_probe.py [fields]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, torch
import _libs as L
import ntscsim
from ntscsim import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
w, h = 720, 486
for flags in (["-vhs"], [], ["-vhs", "-vhs-svideo", "1"], ["-vhs", "-tvstd", "pal"]):
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 40 + j) for j in range(4)]
    src = torch.from_numpy(np.stack(srcs)).cuda()
    for mode, name in ((_capi.MODE_EXACT, "exact"), (_capi.MODE_FLOAT, "float")):
        outs = []
        for batch in (8, n):
            sim = ntscsim.FieldSimulator(params=p)
            sim.set_mode(mode)
            dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
            jobs = [((k // 2) % 4, k, (k & 1) ^ 1, k) for k in range(n)]
            sim.rng_pos = 0
            for a in range(0, n, batch):
                sim.fields(src, dst, jobs[a:a + batch])
            sim.sync()
            kern = sim.last_kernels()
            sim.close()
            outs.append(dst)
        diff = (outs[0] != outs[1]).flatten(1).any(dim=1).nonzero().flatten().tolist()
        print("%-28s %-5s %d fields in one launch (%s) vs launches of 8: %d fields differ %s" % (" ".join(flags) or "default", name, n, ",".join(kern), len(diff), diff[:5]))
        del outs, dst
