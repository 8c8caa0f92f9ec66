"""The ghosting extension folded into the encoder vs as its own pass: fields/s of the -vhs preset with 0 / 2 / 4 taps.
   python tools/ghost_probe.py            (NTSCSIM_DEBUG_DECODE=8 python tools/ghost_probe.py = k_ghost for every delay)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "composite-video-simulator_amd"))
import torch
import ntscsim
import bench_side

dev = torch.device("cuda:0")
w, h, frames = 720, 486, 300
for name, taps in (("vhs", ()), ("ghost2", ((12, 64), (31, -32))), ("ghost4", ((9, 80), (31, -40), (47, 24), (63, -12))),
                   ("ghost2_long", ((12, 64), (144, -32)))):
    prm = ntscsim.make_params(["-vhs"])
    prm.ghost_taps = len(taps)
    for k, (d, g) in enumerate(taps):
        prm.ghost_delay[k], prm.ghost_gain[k] = d, g
    kn = []
    v = max(bench_side.device_rate(torch, ntscsim, dev, 0, None, w, h, frames, 24, 4, params=prm, kernels=kn) for _ in range(2))
    print("%-12s %9.0f /s  %s" % (name, v, [k for k in kn if not k.startswith(("k_field", "k_row"))]), flush=True)
