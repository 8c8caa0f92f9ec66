"""Developer tool (GPU box): how long ONE launch of n fields takes as wavefront roles (k_field_pipe, NTSCSIM_PIPE_ALWAYS=1)
and as the one-launch chain (k_encode_fast + k_decode_fast), device-resident frames, 720x486 -vhs, n = 1 ... 128.
    NTSCSIM_PIPE_ALWAYS=1 python tools/pipe_scaling_probe.py ; python tools/pipe_scaling_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, torch
import _libs as L
import ntscsim
w, h = 720, 486
p = L.make_params(sys.argv[1:] or ["-vhs"])
src = torch.from_numpy(np.stack([L.noise_frame(w, h, 40 + j) for j in range(4)])).cuda()
print("forced roles" if os.environ.get("NTSCSIM_PIPE_ALWAYS") == "1" else "one-launch chain")
for n in (1, 2, 4, 8, 16, 32, 64, 128, 192, 256, 384, 600):
    sim = ntscsim.FieldSimulator(params=p)
    dst = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    jobs = [((k // 2) % 4, k, (k & 1) ^ 1, k) for k in range(n)]
    descs = sim.build_descs(src, dst, jobs)
    for _ in range(5): sim.run_descs(descs, w, h)
    sim.sync()
    t0 = time.perf_counter()
    reps = 30
    for _ in range(reps): sim.run_descs(descs, w, h)
    sim.sync()
    dt = (time.perf_counter() - t0) / reps
    print("  n %4d: %8.1f us per launch, %7.1f us per field   %s" % (n, dt * 1e6, dt * 1e6 / n, ",".join(k for k in sim.last_kernels() if "setup" not in k and "states" not in k)))
    sim.close()
