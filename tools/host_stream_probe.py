"""Developer probe: end-to-end (PCIe-inclusive) throughput of ntscsim_frames_host."""
import os, sys, time
import numpy as np
import torch
torch.zeros(1, device='cuda')
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import ntscsim, _libs as L
w, h, n = 720, 486, 300
p = ntscsim.make_params(["-vhs"])
src = np.stack([L.bars(w, h, j) for j in range(n)])
dst = np.zeros((2 * n, h, w, 4), np.uint8)
sim = ntscsim.FieldSimulator(params=p)
sim.frames_host(dst[:8], src[:4])
for ch in (16, 32, 64, 150, 300):
    sim.rng_pos = 0
    t0 = time.perf_counter()
    sim.frames_host(dst, src, first_fieldno=0, chunk_frames=ch)
    dt = time.perf_counter() - t0
    print("chunk %d frames: %d fields in %.3f s = %.0f fields/s (H2D %.1f MB + D2H %.1f MB => %.1f GB/s combined)" % (
        ch, 2 * n, dt, 2 * n / dt, src.nbytes / 1e6, dst.nbytes / 1e6, (src.nbytes + dst.nbytes) / dt / 1e9))

# same with caller-pinned buffers (hipHostRegister inside the call then fails harmlessly)
import torch
tsrc = torch.empty(src.shape, dtype=torch.uint8).pin_memory()
tdst = torch.empty(dst.shape, dtype=torch.uint8).pin_memory()
tsrc.numpy()[:] = src
for ch in (16, 32, 64, 150, 300):
    sim.rng_pos = 0
    t0 = time.perf_counter()
    sim.frames_host(tdst.numpy(), tsrc.numpy(), first_fieldno=0, chunk_frames=ch)
    dt = time.perf_counter() - t0
    print("pinned, chunk %d frames: %.0f fields/s (%.1f GB/s combined)" % (ch, 2 * n / dt, (src.nbytes + dst.nbytes) / dt / 1e9))
assert (tdst.numpy() == dst).all()

# encoder pixel format made on the GPU: planar YUV 4:2:0 out (1.5 B/pixel instead of 4)
fb = w * h + 2 * (w // 2) * ((h + 1) // 2)
tyuv = torch.empty((2 * n, fb), dtype=torch.uint8).pin_memory()
for ch in (16, 32, 64):
    sim.rng_pos = 0
    t0 = time.perf_counter()
    sim.frames_host(tyuv.numpy(), tsrc.numpy(), first_fieldno=0, chunk_frames=ch, yuv="420")
    dt = time.perf_counter() - t0
    print("pinned, YUV420P out, chunk %d frames: %.0f fields/s (%.1f GB/s combined)" % (
        ch, 2 * n / dt, (src.nbytes + tyuv.numel()) / dt / 1e9))
