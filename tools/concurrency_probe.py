"""Developer probe (GPU box): four contexts in flight on four streams (the bench's arrangement) == the same four launches one
after the other, byte for byte?  Both tools, exact mode, 600 fields per launch.
    python tools/concurrency_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, torch
import _libs as L
import ntscsim

n, NC = 600, 4
lib = L.product()
def bgra(concurrent):
    w, h = 720, 486
    p = L.make_params(["-vhs"])
    src = torch.from_numpy(np.stack([L.noise_frame(w, h, 40 + j) for j in range(4)])).cuda()
    sims = [ntscsim.FieldSimulator(params=p) for _ in range(NC)]
    streams = [torch.cuda.Stream() for _ in range(NC)]
    dsts = [torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(NC)]
    torch.cuda.synchronize()
    for rep in range(2):
        for c in range(NC):
            with torch.cuda.stream(streams[c]):
                sims[c].rng_pos = 1000 * c
                sims[c].fields(src, dsts[c], [((k // 2 + c) % 4, k, (k & 1) ^ 1, k) for k in range(n)])
            if not concurrent:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    for s in sims: s.close()
    return dsts
def v422(concurrent):
    w, h = 720, 480
    p = L.make_params_tocomp(["-vhs"])
    srcs = [L.yuv_noise(w, h, 70 + j) for j in range(4)]
    base = [[torch.from_numpy(np.ascontiguousarray(s.plane(i))).cuda() for i in range(3)] for s in srcs]
    sims = [ntscsim.FieldSimulator(params=p) for _ in range(NC)]
    streams = [torch.cuda.Stream() for _ in range(NC)]
    devs = [[[t.clone() for t in base[(k // 2 + c) % 4]] for k in range(n)] for c in range(NC)]
    jobs = []
    for c in range(NC):
        pos, jl = 1000 * c, []
        for k in range(n):
            field = (k & 1) ^ 1
            jl.append({"dst": devs[c][k], "field": field, "fieldno": k, "rng_pos": pos})
            pos += lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, field)
        jobs.append(jl)
    torch.cuda.synchronize()
    for c in range(NC):
        with torch.cuda.stream(streams[c]):
            sims[c].fields422(jobs[c], w, h, stream=streams[c].cuda_stream)
        if not concurrent:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for s in sims: s.close()
    return devs
a, b = bgra(False), bgra(True)
bad = [(c, int((a[c] != b[c]).flatten(1).any(dim=1).sum())) for c in range(NC) if not torch.equal(a[c], b[c])]
print("BGRA tool    4 x %d fields, four streams in flight vs one after the other: %s" % (n, "identical" if not bad else "DIFFER %s" % bad))
del a, b
a, b = v422(False), v422(True)
bad = [(c, k, i) for c in range(NC) for k in range(n) for i in range(3) if not torch.equal(a[c][k][i], b[c][k][i])]
print("YUV422P tool 4 x %d fields, four streams in flight vs one after the other: %s" % (n, "identical" if not bad else "DIFFER %s" % bad[:6]))
