"""Developer probe (GPU box): does a LARGE in-place YUV422P batch (more workgroups than the chip holds at once) give the same
bytes as the same fields in small batches?  A workgroup's halo lane re-computes the row above -- a row its neighbour
workgroup rewrites in place -- so a workgroup that starts after its neighbour has finished would read output instead of input.
    python tools/halo_race_probe.py [fields]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, torch
import _libs as L
import ntscsim

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
w, h = 720, 480
FLAGS = [["-vhs"], [], ["-vhs", "-vhs-svideo", "1"], ["-vhs", "-vhs-speed", "ep", "-tvstd", "pal"], ["-vhs", "-yc-recomb", "1"]]
lib = L.product()
srcs = [L.yuv_noise(w, h, 70 + j) for j in range(4)]
p = None
base = [[torch.from_numpy(np.ascontiguousarray(s.plane(i))).cuda() for i in range(3)] for s in srcs]
def run(batch):
    devs = [[t.clone() for t in base[(k // 2) % 4]] for k in range(n)]
    jobs, pos = [], 0
    for k in range(n):
        field = (k & 1) ^ 1
        jobs.append({"dst": devs[k], "field": field, "fieldno": k, "rng_pos": pos})
        pos += lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, field)
    sim = ntscsim.FieldSimulator(params=p)
    for a in range(0, n, batch):
        sim.fields422(jobs[a:a + batch], w, h)
    sim.sync()
    kern = sim.last_kernels()
    sim.close()
    return devs, kern
for flags in FLAGS:
  p = L.make_params_tocomp(flags)
  print("==", " ".join(flags) or "default")
  small, _ = run(8)
  for trial in range(2):
    big, kern = run(n)
    bad = 0
    first = None
    for k in range(n):
        for i in range(3):
            if not torch.equal(big[k][i], small[k][i]):
                bad += 1
                if first is None:
                    d = (big[k][i] != small[k][i]).nonzero()
                    first = (k, i, d[0].tolist(), int(d.shape[0]))
    del big
    print("trial %d: %d fields in ONE batch (%s) vs batches of 8: %d planes differ%s" % (trial, n, ",".join(kern), bad, "" if first is None else "  first: field %d plane %d at %s (%d bytes)" % first))
