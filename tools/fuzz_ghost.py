"""Developer tool (GPU box): the ghosting extension on random taps / switch sets / geometries, HIP == oracle byte for byte;
census of the encoder forms that ran (k_encode_fast_gh<.,2|4> = folded into the encoder, k_ghost = a pass of its own).
Delays straddle the fold's limit (63 samples); widths straddle the 16-pixel chunking (row start, chunks, row end).
    python tools/fuzz_ghost.py 80000 600"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np
import torch  # before libntscsim.so: see ntscsim/_capi.py lib()
import _libs as L
import cases
import ntscsim
import test_gpu_parity as T

s0, n = int(sys.argv[1]), int(sys.argv[2])
bad, t0, census = [], time.time(), {}
FLAGS = [[], ["-vhs"], ["-vhs"], ["-vhs", "-vhs-speed", "ep"], ["-vhs", "-vhs-svideo", "1"], ["-vhs", "-comp-catv2"],
         ["-vhs", "-comp-phase", "90"], ["-tvstd", "pal", "-vhs"], ["-vhs", "-noise", "40"], ["-vhs", "-vhs-head-switching-point", "0.3"],
         ["-vhs", "-out-composite-lowpass-lite", "0"], ["-nocolor-subcarrier"]]
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    flags = r.choice(FLAGS)
    w = r.choice([16, 20, 36, 52, 64, 96, 100, 112, 180, 256, 720])
    h = r.choice([8, 17, 32, 66]) if w < 700 else 24
    nf = r.choice([2, 3, 5])
    nt = r.randint(1, 4)
    short = r.random() < 0.7
    taps = []
    for _ in range(nt):
        d = r.choice([1, 2, 3, 15, 16, 17, 31, 47, 48, 62, 63, r.randint(1, 63)]) if short else r.choice([63, 64, 65, 100, w - 1, w, w + 5, 4096, r.randint(1, 200)])
        taps.append((max(1, d), r.choice([-256, 256, 255, -1, 1, 0, r.randint(-256, 256)])))
    try:
        p = T._ghost_params(flags, taps)
    except Exception as e:
        bad.append((seed, flags, "params", repr(e))); continue
    kind = r.choice(["noise", "bars"])
    srcs = [L.noise_frame(w, h, seed * 5 + j) if kind == "noise" else L.bars(w, h, j) for j in range((nf + 1) // 2)]
    jobs = cases.case_jobs(nf)
    o = L.OracleStream(p)
    exp = np.zeros((nf, h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(jobs):
        o.field(exp[k], srcs[si], field, fieldno)
    sim = ntscsim.FieldSimulator(params=p)
    try:
        got = T.run_hip(p, srcs, jobs, h, w, per_field_dst=True, sim=sim)
        form = [k for k in sim.last_kernels() if k.startswith(("k_encode", "k_ghost"))]
        census[" + ".join(form)] = census.get(" + ".join(form), 0) + 1
        if not np.array_equal(got, exp):
            bad.append((seed, flags, w, h, nf, taps, form))
    except Exception as e:
        bad.append((seed, flags, w, h, nf, taps, repr(e)[:160]))
    sim.close()
print("%d random ghosting cases in %.1f s, %d failures" % (n, time.time() - t0, len(bad)))
for k in sorted(census, key=lambda k: -census[k]):
    print("  %5d  %s" % (census[k], k))
for b in bad[:10]:
    print(b)
