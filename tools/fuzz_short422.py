"""Developer tool (GPU box): the YUV422P tool's round-5 kernel forms (k422_short: no VCR; k422_fused_sv<D>: S-Video out) against
the oracle on random switch sets, geometries and row alignments, with a census of the forms that ran.
    python tools/fuzz_short422.py 60000 1500"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np
import torch
import _libs as L
import cases422
import ntscsim
import test_variant422 as T

s0, n = int(sys.argv[1]), int(sys.argv[2])
forms, bad, t0 = {}, [], time.time()
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    f = []
    if r.random() < 0.45:
        f += ["-vhs", "-vhs-svideo", "1"]
        if r.random() < 0.4: f += ["-vhs-speed", r.choice(["lp", "ep"])]
        if r.random() < 0.15: f += ["-vhs-chroma-vblend", "0"]
    if r.random() < 0.12: f = ["-tvstd", "pal"] + f
    if r.random() < 0.2: f.append(r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3"]))
    if r.random() < 0.25: f += ["-noise", str(r.choice([0, 1, 9, 30]))]
    if r.random() < 0.35: f += ["-chroma-noise", str(r.choice([0, 5, 16, 64]))]
    if r.random() < 0.35: f += ["-chroma-phase-noise", str(r.choice([0, 1, 4, 25]))]
    if r.random() < 0.3: f += ["-chroma-dropout", str(r.choice([0, 2000, 60000]))]
    if r.random() < 0.25: f += ["-out-composite-lowpass", "0"] + (["-out-composite-lowpass-lite", "0"] if r.random() < 0.5 else [])
    if r.random() < 0.2: f += ["-comp-phase", r.choice(["0", "90", "180", "270"])]
    if r.random() < 0.15: f += ["-comp-phase-offset", str(r.randrange(4))]
    if r.random() < 0.15: f += ["-subcarrier-amp", str(r.choice([30, 50, 80]))]
    if r.random() < 0.2: f += ["-vhs-head-switching", "1"]
    w = r.choice([64, 96, 128, 130, 178, 320, 720])
    h = r.choice([6, 17, 38, 63, 130]) if w < 720 else r.choice([38, 480])
    pad = r.choice([0, 2, 6, 16, 32])
    nfields = r.choice([2, 3, 4])
    try:
        p = L.make_params_tocomp(f)
        srcs = [cases422.make_source422(r.choice(["noise", "bars"]), w, h, j + seed, pad) for j in range((nfields + 1) // 2)]
        o = L.TocompOracleStream(p, L.OOB_MEMORY)
        frame = srcs[0].copy()
        mask = T.last_row_margin_mask(frame, pad)
        sim = ntscsim.FieldSimulator(params=p)
        whole, dev = T.to_dev_onebuf(torch, frame)
        for k in range(nfields):
            field = (k & 1) ^ 1
            T.refresh(frame, srcs[k // 2], field)
            o.process(frame, field, k)
            _, srcd = T.to_dev_onebuf(torch, srcs[k // 2])
            sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
            sim.sync()
            got = whole.cpu().numpy()
            miss = (got != frame.buf) & mask
            if miss.any():
                bad.append((seed, f, w, h, pad, k, int(miss.sum())))
                break
            frame.buf[~mask] = got[~mask]
        for kn in sim.last_kernels():
            if kn.startswith("k422"):
                forms[kn] = forms.get(kn, 0) + 1
        if sim.rng_pos != o.rng_pos:
            bad.append((seed, f, "rng_pos"))
        sim.close()
    except ntscsim.NtscsimError as e:
        forms["refused:%d" % e.code] = forms.get("refused:%d" % e.code, 0) + 1
print("%d random switch sets in %.1f s, %d failures; forms: %s" % (n, time.time() - t0, len(bad), dict(sorted(forms.items()))))
for b in bad[:10]:
    print(b)
