"""Developer tool (GPU box): the pre-emphasis family of the BGRA tool (k_encode_fast_pre + k_decode_fast_bk), or its
S-Video family (k_decode_fast_sv), against the oracle at full size, seeded random members:
    python tools/fuzz_catv.py 0 60 [svideo | phase | fullout]"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, ntscsim
import cases
import _libs as L
s0, n = int(sys.argv[1]), int(sys.argv[2])
family = sys.argv[3] if len(sys.argv) > 3 else "catv"
t0, bad, forms = time.time(), [], {}
for seed in range(s0, s0 + n):
    r = random.Random(70000 + seed)
    if family == "catv": f = ["-vhs", r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3", "-comp-catv4"])]
    elif family == "svideo": f = ["-vhs", "-vhs-svideo", "1"]
    elif family == "phase":     # scanline phases of either parity: k_encode_fast_xi + k_decode_fast_xi
        f = ["-vhs", "-comp-phase", r.choice(["0", "90", "180", "270"]), "-comp-phase-offset", str(r.choice([1, 3]) if r.random() < 0.5 else r.randint(0, 3))]
        if f[2] in ("0", "180") and int(f[4]) % 2 == 0: f[4] = str(int(f[4]) + 1)
    else: f = ["-vhs", "-out-composite-lowpass-lite", "0"]          # the full output low-pass: k_decode_fast_fo
    if r.random() < 0.3: f = ["-tvstd", "pal"] + f
    if r.random() < 0.5: f += ["-vhs-speed", r.choice(["sp", "lp", "ep"])]
    if r.random() < 0.4: f += ["-noise", str(r.randint(1, 9))]
    if r.random() < 0.4: f += ["-chroma-noise", str(r.randint(1, 9))]
    if r.random() < 0.3: f += ["-vhs-head-switching-point", "%.4f" % r.uniform(0.001, 0.12), "-vhs-head-switching-phase", "%.4f" % r.uniform(0.0005, 0.01)]
    if r.random() < 0.2: f += ["-vhs-chroma-vblend", "0"]
    if r.random() < 0.2: f += ["-chroma-dropout", str(r.randint(2, 9))]
    pal = "pal" in f
    w, h = (720, 576) if pal else ((720, 486) if seed % 3 else (720, 480))
    p = L.make_params(f)
    src = cases.make_source("noise", w, h, seed)
    o = L.OracleStream(p)
    want = np.full((h, w, 4), 9, np.uint8)
    sim = ntscsim.FieldSimulator(params=p)
    got = np.full((h, w, 4), 9, np.uint8)
    for (si, field, fieldno) in cases.case_jobs(2):
        o.field(want, src, field, fieldno, 0, 0)
        sim.field_host(got, src, field, fieldno, 0, 0)
    k = tuple(x for x in sim.last_kernels() if x.startswith(("k_encode", "k_decode")))
    forms[k] = forms.get(k, 0) + 1
    if not np.array_equal(got, want):
        bad.append((seed, f))
    sim.close()
print({"catv": "pre-emphasis", "svideo": "S-Video", "phase": "any-phase", "fullout": "full output low-pass"}[family] + " family at full size: %d random members x 2 fields in %.0f s, %d failures" % (n, time.time() - t0, len(bad)))
for k, v in sorted(forms.items(), key=lambda kv: -kv[1]):
    print("  %4d x %s" % (v, " + ".join(k)))
for b in bad[:8]:
    print(b)
