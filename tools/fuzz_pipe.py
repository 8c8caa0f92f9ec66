"""Developer tool (GPU box): the synchronous drop-in ntscsim_field() on random geometries and switch sets of the -vhs family
against the oracle, call after call -- the five-role workgroup form (k_field_pipe, csrc/ntsc_pipe.hip) wherever the launcher
takes it, the other forms elsewhere; a census of the forms is printed.
Random: width 16..800 (aligned / odd), height 2..300, row padding, destination pageable / pinned (written in place) /
declared, tape speed, noise levels, head-switch point and phase, dropout, vertical blend, sharpen, NTSC / PAL, interlaced
sources, the field order.
    python tools/fuzz_pipe.py 1000 400"""
import collections, os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np
import _libs as L
import ntscsim
from ntscsim import host_alloc_array, host_free_array

s0, n = int(sys.argv[1]), int(sys.argv[2])
bad, t0 = [], time.time()
census = collections.Counter()
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    flags = ["-vhs"]
    if r.random() < 0.5: flags += ["-vhs-speed", r.choice(["sp", "lp", "ep"])]
    if r.random() < 0.4: flags += ["-noise", str(r.choice([0, 1, 2, 7, 40, 100]))]
    if r.random() < 0.4: flags += ["-chroma-noise", str(r.choice([1, 3, 70, 200]))]
    if r.random() < 0.3: flags += ["-chroma-phase-noise", str(r.choice([1, 2, 9, 30]))]
    if r.random() < 0.4: flags += ["-vhs-head-switching-point", "%.4f" % r.uniform(0.6, 1.0)]
    if r.random() < 0.3: flags += ["-vhs-head-switching-phase", "%.4f" % r.uniform(0.0, 0.2)]
    if r.random() < 0.2: flags += ["-vhs-head-switching", "0"]
    if r.random() < 0.15: flags.remove("-vhs")           # the default preset family: three roles
    if r.random() < 0.3: flags += ["-chroma-dropout", str(r.choice([100, 3000, 30000, 90000]))]
    if r.random() < 0.2: flags += ["-vhs-chroma-vblend", "0"]
    if r.random() < 0.15: flags += ["-tvstd", "pal"]
    if r.random() < 0.1: flags += ["-vhs-svideo", "1"]
    if r.random() < 0.1: flags += ["-comp-phase", r.choice(["0", "90", "180", "270"])]
    if len(sys.argv) > 3 and sys.argv[3] == "catv" and r.random() < 0.6: flags += [r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3", "-comp-catv4"])]
    w = r.choice([16, 17, 20, 33, 36, 64, 100, 180, 256, 333, 360, 640, 720, 800]) if r.random() < 0.7 else r.randrange(16, 801)
    h = r.choice([2, 3, 9, 63, 64, 65, 126, 127, 128, 243, 244, 300]) if r.random() < 0.7 else r.randrange(2, 301)
    try:
        p = L.make_params(flags, output_height=h)
    except Exception as e:
        census["rejected switches"] += 1
        continue
    pad = r.choice([0, 0, 4, 12, 16, 64])
    kind = r.choice(["pageable", "pinned", "pinned", "declared"])
    il, tff = r.choice([(0, 0), (1, 0), (1, 1)])
    try:
        sim = ntscsim.FieldSimulator(params=p)
    except ntscsim.NtscsimError:
        census["rejected switches"] += 1      # (outside the supported domain: ntscsim_params_validate)
        continue
    o = L.OracleStream(p)
    rowb = w * 4 + pad
    nbytes = rowb * h
    blk = None
    if kind == "pinned":
        blk = host_alloc_array((nbytes + 64,))
        base = blk[r.choice([0, 16, 32, 4]):][:nbytes]
    elif kind == "declared":
        import mmap            # (a mapping of its own: ntscsim_host_pin refuses pages of the brk heap)
        raw = np.frombuffer(mmap.mmap(-1, nbytes + 3 * 4096), np.uint8)
        a0 = (-raw.ctypes.data) % 4096
        sim.host_pin(raw[a0:a0 + ((nbytes + 4095) // 4096) * 4096])
        base = raw[a0:a0 + nbytes]
    else:
        base = np.zeros((nbytes + 64,), np.uint8)[r.choice([0, 4, 16]):][:nbytes]
    base[:] = 0x5A
    got = np.lib.stride_tricks.as_strided(base, shape=(h, w, 4), strides=(rowb, 4, 1))
    sblk, src_pin = None, None
    if r.random() < 0.5:
        sblk = host_alloc_array((w * 4 * h + 64,))
        src_pin = sblk[r.choice([0, 16, 32, 4]):][:w * 4 * h].reshape(h, w, 4)
    exp = np.full((h, w, 4), 0x5A, np.uint8)
    order = r.choice([(1, 0), (0, 1)])
    ok = True
    try:
        for k in range(r.choice([2, 4, 5])):
            s = L.noise_frame(w, h, seed * 5 + k // 2) if r.random() < 0.8 else L.bars(w, h, k)
            if src_pin is not None:           # a pinned source frame: read in place by the encoder role
                src_pin[:] = s
                s = src_pin
            field = order[k & 1]
            rc = sim._lib.ntscsim_field(sim._h, s.ctypes.data_as(ntscsim.C.POINTER(ntscsim.C.c_uint8)), s.strides[0], il, tff,
                                        got.ctypes.data_as(ntscsim.C.POINTER(ntscsim.C.c_uint8)), rowb, w, h, field, k)
            sim._chk(rc, "ntscsim_field")
            o.field(exp, s, field, k, interlaced=il, tff=tff)
            if not np.array_equal(got, exp) or sim.rng_pos != o.rng_pos:
                bad.append((seed, flags, w, h, pad, kind, il, tff, "call %d" % k)); ok = False; break
        kern = sim.last_kernels()
        census[next((x for x in kern if "k_field_pipe" in x), None) or ",".join(x for x in kern if "setup" not in x)] += 1
        if pad and ok:
            padv = np.lib.stride_tricks.as_strided(base[w * 4:], shape=(h - 1, pad), strides=(rowb, 1))
            if not (padv == 0x5A).all(): bad.append((seed, "row padding written", flags, w, h, pad, kind))
    finally:
        sim.close()
        if blk is not None: host_free_array(blk)
        if sblk is not None: host_free_array(sblk)
print("fuzz_pipe: %d cases from seed %d, %d failures, %.0f s" % (n, s0, len(bad), time.time() - t0))
for k, v in census.most_common(): print("  %5d  %s" % (v, k))
for b in bad[:20]: print("  FAIL", b)
sys.exit(1 if bad else 0)
