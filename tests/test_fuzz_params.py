"""Seeded random sweeps over the tool's switches and frame geometries.

CPU (-m "not gpu"): the oracle against the reference's own composite_layer() (oracle/_ref) --
widens the pin of the restatement beyond the hand-written case matrix.
GPU (-m gpu): the HIP path against the oracle on the same draws.  Bit-exact (tolerance 0)."""
import random

import numpy as np
import pytest

import _libs as L
import cases

N_REF = 200         # oracle vs reference draws (CPU, small frames)
N_GPU = 200         # HIP vs oracle draws


def draw(seed):
    """One random configuration: (flags, W, H, n_fields, source kind, interlaced, tff)."""
    r = random.Random(seed)
    f = []
    if r.random() < 0.6:
        f.append("-vhs")
    if r.random() < 0.35:
        f += ["-vhs-speed", r.choice(["sp", "lp", "ep"])]
    if r.random() < 0.15:
        f += ["-tvstd", "pal"]
    if r.random() < 0.3:
        f += ["-comp-phase", r.choice(["0", "90", "180", "270"])]
    if r.random() < 0.3:
        f += ["-comp-phase-offset", str(r.randrange(0, 4))]
    if r.random() < 0.3:
        f.append(r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3", "-comp-catv4"]))
    elif r.random() < 0.15:
        f += ["-comp-pre", "%.3f" % r.uniform(0.2, 2.5), "-comp-cut", str(r.randrange(300000, 3000000))]
    if r.random() < 0.35:
        f += ["-noise", str(r.choice([0, 1, 2, 7, 30, 200]))]
    if r.random() < 0.35:
        f += ["-chroma-noise", str(r.choice([0, 1, 5, 16, 64, 300]))]
    if r.random() < 0.35:
        f += ["-chroma-phase-noise", str(r.choice([0, 1, 4, 25, 90]))]
    if r.random() < 0.3:
        f += ["-chroma-dropout", str(r.choice([0, 4, 2000, 30000, 99999]))]
    if r.random() < 0.25:
        f += ["-subcarrier-amp", str(r.choice([10, 30, 50, 75, 120]))]
    if r.random() < 0.2:
        f += ["-in-composite-lowpass", str(r.randrange(0, 2))]
    if r.random() < 0.25:
        f += ["-out-composite-lowpass", str(r.randrange(0, 2))]
    if r.random() < 0.25:
        f += ["-out-composite-lowpass-lite", str(r.randrange(0, 2))]
    if r.random() < 0.12:
        f.append("-nocolor-subcarrier")
    if r.random() < 0.2:
        f += ["-vhs-svideo", str(r.randrange(0, 2))]
    if r.random() < 0.2:
        f += ["-vhs-chroma-vblend", str(r.randrange(0, 2))]
    if r.random() < 0.25:
        # head switch inside a small frame: point in rows, phase in columns
        f += ["-vhs-head-switching", "1", "-vhs-head-switching-point", "%.4f" % r.uniform(0.09, 0.125),
              "-vhs-head-switching-phase", "%.5f" % r.uniform(0.0003, 0.0035)]
        if r.random() < 0.4:
            f += ["-vhs-head-switching-noise-level", r.choice(["0", "0.000005", "0.00002"])]
    w = r.choice([16, 17, 31, 64, 65, 96, 100, 127, 130])
    h = r.choice([2, 3, 7, 16, 31, 32, 33, 40])
    n = r.randrange(1, 5)
    kind = r.choice(["noise", "noise", "bars", "ramp", "impulse"])
    if kind in ("ramp", "impulse") and (w < 4 or h < 3):
        kind = "noise"
    il = r.randrange(0, 2)
    tff = r.randrange(0, 2)
    return f, w, h, n, kind, il, tff


def test_draws_are_valid_and_varied():
    seen = set()
    for s in range(max(N_REF, N_GPU)):
        f, w, h, n, kind, il, tff = draw(s)
        L.make_params(f)                      # must parse and validate
        seen.add(tuple(f))
    assert len(seen) > 0.8 * max(N_REF, N_GPU)


@pytest.mark.skipif(not L.have_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(N_REF))
def test_oracle_equals_reference_on_random_parameters(seed):
    f, w, h, n, kind, il, tff = draw(seed)
    p = L.make_params(f)
    srcs = [cases.make_source(kind, w, h, j + seed) for j in range((n + 1) // 2)]
    o, r = L.OracleStream(p), L.RefStream(p)
    do = np.full((h, w, 4), 9, np.uint8)
    dr = np.full((h, w, 4), 9, np.uint8)
    for (si, field, fieldno) in cases.case_jobs(n):
        if field >= h:
            continue
        r.field(dr, srcs[si], field, fieldno, il, tff)
        o.field(do, srcs[si], field, fieldno, il, tff)
        assert np.array_equal(do, dr), (f, w, h, fieldno)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_GPU))
def test_hip_equals_oracle_on_random_parameters(seed):
    import torch
    import ntscsim
    f, w, h, n, kind, il, tff = draw(1000 + seed)
    p = L.make_params(f)
    srcs = [cases.make_source(kind, w, h, j + seed) for j in range((n + 1) // 2)]
    o = L.OracleStream(p)
    want = np.full((h, w, 4), 9, np.uint8)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.full((1, h, w, 4), 9, dtype=torch.uint8, device="cuda")
    for (si, field, fieldno) in cases.case_jobs(n):
        if field >= h:
            continue
        o.field(want, srcs[si], field, fieldno, il, tff)
        sim.fields(src, dst, [(si, 0, field, fieldno)], interlaced=il, tff=tff)
        sim.sync()
        assert np.array_equal(dst[0].cpu().numpy(), want), (f, w, h, fieldno)
    assert sim.rng_pos == o.rng_pos
    sim.close()


# The BGRA tool's hand-tuned families beside the two presets: -vhs with pre-emphasis (-comp-catv*: k_encode_fast_pre +
# k_decode_fast_bk) and -vhs with S-Video out (k_decode_fast_sv), random tape speed / PAL / noise levels / head-switch
# point / blend / dropout -- the draws of tools/fuzz_catv.py at small sizes with 16-byte aligned rows.
N_BGRA_FAMILY = 48


def draw_bgra_family(seed):
    r = random.Random(90000 + seed)
    sv = seed % 3 == 2
    f = ["-vhs", "-vhs-svideo", "1"] if sv else ["-vhs", r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3", "-comp-catv4"])]
    if r.random() < 0.3: f = ["-tvstd", "pal"] + f
    if r.random() < 0.5: f += ["-vhs-speed", r.choice(["sp", "lp", "ep"])]
    if r.random() < 0.4: f += ["-noise", str(r.randint(1, 9))]
    if r.random() < 0.4: f += ["-chroma-noise", str(r.randint(1, 9))]
    if r.random() < 0.3: f += ["-vhs-head-switching-point", "%.4f" % r.uniform(0.001, 0.12), "-vhs-head-switching-phase", "%.4f" % r.uniform(0.0005, 0.01)]
    if r.random() < 0.2: f += ["-vhs-chroma-vblend", "0"]
    if r.random() < 0.2: f += ["-chroma-dropout", str(r.randint(2, 9))]
    w = r.choice([16, 32, 64, 96, 100, 128, 132])          # (4 w is a multiple of 16: the hand-tuned kernels' rows)
    h = r.choice([2, 3, 7, 16, 31, 32, 38, 40])
    n = r.randrange(1, 5)
    kind = r.choice(["noise", "noise", "bars", "ramp"])
    if kind == "ramp" and (w < 4 or h < 3):
        kind = "noise"
    return f, w, h, n, kind, ("k_decode_fast_sv<double>" if sv else "k_decode_fast_bk<true,double>")


@pytest.mark.skipif(not L.have_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(N_BGRA_FAMILY))
def test_bgra_family_oracle_equals_reference(seed):
    f, w, h, n, kind, _ = draw_bgra_family(seed)
    p = L.make_params(f)
    srcs = [cases.make_source(kind, w, h, j + seed) for j in range((n + 1) // 2)]
    o, r = L.OracleStream(p), L.RefStream(p)
    do = np.full((h, w, 4), 9, np.uint8)
    dr = np.full((h, w, 4), 9, np.uint8)
    for (si, field, fieldno) in cases.case_jobs(n):
        if field >= h:
            continue
        r.field(dr, srcs[si], field, fieldno, 0, 0)
        o.field(do, srcs[si], field, fieldno, 0, 0)
        assert np.array_equal(do, dr), (f, w, h, fieldno)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_BGRA_FAMILY))
def test_hip_bgra_family_equals_oracle(seed):
    """the same draws on the GPU; the hand-tuned form the launcher must pick is asserted by name"""
    import torch
    import ntscsim
    f, w, h, n, kind, form = draw_bgra_family(seed)
    p = L.make_params(f)
    srcs = [cases.make_source(kind, w, h, j + seed) for j in range((n + 1) // 2)]
    o = L.OracleStream(p)
    want = np.full((h, w, 4), 9, np.uint8)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.full((1, h, w, 4), 9, dtype=torch.uint8, device="cuda")
    for (si, field, fieldno) in cases.case_jobs(n):
        if field >= h:
            continue
        o.field(want, srcs[si], field, fieldno, 0, 0)
        sim.fields(src, dst, [(si, 0, field, fieldno)])
        sim.sync()
        assert np.array_equal(dst[0].cpu().numpy(), want), (f, w, h, fieldno)
        assert form in sim.last_kernels(), (f, sim.last_kernels())
    assert sim.rng_pos == o.rng_pos
    sim.close()


# ------------------------------------------------------------- 8-bit YUV422P variant ----------
N_REF422 = 120
N_GPU422 = 120


def draw422(seed):
    """One random ffmpeg_to_composite configuration: (flags, W (even), H, n_fields, source)."""
    r = random.Random(50000 + seed)
    f = []
    if r.random() < 0.6:
        f.append("-vhs")
    if r.random() < 0.35:
        f += ["-vhs-speed", r.choice(["sp", "lp", "ep"])]
    if r.random() < 0.15:
        f += ["-tvstd", "pal"]
    if r.random() < 0.3:
        f += ["-comp-phase", r.choice(["0", "90", "180", "270"])]
    if r.random() < 0.3:
        f += ["-comp-phase-offset", str(r.randrange(0, 4))]
    if r.random() < 0.3:
        f.append(r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3"]))
    if r.random() < 0.35:
        f += ["-noise", str(r.choice([0, 1, 2, 7, 30]))]
    if r.random() < 0.35:
        f += ["-chroma-noise", str(r.choice([0, 1, 5, 16, 64]))]
    if r.random() < 0.35:
        f += ["-chroma-phase-noise", str(r.choice([0, 1, 4, 25, 90]))]
    if r.random() < 0.3:
        f += ["-chroma-dropout", str(r.choice([0, 4, 2000, 30000, 99999]))]
    if r.random() < 0.25:
        f += ["-subcarrier-amp", str(r.choice([10, 30, 50, 75, 120]))]
    if r.random() < 0.2:
        f += ["-in-composite-lowpass", str(r.randrange(0, 2))]
    if r.random() < 0.25:
        f += ["-out-composite-lowpass", str(r.randrange(0, 2))]
    if r.random() < 0.25:
        f += ["-out-composite-lowpass-lite", str(r.randrange(0, 2))]
    if r.random() < 0.12:
        f.append("-nocolor-subcarrier")
    if r.random() < 0.12:
        f.append("-nocolor-subcarrier-after-yc-sep")
    if r.random() < 0.2:
        f += ["-yc-recomb", str(r.randrange(0, 4))]
    if r.random() < 0.2:
        f += ["-vhs-svideo", str(r.randrange(0, 2))]
    if r.random() < 0.2:
        f += ["-vhs-chroma-vblend", str(r.randrange(0, 2))]
    if r.random() < 0.25:
        f += ["-vhs-head-switching", "1", "-vhs-head-switching-point", "%.4f" % r.uniform(0.09, 0.125)]
    w = r.choice([32, 34, 64, 66, 96, 100, 128, 130])
    h = r.choice([4, 7, 16, 31, 32, 33, 40])
    n = r.randrange(1, 5)
    kind = r.choice(["noise", "noise", "bars"])
    return f, w, h, n, kind


def _refresh(dst, src, field):
    for i in range(3):
        dst.plane(i)[field::2] = src.plane(i)[field::2]


@pytest.mark.skipif(not L.have_tocomp_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(N_REF422))
def test_variant_oracle_equals_reference_on_random_parameters(seed):
    import cases422
    f, w, h, n, kind = draw422(seed)
    p = L.make_params_tocomp(f)
    pad = 32 if seed & 1 else 0
    srcs = [cases422.make_source422(kind, w, h, j + seed, pad) for j in range((n + 1) // 2)]
    a, b = srcs[0].copy(), srcs[0].copy()
    r, o = L.TocompRefStream(p), L.TocompOracleStream(p, L.OOB_MEMORY)
    for k in range(n):
        field = (k & 1) ^ 1
        _refresh(a, srcs[k // 2], field)
        _refresh(b, srcs[k // 2], field)
        r.process(a, field, k)
        o.process(b, field, k)
        assert np.array_equal(a.buf, b.buf), (f, w, h, k)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_GPU422))
def test_variant_hip_equals_oracle_on_random_parameters(seed):
    import torch
    import cases422
    import ntscsim
    f, w, h, n, kind = draw422(7000 + seed)
    p = L.make_params_tocomp(f)
    # reference-memory semantics of the two-byte read past each luma row (ffmpeg_to_composite.cpp
    # :496): odd seeds pad the rows (the read stays inside the plane everywhere), even seeds use
    # linesize == width (the frame's last row reads 16 instead of the neighbouring allocation)
    import test_variant422 as T
    pad = 32 if seed & 1 else 0
    srcs = [cases422.make_source422(kind, w, h, j + seed, pad) for j in range((n + 1) // 2)]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    frame = srcs[0].copy()
    mask = T.last_row_margin_mask(frame, pad)
    sim = ntscsim.FieldSimulator(params=p)
    whole, dev = T.to_dev_onebuf(torch, frame)
    for k in range(n):
        field = (k & 1) ^ 1
        _refresh(frame, srcs[k // 2], field)
        o.process(frame, field, k)
        _, srcd = T.to_dev_onebuf(torch, srcs[k // 2])
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        got = whole.cpu().numpy()
        bad = (got != frame.buf) & mask
        assert not bad.any(), (f, w, h, k, int(bad.sum()))
        frame.buf[~mask] = got[~mask]
        assert sim.rng_pos == o.rng_pos
    sim.close()


# ---- the streamed kernels of the YUV422P tool's VHS family: random geometry, alignment and switch mix
N_FAMILY = 96


def draw422_family(seed):
    """A member of the -vhs family that the launcher runs as sweep A + one streamed pass when the frame rows
    are aligned: about a third of the seeds are exactly the preset (`-vhs`), the rest change a few switches.
    Returns (flags, W, H, pad, n_fields, source, expected kernel form or None for unaligned rows)."""
    r = random.Random(90000 + seed)
    f = ["-vhs"]
    speed, preset = "sp", True
    if r.random() < 0.35:
        speed = r.choice(["lp", "ep"]); f += ["-vhs-speed", speed]; preset = False
    if r.random() < 0.12:
        f = ["-tvstd", "pal"] + f; preset = False
    if r.random() < 0.2:
        f.append(r.choice(["-comp-catv", "-comp-catv2", "-comp-catv3"])); preset = False
    if r.random() < 0.15:
        f += ["-noise", str(r.choice([0, 1, 30]))]; preset = preset and f[-1] != "0"
    if r.random() < 0.15:
        f += ["-chroma-noise", str(r.choice([0, 5, 64]))]; preset = preset and f[-1] != "0"
    if r.random() < 0.15:
        f += ["-chroma-phase-noise", str(r.choice([0, 1, 25]))]; preset = preset and f[-1] != "0"
    if r.random() < 0.2:
        f += ["-chroma-dropout", str(r.choice([0, 2000, 60000]))]
    if r.random() < 0.15:
        f += ["-out-composite-lowpass", "0"]; preset = False
        if r.random() < 0.5:
            f += ["-out-composite-lowpass-lite", "0"]
    if r.random() < 0.12:
        f += ["-subcarrier-amp", str(r.choice([10, 30, 75]))]; preset = False
    if r.random() < 0.12:
        f += ["-comp-phase", r.choice(["0", "90", "270"])]; preset = preset and f[-1] == "0"
    if r.random() < 0.12:
        off = r.randrange(1, 4); f += ["-comp-phase-offset", str(off)]; preset = preset and (off % 2 == 0 or "-comp-phase" in f)
    if r.random() < 0.2:
        f += ["-vhs-chroma-vblend", "0"]
    if r.random() < 0.3:
        f += ["-vhs-head-switching-point", "%.4f" % r.uniform(0.09, 0.125)]
    aligned = r.random() < 0.8
    w = 16 * r.randrange(1, 14) if aligned else r.choice([18, 34, 50, 98, 130, 22])
    pad = r.choice([0, 16, 32]) if aligned else r.choice([0, 3, 16])
    h = r.choice([2, 3, 5, 8, 17, 32, 41])
    n = r.randrange(1, 4)
    kind = r.choice(["noise", "noise", "bars"])
    form = None
    if aligned:
        d = {"sp": 4, "lp": 5, "ep": 6}[speed]
        form = "k422_fused<true,true,4>" if preset else "k422_fused<false,true,%d>" % d
    return f, w, h, pad, n, kind, form


@pytest.mark.skipif(not L.have_tocomp_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed", range(N_FAMILY))
def test_variant_family_oracle_equals_reference(seed):
    """the same draws, oracle == reference extract over the whole buffer (CPU)"""
    import cases422
    f, w, h, pad, n, kind, _ = draw422_family(seed)
    p = L.make_params_tocomp(f)
    srcs = [cases422.make_source422(kind, w, h, j + seed, pad) for j in range((n + 1) // 2)]
    a, b = srcs[0].copy(), srcs[0].copy()
    r, o = L.TocompRefStream(p), L.TocompOracleStream(p, L.OOB_MEMORY)
    for k in range(n):
        field = (k & 1) ^ 1
        _refresh(a, srcs[k // 2], field)
        _refresh(b, srcs[k // 2], field)
        r.process(a, field, k)
        o.process(b, field, k)
        assert np.array_equal(a.buf, b.buf), (f, w, h, pad, k)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_FAMILY))
def test_variant_streamed_family_on_random_geometry(seed):
    """HIP == oracle (reference-memory semantics, whole buffer) for random members of the -vhs family at random
    widths (16 ... 208), heights (2 ... 41), row paddings and sources; with aligned rows the streamed form the
    launcher must pick is asserted by name."""
    import torch
    import cases422
    import ntscsim
    import test_variant422 as T
    f, w, h, pad, n, kind, form = draw422_family(seed)
    p = L.make_params_tocomp(f)
    srcs = [cases422.make_source422(kind, w, h, j + seed, pad) for j in range((n + 1) // 2)]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    frame = srcs[0].copy()
    mask = T.last_row_margin_mask(frame, pad)
    sim = ntscsim.FieldSimulator(params=p)
    whole, dev = T.to_dev_onebuf(torch, frame)
    for k in range(n):
        field = (k & 1) ^ 1
        _refresh(frame, srcs[k // 2], field)
        o.process(frame, field, k)
        _, srcd = T.to_dev_onebuf(torch, srcs[k // 2])
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        ran = [x for x in sim.last_kernels() if x.startswith("k422_fused") or x == "k422_process"]
        al = all(o_ % a == 0 and l_ % a == 0 for o_, l_, a in zip(frame.off, frame.ls, (16, 8, 8)))
        if form is not None and al:
            assert ran == [form], (f, w, h, pad, ran)
        got = whole.cpu().numpy()
        bad = (got != frame.buf) & mask
        assert not bad.any(), (f, w, h, pad, k, int(bad.sum()), ran)
        frame.buf[~mask] = got[~mask]
        assert sim.rng_pos == o.rng_pos
    sim.close()
