"""The 8-bit YUV422P sibling (ffmpeg_to_composite.cpp): oracle pinning (CPU), product parity (GPU).

Chain of evidence:
  reference extract  ==  oracle in MEMORY mode      (bit for bit, whole buffer; needs /root/reference)
  reference fixtures ==  oracle in MEMORY mode      (tests/golden/tocomp_golden.npz, runs everywhere)
  oracle MEMORY      ==  HIP path                   (bit for bit, GPU tests; the reference's two-byte
                                                     read past each luma row, :496, returns the
                                                     caller's own bytes wherever they lie inside the
                                                     luma plane, i.e. every row but a frame's last
                                                     when linesize < W + 2 -- there it returns 16)
  reference fixtures ==  HIP path                   (GPU, same exclusion)
  oracle MEMORY      ~=  oracle DEFINED             (the all-16 variant differs only in a bounded
                                                     right margin; kept as a documented property)
"""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import cases422
import ntscsim
from ntscsim import _capi

MARGIN_Y, MARGIN_C = 24, 14   # samples at the right edge that the out-of-row read can influence


def refresh(frame, src, field):
    """What render_field does for a same-size progressive source: overwrite the field's rows."""
    for i in range(3):
        frame.plane(i)[field::2] = src.plane(i)[field::2]


@pytest.mark.skipif(not L.have_tocomp_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("c", cases422.CASES422, ids=[c[0] for c in cases422.CASES422])
@pytest.mark.parametrize("pad", [0, 32])
def test_oracle_equals_reference_extract(c, pad):
    name, flags, w, h, n, kind = c
    p = L.make_params_tocomp(flags)
    srcs = [cases422.make_source422(kind, w, h, j, pad) for j in range((n + 1) // 2)]
    a, b, d = srcs[0].copy(), srcs[0].copy(), srcs[0].copy()
    r, o, od = L.TocompRefStream(p), L.TocompOracleStream(p, L.OOB_MEMORY), L.TocompOracleStream(p, L.OOB_DEFINED)
    for k in range(n):
        field = (k & 1) ^ 1
        for fr in (a, b, d):
            refresh(fr, srcs[k // 2], field)
        r.process(a, field, k)
        o.process(b, field, k)
        od.process(d, field, k)
        assert np.array_equal(a.buf, b.buf), "field %d" % k          # incl. padding bytes
        # the defined-behaviour oracle differs from the reference only inside the right margin
        assert np.array_equal(b.pix(0)[:, :w - MARGIN_Y], d.pix(0)[:, :w - MARGIN_Y])
        for i in (1, 2):
            assert np.array_equal(b.pix(i)[:, :w // 2 - MARGIN_C], d.pix(i)[:, :w // 2 - MARGIN_C])
        assert o.rng_pos == od.rng_pos
    lib = L.product()
    exp = sum(lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, (k & 1) ^ 1) for k in range(n))
    assert o.rng_pos == exp


@pytest.mark.skipif(not L.have_tocomp_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("is420,il,tff,second,sh", [(0, 0, 0, 0, 32), (0, 0, 0, 0, 50), (1, 0, 0, 0, 48),
                                                     (0, 1, 1, 0, 48), (0, 1, 1, 1, 48), (0, 1, 0, 0, 40),
                                                     (1, 1, 0, 1, 64), (1, 1, 1, 0, 36)])
def test_render_field_equals_reference_extract(is420, il, tff, second, sh):
    w, h = 64, 32
    rng = np.random.RandomState(sh)
    src = L.Yuv422(w, sh)
    src.buf[:] = rng.randint(0, 256, size=src.buf.shape, dtype=np.uint8)
    for field in (0, 1):
        a, b = L.Yuv422(w, h, fill=7), L.Yuv422(w, h, fill=7)
        da, la = a.ptr_arrays()
        ds, ls = src.ptr_arrays()
        pp = C.POINTER(C.POINTER(C.c_uint8))
        L.tocomp_ref().tocomp_ref_render_field(C.cast(da, pp), la, w, h, C.cast(ds, pp), ls, sh, is420,
                                               il, tff, second, field)
        L.tocomp_oracle_render_field(b, src, is420, il, tff, second, field)
        for i in range(3):
            assert np.array_equal(a.pix(i), b.pix(i)), (field, i)


@pytest.mark.skipif(not L.have_tocomp_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_black_key_equals_reference_extract():
    w, h = 64, 16
    p = L.make_params_tocomp(["-bkey-feedback", "6"])
    assert p.black_key_level_feedback == 6
    L.tocomp_ref().tocomp_ref_set_params(C.byref(p))
    rng = np.random.RandomState(3)
    for field in (0, 1):
        d = L.Yuv422(w, h); f = L.Yuv422(w, h)
        d.buf[:] = rng.randint(10, 40, size=d.buf.shape, dtype=np.uint8)
        d.pix(1)[:] = rng.randint(120, 136, size=d.pix(1).shape, dtype=np.uint8)
        d.pix(2)[:] = rng.randint(120, 136, size=d.pix(2).shape, dtype=np.uint8)
        f.buf[:] = rng.randint(0, 256, size=f.buf.shape, dtype=np.uint8)
        d2, f2 = d.copy(), f.copy()
        pp = C.POINTER(C.POINTER(C.c_uint8))
        da, la = d.ptr_arrays(); fa, lf = f.ptr_arrays()
        L.tocomp_ref().tocomp_ref_black_key_feedback(C.cast(da, pp), la, C.cast(fa, pp), lf, w, h, field)
        L.tocomp_oracle_black_key(d2, f2, field, 6)
        assert np.array_equal(d.buf, d2.buf) and np.array_equal(f.buf, f2.buf)


def _bob_frame(w, h, mode, rng):
    """An encoder frame for output_frame(): YUV422P for mode 0, else YUV420P ((h+1)//2 chroma
    rows; held in a Yuv422 of full height so the helpers apply), pre-filled with a pattern."""
    b = L.Yuv422(w, h, pad=0)
    b.buf[:] = rng.randint(0, 256, size=b.buf.shape, dtype=np.uint8)
    return b


_OUT_CASES = [(w, h, mode, field) for (w, h) in ((64, 32), (48, 15), (32, 6), (40, 3))
              for mode in (0, 1, 2) for field in (0, 1)]


@pytest.mark.skipif(not L.have_tocomp_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("w,h,mode,field", _OUT_CASES)
def test_output_frame_equals_reference_extract(w, h, mode, field):
    rng = np.random.RandomState(w * 100 + h * 7 + mode * 2 + field)
    frame = L.yuv_noise(w, h, 5)
    a, b = _bob_frame(w, h, mode, rng), None
    b = a.copy()
    pp = C.POINTER(C.POINTER(C.c_uint8))
    ba, bl = a.ptr_arrays(); fa, fl = frame.ptr_arrays()
    L.tocomp_ref().tocomp_ref_output_frame(C.cast(ba, pp), bl, C.cast(fa, pp), fl, w, h, field, mode)
    L.tocomp_oracle_output_frame(b, frame, field, mode)
    assert np.array_equal(a.buf, b.buf)


def test_output_frame_properties():
    """Size-independent properties of the bob copy: every luma row of the bob frame is a row of
    the same parity as `field` (or the identity for interlaced 4:2:0); 4:2:0 chroma row j is the
    frame's chroma row sy(2j)."""
    w, h = 64, 31
    frame = L.yuv_noise(w, h, 9)
    for field in (0, 1):
        b = L.Yuv422(w, h)
        L.tocomp_oracle_output_frame(b, frame, field, 0)
        for y in range(h):
            sy = (y | 1) if field else ((y + 1) & ~1)
            if sy >= h:
                sy -= 2
            assert sy % 2 == field
            for i in range(3):
                assert np.array_equal(b.pix(i)[y], frame.pix(i)[sy])
        b = L.Yuv422(w, h)
        L.tocomp_oracle_output_frame(b, frame, field, 1)
        for j in range((h + 1) // 2):
            sy = ((2 * j) | 1) if field else ((2 * j + 1) & ~1)
            if sy >= h:
                sy -= 2
            assert np.array_equal(b.pix(1)[j], frame.pix(1)[sy])
        b = L.Yuv422(w, h)
        L.tocomp_oracle_output_frame(b, frame, field, 2)
        assert np.array_equal(b.pix(0), frame.pix(0))


import json as _json
import os as _os
_G422 = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")
_GOLD422 = np.load(_os.path.join(_G422, "tocomp_golden.npz"))
_MAN422 = _json.load(open(_os.path.join(_G422, "tocomp_golden.json")))["cases"]


@pytest.mark.parametrize("m", _MAN422, ids=[m["name"] for m in _MAN422])
def test_oracle_reproduces_reference_golden(m):
    """Runs everywhere (also where /root/reference is absent): the oracle in MEMORY mode against
    whole-buffer snapshots recorded from the reference extract by tests/golden/make_golden422.py."""
    name, w, h, n = m["name"], m["w"], m["h"], m["n"]
    p = L.make_params_tocomp(m["flags"])
    fr = L.Yuv422(w, h)
    fr.buf[:] = _GOLD422["%s__init" % name]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    for k in range(n):
        field = (k & 1) ^ 1
        src = L.Yuv422(w, h)
        src.buf[:] = _GOLD422["%s__src%d" % (name, k // 2)]
        refresh(fr, src, field)
        o.process(fr, field, k)
        assert np.array_equal(fr.buf, _GOLD422["%s__after%d" % (name, k)]), "field %d" % k


def test_oracle_plane_mode_is_memory_mode_inside_the_plane():
    """TOCOMP_OOB_PLANE -- the product's contract for the separator's read past a row (:496): the caller's bytes where
    they lie inside the luma plane, 16 where they do not -- equals the reference's literal read (MEMORY mode) on padded
    rows everywhere, and on tight rows everywhere except the right margin of the frame's last row."""
    for flags in (["-vhs"], [], ["-vhs", "-vhs-speed", "ep", "-vhs-svideo", "1"]):
        p = L.make_params_tocomp(flags)
        for pad in (0, 1, 2, 32):
            a = cases422.make_source422("noise", 96, 20, 3, pad)
            b = a.copy()
            oa, ob = L.TocompOracleStream(p, L.OOB_MEMORY), L.TocompOracleStream(p, L.OOB_PLANE)
            for k in range(4):
                oa.process(a, (k & 1) ^ 1, k)
                ob.process(b, (k & 1) ^ 1, k)
            assert oa.rng_pos == ob.rng_pos
            if pad >= 2:
                assert np.array_equal(a.buf, b.buf), (flags, pad)
            else:
                diff = a.buf != b.buf
                for i in range(3):
                    n = a.ls[i] * a.h
                    d = diff[a.off[i]:a.off[i] + n].reshape(a.h, a.ls[i])
                    assert not d[:a.h - 1].any(), (flags, pad, i)          # only the last row may differ


def test_to_composite_flag_mirror():
    p = L.make_params_tocomp([])
    assert p.vhs_head_switching_phase == 1.0 - ((4.5 + 0.01) / 262.5)          # :274
    assert p.vhs_head_switching_phase_noise == (1.0 / 300) / 262.5             # :275
    assert p.vhs_out_sharpen_chroma == 0.85 and p.black_key_level_feedback == -1
    p = L.make_params_tocomp(["-comp-catv2"])                                   # :1429-1433, :1627
    assert (p.composite_preemphasis, p.composite_preemphasis_cut) == (2.5, 315000000 // 88 // 2)
    assert p.subcarrier_amplitude_back == int(50 + (50 * 2.5) / 4)
    p = L.make_params_tocomp(["-vhs-head-switching-point", "0.25"])            # sets the PHASE here
    assert p.vhs_head_switching_phase == 0.25
    for bad in (["-d", "2"], ["-comp-catv4"], ["-vhs-head-switching-phase", "0.1"]):
        with pytest.raises(ntscsim.NtscsimError):
            L.make_params_tocomp(bad)
    for ok in (["-ss", "1"], ["-vp"], ["-an"], ["-bkey-feedback", "3"], ["-t", "5"]):
        L.make_params_tocomp(ok)


# ------------------------------------------------------------------------------------- GPU -----

def to_dev(torch, frame):
    return [torch.from_numpy(np.ascontiguousarray(frame.plane(i))).cuda() for i in range(3)]


def to_dev_onebuf(torch, frame):
    """The frame's single host buffer (Y | U | V + slack) as ONE device tensor, planes as views of
    it with the host linesizes -- the reference's memory layout in the fixtures."""
    whole = torch.from_numpy(frame.buf.copy()).cuda()
    views = []
    for i in range(3):
        n = frame.ls[i] * frame.h
        views.append(whole[frame.off[i]:frame.off[i] + n].view(frame.h, frame.ls[i]))
    return whole, views


def last_row_margin_mask(frame, pad):
    """True where HIP must equal the reference.  The only exclusion: the right margin of the LAST
    luma row (and of the chroma row beside it) when the two bytes behind that row are outside the
    luma plane (linesize < W + 2) -- the reference reads the neighbouring allocation there."""
    m = np.ones(frame.buf.shape, bool)
    if pad >= 2:
        return m
    w, h = frame.w, frame.h
    for i, marg in ((0, MARGIN_Y), (1, MARGIN_C), (2, MARGIN_C)):
        n = frame.ls[i] * h
        pm = m[frame.off[i]:frame.off[i] + n].reshape(h, frame.ls[i])
        width = w if i == 0 else w // 2
        pm[h - 1, max(0, width - marg):width] = False
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("c", cases422.CASES422, ids=[c[0] for c in cases422.CASES422])
@pytest.mark.parametrize("pad", [0, 32])
def test_hip_equals_oracle_memory(c, pad):
    """HIP == the oracle in MEMORY mode (== the reference extract), whole buffer incl. padding, for
    every row whose out-of-row read stays inside the luma plane."""
    import torch
    name, flags, w, h, n, kind = c
    p = L.make_params_tocomp(flags)
    srcs = [cases422.make_source422(kind, w, h, j + 5, pad) for j in range((n + 1) // 2)]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    frame = srcs[0].copy()
    mask = last_row_margin_mask(frame, pad)
    sim = ntscsim.FieldSimulator(params=p)
    whole, dev = to_dev_onebuf(torch, frame)
    for k in range(n):
        field = (k & 1) ^ 1
        refresh(frame, srcs[k // 2], field)
        o.process(frame, field, k)
        _, srcd = to_dev_onebuf(torch, srcs[k // 2])
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        got = whole.cpu().numpy()
        bad = (got != frame.buf) & mask
        assert not bad.any(), "field %d: %d bytes differ, first at %d" % (k, int(bad.sum()), int(np.argmax(bad)))
        if pad < 2:
            # keep the oracle's frame in step with the device where the excluded margin differs
            frame.buf[~mask] = got[~mask]
        assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("m", _MAN422, ids=[m["name"] for m in _MAN422])
def test_hip_reproduces_reference_golden(m):
    """HIP against the whole-buffer snapshots recorded from the REFERENCE extract
    (tests/golden/tocomp_golden.npz; linesize == width, one allocation)."""
    import torch
    name, w, h, n = m["name"], m["w"], m["h"], m["n"]
    p = L.make_params_tocomp(m["flags"])
    fr = L.Yuv422(w, h)
    fr.buf[:] = _GOLD422["%s__init" % name]
    mask = last_row_margin_mask(fr, 0)
    sim = ntscsim.FieldSimulator(params=p)
    whole, dev = to_dev_onebuf(torch, fr)
    for k in range(n):
        field = (k & 1) ^ 1
        src = L.Yuv422(w, h)
        src.buf[:] = _GOLD422["%s__src%d" % (name, k // 2)]
        _, srcd = to_dev_onebuf(torch, src)
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        got = whole.cpu().numpy()
        exp = _GOLD422["%s__after%d" % (name, k)]
        bad = (got != exp) & mask
        assert not bad.any(), "field %d: %d bytes differ, first at %d" % (k, int(bad.sum()), int(np.argmax(bad)))
    sim.close()


# (flags, the kernel form the launcher must pick by default; names as rocprofv3 prints them, without blanks).
# `ffmpeg_to_composite -vhs` runs the FULL output chroma low-pass (ffmpeg_to_composite.cpp:278, :948-951):
# that switch set is the preset kernel k422_fused<true,true,4> (sweep A + one streamed pass with the preset's
# arithmetic identities).  The rest of the VHS family takes the same structure with the switches read at run
# time, one instantiation per chroma delay (tape speed): k422_fused<false,true,D>.
_FORM_CASES = [
    (["-vhs"], "k422_fused<true,true,4>"),
    (["-vhs", "-out-composite-lowpass-lite", "0"], "k422_fused<true,true,4>"),   # full low-pass still on
    (["-vhs", "-out-composite-lowpass", "0"], "k422_fused<false,true,4>"),       # lite output low-pass
    (["-vhs", "-out-composite-lowpass", "0", "-out-composite-lowpass-lite", "0"], "k422_fused<false,true,4>"),
    (["-vhs", "-vhs-speed", "lp"], "k422_fused<false,true,5>"),
    (["-vhs", "-vhs-speed", "ep"], "k422_fused<false,true,6>"),
    (["-tvstd", "pal", "-vhs"], "k422_fused<false,true,4>"),
    (["-tvstd", "pal", "-vhs", "-vhs-speed", "ep", "-chroma-noise", "0"], "k422_fused<false,true,6>"),
    (["-vhs", "-comp-catv"], "k422_fused<false,true,4>"),
    (["-vhs", "-noise", "0"], "k422_fused<false,true,4>"),
    # (the preset's filter switches, but not its phase / amplitude identities: streamed with run-time switches;
    # as four sweeps it is still the preset instantiation, which does not use those identities)
    (["-vhs", "-comp-phase", "90", "-subcarrier-amp", "30"], "k422_fused<false,true,4>|k422_fused<true,false,4>"),
    (["-vhs", "-chroma-dropout", "50000", "-chroma-phase-noise", "0"], "k422_fused<false,true,4>"),
    (["-vhs", "-vhs-svideo", "1"], "k422_fused_sv<4>"),                          # round 5: the streamed pass without its re-modulation
    (["-vhs", "-vhs-svideo", "1", "-vhs-speed", "lp", "-chroma-dropout", "30000"], "k422_fused_sv<5>"),
    (["-vhs", "-vhs-svideo", "1", "-vhs-speed", "ep", "-comp-phase", "90", "-out-composite-lowpass", "0"], "k422_fused_sv<6>"),
    ([], "k422_direct_fast"),                                                     # round 5: the default preset, two sweeps
    (["-chroma-noise", "8", "-chroma-phase-noise", "6", "-comp-catv", "-vhs-head-switching", "1"], "k422_direct"),
    (["-tvstd", "pal", "-out-composite-lowpass", "0"], "k422_direct"),
    (["-yc-recomb", "1"], "k422_process"),                                        # what stays on the twelve sweeps
    (["-nocolor-subcarrier"], "k422_process"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("flags,form", [(["-vhs"], "k422_pipe<true,4>"), (["-vhs", "-vhs-speed", "ep"], "k422_pipe<false,6>"),
                                        (["-vhs", "-vhs-svideo", "1"], "k422_pipe_sv<4>"), ([], "k422_direct_pipe"),
                                        (["-yc-recomb", "1"], "k422_process")])
def test_device_resident_launches_take_the_latency_form_when_asked(flags, form):
    """ntscsim_set_launch_form(NTSCSIM_FORM_LATENCY): ntscsim_fields422_device() launches of up to 64 fields run the streamed
    kernels as wavefront roles (k422_pipe / k422_direct_pipe) -- same bytes as the oracle; switch sets the roles do not cover
    keep their form; back to NTSCSIM_FORM_THROUGHPUT the one-launch kernels return."""
    import torch
    w, h, n = 128, 38, 4
    p = L.make_params_tocomp(flags)
    srcs = [cases422.make_source422("noise" if j else "bars", w, h, j + 9, 0) for j in range(n // 2)]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    frame = srcs[0].copy()
    mask = last_row_margin_mask(frame, 0)
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_launch_form(True)
    whole, dev = to_dev_onebuf(torch, frame)
    for k in range(n):
        if k == n - 1:
            sim.set_launch_form(False)
        field = (k & 1) ^ 1
        refresh(frame, srcs[k // 2], field)
        o.process(frame, field, k)
        _, srcd = to_dev_onebuf(torch, srcs[k // 2])
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        ran = sim.last_kernels()
        if k < n - 1:
            assert form in ran, ran
        else:
            assert not any("pipe" in x for x in ran), ran
        got = whole.cpu().numpy()
        bad = (got != frame.buf) & mask
        assert not bad.any(), "field %d: %d bytes differ, first at %d" % (k, int(bad.sum()), int(np.argmax(bad)))
        frame.buf[~mask] = got[~mask]
    assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("flags,form", _FORM_CASES,
                         ids=["vhs", "vhs-lite0", "vhs-litelp", "vhs-nolp", "vhs-lp", "vhs-ep", "pal-vhs", "pal-vhs-ep", "vhs-catv",
                              "vhs-nonoise", "vhs-phase90-amp30", "vhs-dropout", "vhs-svideo", "vhs-svideo-lp-dropout", "vhs-svideo-phase90",
                              "default",
                              "default-noises-catv-hs", "pal-default-lite", "yc-recomb", "nocolor"])
@pytest.mark.parametrize("mode", [0, 1, 2, 4], ids=["default", "twelve-sweep", "general-fused", "preset-four-sweep"])
def test_every_variant_kernel_form_agrees_with_the_oracle(flags, form, mode):
    """The kernel forms of the -vhs family (streamed: k422_fused<true,true,4> for the preset's own switch
    set, k422_fused<false,true,D> for the others; four sweeps: k422_fused<true,false,4> / <false,false,4>;
    twelve sweeps: k422_process) on the same fields: each must equal the oracle, and the form that ran must
    be the one the case names (ntscsim_debug_last_kernels).  mode = the ntscsim_debug_no_fast_decode() bits
    (1: twelve-sweep form, 2: no preset instantiation and no streamed pass, 4: four sweeps instead of the
    streamed pass)."""
    import torch
    w, h, n = 128, 38, 4
    p = L.make_params_tocomp(flags)
    srcs = [cases422.make_source422("noise" if j else "bars", w, h, j + 9, 0) for j in range(n // 2)]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    frame = srcs[0].copy()
    mask = last_row_margin_mask(frame, 0)
    sim = ntscsim.FieldSimulator(params=p)
    sim.debug_no_fast_decode(mode)
    form, _, four = form.partition("|")
    preset, fam = form == "k422_fused<true,true,4>" or four == "k422_fused<true,false,4>", form.startswith("k422_fused<")
    want = form
    if mode == 1:
        want = "k422_process"
    elif mode == 2 and fam:
        want = "k422_fused<false,false,4>"
    elif mode == 4 and fam:
        want = "k422_fused<true,false,4>" if preset else "k422_fused<false,false,4>"
    elif mode == 2 and form.endswith("_fast"):
        want = form[:-5]                       # (the same debug bit keeps the general sweep A of the no-VCR form)
    elif mode in (2, 4) and form.startswith("k422_fused_sv"):
        want = "k422_process"                  # (no streamed pass: the S-Video family falls back to the twelve sweeps)
    whole, dev = to_dev_onebuf(torch, frame)
    for k in range(n):
        field = (k & 1) ^ 1
        refresh(frame, srcs[k // 2], field)
        o.process(frame, field, k)
        _, srcd = to_dev_onebuf(torch, srcs[k // 2])
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        ran = sim.last_kernels()
        assert want in ran and sum(x.startswith(("k422_fused", "k422_process", "k422_direct")) for x in ran) == 1, ran
        got = whole.cpu().numpy()
        bad = (got != frame.buf) & mask
        assert not bad.any(), "field %d: %d bytes differ, first at %d" % (k, int(bad.sum()), int(np.argmax(bad)))
        frame.buf[~mask] = got[~mask]
    assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
def test_422_batch_accepts_views_of_one_allocation_that_never_meet():
    """Side-by-side tiles of one wide allocation (same linesize, pixel columns -- and the separator's two bytes behind
    each row, :496 -- that never meet) write disjoint bytes: one batch, even with the same field parity, == the oracle
    on each view (the byte-range test alone refused them)."""
    import torch
    w, h = 64, 16
    pad = w + 32                                   # luma linesize 2w + 32, chroma linesize w/2 + w + 32
    p = L.make_params_tocomp(["-vhs"])
    lib = L.product()
    pos1 = lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, 1)
    wide = cases422.make_source422("noise", w, h, 11, pad)
    rng = np.random.RandomState(4)
    wide.buf[:] = rng.randint(16, 236, size=wide.buf.size, dtype=np.uint8)     # both tiles (and the gaps) hold pixels
    cols = [w + 16, w // 2 + 16, w // 2 + 16]      # where tile B starts in each plane's rows

    class View:                                    # tile B for the oracle: the same buffer, planes shifted by `cols`
        pass
    vb = View()
    vb.w, vb.h, vb.ls, vb.buf = w, h, list(wide.ls), wide.buf
    vb.off = [wide.off[i] + cols[i] for i in range(3)]
    vb.cplanes = lambda: L.Yuv422.cplanes(vb)
    exp = wide.copy()
    vb_exp = View()
    vb_exp.w, vb_exp.h, vb_exp.ls, vb_exp.buf, vb_exp.off = w, h, list(exp.ls), exp.buf, list(vb.off)
    vb_exp.cplanes = lambda: L.Yuv422.cplanes(vb_exp)
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    o.process(exp, 1, 0)
    o.process(vb_exp, 1, 1)
    whole = torch.from_numpy(wide.buf.copy()).cuda()
    full = [whole[wide.off[i]:wide.off[i] + wide.ls[i] * h].view(h, wide.ls[i]) for i in range(3)]
    tile_a = [full[i][:, :(w if i == 0 else w // 2)] for i in range(3)]
    tile_b = [full[i][:, cols[i]:cols[i] + (w if i == 0 else w // 2)] for i in range(3)]
    sim = ntscsim.FieldSimulator(params=p)
    sim.fields422([{"dst": tile_a, "field": 1, "fieldno": 0, "rng_pos": 0},
                   {"dst": tile_b, "field": 1, "fieldno": 1, "rng_pos": pos1}], w, h)
    sim.sync()
    got = whole.cpu().numpy()
    assert np.array_equal(got, exp.buf), int((got != exp.buf).sum())
    assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
def test_422_batch_refuses_racing_descriptors():
    """Two descriptors of one call on the same destination frame: same field = write-write race; both
    fields with luma rows tighter than width + 2 = the Y/C separator of one field would read bytes the
    other rewrites (ffmpeg_to_composite.cpp:496 vs the tool's sequential loop :1783-1800).  Both are
    refused; with padded rows, or as separate calls, both fields of a frame are fine and equal the oracle."""
    import torch
    w, h = 64, 16
    p = L.make_params_tocomp(["-vhs"])
    lib = L.product()
    pos1 = lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, 1)
    sim = ntscsim.FieldSimulator(params=p)
    tight = cases422.make_source422("noise", w, h, 5, 0)
    _, dev = to_dev_onebuf(torch, tight)
    for jobs in ([{"dst": dev, "field": 1, "fieldno": 0, "rng_pos": 0}, {"dst": dev, "field": 1, "fieldno": 1, "rng_pos": pos1}],
                 [{"dst": dev, "field": 1, "fieldno": 0, "rng_pos": 0}, {"dst": dev, "field": 0, "fieldno": 1, "rng_pos": pos1}]):
        with pytest.raises(ntscsim.NtscsimError) as e:
            sim.fields422(jobs, w, h)
        assert e.value.code == _capi.E_ARG
        assert sim.rng_pos == 0
    # frames that overlap without being the same frame (here: a luma view that starts one row further down in the
    # same allocation, the other field) race in the same way and are refused as well
    wide = cases422.make_source422("noise", w, h + 1, 5, 16)
    _, devw = to_dev_onebuf(torch, wide)
    shifted = [devw[0][1:], devw[1][:h], devw[2][:h]]
    same = [devw[0][:h], devw[1][:h], devw[2][:h]]
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.fields422([{"dst": same, "field": 1, "fieldno": 0, "rng_pos": 0},
                       {"dst": shifted, "field": 0, "fieldno": 1, "rng_pos": pos1}], w, h)
    assert e.value.code == _capi.E_ARG and sim.rng_pos == 0
    # padded rows: one batch with both fields == the oracle's sequential result
    padded = cases422.make_source422("noise", w, h, 5, 16)
    exp = padded.copy()
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    o.process(exp, 1, 0); o.process(exp, 0, 1)
    whole, devp = to_dev_onebuf(torch, padded)
    sim.fields422([{"dst": devp, "field": 1, "fieldno": 0, "rng_pos": 0}, {"dst": devp, "field": 0, "fieldno": 1, "rng_pos": pos1}], w, h)
    sim.sync()
    mask = last_row_margin_mask(padded, 16)
    bad = (whole.cpu().numpy() != exp.buf) & mask
    assert not bad.any(), int(bad.sum())
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w", [16, 18, 30, 62, 64, 66, 126, 722])
@pytest.mark.parametrize("pad", [0, 3, 16])
def test_fused_kernel_widths_and_row_alignments(w, pad):
    """-vhs at widths around the four-sweep kernel's 16 / 64-sample block sizes, with row paddings that
    make luma / chroma rows 16- / 8-byte aligned (k422_fused<true,true,4>, the preset instantiation) or not
    (k422_fused<false,false,4>, four sweeps with scalar row accesses); the form that ran is asserted."""
    import torch
    h, n = 10, 3
    p = L.make_params_tocomp(["-vhs"])
    srcs = [cases422.make_source422("noise", w, h, j + 31 + w, pad) for j in range((n + 1) // 2)]
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    frame = srcs[0].copy()
    mask = last_row_margin_mask(frame, pad)
    sim = ntscsim.FieldSimulator(params=p)
    whole, dev = to_dev_onebuf(torch, frame)
    for k in range(n):
        field = (k & 1) ^ 1
        refresh(frame, srcs[k // 2], field)
        o.process(frame, field, k)
        _, srcd = to_dev_onebuf(torch, srcs[k // 2])
        sim.fields422([{"dst": dev, "src": srcd, "src_height": h, "field": field, "fieldno": k}], w, h)
        sim.sync()
        # aligned luma (16) and chroma (8) rows and plane starts -> the preset instantiation
        al = all(o_ % a == 0 and l_ % a == 0 for o_, l_, a in zip(frame.off, frame.ls, (16, 8, 8)))
        assert ("k422_fused<true,true,4>" if al else "k422_fused<false,false,4>") in sim.last_kernels(), (al, sim.last_kernels())
        got = whole.cpu().numpy()
        bad = (got != frame.buf) & mask
        assert not bad.any(), "field %d: %d bytes differ, first at %d" % (k, int(bad.sum()), int(np.argmax(bad)))
        if pad < 2:
            frame.buf[~mask] = got[~mask]
    assert sim.rng_pos == o.rng_pos
    sim.close()


@pytest.mark.gpu
def test_prepared_batch422_equals_direct_call():
    """ntscsim_batch422_create/run == ntscsim_fields422_device on the same descriptors, twice in a row
    (the second run starts from the frames the first one left, like a direct call would), and the ctx
    rand() position ends where the direct call leaves it."""
    import torch
    w, h, n = 128, 40, 6
    p = L.make_params_tocomp(["-vhs"])
    lib = L.product()
    srcs = [cases422.make_source422("noise", w, h, 50 + j, 0) for j in range(n)]

    def run(prepared):
        sim = ntscsim.FieldSimulator(params=p)
        devs = [to_dev(torch, s_) for s_ in srcs]
        jobs, pos = [], 0
        for k in range(n):
            field = (k & 1) ^ 1
            jobs.append({"dst": devs[k], "field": field, "fieldno": k, "rng_pos": pos})
            pos += lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, field)
        arr = sim.build_descs422(jobs)
        if prepared:
            b = sim.prepare422(arr, w, h)
            assert sim.rng_pos == 0                          # creating a batch does not move the stream
            sim.run_prepared422(b); sim.run_prepared422(b)
            sim.sync()
            sim.free_prepared422(b)
        else:
            sim.run_descs422(arr, w, h); sim.run_descs422(arr, w, h)
            sim.sync()
        out = [[t.cpu().numpy().copy() for t in d] for d in devs]
        rp = sim.rng_pos
        sim.close()
        return out, rp

    a, rpa = run(False)
    b, rpb = run(True)
    assert rpa == rpb and rpa > 0
    for k in range(n):
        for i in range(3):
            assert np.array_equal(a[k][i], b[k][i]), (k, i)


@pytest.mark.gpu
def test_hip_batch_of_fields_full_size():
    """720x480 -vhs, 8 fields in ONE batch (each field its own frame), explicit rand() positions."""
    import torch
    w, h, n = 720, 480, 8
    p = L.make_params_tocomp(["-vhs"])
    lib = L.product()
    srcs = [L.yuv_bars(w, h, j) if j % 2 == 0 else L.yuv_noise(w, h, 70 + j) for j in range(n // 2)]
    exp, jobs, devs = [], [], []
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    pos = 0
    for k in range(n):
        field = (k & 1) ^ 1
        fr = srcs[k // 2].copy()
        o.process(fr, field, k)
        exp.append(fr)
        d = to_dev(torch, srcs[k // 2])
        devs.append(d)
        jobs.append({"dst": d, "field": field, "fieldno": k, "rng_pos": pos})
        pos += lib.ntscsim_rng_calls_per_field_422(C.byref(p), w, h, field)
    sim = ntscsim.FieldSimulator(params=p)
    sim.fields422(jobs, w, h)
    sim.sync()
    for k in range(n):
        for i in range(3):
            # (separate plane tensors with linesize == width: the frame's last row reads 16)
            got, want = devs[k][i].cpu().numpy(), exp[k].pix(i)
            assert np.array_equal(got[:h - 1], want[:h - 1]), (k, i)
            marg = MARGIN_Y if i == 0 else MARGIN_C
            assert np.array_equal(got[h - 1, :want.shape[1] - marg], want[h - 1, :want.shape[1] - marg]), (k, i)
    sim.close()


@pytest.mark.gpu
def test_a_launch_of_more_workgroups_than_the_chip_holds_equals_small_launches():
    """composite_video_process() runs IN PLACE, and a workgroup's halo lane re-computes the row above -- a row of its
    neighbour's.  With more workgroups than the chip holds at once (here 2,667 for 2,048 slots) a workgroup starts after its
    neighbour has rewritten that row; the halo lanes therefore read copies taken before the kernel starts (k422_halo).
    700 fields in ONE launch == the same fields in launches of 50, byte for byte (before the fix: the first row of
    workgroup 2,048 -- field 537, row 288 -- differed; tools/halo_race_probe.py)."""
    import torch
    w, h, n = 720, 480, 700
    p = L.make_params_tocomp(["-vhs"])
    lib = L.product()
    srcs = [L.yuv_noise(w, h, 70 + j) for j in range(4)]
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
        sim.close()
        return devs

    small = run(50)
    for _ in range(2):
        big = run(n)
        bad = [(k, i) for k in range(n) for i in range(3) if not torch.equal(big[k][i], small[k][i])]
        assert not bad, bad[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("is420,il,tff,second,sh", [(0, 0, 0, 0, 50), (1, 0, 0, 0, 48), (0, 1, 1, 1, 48),
                                                     (1, 1, 0, 1, 64)])
def test_hip_render_field_and_black_key(is420, il, tff, second, sh):
    import torch
    w, h = 64, 32
    p = L.make_params_tocomp(["-vhs", "-bkey-feedback", "8"])
    rng = np.random.RandomState(sh)
    src = L.Yuv422(w, sh)
    src.buf[:] = rng.randint(0, 60, size=src.buf.shape, dtype=np.uint8)
    src.pix(1)[:] = rng.randint(118, 138, size=src.pix(1).shape, dtype=np.uint8)
    src.pix(2)[:] = rng.randint(118, 138, size=src.pix(2).shape, dtype=np.uint8)
    dst, flt = L.Yuv422(w, h, fill=9), L.Yuv422(w, h)
    flt.buf[:] = rng.randint(0, 256, size=flt.buf.shape, dtype=np.uint8)
    o = L.TocompOracleStream(p, L.OOB_MEMORY)
    mask = last_row_margin_mask(dst, 0)
    sim = ntscsim.FieldSimulator(params=p)
    (dwhole, dd), (fwhole, fd), (_, sd) = to_dev_onebuf(torch, dst), to_dev_onebuf(torch, flt), to_dev_onebuf(torch, src)
    flags = (_capi.F422_SRC420 if is420 else 0) | (_capi.F422_INTERLACED if il else 0) | \
            (_capi.F422_TFF if tff else 0) | (_capi.F422_SECOND if second else 0)
    for k in range(3):
        field = (k & 1) ^ 1
        L.tocomp_oracle_render_field(dst, src, is420, il, tff, second, field)
        L.tocomp_oracle_black_key(dst, flt, field, 8)
        o.process(dst, field, k)
        sim.fields422([{"dst": dd, "src": sd, "src_height": sh, "flt": fd, "field": field,
                        "fieldno": k, "flags": flags}], w, h)
        sim.sync()
        got = dwhole.cpu().numpy()
        bad = (got != dst.buf) & mask
        assert not bad.any(), (k, int(bad.sum()))
        dst.buf[~mask] = got[~mask]
        assert np.array_equal(fwhole.cpu().numpy(), flt.buf), k
    sim.close()


@pytest.mark.gpu
def test_hip_422_rejects_odd_width():
    import torch
    p = L.make_params_tocomp([])
    sim = ntscsim.FieldSimulator(params=p)
    t = [torch.zeros((8, 64), dtype=torch.uint8, device="cuda") for _ in range(3)]
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.fields422([{"dst": t, "field": 0, "fieldno": 0}], 33, 8)
    assert e.value.code == _capi.E_SIZE
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,mode,field", _OUT_CASES + [(720, 486, 0, 1), (720, 486, 1, 0), (720, 480, 2, 1)])
def test_hip_output_frame(w, h, mode, field):
    import torch
    rng = np.random.RandomState(w + h + mode + field)
    frame = L.yuv_noise(w, h, 11)
    bob = _bob_frame(w, h, mode, rng)
    sim = ntscsim.FieldSimulator(params=L.make_params_tocomp([]))
    fd, bd = to_dev(torch, frame), to_dev(torch, bob)
    L.tocomp_oracle_output_frame(bob, frame, field, mode)
    sim.output422([{"frame": fd, "bob": bd, "field": field, "mode": mode}], w, h)
    sim.sync()
    for i in range(3):
        assert np.array_equal(bd[i].cpu().numpy(), bob.pix(i)), i
    sim.close()


@pytest.mark.gpu
def test_hip_output_frame_batch_unaligned():
    """A batch of descriptors, odd pointers/linesizes (byte path) and error codes."""
    import torch
    w, h = 66, 20
    sim = ntscsim.FieldSimulator(params=L.make_params_tocomp([]))
    jobs, want = [], []
    for k in range(5):
        frame = L.yuv_noise(w, h, 20 + k, pad=3)
        bob = L.Yuv422(w, h, pad=1, fill=k)
        fd, bd = to_dev(torch, frame), to_dev(torch, bob)
        L.tocomp_oracle_output_frame(bob, frame, k & 1, k % 3)
        jobs.append({"frame": fd, "bob": bd, "field": k & 1, "mode": k % 3})
        want.append(bob)
    sim.output422(jobs, w, h)
    sim.sync()
    for j, b in zip(jobs, want):
        for i in range(3):
            assert np.array_equal(j["bob"][i].cpu().numpy(), b.plane(i)), i
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.output422([dict(jobs[0], mode=4)], w, h)
    assert e.value.code == _capi.E_ARG
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.output422(jobs[:1], w + 62, h)
    assert e.value.code == _capi.E_SIZE
    sim.close()
