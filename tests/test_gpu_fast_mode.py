"""NTSCSIM_MODE_FAST32: fp32 filters / colour matrices / rotation, everything else identical.

BASELINE north_star: "bit-exact output vs the reference for integer pixel paths and within a stated
fp32 tolerance for the filtered signal".  Stated tolerance (8-bit BGRA output vs the oracle):
  * every channel of every pixel within +-1 LSB                         (max |diff| <= 1)
  * at most 1 % of the pixels differ at all                             (observed <= 0.3 %)
  * the rand() stream, the integer stages and the untouched rows are identical (same noise
    pattern, same dropouts, same head-switch geometry)
The default mode stays EXACT; nothing else in the suite runs in FAST32.
"""
import numpy as np
import pytest

import _libs as L
import ntscsim
from ntscsim import _capi

pytestmark = pytest.mark.gpu

MAX_ABS = 1
MAX_FRACTION_DIFFERENT = 0.01


@pytest.mark.parametrize("flags,w,h,n,kind", [
    ([], 720, 486, 4, "bars"), ([], 720, 486, 2, "noise"),
    (["-vhs"], 720, 486, 4, "bars"), (["-vhs"], 720, 486, 4, "noise"),
    (["-vhs", "-vhs-speed", "ep"], 720, 480, 2, "noise"),
    (["-vhs", "-comp-catv3"], 720, 480, 2, "noise"),
    (["-vhs", "-vhs-svideo", "1", "-out-composite-lowpass-lite", "0"], 360, 240, 2, "noise"),
    (["-vhs"], 1920, 1080, 2, "noise"),
])
def test_fast32_within_stated_tolerance(flags, w, h, n, kind):
    import torch
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 5 + j) if kind == "noise" else L.bars(w, h, j) for j in range((n + 1) // 2)]
    o = L.OracleStream(p)
    exp = np.full((n, h, w, 4), 0x77, np.uint8)
    for k in range(n):
        o.field(exp[k], srcs[k // 2], (k & 1) ^ 1, k)
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_mode(_capi.MODE_FAST32)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.full((n, h, w, 4), 0x77, dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)])
    sim.sync()
    assert sim.rng_pos == o.rng_pos
    got = dst.cpu().numpy()
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= MAX_ABS
    for k in range(n):
        field = (k & 1) ^ 1
        assert (got[k][1 - field::2] == 0x77).all()           # other field's rows untouched
        assert not got[k][field::2, :, 3].any()                # alpha 0
        frac = (d[k][field::2].max(axis=-1) > 0).mean()
        assert frac <= MAX_FRACTION_DIFFERENT, (k, frac)
    sim.close()


# ---- NTSCSIM_MODE_FLOAT: the all-float pipeline (csrc/ntsc_float.hip) ------------------------------------------------
# Stated tolerance: every channel of every pixel within +-1 LSB, on every input.  How many pixels differ at all depends on
# the input: every dropped `(int)` is worth a uniform +-1/2 of 1/256 of an output step on natural / noisy pictures (mean
# compensated: measured mean signed difference < 0.002 LSB per channel) -- 0.5 % of the channels, 1.5 % of the pixels --
# but FLAT colour (bars) is the reference's worst case for truncation: its filters converge towards integers from one
# side, so `(int)` takes a whole unit off for hundreds of samples where the float pipeline keeps the fraction.
# Measured (tools/float_err.py, profiles/r06_float_err.txt): noise 1.2-1.5 % of the pixels, ramps 2.6 %, -vhs bars
# 2.7-4 %, default-preset bars 7 %.  rand() stream, noise accumulators, head-switch geometry, dropout rows: identical.
FLOAT_MAX_FRACTION = {"noise": 0.02, "ramp": 0.035, "bars": 0.08}


def _ramp(w, h):
    x = np.arange(w)[None, :]
    y = np.arange(h)[:, None]
    fr = np.zeros((h, w, 4), np.uint8)
    fr[..., 0] = x * 255 // max(1, w - 1)
    fr[..., 1] = y * 255 // max(1, h - 1)
    fr[..., 2] = (x + y) * 255 // (w + h - 2)
    return fr


@pytest.mark.parametrize("flags,w,h,n,kind,fp", [
    ([], 720, 486, 4, "bars", True), ([], 720, 486, 2, "noise", True),
    (["-vhs"], 720, 486, 4, "bars", True), (["-vhs"], 720, 486, 4, "noise", True), (["-vhs"], 720, 486, 2, "ramp", True),
    (["-vhs", "-vhs-speed", "ep"], 720, 480, 2, "noise", True), (["-vhs", "-vhs-speed", "lp"], 704, 480, 2, "noise", True),
    (["-vhs"], 1920, 1080, 2, "noise", True), (["-vhs"], 3840, 2160, 2, "noise", True),
    # geometries whose rows are all row start / row end (the guarded steps), the head switch inside the frame
    (["-vhs"], 36, 17, 4, "noise", True), (["-vhs"], 100, 31, 4, "noise", True), ([], 20, 9, 4, "noise", True),
    (["-vhs"], 33, 17, 4, "noise", False),            # rows that are not 16-byte aligned: the FAST32 forms
    (["-vhs", "-vhs-head-switching-point", "0.8"], 360, 244, 4, "noise", True),
    # switch sets outside the float forms run the FAST32 kernels in this mode
    (["-vhs", "-vhs-svideo", "1"], 360, 240, 2, "noise", False), (["-vhs", "-comp-phase", "90"], 360, 240, 2, "noise", False),
])
def test_float_pipeline_within_stated_tolerance(flags, w, h, n, kind, fp):
    import torch
    p = L.make_params(flags)
    srcs = [L.noise_frame(w, h, 5 + j) if kind == "noise" else (L.bars(w, h, j) if kind == "bars" else _ramp(w, h))
            for j in range((n + 1) // 2)]
    o = L.OracleStream(p)
    exp = np.full((n, h, w, 4), 0x77, np.uint8)
    for k in range(n):
        o.field(exp[k], srcs[k // 2], (k & 1) ^ 1, k)
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_mode(_capi.MODE_FLOAT)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.full((n, h, w, 4), 0x77, dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(n)])
    sim.sync()
    assert sim.rng_pos == o.rng_pos
    kern = sim.last_kernels()
    if fp:
        assert "k_encode_fp" in kern and ("k_decode_fp2" if "-vhs" in flags else "k_decode_fp<false>") in kern, kern      # (short batch: the two-role form)
    else:
        assert not any(k_.endswith("_fp") or "_fp<" in k_ for k_ in kern), kern
    got = dst.cpu().numpy()
    d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= MAX_ABS
    limit = FLOAT_MAX_FRACTION[kind] if fp else MAX_FRACTION_DIFFERENT
    if w < 64:
        limit = 0.06            # (a handful of pixels per row: the share is noisy)
    for k in range(n):
        field = (k & 1) ^ 1
        assert (got[k][1 - field::2] == 0x77).all()           # other field's rows untouched
        assert not got[k][field::2, :, 3].any()                # alpha 0
        frac = (d[k][field::2].max(axis=-1) > 0).mean()
        assert frac <= limit, (k, frac)
        # the compensated luma bias: no channel is off on average
        sd = (got[k][field::2, :, :3].astype(np.int16) - exp[k][field::2, :, :3].astype(np.int16))
        if kind != "bars" and w >= 360:
            assert abs(sd.mean()) < 0.004, sd.mean()
    sim.close()


def test_float_long_batches_take_the_one_wave_form_and_agree_with_the_two_role_form():
    """more than 128 fields per launch: k_decode_fp<true> (one wave per 63 rows); the same pictures as the two-role
    workgroup form the short batches take (the oracle comparison above covers the two-role form, this one the other)"""
    import torch
    w, h, n = 360, 120, 160
    p = L.make_params(["-vhs"])
    srcs = [L.noise_frame(w, h, 70 + j) for j in range(4)]
    src = torch.from_numpy(np.stack(srcs)).cuda()
    jobs = [(k % 4, k, (k & 1) ^ 1, k) for k in range(n)]
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_mode(_capi.MODE_FLOAT)
    a = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, a, jobs)
    sim.sync()
    assert "k_decode_fp<true>" in sim.last_kernels()
    sim.rng_pos = 0
    b = torch.zeros((n, h, w, 4), dtype=torch.uint8, device="cuda")
    for i in range(0, n, 40):
        sim.fields(src, b[i:i + 40], [(k % 4, k - i, (k & 1) ^ 1, k) for k in range(i, i + 40)])
    sim.sync()
    assert "k_decode_fp2" in sim.last_kernels()
    # (not bit for bit: the two forms change from their guarded to their steady steps at different positions of a row, and
    #  the steady separator sums its box as pair sums -- another association of the same four floats)
    d = (a.to(torch.int16) - b.to(torch.int16)).abs()
    assert int(d.max()) <= 1 and float((d.amax(dim=-1) > 0).float().mean()) < 0.002
    sim.close()


def test_float_mode_through_the_host_call_and_the_submit_engine():
    """the mode is the ctx's: ntscsim_field() and ntscsim_submit() lanes run in the tolerance mode too -- their short launches
    as the role kernels with float filter states (k_field_pipe<float>: 4.8k calls per second against 2.3k for the float
    pipeline's own short-batch form) -- inside the mode's stated tolerance against the oracle, same rand() stream"""
    w, h = 720, 486
    p = L.make_params(["-vhs"])
    srcs = [L.noise_frame(w, h, 40 + j) for j in range(2)]
    o = L.OracleStream(p)
    exp = np.zeros((4, h, w, 4), np.uint8)
    for k in range(4):
        o.field(exp[k], srcs[k // 2], (k & 1) ^ 1, k)
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_mode(_capi.MODE_FLOAT)

    def check(got, what):
        d = np.abs(got.astype(np.int16) - exp.astype(np.int16))
        assert d.max() <= MAX_ABS, what
        for k in range(4):
            field = (k & 1) ^ 1
            assert not got[k][1 - field::2].any(), what              # the other field's rows untouched
            assert (d[k][field::2].max(axis=-1) > 0).mean() <= MAX_FRACTION_DIFFERENT, what

    one = np.zeros((4, h, w, 4), np.uint8)
    for k in range(4):
        sim.field_host(one[k], srcs[k // 2], (k & 1) ^ 1, k)
        assert "k_field_pipe<float>" in sim.last_kernels()
    assert sim.rng_pos == o.rng_pos
    check(one, "ntscsim_field")
    sim.rng_pos = 0
    two = np.zeros((4, h, w, 4), np.uint8)
    ts = [sim.submit(two[k], srcs[k // 2], (k & 1) ^ 1, k) for k in range(4)]
    sim.wait(ts[-1])
    check(two, "ntscsim_submit")
    assert np.array_equal(one, two)
    sim.close()


@pytest.mark.parametrize("w,h", [(720, 486), (100, 37)])
def test_fast32_through_the_host_call_takes_the_pipelined_form_and_equals_the_batch(w, h):
    """FAST32 is the exact kernels with float filter states: ntscsim_field() runs it as k_field_pipe<float> (five roles of
    one workgroup), same bytes as the device-resident batch (k_encode_fast<float> | k_decode_fast<true,float>)"""
    import torch
    p = L.make_params(["-vhs"], output_height=h)
    srcs = [L.noise_frame(w, h, 50 + j) for j in range(2)]
    sim = ntscsim.FieldSimulator(params=p)
    sim.set_mode(_capi.MODE_FAST32)
    src = torch.from_numpy(np.stack(srcs)).cuda()
    dst = torch.zeros((4, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k, (k & 1) ^ 1, k) for k in range(4)])
    sim.sync()
    assert "k_field_pipe<float>" not in sim.last_kernels()
    ref = dst.cpu().numpy()
    sim.rng_pos = 0
    one = np.zeros((4, h, w, 4), np.uint8)
    for k in range(4):
        sim.field_host(one[k], srcs[k // 2], (k & 1) ^ 1, k)
        assert "k_field_pipe<float>" in sim.last_kernels()
    assert np.array_equal(one, ref)
    sim.close()


def test_mode_switch_roundtrip():
    """EXACT after FAST32 on the same ctx is bit-exact again."""
    import torch
    w, h = 96, 32
    p = L.make_params(["-vhs"])
    s = L.noise_frame(w, h, 3)
    e = np.zeros((h, w, 4), np.uint8)
    L.OracleStream(p).field(e, s, 1, 0)
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(s[None]).cuda()
    dst = torch.zeros((1, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.set_mode(_capi.MODE_FAST32)
    sim.fields(src, dst, [(0, 0, 1, 0)], rng_pos=[0])
    sim.set_mode(_capi.MODE_FLOAT)
    sim.fields(src, dst, [(0, 0, 1, 0)], rng_pos=[0])
    sim.set_mode(_capi.MODE_EXACT)
    dst.zero_()
    sim.fields(src, dst, [(0, 0, 1, 0)], rng_pos=[0])
    sim.sync()
    assert np.array_equal(dst[0].cpu().numpy(), e)
    with pytest.raises(ntscsim.NtscsimError):
        sim.set_mode(7)
    sim.close()
