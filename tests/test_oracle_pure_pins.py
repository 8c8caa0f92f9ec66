"""The stand-in-free pin.

oracle/build_ref_pure.sh compiles the reference's own text for the pieces of the three hot paths that need
nothing but libc / STL headers -- class LowpassFilter (ffmpeg_ntsc.cpp:74-106 and its two copies), RGB_to_YIQ
:1375-1383, YIQ_to_RGB :1385-1396, clampu8 and black_key of ffmpeg_to_composite.cpp (:335-342, :954-972) and the
raw-capture front end hsync_dc_proc of ffmpeg_raw28ntsc.cpp (:556-594) -- with NO declaration of ours standing
in for a missing header.  Two layers:
  * where libref_pure.so exists (it is built where /root/reference is, and travels as a binary): the oracle's
    primitives == the reference's, exhaustively (all 2^24 RGB triples) and on seeded sequences, bit for bit;
  * everywhere: the oracle's primitives == tests/golden/pure_pins.npz, the reference's outputs on the same
    seeded inputs (tests/golden/make_pure_pins.py).
These are the only rows of the oracle whose parity is pinned without a stand-in; the frame-level paths stay
"parity unpinned" (DESIGN.md section 4).
"""
import numpy as np
import pytest

import _pure as P

needs_pure = pytest.mark.skipif(not P.have_pure(), reason="oracle/_ref/libref_pure.so not built (no /root/reference)")


@pytest.fixture(scope="module")
def pins():
    return np.load(P.PINS)


# ------------------------------------------------------------------ against the committed vectors
def test_rgb_to_yiq_cube_hash_and_sample_match_the_reference(pins):
    o = P.oracle()
    assert P.cube_hash(o, "oracle") == int(pins["cube_fnv"][0])
    idx = P.cube_sample_index()
    assert np.array_equal(P.rgb_to_yiq(o, "oracle", P.cube_triples(idx)), pins["cube_sample"])


@pytest.mark.parametrize("k", range(len(P.FILTER_CASES)))
def test_one_pole_sequences_match_the_reference(pins, k):
    tool, rate, hz, reset, hp, seed, n = P.FILTER_CASES[k]
    y, alpha = P.run_filter(P.oracle(), "oracle", tool, rate, hz, reset, hp, P.filter_input(seed, n))
    assert np.array([alpha]).view(np.uint64)[0] == pins["alpha%02d" % k][0]
    assert np.array_equal(y.view(np.uint64), pins["filter%02d" % k])      # bit patterns, not values


def test_yiq_to_rgb_matches_the_reference(pins):
    rgb = P.yiq_to_rgb(P.oracle(), "oracle", P.yiq_input())
    assert P.fnv(rgb) == int(pins["yiq_rgb_fnv"][0])
    assert np.array_equal(rgb[:4096].astype(np.uint8), pins["yiq_rgb_sample"])
    assert rgb.min() == 0 and rgb.max() == 255                            # both clamps were exercised


def test_clampu8_and_black_key_match_the_reference(pins):
    o = P.oracle()
    assert P.fnv(P.clampu8(o, "oracle", P.clamp_input())) == int(pins["clamp_fnv"][0])
    for level in P.BKEY_LEVELS:
        for wch in (0, 1):
            d, f = P.bkey_input(level)
            d0 = d.copy()
            P.black_key(o, "oracle", level, wch, d, f)
            assert [P.fnv(d), P.fnv(f)] == [int(v) for v in pins["bkey_%d_%d" % (level, wch)]]
            assert (d != d0).any()                                        # the key fired somewhere


@pytest.mark.parametrize("k", range(len(P.FRONT_CASES)))
def test_raw28_front_end_matches_the_reference(pins, k):
    rate, mark, fields, seed, noise, cut = P.FRONT_CASES[k]
    cap = P.front_capture(fields, seed, noise, cut)
    h, r = P.raw28_front(P.oracle(), "oracle", rate, mark, cap)
    assert [P.fnv(h), P.fnv(r)] == [int(v) for v in pins["front%d" % k]]
    assert np.array_equal(np.stack([h[:8192], r[:8192]]), pins["front%d_head" % k])


# ------------------------------------------------------------------ against the extract itself
@needs_pure
def test_rgb_to_yiq_exhaustive_against_the_extract():
    a = P.cube_full(P.pure_ref(), "ref")
    b = P.cube_full(P.oracle(), "oracle")
    assert np.array_equal(a, b)                                           # all 16,777,216 triples


@needs_pure
def test_random_filter_sequences_against_the_extract():
    r = np.random.RandomState(2025)
    for trial in range(200):
        tool = int(r.randint(0, 3))
        rate = float(r.choice([P.NTSC_RATE, P.NTSC_RATE / 2, P.R28, 40e6, 48000.0]))
        hz = float(np.exp(r.uniform(np.log(50.0), np.log(rate * 0.9))))
        reset = float(r.choice([0.0, 16.0, 128.0, -300.5]))
        hp = int(r.randint(0, 2))
        x = P.filter_input(int(r.randint(0, 1 << 20)), 1500)
        if trial % 3 == 0:
            x = x + r.uniform(-1, 1, size=x.size)                         # non-integers too
        ya, aa = P.run_filter(P.pure_ref(), "ref", tool, rate, hz, reset, hp, x)
        yb, ab = P.run_filter(P.oracle(), "oracle", tool, rate, hz, reset, hp, x)
        assert aa == ab and np.array_equal(ya.view(np.uint64), yb.view(np.uint64)), (trial, tool, rate, hz)


@needs_pure
def test_yiq_to_rgb_clamp_and_key_against_the_extract():
    yiq = P.yiq_input()
    assert np.array_equal(P.yiq_to_rgb(P.pure_ref(), "ref", yiq), P.yiq_to_rgb(P.oracle(), "oracle", yiq))
    x = P.clamp_input()
    assert np.array_equal(P.clampu8(P.pure_ref(), "ref", x), P.clampu8(P.oracle(), "oracle", x))
    for level in range(0, 24, 3):
        for wch in (0, 1):
            d1, f1 = P.bkey_input(level)
            d2, f2 = d1.copy(), f1.copy()
            P.black_key(P.pure_ref(), "ref", level, wch, d1, f1)
            P.black_key(P.oracle(), "oracle", level, wch, d2, f2)
            assert np.array_equal(d1, d2) and np.array_equal(f1, f2)


@needs_pure
@pytest.mark.parametrize("rate,mark,noise", [(0.0, 0, 3), (0.0, 1, 24), (40000000.0, 0, 12)])
def test_raw28_front_end_against_the_extract(rate, mark, noise):
    cap = P.front_capture(2, 11, noise, 777)
    ha, ra = P.raw28_front(P.pure_ref(), "ref", rate, mark, cap)
    hb, rb = P.raw28_front(P.oracle(), "oracle", rate, mark, cap)
    assert np.array_equal(ha, hb) and np.array_equal(ra, rb)
