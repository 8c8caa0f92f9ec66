"""The CPU oracle against the committed golden vectors (generated from the reference's own
hot-path code by tests/golden/make_golden.py) and the full-size hash table."""
import json
import os

import numpy as np
import pytest

import _libs as L
import cases

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "ntsc_golden.npz"))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "ntsc_golden.json")))["cases"]
FULL = json.load(open(os.path.join(HERE, "golden", "ntsc_fullsize_hashes.json")))["cases"]


def test_manifest_matches_case_table():
    assert [m["name"] for m in MANIFEST] == [c[0] for c in cases.CASES]
    for m, c in zip(MANIFEST, cases.CASES):
        assert (m["flags"], m["w"], m["h"], m["n"], m["src"]) == (c[1], c[2], c[3], c[4], c[5])


@pytest.mark.parametrize("m", MANIFEST, ids=[m["name"] for m in MANIFEST])
def test_oracle_reproduces_golden(m):
    name, w, h, n = m["name"], m["w"], m["h"], m["n"]
    p = L.make_params(m["flags"])
    srcs = GOLD["%s__src" % name]
    # the stored inputs are what the generator says they are
    for j in range(srcs.shape[0]):
        assert np.array_equal(srcs[j], cases.make_source(m["src"], w, h, j))
    o = L.OracleStream(p)
    dst = np.zeros((h, w, 4), np.uint8)
    for k, (si, field, fieldno) in enumerate(cases.case_jobs(n)):
        o.field(dst, np.ascontiguousarray(srcs[si]), field, fieldno, m["interlaced"], m["tff"])
        assert np.array_equal(dst[field::2], GOLD["%s__field%d" % (name, k)]), "field %d" % k
    assert np.array_equal(dst, GOLD["%s__final" % name])
    assert "%016x" % L.fnv1a(dst) == m["fnv1a_final"]
    # alpha byte is 0 on every written pixel (ffmpeg_ntsc.cpp:1914)
    assert not dst[..., 3].any()
    # the oracle consumed exactly the number of draws the product's closed form predicts
    lib = L.product()
    import ctypes as C
    exp = sum(lib.ntscsim_rng_calls_per_field(C.byref(p), w, h, f) for (_, f, _) in cases.case_jobs(n))
    assert o.rng_pos == exp


@pytest.mark.parametrize("c", [c for c in FULL if c["w"] <= 1920],
                         ids=lambda c: "%dx%d%s" % (c["w"], c["h"], "".join(c["flags"])))
def test_oracle_fullsize_hashes(c):
    w, h = c["w"], c["h"]
    p = L.make_params(c["flags"])
    o = L.OracleStream(p)
    dst = np.zeros((h, w, 4), np.uint8)
    for k in range(c["n"]):
        o.field(dst, L.bars(w, h, k // 2), (k & 1) ^ 1, k)
        assert "%016x" % L.fnv1a(dst) == c["fnv1a_after_each_field"][k], "field %d" % k


def test_rows_of_other_field_untouched():
    p = L.make_params(["-vhs"])
    src = L.noise_frame(96, 32)
    dst = np.full((32, 96, 4), 0xAB, np.uint8)
    L.OracleStream(p).field(dst, src, 1, 0)
    assert (dst[0::2] == 0xAB).all() and not (dst[1::2] == 0xAB).all()


def test_bob_matches_loop_description():
    # field 1: odd rows copied upward; field 0: row y+1 copied onto odd y while y+1 < H
    h, w = 8, 16
    f = np.arange(h, dtype=np.uint8)[:, None, None].repeat(w, 1).repeat(4, 2).copy()
    a = f.copy()
    L.oracle().ntsc_oracle_bob(L._ptr(a), w * 4, w, h, 1)
    assert [int(a[y, 0, 0]) for y in range(h)] == [1, 1, 3, 3, 5, 5, 7, 7]
    a = f.copy()
    L.oracle().ntsc_oracle_bob(L._ptr(a), w * 4, w, h, 0)
    assert [int(a[y, 0, 0]) for y in range(h)] == [0, 2, 2, 4, 4, 6, 6, 7]


_BOB = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ntsc_bob_golden.npz"))
BOB_CASES = [("bob_default_even", [], 96, 32, 4), ("bob_default_odd", [], 96, 33, 4),
             ("bob_vhs_even", ["-vhs"], 96, 32, 4), ("bob_vhs_odd", ["-vhs"], 100, 35, 4),
             ("bob_vhs_h2", ["-vhs"], 64, 2, 2), ("bob_vhs_h3", ["-vhs"], 64, 3, 2)]


@pytest.mark.parametrize("c", BOB_CASES, ids=[c[0] for c in BOB_CASES])
def test_field_loop_with_bob_matches_reference_golden(c):
    """composite_layer() + the loop's bob block (ffmpeg_ntsc.cpp:2229-2257), field after field into one
    frame: oracle == the frames the reference text produced (tests/golden/make_golden_bob.py)."""
    name, flags, w, h, n = c
    p = L.make_params(flags)
    srcs = [np.ascontiguousarray(a) for a in _BOB["%s__src" % name]]
    o = L.OracleStream(p)
    dst = np.full((h, w, 4), 0x5A, np.uint8)
    for k in range(n):
        field = (k & 1) ^ 1
        o.field(dst, srcs[k // 2], field, k)
        L.oracle().ntsc_oracle_bob(L._ptr(dst), w * 4, w, h, field)
        assert np.array_equal(dst, _BOB["%s__after%d" % (name, k)]), "field %d" % k
