"""Pins the oracle against the reference's own hot-path functions (oracle/_ref, built by
oracle/build_ref.sh from /root/reference).  Skipped where the reference is absent (GPU box)."""
import numpy as np
import pytest

import _libs as L
import cases

pytestmark = pytest.mark.skipif(not L.have_ref(), reason="oracle/_ref not built (no /root/reference)")


def _run_both(flags, w, h, n, kind, il=0, tff=0, seed_shift=0):
    p = L.make_params(flags)
    srcs = [cases.make_source(kind, w, h, j + seed_shift) for j in range((n + 1) // 2)]
    o, r = L.OracleStream(p), L.RefStream(p)
    do = np.zeros((h, w, 4), np.uint8)
    dr = np.zeros((h, w, 4), np.uint8)
    for (si, field, fieldno) in cases.case_jobs(n):
        r.field(dr, srcs[si], field, fieldno, il, tff)
        o.field(do, srcs[si], field, fieldno, il, tff)
        assert np.array_equal(do, dr), "diverged at field %d" % fieldno
    return o


@pytest.mark.parametrize("c", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_case_matrix_other_seeds(c):
    name, flags, w, h, n, kind, il, tff = c
    _run_both(flags, w, h, n + 2, kind, il, tff, seed_shift=7)


@pytest.mark.parametrize("flags,w,h,n", [
    ([], 720, 480, 4), (["-vhs"], 720, 480, 4), (["-vhs"], 720, 486, 4),
    (["-vhs", "-vhs-speed", "ep", "-comp-catv2"], 720, 480, 2), (["-vhs"], 1920, 1080, 2),
])
def test_full_size(flags, w, h, n):
    _run_both(flags, w, h, n, "bars")


def test_rand_stream_is_libc():
    r = L.ref()
    r.ntsc_ref_srand(1)
    g = L.OracleRng()
    L.oracle().ntsc_oracle_rng_seed(g, 1)
    import ctypes as C
    for _ in range(5000):
        assert r.ntsc_ref_rand() == L.oracle().ntsc_oracle_rng_next(C.byref(g))


@pytest.mark.parametrize("h", [2, 3, 8, 9, 32, 33])
@pytest.mark.parametrize("current", [0, 1, 2, 5])
def test_bob_block_of_the_field_loop(h, current):
    """ntsc_oracle_bob == the reference's own "field deinterlace" block (ffmpeg_ntsc.cpp:2233-2257,
    extracted verbatim by oracle/build_ref.sh) for both field parities, even and odd heights (the
    field-0 branch's y + 1 < height bound, :2248) and a padded linesize."""
    import ctypes as C
    w, pad = 24, 3
    rng = np.random.RandomState(100 * h + current)
    a = rng.randint(0, 256, size=(h, w + pad, 4), dtype=np.uint8)
    b = a.copy()
    L.ref().ntsc_ref_bob(L._ptr(a), (w + pad) * 4, w, h, current)
    L.oracle().ntsc_oracle_bob(L._ptr(b), (w + pad) * 4, w, h, (current & 1) ^ 1)
    assert np.array_equal(a, b)
