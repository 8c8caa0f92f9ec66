"""Test infrastructure for the stand-in-free pin: the same unit entry points of
  * oracle/_ref/libref_pure.so  ("ref": the reference's own text behind libc headers alone,
                                 oracle/build_ref_pure.sh; exists where /root/reference did at build time)
  * oracle/libntsc_oracle.so    ("oracle": our restatement)
and the seeded inputs both sides of tests/golden/pure_pins.npz are generated from."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PURE_SO = os.path.join(ROOT, "oracle", "_ref", "libref_pure.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "libntsc_oracle.so")
PINS = os.path.join(ROOT, "tests", "golden", "pure_pins.npz")

NTSC_RATE = (315000000.00 * 4) / 88
R28 = (315000000.00 * 8.0) / 88.00
_SCAN28 = (R28 / (30000.00 / 1001.00)) / 525.00

# (tool, rate, cutoff, reset, highpass, seed, n): every (rate, cutoff) pair the three paths configure --
# ffmpeg_ntsc.cpp:1415, :1446, :1621, :1801-1805, :1822-1825, :1874; ffmpeg_to_composite.cpp:377-381, :419,
# :642, :817-821, :838-841, :890, :908; ffmpeg_raw28ntsc.cpp:889-892 -- and the value ranges they see
FILTER_CASES = [
    (0, NTSC_RATE, 2600000.0, 0.0, 0, 1, 4096),
    (0, NTSC_RATE, 1300000.0, 0.0, 0, 2, 4096),
    (0, NTSC_RATE, 600000.0, 0.0, 0, 3, 4096),
    (0, NTSC_RATE, 3000000.0, 16.0, 0, 4, 4096),
    (0, NTSC_RATE, 2400000.0, 16.0, 0, 5, 4096),
    (0, NTSC_RATE, 1900000.0, 16.0, 0, 6, 4096),
    (0, NTSC_RATE, 1400000.0, 16.0, 0, 7, 4096),
    (0, NTSC_RATE, 320000.0, 0.0, 0, 8, 4096),
    (0, NTSC_RATE, 290000.0, 0.0, 0, 9, 4096),
    (0, NTSC_RATE, 1000000.0, 16.0, 1, 10, 4096),
    (0, NTSC_RATE, 6000000.0, 0.0, 1, 11, 4096),
    (0, NTSC_RATE, 2400000.0 * 4, 0.0, 0, 12, 4096),
    (1, NTSC_RATE / 2, 1300000.0, 128.0, 0, 13, 4096),
    (1, NTSC_RATE / 2, 2600000.0, 128.0, 0, 14, 4096),
    (1, NTSC_RATE, 2400000.0, 16.0, 0, 15, 4096),
    (1, NTSC_RATE, 1000000.0, 16.0, 1, 16, 4096),
    (1, NTSC_RATE / 2, 320000.0, 128.0, 0, 17, 4096),
    (2, R28, R28 / (_SCAN28 * 0.075 * 0.75), 0.0, 0, 18, 8192),
    (2, 40000000.0, 40000000.0 / 190.0, 0.0, 0, 19, 8192),
]
BKEY_LEVELS = (0, 4, 20)
# (sample_rate or 0, mark_sync, fields, seed, noise, cut)
FRONT_CASES = [(0.0, 0, 2, 1, 3, 0), (0.0, 1, 2, 7, 12, 4321), (0.0, 0, 1, 3, 36, 100)]


def _bind(lib, prefix_filter, has_tool):
    dp = C.POINTER(C.c_double)
    f = getattr(lib, prefix_filter)
    f.argtypes = ([C.c_int] if has_tool else []) + [C.c_double, C.c_double, C.c_double, C.c_int, dp, C.c_size_t, dp, dp]
    f.restype = C.c_int if has_tool else None


_pure = None
_oracle = None


def have_pure():
    return os.path.exists(PURE_SO)


def pure_ref():
    global _pure
    if _pure is None:
        lib = C.CDLL(PURE_SO)
        _bind(lib, "ref_pure_filter", True)
        lib.ref_pure_rgb_to_yiq_cube.restype = C.c_ulonglong
        lib.ref_pure_rgb_to_yiq_cube.argtypes = [C.c_void_p]
        for n in ("ref_pure_rgb_to_yiq", "ref_pure_yiq_to_rgb", "ref_pure_clampu8"):
            getattr(lib, n).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
            getattr(lib, n).restype = None
        lib.ref_pure_black_key.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.ref_pure_black_key.restype = None
        lib.ref_pure_raw28_front.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.ref_pure_raw28_front.restype = None
        _pure = lib
    return _pure


class _RawOpts(C.Structure):
    _fields_ = [("sample_rate", C.c_double)] + [(n, C.c_int32) for n in (
        "mark_sync", "disable_sync", "disable_wp_equ", "show_subcarrier", "disable_subcarrier",
        "disable_equalization")]


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ORACLE_SO)
        _bind(lib, "ntsc_oracle_unit_filter", False)
        _bind(lib, "tocomp_oracle_unit_filter", False)
        lib.ntsc_oracle_unit_rgb_to_yiq_cube.restype = C.c_uint64
        lib.ntsc_oracle_unit_rgb_to_yiq_cube.argtypes = [C.c_void_p]
        for n in ("ntsc_oracle_unit_rgb_to_yiq", "ntsc_oracle_unit_yiq_to_rgb", "tocomp_oracle_unit_clampu8"):
            getattr(lib, n).argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
            getattr(lib, n).restype = None
        lib.tocomp_oracle_unit_black_key.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.tocomp_oracle_unit_black_key.restype = None
        lib.raw28_oracle_front.argtypes = [C.POINTER(_RawOpts), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        lib.raw28_oracle_front.restype = None
        lib.raw28_synth_capture.restype = C.c_size_t
        lib.raw28_synth_capture.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int]
        lib.ntsc_oracle_fnv1a.argtypes = [C.c_void_p, C.c_size_t]
        lib.ntsc_oracle_fnv1a.restype = C.c_uint64
        _oracle = lib
    return _oracle


def fnv(a):
    a = np.ascontiguousarray(a)
    return int(oracle().ntsc_oracle_fnv1a(a.ctypes.data, a.nbytes))


# ------------------------------------------------------------------ seeded inputs ------------
def filter_input(seed, n):
    """Integer-valued samples like the path's (every filter input is an int plane value), in bursts of
    different amplitude with a few extreme steps."""
    r = np.random.RandomState(1000 + seed)
    x = np.empty(n, np.float64)
    amp = [256.0, 65535.0, 450000.0, 16.0]
    for b in range(0, n, 512):
        a = amp[(b // 512) % 4]
        x[b:b + 512] = np.round(r.uniform(-a, a, size=min(512, n - b)))
    x[::97] = 0.0
    return x


def cube_sample_index():
    return (np.arange(4096, dtype=np.int64) * 4099 + 17) % (1 << 24)


def cube_triples(idx):
    idx = np.asarray(idx, np.int64)
    return np.stack([idx >> 16, (idx >> 8) & 255, idx & 255], axis=1).astype(np.int32)


def yiq_input(n=1 << 20):
    r = np.random.RandomState(77)
    y = r.randint(-20000, 90000, size=n)
    i = r.randint(-60000, 60000, size=n)
    q = r.randint(-60000, 60000, size=n)
    a = np.stack([y, i, q], axis=1).astype(np.int32)
    a[:4096, 0] = np.linspace(-512, 66000, 4096).astype(np.int32)     # a sweep through both clamps
    a[:4096, 1:] //= 64
    return a


def clamp_input():
    return np.concatenate([np.arange(-70000, 70000, dtype=np.int32),
                           np.array([-2 ** 31, 2 ** 31 - 1, -1, 0, 255, 256], np.int32)])


def bkey_input(level):
    r = np.random.RandomState(500 + level)
    n = 1 << 18
    d = r.randint(0, 256, size=(n, 3)).astype(np.uint8)
    d[: n // 2, 0] = r.randint(10, 50, size=n // 2)             # around the key threshold
    d[: n // 2, 1] = r.randint(118, 139, size=n // 2)
    d[: n // 2, 2] = r.randint(118, 139, size=n // 2)
    f = r.randint(0, 256, size=(n, 3)).astype(np.uint8)
    return d, f


def front_capture(fields, seed, noise, cut):
    o = oracle()
    buf = np.zeros(fields * 477750 + 16, np.uint8)
    n = o.raw28_synth_capture(buf.ctypes.data, buf.size, fields, seed, noise)
    return np.ascontiguousarray(buf[cut:n])


# ------------------------------------------------------------------ the two sides -------------
def run_filter(lib, side, tool, rate, hz, reset, hp, x):
    y = np.empty_like(x)
    alpha = C.c_double()
    dp = C.POINTER(C.c_double)
    if side == "ref":
        rc = lib.ref_pure_filter(tool, rate, hz, reset, hp, x.ctypes.data_as(dp), x.size, y.ctypes.data_as(dp),
                                 C.byref(alpha))
        assert rc == 0
    else:
        # the raw28 oracle shares the primitive of ntsc_oracle.c's formula; tools 0 and 2 go through the
        # BGRA tool's restatement, tool 1 through the variant's
        fn = lib.tocomp_oracle_unit_filter if tool == 1 else lib.ntsc_oracle_unit_filter
        fn(rate, hz, reset, hp, x.ctypes.data_as(dp), x.size, y.ctypes.data_as(dp), C.byref(alpha))
    return y, alpha.value


def cube_hash(lib, side):
    return int(lib.ref_pure_rgb_to_yiq_cube(None) if side == "ref" else lib.ntsc_oracle_unit_rgb_to_yiq_cube(None))


def cube_full(lib, side):
    a = np.empty((1 << 24, 3), np.int32)
    (lib.ref_pure_rgb_to_yiq_cube if side == "ref" else lib.ntsc_oracle_unit_rgb_to_yiq_cube)(a.ctypes.data)
    return a


def rgb_to_yiq(lib, side, rgb):
    rgb = np.ascontiguousarray(rgb, np.int32)
    out = np.empty_like(rgb)
    (lib.ref_pure_rgb_to_yiq if side == "ref" else lib.ntsc_oracle_unit_rgb_to_yiq)(rgb.ctypes.data, rgb.shape[0], out.ctypes.data)
    return out


def yiq_to_rgb(lib, side, yiq):
    yiq = np.ascontiguousarray(yiq, np.int32)
    out = np.empty_like(yiq)
    (lib.ref_pure_yiq_to_rgb if side == "ref" else lib.ntsc_oracle_unit_yiq_to_rgb)(yiq.ctypes.data, yiq.shape[0], out.ctypes.data)
    return out


def clampu8(lib, side, x):
    out = np.empty_like(x)
    (lib.ref_pure_clampu8 if side == "ref" else lib.tocomp_oracle_unit_clampu8)(x.ctypes.data, x.size, out.ctypes.data)
    return out


def black_key(lib, side, level, wch, d, f):
    (lib.ref_pure_black_key if side == "ref" else lib.tocomp_oracle_unit_black_key)(level, wch, d.ctypes.data, f.ctypes.data, d.shape[0])


def raw28_front(lib, side, rate, mark, cap):
    h = np.empty(cap.size, np.uint8)
    r = np.empty(cap.size, np.uint8)
    if side == "ref":
        lib.ref_pure_raw28_front(rate, mark, cap.ctypes.data, cap.size, h.ctypes.data, r.ctypes.data)
    else:
        o = _RawOpts()
        o.sample_rate = rate
        o.mark_sync = mark
        lib.raw28_oracle_front(C.byref(o), cap.ctypes.data, cap.size, h.ctypes.data, r.ctypes.data)
    return h, r
