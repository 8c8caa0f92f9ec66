"""ctypes bindings shared by the tests, bench.py and __graft_entry__.py.

Three shared objects are involved:
  * PRODUCT  composite-video-simulator_amd/libntscsim.so -- the C-ABI of include/ntscsim.h
             (HIP kernels + host mirror of the reference's parse_argv).  Loads without a GPU;
             ntscsim_create() then fails with NTSCSIM_E_NODEV.
  * ORACLE   oracle/libntsc_oracle.so -- our CPU restatement (test infrastructure only).
  * REF      oracle/_ref/libntsc_ref.so -- the reference's own hot-path text compiled by
             oracle/build_ref.sh; exists only where /root/reference does (never on the GPU box).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "composite-video-simulator_amd")
PRODUCT_SO = os.path.join(PKG, "libntscsim.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "libntsc_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libntsc_ref.so")


sys.path.insert(0, PKG)
import ntscsim  # noqa: E402  (the product's host-side package)
from ntscsim._capi import (DESC_BOB, DESC_INTERLACED, DESC_TFF, RNG_AUTO, FieldDesc,  # noqa: E402,F401
                           Params, make_params)


class OracleRng(C.Structure):
    _fields_ = [("r", C.c_uint32 * 34), ("i", C.c_int), ("count", C.c_uint64)]


class OracleTaps(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_int32)) for n in (
        "composite_y", "headswitch_y", "demod_y", "demod_i", "demod_q", "noise_i", "noise_q",
        "vhs_y", "vhs_i", "vhs_q", "final_y", "final_i", "final_q")]


_u8p = C.POINTER(C.c_uint8)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


_oracle = None
_ref = None


def product():
    """The product C-ABI (raises if libntscsim.so has not been built)."""
    return ntscsim.lib()


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(ORACLE_SO)
        lib.ntsc_oracle_rng_seed.argtypes = [C.POINTER(OracleRng), C.c_uint32]
        lib.ntsc_oracle_rng_seed.restype = None
        lib.ntsc_oracle_rng_next.argtypes = [C.POINTER(OracleRng)]
        lib.ntsc_oracle_rng_next.restype = C.c_uint32
        lib.ntsc_oracle_rng_discard.argtypes = [C.POINTER(OracleRng), C.c_uint64]
        lib.ntsc_oracle_rng_discard.restype = None
        lib.ntsc_oracle_field.argtypes = [C.POINTER(Params), C.POINTER(OracleRng), _u8p, C.c_int,
                                          C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int,
                                          C.c_uint, C.c_uint64, C.POINTER(OracleTaps)]
        lib.ntsc_oracle_field.restype = C.c_int
        lib.ntsc_oracle_bob.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_uint]
        lib.ntsc_oracle_bob.restype = None
        lib.ntsc_oracle_fnv1a.argtypes = [C.c_void_p, C.c_size_t]
        lib.ntsc_oracle_fnv1a.restype = C.c_uint64
        lib.ntsc_oracle_make_bars.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.ntsc_oracle_make_bars.restype = None
        lib.ntsc_oracle_make_noise.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_uint32]
        lib.ntsc_oracle_make_noise.restype = None
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        lib.ntsc_ref_set_params.argtypes = [C.POINTER(Params)]
        lib.ntsc_ref_set_params.restype = None
        lib.ntsc_ref_srand.argtypes = [C.c_uint]
        lib.ntsc_ref_srand.restype = None
        lib.ntsc_ref_rand.argtypes = []
        lib.ntsc_ref_rand.restype = C.c_uint
        lib.ntsc_ref_composite_layer.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_uint, C.c_ulonglong]
        lib.ntsc_ref_composite_layer.restype = None
        _ref = lib
    return _ref


# ---------------------------------------------------------------------------------------------

def field_rows(h, field):
    return (h - field + 1) // 2 if h > field else 0


def bars(w, h, rot=0):
    a = np.zeros((h, w, 4), dtype=np.uint8)
    oracle().ntsc_oracle_make_bars(_ptr(a), w * 4, w, h, rot)
    return a


def noise_frame(w, h, seed=0x1234567):
    a = np.zeros((h, w, 4), dtype=np.uint8)
    oracle().ntsc_oracle_make_noise(_ptr(a), w * 4, w, h, seed)
    return a


def fnv1a(a):
    a = np.ascontiguousarray(a)
    return oracle().ntsc_oracle_fnv1a(a.ctypes.data_as(C.c_void_p), a.nbytes)


class OracleStream:
    """The oracle driven like the reference's field loop: one rand() stream across calls."""

    def __init__(self, params):
        self.p = params
        self.g = OracleRng()
        oracle().ntsc_oracle_rng_seed(C.byref(self.g), 1)

    @property
    def rng_pos(self):
        return int(self.g.count)

    def skip(self, n):
        oracle().ntsc_oracle_rng_discard(C.byref(self.g), n)

    def field(self, dst, src, field, fieldno, interlaced=0, tff=0, taps=None):
        h, w = src.shape[:2]
        assert dst.shape == src.shape and src.flags.c_contiguous and dst.flags.c_contiguous
        t = None
        keep = {}
        if taps:
            t = OracleTaps()
            n = field_rows(h, field) * w
            for name in taps:
                keep[name] = np.zeros(n, dtype=np.int32)
                setattr(t, name, keep[name].ctypes.data_as(C.POINTER(C.c_int32)))
        rc = oracle().ntsc_oracle_field(C.byref(self.p), C.byref(self.g), _ptr(src), w * 4,
                                        interlaced, tff, _ptr(dst), w * 4, w, h, field, fieldno,
                                        C.byref(t) if t is not None else None)
        assert rc == 0
        return {k: v.reshape(field_rows(h, field), w) for k, v in keep.items()}


class RefStream:
    """The reference extract driven the same way (process-wide libc rand(), re-seeded to 1)."""

    def __init__(self, params):
        ref().ntsc_ref_set_params(C.byref(params))
        ref().ntsc_ref_srand(1)

    def field(self, dst, src, field, fieldno, interlaced=0, tff=0):
        h, w = src.shape[:2]
        ref().ntsc_ref_composite_layer(_ptr(dst), w * 4, _ptr(src), w * 4, interlaced, tff,
                                       w, h, field, fieldno)
