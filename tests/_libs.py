"""ctypes bindings shared by the tests, bench.py and __graft_entry__.py.

Three shared objects are involved:
  * PRODUCT  composite-video-simulator_amd/libntscsim.so -- the C-ABI of include/ntscsim.h
             (HIP kernels + host mirror of the reference's parse_argv).  Loads without a GPU;
             ntscsim_create() then fails with NTSCSIM_E_NODEV.
  * ORACLE   oracle/libntsc_oracle.so -- our CPU restatement (test infrastructure only).
  * REF      oracle/_ref/libntsc_ref.so -- the reference's own hot-path text compiled by
             oracle/build_ref.sh.  It can only be BUILT where /root/reference exists; the built,
             git-ignored binary (GPL-2 text in compiled form) travels to the GPU box inside the gpurun
             snapshot and is loaded there by the checker side of the tests and by bench.py's
             cpu_baseline leg -- never by the product.
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
        lib.ntsc_oracle_bgra_to_yuv.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p,
                                                C.c_int, _u8p, C.c_int, C.c_int]
        lib.ntsc_oracle_bgra_to_yuv.restype = None
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
        lib.ntsc_ref_bob.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_ulonglong]
        lib.ntsc_ref_bob.restype = None
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

    @staticmethod
    def bob(frame, fieldno):
        """the field loop's "field deinterlace" block (ffmpeg_ntsc.cpp:2233-2257) with current = fieldno"""
        h, w = frame.shape[:2]
        ref().ntsc_ref_bob(_ptr(frame), w * 4, w, h, fieldno)


# ------------------------------------------------------------------ 8-bit YUV422P variant -----
TOCOMP_REF_SO = os.path.join(ROOT, "oracle", "_ref", "libtocomp_ref.so")
OOB_DEFINED, OOB_MEMORY, OOB_PLANE = 0, 1, 2


class TocompPlanes(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8) * 3), ("linesize", C.c_int * 3),
                ("width", C.c_int), ("height", C.c_int)]


class Yuv422:
    """A YUV422P frame in ONE buffer (Y | U | V + slack), so that the reference's two-byte read
    past each luma row (ffmpeg_to_composite.cpp:496) lands in memory we own and control."""

    def __init__(self, w, h, pad=0, fill=0):
        self.w, self.h = w, h
        self.ls = [w + pad, w // 2 + pad, w // 2 + pad]
        sizes = [self.ls[0] * h, self.ls[1] * h, self.ls[2] * h]
        self.buf = np.full(sum(sizes) + 64, fill, np.uint8)
        self.off = [0, sizes[0], sizes[0] + sizes[1]]

    def plane(self, i):
        n = self.ls[i] * self.h
        return self.buf[self.off[i]:self.off[i] + n].reshape(self.h, self.ls[i])

    def pix(self, i):
        return self.plane(i)[:, :(self.w if i == 0 else self.w // 2)]

    def copy(self):
        o = Yuv422(self.w, self.h)
        o.ls, o.off = list(self.ls), list(self.off)
        o.buf = self.buf.copy()
        return o

    def cplanes(self):
        t = TocompPlanes()
        for i in range(3):
            t.data[i] = C.cast(self.buf.ctypes.data + self.off[i], C.POINTER(C.c_uint8))
            t.linesize[i] = self.ls[i]
        t.width, t.height = self.w, self.h
        return t

    def ptr_arrays(self):
        d = (C.POINTER(C.c_uint8) * 3)(*[C.cast(self.buf.ctypes.data + self.off[i], C.POINTER(C.c_uint8))
                                         for i in range(3)])
        ls = (C.c_int * 3)(*self.ls)
        return d, ls


def yuv_noise(w, h, seed, pad=0):
    f = Yuv422(w, h, pad)
    rng = np.random.RandomState(seed)
    f.pix(0)[:] = rng.randint(16, 236, size=(h, w), dtype=np.uint8)
    f.pix(1)[:] = rng.randint(16, 241, size=(h, w // 2), dtype=np.uint8)
    f.pix(2)[:] = rng.randint(16, 241, size=(h, w // 2), dtype=np.uint8)
    return f


def yuv_bars(w, h, rot=0, pad=0):
    """BT.601 75% colour bars in YUV422P."""
    yuv = [(180, 128, 128), (162, 44, 142), (131, 156, 44), (112, 72, 58),
           (84, 184, 198), (65, 100, 212), (35, 212, 114), (16, 128, 128)]
    f = Yuv422(w, h, pad)
    for x in range(w):
        c = yuv[(8 * ((x + rot) % w)) // w]
        f.pix(0)[:, x] = c[0]
        if x % 2 == 0:
            f.pix(1)[:, x // 2] = c[1]
            f.pix(2)[:, x // 2] = c[2]
    return f


def _bind_tocomp_oracle():
    o = oracle()
    if not hasattr(o, "_tocomp_bound"):
        o.tocomp_oracle_process.argtypes = [C.POINTER(Params), C.POINTER(OracleRng),
                                            C.POINTER(TocompPlanes), C.c_uint, C.c_uint64, C.c_int]
        o.tocomp_oracle_process.restype = C.c_int
        o.tocomp_oracle_render_field.argtypes = [C.POINTER(TocompPlanes), C.POINTER(TocompPlanes),
                                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
        o.tocomp_oracle_render_field.restype = None
        o.tocomp_oracle_black_key_feedback.argtypes = [C.POINTER(TocompPlanes), C.POINTER(TocompPlanes),
                                                       C.c_uint, C.c_int]
        o.tocomp_oracle_black_key_feedback.restype = None
        o.tocomp_oracle_output_frame.argtypes = [C.POINTER(TocompPlanes), C.POINTER(TocompPlanes),
                                                 C.c_uint, C.c_int]
        o.tocomp_oracle_output_frame.restype = None
        o._tocomp_bound = True
    return o


def make_params_tocomp(flags=(), **overrides):
    return ntscsim.make_params_to_composite(flags, **overrides)


class TocompOracleStream:
    def __init__(self, params, oob=OOB_DEFINED):
        self.p, self.oob = params, oob
        self.g = OracleRng()
        oracle().ntsc_oracle_rng_seed(C.byref(self.g), 1)
        _bind_tocomp_oracle()

    @property
    def rng_pos(self):
        return int(self.g.count)

    def skip(self, n):
        oracle().ntsc_oracle_rng_discard(C.byref(self.g), n)

    def process(self, frame, field, fieldno):
        t = frame.cplanes()
        rc = oracle().tocomp_oracle_process(C.byref(self.p), C.byref(self.g), C.byref(t), field,
                                            fieldno, self.oob)
        assert rc == 0


def tocomp_oracle_render_field(dst, src, is420, interlaced, tff, second, field):
    o = _bind_tocomp_oracle()
    d, s = dst.cplanes(), src.cplanes()
    o.tocomp_oracle_render_field(C.byref(d), C.byref(s), is420, interlaced, tff, second, field)


def tocomp_oracle_black_key(dst, flt, field, level):
    o = _bind_tocomp_oracle()
    d, f = dst.cplanes(), flt.cplanes()
    o.tocomp_oracle_black_key_feedback(C.byref(d), C.byref(f), field, level)


def tocomp_oracle_output_frame(bob, frame, field, mode):
    o = _bind_tocomp_oracle()
    b, f = bob.cplanes(), frame.cplanes()
    o.tocomp_oracle_output_frame(C.byref(b), C.byref(f), field, mode)


def have_tocomp_ref():
    return os.path.exists(TOCOMP_REF_SO)


_tocomp_ref = None


def tocomp_ref():
    global _tocomp_ref
    if _tocomp_ref is None:
        lib = C.CDLL(TOCOMP_REF_SO)
        pp = C.POINTER(C.POINTER(C.c_uint8))
        ip = C.POINTER(C.c_int)
        lib.tocomp_ref_set_params.argtypes = [C.POINTER(Params)]
        lib.tocomp_ref_srand.argtypes = [C.c_uint]
        lib.tocomp_ref_process.argtypes = [pp, ip, C.c_int, C.c_int, C.c_uint, C.c_ulonglong]
        lib.tocomp_ref_render_field.argtypes = [pp, ip, C.c_int, C.c_int, pp, ip, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_uint]
        lib.tocomp_ref_black_key_feedback.argtypes = [pp, ip, pp, ip, C.c_int, C.c_int, C.c_uint]
        lib.tocomp_ref_output_frame.argtypes = [pp, ip, pp, ip, C.c_int, C.c_int, C.c_uint, C.c_int]
        for f in (lib.tocomp_ref_set_params, lib.tocomp_ref_srand, lib.tocomp_ref_process,
                  lib.tocomp_ref_render_field, lib.tocomp_ref_black_key_feedback,
                  lib.tocomp_ref_output_frame):
            f.restype = None
        _tocomp_ref = lib
    return _tocomp_ref


class TocompRefStream:
    def __init__(self, params):
        tocomp_ref().tocomp_ref_set_params(C.byref(params))
        tocomp_ref().tocomp_ref_srand(1)

    def process(self, frame, field, fieldno):
        d, ls = frame.ptr_arrays()
        tocomp_ref().tocomp_ref_process(C.cast(d, C.POINTER(C.POINTER(C.c_uint8))), ls, frame.w,
                                        frame.h, field, fieldno)


def oracle_bgra_to_yuv(bgra, is420):
    """bgra: uint8 [H, W, 4] -> (Y [H, W], U, V [(H+1)//2 or H, W//2]) by the oracle's definition."""
    h, w = bgra.shape[:2]
    ch = (h + 1) // 2 if is420 else h
    y = np.zeros((h, w), np.uint8)
    u = np.zeros((ch, w // 2), np.uint8)
    v = np.zeros((ch, w // 2), np.uint8)
    src = np.ascontiguousarray(bgra)
    oracle().ntsc_oracle_bgra_to_yuv(_ptr(src), w * 4, w, h, _ptr(y), w, _ptr(u), w // 2, _ptr(v), w // 2,
                                     1 if is420 else 0)
    return y, u, v


# ---- f2 input side: the definition of csrc/ntsc_scale.hip restated in numpy (test infrastructure)
def _scale_pos(n_dst, n_src):
    x = np.arange(n_dst, dtype=np.int64)
    pos = (((2 * x + 1) * n_src) << 15) // n_dst - 32768
    pos = np.clip(pos, 0, (n_src - 1) << 16)
    i0 = pos >> 16
    return i0, np.minimum(i0 + 1, n_src - 1), (pos >> 8) & 255


def _plane_resample(p, W, H):
    """uint8 [h, w] or [h, w, c] -> int64 [H, W(, c)] by the 8-bit-weight bilinear rule."""
    p = p.astype(np.int64)
    x0, x1, fx = _scale_pos(W, p.shape[1])
    y0, y1, fy = _scale_pos(H, p.shape[0])
    if p.ndim == 3:
        fx = fx[None, :, None]; fy = fy[:, None, None]
    else:
        fx = fx[None, :]; fy = fy[:, None]
    top = p[y0][:, x0] * (256 - fx) + p[y0][:, x1] * fx
    bot = p[y1][:, x0] * (256 - fx) + p[y1][:, x1] * fx
    return (top * (256 - fy) + bot * fy + 32768) >> 16


def oracle_scale_to_bgra(planes, fmt, W, H):
    """planes: [bgra [h, w, 4]] or [Y [h, w], U, V] (uint8); fmt 0 BGRA, 1 YUV420P, 2 YUV422P."""
    if fmt == 0:
        return np.ascontiguousarray(_plane_resample(planes[0], W, H).astype(np.uint8))
    y, u, v = (_plane_resample(p, W, H) for p in planes)
    c, d, e = y - 16, u - 128, v - 128
    r = np.clip((298 * c + 409 * e + 128) >> 8, 0, 255)
    g = np.clip((298 * c - 100 * d - 208 * e + 128) >> 8, 0, 255)
    b = np.clip((298 * c + 516 * d + 128) >> 8, 0, 255)
    return np.ascontiguousarray(np.stack([b, g, r, np.full_like(b, 255)], axis=-1).astype(np.uint8))


# ---- the raw-composite decoder (ffmpeg_raw28ntsc.cpp): oracle + reference extract ---------------
RAW28_REF_SO = os.path.join(ROOT, "oracle", "_ref", "libraw28_ref.so")
RAW28_FLAGS = ("mark_sync", "disable_sync", "disable_wp_equ", "show_subcarrier", "disable_subcarrier",
               "disable_equalization")


class Raw28OracleOpts(C.Structure):       # struct raw28_opts (oracle/raw28_oracle.h)
    _fields_ = [("sample_rate", C.c_double)] + [(n, C.c_int32) for n in RAW28_FLAGS]


def raw28_oracle_opts(sample_rate=0.0, **kw):
    o = Raw28OracleOpts()
    o.sample_rate = float(sample_rate)
    for k, v in kw.items():
        assert k in RAW28_FLAGS
        setattr(o, k, int(v))
    return o


_raw28_bound = False


def _bind_raw28_oracle():
    global _raw28_bound
    o = oracle()
    if not _raw28_bound:
        o.raw28_synth_capture.restype = C.c_size_t
        o.raw28_synth_capture.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int]
        o.raw28_oracle_open.restype = C.c_void_p
        o.raw28_oracle_open.argtypes = [C.POINTER(Raw28OracleOpts), C.c_void_p, C.c_size_t]
        o.raw28_oracle_next_field.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        o.raw28_oracle_close.argtypes = [C.c_void_p]
        o.raw28_oracle_geometry.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
        o.raw28_oracle_levels.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        o.raw28_oracle_front.argtypes = [C.POINTER(Raw28OracleOpts), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _raw28_bound = True
    return o


def raw28_capture(fields, seed=1, noise=3, cut=0):
    """Synthetic 8 x fsc capture of `fields` fields (oracle/raw28_oracle.c); cut = samples dropped from
    its start (an arbitrary tuning-in point)."""
    o = _bind_raw28_oracle()
    buf = np.zeros(fields * 477750 + 16, np.uint8)
    n = o.raw28_synth_capture(buf.ctypes.data, buf.size, fields, seed, noise)
    return np.ascontiguousarray(buf[cut:n])


def raw28_oracle_run(opts, capture, max_fields=1 << 30):
    """(frames [F, H, W*4], (blank, white, read_pos)) of the oracle on a capture."""
    o = _bind_raw28_oracle()
    d = o.raw28_oracle_open(C.byref(opts), capture.ctypes.data, capture.size)
    w, h, sl = C.c_int(), C.c_int(), C.c_int()
    o.raw28_oracle_geometry(d, C.byref(w), C.byref(h), C.byref(sl))
    frames = []
    while len(frames) < max_fields:
        f = np.empty((h.value, w.value * 4), np.uint8)
        if not o.raw28_oracle_next_field(d, f.ctypes.data, w.value * 4):
            break
        frames.append(f)
    b, wh, rp = C.c_double(), C.c_double(), C.c_uint64()
    o.raw28_oracle_levels(d, C.byref(b), C.byref(wh), C.byref(rp))
    o.raw28_oracle_close(d)
    out = np.stack(frames) if frames else np.zeros((0, h.value, w.value * 4), np.uint8)
    return out, (b.value, wh.value, rp.value)


def raw28_oracle_front(opts, capture):
    o = _bind_raw28_oracle()
    h = np.empty(capture.size, np.uint8)
    r = np.empty(capture.size, np.uint8)
    o.raw28_oracle_front(C.byref(opts), capture.ctypes.data, capture.size, h.ctypes.data, r.ctypes.data)
    return h, r


def have_raw28_ref():
    return os.path.exists(RAW28_REF_SO)


_raw28_ref = None


def raw28_ref_run(opts, capture, path, max_fields=64):
    """The reference extract (oracle/_ref/libraw28_ref.so) on the capture written to `path`."""
    global _raw28_ref
    if _raw28_ref is None:
        _raw28_ref = C.CDLL(RAW28_REF_SO)
        _raw28_ref.raw28_ref_run.argtypes = ([C.POINTER(Raw28OracleOpts), C.c_char_p, C.c_void_p, C.c_int] +
                                             [C.POINTER(C.c_int)] * 3 + [C.POINTER(C.c_double)] * 2 +
                                             [C.POINTER(C.c_ulonglong)])
        _raw28_ref.raw28_ref_front.argtypes = [C.POINTER(Raw28OracleOpts), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    with open(path, "wb") as f:
        f.write(capture.tobytes())
    w, h, sl = C.c_int(), C.c_int(), C.c_int()
    b, wh, rp = C.c_double(), C.c_double(), C.c_ulonglong()
    # frame size of this sample rate (compute_NTSC :247-256, preset_NTSC :395-402)
    rate = opts.sample_rate if opts.sample_rate > 0 else (315000000.00 * 8.0) / 88.00
    length = int((rate / (30000.00 / 1001.00)) / 525.00 + 0.5)
    width = (length + 1) & ~1
    out = np.zeros((max_fields, 262, width * 4), np.uint8)
    n = _raw28_ref.raw28_ref_run(C.byref(opts), str(path).encode(), out.ctypes.data, max_fields, C.byref(w),
                                 C.byref(h), C.byref(sl), C.byref(b), C.byref(wh), C.byref(rp))
    assert n >= 0 and (w.value, h.value, sl.value) == (width, 262, length)
    return out[:n], (b.value, wh.value, rp.value)


def raw28_ref_front(opts, capture):
    raw28_ref_run  # noqa: the binding above
    global _raw28_ref
    if _raw28_ref is None:
        _raw28_ref = C.CDLL(RAW28_REF_SO)
        _raw28_ref.raw28_ref_front.argtypes = [C.POINTER(Raw28OracleOpts), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    h = np.empty(capture.size, np.uint8)
    r = np.empty(capture.size, np.uint8)
    _raw28_ref.raw28_ref_front(C.byref(opts), capture.ctypes.data, capture.size, h.ctypes.data, r.ctypes.data)
    return h, r
