"""ctypes binding of include/ntscsim.h (the product C-ABI, libntscsim.so).

The shared object loads without a GPU (so the CPU test-suite can check the exported symbols and
use the host-side parse_argv mirror); ntscsim_create() then fails with NTSCSIM_E_NODEV.  There is
no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# NTSCSIM_LIB: developer override (A/B builds of the same ABI); default = the in-tree library
PRODUCT_SO = os.environ.get("NTSCSIM_LIB") or os.path.join(PKG_DIR, "libntscsim.so")

OK, E_ARG, E_SIZE, E_NODEV, E_HIP, E_NOMEM, E_PARAM, E_FLAG, E_HELP, E_INTERNAL = \
    0, -1, -2, -3, -4, -5, -6, -7, -8, -9
RNG_AUTO = 0xFFFFFFFFFFFFFFFF
MODE_EXACT, MODE_FAST32, MODE_FLOAT = 0, 1, 2
DESC_INTERLACED, DESC_TFF, DESC_BOB = 1, 2, 0x100
SUBMIT_SAME_SRC, SUBMIT_SRC_STABLE = 0x10000, 0x20000
TICKET_ALL = 0xFFFFFFFFFFFFFFFF

# every symbol include/ntscsim.h declares
EXPORTS = (
    "ntscsim_params_init", "ntscsim_cli_init", "ntscsim_params_parse_argv",
    "ntscsim_params_validate", "ntscsim_rng_calls_per_field", "ntscsim_rng_draw",
    "ntscsim_create", "ntscsim_destroy", "ntscsim_strerror", "ntscsim_last_error",
    "ntscsim_set_mode", "ntscsim_get_rng_pos", "ntscsim_set_rng_pos", "ntscsim_field", "ntscsim_frames_host",
    "ntscsim_fields_device",
    "ntscsim_batch_create", "ntscsim_batch_run", "ntscsim_batch_destroy",
    "ntscsim_sync", "ntscsim_set_profiling", "ntscsim_get_timings_ms", "ntscsim_set_launch_form",
    "ntscsim_debug_read_composite", "ntscsim_debug_set_warmup",
    "ntscsim_debug_force_generic", "ntscsim_debug_no_fast_decode", "ntscsim_debug_last_kernels", "ntscsim_debug_fast_plane_ok", "ntscsim_debug_field_stats",
    "ntscsim_params_init_to_composite", "ntscsim_params_parse_argv_to_composite",
    "ntscsim_fields422_device", "ntscsim_output422_device", "ntscsim_bgra_to_yuv_device", "ntscsim_rng_calls_per_field_422",
    "ntscsim_scale_to_bgra_device", "ntscsim_frames_host_scaled",
    "ntscsim_batch422_create", "ntscsim_batch422_run", "ntscsim_batch422_destroy",
    "ntscsim_raw28_opts_init", "ntscsim_raw28_parse_argv", "ntscsim_raw28_geometry", "ntscsim_raw28_create",
    "ntscsim_raw28_destroy", "ntscsim_raw28_last_error", "ntscsim_raw28_decode", "ntscsim_raw28_decode_device",
    "ntscsim_raw28_stream_reset", "ntscsim_raw28_stream_push",
    "ntscsim_raw28_get_levels", "ntscsim_raw28_debug_set_speculation", "ntscsim_raw28_debug_stats", "ntscsim_raw28_debug_pick_chunk",
    "ntscsim_raw28_debug_read_front",
    "ntscsim_submit_opts_init", "ntscsim_submit_configure", "ntscsim_submit", "ntscsim_flush", "ntscsim_wait",
    "ntscsim_host_unpin", "ntscsim_submit_stats",
    "ntscsim_host_pin", "ntscsim_host_alloc", "ntscsim_host_free", "ntscsim_host_frame_alloc", "ntscsim_set_pin_policy",
    "ntscsim_field422", "ntscsim_submit422", "ntscsim_submit422_configure", "ntscsim_submit422_stats",
    "ntscsim_pool_create", "ntscsim_pool_destroy", "ntscsim_pool_size", "ntscsim_pool_ctx", "ntscsim_pool_set_block",
    "ntscsim_pool_get_rng_pos", "ntscsim_pool_set_rng_pos", "ntscsim_pool_last_error", "ntscsim_pool_frames_host",
)


class SubmitOpts(C.Structure):
    """struct ntscsim_submit_opts -- keep in lock-step with include/ntscsim.h."""
    _fields_ = [("struct_size", C.c_uint32), ("depth", C.c_int32), ("slots", C.c_int32), ("lanes", C.c_int32),
                ("pin_caller_buffers", C.c_int32), ("_pad", C.c_int32), ("min_pin_bytes", C.c_size_t)]


class Frame422(C.Structure):
    """struct ntscsim_frame422 -- keep in lock-step with include/ntscsim.h."""
    _fields_ = [("data", C.c_void_p * 3), ("linesize", C.c_int32 * 3), ("_pad", C.c_int32)]


class Loop422(C.Structure):
    """struct ntscsim_loop422 -- keep in lock-step with include/ntscsim.h."""
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32), ("src_height", C.c_int32),
                ("frame", Frame422), ("src", Frame422), ("filter", Frame422), ("out", Frame422),
                ("field", C.c_uint32), ("flags", C.c_uint32), ("out_mode", C.c_uint32), ("out_field", C.c_uint32),
                ("fieldno", C.c_uint64)]


SUBMIT422_DIRTY = 0x40000


class Raw28Opts(C.Structure):
    """struct ntscsim_raw28_opts -- keep in lock-step with include/ntscsim.h."""
    _fields_ = [("struct_size", C.c_uint32), ("_pad0", C.c_uint32), ("sample_rate", C.c_double),
                ("mark_sync", C.c_int32), ("disable_sync", C.c_int32), ("disable_wp_equ", C.c_int32),
                ("show_subcarrier", C.c_int32), ("disable_subcarrier", C.c_int32),
                ("disable_equalization", C.c_int32)]


class Params(C.Structure):
    """struct ntscsim_params -- keep in lock-step with include/ntscsim.h."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("tv_standard", C.c_int32),
        ("output_width", C.c_int32),
        ("output_height", C.c_int32),
        ("video_scanline_phase_shift", C.c_int32),
        ("video_scanline_phase_shift_offset", C.c_int32),
        ("composite_preemphasis", C.c_double),
        ("composite_preemphasis_cut", C.c_double),
        ("vhs_out_sharpen", C.c_double),
        ("vhs_head_switching", C.c_int32),
        ("_pad0", C.c_int32),
        ("vhs_head_switching_point", C.c_double),
        ("vhs_head_switching_phase", C.c_double),
        ("vhs_head_switching_phase_noise", C.c_double),
        ("composite_in_chroma_lowpass", C.c_int32),
        ("composite_out_chroma_lowpass", C.c_int32),
        ("composite_out_chroma_lowpass_lite", C.c_int32),
        ("video_yc_recombine", C.c_int32),
        ("video_chroma_noise", C.c_int32),
        ("video_chroma_phase_noise", C.c_int32),
        ("video_chroma_loss", C.c_int32),
        ("video_noise", C.c_int32),
        ("subcarrier_amplitude", C.c_int32),
        ("subcarrier_amplitude_back", C.c_int32),
        ("emulating_vhs", C.c_int32),
        ("nocolor_subcarrier", C.c_int32),
        ("nocolor_subcarrier_after_yc_sep", C.c_int32),
        ("vhs_chroma_vert_blend", C.c_int32),
        ("vhs_svideo_out", C.c_int32),
        ("enable_composite_emulation", C.c_int32),
        ("output_vhs_tape_speed", C.c_int32),
        ("black_key_level_feedback", C.c_int32),
        ("vhs_out_sharpen_chroma", C.c_double),
        ("ghost_taps", C.c_int32),
        ("ghost_delay", C.c_int32 * 4),
        ("ghost_gain", C.c_int32 * 4),
        ("_pad1", C.c_int32),
    ]


class FieldDesc(C.Structure):
    """struct ntscsim_field_desc"""
    _fields_ = [
        ("src_dev", C.c_void_p),
        ("dst_dev", C.c_void_p),
        ("src_linesize", C.c_int32),
        ("dst_linesize", C.c_int32),
        ("field", C.c_uint32),
        ("flags", C.c_uint32),
        ("fieldno", C.c_uint64),
        ("rng_pos", C.c_uint64),
    ]


class Field422Desc(C.Structure):
    """struct ntscsim_field422_desc"""
    _fields_ = [
        ("dst_dev", C.c_void_p * 3),
        ("src_dev", C.c_void_p * 3),
        ("flt_dev", C.c_void_p * 3),
        ("dst_linesize", C.c_int32 * 3),
        ("src_linesize", C.c_int32 * 3),
        ("flt_linesize", C.c_int32 * 3),
        ("src_height", C.c_int32),
        ("field", C.c_uint32),
        ("flags", C.c_uint32),
        ("_pad", C.c_uint32),
        ("fieldno", C.c_uint64),
        ("rng_pos", C.c_uint64),
    ]


F422_INTERLACED, F422_TFF, F422_SRC420, F422_SECOND, F422_NOCOMP = 1, 2, 4, 8, 16


class Out422Desc(C.Structure):
    """struct ntscsim_out422_desc"""
    _fields_ = [
        ("frame_dev", C.c_void_p * 3),
        ("bob_dev", C.c_void_p * 3),
        ("frame_linesize", C.c_int32 * 3),
        ("bob_linesize", C.c_int32 * 3),
        ("field", C.c_uint32),
        ("mode", C.c_uint32),
    ]


OUT422_BOB422, OUT422_BOB420, OUT422_INTERLACED420, OUT422_FRAME = 0, 1, 2, 3


class YuvDesc(C.Structure):
    """struct ntscsim_yuv_desc"""
    _fields_ = [
        ("bgra_dev", C.c_void_p),
        ("yuv_dev", C.c_void_p * 3),
        ("bgra_linesize", C.c_int32),
        ("yuv_linesize", C.c_int32 * 3),
    ]


PIX_YUV420P, PIX_YUV422P = 0, 1
HOST_YUV420P, HOST_YUV422P = 0x1000, 0x2000
SRC_BGRA, SRC_YUV420P, SRC_YUV422P = 0, 1, 2


class ScaleDesc(C.Structure):
    """struct ntscsim_scale_desc"""
    _fields_ = [
        ("src_dev", C.c_void_p * 3),
        ("bgra_dev", C.c_void_p),
        ("src_linesize", C.c_int32 * 3),
        ("bgra_linesize", C.c_int32),
        ("src_width", C.c_int32),
        ("src_height", C.c_int32),
        ("src_format", C.c_int32),
        ("_pad", C.c_int32),
    ]


class HostSource(C.Structure):
    """struct ntscsim_host_source"""
    _fields_ = [
        ("format", C.c_int32),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("linesize", C.c_int32 * 3),
        ("plane_offset", C.c_size_t * 3),
        ("frame_bytes", C.c_size_t),
    ]

_u8p = C.POINTER(C.c_uint8)
_lib = None


def lib():
    """Load libntscsim.so; fail loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(PRODUCT_SO):
        raise RuntimeError(
            "%s is missing -- build it with `python -c \"import __graft_entry__ as g; g.build()\"` "
            "(or `make -C composite-video-simulator_amd/csrc`)" % PRODUCT_SO)
    # Load order: torch carries its own copy of the HIP runtime; imported AFTER libntscsim.so (which links /opt/rocm's)
    # it finds no device.  The package's __init__ (the half that hands torch tensors to this library) therefore imports
    # torch before anything here runs; a program that uses _capi alone and torch later must import torch first itself.
    L = C.CDLL(PRODUCT_SO)
    L.ntscsim_params_init.argtypes = [C.POINTER(Params)]
    L.ntscsim_params_init.restype = None
    L.ntscsim_cli_init.argtypes = [C.c_void_p]
    L.ntscsim_cli_init.restype = None
    L.ntscsim_params_parse_argv.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int,
                                            C.POINTER(C.c_char_p), C.c_int]
    L.ntscsim_params_parse_argv.restype = C.c_int
    L.ntscsim_params_validate.argtypes = [C.POINTER(Params)]
    L.ntscsim_params_validate.restype = C.c_int
    L.ntscsim_rng_calls_per_field.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_uint]
    L.ntscsim_rng_calls_per_field.restype = C.c_uint64
    L.ntscsim_rng_draw.argtypes = [C.c_uint64, C.c_size_t, C.POINTER(C.c_uint32)]
    L.ntscsim_rng_draw.restype = None
    L.ntscsim_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
    L.ntscsim_create.restype = C.c_int
    L.ntscsim_destroy.argtypes = [C.c_void_p]
    L.ntscsim_destroy.restype = None
    L.ntscsim_strerror.argtypes = [C.c_int]
    L.ntscsim_strerror.restype = C.c_char_p
    L.ntscsim_last_error.argtypes = [C.c_void_p]
    L.ntscsim_last_error.restype = C.c_char_p
    L.ntscsim_set_mode.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_set_mode.restype = C.c_int
    L.ntscsim_get_rng_pos.argtypes = [C.c_void_p]
    L.ntscsim_get_rng_pos.restype = C.c_uint64
    L.ntscsim_set_rng_pos.argtypes = [C.c_void_p, C.c_uint64]
    L.ntscsim_set_rng_pos.restype = None
    L.ntscsim_field.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int,
                                C.c_int, C.c_int, C.c_uint, C.c_uint64]
    L.ntscsim_field.restype = C.c_int
    L.ntscsim_frames_host.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p,
                                      C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32,
                                      C.c_int]
    L.ntscsim_frames_host.restype = C.c_int
    L.ntscsim_fields_device.argtypes = [C.c_void_p, C.POINTER(FieldDesc), C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]
    L.ntscsim_fields_device.restype = C.c_int
    L.ntscsim_batch_create.argtypes = [C.c_void_p, C.POINTER(FieldDesc), C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_void_p)]
    L.ntscsim_batch_create.restype = C.c_int
    L.ntscsim_batch_run.argtypes = [C.c_void_p, C.c_void_p]
    L.ntscsim_batch_run.restype = C.c_int
    L.ntscsim_batch_destroy.argtypes = [C.c_void_p]
    L.ntscsim_batch_destroy.restype = None
    L.ntscsim_sync.argtypes = [C.c_void_p]
    L.ntscsim_sync.restype = C.c_int
    L.ntscsim_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_set_profiling.restype = None
    L.ntscsim_set_launch_form.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_set_launch_form.restype = C.c_int
    L.ntscsim_get_timings_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.ntscsim_get_timings_ms.restype = C.c_int
    L.ntscsim_debug_read_composite.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t]
    L.ntscsim_debug_read_composite.restype = C.c_int
    L.ntscsim_debug_set_warmup.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ntscsim_debug_set_warmup.restype = None
    L.ntscsim_params_init_to_composite.argtypes = [C.POINTER(Params)]
    L.ntscsim_params_init_to_composite.restype = None
    L.ntscsim_params_parse_argv_to_composite.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int,
                                                         C.POINTER(C.c_char_p), C.c_int]
    L.ntscsim_params_parse_argv_to_composite.restype = C.c_int
    L.ntscsim_fields422_device.argtypes = [C.c_void_p, C.POINTER(Field422Desc), C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]
    L.ntscsim_fields422_device.restype = C.c_int
    L.ntscsim_output422_device.argtypes = [C.c_void_p, C.POINTER(Out422Desc), C.c_int, C.c_int,
                                           C.c_int, C.c_void_p]
    L.ntscsim_output422_device.restype = C.c_int
    L.ntscsim_bgra_to_yuv_device.argtypes = [C.c_void_p, C.POINTER(YuvDesc), C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_void_p]
    L.ntscsim_bgra_to_yuv_device.restype = C.c_int
    L.ntscsim_scale_to_bgra_device.argtypes = [C.c_void_p, C.POINTER(ScaleDesc), C.c_int, C.c_int, C.c_int,
                                               C.c_void_p]
    L.ntscsim_scale_to_bgra_device.restype = C.c_int
    L.ntscsim_frames_host_scaled.argtypes = [C.c_void_p, C.POINTER(HostSource), C.POINTER(C.c_uint8), C.c_size_t,
                                             C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, C.c_int, C.c_int,
                                             C.c_uint64, C.c_uint32, C.c_int]
    L.ntscsim_frames_host_scaled.restype = C.c_int
    L.ntscsim_rng_calls_per_field_422.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_uint]
    L.ntscsim_rng_calls_per_field_422.restype = C.c_uint64
    L.ntscsim_debug_force_generic.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_debug_force_generic.restype = None
    L.ntscsim_debug_no_fast_decode.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_debug_no_fast_decode.restype = None
    L.ntscsim_debug_last_kernels.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.ntscsim_debug_last_kernels.restype = C.c_int
    L.ntscsim_debug_fast_plane_ok.argtypes = [C.c_int] * 4
    L.ntscsim_debug_fast_plane_ok.restype = C.c_int
    L.ntscsim_debug_field_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ntscsim_debug_field_stats.restype = None
    L.ntscsim_batch422_create.argtypes = [C.c_void_p, C.POINTER(Field422Desc), C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_void_p)]
    L.ntscsim_batch422_create.restype = C.c_int
    L.ntscsim_batch422_run.argtypes = [C.c_void_p, C.c_void_p]
    L.ntscsim_batch422_run.restype = C.c_int
    L.ntscsim_batch422_destroy.argtypes = [C.c_void_p]
    L.ntscsim_batch422_destroy.restype = None
    L.ntscsim_raw28_opts_init.argtypes = [C.POINTER(Raw28Opts)]
    L.ntscsim_raw28_opts_init.restype = None
    L.ntscsim_raw28_parse_argv.argtypes = [C.POINTER(Raw28Opts), C.c_int, C.POINTER(C.c_char_p), C.c_int]
    L.ntscsim_raw28_parse_argv.restype = C.c_int
    L.ntscsim_raw28_geometry.argtypes = [C.POINTER(Raw28Opts)] + [C.POINTER(C.c_int)] * 3
    L.ntscsim_raw28_geometry.restype = C.c_int
    L.ntscsim_raw28_create.argtypes = [C.POINTER(Raw28Opts), C.c_int, C.POINTER(C.c_void_p)]
    L.ntscsim_raw28_create.restype = C.c_int
    L.ntscsim_raw28_destroy.argtypes = [C.c_void_p]
    L.ntscsim_raw28_destroy.restype = None
    L.ntscsim_raw28_last_error.argtypes = [C.c_void_p]
    L.ntscsim_raw28_last_error.restype = C.c_char_p
    for fn in (L.ntscsim_raw28_decode, L.ntscsim_raw28_decode_device):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)]
        fn.restype = C.c_int
    L.ntscsim_raw28_stream_reset.argtypes = [C.c_void_p]
    L.ntscsim_raw28_stream_reset.restype = C.c_int
    L.ntscsim_raw28_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                            C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.ntscsim_raw28_stream_push.restype = C.c_int
    L.ntscsim_raw28_get_levels.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.ntscsim_raw28_get_levels.restype = C.c_int
    L.ntscsim_raw28_debug_set_speculation.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ntscsim_raw28_debug_set_speculation.restype = None
    L.ntscsim_raw28_debug_pick_chunk.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ntscsim_raw28_debug_pick_chunk.restype = None
    L.ntscsim_raw28_debug_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.ntscsim_raw28_debug_stats.restype = None
    L.ntscsim_raw28_debug_read_front.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ntscsim_raw28_debug_read_front.restype = C.c_int
    L.ntscsim_submit_opts_init.argtypes = [C.POINTER(SubmitOpts)]
    L.ntscsim_submit_opts_init.restype = None
    L.ntscsim_submit_configure.argtypes = [C.c_void_p, C.POINTER(SubmitOpts)]
    L.ntscsim_submit_configure.restype = C.c_int
    L.ntscsim_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_uint, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]
    L.ntscsim_submit.restype = C.c_int
    L.ntscsim_flush.argtypes = [C.c_void_p]
    L.ntscsim_flush.restype = C.c_int
    L.ntscsim_wait.argtypes = [C.c_void_p, C.c_uint64]
    L.ntscsim_wait.restype = C.c_int
    L.ntscsim_host_unpin.argtypes = [C.c_void_p, C.c_void_p]
    L.ntscsim_host_unpin.restype = C.c_int
    L.ntscsim_submit_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ntscsim_submit_stats.restype = None
    L.ntscsim_host_pin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.ntscsim_host_pin.restype = C.c_int
    L.ntscsim_host_alloc.argtypes = [C.c_size_t]
    L.ntscsim_host_alloc.restype = C.c_void_p
    L.ntscsim_host_free.argtypes = [C.c_void_p]
    L.ntscsim_host_free.restype = None
    L.ntscsim_host_frame_alloc.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.ntscsim_host_frame_alloc.restype = C.c_int
    L.ntscsim_set_pin_policy.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_set_pin_policy.restype = C.c_int
    L.ntscsim_field422.argtypes = [C.c_void_p, C.POINTER(Loop422)]
    L.ntscsim_field422.restype = C.c_int
    L.ntscsim_submit422.argtypes = [C.c_void_p, C.POINTER(Loop422), C.c_uint32, C.POINTER(C.c_uint64)]
    L.ntscsim_submit422.restype = C.c_int
    L.ntscsim_submit422_configure.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.ntscsim_submit422_configure.restype = C.c_int
    L.ntscsim_submit422_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ntscsim_submit422_stats.restype = None
    L.ntscsim_pool_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.ntscsim_pool_create.restype = C.c_int
    L.ntscsim_pool_destroy.argtypes = [C.c_void_p]
    L.ntscsim_pool_destroy.restype = None
    L.ntscsim_pool_size.argtypes = [C.c_void_p]
    L.ntscsim_pool_size.restype = C.c_int
    L.ntscsim_pool_ctx.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_pool_ctx.restype = C.c_void_p
    L.ntscsim_pool_set_block.argtypes = [C.c_void_p, C.c_int]
    L.ntscsim_pool_set_block.restype = C.c_int
    L.ntscsim_pool_get_rng_pos.argtypes = [C.c_void_p]
    L.ntscsim_pool_get_rng_pos.restype = C.c_uint64
    L.ntscsim_pool_set_rng_pos.argtypes = [C.c_void_p, C.c_uint64]
    L.ntscsim_pool_set_rng_pos.restype = None
    L.ntscsim_pool_last_error.argtypes = [C.c_void_p]
    L.ntscsim_pool_last_error.restype = C.c_char_p
    L.ntscsim_pool_frames_host.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.c_int, C.c_int, _u8p, C.c_size_t, C.c_int,
                                           C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int]
    L.ntscsim_pool_frames_host.restype = C.c_int
    _lib = L
    return L


class NtscsimError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().ntscsim_strerror(code).decode()
        super().__init__("ntscsim error %d: %s%s" % (code, msg, (" -- " + detail) if detail else ""))


def make_params_to_composite(flags=(), **overrides):
    """ntscsim_params from ffmpeg_to_composite's switches (ffmpeg_to_composite.cpp :1325)."""
    L = lib()
    p = Params()
    L.ntscsim_params_init_to_composite(C.byref(p))
    argv = [b"ffmpeg_to_composite"] + [str(f).encode() for f in flags]
    arr = (C.c_char_p * len(argv))(*argv)
    rc = L.ntscsim_params_parse_argv_to_composite(C.byref(p), None, len(argv), arr, 0)
    if rc != OK:
        raise NtscsimError(rc, "parse_argv_to_composite(%r)" % (list(flags),))
    for k, v in overrides.items():
        if k not in dict(Params._fields_):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def make_params(flags=(), **overrides):
    """ntscsim_params from the reference's CLI switches (ffmpeg_ntsc.cpp parse_argv :972)."""
    L = lib()
    p = Params()
    L.ntscsim_params_init(C.byref(p))
    argv = [b"ffmpeg_ntsc"] + [str(f).encode() for f in flags]
    arr = (C.c_char_p * len(argv))(*argv)
    rc = L.ntscsim_params_parse_argv(C.byref(p), None, len(argv), arr, 0)
    if rc != OK:
        raise NtscsimError(rc, "parse_argv(%r)" % (list(flags),))
    for k, v in overrides.items():
        if k not in dict(Params._fields_):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def make_raw28_opts(flags=()):
    """ntscsim_raw28_opts from ffmpeg_raw28ntsc's switches (ffmpeg_raw28ntsc.cpp parse_argv :442)."""
    L = lib()
    o = Raw28Opts()
    L.ntscsim_raw28_opts_init(C.byref(o))
    argv = [b"ffmpeg_raw28ntsc"] + [str(f).encode() for f in flags]
    arr = (C.c_char_p * len(argv))(*argv)
    rc = L.ntscsim_raw28_parse_argv(C.byref(o), len(argv), arr, 1)
    if rc != OK:
        raise NtscsimError(rc, "raw28 parse_argv(%r)" % (list(flags),))
    return o
