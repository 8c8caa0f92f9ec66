"""ntscsim -- host-side Python mirror of the reference's per-field interface
(composite_layer(), ffmpeg_ntsc.cpp:1570, and the field loop around it, :2202-2282), calling the
hand-written HIP kernels through the C-ABI of include/ntscsim.h.

torch is used only for device memory, streams and torch.distributed plumbing.
"""
import ctypes as C

try:                # before libntscsim.so is loaded: one HIP runtime for torch and the library (_capi.lib)
    import torch  # noqa: F401
except ImportError:
    pass

from . import _capi
from ._capi import (DESC_BOB, DESC_INTERLACED, DESC_TFF, RNG_AUTO, Field422Desc, FieldDesc,
                    NtscsimError, Out422Desc, YuvDesc, ScaleDesc, HostSource, Params, lib, make_params,
                    make_params_to_composite)

__all__ = ["FieldSimulator", "Pool", "Params", "FieldDesc", "make_params", "NtscsimError", "lib",
           "field_rows", "calls_per_field", "field_schedule"]


def field_rows(height, field):
    """Rows composite_layer touches: y = field, field+2, ... < height."""
    return (height - field + 1) // 2 if height > field else 0


def calls_per_field(params, width, height, field):
    return int(lib().ntscsim_rng_calls_per_field(C.byref(params), width, height, field))


def field_schedule(n_fields, first=0):
    """The reference's field loop (ffmpeg_ntsc.cpp:2202-2229): output field `current` uses
    field parity (current & 1) ^ 1 and fieldno = current."""
    return [((cur & 1) ^ 1, cur) for cur in range(first, first + n_fields)]


class FieldSimulator:
    """One GPU context.  `flags` are the reference's CLI switches, e.g. ("-vhs",)."""

    def __init__(self, flags=(), device=0, params=None):
        self.params = params if params is not None else make_params(flags)
        self._lib = lib()
        h = C.c_void_p()
        rc = self._lib.ntscsim_create(C.byref(self.params), int(device), C.byref(h))
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_create(device=%d)" % device)
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ntscsim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != _capi.OK:
            raise NtscsimError(rc, "%s: %s" % (what, self._lib.ntscsim_last_error(self._h).decode()))

    def set_mode(self, mode):
        """_capi.MODE_EXACT (default, bit-identical to the reference), _capi.MODE_FAST32 or _capi.MODE_FLOAT."""
        self._chk(self._lib.ntscsim_set_mode(self._h, int(mode)), "ntscsim_set_mode")

    # ---- rand() stream position -----------------------------------------------------------
    @property
    def rng_pos(self):
        return int(self._lib.ntscsim_get_rng_pos(self._h))

    @rng_pos.setter
    def rng_pos(self, pos):
        self._lib.ntscsim_set_rng_pos(self._h, int(pos))

    # ---- drop-in for composite_layer(): host (numpy) frames ---------------------------------
    def field_host(self, dst, src, field, fieldno, interlaced=0, tff=0):
        """dst, src: numpy uint8 [H, W, 4] BGRA (C-contiguous).  Writes rows field, field+2, ..."""
        h, w = src.shape[:2]
        assert src.shape == dst.shape and src.shape[2] == 4
        u8p = C.POINTER(C.c_uint8)
        rc = self._lib.ntscsim_field(self._h, src.ctypes.data_as(u8p), src.strides[0],
                                     int(interlaced), int(tff), dst.ctypes.data_as(u8p),
                                     dst.strides[0], w, h, int(field), int(fieldno))
        self._chk(rc, "ntscsim_field")

    # ---- the asynchronous form of the same drop-in: ntscsim_submit() / ntscsim_wait() ------
    def submit_configure(self, depth=32, slots=0, lanes=3, pin=True, min_pin_bytes=256 << 10):
        o = _capi.SubmitOpts()
        self._lib.ntscsim_submit_opts_init(C.byref(o))
        o.depth, o.slots, o.lanes = int(depth), int(slots), int(lanes)
        o.pin_caller_buffers, o.min_pin_bytes = int(bool(pin)), int(min_pin_bytes)
        self._chk(self._lib.ntscsim_submit_configure(self._h, C.byref(o)), "ntscsim_submit_configure")

    def submit(self, dst, src, field, fieldno, interlaced=0, tff=0, bob=False, same_src=False,
               src_stable=False):
        """As field_host(), asynchronously: returns a ticket; dst is complete after wait(ticket).
        dst, src: numpy uint8 [H, W, 4] (rows contiguous; any row stride)."""
        h, w = src.shape[:2]
        assert src.shape == dst.shape and src.shape[2] == 4 and src.strides[1] == 4 and dst.strides[1] == 4
        flags = (DESC_BOB if bob else 0) | (_capi.SUBMIT_SAME_SRC if same_src else 0) | \
                (_capi.SUBMIT_SRC_STABLE if src_stable else 0)
        t = C.c_uint64(0)
        rc = self._lib.ntscsim_submit(self._h, src.ctypes.data, src.strides[0], int(interlaced), int(tff),
                                      dst.ctypes.data, dst.strides[0], w, h, int(field), int(fieldno),
                                      flags, C.byref(t))
        self._chk(rc, "ntscsim_submit")
        return int(t.value)

    def wait(self, ticket=None):
        self._chk(self._lib.ntscsim_wait(self._h, _capi.TICKET_ALL if ticket is None else int(ticket)),
                  "ntscsim_wait")

    def flush(self):
        self._chk(self._lib.ntscsim_flush(self._h), "ntscsim_flush")

    def host_unpin(self, array=None):
        self._chk(self._lib.ntscsim_host_unpin(self._h, None if array is None else array.ctypes.data),
                  "ntscsim_host_unpin")

    def host_pin(self, array):
        """ntscsim_host_pin(): declare a numpy array's memory as the caller's own to pin."""
        self._chk(self._lib.ntscsim_host_pin(self._h, array.ctypes.data, array.nbytes), "ntscsim_host_pin")

    def set_pin_policy(self, policy):
        self._chk(self._lib.ntscsim_set_pin_policy(self._h, int(policy)), "ntscsim_set_pin_policy")

    def submit_stats(self):
        out = (C.c_uint64 * 8)()
        self._lib.ntscsim_submit_stats(self._h, out)
        keys = ("submitted", "launches", "uploads", "uploads_staged", "delivered_direct", "delivered_staged",
                "registrations", "ring_full_waits")
        return dict(zip(keys, [int(v) for v in out]))

    def frames_host(self, dst, src, first_fieldno=0, bob=True, chunk_frames=0, yuv=None):
        """The field loop over host frames: src numpy uint8 [N, H, W, 4], dst [2N, H, W, 4].
        yuv = "420" / "422": dst is uint8 [2N, frame_bytes], each frame Y|U|V planes packed at
        linesize W (chroma W/2)."""
        n, h, w = src.shape[:3]
        assert src.flags.c_contiguous and dst.flags.c_contiguous
        u8p = C.POINTER(C.c_uint8)
        flags = DESC_BOB if bob else 0
        if yuv is None:
            assert dst.shape == (2 * n, h, w, 4)
            dls = dst.strides[1]
        else:
            flags |= _capi.HOST_YUV420P if yuv == "420" else _capi.HOST_YUV422P
            assert dst.ndim == 2 and dst.shape[0] == 2 * n
            dls = w
        rc = self._lib.ntscsim_frames_host(self._h, src.ctypes.data_as(u8p), src.strides[0],
                                           src.strides[1], n, dst.ctypes.data_as(u8p),
                                           dst.strides[0], dls, w, h, int(first_fieldno),
                                           flags, int(chunk_frames))
        self._chk(rc, "ntscsim_frames_host")

    def bgra_to_yuv(self, jobs, width, height, pix_fmt, stream=None):
        """jobs: list of (bgra CUDA uint8 tensor [H, ls], (y, u, v) CUDA uint8 tensors [rows, ls])."""
        arr = (YuvDesc * len(jobs))()
        for d, (bgra, planes) in zip(arr, jobs):
            d.bgra_dev = bgra.data_ptr()
            d.bgra_linesize = bgra.stride(0)
            for k in range(3):
                d.yuv_dev[k] = planes[k].data_ptr()
                d.yuv_linesize[k] = planes[k].stride(0)
        if stream is None:
            stream = self._torch_stream()
        rc = self._lib.ntscsim_bgra_to_yuv_device(self._h, arr, len(jobs), int(width), int(height),
                                                  int(pix_fmt), C.c_void_p(stream))
        self._chk(rc, "ntscsim_bgra_to_yuv_device")

    def scale_to_bgra(self, jobs, width, height, stream=None):
        """jobs: list of (src planes: 1 (BGRA) or 3 (Y, U, V) CUDA uint8 tensors [rows, ls], src_width,
        src_height, format _capi.SRC_*, dst BGRA CUDA uint8 tensor [height, ls])."""
        arr = (ScaleDesc * len(jobs))()
        for d, (planes, sw, sh, fmt, dst) in zip(arr, jobs):
            for k, pl in enumerate(planes):
                d.src_dev[k] = pl.data_ptr()
                d.src_linesize[k] = pl.stride(0)
            d.bgra_dev = dst.data_ptr()
            d.bgra_linesize = dst.stride(0)
            d.src_width, d.src_height, d.src_format = int(sw), int(sh), int(fmt)
        if stream is None:
            stream = self._torch_stream()
        rc = self._lib.ntscsim_scale_to_bgra_device(self._h, arr, len(jobs), int(width), int(height),
                                                    C.c_void_p(stream))
        self._chk(rc, "ntscsim_scale_to_bgra_device")

    def frames_host_scaled(self, dst, src, source, width, height, first_fieldno=0, bob=True, chunk_frames=0, yuv=None):
        """src: numpy uint8 [N, frame_bytes] (frames in the layout `source`, a HostSource);
        dst: numpy uint8 [2N, height, width, 4], or with yuv = "420" / "422" uint8 [2N, frame_bytes]
        (Y|U|V planes packed at linesize width, chroma width/2)."""
        n = src.shape[0]
        # (frames may be strided: only each frame's own bytes must be contiguous)
        assert src.ndim == 2 and src.strides[1] == 1 and dst.flags.c_contiguous
        flags = DESC_BOB if bob else 0
        if yuv is None:
            assert dst.shape == (2 * n, height, width, 4)
            dls = dst.strides[1]
        else:
            flags |= _capi.HOST_YUV420P if yuv == "420" else _capi.HOST_YUV422P
            assert dst.ndim == 2 and dst.shape[0] == 2 * n
            dls = width
        u8p = C.POINTER(C.c_uint8)
        rc = self._lib.ntscsim_frames_host_scaled(self._h, C.byref(source), src.ctypes.data_as(u8p), src.strides[0], n,
                                                  dst.ctypes.data_as(u8p), dst.strides[0], dls,
                                                  int(width), int(height), int(first_fieldno),
                                                  flags, int(chunk_frames))
        self._chk(rc, "ntscsim_frames_host_scaled")

    # ---- batched, device-resident -----------------------------------------------------------
    def build_descs(self, src, dst, jobs, bob=False, interlaced=0, tff=0, rng_pos=None):
        """src, dst: torch uint8 CUDA tensors [N, H, W, 4].  jobs: iterable of
        (src_index, dst_index, field, fieldno).  rng_pos: None (sequential from the ctx
        position) or a list of absolute rand() stream positions."""
        assert src.is_cuda and dst.is_cuda and src.dim() == 4 and dst.dim() == 4
        assert src.stride(2) == 4 and src.stride(3) == 1 and dst.stride(2) == 4 and dst.stride(3) == 1
        jobs = list(jobs)
        arr = (FieldDesc * len(jobs))()
        sbase, dbase = src.data_ptr(), dst.data_ptr()
        sfs, dfs = src.stride(0), dst.stride(0)
        flags = (DESC_INTERLACED if interlaced else 0) | (DESC_TFF if tff else 0) | \
                (DESC_BOB if bob else 0)
        for i, (si, di, field, fieldno) in enumerate(jobs):
            d = arr[i]
            d.src_dev = sbase + si * sfs
            d.dst_dev = dbase + di * dfs
            d.src_linesize = src.stride(1)
            d.dst_linesize = dst.stride(1)
            d.field = field
            d.flags = flags
            d.fieldno = fieldno
            d.rng_pos = RNG_AUTO if rng_pos is None else int(rng_pos[i])
        return arr

    def run_descs(self, descs, width, height, stream=None):
        """Enqueue; does not synchronise."""
        if stream is None:
            stream = self._torch_stream()
        rc = self._lib.ntscsim_fields_device(self._h, descs, len(descs), int(width), int(height),
                                             C.c_void_p(stream))
        self._chk(rc, "ntscsim_fields_device")

    def prepare(self, descs, width, height):
        """Prepared batch (ntscsim_batch_create): returns a handle for run_prepared()."""
        h = C.c_void_p()
        rc = self._lib.ntscsim_batch_create(self._h, descs, len(descs), int(width), int(height),
                                            C.byref(h))
        self._chk(rc, "ntscsim_batch_create")
        return h

    def run_prepared(self, batch, stream=None):
        if stream is None:
            stream = self._torch_stream()
        self._chk(self._lib.ntscsim_batch_run(batch, C.c_void_p(stream)), "ntscsim_batch_run")

    def free_prepared(self, batch):
        self._lib.ntscsim_batch_destroy(batch)

    def fields(self, src, dst, jobs, **kw):
        descs = self.build_descs(src, dst, jobs, **kw)
        self.run_descs(descs, src.shape[2], src.shape[1])
        return descs

    # ---- 8-bit YUV422P sibling (ffmpeg_to_composite) ---------------------------------------
    def build_descs422(self, jobs):
        """jobs: list of dicts with keys dst (3 CUDA uint8 tensors [H, ls]), optional src (3
        tensors) + src_height, optional flt (3 tensors), field, fieldno, flags, rng_pos.
        Returns the ntscsim_field422_desc array (reusable: run_descs422)."""
        arr = (Field422Desc * len(jobs))()
        for d, j in zip(arr, jobs):
            for k in range(3):
                d.dst_dev[k] = j["dst"][k].data_ptr()
                d.dst_linesize[k] = j["dst"][k].stride(0)
                if j.get("src") is not None:
                    d.src_dev[k] = j["src"][k].data_ptr()
                    d.src_linesize[k] = j["src"][k].stride(0)
                if j.get("flt") is not None:
                    d.flt_dev[k] = j["flt"][k].data_ptr()
                    d.flt_linesize[k] = j["flt"][k].stride(0)
            d.src_height = j.get("src_height", 0)
            d.field = j["field"]
            d.flags = j.get("flags", 0)
            d.fieldno = j["fieldno"]
            d.rng_pos = RNG_AUTO if j.get("rng_pos") is None else int(j["rng_pos"])
        return arr

    def run_descs422(self, arr, width, height, stream=None):
        if stream is None:
            stream = self._torch_stream()
        rc = self._lib.ntscsim_fields422_device(self._h, arr, len(arr), int(width), int(height),
                                                C.c_void_p(stream))
        self._chk(rc, "ntscsim_fields422_device")

    def prepare422(self, arr, width, height):
        """ntscsim_batch422_create on a descriptor array (keep `arr` and its tensors alive)."""
        b = C.c_void_p()
        rc = self._lib.ntscsim_batch422_create(self._h, arr, len(arr), int(width), int(height), C.byref(b))
        self._chk(rc, "ntscsim_batch422_create")
        return b

    def run_prepared422(self, batch, stream=None):
        if stream is None:
            stream = self._torch_stream()
        self._chk(self._lib.ntscsim_batch422_run(batch, C.c_void_p(stream)), "ntscsim_batch422_run")

    def free_prepared422(self, batch):
        self._lib.ntscsim_batch422_destroy(batch)

    def fields422(self, jobs, width, height, stream=None):
        """build_descs422 + run_descs422."""
        self.run_descs422(self.build_descs422(jobs), width, height, stream)

    def output422(self, jobs, width, height, stream=None):
        """jobs: list of dicts {frame: 3 CUDA uint8 tensors, bob: 3 tensors, field, mode}."""
        arr = (Out422Desc * len(jobs))()
        for d, j in zip(arr, jobs):
            for k in range(3):
                d.frame_dev[k] = j["frame"][k].data_ptr()
                d.frame_linesize[k] = j["frame"][k].stride(0)
                d.bob_dev[k] = j["bob"][k].data_ptr()
                d.bob_linesize[k] = j["bob"][k].stride(0)
            d.field = j["field"]
            d.mode = j["mode"]
        if stream is None:
            stream = self._torch_stream()
        rc = self._lib.ntscsim_output422_device(self._h, arr, len(jobs), int(width), int(height),
                                                C.c_void_p(stream))
        self._chk(rc, "ntscsim_output422_device")

    def _torch_stream(self):
        """Handle of torch's current stream.  The legacy default stream has handle 0, which the C
        ABI reads as "the context's own (non-blocking) stream": nothing would order the kernels
        behind torch work still queued on the default stream, so that work is waited for here."""
        import torch
        st = torch.cuda.current_stream(self.device)
        if st.cuda_stream == 0:
            st.synchronize()
        return st.cuda_stream

    def sync(self):
        self._chk(self._lib.ntscsim_sync(self._h), "ntscsim_sync")

    def debug_field_stats(self):
        """[calls of ntscsim_field, sources read in place, destinations written in place, early setup kernels used]"""
        a = (C.c_uint64 * 4)()
        self._lib.ntscsim_debug_field_stats(self._h, a)
        return list(a)

    def set_launch_form(self, latency=True):
        """Device-resident launches of up to 64 fields as wavefront roles (ntscsim_set_launch_form: NTSCSIM_FORM_LATENCY)."""
        self._chk(self._lib.ntscsim_set_launch_form(self._h, 1 if latency else 0), "ntscsim_set_launch_form")

    def set_profiling(self, on=True):
        self._lib.ntscsim_set_profiling(self._h, 1 if on else 0)

    def timings_ms(self):
        """Summed hipEvent timings (ms) of the calls made since the last query."""
        out = (C.c_float * 4)()
        n = C.c_int(0)
        self._chk(self._lib.ntscsim_get_timings_ms(self._h, out, C.byref(n)),
                  "ntscsim_get_timings_ms")
        return {"setup": out[0], "encode": out[1], "decode": out[2], "total": out[3],
                "calls": n.value}

    def debug_force_generic(self, on=True):
        self._lib.ntscsim_debug_force_generic(self._h, 1 if on else 0)

    def debug_no_fast_decode(self, on=True):
        """on: True/1 = template PRESET kernels, 2 = two-launch (VCR half / TV half) VHS decoder."""
        self._lib.ntscsim_debug_no_fast_decode(self._h, int(on))

    def last_kernels(self):
        """Names of the kernel forms the last fields / fields422 / batch call enqueued, in order."""
        buf = C.create_string_buffer(1024)
        n = self._lib.ntscsim_debug_last_kernels(self._h, buf, len(buf))
        if n < 0:
            raise RuntimeError("ntscsim_debug_last_kernels: %d" % n)
        return [k for k in buf.value.decode().split(";") if k]

    def debug_set_warmup(self, luma_draws, chroma_draws):
        self._lib.ntscsim_debug_set_warmup(self._h, int(luma_draws), int(chroma_draws))

    def debug_composite(self, n_fields, width, height):
        import numpy as np
        lslot = (height + 1) // 2
        a = np.zeros((n_fields, lslot, width), dtype=np.int32)
        rc = self._lib.ntscsim_debug_read_composite(
            self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), a.size)
        self._chk(rc, "ntscsim_debug_read_composite")
        return a


class Pool:
    """ntscsim_pool_*: one context per GPU (an ordinal may repeat), a run of host frames dealt block-cyclically
    over them -- byte-identical to one FieldSimulator.frames_host() over the whole run."""

    def __init__(self, flags=(), devices=(0,), params=None, block_frames=None):
        self.params = params if params is not None else make_params(flags)
        self._lib = lib()
        h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = self._lib.ntscsim_pool_create(C.byref(self.params), devs, len(devices), C.byref(h))
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_pool_create(%r)" % (list(devices),))
        self._h = h
        if block_frames is not None:
            rc = self._lib.ntscsim_pool_set_block(self._h, int(block_frames))
            if rc != _capi.OK:
                raise NtscsimError(rc, "ntscsim_pool_set_block")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ntscsim_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return int(self._lib.ntscsim_pool_size(self._h))

    @property
    def rng_pos(self):
        return int(self._lib.ntscsim_pool_get_rng_pos(self._h))

    @rng_pos.setter
    def rng_pos(self, pos):
        self._lib.ntscsim_pool_set_rng_pos(self._h, int(pos))

    def frames_host(self, dst, src, first_fieldno=0, bob=True, chunk_frames=0):
        """src numpy uint8 [N, H, W, 4], dst [2N, H, W, 4] (as FieldSimulator.frames_host)."""
        n, h, w = src.shape[:3]
        assert src.flags.c_contiguous and dst.flags.c_contiguous and dst.shape == (2 * n, h, w, 4)
        u8p = C.POINTER(C.c_uint8)
        rc = self._lib.ntscsim_pool_frames_host(self._h, src.ctypes.data_as(u8p), src.strides[0], src.strides[1], n,
                                                dst.ctypes.data_as(u8p), dst.strides[0], dst.strides[1], w, h,
                                                int(first_fieldno), DESC_BOB if bob else 0, int(chunk_frames))
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_pool_frames_host: %s" % self._lib.ntscsim_pool_last_error(self._h).decode())


class Raw28Decoder:
    """The raw-composite decoder (ffmpeg_raw28ntsc): one decode() = one run of the tool on a capture."""

    def __init__(self, flags=(), device=0, opts=None):
        self._lib = lib()
        self.opts = opts if opts is not None else _capi.make_raw28_opts(flags)
        w, h, sl = C.c_int(), C.c_int(), C.c_int()
        rc = self._lib.ntscsim_raw28_geometry(C.byref(self.opts), C.byref(w), C.byref(h), C.byref(sl))
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_raw28_geometry")
        self.width, self.height, self.scanline = w.value, h.value, sl.value
        hnd = C.c_void_p()
        rc = self._lib.ntscsim_raw28_create(C.byref(self.opts), int(device), C.byref(hnd))
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_raw28_create")
        self._h = hnd

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ntscsim_raw28_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, capture, frames, max_fields=None):
        """capture: numpy uint8 (host) or CUDA uint8 tensor; frames: CUDA uint8 tensor
        [F, height, >= 4*width].  Returns the number of fields decoded."""
        n = C.c_int()
        mf = int(frames.shape[0] if max_fields is None else max_fields)
        if hasattr(capture, "data_ptr"):
            fn, ptr, ns = self._lib.ntscsim_raw28_decode_device, capture.data_ptr(), capture.numel()
        else:
            fn, ptr, ns = self._lib.ntscsim_raw28_decode, capture.ctypes.data, capture.size
        rc = fn(self._h, C.c_void_p(ptr), ns, C.c_void_p(frames.data_ptr()), frames.stride(0), frames.stride(1),
                mf, C.byref(n))
        if rc != _capi.OK:
            raise NtscsimError(rc, self._lib.ntscsim_raw28_last_error(self._h).decode())
        return n.value

    def stream_reset(self):
        rc = self._lib.ntscsim_raw28_stream_reset(self._h)
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_raw28_stream_reset")

    def stream_push(self, samples, frames, final=False, max_fields=None):
        """samples: numpy uint8 (host), a CUDA uint8 tensor, or None / empty; frames: CUDA uint8 tensor
        [F, height, >= 4*width].  Returns the number of fields this push wrote to frames[0:]."""
        n = C.c_int()
        mf = int(frames.shape[0] if max_fields is None else max_fields)
        if samples is None:
            ptr, ns, dev = None, 0, 0
        elif hasattr(samples, "data_ptr"):
            import torch
            if samples.dtype != torch.uint8 or not samples.is_contiguous():
                raise TypeError("stream_push: a CUDA tensor of samples must be contiguous uint8 (one byte per sample)")
            ptr, ns, dev = samples.data_ptr(), samples.numel(), 1
        else:
            import numpy as np
            keep = np.ascontiguousarray(samples)      # (a strided view would push the bytes in between)
            if keep.dtype != np.uint8:
                raise TypeError("stream_push: samples must be uint8 (one byte per sample), got %s" % keep.dtype)
            ptr, ns, dev = keep.ctypes.data, keep.size, 0
        rc = self._lib.ntscsim_raw28_stream_push(self._h, C.c_void_p(ptr), ns, dev, 1 if final else 0,
                                                 C.c_void_p(frames.data_ptr()), frames.stride(0), frames.stride(1),
                                                 mf, C.byref(n))
        if rc != _capi.OK:
            raise NtscsimError(rc, self._lib.ntscsim_raw28_last_error(self._h).decode())
        return n.value

    def levels(self):
        b, w, p = C.c_double(), C.c_double(), C.c_uint64()
        self._lib.ntscsim_raw28_get_levels(self._h, C.byref(b), C.byref(w), C.byref(p))
        return b.value, w.value, p.value

    def set_speculation(self, warm_lines=-1, chunk_samples=0):
        self._lib.ntscsim_raw28_debug_set_speculation(self._h, int(warm_lines), int(chunk_samples))

    def stats(self):
        a = (C.c_int64 * 16)()
        self._lib.ntscsim_raw28_debug_stats(self._h, a)
        return dict(zip(("front_rounds", "chunks_repaired", "tail_rounds", "sync_runs", "scanlines", "cal_pulses",
                         "us_front", "us_runs", "us_walk", "us_queue", "us_wait", "us_redo",
                         "pulses_past_stream", "zero_records", "compactions", "max_samples_held"), list(a)))

    def read_front(self, n):
        import numpy as np
        out = np.empty(n, np.uint8)
        rc = self._lib.ntscsim_raw28_debug_read_front(self._h, C.c_void_p(out.ctypes.data), n)
        if rc != _capi.OK:
            raise NtscsimError(rc, "ntscsim_raw28_debug_read_front")
        return out


_host_blocks = {}


def host_alloc_array(shape, align_offset=0):
    """A uint8 array of `shape` in pinned memory from ntscsim_host_alloc(), starting `align_offset` bytes into the
    block (tests: frames that do not start on a page boundary).  Free with host_free_array()."""
    import numpy as np
    n = int(np.prod(shape))
    p = lib().ntscsim_host_alloc(n + int(align_offset))
    if not p:
        raise MemoryError("ntscsim_host_alloc(%d)" % n)
    buf = (C.c_uint8 * (n + int(align_offset))).from_address(p)
    a = np.frombuffer(buf, dtype=np.uint8)[int(align_offset):].reshape(shape)
    _host_blocks[a.ctypes.data] = p
    return a


def host_free_array(a):
    p = _host_blocks.pop(a.ctypes.data, None)
    if p:
        lib().ntscsim_host_free(p)
