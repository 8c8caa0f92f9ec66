"""SURVEY 8(f) row f2, input side: source scaling / pixel-format conversion to BGRA W x H on the GPU
(what the tool does with libswscale at ffmpeg_ntsc.cpp:573-583 / :603).  libswscale is third-party
and absent from the reference tree: PARITY UNPINNED.  What is tested: the product == the documented
definition (csrc/ntsc_scale.hip, restated in numpy in tests/_libs.py) bit for bit, the properties any
sane conversion has, and the wiring into the host-frame field loop."""
import ctypes as C

import numpy as np
import pytest

import _libs as L
import ntscsim
from ntscsim import _capi


def test_definition_properties():
    """CPU: identity at equal size, constants stay constant, BT.601 anchors, monotone ramps."""
    rng = np.random.RandomState(1)
    a = rng.randint(0, 256, size=(9, 14, 4), dtype=np.uint8)
    assert np.array_equal(L.oracle_scale_to_bgra([a], 0, 14, 9), a)
    c = np.full((5, 7, 4), 93, np.uint8)
    assert (L.oracle_scale_to_bgra([c], 0, 31, 17) == 93).all()
    for (yy, want) in ((16, 0), (235, 255), (126, 128)):
        o = L.oracle_scale_to_bgra([np.full((8, 8), yy, np.uint8), np.full((4, 4), 128, np.uint8),
                                    np.full((4, 4), 128, np.uint8)], 1, 16, 16)
        assert (o[..., :3] == want).all() and (o[..., 3] == 255).all()
    ramp = np.tile(np.arange(0, 256, 4, dtype=np.uint8)[None, :, None], (3, 1, 4))
    up = L.oracle_scale_to_bgra([ramp], 0, 200, 3)[0, :, 0].astype(int)
    assert (np.diff(up) >= 0).all() and up[0] == 0 and up[-1] == 252
    # red in YUV (Y 81, U 90, V 240) comes back as red
    o = L.oracle_scale_to_bgra([np.full((4, 4), 81, np.uint8), np.full((2, 2), 90, np.uint8),
                                np.full((2, 2), 240, np.uint8)], 1, 4, 4)
    assert abs(int(o[0, 0, 2]) - 255) <= 2 and o[0, 0, 1] <= 2 and o[0, 0, 0] <= 2


def test_scale_exports_and_struct_sizes():
    lib = L.product()
    assert hasattr(lib, "ntscsim_scale_to_bgra_device") and hasattr(lib, "ntscsim_frames_host_scaled")
    assert C.sizeof(_capi.ScaleDesc) == 64 and C.sizeof(_capi.HostSource) == 56


def _planes(rng, fmt, sw, sh, pad):
    if fmt == _capi.SRC_BGRA:
        return [rng.randint(0, 256, size=(sh, sw * 4 + pad), dtype=np.uint8)], [sw * 4]
    cw, ch = (sw + 1) // 2, ((sh + 1) // 2 if fmt == _capi.SRC_YUV420P else sh)
    return ([rng.randint(0, 256, size=(sh, sw + pad), dtype=np.uint8),
             rng.randint(0, 256, size=(ch, cw + pad), dtype=np.uint8),
             rng.randint(0, 256, size=(ch, cw + pad), dtype=np.uint8)], [sw, cw, cw])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [_capi.SRC_BGRA, _capi.SRC_YUV420P, _capi.SRC_YUV422P])
@pytest.mark.parametrize("sw,sh,w,h,pad", [(64, 48, 64, 48, 0), (1920, 1080, 720, 486, 0), (352, 240, 720, 480, 0),
                                           (33, 17, 96, 32, 3), (720, 576, 720, 480, 0), (16, 2, 50, 7, 0)])
def test_hip_scale_equals_definition(fmt, sw, sh, w, h, pad):
    import torch
    rng = np.random.RandomState(sw * 7 + sh + fmt)
    planes, widths = _planes(rng, fmt, sw, sh, pad)
    if fmt == _capi.SRC_BGRA:
        want = L.oracle_scale_to_bgra([planes[0][:, :sw * 4].reshape(sh, sw, 4)], 0, w, h)
    else:
        want = L.oracle_scale_to_bgra([p[:, :n] for p, n in zip(planes, widths)], fmt, w, h)
    sim = ntscsim.FieldSimulator([])
    dev = [torch.from_numpy(p).cuda() for p in planes]
    dst = torch.full((h, w * 4 + 8), 7, dtype=torch.uint8, device="cuda")
    sim.scale_to_bgra([(dev, sw, sh, fmt, dst)], w, h)
    sim.sync()
    got = dst.cpu().numpy()
    assert np.array_equal(got[:, :w * 4].reshape(h, w, 4), want)
    assert (got[:, w * 4:] == 7).all()
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.scale_to_bgra([(dev, sw, sh, 9, dst)], w, h)
    assert e.value.code == _capi.E_ARG
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,sw,sh", [(_capi.SRC_YUV420P, 352, 240), (_capi.SRC_BGRA, 100, 50), (_capi.SRC_YUV422P, 96, 32)])
def test_frames_host_scaled_equals_scale_then_field_loop(fmt, sw, sh):
    """The scaled host loop == scale every frame by the definition, then the plain host loop (itself
    oracle-checked in test_gpu_parity.py)."""
    w, h, n = 96, 32, 5
    rng = np.random.RandomState(fmt + sw)
    if fmt == _capi.SRC_BGRA:
        sizes, ls = [sw * 4 * sh], [sw * 4, 0, 0]
    else:
        cw, ch = (sw + 1) // 2, ((sh + 1) // 2 if fmt == _capi.SRC_YUV420P else sh)
        sizes, ls = [sw * sh, cw * ch, cw * ch], [sw, cw, cw]
    fb = (sum(sizes) + 15) // 16 * 16
    src = rng.randint(0, 256, size=(n, fb), dtype=np.uint8)
    hs = _capi.HostSource()
    hs.format, hs.width, hs.height, hs.frame_bytes = fmt, sw, sh, fb
    off = 0
    for k, sz in enumerate(sizes):
        hs.linesize[k], hs.plane_offset[k] = ls[k], off
        off += sz
    bgra = []
    for j in range(n):
        if fmt == _capi.SRC_BGRA:
            pl = [src[j, :sizes[0]].reshape(sh, sw, 4)]
        else:
            o1, o2 = sizes[0], sizes[0] + sizes[1]
            pl = [src[j, :o1].reshape(sh, sw), src[j, o1:o2].reshape(-1, ls[1]), src[j, o2:o2 + sizes[2]].reshape(-1, ls[2])]
        bgra.append(L.oracle_scale_to_bgra(pl, fmt, w, h))
    bgra = np.stack(bgra)
    p = L.make_params(["-vhs"])
    a = np.zeros((2 * n, h, w, 4), np.uint8)
    b = np.zeros_like(a)
    sim = ntscsim.FieldSimulator(params=p)
    sim.frames_host_scaled(a, src, hs, w, h, chunk_frames=2)
    sim.rng_pos = 0
    sim.frames_host(b, bgra, chunk_frames=2)
    sim.close()
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_frames_host_scaled_yuv_in_yuv_out_and_tight_last_frame():
    """YUV420P frames in -> YUV420P bob frames out (1.5 bytes per pixel each way over the link) == the
    scaled source through the plain host loop with the same output format.  The source frames are
    4998 bytes each at a stride of 5008 (the device copy's 16-byte rounding) in a buffer that ENDS with
    the last frame's last byte: the whole-chunk upload must not read the rounding bytes behind it."""
    sw, sh, w, h, n = 98, 34, 96, 32, 5
    cw, ch = (sw + 1) // 2, (sh + 1) // 2
    sizes = [sw * sh, cw * ch, cw * ch]
    fb = sum(sizes)
    stride = (fb + 15) // 16 * 16
    assert fb % 16 != 0
    rng = np.random.RandomState(11)
    flat = rng.randint(0, 256, size=stride * (n - 1) + fb, dtype=np.uint8)
    src = np.lib.stride_tricks.as_strided(flat, shape=(n, fb), strides=(stride, 1))
    hs = _capi.HostSource()
    hs.format, hs.width, hs.height, hs.frame_bytes = _capi.SRC_YUV420P, sw, sh, fb
    off = 0
    for k, sz in enumerate(sizes):
        hs.linesize[k], hs.plane_offset[k] = [sw, cw, cw][k], off
        off += sz
    bgra = np.stack([L.oracle_scale_to_bgra([src[j, :sizes[0]].reshape(sh, sw),
                                             src[j, sizes[0]:sizes[0] + sizes[1]].reshape(ch, cw),
                                             src[j, sizes[0] + sizes[1]:].reshape(ch, cw)], _capi.SRC_YUV420P, w, h)
                     for j in range(n)])
    ob = w * h + 2 * (w // 2) * ((h + 1) // 2)
    a = np.zeros((2 * n, ob), np.uint8)
    b = np.zeros_like(a)
    sim = ntscsim.FieldSimulator(params=L.make_params(["-vhs"]))
    sim.frames_host_scaled(a, src, hs, w, h, chunk_frames=2, yuv="420")
    sim.rng_pos = 0
    sim.frames_host(b, bgra, chunk_frames=2, yuv="420")
    sim.close()
    assert a.any() and np.array_equal(a, b)
