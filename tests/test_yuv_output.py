"""Encoder-side colour conversion (output side): BGRA -> YUV420P / YUV422P.
PARITY UNPINNED against libswscale (absent from the reference tree): the oracle states the
product's own BT.601 definition; the HIP kernel must match it bit for bit, and the definition
itself is checked through colour-science properties."""
import numpy as np
import pytest

import _libs as L
import ntscsim
from ntscsim import _capi


def test_definition_properties():
    # greys: no chroma, luma on the 16..235 scale
    g = np.arange(256, dtype=np.uint8)
    img = np.zeros((2, 256, 4), np.uint8)
    for k in range(3):
        img[:, :, k] = g[None, :]
    for is420 in (0, 1):
        y, u, v = L.oracle_bgra_to_yuv(img, is420)
        assert (u == 128).all() and (v == 128).all()
        want = np.floor(16.5 + g.astype(np.float64) * 219 / 255 + 1e-9)
        assert np.abs(y[0].astype(int) - want).max() <= 1
        assert y.min() == 16 and y.max() == 235
    # primaries (B, G, R byte order): BT.601 limited-range values
    px = {"white": (255, 255, 255), "red": (0, 0, 255), "green": (0, 255, 0), "blue": (255, 0, 0),
          "black": (0, 0, 0)}
    want = {"white": (235, 128, 128), "red": (81, 90, 240), "green": (145, 54, 34),
            "blue": (41, 240, 110), "black": (16, 128, 128)}
    for name, bgr in px.items():
        img = np.zeros((2, 2, 4), np.uint8)
        img[:, :, 0], img[:, :, 1], img[:, :, 2] = bgr
        y, u, v = L.oracle_bgra_to_yuv(img, 1)
        got = (int(y[0, 0]), int(u[0, 0]), int(v[0, 0]))
        assert all(abs(a - b) <= 1 for a, b in zip(got, want[name])), (name, got)
    # alpha is ignored
    rng = np.random.RandomState(1)
    a = rng.randint(0, 256, size=(6, 8, 4), dtype=np.uint8)
    b = a.copy(); b[:, :, 3] = 255 - b[:, :, 3]
    for is420 in (0, 1):
        assert all(np.array_equal(p, q) for p, q in zip(L.oracle_bgra_to_yuv(a, is420), L.oracle_bgra_to_yuv(b, is420)))
    # 4:2:0 of a frame whose row pairs are equal == 4:2:2 of it, subsampled; odd height: last row twice
    a = rng.randint(0, 256, size=(5, 8, 4), dtype=np.uint8)
    a[1] = a[0]; a[3] = a[2]
    y0, u0, v0 = L.oracle_bgra_to_yuv(a, 1)
    y2, u2, v2 = L.oracle_bgra_to_yuv(a, 0)
    assert np.array_equal(y0, y2) and np.array_equal(u0, u2[::2]) and np.array_equal(v0, v2[::2])


def test_capi_constants_match_header():
    import os, re
    hdr = open(os.path.join(L.ROOT, "include", "ntscsim.h")).read()
    assert int(re.search(r"#define NTSCSIM_HOST_YUV420P (0x[0-9a-fA-F]+)u", hdr).group(1), 16) == _capi.HOST_YUV420P
    assert int(re.search(r"#define NTSCSIM_HOST_YUV422P (0x[0-9a-fA-F]+)u", hdr).group(1), 16) == _capi.HOST_YUV422P
    assert int(re.search(r"#define NTSCSIM_PIX_YUV420P (\d+)", hdr).group(1)) == _capi.PIX_YUV420P
    assert int(re.search(r"#define NTSCSIM_PIX_YUV422P (\d+)", hdr).group(1)) == _capi.PIX_YUV422P


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,pad", [(720, 486, 0), (720, 480, 0), (64, 33, 0), (66, 7, 4), (18, 2, 0),
                                     (1920, 1080, 0), (40, 5, 12)])
@pytest.mark.parametrize("is420", [1, 0])
def test_hip_bgra_to_yuv_equals_definition(w, h, pad, is420):
    import torch
    rng = np.random.RandomState(w + h)
    frames = [rng.randint(0, 256, size=(h, w, 4), dtype=np.uint8), L.bars(w, h, 3)]
    sim = ntscsim.FieldSimulator(["-vhs"])
    jobs, want = [], []
    ch = (h + 1) // 2 if is420 else h
    for f in frames:
        # pad columns make linesizes that are not multiples of 16 (the scalar path)
        bg = torch.zeros((h, w * 4 + pad), dtype=torch.uint8, device="cuda")
        bg[:, :w * 4] = torch.from_numpy(f.reshape(h, w * 4)).cuda()
        planes = [torch.full((h, w + pad), 7, dtype=torch.uint8, device="cuda"),
                  torch.full((ch, w // 2 + pad), 7, dtype=torch.uint8, device="cuda"),
                  torch.full((ch, w // 2 + pad), 7, dtype=torch.uint8, device="cuda")]
        jobs.append((bg, planes))
        want.append(L.oracle_bgra_to_yuv(f, is420))
    sim.bgra_to_yuv(jobs, w, h, _capi.PIX_YUV420P if is420 else _capi.PIX_YUV422P)
    sim.sync()
    for (bg, planes), wnt in zip(jobs, want):
        for k in range(3):
            got = planes[k].cpu().numpy()
            ww = w if k == 0 else w // 2
            assert np.array_equal(got[:, :ww], wnt[k]), k
            assert (got[:, ww:] == 7).all()          # nothing written past the row
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.bgra_to_yuv(jobs, w + 1, h, 0)
    assert e.value.code == _capi.E_SIZE
    with pytest.raises(ntscsim.NtscsimError) as e:
        sim.bgra_to_yuv(jobs, w, h, 5)
    assert e.value.code == _capi.E_ARG
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("yuv,w,h", [("420", 96, 32), ("422", 96, 32), ("420", 72, 21), ("420", 720, 486)])
def test_frames_host_yuv_output(yuv, w, h):
    """ntscsim_frames_host with NTSCSIM_HOST_YUV*: the planar frames delivered to the host are the
    conversion of exactly the BGRA bob frames the plain call delivers (themselves oracle-checked)."""
    n = 5
    src = np.stack([L.bars(w, h, j) if j % 2 else L.noise_frame(w, h, j + 1) for j in range(n)])
    sim = ntscsim.FieldSimulator(["-vhs"])
    bgra = np.zeros((2 * n, h, w, 4), np.uint8)
    sim.rng_pos = 0
    sim.frames_host(bgra, src, first_fieldno=0, chunk_frames=2)
    ch = (h + 1) // 2 if yuv == "420" else h
    fb = w * h + 2 * (w // 2) * ch
    out = np.zeros((2 * n, fb), np.uint8)
    sim.rng_pos = 0
    sim.frames_host(out, src, first_fieldno=0, chunk_frames=2, yuv=yuv)
    for k in range(2 * n):
        y, u, v = L.oracle_bgra_to_yuv(bgra[k], 1 if yuv == "420" else 0)
        assert np.array_equal(out[k, :w * h].reshape(h, w), y), k
        assert np.array_equal(out[k, w * h:w * h + (w // 2) * ch].reshape(ch, w // 2), u), k
        assert np.array_equal(out[k, w * h + (w // 2) * ch:].reshape(ch, w // 2), v), k
    sim.close()
