"""ntsc_cli: the ffmpeg_ntsc-compatible host (raw BGRA I/O).  CPU part: flag handling mirrors the
reference (exit code 1 on unknown switches / missing -i/-o / -h).  GPU part: the byte stream it
writes equals the reference's field loop (composite_layer + bob + frame-delay ring) replayed with
the oracle."""
import os
import subprocess

import numpy as np
import pytest

import _libs as L

CLI = os.path.join(L.PKG, "ntsc_cli")


def run(args, **kw):
    return subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_cli_exists_and_help():
    assert os.path.exists(CLI)
    r = run(["-h"])
    assert r.returncode == 1 and b"-vhs" in r.stderr and b"-comp-phase" in r.stderr


@pytest.mark.parametrize("args,msg", [
    (["-bogus"], b"Unknown switch 'bogus'"),
    (["stray"], b"Unhandled arg 'stray'"),
    (["-i", "bars:2"], b"No output file specified"),
    (["-o", "null:"], b"No input files specified"),
    (["-i", "bars:2", "-o", "null:", "-comp-phase", "45"], b"Invalid phase"),
    (["-i", "bars:2", "-o", "null:", "-d", "0"], b"Invalid delay"),
    (["-i", "bars:2", "-o", "null:", "-tvstd", "secam"], b"Unknown tv std 'secam'"),
    (["-i", "bars:2", "-o", "null:", "-ss", "1"], b"Unknown switch 'ss'"),   # help()-only switch
])
def test_cli_rejects_like_the_reference(args, msg):
    r = run(args)
    assert r.returncode == 1 and msg in r.stderr


def oracle_field_loop(p, frames, delay):
    """main() :2140-2283 for one input: ring of `delay` zeroed frames, composite_layer, bob."""
    h, w = frames[0].shape[:2]
    ring = [np.zeros((h, w, 4), np.uint8) for _ in range(delay)]
    o = L.OracleStream(p)
    out = []
    idx = 0
    for cur in range(2 * len(frames)):
        field = (cur & 1) ^ 1
        o.field(ring[idx], frames[cur // 2], field, cur)
        L.oracle().ntsc_oracle_bob(L._ptr(ring[idx]), w * 4, w, h, field)
        out.append(ring[idx].copy())
        idx = (idx + 1) % delay
    return np.stack(out)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,w,h,nfr,delay,batch", [
    (["-vhs"], 720, 480, 3, 1, 4),
    (["-vhs", "-width", "96"], 96, 34, 5, 1, 4),
    (["-vhs", "-width", "96"], 96, 33, 5, 3, 2),       # odd height, ring of 3
    (["-width", "64", "-d", "2"], 64, 20, 4, 2, 256),
])
def test_cli_stream_equals_reference_field_loop(tmp_path, flags, w, h, nfr, delay, batch):
    frames = [L.noise_frame(w, h, 0x1234567 + k) for k in range(nfr)]   # == noise:N source
    inp = tmp_path / "in.bgra"
    outp = tmp_path / "out.bgra"
    inp.write_bytes(b"".join(f.tobytes() for f in frames))
    fl = list(flags)
    if "-d" not in fl:
        fl += ["-d", str(delay)]
    args = fl + ["-i", str(inp), "-o", str(outp), "--height", str(h), "--batch", str(batch)]
    r = run(args)
    assert r.returncode == 0, r.stderr.decode()
    got = np.frombuffer(outp.read_bytes(), np.uint8).reshape(2 * nfr, h, w, 4)
    p = L.make_params([a for a in flags], output_height=h)
    exp = oracle_field_loop(p, frames, delay)
    assert np.array_equal(got, exp)
    # synthetic source spelled noise:N gives the same stream
    r2 = run(fl + ["-i", "noise:%d" % nfr, "-o", str(outp), "--height", str(h), "--batch", str(batch)])
    assert r2.returncode == 0
    assert np.array_equal(np.frombuffer(outp.read_bytes(), np.uint8).reshape(2 * nfr, h, w, 4), exp)


@pytest.mark.gpu
def test_cli_layered_inputs_advance_the_rand_stream(tmp_path):
    """Two -i layers: the last one wins but the first still consumes rand() draws (:2203-2230)."""
    w, h, nfr = 96, 32, 2
    outp = tmp_path / "o.bgra"
    r = run(["-vhs", "-width", str(w), "--height", str(h), "-i", "bars:%d" % nfr, "-i",
             "noise:%d" % nfr, "-o", str(outp)])
    assert r.returncode == 0, r.stderr.decode()
    got = np.frombuffer(outp.read_bytes(), np.uint8).reshape(2 * nfr, h, w, 4)
    p = L.make_params(["-vhs"], output_height=h, output_width=w)
    o = L.OracleStream(p)
    ring = np.zeros((h, w, 4), np.uint8)
    for cur in range(2 * nfr):
        field = (cur & 1) ^ 1
        o.field(ring, L.bars(w, h, cur // 2), field, cur)
        o.field(ring, L.noise_frame(w, h, 0x1234567 + cur // 2), field, cur)
        L.oracle().ntsc_oracle_bob(L._ptr(ring), w * 4, w, h, field)
        assert np.array_equal(got[cur], ring), cur


@pytest.mark.gpu
def test_cli_lower_layer_ending_early_does_not_end_the_run(tmp_path):
    """The reference keeps compositing an ended layer's last frame (:2218-2226): it goes on drawing
    from rand() until every input has ended."""
    w, h = 96, 32
    outp = tmp_path / "o.bgra"
    r = run(["-vhs", "-width", str(w), "--height", str(h), "-i", "bars:1", "-i", "noise:3", "-o", str(outp)])
    assert r.returncode == 0, r.stderr.decode()
    got = np.frombuffer(outp.read_bytes(), np.uint8).reshape(6, h, w, 4)
    p = L.make_params(["-vhs"], output_height=h, output_width=w)
    o = L.OracleStream(p)
    ring = np.zeros((h, w, 4), np.uint8)
    for cur in range(6):
        field = (cur & 1) ^ 1
        o.field(ring, L.bars(w, h, 0), field, cur)                       # layer 1: its last frame
        o.field(ring, L.noise_frame(w, h, 0x1234567 + cur // 2), field, cur)
        L.oracle().ntsc_oracle_bob(L._ptr(ring), w * 4, w, h, field)
        assert np.array_equal(got[cur], ring), cur


@pytest.mark.gpu
def test_cli_top_layer_ending_early_keeps_its_last_frame(tmp_path):
    """The loop runs until EVERY input has ended (ffmpeg_ntsc.cpp:2149-2153, :2283); an ended input keeps
    compositing the last frame it delivered (:2192-2197) -- also the top one, while a longer lower layer
    is still running: 3 frames of bars under 1 frame of noise = 6 fields, all of them the noise frame,
    with the rand() stream advanced by both layers every field."""
    w, h = 96, 32
    outp = tmp_path / "o.bgra"
    r = run(["-vhs", "-width", str(w), "--height", str(h), "-i", "bars:3", "-i", "noise:1", "-o", str(outp)])
    assert r.returncode == 0, r.stderr.decode()
    got = np.frombuffer(outp.read_bytes(), np.uint8).reshape(6, h, w, 4)
    p = L.make_params(["-vhs"], output_height=h, output_width=w)
    o = L.OracleStream(p)
    ring = np.zeros((h, w, 4), np.uint8)
    for cur in range(6):
        field = (cur & 1) ^ 1
        o.field(ring, L.bars(w, h, cur // 2), field, cur)
        o.field(ring, L.noise_frame(w, h, 0x1234567), field, cur)        # top layer: its only frame, again and again
        L.oracle().ntsc_oracle_bob(L._ptr(ring), w * 4, w, h, field)
        assert np.array_equal(got[cur], ring), cur


@pytest.mark.gpu
def test_cli_truncated_final_frame_is_reported(tmp_path):
    w, h = 96, 32
    inp, outp = tmp_path / "in.bgra", tmp_path / "o.bgra"
    inp.write_bytes(L.noise_frame(w, h, 5).tobytes() + b"\x00" * 100)
    r = run(["-width", str(w), "--height", str(h), "-i", str(inp), "-o", str(outp)])
    assert r.returncode == 0 and b"truncated final frame" in r.stderr
    assert len(outp.read_bytes()) == 2 * w * h * 4
