"""The raw-composite decoder (ffmpeg_raw28ntsc.cpp): oracle == reference
extract (CPU, build container), oracle == committed hashes (CPU, everywhere), HIP == oracle (GPU)."""
import hashlib
import json
import os

import numpy as np
import pytest

import _libs as L
from ntscsim import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "raw28_hashes.json")

# (name, oracle options, product switches)
CASES = [
    ("default", {}, []),
    ("marksig", {"mark_sync": 1}, ["-marksig"]),
    ("nosig", {"disable_sync": 1}, ["-nosig"]),
    ("nowequ", {"disable_wp_equ": 1}, ["-nowequ"]),
    ("showsc", {"show_subcarrier": 1}, ["-showsc"]),
    ("nosc", {"disable_subcarrier": 1}, ["-nosc"]),
    ("noequ", {"disable_equalization": 1}, ["-noequ"]),
    ("nosc_showsc_marksig", {"disable_subcarrier": 1, "show_subcarrier": 1, "mark_sync": 1}, ["-nosc", "-showsc", "-marksig"]),
    # another geometry: the same captures read as if sampled at 40 MHz (2,542 samples per scanline, other
    # pulse-length thresholds, frame 2542 x 262) -- not a meaningful picture, but every code path
    ("rate40", {"sample_rate": 40e6}, ["-s", "40mhz"]),
    ("rate40_nosig", {"sample_rate": 40e6, "disable_sync": 1}, ["-s", "40mhz", "-nosig"]),
]
# (fields, seed, noise, samples cut from the start)
CAPTURES = {"clean": (5, 1, 0, 0), "noisy_cut": (6, 7, 4, 123457), "long": (13, 3, 2, 600001)}


def _odd_capture(kind):
    """Captures that drive the tool into the corners of its sample buffer: no vertical sync in the
    last scanlines and an equalising-length pulse right at the end of the data, so that the
    calibration sums (:661-676) run past the buffered stream -- into never-filled records (short
    capture) or into stale records of an earlier window (capture longer than the buffer); random
    bytes; a constant."""
    if kind in ("tail_short", "tail_long"):
        base = L.raw28_capture(3 if kind == "tail_short" else 10, 21, 2, 0)
        line = base[(9 + 100) * 1820:(9 + 101) * 1820]
        tail = np.concatenate([np.tile(line, 330), np.full(73, 18, np.uint8), np.full(90, 62, np.uint8)])
        return np.ascontiguousarray(np.concatenate([base, tail]))
    if kind == "random":
        return np.random.RandomState(5).randint(0, 256, 1820 * 900, dtype=np.uint8)
    if kind == "random_low":
        return (np.random.RandomState(6).randint(0, 64, 1820 * 700) + 10).astype(np.uint8)
    if kind == "constant":
        return np.full(1820 * 600, 100, np.uint8)
    raise KeyError(kind)


ODD = ["tail_short", "tail_long", "random", "random_low", "constant"]


def _digest(frames, levels):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(frames).tobytes())
    h.update(np.array(levels[:2], np.float64).tobytes())
    h.update(np.array([levels[2]], np.uint64).tobytes())
    return h.hexdigest()


@pytest.mark.skipif(not L.have_raw28_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("cap", sorted(CAPTURES))
@pytest.mark.parametrize("c", CASES, ids=[c[0] for c in CASES])
def test_oracle_equals_reference_extract(c, cap, tmp_path):
    if cap == "long" and c[0] not in ("default", "marksig"):
        pytest.skip("the long capture runs with two switch sets")
    capture = L.raw28_capture(*CAPTURES[cap])
    opts = L.raw28_oracle_opts(**c[1])
    got, lv = L.raw28_oracle_run(opts, capture)
    want, lv2 = L.raw28_ref_run(opts, capture, tmp_path / "cap.u8")
    assert got.shape == want.shape and got.shape[0] >= (CAPTURES[cap][0] - 2) * (1 if "sample_rate" not in c[1] else 0.5)
    assert np.array_equal(got, want)
    assert lv == lv2


@pytest.mark.skipif(not L.have_raw28_ref(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("kind", ODD)
def test_oracle_equals_reference_extract_on_odd_captures(kind, tmp_path):
    capture = _odd_capture(kind)
    for kw in ({}, {"mark_sync": 1}):
        opts = L.raw28_oracle_opts(**kw)
        got, lv = L.raw28_oracle_run(opts, capture)
        want, lv2 = L.raw28_ref_run(opts, capture, tmp_path / "cap.u8")
        assert got.shape == want.shape and got.shape[0] >= 1
        assert np.array_equal(got, want) and lv == lv2


@pytest.mark.skipif(not L.have_raw28_ref(), reason="oracle/_ref not built (no /root/reference)")
def test_front_end_equals_reference_extract():
    capture = L.raw28_capture(3, 5, 3, 1000)
    for kw in ({}, {"mark_sync": 1}):
        opts = L.raw28_oracle_opts(**kw)
        h1, r1 = L.raw28_oracle_front(opts, capture)
        h2, r2 = L.raw28_ref_front(opts, capture)
        assert np.array_equal(h1, h2) and np.array_equal(r1, r2)


def test_oracle_reproduces_reference_hashes():
    """tests/golden/raw28_hashes.json was written from the REFERENCE extract (make_golden.py)."""
    gold = json.load(open(GOLD))
    for key, want in gold.items():
        cname, cap = key.split("@")
        c = [x for x in CASES if x[0] == cname][0]
        capture = L.raw28_capture(*CAPTURES[cap])
        frames, lv = L.raw28_oracle_run(L.raw28_oracle_opts(**c[1]), capture)
        assert _digest(frames, lv) == want["sha256"], key
        assert frames.shape[0] == want["fields"]


def test_flag_mirror_and_geometry():
    lib = L.product()
    o = _capi.make_raw28_opts(["-marksig", "-nosig", "-s", "40mhz", "-width", "800", "-i", "x", "-o", "y", "-420"])
    assert (o.mark_sync, o.disable_sync, o.sample_rate) == (1, 1, 40e6)
    with pytest.raises(_capi.NtscsimError):
        _capi.make_raw28_opts(["-bogus"])
    with pytest.raises(_capi.NtscsimError):
        _capi.make_raw28_opts(["-width", "16"])
    import ctypes as C
    w, h, sl = C.c_int(), C.c_int(), C.c_int()
    d = _capi.make_raw28_opts([])
    assert lib.ntscsim_raw28_geometry(C.byref(d), C.byref(w), C.byref(h), C.byref(sl)) == 0
    assert (w.value, h.value, sl.value) == (1820, 262, 1820)
    assert lib.ntscsim_raw28_geometry(C.byref(o), C.byref(w), C.byref(h), C.byref(sl)) == 0
    assert (w.value, h.value, sl.value) == (2542, 262, 2542)


# ------------------------------------------------------------------------------------------- GPU
def _hip_run(flags, capture, on_device=False, warm=None, chunk=None, max_fields=None):
    import torch
    import ntscsim
    dec = ntscsim.Raw28Decoder(flags)
    if warm is not None or chunk is not None:
        dec.set_speculation(-1 if warm is None else warm, 0 if chunk is None else chunk)
    cap_f = capture.size // (dec.scanline * 262) + 2
    frames = torch.full((cap_f if max_fields is None else max_fields, dec.height, dec.width * 4 + 16), 9,
                        dtype=torch.uint8, device="cuda")
    src = torch.from_numpy(capture).cuda() if on_device else capture
    n = dec.decode(src, frames)
    out = frames[:n].cpu().numpy()
    assert (out[:, :, dec.width * 4:] == 0).all()          # the tool's memset covers linesize x height (:1024)
    res = (np.ascontiguousarray(out[:, :, :dec.width * 4]), dec.levels(), dec.stats(), dec)
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("cap", sorted(CAPTURES))
@pytest.mark.parametrize("c", CASES, ids=[c[0] for c in CASES])
def test_hip_equals_oracle(c, cap):
    if cap == "long" and c[0] not in ("default", "marksig", "nosig", "rate40"):
        pytest.skip("the long capture runs with four switch sets")
    capture = L.raw28_capture(*CAPTURES[cap])
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(**c[1]), capture)
    got, lv2, st, dec = _hip_run(c[2], capture, on_device=(cap == "noisy_cut"))
    assert got.shape == want.shape, (got.shape, want.shape)
    bad = [int(f) for f in range(got.shape[0]) if not np.array_equal(got[f], want[f])]
    assert not bad, (bad, st)
    assert lv2 == lv
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ODD)
def test_hip_equals_oracle_on_odd_captures(kind):
    """incl. the records the tool reads past the buffered stream (never filled / stale)"""
    capture = _odd_capture(kind)
    for kw, flags in (({}, []), ({"mark_sync": 1}, ["-marksig"])):
        want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(**kw), capture)
        got, lv2, st, dec = _hip_run(flags, capture)
        assert got.shape == want.shape, (got.shape, want.shape)
        assert np.array_equal(got, want), st
        assert lv2 == lv, (lv2, lv, st)
        if kind == "tail_short":
            assert st["pulses_past_stream"] >= 1 and st["zero_records"] > 0, st        # never-filled records
        if kind == "tail_long":
            assert st["pulses_past_stream"] >= 1 and st["zero_records"] == 0, st       # stale records of an earlier window
        dec.close()


@pytest.mark.gpu
def test_hip_front_end_equals_oracle_and_repairs_are_exact():
    """hsync_dc_raw of every sample; then with the speculation crippled (no warm-up, or a tiny one
    with large chunks) so that the repair rounds do the work -- results must not change."""
    capture = L.raw28_capture(4, 11, 3, 4321)
    h, _ = L.raw28_oracle_front(L.raw28_oracle_opts(), capture)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    for warm, chunk in ((None, None), (0, 65536), (2, 16384), (64, 1024)):
        got, lv2, st, dec = _hip_run([], capture, warm=warm, chunk=chunk)
        assert np.array_equal(dec.read_front(capture.size), h), (warm, chunk)
        assert np.array_equal(got, want) and lv2 == lv
        if warm == 0:
            assert st["front_rounds"] >= 1 and st["chunks_repaired"] >= 1
        if warm is None:
            assert st["front_rounds"] <= 3, st               # the default warm-up leaves a few links to one or two repair rounds
        dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [0, 3, 16, 1000])
def test_hip_front_end_cheap_warm_up_is_only_a_guess(monkeypatch, exact):
    """The follower's warm-up takes its first scanlines in closed form from sweep 1's 64-sample records (round 4) and
    only the last NTSCSIM_RAW28_EXACT scanlines sample by sample.  Whatever the split -- no exact part at all (every link
    is then closed by the repair rounds), a short one, the default, all of it exact (the round-3 walk) -- the bytes,
    frames and levels are the oracle's; also with chunks that are not a multiple of 64 (no records, exact walk)."""
    monkeypatch.setenv("NTSCSIM_RAW28_EXACT", str(exact))
    capture = L.raw28_capture(5, 31, 4, 1234)
    h, _ = L.raw28_oracle_front(L.raw28_oracle_opts(), capture)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    for warm, chunk in ((None, None), (None, 8192), (40, 16384), (None, 4112), (112, 64 * 37)):
        got, lv2, st, dec = _hip_run([], capture, warm=warm, chunk=chunk)
        assert np.array_equal(dec.read_front(capture.size), h), (exact, warm, chunk, st)
        assert np.array_equal(got, want) and lv2 == lv, (exact, warm, chunk)
        if exact == 0 and chunk != 4112:
            assert st["chunks_repaired"] >= 1, st            # a closed-form guess is a few ulp off: the links cannot all hold
        dec.close()


@pytest.mark.gpu
def test_hip_front_end_on_a_very_noisy_capture():
    """At the generator's noise level 24 the follower's default warm-up leaves most links open (profiles/r04_raw28_noise.txt):
    several repair rounds in a row, with the closed-form warm-up and without it, and the bytes are still the oracle's."""
    capture = L.raw28_capture(5, 41, 24, 999)
    h, _ = L.raw28_oracle_front(L.raw28_oracle_opts(), capture)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    for warm, chunk in ((None, None), (20, None), (None, 8192)):
        got, lv2, st, dec = _hip_run([], capture, warm=warm, chunk=chunk)
        assert np.array_equal(dec.read_front(capture.size), h), (warm, chunk, st)
        assert np.array_equal(got, want) and lv2 == lv, (warm, chunk)
        assert st["tail_rounds"] == 1, st          # the comb tails' first guess settles at this noise level too (16 + 4 scanlines)
        dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("group,rounds", [(1, 0), (3, 0), (192, 1), (2, 1)])
def test_hip_back_half_in_groups_behind_the_walk(monkeypatch, group, rounds):
    """Levels, comb tails and rendering run per group of fields on the GPU while the host's sync walk goes on (round 4):
    whatever the group size -- one field per group puts every field boundary on a group boundary -- and also when the
    comb tails' first guess is declared unsettled (test hook: rounds over all scanlines, everything rendered again),
    frames and levels are the oracle's; as a stream in pieces too."""
    monkeypatch.setenv("NTSCSIM_RAW28_GROUP", str(group))
    monkeypatch.setenv("NTSCSIM_RAW28_TAILROUNDS", str(rounds))
    capture = L.raw28_capture(6, 17, 3, 4000)
    for kw, flags in (({}, []), ({"disable_subcarrier": 1}, ["-nosc"]), ({"mark_sync": 1, "show_subcarrier": 1}, ["-marksig", "-showsc"])):
        want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(**kw), capture)
        got, lv2, st, dec = _hip_run(flags, capture)
        assert got.shape == want.shape and np.array_equal(got, want), (group, rounds, flags, st)
        assert lv2 == lv
        if rounds and "-nosc" not in flags:
            assert st["tail_rounds"] >= 5, st                  # the first look + at least one batch of four rounds
        dec.close()
        got, lv3, _, dec2 = _hip_stream(flags, capture, [700001, 1, 250000])
        assert np.array_equal(got, want) and lv3 == lv, (group, rounds, flags)
        dec2.close()


@pytest.mark.gpu
def test_hip_front_end_in_segments(monkeypatch):
    """The front end works through at most 2^29 new samples at a time (8 bytes of fp64 plane per sample); with the
    segment shrunk by its test hook the loop over segments runs on a small capture: odd segment sizes, segments
    shorter than the warm-up, one that leaves fewer than 16 samples for the last segment -- same bytes, same frames."""
    capture = L.raw28_capture(3, 21, 3, 777)
    h, _ = L.raw28_oracle_front(L.raw28_oracle_opts(), capture)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    for seg in (1000003, 65537, capture.size - 5, 4096):
        monkeypatch.setenv("NTSCSIM_RAW28_SEG", str(seg))
        got, lv2, st, dec = _hip_run([], capture)
        assert np.array_equal(dec.read_front(capture.size), h), seg
        assert np.array_equal(got, want) and lv2 == lv, seg
        dec.close()
    monkeypatch.delenv("NTSCSIM_RAW28_SEG")


@pytest.mark.gpu
def test_hip_reproduces_reference_hashes():
    gold = json.load(open(GOLD))
    for key, want in gold.items():
        cname, cap = key.split("@")
        if cap == "long" and cname != "default":
            continue
        c = [x for x in CASES if x[0] == cname][0]
        got, lv, _, dec = _hip_run(c[2], L.raw28_capture(*CAPTURES[cap]))
        assert _digest(got, lv) == want["sha256"], key
        dec.close()


@pytest.mark.gpu
def test_hip_max_fields_and_errors():
    import torch
    import ntscsim
    capture = L.raw28_capture(4, 2, 1, 0)
    want, _ = L.raw28_oracle_run(L.raw28_oracle_opts(), capture, max_fields=2)
    got, _, _, dec = _hip_run([], capture, max_fields=2)
    assert got.shape[0] == 2 and np.array_equal(got, want)
    small = torch.zeros((1, dec.height, dec.width * 4 - 4), dtype=torch.uint8, device="cuda")
    with pytest.raises(ntscsim.NtscsimError) as e:
        dec.decode(capture, small)
    assert e.value.code == _capi.E_SIZE
    # a capture shorter than 256 scanlines decodes to zero fields, like the tool
    frames = torch.zeros((2, dec.height, dec.width * 4), dtype=torch.uint8, device="cuda")
    assert dec.decode(capture[:1820 * 200].copy(), frames) == 0
    dec.close()



def _hip_stream(flags, capture, sizes, frames_per_push=None, device_every=0, warm=None, chunk=None):
    """The capture pushed in pieces of the given sizes (cycled), the last piece flagged final; then drained
    with empty pushes while fields keep coming.  Returns (frames, levels, number of pushes)."""
    import torch
    import ntscsim
    dec = ntscsim.Raw28Decoder(flags)
    if warm is not None or chunk is not None:
        dec.set_speculation(-1 if warm is None else warm, 0 if chunk is None else chunk)
    cap_f = capture.size // (dec.scanline * 240) + 4
    ring = torch.full((cap_f if frames_per_push is None else frames_per_push, dec.height, dec.width * 4), 9,
                      dtype=torch.uint8, device="cuda")
    out, pos, k = [], 0, 0
    dec.stream_reset()
    while True:
        n = min(sizes[k % len(sizes)], capture.size - pos)
        piece = capture[pos:pos + n]
        pos += n
        final = pos >= capture.size
        src = piece if n else None
        if n and device_every and k % device_every == 0:
            src = torch.from_numpy(np.ascontiguousarray(piece)).cuda()
        got = dec.stream_push(src, ring, final=final)
        out.append(ring[:got].cpu().numpy().copy())
        k += 1
        if final:
            while got == ring.shape[0]:               # held back by max_fields: drain
                got = dec.stream_push(None, ring, final=True)
                out.append(ring[:got].cpu().numpy().copy())
                k += 1
            break
    frames = np.concatenate(out) if out else np.zeros((0, dec.height, dec.width * 4), np.uint8)
    res = (frames, dec.levels(), k, dec)
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("name,sizes,fpp,devn", [
    ("one-megasample", [1000003], None, 0),                 # not a multiple of the 16-sample load grid
    ("uneven", [1, 15, 4099, 2500000, 17, 777777, 3000001], None, 3),
    ("one-field-at-a-time", [2 * 477750 + 5], 1, 0),        # every push may hand back a single field
    ("whole-windows", [2048 * 1820], 12, 2),
])
def test_hip_stream_equals_one_shot(name, sizes, fpp, devn):
    """ntscsim_raw28_stream_push: a 40-field capture (19 M samples, five buffer windows of the tool, several
    compactions of the device buffer) pushed in pieces == the oracle's run on the whole capture: frames,
    levels and stream position, whatever the piece sizes, the frames-per-push limit and where the pieces
    live."""
    capture = L.raw28_capture(40, 17, 3, 250001)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    got, lv2, npush, dec = _hip_stream([], capture, sizes, fpp, devn)
    assert got.shape == want.shape, (got.shape, want.shape, npush)
    bad = [int(f) for f in range(got.shape[0]) if not np.array_equal(got[f], want[f])]
    assert not bad, bad
    assert lv2 == lv
    st = dec.stats()
    # the stream never sat on the device as a whole: old samples were dropped as the window moved on
    # (unless the caller takes one field per push while handing in two fields' worth of samples)
    if fpp != 1:
        assert st["compactions"] >= 1 and st["max_samples_held"] < capture.size * 3 // 4, st
    dec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["tail_long", "tail_short", "random"])
def test_hip_stream_on_odd_captures(kind):
    """the stale / never-filled records of the tool's sample array (:655-676) when the capture arrives in
    pieces, with marksig, and with the front end's speculation crippled"""
    capture = _odd_capture(kind)
    for kw, flags, warm, chunk in (({}, [], None, None), ({"mark_sync": 1}, ["-marksig"], 2, 16384)):
        want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(**kw), capture)
        got, lv2, _, dec = _hip_stream(flags, capture, [1234567, 333, 2000000], None, 2, warm, chunk)
        assert got.shape == want.shape and np.array_equal(got, want), kind
        assert lv2 == lv
        dec.close()


@pytest.mark.gpu
def test_hip_stream_api_contract():
    import torch
    import ntscsim
    dec = ntscsim.Raw28Decoder([])
    frames = torch.zeros((4, dec.height, dec.width * 4), dtype=torch.uint8, device="cuda")
    # an empty capture is a capture: zero fields, no error
    assert dec.decode(np.zeros(0, np.uint8), frames) == 0
    dec.stream_reset()
    assert dec.stream_push(None, frames, final=True) == 0
    # samples after the end of a stream are refused until the next reset
    with pytest.raises(ntscsim.NtscsimError) as e:
        dec.stream_push(np.zeros(100, np.uint8), frames)
    assert e.value.code == _capi.E_ARG
    # a decode() after a stream starts from scratch, and a stream after a decode() too
    capture = L.raw28_capture(5, 1, 0, 0)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    big = torch.zeros((8, dec.height, dec.width * 4), dtype=torch.uint8, device="cuda")
    n = dec.decode(capture, big)
    assert n == want.shape[0] and np.array_equal(big[:n].cpu().numpy(), want) and dec.levels() == lv
    dec.stream_reset()
    n1 = dec.stream_push(capture[:3000000], big)
    n2 = dec.stream_push(capture[3000000:], big[n1:], final=True)
    assert n1 + n2 == want.shape[0] and np.array_equal(big[:n1 + n2].cpu().numpy(), want) and dec.levels() == lv
    dec.close()

# ------------------------------------------------------------------------------- command line host
RAW28_CLI = os.path.join(L.PKG, "raw28_cli")


def test_raw28_cli_exists_and_rejects_like_the_tool():
    import subprocess
    assert os.path.exists(RAW28_CLI), "build with make -C composite-video-simulator_amd/csrc"
    r = subprocess.run([RAW28_CLI, "-bogus"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1
    r = subprocess.run([RAW28_CLI, "-i", "x.u8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"No output file specified" in r.stderr          # :510-513
    r = subprocess.run([RAW28_CLI, "-o", "null:"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"No input file specified" in r.stderr           # :514-517


@pytest.mark.gpu
def test_raw28_cli_equals_oracle(tmp_path):
    import subprocess
    capture = L.raw28_capture(4, 9, 2, 31337)
    src, dst = tmp_path / "cap.u8", tmp_path / "out.bgra"
    src.write_bytes(capture.tobytes())
    for flags, kw in ((["-showsc"], {"show_subcarrier": 1}), ([], {})):
        r = subprocess.run([RAW28_CLI] + flags + ["-i", str(src), "-o", str(dst)], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        want, _ = L.raw28_oracle_run(L.raw28_oracle_opts(**kw), capture)
        got = np.frombuffer(dst.read_bytes(), np.uint8).reshape(-1, 262, 1820 * 4)
        assert got.shape == want.shape and np.array_equal(got, want)
    r = subprocess.run([RAW28_CLI, "--max-fields", "2", "-i", str(src), "-o", "null:"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"2 fields of 1820x262" in r.stderr
    # the capture through a pipe, read in 1 MB pieces with a ring of two frames: the same frames
    want, _ = L.raw28_oracle_run(L.raw28_oracle_opts(), capture)
    r = subprocess.run([RAW28_CLI, "--chunk-bytes", "1000000", "--ring-fields", "2", "-i", "-", "-o", str(dst)],
                       input=capture.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    got = np.frombuffer(dst.read_bytes(), np.uint8).reshape(-1, 262, 1820 * 4)
    assert got.shape == want.shape and np.array_equal(got, want)
    # an empty capture: zero fields, exit code 0
    empty = tmp_path / "empty.u8"
    empty.write_bytes(b"")
    r = subprocess.run([RAW28_CLI, "-i", str(empty), "-o", "null:"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"0 fields of 1820x262" in r.stderr


def test_front_end_chunk_length_is_whole_scanlines_of_whole_superblocks():
    """The chunk length of the front end's second sweep (host logic, no GPU): sub-chunks of whole 64-sample superblocks
    of about 2048 samples, as near a whole number of scanlines as that allows -- 16 scanlines of 1820 samples at 8 x fsc
    whatever the stream length the default chunk count divides -- and sensible lengths for other line lengths."""
    import ctypes as C
    lib = _capi.lib()
    def pick(line, target):
        m, q = C.c_int(), C.c_int()
        lib.ntscsim_raw28_debug_pick_chunk(C.c_double(line), C.c_double(target), C.byref(m), C.byref(q))
        return m.value, q.value
    for target in (1000.0, 8000.0, 17480.0, 29120.0, 32768.0):
        assert pick(1820.0, target) == (29120, 13), target
    for line in (1820.0, 1820.3, 2542.2, 1135.0, 910.0, 3640.0):
        for target in (500.0, 17480.0, 30000.0):
            m, q = pick(line, target)
            assert q >= 1 and m % (64 * q) == 0 and 2048 <= m // q <= 4096, (line, target, m, q)
            t = max(target, 16 * line)
            assert 0.55 * t <= m <= 2.0 * t, (line, target, m)
            lines = m / line
            assert abs(lines - round(lines)) * line <= 16.0, (line, target, m)      # at most a quarter superblock per lane
