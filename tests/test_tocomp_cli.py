"""tocomp_cli: the ffmpeg_to_composite-compatible host (raw planar YUV in / out).  CPU part: the flag parser mirrors
the tool's (exit 1 on unknown switches, -h).  GPU part: the byte stream it writes equals the tool's loop
(ffmpeg_to_composite.cpp:1783-1800: render_field -> black_key_feedback -> composite_video_process ->
output_frame) replayed with the oracle on ONE persistent frame, like the tool."""
import os
import subprocess

import numpy as np
import pytest

import _libs as L

CLI = os.path.join(L.PKG, "tocomp_cli")


def run(args, **kw):
    return subprocess.run([CLI] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def test_cli_exists_help_and_rejects_like_the_tool():
    assert os.path.exists(CLI)
    r = run(["-h"])
    assert r.returncode == 1
    r = run(["-bogus"])
    assert r.returncode == 1 and b"Unknown switch" in r.stderr
    r = run(["-i", "bars:2"])
    assert r.returncode == 1 and b"You must specify an input and output file" in r.stderr   # :1634
    r = run(["-i", "bars:2", "-o", "null:", "-comp-catv4"])          # ffmpeg_ntsc only
    assert r.returncode == 1


def oracle_loop(p, sources, is420, w, h, interlaced_out, out422, il=0, tff=0, nocomp=False):
    """do_video_decode_and_render() :1783-1800 on one persistent output frame (rows padded by 64 zero bytes, as
    the CLI's frames are); returns the frames output_frame() would have encoded, tightly packed."""
    frame = L.Yuv422(w, h, pad=64)
    flt = L.Yuv422(w, h, pad=64)
    o = L.TocompOracleStream(p, oob=L.OOB_MEMORY)
    level = p.black_key_level_feedback
    outs = []
    vf = 0
    for src in sources:
        for sub in (0, 1):
            field = (vf & 1) ^ 1
            L.tocomp_oracle_render_field(frame, src, is420, il, tff, sub, field)
            if level >= 0:
                L.tocomp_oracle_black_key(frame, flt, field, level)
            if not nocomp:
                o.process(frame, field, vf)
            emit = (vf & 1) == 1 if interlaced_out else True
            if emit:
                if interlaced_out and out422:
                    outs.append(np.concatenate([frame.pix(i).reshape(-1) for i in range(3)]))
                else:
                    mode = 0 if out422 else (2 if interlaced_out else 1)
                    bob = L.Yuv422(w, h)
                    bob.buf[:] = 0
                    f_arg = ((vf - 1) & 1) ^ 1 if interlaced_out else field
                    L.tocomp_oracle_output_frame(bob, frame, f_arg, mode)
                    crows = h if mode == 0 else (h + 1) // 2
                    outs.append(np.concatenate([bob.pix(0).reshape(-1), bob.pix(1)[:crows].reshape(-1),
                                                bob.pix(2)[:crows].reshape(-1)]))
            vf += 1
    return outs


def write_sources(path, sources, is420):
    with open(path, "wb") as f:
        for s in sources:
            crows = (s.h + 1) // 2 if is420 else s.h
            f.write(s.pix(0).tobytes())
            f.write(s.pix(1)[:crows].tobytes())
            f.write(s.pix(2)[:crows].tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("flags,is420,sh,extra", [
    (["-vhs"], 0, 36, []),                                   # bob, 4:2:0 out (the tool's default)
    (["-vhs", "-422"], 0, 36, []),                           # bob, 4:2:2 out
    (["-vhs", "-vi"], 0, 36, []),                            # interlaced 4:2:0 repack, one frame per pair
    (["-vhs", "-vi", "-422"], 0, 36, []),                    # the processed frame itself
    (["-vhs", "-vhs-speed", "ep", "-422"], 1, 50, ["--src-420"]),         # 4:2:0 source of another height
    (["-422", "-bkey-feedback", "40"], 0, 36, []),           # the frame-to-frame recurrence: one field per call
    (["-vhs", "-nocomp", "-422"], 0, 40, []),                # render only
    (["-vhs", "-422"], 0, 36, ["--src-interlaced", "--src-tff"]),
])
def test_cli_stream_equals_the_tools_loop(tmp_path, flags, is420, sh, extra):
    # (-vi at height 34 = 2 mod 4: the tool's interlaced 4:2:0 repack writes chroma row (h + 1) / 2, one past the plane,
    #  :1215-1223; the CLI keeps a spare row for it and writes the plane's own rows out)
    w, n = 96, 7
    h = 34 if flags == ["-vhs", "-vi"] else 36
    rng = np.random.RandomState(sh)
    sources = []
    for j in range(n):
        s = L.yuv_noise(w, sh, 100 + j)
        if "-bkey-feedback" in flags:          # dark frames so that the key fires on part of the picture
            s.pix(0)[:, : w // 2] = rng.randint(16, 40, size=(sh, w // 2), dtype=np.uint8)
            s.pix(1)[:, : w // 4] = 128
            s.pix(2)[:, : w // 4] = 128
        sources.append(s)
    inp, outp = tmp_path / "in.yuv", tmp_path / "out.yuv"
    write_sources(inp, sources, is420)
    args = flags + ["-width", str(w), "--height", str(h), "--src-height", str(sh), "--batch", "3",
                    "-i", str(inp), "-o", str(outp)] + extra
    r = run(args)
    assert r.returncode == 0, r.stderr.decode()
    p = L.make_params_tocomp([f for f in flags if f not in ("-422", "-vi")] + ["-width", str(w)], output_height=h)
    il, tff = int("--src-interlaced" in extra), int("--src-tff" in extra)
    exp = oracle_loop(p, sources, is420, w, h, "-vi" in flags, "-422" in flags, il, tff, "-nocomp" in flags)
    got = np.frombuffer(outp.read_bytes(), np.uint8)
    exp_all = np.concatenate(exp)
    assert got.size == exp_all.size, (got.size, exp_all.size, len(exp))
    bad = np.nonzero(got != exp_all)[0]
    assert bad.size == 0, "first mismatch at byte %d of %d" % (bad[0], got.size)
