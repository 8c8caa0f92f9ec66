"""The flag matrix shared by the golden-vector generator, the oracle tests and the GPU parity
tests.  Each case: (name, reference CLI switches, width, height, n_fields, source kind, extra)."""

CASES = [
    # name, flags, W, H, nfields, src, interlaced, tff
    ("default", [], 96, 32, 4, "noise", 0, 0),
    ("default_64x16", [], 64, 16, 2, "noise", 0, 0),
    ("default_bars", [], 96, 32, 4, "bars", 0, 0),
    ("impulse", [], 64, 16, 2, "impulse", 0, 0),
    ("ramp", ["-vhs"], 64, 16, 2, "ramp", 0, 0),
    ("noise0", ["-noise", "0"], 96, 32, 4, "noise", 0, 0),
    ("vhs", ["-vhs"], 96, 32, 4, "noise", 0, 0),
    ("vhs_odd", ["-vhs"], 97, 33, 3, "noise", 0, 0),
    ("vhs_w100", ["-vhs"], 100, 34, 4, "bars", 0, 0),
    ("vhs_ep", ["-vhs", "-vhs-speed", "ep"], 96, 32, 4, "noise", 0, 0),
    ("vhs_lp_only", ["-vhs-speed", "lp"], 96, 32, 4, "noise", 0, 0),
    ("vhs_svideo", ["-vhs", "-vhs-svideo", "1"], 96, 32, 4, "noise", 0, 0),
    ("vhs_noblend", ["-vhs", "-vhs-chroma-vblend", "0"], 96, 32, 4, "noise", 0, 0),
    ("vhs_pal", ["-tvstd", "pal", "-vhs"], 96, 36, 4, "noise", 0, 0),
    ("vhs_full_outlp", ["-vhs", "-out-composite-lowpass-lite", "0"], 96, 32, 4, "noise", 0, 0),
    ("no_outlp", ["-out-composite-lowpass", "0"], 96, 32, 4, "noise", 0, 0),
    ("no_inlp", ["-in-composite-lowpass", "0"], 96, 32, 4, "noise", 0, 0),
    ("catv", ["-comp-catv"], 96, 32, 4, "noise", 0, 0),
    ("catv2", ["-comp-catv2"], 96, 32, 4, "noise", 0, 0),
    ("catv3", ["-comp-catv3"], 96, 32, 4, "noise", 0, 0),
    ("catv4_vhs", ["-vhs", "-comp-catv4"], 96, 32, 4, "noise", 0, 0),
    ("phase0", ["-comp-phase", "0", "-comp-phase-offset", "1"], 96, 32, 4, "noise", 0, 0),
    ("phase90", ["-vhs", "-comp-phase", "90"], 96, 32, 4, "noise", 0, 0),
    ("phase270", ["-vhs", "-comp-phase", "270", "-comp-phase-offset", "3"], 96, 32, 4, "noise", 0, 0),
    ("phase180_off2", ["-comp-phase-offset", "2"], 96, 32, 4, "noise", 0, 0),
    ("nocolor", ["-nocolor-subcarrier"], 96, 32, 4, "noise", 0, 0),
    ("nocolor_vhs", ["-vhs", "-nocolor-subcarrier"], 96, 32, 4, "noise", 0, 0),
    ("amp30", ["-vhs", "-subcarrier-amp", "30"], 96, 32, 4, "noise", 0, 0),
    ("amp1", ["-vhs", "-subcarrier-amp", "1"], 96, 32, 4, "noise", 0, 0),
    ("dropout_often", ["-vhs", "-chroma-dropout", "50000"], 96, 32, 4, "noise", 0, 0),
    ("phase_noise20", ["-chroma-phase-noise", "20"], 96, 32, 4, "noise", 0, 0),
    ("chroma_noise_only", ["-noise", "0", "-chroma-noise", "40"], 96, 32, 4, "noise", 0, 0),
    # head switch moved into the frame: negative and positive displacement
    ("hs_neg", ["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.002"],
     96, 32, 4, "noise", 0, 0),
    ("hs_pos", ["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase", "0.0012"],
     96, 32, 4, "noise", 0, 0),
    ("hs_nonoise", ["-vhs", "-vhs-head-switching-point", "0.105", "-vhs-head-switching-phase",
                    "0.002", "-vhs-head-switching-noise-level", "0"], 96, 32, 4, "noise", 0, 0),
    ("hs_default_tall", ["-vhs"], 48, 480, 2, "noise", 0, 0),
    ("interlaced_tff", ["-vhs"], 96, 32, 4, "noise", 1, 1),
    ("interlaced_bff", [], 96, 32, 4, "noise", 1, 0),
    ("wide", ["-vhs"], 720, 12, 2, "bars", 0, 0),
]


def make_source(kind, w, h, idx):
    import numpy as np
    import _libs as L
    if kind == "noise":
        return L.noise_frame(w, h, 0x1234567 + idx)
    if kind == "bars":
        return L.bars(w, h, idx)
    a = np.zeros((h, w, 4), np.uint8)
    if kind == "impulse":
        a[(h // 2 + idx) % h, w // 3, :3] = 255
        a[1, w - 2, 2] = 255
        return a
    if kind == "ramp":
        x = np.arange(w, dtype=np.int64)
        a[:, :, 0] = (x * 255 // (w - 1))[None, :]
        a[:, :, 1] = ((w - 1 - x) * 255 // (w - 1))[None, :]
        a[:, :, 2] = (np.arange(h) * 255 // (h - 1))[:, None]
        return a
    raise ValueError(kind)


def case_jobs(nfields):
    """(src frame index, field, fieldno) per output field, as the field loop issues them
    (ffmpeg_ntsc.cpp:2229): frame k//2, field (k&1)^1, fieldno k."""
    return [(k // 2, (k & 1) ^ 1, k) for k in range(nfields)]
