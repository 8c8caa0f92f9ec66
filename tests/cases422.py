"""Flag matrix of the 8-bit YUV422P variant (ffmpeg_to_composite switches)."""

CASES422 = [
    # name, flags, W, H, nfields, source
    ("default", [], 96, 32, 4, "noise"),
    ("default_bars", [], 96, 32, 4, "bars"),
    ("noise0", ["-noise", "0"], 96, 32, 3, "noise"),
    ("vhs", ["-vhs"], 96, 32, 4, "noise"),
    ("vhs_bars", ["-vhs"], 100, 34, 4, "bars"),
    ("vhs_oddh", ["-vhs"], 96, 33, 3, "noise"),
    ("vhs_ep", ["-vhs", "-vhs-speed", "ep"], 96, 32, 4, "noise"),
    ("vhs_lp_only", ["-vhs-speed", "lp"], 96, 32, 4, "noise"),
    ("vhs_svideo", ["-vhs", "-vhs-svideo", "1"], 96, 32, 4, "noise"),
    ("vhs_noblend", ["-vhs", "-vhs-chroma-vblend", "0"], 96, 32, 4, "noise"),
    ("vhs_pal", ["-tvstd", "pal", "-vhs"], 96, 36, 4, "noise"),
    ("out_lite_only", ["-out-composite-lowpass", "0"], 96, 32, 4, "noise"),
    ("no_outlp", ["-out-composite-lowpass", "0", "-out-composite-lowpass-lite", "0"], 96, 32, 4, "noise"),
    ("no_inlp", ["-in-composite-lowpass", "0"], 96, 32, 4, "noise"),
    ("catv", ["-comp-catv"], 96, 32, 4, "noise"),
    ("catv3_vhs", ["-vhs", "-comp-catv3"], 96, 32, 4, "noise"),
    ("phase0", ["-comp-phase", "0", "-comp-phase-offset", "1"], 96, 32, 4, "noise"),
    ("phase90", ["-vhs", "-comp-phase", "90"], 96, 32, 4, "noise"),
    ("phase270", ["-vhs", "-comp-phase", "270", "-comp-phase-offset", "3"], 96, 32, 4, "noise"),
    ("nocolor", ["-nocolor-subcarrier"], 96, 32, 4, "noise"),
    ("after_yc_sep", ["-vhs", "-nocolor-subcarrier-after-yc-sep"], 96, 32, 4, "noise"),
    ("yc_recomb2", ["-vhs", "-yc-recomb", "2"], 96, 32, 4, "noise"),
    ("amp30", ["-vhs", "-subcarrier-amp", "30"], 96, 32, 4, "noise"),
    ("amp1", ["-vhs", "-subcarrier-amp", "1"], 96, 32, 4, "noise"),
    ("dropout_often", ["-vhs", "-chroma-dropout", "50000"], 96, 32, 4, "noise"),
    ("phase_noise20", ["-chroma-phase-noise", "20"], 96, 32, 4, "noise"),
    ("hs_inframe", ["-vhs", "-vhs-head-switching-point", "0.105"], 96, 32, 4, "noise"),
    ("hs_inframe_b", ["-vhs", "-vhs-head-switching-point", "0.1013"], 96, 32, 4, "noise"),
    ("hs_tall", ["-vhs"], 48, 480, 2, "noise"),
    ("wide", ["-vhs"], 720, 12, 2, "bars"),
]


def make_source422(kind, w, h, idx, pad=0):
    import _libs as L
    return L.yuv_noise(w, h, 1000 + idx, pad) if kind == "noise" else L.yuv_bars(w, h, idx, pad)
