"""Developer tool (GPU box): ntscsim_field422() / ntscsim_submit422() on random loops (switch set, geometry, row padding,
output mode, interlaced output, source shape, depth, how far behind the submits the caller waits, page-owned planes)
against the oracle's loop on byte-identical buffers, whole buffers (the oracle in PLANE mode: the library's contract for the
separator's read past a row -- every byte is compared, tight rows and iterations without a source included).
    python tools/fuzz_host422.py 70000 400"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import _libs as L
import test_host422 as H

s0, n = int(sys.argv[1]), int(sys.argv[2])
bad, t0, paths = [], time.time(), [0, 0]
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    f = r.choice([["-vhs"], [], ["-vhs", "-vhs-speed", "ep"], ["-vhs", "-vhs-svideo", "1"], ["-tvstd", "pal", "-vhs"], ["-vhs", "-comp-catv"]])
    bkey = r.random() < 0.15
    if bkey: f = f + ["-bkey-feedback", str(r.choice([10, 40]))]
    w = r.choice([64, 96, 130, 178])
    h = r.choice([10, 17, 34, 36, 63])
    pad = r.choice([0, 1, 2, 8, 64])
    il = r.random() < 0.3
    out_mode = r.choice([H.OUT_INT420, H.OUT_FRAME, None]) if il else r.choice([H.OUT_BOB420, H.OUT_BOB422])
    src_flags, sh = 0, None
    q = r.random()
    if q < 0.15: src_flags, sh = H.F_420, r.choice([h, h + 14])
    elif q < 0.3: src_flags, sh = H.F_IL | (H.F_TFF if r.random() < 0.5 else 0), r.choice([h, h + 4])
    elif q < 0.36: src_flags = H.F_NOCOMP
    mode = r.choice(["sync", "submit", "submit", "submit"])
    depth = r.choice([1, 2, 3, 8, 32])
    lag = r.choice([None, None, 0, 1, 5])
    with_src = r.random() > 0.08
    try:
        p = L.make_params_tocomp(f + ["-width", str(w)], output_height=h)
        st = H.run_loop(p, w, h, pad, r.choice([2, 3, 5, 9]), out_mode, mode, sh=sh, src_flags=src_flags, bkey=bkey, interlaced_out=il,
                        depth=depth, with_src=with_src, frame_seed=seed & 0xFFFF, lag=lag if mode == "submit" else None)
        paths[0] += st[3]; paths[1] += st[4]
    except AssertionError as e:
        bad.append((seed, f, w, h, pad, out_mode, il, mode, depth, lag, str(e)[:160]))
print("%d random loops in %.1f s, %d failures; iterations batched / one at a time: %d / %d" % (n, time.time() - t0, len(bad), paths[0], paths[1]))
for b in bad[:10]:
    print(b)
