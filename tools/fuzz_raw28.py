"""Developer tool (GPU box): the raw-composite decoder against its oracle on seeded random captures
(fields, noise level, starting sample, truncation) and random switch sets, speculation settings (warm-up, chunk length, exact
part of the warm-up, chunks per wavefront), front-end segment sizes and -- a third of the time -- as a stream pushed in random pieces:  python tools/fuzz_raw28.py 0 100"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np, torch, ntscsim
import _libs as L
s0, n = int(sys.argv[1]), int(sys.argv[2])
FL = [("mark_sync", "-marksig"), ("disable_sync", "-nosig"), ("disable_wp_equ", "-nowequ"), ("show_subcarrier", "-showsc"),
      ("disable_subcarrier", "-nosc"), ("disable_equalization", "-noequ")]
bad, t0 = [], time.time()
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    cap = L.raw28_capture(r.randrange(2, 6), seed, r.choice([0, 1, 3, 6, 12]), r.randrange(0, 400000))
    cap = np.ascontiguousarray(cap[:cap.size - r.randrange(0, 300000)])
    kw, flags = {}, []
    for k, f in FL:
        if r.random() < 0.25:
            kw[k] = 1; flags.append(f)
    want, lv = L.raw28_oracle_run(L.raw28_oracle_opts(**kw), cap)
    if r.random() < 0.25:
        os.environ["NTSCSIM_RAW28_SEG"] = str(r.choice([4096, 65537, 300001, 1000003]))
    else:
        os.environ.pop("NTSCSIM_RAW28_SEG", None)
    for var, choices in (("NTSCSIM_RAW28_EXACT", [None, None, 0, 2, 12, 1000]), ("NTSCSIM_RAW28_LANES", [None, None, 5, 64])):
        v = r.choice(choices)              # round 4: how much of the follower's warm-up is exact, chunks per wavefront
        if v is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = str(v)
    dec = ntscsim.Raw28Decoder(flags)
    if r.random() < 0.3:
        dec.set_speculation(r.choice([0, 8, 40]), r.choice([1024, 8192, 65536]))
    fr = torch.empty((want.shape[0] + 2, dec.height, dec.width * 4), dtype=torch.uint8, device="cuda")
    if r.random() < 0.33:                          # a stream: random pieces, everything a push yields is collected
        dec.stream_reset()
        pos, nf, ring = 0, 0, torch.empty_like(fr)
        while True:
            step = r.choice([1, 4097, 200003, 700001, 3000001])
            piece = cap[pos:pos + step]
            pos += piece.size
            final = pos >= cap.size
            k = dec.stream_push(piece if piece.size else None, ring, final=final)
            fr[nf:nf + k] = ring[:k]
            nf += k
            while k == ring.shape[0]:              # (never here: the ring holds every field of the capture)
                k = dec.stream_push(None, ring, final=final); fr[nf:nf + k] = ring[:k]; nf += k
            if final:
                break
    else:
        nf = dec.decode(cap, fr)
    got = fr[:nf].cpu().numpy()
    if got.shape != want.shape or not np.array_equal(got, want) or dec.levels() != lv:
        bad.append((seed, flags, got.shape, want.shape))
    dec.close()
print("%d captures in %.1f s, %d failures" % (n, time.time() - t0, len(bad)))
for b in bad[:10]:
    print(b)
