"""Developer tool (GPU box): ntscsim_submit() / ntscsim_wait() on random loops against the oracle's loop.
  mode A  a ring deeper than the lag (the INTEGRATION 1b patch): every consumed frame == the synchronous loop's snapshot;
  mode B  rings SHALLOWER than the fields in flight (1-3 frames): fields share destination frames while in flight; at
          random points everything is waited for and every ring frame must be what the in-order loop left there (the
          header's "delivered in submit order", with the decoder writing into the caller's pinned frames itself).
Random: geometry (aligned / unaligned rows, padding), switch set, depth, lanes, line doubling, pinned / staged, lag.
    python tools/fuzz_submit.py 30000 300"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import numpy as np
import _libs as L
import ntscsim
import test_submit as T

s0, n = int(sys.argv[1]), int(sys.argv[2])
bad, t0, cnt = [], time.time(), [0, 0]
for seed in range(s0, s0 + n):
    r = random.Random(seed)
    flags = r.choice([["-vhs"], [], ["-vhs", "-vhs-speed", "ep"], ["-vhs", "-comp-catv2"], ["-vhs", "-vhs-svideo", "1"], ["-tvstd", "pal", "-vhs"]])
    pin = r.random() < 0.7
    w = r.choice([192, 256, 180, 320]) if pin else r.choice([96, 100, 192])
    h = r.choice([96, 99, 130]) if pin else r.choice([32, 33, 64])
    nf = r.choice([12, 20, 34])
    depth = r.choice([1, 2, 3, 4, 8])
    lanes = r.choice([1, 2, 3, 4])
    bob = r.random() < 0.4
    p = L.make_params(flags, output_height=h)
    frames = [L.noise_frame(w, h, seed * 7 + j) for j in range(nf // 2)]
    sim = ntscsim.FieldSimulator(params=p)
    try:
        if r.random() < 0.5:
            # ---- mode A
            lag = r.choice([depth, 2 * depth + 1, 9])
            ring = lag + r.choice([1, 3])
            sim.submit_configure(depth=depth, slots=max(2 * depth, ring + depth + 2), lanes=lanes, pin=pin, min_pin_bytes=0)
            exp, exp_pos = T.reference_loop(p, frames, nf, w, h, ring, bob)
            got = T.run_submit_loop(sim, frames, nf, w, h, ring, bob, lag, pad=r.choice([0, 4, 8]))
            for k in range(nf):
                if not np.array_equal(got[k], exp[k]):
                    bad.append((seed, "A", flags, w, h, depth, lanes, bob, pin, ring, lag, "field %d" % k)); break
            if sim.rng_pos != exp_pos: bad.append((seed, "A rng_pos"))
            cnt[0] += 1
        else:
            # ---- mode B
            ring = r.choice([1, 2, 3])
            sim.submit_configure(depth=depth, slots=max(2 * depth, 8), lanes=lanes, pin=pin, min_pin_bytes=0)
            o = L.OracleStream(p)
            ebuf = [np.full((h, w, 4), 0x5A, np.uint8) for _ in range(ring)]
            src = T.page_frame(h, w)
            gbuf = [T.page_frame(h, w, 0x5A) for _ in range(ring)]
            for k in range(nf):
                field = (k & 1) ^ 1
                o.field(ebuf[k % ring], frames[k // 2], field, k)
                if bob: T.oracle_bob(ebuf[k % ring], field)
                if k % 2 == 0: src[:] = frames[k // 2]
                sim.submit(gbuf[k % ring], src, field, k, bob=bob, same_src=(k % 2 == 1))
                if r.random() < 0.15 or k == nf - 1:
                    sim.wait()
                    for q in range(ring):
                        if not np.array_equal(gbuf[q], ebuf[q]):
                            bad.append((seed, "B", flags, w, h, depth, lanes, bob, pin, ring, "after field %d frame %d" % (k, q))); break
            if sim.rng_pos != o.rng_pos: bad.append((seed, "B rng_pos"))
            sim.host_unpin()
            cnt[1] += 1
    except AssertionError as e:
        bad.append((seed, flags, w, h, depth, lanes, bob, pin, str(e)[:120]))
    sim.close()
print("%d random loops (%d ring > lag, %d shared frames) in %.1f s, %d failures" % (n, cnt[0], cnt[1], time.time() - t0, len(bad)))
for b in bad[:10]:
    print(b)
