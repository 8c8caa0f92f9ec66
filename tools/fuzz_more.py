"""Developer tool (GPU box): the seeded random-parameter parity sweeps of tests/test_fuzz_params.py with
OTHER seeds (python tools/fuzz_more.py 5000 800 -> seeds 5000..5799 for both tools)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import test_fuzz_params as T
s0, n = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = []
for seed in range(s0, s0 + n):
    for fn in (T.test_hip_equals_oracle_on_random_parameters, T.test_variant_hip_equals_oracle_on_random_parameters):
        try:
            fn(seed)
        except AssertionError as e:
            bad.append((fn.__name__, seed, str(e)[:200]))
        except Exception as e:          # parameter sets the product refuses are drawn again by the test itself
            bad.append((fn.__name__, seed, repr(e)[:200]))
print("%d seeds x 2 tools in %.1f s, %d failures" % (n, time.time() - t0, len(bad)))
for b in bad[:10]:
    print(b)
