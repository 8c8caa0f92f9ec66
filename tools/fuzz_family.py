"""Developer tool (GPU box): tests/test_fuzz_params.py::test_variant_streamed_family_on_random_geometry with
OTHER seeds (python tools/fuzz_family.py 1000 3000 -> seeds 1000..3999), and which kernel forms they hit."""
import collections, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "composite-video-simulator_amd"))
import test_fuzz_params as T
s0, n = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad, forms = [], collections.Counter()
for seed in range(s0, s0 + n):
    forms[T.draw422_family(seed)[6] or "unaligned rows (four sweeps)"] += 1
    try:
        T.test_variant_streamed_family_on_random_geometry(seed)
    except Exception as e:
        bad.append((seed, repr(e)[:300]))
print("%d seeds in %.1f s, %d failures; forms: %s" % (n, time.time() - t0, len(bad), dict(forms)))
for b in bad[:10]:
    print(b)
