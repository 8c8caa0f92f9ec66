"""Developer tool (build container, CPU): the oracles against the reference extracts (oracle/_ref) on many
more seeds than the committed tests use -- random switch sets / geometries for composite_layer() and
composite_video_process(), random captures / switch sets for the raw-composite decoder."""
import os, sys, time, random
sys.path.insert(0,'tests'); sys.path.insert(0,'composite-video-simulator_amd')
import numpy as np
import test_fuzz_params as T
import _libs as L
t0=time.time(); bad=[]
for seed in range(100000, 200000):
    for fn in (T.test_oracle_equals_reference_on_random_parameters, T.test_variant_oracle_equals_reference_on_random_parameters):
        try: fn(seed)
        except AssertionError as e: bad.append((fn.__name__, seed))
print("oracle vs reference extract: 100000 seeds x 2 tools in %.0f s, %d failures" % (time.time()-t0, len(bad)), bad[:5])
# raw28 oracle vs extract on random captures
FL=["mark_sync","disable_sync","disable_wp_equ","show_subcarrier","disable_subcarrier","disable_equalization"]
t0=time.time(); bad=[]
for seed in range(5000, 6500):
    r=random.Random(seed)
    cap=L.raw28_capture(r.randrange(2,6), seed, r.choice([0,1,3,6,12]), r.randrange(0,400000))
    cap=np.ascontiguousarray(cap[:cap.size-r.randrange(0,300000)])
    kw={k:1 for k in FL if r.random()<0.25}
    o=L.raw28_oracle_opts(**kw)
    a,la=L.raw28_oracle_run(o,cap); b,lb=L.raw28_ref_run(o,cap,'/tmp/fz_cap.u8')
    if a.shape!=b.shape or not np.array_equal(a,b) or la!=lb: bad.append(seed)
print("raw28 oracle vs reference extract: 1500 captures in %.0f s, %d failures" % (time.time()-t0, len(bad)), bad[:5])
