#!/bin/bash
# round-3 GPU call 1: tests, bench lines, kernel names, A/B variants, in-flight sweep
O=gpurun_out/c1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --tool to_composite --cpu-fields 100 > $O/bench_tocomp.json 2> $O/bench_tocomp.err; echo "tocomp rc=$?" >> $O/rc.txt
timeout 250 tools/kstats.sh c1/ks_tocomp --tool to_composite --inflight 1 --steps 20 --sustain-seconds 0 > $O/ks_tocomp.txt 2>&1
for q in 2 3 6 8; do
  timeout 120 python bench.py --cpu-fields 0 --no-extras --inflight $q --steps 60 --warmup 20 > $O/inflight$q.json 2>/dev/null
done
timeout 600 tools/run_variants.sh $O/variants cur randp nosb ilp dflt > $O/variants.txt 2>&1
cat $O/rc.txt; cat $O/variants.txt; cat $O/ks_tocomp.txt | head -12
python - <<'PY'
import json
for q in (2,3,6,8):
    try:
        d=json.load(open("gpurun_out/c1/inflight%d.json"%q)); print("inflight",q,round(d["value"]),round(d["value_sustained"]))
    except Exception as e: print("inflight",q,"failed",e)
for f in ("bench","bench_tocomp"):
    try:
        d=json.load(open("gpurun_out/c1/%s.json"%f)); print(f, round(d["value"]), round(d.get("value_sustained",0)), d["roofline"]["frac"], d["roofline"].get("valu",{}) and {k:d["roofline"]["valu"].get(k) for k in ("path_frac","path_frac_nominal","k_decode_frac_nominal","hbm_frac_ceiling_exact_mode")})
        for k in ("end_to_end","variant422","raw28","presets","sizes"):
            if k in d: print(k, json.dumps(d[k])[:1500])
    except Exception as e: print(f,"failed",e)
PY
