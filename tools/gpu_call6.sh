#!/bin/bash
O=gpurun_out/c6; mkdir -p $O
timeout 600 python -m pytest tests/test_variant422.py tests/test_fuzz_params.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
timeout 300 python bench.py --tool to_composite --cpu-fields 50 > $O/bench_tocomp.json 2> $O/bench_tocomp.err
NTSCSIM_DEBUG_DECODE=4 timeout 300 python bench.py --tool to_composite --cpu-fields 0 > $O/bench_tocomp_4sweep.json 2>> $O/bench_tocomp.err
python - <<'PY'
import json
for f in ("bench_tocomp","bench_tocomp_4sweep"):
    try:
        d=json.load(open("gpurun_out/c6/%s.json"%f)); print(f, round(d["value"]), round(d.get("value_sustained",0)), d["roofline"]["kernel_ms_all"])
    except Exception as e: print(f,"failed",e)
PY
