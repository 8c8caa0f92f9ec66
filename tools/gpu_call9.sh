#!/bin/bash
O=gpurun_out/c9; mkdir -p $O
timeout 600 python -m pytest tests/test_variant422.py tests/test_fuzz_params.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python bench.py --tool to_composite --cpu-fields 0 > $O/bench_tocomp.json 2> $O/bench_tocomp.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/c9/bench_tocomp.json")); print(round(d["value"]), round(d.get("value_sustained",0)), d["roofline"]["kernel_ms_all"])
PY
timeout 500 bash tools/pmc422.sh c9/pmc422 2>&1 | grep "FETCH\|WRITE_SIZE\|INSTS_VALU"
