#!/bin/bash
O=gpurun_out/c5; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "decoder_path" 2>&1 | tail -2
for rep in 1 2 3; do
 for n in cur randp nosb both; do
  NTSCSIM_LIB=$PWD/tools/bin/variants/lib_$n.so timeout 120 python bench.py --cpu-fields 0 --no-extras > $O/$n.$rep.json 2>/dev/null
  python - $O/$n.$rep.json $n $rep <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["roofline"]["kernel_ms_all"]
print("%-6s rep %s value %.0f sustained %.0f  enc %.3f dec %.3f" % (sys.argv[2], sys.argv[3], d["value"], d["value_sustained"], k["encode"], k["decode"]))
PY
 done
done
