#!/bin/bash
O=gpurun_out/c12; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "decoder_path or golden" 2>&1 | tail -2
for rep in 1 2; do
for m in 0 2; do
  NTSCSIM_DEBUG_DECODE=$m timeout 120 python bench.py --cpu-fields 0 --no-extras > $O/m$m.$rep.json 2>/dev/null
  python - $O/m$m.$rep.json $m <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d["roofline"]["kernel_ms_all"]
print("mode %s value %.0f sustained %.0f  enc %.3f dec %.3f" % (sys.argv[2], d["value"], d["value_sustained"], k["encode"], k["decode"]))
PY
done; done
for q in 6 8; do NTSCSIM_DEBUG_DECODE=2 timeout 120 python bench.py --cpu-fields 0 --no-extras --inflight $q > $O/m2q$q.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/m2q$q.json')); print('mode 2 inflight $q', round(d['value']), round(d['value_sustained']))"; done
