#!/bin/bash
# Developer tool (GPU box): the shader clock while the bench's sustained leg runs (rocm-smi, every 0.5 s).
#   bash tools/clock_probe.sh [seconds]
S=${1:-6}
python bench.py --cpu-fields 0 --no-extras --steps 100 --warmup 40 --sustain-seconds $S > /tmp/clock_bench.json 2>/dev/null &
BP=$!
sleep 4             # torch import + set-up (the sustained leg comes first)
for i in $(seq 1 $((S * 2 + 4))); do
  /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | head -1 | sed 's/.*sclk clock level[^(]*//'
  sleep 0.5
done
wait $BP
python - <<'PY'
import json
d = json.load(open('/tmp/clock_bench.json'))
print("value_sustained %.0f over %.1f s; value %.0f" % (d["value_sustained"], d["sustained"]["seconds"], d["value"]))
PY
