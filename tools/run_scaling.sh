#!/bin/sh
# Developer tool (GPU box): kernel times of A/B builds at batch sizes that put exactly 1, 2, 3 decode
# waves on every SIMD (134 / 269 / 404 frames of 720x486).
#   tools/run_scaling.sh names...
for n in "$@"; do
  lib=$(pwd)/tools/bin/variants/lib_$n.so
  for fr in 134 269 404; do
    NTSCSIM_LIB=$lib timeout 120 python bench.py --cpu-fields 0 --inflight 1 --steps 20 --frames $fr > /tmp/s.json 2>/dev/null
    python - "$n" $fr <<'PY'
import json, sys
d = json.load(open("/tmp/s.json")); k = d["roofline"]["kernel_ms_all"]; n = 2 * int(sys.argv[2])
print("%-10s %4d fields (%4d dec waves): setup %.3f enc %.3f dec %.3f ms" % (sys.argv[1], n, (n * 243 + 62) // 63, k["setup"], k["encode"], k["decode"]))
PY
  done
done
