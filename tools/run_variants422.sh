#!/bin/sh
# Developer tool (GPU box): time the YUV422P variant with every A/B build in tools/bin/variants/.
for lib in tools/bin/variants/lib_*.so; do
  for q in 1 4; do
    NTSCSIM_LIB=$(pwd)/$lib timeout 120 python bench.py --cpu-fields 0 --steps 3 --sustain-seconds 0 --inflight $q 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['variant422']
print('%-14s inflight $q: %.3f ms/step  %.0f frames/s' % ('$lib'.split('lib_')[-1][:-3], v['ms_per_step'], v['value']))"
  done
done
