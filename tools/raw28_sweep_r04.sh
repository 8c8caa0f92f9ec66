#!/bin/sh
# Round 4: the raw-composite decoder's second sweep with a cheap warm-up -- exact scanlines behind it, chunks per wavefront.
# (GPU box; results never depend on the switches, tests/test_raw28.py)  usage: tools/raw28_sweep_r04.sh > gpurun_out/raw28_sweep.txt
cd "$(dirname "$0")/.."
run() { echo "# $*"; env "$@" python tools/raw28_probe.py 2>&1 | tail -1 | cut -c1-330; }
run NTSCSIM_RAW28_EXACT=1000 NTSCSIM_RAW28_LANES=64
for l in 64 32 16 12; do run NTSCSIM_RAW28_LANES=$l; done
for e in 12 16 20 24 30 36; do run NTSCSIM_RAW28_EXACT=$e; done
echo "# chunks that are NOT a whole number of scanlines (forced 17472 samples): the walk falls back to exact steps"
python tools/raw28_probe.py 112 17472 2>&1 | tail -1 | cut -c1-330
