#!/bin/sh
# Round 4: the raw-composite decoder's second sweep with a cheap warm-up -- exact scanlines behind it, chunks per wavefront.
# (GPU box; results never depend on the switches, tests/test_raw28.py)  usage: tools/raw28_sweep_r04.sh > gpurun_out/raw28_sweep.txt
cd "$(dirname "$0")/.."
run() { echo "# $*"; env "$@" python tools/raw28_probe.py 2>&1 | tail -1 | cut -c1-330; }
run NTSCSIM_RAW28_EXACT=1000 NTSCSIM_RAW28_LANES=64
run NTSCSIM_RAW28_EXACT=20 NTSCSIM_RAW28_LANES=64
for l in 32 16 12 10; do run NTSCSIM_RAW28_EXACT=20 NTSCSIM_RAW28_LANES=$l; done
for e in 12 16 24 30; do run NTSCSIM_RAW28_EXACT=$e NTSCSIM_RAW28_LANES=16; done
echo "# chunks that are NOT a whole number of scanlines (forced 17472 samples): the walk falls back to exact steps"
python tools/raw28_probe.py 112 17472 2>&1 | tail -1 | cut -c1-330
