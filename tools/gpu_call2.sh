#!/bin/bash
O=gpurun_out/c2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for a in "" "112 4384" "56 4384" "112 8768"; do echo "== pin $a"; timeout 100 python tools/raw28_probe.py $a 2>&1 | tail -1; done
echo "== nopin"; NTSCSIM_RAW28_NOPIN=1 timeout 100 python tools/raw28_probe.py 2>&1 | tail -1
