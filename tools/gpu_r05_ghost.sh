#!/bin/bash
# round 5: the ghosting extension folded into the encoder -- tests + A/B rates
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ghost" > gpurun_out/ghost_tests.log 2>&1; tail -5 gpurun_out/ghost_tests.log
timeout 200 python tools/ghost_probe.py > gpurun_out/ghost_probe.txt 2>&1
echo "--- NTSCSIM_DEBUG_DECODE=8 (k_ghost for every delay)" >> gpurun_out/ghost_probe.txt
NTSCSIM_DEBUG_DECODE=8 timeout 200 python tools/ghost_probe.py >> gpurun_out/ghost_probe.txt 2>&1
cat gpurun_out/ghost_probe.txt
