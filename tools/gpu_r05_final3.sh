#!/bin/bash
# round 5, last build: the GPU suite + smoke, the whole profile refresh on one box, the fuzz sweeps
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/final_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/final_tests.log 2>&1
bash tools/gpu_r05_refresh_onebox.sh
timeout 1500 bash tools/fuzz_r05.sh > /dev/null 2>&1
tail -4 gpurun_out/final_tests.log
grep -c "0 failures" gpurun_out/fuzz_r05.txt; grep "failures" gpurun_out/fuzz_r05.txt
