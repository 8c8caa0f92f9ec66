#!/bin/bash
# Longer sweeps on the final build of round 6 (MI355X) -> gpurun_out/fuzz_long_r06.txt
O=gpurun_out/fuzz_long_r06.txt
{
echo "# Longer sweeps on the FINAL build of round 6 (MI355X): role kernels of every switch set (incl. the pre-emphasis presets), the"
echo "# synchronous calls with the setup kernel launched ahead of time, in-place source planes, k422_deliver, the latency form"
echo '$ python tools/fuzz_pipe.py 600000 20000 catv'
timeout 1200 python tools/fuzz_pipe.py 600000 20000 catv 2>&1 | grep -v amdgpu.ids | head -14
echo '$ python tools/fuzz_pipe.py 700000 10000'
timeout 1200 python tools/fuzz_pipe.py 700000 10000 2>&1 | grep -v amdgpu.ids | head -3
echo '$ python tools/fuzz_host422.py 800000 12000'
timeout 1200 python tools/fuzz_host422.py 800000 12000 2>&1 | grep -v amdgpu.ids | tail -2
echo '$ python tools/fuzz_submit.py 900000 5000'
timeout 1200 python tools/fuzz_submit.py 900000 5000 2>&1 | grep -v amdgpu.ids | tail -2
echo '$ python tools/halo_race_probe.py 1300'
timeout 900 python tools/halo_race_probe.py 1300 2>&1 | grep -v amdgpu.ids | tail -3
echo '$ python tools/concurrency_probe.py'
timeout 600 python tools/concurrency_probe.py 2>&1 | grep -v amdgpu.ids | tail -3
echo '$ python tools/fuzz_more.py 950000 3000; python tools/fuzz_fullsize.py 960000 600'
timeout 900 python tools/fuzz_more.py 950000 3000 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python tools/fuzz_fullsize.py 960000 600 2>&1 | grep -v amdgpu.ids | tail -1
} > $O 2>&1
cat $O
