#!/bin/bash
O=gpurun_out/fuzz_r06.txt
{
echo "# One-off parity sweeps on the final build of round 6 (MI355X).  Exact mode: HIP output == oracle byte for byte; float mode: within +-1 LSB."
echo '$ python tools/fuzz_host422.py 170000 3000    # ntscsim_field422 / ntscsim_submit422 (incl. tight rows >= 128 wide: the chained launches; staged delivery on the copy threads)'
timeout 1500 python tools/fuzz_host422.py 170000 3000 2>&1 | grep -v amdgpu.ids | tail -6
echo '$ python tools/fuzz_submit.py 130000 2000     # ntscsim_submit / ntscsim_wait (staged delivery on the copy threads, pin policy 1)'
timeout 1500 python tools/fuzz_submit.py 130000 2000 2>&1 | grep -v amdgpu.ids | tail -6
echo '$ python tools/fuzz_float.py 90000 1500       # NTSCSIM_MODE_FLOAT: random switch sets / geometries, <= 1 LSB, forms census'
timeout 1500 python tools/fuzz_float.py 90000 1500 2>&1 | grep -v amdgpu.ids | tail -8
echo '$ python tools/fuzz_pipe.py 50000 6000      # ntscsim_field(): the five-role workgroup form (k_field_pipe) on random geometries / -vhs switch sets / frame memory kinds'
timeout 900 python tools/fuzz_pipe.py 50000 6000 2>&1 | grep -v amdgpu.ids | head -8
echo '$ python tools/fuzz_pipe.py 70000 4000 catv # the same with the pre-emphasis presets mixed in (k_field_pipe_catv, k_field_pipe_tv_catv)'
timeout 900 python tools/fuzz_pipe.py 70000 4000 catv 2>&1 | grep -v amdgpu.ids | head -10
echo '$ python tools/fuzz_more.py 150000 1000      # random switch sets / geometries / sources, both tools, exact mode'
timeout 900 python tools/fuzz_more.py 150000 1000 2>&1 | grep -v amdgpu.ids | tail -3
echo '$ python tools/fuzz_fullsize.py 19000 300    # 720x486 / 720x480, random switch sets, both tools, two fields each'
timeout 900 python tools/fuzz_fullsize.py 19000 300 2>&1 | grep -v amdgpu.ids | tail -3
echo '$ python tools/fuzz_raw28.py 40000 200       # the raw-composite decoder (own scans instead of hipcub) against its oracle on random captures'
timeout 900 python tools/fuzz_raw28.py 40000 200 2>&1 | grep -v amdgpu.ids | tail -3
} > $O 2>&1
cat $O
