#!/bin/bash
O=gpurun_out/fuzz_r03.txt
{
echo "# One-off parity sweeps on the final build of round 3 (MI355X): HIP output == oracle byte for byte."
echo '$ python tools/fuzz_more.py 30000 6000      # random switch sets / geometries / sources, both tools (tests/test_fuzz_params.py, other seeds)'
timeout 900 python tools/fuzz_more.py 30000 6000 2>&1 | tail -3
echo '$ python tools/fuzz_family.py 1000 6000     # the YUV422P tool'"'"'s -vhs family: random geometry, alignment, switch mix; streamed forms asserted by name'
timeout 900 python tools/fuzz_family.py 1000 6000 2>&1 | tail -3
echo '$ python tools/fuzz_fullsize.py 5000 1500   # 720x486 / 720x480, random switch sets, both tools, two fields each'
timeout 900 python tools/fuzz_fullsize.py 5000 1500 2>&1 | tail -3
echo '$ python tools/fuzz_catv.py 1000 1500      # the BGRA tool'"'"'s pre-emphasis family at full size (k_encode_fast_pre + k_decode_fast_bk), forms listed'
timeout 900 python tools/fuzz_catv.py 1000 1500 2>&1 | grep -v amdgpu.ids | tail -4
echo '$ python tools/fuzz_catv.py 3000 1500 svideo   # the -vhs -vhs-svideo 1 family at full size (k_decode_fast_sv)'
timeout 900 python tools/fuzz_catv.py 3000 1500 svideo 2>&1 | grep -v amdgpu.ids | tail -4
echo '$ python tools/fuzz_raw28.py 5000 1000      # raw-composite decoder: random captures / switch sets / crippled speculation'
timeout 900 python tools/fuzz_raw28.py 5000 1000 2>&1 | tail -3
} > $O 2>&1
cat $O
