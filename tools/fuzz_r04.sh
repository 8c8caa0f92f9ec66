#!/bin/bash
O=gpurun_out/fuzz_r04.txt
{
echo "# One-off parity sweeps on the final build of round 4 (MI355X): HIP output == oracle byte for byte."
echo '$ python tools/fuzz_more.py 40000 4000      # random switch sets / geometries / sources, both tools (tests/test_fuzz_params.py, other seeds)'
timeout 900 python tools/fuzz_more.py 40000 4000 2>&1 | tail -3
echo '$ python tools/fuzz_fullsize.py 7000 1000   # 720x486 / 720x480, random switch sets, both tools, two fields each'
timeout 900 python tools/fuzz_fullsize.py 7000 1000 2>&1 | tail -3
echo '$ python tools/fuzz_catv.py 5000 1000 phase     # -vhs with scanline phases of either parity at full size (k_encode_fast_xi + k_decode_fast_xi)'
timeout 900 python tools/fuzz_catv.py 5000 1000 phase 2>&1 | grep -v amdgpu.ids | tail -5
echo '$ python tools/fuzz_catv.py 6000 1000 fullout   # -vhs -out-composite-lowpass-lite 0 at full size (k_decode_fast_fo)'
timeout 900 python tools/fuzz_catv.py 6000 1000 fullout 2>&1 | grep -v amdgpu.ids | tail -5
echo '$ python tools/fuzz_catv.py 7000 700           # the pre-emphasis family (k_encode_fast_pre + k_decode_fast_bk)'
timeout 900 python tools/fuzz_catv.py 7000 700 2>&1 | grep -v amdgpu.ids | tail -4
echo '$ python tools/fuzz_catv.py 8000 700 svideo    # the S-Video family (k_decode_fast_sv)'
timeout 900 python tools/fuzz_catv.py 8000 700 svideo 2>&1 | grep -v amdgpu.ids | tail -4
echo '$ python tools/fuzz_raw28.py 9000 600       # raw-composite decoder: random captures / switch sets / streams / speculation settings (exact part of the warm-up, chunks per wavefront)'
timeout 900 python tools/fuzz_raw28.py 9000 600 2>&1 | grep -v amdgpu.ids | tail -3
} > $O 2>&1
cat $O
