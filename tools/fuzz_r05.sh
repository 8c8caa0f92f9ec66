#!/bin/bash
O=gpurun_out/fuzz_r05.txt
{
echo "# One-off parity sweeps on the final build of round 5 (MI355X): HIP output == oracle byte for byte.  (The BGRA kernels, the"
echo "# streamed YUV422P family and the raw-composite decoder are unchanged since round 4: profiles/r04_fuzz_sweep.txt.)"
echo '$ python tools/fuzz_short422.py 60000 3000   # the short forms of the YUV422P tool (no VCR / S-Video out): random switch sets, geometries, paddings; census of the forms that ran'
timeout 1200 python tools/fuzz_short422.py 60000 3000 2>&1 | grep -v amdgpu.ids | tail -6
echo '$ python tools/fuzz_host422.py 70000 3000     # ntscsim_field422 / ntscsim_submit422: random loops against the oracle on byte-identical buffers'
timeout 1200 python tools/fuzz_host422.py 70000 3000 2>&1 | grep -v amdgpu.ids | tail -6
echo '$ python tools/fuzz_submit.py 30000 2000      # ntscsim_submit / ntscsim_wait: random loops (ring deeper than the lag; rings of 1-3 frames shared by the fields in flight) against the oracle'"'"'s loop'
timeout 1200 python tools/fuzz_submit.py 30000 2000 2>&1 | grep -v amdgpu.ids | tail -6
echo '$ python tools/fuzz_ghost.py 80000 5000     # the ghosting extension (folded into the encoder / a pass of its own): random taps, switch sets, geometries'
timeout 600 python tools/fuzz_ghost.py 80000 5000 2>&1 | grep -v amdgpu.ids | tail -8
echo '$ python tools/fuzz_more.py 50000 1500       # random switch sets / geometries / sources, both tools (tests/test_fuzz_params.py, other seeds)'
timeout 900 python tools/fuzz_more.py 50000 1500 2>&1 | grep -v amdgpu.ids | tail -3
echo '$ python tools/fuzz_fullsize.py 9000 400     # 720x486 / 720x480, random switch sets, both tools, two fields each'
timeout 900 python tools/fuzz_fullsize.py 9000 400 2>&1 | grep -v amdgpu.ids | tail -3
} > $O 2>&1
cat $O
