#!/bin/bash
# round 5, after the ghosting fold (every other kernel's ISA unchanged): the whole GPU suite, smoke, the bench line
# with its side legs (presets.vhs_ghost2 is the one that moved), the driver's command, the ghosting fuzz
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/final_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/final_tests.log 2>&1
tail -4 gpurun_out/final_tests.log
timeout 900 python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-fields 0 --no-extras > gpurun_out/bench_r05_driver_cmd.json 2>> gpurun_out/bench_r05.err
{ echo '$ python tools/fuzz_ghost.py 80000 5000     # the ghosting extension (folded into the encoder / a pass of its own): random taps, switch sets, geometries'
  timeout 600 python tools/fuzz_ghost.py 80000 5000 2>&1 | grep -v amdgpu.ids | tail -8; } > gpurun_out/fuzz_ghost_r05.txt
cat gpurun_out/fuzz_ghost_r05.txt
tail -c 700 gpurun_out/bench_r05.json
