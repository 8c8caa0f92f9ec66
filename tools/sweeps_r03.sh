#!/bin/bash
{ echo "# NTSCSIM_LIB=tools/bin/variants/lib_times.so (tools/build_variants.sh times -DF422_AB_TIMES) python tools/sweep_times.py <steps in flight>"
  echo "# k422_fused<true,true,4>: sweep A | head switch | (B1, B2: none) | 'B3' = the streamed pass"
  NTSCSIM_LIB=$PWD/tools/bin/variants/lib_times.so timeout 120 python tools/sweep_times.py 4 2>&1 | grep -v amdgpu.ids
  NTSCSIM_LIB=$PWD/tools/bin/variants/lib_times.so timeout 120 python tools/sweep_times.py 1 2>&1 | grep -v amdgpu.ids; } > gpurun_out/variant_sweeps_r03.txt
cat gpurun_out/variant_sweeps_r03.txt
