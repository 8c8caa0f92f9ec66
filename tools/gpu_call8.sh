#!/bin/bash
O=gpurun_out/c8; mkdir -p $O
NTSCSIM_LIB=$PWD/tools/bin/variants/lib_times.so timeout 120 python tools/sweep_times.py 4 2>&1 | tail -7
NTSCSIM_LIB=$PWD/tools/bin/variants/lib_times.so timeout 120 python tools/sweep_times.py 1 2>&1 | tail -7
timeout 500 bash tools/pmc422.sh c8/pmc422 2>&1 | tail -22
