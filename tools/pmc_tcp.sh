#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters for the bench workload's kernels, one group per pass.
# Usage (GPU box): bash tools/pmc_tcp.sh <outdir-under-gpurun_out> [bench args]
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-pmc_tcp}; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --cpu-fields 0 --inflight 1 --no-extras --sustain-seconds 0 "$@" > $OUT/$tag.log 2>&1 < /dev/null
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
