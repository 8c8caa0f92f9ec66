#!/bin/bash
# Collect PMC counters for the bench workload, one counter group per pass (the guide's recipe:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; never combine --pmc with sys/hip traces).
# Usage (GPU box): bash tools/pmc.sh <outdir-under-gpurun_out>
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $R/bench.py --steps 2 --warmup 1 --cpu-fields 0 --inflight 1 --no-extras --sustain-seconds 0 > $OUT/$tag.log 2>&1 < /dev/null
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
