#!/bin/bash
# PMC counters of the YUV422P tool's kernels (one 600-field launch per measured call), one counter
# group per pass like tools/pmc.sh.   Usage (GPU box): bash tools/pmc422.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp; R=$PWD; OUT=$R/gpurun_out/${1:-pmc422}; mkdir -p $OUT; cd /tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $R/tools/variant_probe.py > $OUT/$tag.log 2>&1 < /dev/null
done
cd $R; python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A18 k422_ $OUT/summary.txt
