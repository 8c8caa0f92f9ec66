#!/bin/bash
# tools/fp_probe.sh -- GPU box: the float pipeline's variants (NTSCSIM_FP_VARIANT), kernel times and stall counters.
# usage: bash tools/fp_probe.sh <outdir-under-gpurun_out> "<variants>"
set -u
R=$PWD
OUT=$R/gpurun_out/${1:-fp_probe}
mkdir -p $OUT
for v in ${2:-0 1}; do
  echo "== variant $v"
  NTSCSIM_FP_VARIANT=$v python bench.py --mode float --no-extras --cpu-fields 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f sustained %.0f ms/step %.4f kernels %s' % (d['value'], d.get('value_sustained',0), d['ms_per_step'], d['roofline']['kernel_ms_all']))"
done
export TMPDIR=/tmp
cd /tmp
for v in ${3:-0}; do
NTSCSIM_FP_VARIANT=$v timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc$v -o pmc -- python $R/bench.py --mode float --steps 2 --warmup 1 --cpu-fields 0 --inflight 1 --no-extras --sustain-seconds 0 > $OUT/pmc$v.log 2>&1 < /dev/null
python - $OUT/pmc$v <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "fp" not in k and "row_states" not in k: continue
    print(k)
    for c,vals in sorted(v.items()): print("   %-22s n=%d mean=%g" % (c,len(vals),sum(vals)/len(vals)))
PY
done
cd $R
