#!/bin/bash
O=$PWD/gpurun_out/c3; mkdir -p $O; R=$PWD
export TMPDIR=/tmp; cd /tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$tag -o pmc -- python $R/tools/raw28_probe.py > $O/$tag.log 2>&1 < /dev/null
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/c3/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "raw28_front" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print("   %-24s n=%d max=%.6g mean=%.6g" % (c, len(v), max(v), sum(v)/len(v)))
PY
