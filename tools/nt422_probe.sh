#!/bin/bash
# Round 4 A/B (GPU box): streaming stores in the YUV422P variant's burst writer -- speed and HBM-side counters.
R=$PWD; export TMPDIR=/tmp
for n in b422 nt422 b422 nt422; do
  lib=$R/tools/bin/variants/lib_$n.so
  NTSCSIM_LIB=$lib timeout 120 python bench.py --cpu-fields 0 --steps 3 --sustain-seconds 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['variant422']
print('$n: %.4f ms/step  %.0f frames/s' % (v['ms_per_step'], v['value']))"
done
cd /tmp
for n in b422 nt422; do
  lib=$R/tools/bin/variants/lib_$n.so
  for grp in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/nt422_${n}_$grp; mkdir -p $O
    NTSCSIM_LIB=$lib timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -o pmc -- python $R/tools/variant_probe.py > $O.log 2>&1 < /dev/null
    f=$(find $O -name "*counter_collection.csv" | head -1)
    python - "$f" "$n" "$grp" <<'PY'
import csv, sys, collections
f, n, g = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == g: acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k422" in k: print("%-8s %-10s %-42s per launch %10.0f KiB (%d launches)" % (n, g, k, sum(v) / len(v), len(v)))
PY
  done
done
