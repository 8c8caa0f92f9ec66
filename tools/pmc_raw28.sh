#!/bin/bash
# Counters of the raw-composite decoder's front end (k_raw28_lp, k_raw28_follow) on the 600-field capture of
# tools/raw28_probe.py, two passes.  Usage (GPU box): bash tools/pmc_raw28.sh <outfile>
set -u
R=$PWD
OUT=$R/gpurun_out/pmc_raw28; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY FETCH_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $R/tools/raw28_probe.py > $OUT/$tag.log 2>&1 < /dev/null
done
cd $R
python - "$OUT" > "${1:-gpurun_out/raw28_front_pmc.txt}" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/*/pmc_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        for k in ('k_raw28_lp', 'k_raw28_follow'):
            if k in r['Kernel_Name']:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
print("rocprofv3 --pmc (two passes) -- python tools/raw28_probe.py: the two sweeps of the front end on the 600-field capture")
print("(286.4 M samples); per kernel, the launches of the 4 decodes in order (k_raw28_follow: a repair round follows a sweep")
print("when a link did not hold)\n")
for k in ('k_raw28_lp', 'k_raw28_follow'):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-22s %s" % (c, [int(x) for x in v[:8]]))
PY
cat "${1:-gpurun_out/raw28_front_pmc.txt}"
