#!/bin/bash
# Developer tool (GPU box): per-kernel times of the raw-composite decoder (tools/raw28_probe.py under rocprofv3).
#   tools/raw28_kstats.sh <tag> [ENV=VALUE ...]
R=$PWD; tag=$1; shift
O=$R/gpurun_out/ks28_$tag; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ks -- python $R/tools/raw28_probe.py > $O/probe.log 2>&1 < /dev/null
cd $R
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "## $tag $*"; tail -1 $O/probe.log | cut -c1-200
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "raw28" in r["Name"] or float(r["Percentage"]) > 1:
        print("%-40s calls %4s avg %9.1f us  min %9.1f  max %9.1f  total %9.1f us" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
