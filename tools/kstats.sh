#!/bin/bash
# Developer tool (GPU box): rocprofv3 --kernel-trace --stats of bench.py, kernel rows only.
#   tools/kstats.sh <outdir under gpurun_out> [bench.py args...]
R=$PWD
OUT=$R/gpurun_out/${1:-kstats}; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ks -- python "$R/bench.py" --cpu-fields 0 "$@" > "$OUT/bench.log" 2>&1 < /dev/null
cd "$R"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats.csv"; python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
else echo "no kernel_stats.csv"; tail -5 "$OUT/bench.log"; fi
