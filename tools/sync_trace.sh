#!/bin/bash
# Developer tool (GPU box): the synchronous one-field call (field_loop --mode sync) under the kernel + memory-copy trace, per
# frame-memory kind.   tools/sync_trace.sh <tag>   ->  gpurun_out/sync_<tag>_<alloc>/{kernel_stats.csv,memory_copy_stats.csv,probe.log}
tag=${1:-x}
R=$PWD
export TMPDIR=/tmp
for alloc in malloc pinned; do
  O=$R/gpurun_out/sync_${tag}_${alloc}; mkdir -p $O
  # untraced rate first
  $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 600 --warmup 50 --alloc $alloc > $O/rate.json 2>&1 < /dev/null
  ( cd /tmp; timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -o ks -- \
      $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 300 --warmup 50 --alloc $alloc > $O/probe.log 2>&1 < /dev/null )
  for k in kernel_stats memory_copy_stats; do f=$(find $O -name "*_${k}.csv" | head -1); [ -n "$f" ] && cp $f $O/${k}.csv; done
  echo "== $alloc"; cat $O/rate.json | tail -1 | cut -c1-200
  python - $O <<'PY'
import csv, sys, os
for k in ("kernel_stats", "memory_copy_stats"):
    p = os.path.join(sys.argv[1], k + ".csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            print("%-60s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
