#!/bin/bash
# Developer tool (GPU box): ntscsim_field422() through host/field_loop422 (--mode sync) under the kernel / copy trace:
# per-kernel averages and the timeline of the last calls.   tools/sync422_trace.sh [flags] -> gpurun_out/sync422_timeline.txt
R=$PWD; export TMPDIR=/tmp
for alloc in malloc pinned; do
  O=$R/gpurun_out/sync422_tl_$alloc; rm -rf $O; mkdir -p $O
  ( cd /tmp; timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -o ks -- \
      $R/composite-video-simulator_amd/field_loop422 ${1:--vhs} --mode sync --fields 300 --warmup 50 --alloc $alloc > $O/probe.log 2>&1 < /dev/null )
  echo "== frames: $alloc"
  python3 $R/tools/call_timeline.py $O -26 26
done > $R/gpurun_out/sync422_timeline.txt 2>&1
cat $R/gpurun_out/sync422_timeline.txt
