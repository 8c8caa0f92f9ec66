#!/bin/bash
# Developer tool (GPU box): the synchronous one-field call ntscsim_field() through host/field_loop.cpp (--mode sync), 720x486.
#   tools/sync_trace.sh <tag>   ->  gpurun_out/sync_<tag>.txt (+ gpurun_out/sync_<tag>_<preset>_<alloc>/ rocprofv3 csv)
# Per preset (-vhs: k_field_pipe, five roles; default: k_field_pipe_tv, three roles) and frame memory kind (malloc =
# posix_memalign, the tool unpatched; pinned = ntscsim_host_frame_alloc, the frames of ntscsim_av_frame_get_buffer: written in
# place): the untraced rate, the one-launch chain for comparison (NTSCSIM_PIPE=0), rocprofv3 kernel / copy averages; then the
# roles' clocks inside one call (NTSCSIM_PIPE_TIMING=1) and the asynchronous form at small depths with and without the roles.
tag=${1:-x}
R=$PWD
export TMPDIR=/tmp
FL=$R/composite-video-simulator_amd/field_loop
rate() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f fields/s  %6.1f us per call' % (d['fields_per_s'], 1e6/d['fields_per_s']))"; }
{
for preset in "-vhs" ""; do
  pname=${preset:+vhs}; pname=${pname:-default}
  for alloc in malloc pinned; do
    echo "== preset ${pname}, frames ${alloc}"
    echo -n "   roles (shipped)      : "; $FL $preset --mode sync --fields 1500 --warmup 100 --alloc $alloc 2>/dev/null | rate
    echo -n "   one-launch chain     : "; NTSCSIM_PIPE=0 $FL $preset --mode sync --fields 1500 --warmup 100 --alloc $alloc 2>/dev/null | rate
    O=$R/gpurun_out/sync_${tag}_${pname}_${alloc}; mkdir -p $O
    ( cd /tmp; timeout 120 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O -o ks -- \
        $FL $preset --mode sync --fields 300 --warmup 50 --alloc $alloc > $O/probe.log 2>&1 < /dev/null )
    python3 - $O <<'PY'
import csv, sys, os
for k in ("kernel_stats", "memory_copy_stats"):
    p = os.path.join(sys.argv[1], "ks_" + k + ".csv")
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            print("   %-58s calls %4s avg %7.1f us  min %7.1f  max %7.1f" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  done
done
echo "== the other switch sets on role kernels: fields/s, frames malloc | pinned, roles (one-launch chain)"
for preset in "-vhs -tvstd pal" "-vhs -vhs-svideo 1" "-vhs -comp-phase 90" "-vhs -comp-catv" "-vhs -comp-catv3" "-comp-catv" "-comp-catv3"; do
  line="   [$preset]"
  for alloc in malloc pinned; do
    a=$($FL $preset --mode sync --fields 1500 --warmup 100 --alloc $alloc 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
    b=$(NTSCSIM_PIPE=0 $FL $preset --mode sync --fields 1500 --warmup 100 --alloc $alloc 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
    line="$line  $alloc $a ($b)"
  done
  echo "$line"
done
echo "== the next call's setup kernel launched ahead of time (NTSCSIM_SETUP_AHEAD=0: off), -vhs, pinned frames: BGRA tool | YUV422P tool"
FL4=$R/composite-video-simulator_amd/field_loop422
for v in 1 0; do
  a=$(NTSCSIM_SETUP_AHEAD=$v $FL -vhs --mode sync --fields 3000 --warmup 200 --alloc pinned 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
  b=$(NTSCSIM_SETUP_AHEAD=$v $FL4 -vhs --mode sync --fields 3000 --warmup 200 --alloc pinned 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
  echo "   ahead=$v: $a | $b"
done
echo "== pinned source frames read in place by the encoder (NTSCSIM_FIELD_SRC_DIRECT=0: uploaded), pinned frames: -vhs | default preset"
for v in 1 0; do
  a=$(NTSCSIM_FIELD_SRC_DIRECT=$v $FL -vhs --mode sync --fields 3000 --warmup 200 --alloc pinned 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
  b=$(NTSCSIM_FIELD_SRC_DIRECT=$v $FL --mode sync --fields 3000 --warmup 200 --alloc pinned 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
  echo "   in place=$v: $a | $b"
done
NTSCSIM_SETUP_AHEAD_STATS=1 $FL -vhs --mode sync --fields 500 --warmup 20 2>&1 | grep "setup ahead" | sed 's/^/   BGRA tool: /'
NTSCSIM_SETUP_AHEAD_STATS=1 $FL4 -vhs --mode sync --fields 500 --warmup 20 2>&1 | grep "setup ahead" | sed 's/^/   YUV422P tool: /'
echo "== the roles' clocks inside one call (-vhs; 100 MHz wall clock from the workgroup's start; polling = time spent waiting for a hand-off)"
NTSCSIM_PIPE_TIMING=1 $FL -vhs --mode sync --fields 150 --warmup 10 2>&1 | grep "pipe_timing wg" | head -20
echo "== ntscsim_submit() at small depths, pinned frames, lag = depth: fields/s with the one-launch chain | with the roles"
for depth in 1 2 4 8 16 32 64; do
  a=$(NTSCSIM_PIPE=0 $FL -vhs --mode submit --depth $depth --lag $depth --ring $((depth*2+2)) --fields 3000 --warmup 200 --alloc pinned 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
  b=$($FL -vhs --mode submit --depth $depth --lag $depth --ring $((depth*2+2)) --fields 3000 --warmup 200 --alloc pinned 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])")
  echo "   depth $depth: $a | $b"
done
} > $R/gpurun_out/sync_${tag}.txt 2>&1
cat $R/gpurun_out/sync_${tag}.txt
