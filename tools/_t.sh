R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out/sync_two_launch; mkdir -p $O; cd /tmp
NTSCSIM_PIPE=0 NTSCSIM_DEBUG_DECODE=2 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ks -- $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 300 --warmup 50 > $O/probe.log 2>&1 < /dev/null
python - $O/ks_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
