set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variant422.py tests/test_tocomp_cli.py tests/test_host422.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/t4.log
timeout 600 python bench.py --tool to_composite --steps 20 --warmup 5 --cpu-fields 0 > gpurun_out/bench422.json 2> gpurun_out/bench422.err
tail -3 gpurun_out/bench422.err >> gpurun_out/t4.log
cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/composite-video-simulator_amd
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_sync -o sync -- $P/field_loop -vhs --mode sync --fields 300 --warmup 50 > /tmp/sync_prof.log 2>&1
for f in $(find /tmp/prof_sync -name "*stats*.csv"); do echo "== $f"; head -12 $f; done > $GRAFT_REPO_ROOT/gpurun_out/sync_prof.log 2>&1
