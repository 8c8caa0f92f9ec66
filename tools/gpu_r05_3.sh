set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pool.py -q -m gpu 2>&1 | tail -15 > gpurun_out/t3.log
P=composite-video-simulator_amd
$P/rank_bench -vhs --spawn 1 --frames 300 --steps 20 --warmup 5 > gpurun_out/rank_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/prof_sync -o sync -- $GRAFT_REPO_ROOT/$P/field_loop -vhs --mode sync --fields 300 --warmup 50 > $GRAFT_REPO_ROOT/gpurun_out/sync_prof.log 2>&1
ls -R /tmp/prof_sync | head -20 >> $GRAFT_REPO_ROOT/gpurun_out/sync_prof.log
for f in $(find /tmp/prof_sync -name "*stats*.csv"); do echo "== $f"; head -15 $f; done >> $GRAFT_REPO_ROOT/gpurun_out/sync_prof.log 2>&1
