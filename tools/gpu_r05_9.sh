set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=$GRAFT_REPO_ROOT/composite-video-simulator_amd
timeout 600 python -m pytest tests/test_host422.py -q -m gpu -x -k "cpp or pinned" 2>&1 | tail -12 > gpurun_out/t9.log
{
for pf in 1 0; do
$P/field_loop422 -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 32 --page-frames $pf
$P/field_loop422 -vhs --mode submit --fields 6000 --warmup 600 --depth 32 --page-frames $pf
done
$P/field_loop422 -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 64 --page-frames 1
$P/field_loop422 -422 --mode submit --fields 6000 --warmup 600 --depth 32 --page-frames 1
$P/field_loop422 -vhs -422 --mode sync --fields 600 --warmup 100 --page-frames 1
} > gpurun_out/loop422_e.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof422 -o p -- $P/field_loop422 -vhs -422 --mode submit --fields 3000 --warmup 300 --depth 32 --page-frames 1 > /tmp/p422.log 2>&1
for f in $(find /tmp/prof422 -name "*stats*.csv"); do echo "== $f"; head -14 $f | cut -c1-220; done > $GRAFT_REPO_ROOT/gpurun_out/prof422.log 2>&1
