set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=$GRAFT_REPO_ROOT/composite-video-simulator_amd
{
for pin in 1 0; do
NTSCSIM_SUBMIT422_PIN=$pin $P/field_loop422 -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 32
done
$P/field_loop422 -vhs --mode submit --fields 6000 --warmup 600 --depth 32
} > gpurun_out/loop422_d.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof422 -o p -- $P/field_loop422 -vhs -422 --mode submit --fields 3000 --warmup 300 --depth 32 > /tmp/p422.log 2>&1
for f in $(find /tmp/prof422 -name "*stats*.csv"); do echo "== $f"; head -14 $f; done > $GRAFT_REPO_ROOT/gpurun_out/prof422.log 2>&1
