set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=composite-video-simulator_amd
{
echo "# sync call alone"
$P/field_loop -vhs --mode sync --fields 1500 --warmup 200
echo "# sync call while another process keeps the GPU busy (clocks up)"
$P/rank_bench -vhs --spawn 1 --frames 300 --steps 4000 --warmup 5 --verify 0 > /tmp/bg.log 2>&1 &
BG=$!
sleep 1.5
$P/field_loop -vhs --mode sync --fields 1500 --warmup 200
$P/field_loop422 -vhs --mode sync --fields 600 --warmup 100
kill $BG; wait $BG
echo "# rank_bench with 4 steps in flight"
$P/rank_bench -vhs --spawn 1 --frames 300 --steps 40 --warmup 8
rocm-smi --showclocks 2>/dev/null | head -20
} > gpurun_out/sync2.log 2>&1
