set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host422.py tests/test_bench_contract.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/t6.log
P=composite-video-simulator_amd
{
$P/field_loop422 -vhs --mode submit --fields 6000 --warmup 600 --depth 32
$P/field_loop422 --mode submit --fields 6000 --warmup 600 --depth 32
$P/field_loop422 -vhs -vi -422 --mode submit --fields 6000 --warmup 600 --depth 32
} > gpurun_out/loop422_b.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_a.json 2> gpurun_out/bench_r05_a.err
tail -3 gpurun_out/bench_r05_a.err >> gpurun_out/t6.log
