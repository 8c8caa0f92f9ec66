set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host422.py tests/test_submit.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/t1.log
timeout 300 python -m pytest tests/test_variant422.py tests/test_tocomp_cli.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/t1.log
P=composite-video-simulator_amd
{
echo "# BGRA field_loop -vhs 720x486 depth 32: direct delivery vs ring + k_deliver"
for d in 1 0; do
  NTSCSIM_SUBMIT_DIRECT=$d $P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 32
  NTSCSIM_SUBMIT_DIRECT=$d $P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --rewrite-src 1
  NTSCSIM_SUBMIT_DIRECT=$d $P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --src-stable 1
done
$P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --bob 1
$P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --lanes 1
$P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --lanes 2
$P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 64
$P/field_loop -vhs --mode submit --fields 20000 --warmup 2000 --depth 128
$P/field_loop -vhs --mode sync --fields 1500 --warmup 200
NTSCSIM_SUBMIT_TIMING=1 $P/field_loop -vhs --mode submit --fields 1200 --warmup 1000 --depth 32 2>&1 | tail -12
echo "# 422 field_loop422 -vhs 720x480"
$P/field_loop422 -vhs --mode submit --fields 6000 --warmup 600 --depth 32
$P/field_loop422 -vhs --mode submit --fields 6000 --warmup 600 --depth 64
$P/field_loop422 --mode submit --fields 6000 --warmup 600 --depth 32
$P/field_loop422 -vhs --mode sync --fields 600 --warmup 100
$P/field_loop422 -vhs -width 704 --mode submit --fields 600 --warmup 100
} > gpurun_out/loop1.log 2>&1
