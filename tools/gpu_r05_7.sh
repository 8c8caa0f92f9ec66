set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host422.py tests/test_submit.py tests/test_variant422.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/t7.log
P=composite-video-simulator_amd
{
for pin in 1 0; do
NTSCSIM_SUBMIT422_PIN=$pin $P/field_loop422 -vhs --mode submit --fields 6000 --warmup 600 --depth 32
NTSCSIM_SUBMIT422_PIN=$pin $P/field_loop422 -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 32
NTSCSIM_SUBMIT422_PIN=$pin $P/field_loop422 -vhs -vi -422 --mode submit --fields 6000 --warmup 600 --depth 32
done
$P/field_loop422 -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 64
$P/field_loop422 -422 --mode submit --fields 6000 --warmup 600 --depth 32
$P/field_loop422 -vhs -422 --mode sync --fields 600 --warmup 100
} > gpurun_out/loop422_c.log 2>&1
