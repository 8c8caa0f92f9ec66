set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variant422.py tests/test_tocomp_cli.py tests/test_host422.py -q -m gpu -x 2>&1 | tail -12 > gpurun_out/t10.log
timeout 600 python bench.py --tool to_composite --steps 20 --warmup 5 --cpu-fields 0 > gpurun_out/bench422_b.json 2> gpurun_out/bench422_b.err
tail -3 gpurun_out/bench422_b.err >> gpurun_out/t10.log
NTSCSIM_DEBUG_DECODE=2 timeout 600 python bench.py --tool to_composite --steps 20 --warmup 5 --cpu-fields 0 > gpurun_out/bench422_b_nofast.json 2>> gpurun_out/bench422_b.err
