set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/final_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/final_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-fields 0 --no-extras > gpurun_out/bench_r05_driver_cmd.json 2>> gpurun_out/bench_r05.err
timeout 600 python bench.py --tool to_composite --cpu-fields 200 > gpurun_out/bench_r05_tocomp.json 2>> gpurun_out/bench_r05.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r05_driver_full.json 2>> gpurun_out/bench_r05.err
tail -5 gpurun_out/bench_r05.err >> gpurun_out/final_tests.log
