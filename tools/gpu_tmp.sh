cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_variant422.py tests/test_tocomp_cli.py tests/test_host422.py tests/test_fuzz_params.py -q -m gpu -x 2>&1 | tail -10 > gpurun_out/t_sv.log
timeout 600 python tools/fuzz_short422.py 60000 3000 2>&1 | grep -v amdgpu.ids | tail -6 >> gpurun_out/t_sv.log
timeout 600 python bench.py --tool to_composite --steps 20 --warmup 5 --cpu-fields 0 > gpurun_out/bench422_sv.json 2> gpurun_out/bench422_sv.err
