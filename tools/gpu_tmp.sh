cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_host422.py -q -m gpu 2>&1 | tail -6 > gpurun_out/t_mm.log
