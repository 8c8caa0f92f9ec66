set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_host422.py -q -m gpu 2>&1 | tail -30 > gpurun_out/t2.log
timeout 900 python -m pytest tests/test_submit.py -q -m gpu 2>&1 | tail -30 >> gpurun_out/t2.log
