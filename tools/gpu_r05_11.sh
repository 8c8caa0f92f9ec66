set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_variant422.py tests/test_raw28.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/t11.log
