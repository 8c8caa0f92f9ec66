set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/refresh_profiles.sh gpu r05 > gpurun_out/refresh_r05.log 2>&1
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/final_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/final_tests.log 2>&1
timeout 600 bash tools/fuzz_r05.sh > /dev/null 2>&1
