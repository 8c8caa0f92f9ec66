R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined" 2>&1 | tail -2
NTSCSIM_PIPE_TIMING=1 $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 150 --warmup 10 2>&1 | grep "pipe_timing wg" | head -20
for i in 1 2; do $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 600 --warmup 50 2>&1 | cut -c1-90; done
