R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or setup or dropin" 2>&1 | tail -4
NTSCSIM_PIPE_TIMING=1 $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 150 --warmup 10 2>&1 | grep "pipe_timing wg [01]" | head -10
tools/sync_trace.sh pipe5b
