R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or pinned_destination" 2>&1 | tail -5
for a in malloc pinned; do $R/composite-video-simulator_amd/field_loop --mode sync --fields 600 --warmup 50 --alloc $a 2>&1 | cut -c1-90; done
NTSCSIM_PIPE=0 $R/composite-video-simulator_amd/field_loop --mode sync --fields 600 --warmup 50 --alloc malloc 2>&1 | cut -c1-90
timeout 600 python tools/fuzz_pipe.py 3000 1500 2>&1 | tail -14
