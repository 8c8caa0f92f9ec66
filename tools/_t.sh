R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pinned_destination or pipelined or dropin" 2>&1 | tail -5
for a in malloc pinned; do for i in 1 2; do $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 600 --warmup 50 --alloc $a 2>&1 | cut -c1-90; done; done
NTSCSIM_FIELD_DIRECT=0 $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 600 --warmup 50 --alloc pinned 2>&1 | cut -c1-90
