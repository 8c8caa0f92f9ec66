R=$PWD
timeout 900 python -m pytest tests/test_host422.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/fuzz_host422.py 210000 1000 2>&1 | tail -3
for a in malloc pinned; do for pp in 1 0; do echo -n "pipe=$pp $a: "; NTSCSIM_PIPE=$pp $R/composite-video-simulator_amd/field_loop422 -vhs --mode sync --fields 600 --warmup 50 --alloc $a 2>&1 | cut -c1-95; done; done
for d in 4 16 32; do for pp in 1 0; do echo -n "pipe=$pp depth $d: "; NTSCSIM_PIPE=$pp $R/composite-video-simulator_amd/field_loop422 -vhs --mode submit --depth $d --lag $((2*d)) --fields 3000 --warmup 200 --alloc pinned 2>&1 | cut -c1-95; done; done
