R=$PWD
timeout 900 python -m pytest tests/test_host422.py tests/test_variant422.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/fuzz_host422.py 220000 1500 2>&1 | tail -2
for pp in 1 0; do for a in malloc pinned; do echo -n "default preset pipe=$pp $a: "; NTSCSIM_PIPE=$pp $R/composite-video-simulator_amd/field_loop422 --mode sync --fields 1000 --warmup 50 --alloc $a 2>&1 | cut -c1-95; done; done
for pp in 1 0; do echo -n "default preset pipe=$pp submit depth 8: "; NTSCSIM_PIPE=$pp $R/composite-video-simulator_amd/field_loop422 --mode submit --depth 8 --lag 16 --fields 3000 --warmup 200 --alloc pinned 2>&1 | cut -c1-95; done
