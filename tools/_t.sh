R=$PWD
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_submit.py -m gpu -x -q 2>&1 | tail -3
for e in 1 0; do for a in malloc pinned; do echo -n "inplace=$e $a: "; NTSCSIM_RECORDS_INPLACE=$e $R/composite-video-simulator_amd/field_loop -vhs --mode sync --fields 1000 --warmup 50 --alloc $a 2>&1 | cut -c1-90; done; done
for e in 1 0; do for d in 4 16; do echo -n "inplace=$e depth $d: "; NTSCSIM_RECORDS_INPLACE=$e $R/composite-video-simulator_amd/field_loop -vhs --mode submit --depth $d --lag $d --ring $((d*2+2)) --fields 3000 --warmup 200 --alloc pinned 2>&1 | cut -c1-90; done; done
