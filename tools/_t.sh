R=$PWD
timeout 900 python -m pytest tests/test_host422.py tests/test_variant422.py -m gpu -x -q 2>&1 | tail -3
for e in 1 0; do for a in malloc pinned; do echo -n "inplace=$e $a: "; NTSCSIM_RECORDS_INPLACE=$e $R/composite-video-simulator_amd/field_loop422 -vhs --mode sync --fields 1000 --warmup 50 --alloc $a 2>&1 | cut -c1-95; done; done
echo -n "default preset pinned: "; $R/composite-video-simulator_amd/field_loop422 --mode sync --fields 1000 --warmup 50 --alloc pinned 2>&1 | cut -c1-95
echo -n "svideo pinned: "; $R/composite-video-simulator_amd/field_loop422 -vhs -vhs-svideo 1 --mode sync --fields 1000 --warmup 50 --alloc pinned 2>&1 | cut -c1-95
echo -n "svideo pinned one-wave: "; NTSCSIM_PIPE=0 $R/composite-video-simulator_amd/field_loop422 -vhs -vhs-svideo 1 --mode sync --fields 1000 --warmup 50 --alloc pinned 2>&1 | cut -c1-95
echo -n "ep pal pinned: "; $R/composite-video-simulator_amd/field_loop422 -vhs -vhs-speed ep --mode sync --fields 1000 --warmup 50 --alloc pinned 2>&1 | cut -c1-95
echo -n "ep pal pinned one-wave: "; NTSCSIM_PIPE=0 $R/composite-video-simulator_amd/field_loop422 -vhs -vhs-speed ep --mode sync --fields 1000 --warmup 50 --alloc pinned 2>&1 | cut -c1-95
