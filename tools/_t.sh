R=$PWD
for depth in 96 128 256; do for mx in 64 $depth; do
echo -n "bgra depth $depth pipe_max $mx: "; NTSCSIM_PIPE_MAX=$mx $R/composite-video-simulator_amd/field_loop -vhs --mode submit --depth $depth --lag $depth --ring $((depth*2+2)) --fields 6000 --warmup 600 --alloc pinned 2>&1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])"
echo -n "422  depth $depth pipe_max $mx: "; NTSCSIM_PIPE_MAX=$mx $R/composite-video-simulator_amd/field_loop422 -vhs --mode submit --depth $depth --lag $((2*depth)) --fields 6000 --warmup 600 --alloc pinned 2>&1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['fields_per_s'])"
done; done
