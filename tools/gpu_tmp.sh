cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_host422.py -q -m gpu 2>&1 | tail -4 > gpurun_out/t_mm.log
timeout 400 sh tools/host422_loop_probe.sh > /dev/null 2>&1
grep -A30 "# throughput" gpurun_out/host422_loop_probe.txt | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['mode'],round(d['fields_per_s']),'out',d['out_mode'],'depth',d['depth'],'pf',d['page_frames'],'mm',d['mmap_threshold'],d['host_us_per_call'],d['stats']['dma_uploads'])
    else: print(l)
" >> gpurun_out/t_mm.log
