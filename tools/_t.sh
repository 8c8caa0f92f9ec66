FL=composite-video-simulator_amd/field_loop422
rate() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%8.1f fields/s' % (d['fields_per_s']))"; }
{
for preset in "-vhs" "" "-vhs -422"; do
 for alloc in malloc pinned; do
  for rep in 1 2; do
  echo -n "[$preset] $alloc new: "; $FL $preset --mode sync --fields 3000 --warmup 200 --alloc $alloc 2>/dev/null | rate
  echo -n "[$preset] $alloc no src direct: "; NTSCSIM_FIELD422_SRCDIRECT=0 $FL $preset --mode sync --fields 3000 --warmup 200 --alloc $alloc 2>/dev/null | rate
  echo -n "[$preset] $alloc two delivery launches: "; NTSCSIM_SUBMIT422_DELIVER1=0 $FL $preset --mode sync --fields 3000 --warmup 200 --alloc $alloc 2>/dev/null | rate
  done
 done
done
for d in 4 16 32; do
  echo -n "submit depth $d new: "; $FL -vhs --mode submit --fields 12000 --warmup 600 --depth $d 2>/dev/null | rate
  echo -n "submit depth $d two delivery launches: "; NTSCSIM_SUBMIT422_DELIVER1=0 $FL -vhs --mode submit --fields 12000 --warmup 600 --depth $d  2>/dev/null | rate
done
python -m pytest tests/test_host422.py tests/test_variant422.py -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_host422.py 220000 1500 2>&1 | tail -3
bash tools/sync422_trace.sh
} > gpurun_out/sync422_early.txt 2>&1
