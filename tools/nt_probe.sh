#!/bin/bash
# Round 4 A/B (GPU box): streaming (nt) stores / loads on the path's planes -- speed and the HBM-side counters.
#   tools/nt_probe.sh > gpurun_out/nt_probe.txt      (variants built by tools/build_variants.sh: base nt_out nt_all nt_all2)
R=$PWD
export TMPDIR=/tmp
for n in base nt_all2 nt_encld base nt_all2; do
  lib=$R/tools/bin/variants/lib_$n.so
  [ -f $lib ] || continue
  v=$(NTSCSIM_LIB=$lib timeout 120 python bench.py --cpu-fields 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); k=d['roofline']['kernel_ms_all']; print('value %.0f sustained %.0f  enc %.4f dec %.4f ms' % (d['value'], d['value_sustained'], k['encode'], k['decode']))")
  echo "$n: $v"
done
cd /tmp
for n in nt_encld; do
  lib=$R/tools/bin/variants/lib_$n.so
  [ -f $lib ] || continue
  for grp in FETCH_SIZE WRITE_SIZE; do
    O=$R/gpurun_out/ntp_${n}_$grp; mkdir -p $O
    NTSCSIM_LIB=$lib timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -o pmc -- python $R/bench.py --steps 2 --warmup 1 --cpu-fields 0 --inflight 1 --no-extras --sustain-seconds 0 > $O.log 2>&1 < /dev/null
    f=$(find $O -name "*counter_collection.csv" | head -1)
    python - "$f" "$n" "$grp" <<'PY'
import csv, sys, collections
f, n, g = sys.argv[1:4]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == g: acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if "k_decode" in k or "k_encode" in k: print("%-8s %-10s %-42s per launch %10.0f KiB (%d launches)" % (n, g, k, sum(v) / len(v), len(v)))
PY
  done
done
