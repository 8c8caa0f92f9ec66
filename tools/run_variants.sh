#!/bin/sh
# Developer tool (GPU box): bench every A/B build in tools/bin/variants/, 1 and 3 steps in flight.
#   tools/run_variants.sh outdir [names...]
out=$1; shift
mkdir -p "$out"
names="$@"
[ -n "$names" ] || names=$(ls tools/bin/variants/lib_*.so | sed 's/.*lib_\(.*\)\.so/\1/')
for n in $names; do
  lib=$(pwd)/tools/bin/variants/lib_$n.so
  for q in 1 3; do
    NTSCSIM_LIB=$lib timeout 120 python bench.py --cpu-fields 0 --inflight $q --steps 30 --warmup 5 > "$out/$n.if$q.json" 2> "$out/$n.if$q.err"
  done
  python - "$out" "$n" <<'PY'
import json, sys
o, n = sys.argv[1], sys.argv[2]
try:
    a = json.load(open("%s/%s.if1.json" % (o, n))); b = json.load(open("%s/%s.if3.json" % (o, n)))
    k = a["roofline"]["kernel_ms_all"]
    print("%-22s if1 %.0f f/s  if3 %.0f f/s   setup %.3f enc %.3f dec %.3f ms" % (n, a["value"], b["value"], k["setup"], k["encode"], k["decode"]))
except Exception as e:
    print(n, "FAILED", e)
PY
done
