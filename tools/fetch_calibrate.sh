#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of tools/bin/fetch_probe's kernels against their known byte counts.
# Output: gpurun_out/fetch_calibration.txt
R=$PWD
OUT=$R/gpurun_out/fetchcal
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o p -- $R/tools/bin/fetch_probe > $OUT/$c.log 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/fetch_calibration.txt
import csv, glob, collections, re
known = {}
for l in open("gpurun_out/fetchcal/FETCH_SIZE.log"):
    m = re.match(r"known_bytes (\S+) (\d+)", l)
    if m: known[m.group(1)] = int(m.group(2))
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) per dispatch of tools/fetch_probe.hip against the bytes each kernel is known to move")
print("# (buffer 1.26 GB >> 256 MiB Infinity Cache; median of 3 dispatches)")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/fetchcal/%s/**/*counter_collection.csv" % c, recursive=True)
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        if k not in known: continue
        v.sort(); med = v[len(v) // 2] * 1024
        rd = k.startswith("k_read")
        if (c == "FETCH_SIZE") == rd:
            print("%-18s %-10s counter %14.0f B   known %14d B   known / counter = %.3f" % (k, c, med, known[k], known[k] / med if med else 0))
PY
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
