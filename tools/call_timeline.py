"""Developer tool: a few consecutive calls out of a rocprofv3 --kernel-trace --memory-copy-trace run (csv), as one timeline.
    python tools/call_timeline.py <dir with *_kernel_trace.csv / *_memory_copy_trace.csv> [first event] [events]
Times in us from the first event shown; used by tools/sync422_trace.sh."""
import csv, glob, os, sys
d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else -60
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ev = []
for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for p in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy")))
ev.sort()
ev = ev[first:][:count] if first < 0 else ev[first:first + count]
t0 = ev[0][0]
for a, b, n in ev:
    n = n.replace("ntscsim::", "").replace("(anonymous namespace)::", "").split("(")[0]
    print("%9.1f %9.1f  dur %7.1f  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n[:60]))
