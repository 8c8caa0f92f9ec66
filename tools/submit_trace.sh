#!/bin/sh
# GPU box: kernel timeline of the submit engine (are launches of different lanes overlapping?)
# usage: tools/submit_trace.sh [extra field_loop args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/subtrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/subtrace -o t -- $R/composite-video-simulator_amd/field_loop -vhs --mode submit --fields 1280 --warmup 640 --depth 32 "$@" > $R/gpurun_out/subtrace/run.txt 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/subtrace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one line per batch: from its k_field_setup to its last k_deliver, per queue
t0 = int(rows[0]["Start_Timestamp"])
cur = {}
out = []
for r in rows:
    q = r["Queue_Id"]; name = r["Kernel_Name"]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if "k_field_setup" in name:
        if q in cur: out.append(cur[q])
        cur[q] = [q, s, e, 0.0, 0]
    elif q in cur:
        cur[q][2] = e
        if "k_deliver" in name: cur[q][3] += e - s; cur[q][4] += 1
out += list(cur.values())
out.sort(key=lambda b: b[1])
lines = ["queue  start_us    end_us   busy_us  deliver_us(n)  gap_to_prev_end"]
prev_end = 0
for b in out[20:60]:
    lines.append("q=%s %10.1f %10.1f %8.1f %8.1f(%d) %8.1f" % (b[0], b[1], b[2], b[2] - b[1], b[3], b[4], b[1] - prev_end))
    prev_end = b[2]
open("gpurun_out/subtrace/timeline.txt", "w").write("\n".join(lines))
print("\n".join(lines))
PY
rm -f gpurun_out/subtrace/*/*.csv gpurun_out/subtrace/*.csv 2>/dev/null
