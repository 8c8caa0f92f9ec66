#!/bin/bash
# round 5: the whole profile refresh on ONE box (boxes differ by up to 4 %): counters + stats + probes + bench legs,
# profiles/traffic.json assembled on the box from those counters, then the bench legs again (bench.py reads it)
cd /root/repo
mkdir -p gpurun_out
tag=r05
timeout 1200 bash tools/refresh_profiles.sh gpu $tag > gpurun_out/refresh_$tag.log 2>&1
timeout 600 bash tools/refresh_profiles.sh local $tag > gpurun_out/refresh_${tag}_local.log 2>&1
cp profiles/traffic.json gpurun_out/traffic_${tag}_box.json
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
timeout 200 python bench.py --mode fast32 --cpu-fields 0 --no-extras > gpurun_out/bench_${tag}_fast32.json 2>> gpurun_out/bench_$tag.err
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-fields 0 --no-extras > gpurun_out/bench_${tag}_driver_cmd.json 2>> gpurun_out/bench_$tag.err
timeout 600 python bench.py --tool to_composite --cpu-fields 200 > gpurun_out/bench_${tag}_tocomp.json 2>> gpurun_out/bench_$tag.err
python - <<'PY'
import json
for f in ("bench_r05", "bench_r05_driver_cmd", "bench_r05_fast32", "bench_r05_tocomp"):
    d = json.load(open("gpurun_out/%s.json" % f)); print(f, round(d["value"]), round(d.get("value_sustained") or 0))
PY
