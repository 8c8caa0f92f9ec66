#!/bin/bash
# Developer tool (multi-GPU node): the scaling curve of BASELINE's metric, one process per GPU over RCCL.
#   tools/run_multi_gpu.sh [max_gpus] [extra bench.py args...]     e.g.  tools/run_multi_gpu.sh 8 --scaling strong
# Prints one line per N in {1, 2, 4, 8} <= max_gpus: frames/s, ms per step, per-GPU rate, the ratio to
# N x the 1-GPU rate, and whether rank 0 could reproduce every rank's checksum (rank_checksums_verified).
# BASELINE configs[3] (8 independent streams): tools/run_multi_gpu.sh 8 --streams 8
max=${1:-8}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
base=""
for n in 1 2 4 8; do
  [ "$n" -le "$max" ] || break
  port=$((29700 + n))
  if [ "$n" -eq 1 ]; then
    line=$(python bench.py --gpus 1 --no-extras --cpu-fields 0 --force-dist "$@" 2> /tmp/run_multi_gpu_$n.err | grep '^{' | tail -1)
  else
    line=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" \
           bench.py --gpus "$n" --no-extras --cpu-fields 0 "$@" 2> /tmp/run_multi_gpu_$n.err | grep '^{' | tail -1)
  fi
  if [ -z "$line" ]; then echo "N=$n FAILED (see /tmp/run_multi_gpu_$n.err)"; tail -3 /tmp/run_multi_gpu_$n.err; continue; fi
  base=$(python - "$n" "$base" "$line" <<'PY'
import json, sys
n, base, d = int(sys.argv[1]), sys.argv[2], json.loads(sys.argv[3])
v = d["value"]
b = float(base) if base else v
sys.stderr.write("N=%d  %10.0f frames/s  %.3f ms/step  %9.0f per GPU  x%.2f of N x (1-GPU rate)  scaling=%s  checksums_verified=%s\n"
                 % (n, v, d["ms_per_step"], v / n, v / (n * b), d["scaling"], d["config"]["rank_checksums_verified"]))
print(b)
PY
)
done
# ... and the same deal with the C++ host (host/rank_bench.cpp: rccl.h directly, no torch): one JSON line per N and leg --
# BASELINE configs[1] (the clip frame-round-robin), configs[3] (8 independent streams) and configs[4] (3840x2160, 600 frames)
rb=composite-video-simulator_amd/rank_bench
for n in 1 2 4 8; do
  [ "$n" -le "$max" ] || break
  for leg in "--frames 300" "--streams 8 --frames 300" "--size 3840x2160 --frames 64 --steps 6 --warmup 2"; do
    echo "rank_bench N=$n $leg" >&2
    $rb -vhs --spawn "$n" --steps 20 --warmup 5 $leg 2> /tmp/rank_bench_$n.err | grep '^{' | tail -1 \
      || { echo "rank_bench N=$n $leg FAILED (see /tmp/rank_bench_$n.err)"; tail -3 /tmp/rank_bench_$n.err; }
  done
done
