#!/bin/bash
# Dry run of the N > 1 path of bench.py on a box with ONE GPU: two ranks, both on GPU 0.
#   tools/dryrun_two_ranks_one_gpu.sh [nccl|gloo]
# With nccl this asks RCCL for a communicator of two ranks on the same device.  RCCL refuses that ("Duplicate GPU
# detected"); the script then says so and repeats the run with gloo for the exchange (the HIP path per rank, the
# barrier / MAX / all-gather branches and the checksum verification are the same code).  Output:
# gpurun_out/dryrun_two_ranks.txt
backend=${1:-nccl}
out=gpurun_out/dryrun_two_ranks.txt
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $2 \
      bench.py --gpus 2 --steps 10 --warmup 3 --no-extras --cpu-fields 0 --sustain-seconds 0 --dist-backend $1 2> /tmp/dryrun_$1.err | grep '^{' | tail -1
}
{
echo "# two ranks on one GPU, backend $backend"
line=$(run $backend 29811)
if [ -n "$line" ]; then
  echo "$line" | python -c 'import json,sys; d=json.load(sys.stdin); print("backend '$backend': OK  n_gpus=%d value=%.0f frames/s scaling=%s checksums_verified=%s" % (d["n_gpus"], d["value"], d["scaling"], d["config"]["rank_checksums_verified"]))'
else
  echo "backend $backend: FAILED -- last lines of stderr:"
  grep -i -m3 "duplicate\|invalid usage\|error" /tmp/dryrun_$backend.err | cut -c1-300
  if [ "$backend" = nccl ]; then
    echo "# RCCL does not accept two ranks on one device; the same run with gloo for the exchange:"
    line=$(run gloo 29812)
    if [ -n "$line" ]; then
      echo "$line" | python -c 'import json,sys; d=json.load(sys.stdin); print("backend gloo: OK  n_gpus=%d value=%.0f frames/s scaling=%s checksums_verified=%s" % (d["n_gpus"], d["value"], d["scaling"], d["config"]["rank_checksums_verified"]))'
    else
      echo "backend gloo: FAILED"; tail -5 /tmp/dryrun_gloo.err | cut -c1-300
    fi
  fi
fi
} > $out 2>&1
cat $out
