#!/bin/sh
# GPU box: the asynchronous drop-in (ntscsim_submit / ntscsim_wait) against the synchronous one, through the
# reference-shaped loop of host/field_loop.cpp.  Output: gpurun_out/submit_probe.txt
OUT=gpurun_out/submit_probe.txt
mkdir -p gpurun_out
FL=composite-video-simulator_amd/field_loop
{
echo "# byte identity (FNV-1a over every consumed frame, same ring): sync vs submit, bob off/on"
for bob in 0 1; do
  $FL -vhs --mode sync   --fields 300 --warmup 0 --ring 140 --bob $bob --hash 1
  $FL -vhs --mode submit --fields 300 --warmup 0 --ring 140 --bob $bob --hash 1 --depth 32
  $FL -vhs --mode submit --fields 300 --warmup 0 --ring 140 --bob $bob --hash 1 --depth 32 --pin 0
done
echo "# throughput, 720x486 -vhs"
$FL -vhs --mode sync --fields 1500 --warmup 100
for depth in 8 16 32 64 128; do
  $FL -vhs --mode submit --fields 20000 --warmup 2000 --depth $depth
done
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --lanes 1
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --lanes 2
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --lanes 4
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --bob 1
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --rewrite-src 1
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --pin 0
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --pin 0 --bob 1
$FL --mode submit --fields 20000 --warmup 2000 --depth 32
echo "# how far behind the submits the loop waits (default 4 x depth), and more hardware queues for the lanes"
for lag in 64 96 128 256; do
  $FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --lag $lag
done
GPU_MAX_HW_QUEUES=8 $FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32
GPU_MAX_HW_QUEUES=8 $FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --src-stable 1
$FL -vhs --mode submit --fields 20000 --warmup 2000 --depth 32 --src-stable 1
echo "# GPU spans of consecutive launches (NTSCSIM_SUBMIT_TIMING=1)"
NTSCSIM_SUBMIT_TIMING=1 $FL -vhs --mode submit --fields 640 --warmup 640 --depth 32 2>&1 | grep "^launch" | tail -8
echo "# the link (tools/link_probe.hip)"
tools/bin/link_probe
} > $OUT 2>&1
cat $OUT
