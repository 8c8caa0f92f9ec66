#!/bin/sh
# GPU box: the YUV422P tool's loop on host frames (ntscsim_field422 / ntscsim_submit422) through host/field_loop422.cpp.
# Output: gpurun_out/host422_loop_probe.txt
OUT=gpurun_out/host422_loop_probe.txt
mkdir -p gpurun_out
FL="timeout 120 composite-video-simulator_amd/field_loop422"
{
echo "# byte identity (FNV-1a over every encoder frame): sync vs submit vs submit through the staging rings vs submit on page-owned planes"
for fl in "-vhs" "-vhs -422" "-vhs -vi" "-vhs -vi -422" "-422 -bkey-feedback 40" "-vhs -width 704"; do
  $FL $fl --mode sync   --fields 120 --warmup 0 --hash 1
  $FL $fl --mode submit --fields 120 --warmup 0 --hash 1 --depth 16
  NTSCSIM_SUBMIT422_PIN=0 $FL $fl --mode submit --fields 120 --warmup 0 --hash 1 --depth 16 --page-frames 1
  $FL $fl --mode submit --fields 120 --warmup 0 --hash 1 --depth 16 --page-frames 1
done
echo "# throughput, 720x480"
$FL -vhs --mode sync --fields 600 --warmup 100
for pf in 0 1; do
  for fl in "-vhs" "-vhs -422" "-vhs -vi -422" "" "-422"; do
    $FL $fl --mode submit --fields 6000 --warmup 600 --depth 32 --page-frames $pf
  done
done
echo "# source snapshot by DMA out of the pinned planes instead of the memcpy (NTSCSIM_SUBMIT422_SRCDMA=1)"
NTSCSIM_SUBMIT422_SRCDMA=1 $FL -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 32
NTSCSIM_SUBMIT422_SRCDMA=1 $FL -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 64
echo "# without mallopt(M_MMAP_THRESHOLD, 64 KiB): planes are heap blocks, staged"
$FL -vhs --mode submit --fields 6000 --warmup 600 --depth 32 --mmap-threshold 0
$FL -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 32 --mmap-threshold 0
$FL -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 8 --page-frames 1
$FL -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 64 --page-frames 1
$FL -vhs -422 --mode submit --fields 6000 --warmup 600 --depth 128 --page-frames 1
$FL -vhs -width 704 --mode submit --fields 600 --warmup 100
$FL -422 -bkey-feedback 40 --mode submit --fields 600 --warmup 100
} > $OUT 2>&1
tail -3 $OUT
