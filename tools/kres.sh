#!/bin/sh
# Register / scratch / LDS use of every kernel in a built object (code-object metadata).
# usage: tools/kres.sh [object ...]   (default: the two product objects with device code)
B=/opt/rocm/lib/llvm/bin
[ $# -eq 0 ] && set -- composite-video-simulator_amd/csrc/ntscsim_hip.o composite-video-simulator_amd/csrc/raw28_decode.o
for f in "$@"; do
  fat=$(mktemp); co=$(mktemp)
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin "$f" "$fat" || exit 1
  $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$fat" --output="$co" || exit 1
  $B/llvm-readelf --notes "$co" | awk '
    /\.agpr_count:/ {a=$NF}
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2}
    /\.group_segment_fixed_size:/ {l=$2} /\.vgpr_spill_count:/ {sp=$2}
    /\.wavefront_size:/ {printf "%s\tvgpr %d agpr %d sgpr %d scratch %d lds %d spill %d\n", name, v, a, s, p, l, sp}' | c++filt | grep -v "^void rocprim::"
  rm -f "$fat" "$co"
done
