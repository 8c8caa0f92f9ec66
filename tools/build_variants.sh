#!/bin/sh
# Developer tool: A/B builds of libntscsim.so (same ABI, different -D switches) into tools/bin/variants/.
#   tools/build_variants.sh name1 "-DFOO -DBAR=2" name2 "..." ...
# Run one with NTSCSIM_LIB=tools/bin/variants/lib_<name>.so python bench.py ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/composite-video-simulator_amd/csrc
OUT=$ROOT/tools/bin/variants
mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -I$ROOT/include -I$SRC ${SCHED--mllvm -amdgpu-sched-strategy=iterative-maxocc}"   # SCHED="" = the compiler's default strategy
[ -f "$SRC/params.o" ] && [ -f "$SRC/raw28_decode.o" ] || (cd "$SRC" && make -s params.o glibc_rand.o raw28_decode.o)
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  (
    /opt/rocm/bin/hipcc $FLAGS $defs --offload-arch=gfx950 -c "$SRC/ntscsim_hip.hip" -o "$OUT/$name.o" \
        -Rpass-analysis=kernel-resource-usage 2> "$OUT/$name.log" || { tail -20 "$OUT/$name.log"; exit 1; }
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$SRC/params.o" "$SRC/glibc_rand.o" "$SRC/raw28_decode.o" "$OUT/$name.o" -o "$OUT/lib_$name.so"
    rm -f "$OUT/$name.o"
    echo "built $name: $(grep -A8 'k_decode_fastILb1Ed' "$OUT/$name.log" | grep -E 'VGPRs:|Spill|Occupancy|LDS|Scratch' | sed 's/.*remark: *//' | tr '\n' ' ')"
  ) &
done
wait
