#!/bin/sh
# build_ref.sh -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.
#
# Compiles the reference's OWN per-field DSP text (line ranges of ffmpeg_ntsc.cpp, SURVEY.md
# Appendix C) into oracle/_ref/libntsc_ref.so so that oracle/ntsc_oracle.c can be pinned against
# it bit-for-bit.  The reference text is streamed from /root/reference straight into g++'s stdin:
# no reference source is ever written into this repo (oracle/_ref/ holds the .so only and is
# git-ignored).  -ffp-contract=off + baseline x86-64 = the reference's default build semantics
# (plain `g++ -O2` via automake has no FMA on x86-64).
#
# Ranges: 72-106 LowpassFilter | 205-214 output/phase globals | 756-809 L1 globals + VHS enum |
#         1375-1921 RGB_to_YIQ ... composite_layer
set -e
here=$(cd "$(dirname "$0")" && pwd)
ref=${NTSC_REFERENCE_DIR:-/root/reference}
src="$ref/ffmpeg_ntsc.cpp"
[ -f "$src" ] || { echo "build_ref.sh: $src not present (GPU box?) -- skipping" >&2; exit 0; }
mkdir -p "$here/_ref"
{
    cat "$here/ref_shim_pre.hpp"
    sed -n '72,106p' "$src"
    sed -n '205,214p' "$src"
    sed -n '756,809p' "$src"
    sed -n '1375,1921p' "$src"
    cat "$here/ref_shim_post.cpp"
} | g++ -x c++ -O2 -w -ffp-contract=off -fPIC -shared -I"$here/../include" - -o "$here/_ref/libntsc_ref.so"
echo "built $here/_ref/libntsc_ref.so"
