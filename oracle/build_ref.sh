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
# Ranges: 72-106 LowpassFilter | 205-214 output/phase globals | 223, 226 the output frame ring and its
#         index | 756-809 L1 globals + VHS enum | 1375-1921 RGB_to_YIQ ... composite_layer |
#         2233-2257 the "field deinterlace" (bob) block of main()'s field loop, given a function head
#         and tail of our own (it only touches the frame ring, its index and `current`)
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
    sed -n '223p;226p' "$src"
    sed -n '756,809p' "$src"
    sed -n '1375,1921p' "$src"
    # the bob block of main() :2233-2257 as a function of the loop variable it reads
    echo 'static void ref_field_deinterlace(unsigned long long current)'
    sed -n '2233,2257p' "$src"
    cat "$here/ref_shim_post.cpp"
} | g++ -x c++ -O2 -w -ffp-contract=off -fPIC -shared -I"$here/../include" - -o "$here/_ref/libntsc_ref.so"
echo "built $here/_ref/libntsc_ref.so"

# ---- the 8-bit YUV422P sibling, ffmpeg_to_composite.cpp:
# 97-131 LowpassFilter | 261 use_422_colorspace | 267-333 L1 globals | 335-351 clampu8/clips16 |
# 353-553 chroma low-pass / modulate / demodulate | 629-1129 composite_video_process,
# black_key_feedback, render_field | 1177-1236 output_frame's bob copy loops
src2="$ref/ffmpeg_to_composite.cpp"
{
    cat "$here/ref_tocomp_pre.hpp"
    sed -n '97,131p' "$src2"
    sed -n '261,261p' "$src2"
    sed -n '267,333p' "$src2"
    sed -n '335,351p' "$src2"
    sed -n '353,553p' "$src2"
    sed -n '629,1129p' "$src2"
    # the bob copy loops of output_frame() :1177-1236, given a function head/tail of our own
    echo 'static AVFrame *output_avstream_video_bob_frame;'
    echo 'static void ref_output_bob(AVFrame *frame, unsigned int field) {'
    sed -n '1177,1236p' "$src2"
    echo '}'
    cat "$here/ref_tocomp_post.cpp"
} | g++ -x c++ -O2 -w -ffp-contract=off -fPIC -shared -I"$here/../include" - -o "$here/_ref/libtocomp_ref.so"
echo "built $here/_ref/libtocomp_ref.so"

# ---- the raw-composite decoder, ffmpeg_raw28ntsc.cpp:
# 74-108 LowpassFilter | 207-366 globals, rate/geometry, sample buffer (open/flush/refill/close_src),
# RGBTRIPLET | 383-399 preset_PAL / preset_NTSC | 544-598 hsync_dc_proc | 600-849 composite_layer
src3="$ref/ffmpeg_raw28ntsc.cpp"
{
    cat "$here/ref_raw28_pre.hpp"
    sed -n '74,108p' "$src3"
    sed -n '207,366p' "$src3"
    sed -n '383,399p' "$src3"
    sed -n '544,598p' "$src3"
    sed -n '600,849p' "$src3"
    cat "$here/ref_raw28_post.cpp"
} | g++ -x c++ -O2 -w -ffp-contract=off -fPIC -shared - -o "$here/_ref/libraw28_ref.so"
echo "built $here/_ref/libraw28_ref.so"
