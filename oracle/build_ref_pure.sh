#!/bin/sh
# build_ref_pure.sh -- TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.
#
# The part of the reference that compiles with NOTHING but libc / libstdc++ headers, built into
# oracle/_ref/libref_pure.so.  Unlike build_ref.sh there is no stand-in here: no declaration of our
# own replaces a header, a type, a library or a tool the image lacks.  What g++ sees is
#   * the reference's own #include lines for libc / STL headers and its `using namespace std;`
#     (streamed from the file, the libav* includes between them left out),
#   * the reference's text for the pieces listed below, each inside a namespace (three tools define a
#     class of the same name),
#   * ref_pure_harness.cpp: extern "C" entry points of OURS that CALL the reference's functions
#     (a test harness, not a stand-in: it declares nothing the reference's text depends on).
# The reference text is streamed from /root/reference into g++'s stdin; nothing of it is written into
# this repo (oracle/_ref/ is git-ignored).
#
#   ffmpeg_ntsc.cpp          72-106   class LowpassFilter (setFilter / resetFilter / lowpass / highpass)
#                            1375-1396 RGB_to_YIQ, YIQ_to_RGB
#   ffmpeg_to_composite.cpp  97-131   class LowpassFilter (the variant's copy)
#                            267-291, 293-333  the L1 globals (every line but :292, the one AVRational)
#                            335-351  clampu8, clips16
#                            954-972  black_key (reads the global black_key_level_feedback)
#   ffmpeg_raw28ntsc.cpp     74-108   class LowpassFilter
#                            207-218, 220-366  globals, rate / scanline geometry (NTSC28MHz, compute_NTSC),
#                                     the sample buffer (every line but :219, the one AVRational)
#                            544-598  hsync_dc_proc, do_filter_new_input: the whole sample front end
set -e
here=$(cd "$(dirname "$0")" && pwd)
ref=${NTSC_REFERENCE_DIR:-/root/reference}
a="$ref/ffmpeg_ntsc.cpp"; b="$ref/ffmpeg_to_composite.cpp"; c="$ref/ffmpeg_raw28ntsc.cpp"
[ -f "$a" ] && [ -f "$b" ] && [ -f "$c" ] || { echo "build_ref_pure.sh: reference not present (GPU box?) -- skipping" >&2; exit 0; }
mkdir -p "$here/_ref"
{
    # the reference's own libc / STL includes (ffmpeg_raw28ntsc.cpp:10-18, :44-49) and :42
    sed -n '10,18p' "$c"
    echo '#include <string.h>'        # memmove/memset: the tools get it through libavutil's headers
    sed -n '44,49p' "$c"
    sed -n '42p' "$c"
    echo 'namespace pure_ntsc {'
    sed -n '72,106p' "$a"
    sed -n '1375,1396p' "$a"
    echo '}'
    echo 'namespace pure_tocomp {'
    sed -n '97,131p' "$b"
    sed -n '267,291p;293,333p' "$b"
    sed -n '335,351p' "$b"
    sed -n '954,972p' "$b"
    echo '}'
    echo 'namespace pure_raw28 {'
    sed -n '74,108p' "$c"
    sed -n '207,218p;220,366p' "$c"
    sed -n '544,598p' "$c"
    echo '}'
    cat "$here/ref_pure_harness.cpp"
} | g++ -x c++ -O2 -w -ffp-contract=off -fPIC -shared - -o "$here/_ref/libref_pure.so"
echo "built $here/_ref/libref_pure.so"
