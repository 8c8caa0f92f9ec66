// ref_shim_pre.hpp -- TEST INFRASTRUCTURE ONLY.  Prepended (on g++'s stdin, never on disk) to the
// line ranges of /root/reference/ffmpeg_ntsc.cpp that hold the per-field DSP (see build_ref.sh).
// The DSP touches exactly six members of AVFrame (ffmpeg_ntsc.cpp:1579-1586,:1599,:1911) and never
// touches InputFile, so these PODs are all it needs to compile.  This is NOT a build of the
// reference program (that needs FFmpeg 3.x libav* headers/libraries, absent here); it is the
// reference's own hot-path function text, compiled where it lies, used to pin oracle/ntsc_oracle.c.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <algorithm>
struct AVFrame {
    uint8_t *data[8];
    int linesize[8];
    int width, height;
    int format;
    int interlaced_frame, top_field_first;
};
struct AVRational { int num, den; };
struct AVFormatContext; struct AVStream; struct AVCodecContext; struct SwsContext;
class InputFile {};
#include <vector>
