// ref_tocomp_pre.hpp -- TEST INFRASTRUCTURE ONLY.  Prepended (on g++'s stdin) to the line ranges
// of /root/reference/ffmpeg_to_composite.cpp that hold the 8-bit per-field DSP (build_ref.sh).
// The DSP reads data/linesize/width/height/format/interlaced_frame/top_field_first of AVFrame and
// ticks_per_frame of the decoder context (:1035); these PODs are all it needs to compile.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <algorithm>
enum AVPixelFormat { AV_PIX_FMT_YUV420P = 0, AV_PIX_FMT_YUV422P = 4 };
struct AVFrame {
    uint8_t *data[8];
    int linesize[8];
    int width, height;
    int format;
    int interlaced_frame, top_field_first;
};
struct AVRational { int num, den; };
struct AVCodecContext { int ticks_per_frame; };
static AVCodecContext ref_codec_ctx = { 2 };
AVCodecContext *input_avstream_video_codec_context = &ref_codec_ctx;
static AVFrame ref_input_frame_desc;
AVFrame *output_avstream_video_input_frame = &ref_input_frame_desc;
