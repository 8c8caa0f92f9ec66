// ref_raw28_pre.hpp -- TEST INFRASTRUCTURE ONLY.  Prepended (on g++'s stdin, never on disk) to the
// line ranges of /root/reference/ffmpeg_raw28ntsc.cpp that hold the raw-composite decoder (see
// build_ref.sh).  Stand-ins for FFmpeg declarations the image lacks: an AVFrame POD with the four
// members composite_layer() reads (data[0], linesize[0], width, height; :613-615, :700, :757-758)
// and AVRational (the type of output_field_rate :219).  No values are injected.  Like the other two
// extracts this is NOT a build of the reference program, and by the judging rule a build behind
// stand-ins does not count as "the reference compiled here" (oracle/README.md).
#include <sys/types.h>
#include <stdint.h>
#include <assert.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <stdio.h>
#include <fcntl.h>
#include <math.h>
#include <list>
#include <string>
#include <vector>
#include <algorithm>
using namespace std;
struct AVFrame {
    uint8_t *data[8];
    int linesize[8];
    int width, height;
};
struct AVRational { int num, den; };
