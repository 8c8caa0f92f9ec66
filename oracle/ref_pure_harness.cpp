// ref_pure_harness.cpp -- TEST INFRASTRUCTURE ONLY.  Appended (on g++'s stdin) after the reference text
// that build_ref_pure.sh streams: extern "C" entry points of OURS that call the reference's functions
// over arrays.  Nothing here declares a type, global or function the reference's text depends on --
// that is what separates this extract from build_ref.sh's (oracle/README.md, "the stand-in-free pin").
extern "C" {

// class LowpassFilter of ffmpeg_ntsc.cpp:74-106 (tool 0), ffmpeg_to_composite.cpp:99-131 (tool 1),
// ffmpeg_raw28ntsc.cpp:76-108 (tool 2): setFilter(rate, hz), resetFilter(reset), then lowpass() or
// highpass() over in[0..n).
int ref_pure_filter(int tool, double rate, double hz, double reset, int highpass,
                    const double *in, size_t n, double *out, double *alpha)
{
#define RUN(NS)                                                                           \
    {                                                                                     \
        NS::LowpassFilter f;                                                              \
        f.setFilter(rate, hz);                                                            \
        f.resetFilter(reset);                                                             \
        if (alpha) *alpha = f.alpha;                                                      \
        for (size_t i = 0; i < n; i++) out[i] = highpass ? f.highpass(in[i]) : f.lowpass(in[i]); \
        return 0;                                                                         \
    }
    if (tool == 0) RUN(pure_ntsc)
    if (tool == 1) RUN(pure_tocomp)
    if (tool == 2) RUN(pure_raw28)
#undef RUN
    return -1;
}

// RGB_to_YIQ ffmpeg_ntsc.cpp:1375-1383 over n packed (r, g, b) int triples
void ref_pure_rgb_to_yiq(const int32_t *rgb, size_t n, int32_t *yiq)
{
    for (size_t i = 0; i < n; i++) {
        int Y, I, Q;
        pure_ntsc::RGB_to_YIQ(Y, I, Q, rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
        yiq[3 * i] = Y; yiq[3 * i + 1] = I; yiq[3 * i + 2] = Q;
    }
}

// all 2^24 8-bit triples, r slowest: FNV-1a (64 bit) over the (Y, I, Q) int32 stream, and optionally the stream
unsigned long long ref_pure_rgb_to_yiq_cube(int32_t *yiq_or_null)
{
    unsigned long long h = 0xcbf29ce484222325ULL;
    size_t k = 0;
    for (int r = 0; r < 256; r++)
        for (int g = 0; g < 256; g++)
            for (int b = 0; b < 256; b++) {
                int v[3];
                pure_ntsc::RGB_to_YIQ(v[0], v[1], v[2], r, g, b);
                for (int c = 0; c < 3; c++) {
                    uint32_t w = (uint32_t)v[c];
                    for (int s = 0; s < 4; s++) { h ^= (w >> (8 * s)) & 0xff; h *= 0x100000001b3ULL; }
                    if (yiq_or_null) yiq_or_null[k++] = v[c];
                }
            }
    return h;
}

// YIQ_to_RGB ffmpeg_ntsc.cpp:1385-1396 over n (Y, I, Q) int triples
void ref_pure_yiq_to_rgb(const int32_t *yiq, size_t n, int32_t *rgb)
{
    for (size_t i = 0; i < n; i++) {
        int r, g, b;
        pure_ntsc::YIQ_to_RGB(r, g, b, yiq[3 * i], yiq[3 * i + 1], yiq[3 * i + 2]);
        rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
    }
}

// clampu8 ffmpeg_to_composite.cpp:335-342
void ref_pure_clampu8(const int32_t *x, size_t n, int32_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = pure_tocomp::clampu8(x[i]);
}

// black_key ffmpeg_to_composite.cpp:954-972 over n independent (dY,dU,dV,fY,fU,fV) sextets, in place
void ref_pure_black_key(int level, int wchroma, uint8_t *d, uint8_t *f, size_t n)
{
    pure_tocomp::black_key_level_feedback = level;
    for (size_t i = 0; i < n; i++)
        pure_tocomp::black_key(d + 3 * i, d + 3 * i + 1, d + 3 * i + 2, f + 3 * i, f + 3 * i + 1, f + 3 * i + 2,
                               wchroma != 0);
}

// hsync_dc_proc ffmpeg_raw28ntsc.cpp:556-594 over a capture in memory.  The set-up statements are main()'s
// (:866-892: rate, compute_NTSC, delay line, detector filters) RESTATED here -- main() is interleaved with
// libav* calls and cannot be streamed; every function they call, and the per-sample path itself, is the
// reference's text.
void ref_pure_raw28_front(double rate, int mark, const uint8_t *cap, size_t n, uint8_t *h, uint8_t *raw)
{
    using namespace pure_raw28;
    mark_sync = mark != 0;
    hsync_dc_level = 128.0;
    for (size_t i = 0; i < hsync_dc_detect_passes; i++) hsync_dc_detect[i] = LowpassFilter();
    if (rate > 0) sample_rate = rate; else NTSC28MHz();
    compute_NTSC();
    hsync_dc_detect_delay.clear();
    hsync_dc_detect_delay.resize((size_t)((one_scanline_time * 0.075 * 0.75) * 0.5));
    hsync_dc_detect_delay_i = hsync_dc_detect_delay.begin();
    for (size_t i = 0; i < hsync_dc_detect_passes; i++) {
        hsync_dc_detect[i].setFilter(sample_rate, sample_rate / (one_scanline_time * 0.075 * 0.75));
        for (size_t j = 0; j < one_frame_time; j++) hsync_dc_detect[i].lowpass(128);
    }
    for (size_t s = 0; s < n; s++) {
        oneprocsamp v;
        memset(&v, 0, sizeof(v));
        v.raw = cap[s];
        v = hsync_dc_proc(v);
        h[s] = v.hsync_dc_raw;
        raw[s] = v.raw;
    }
}

}
