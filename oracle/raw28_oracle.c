/*
 * raw28_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md, raw28_oracle.h).
 *
 * Scalar restatement of ffmpeg_raw28ntsc.cpp; every function names the lines it follows.  The
 * sample buffer is kept the way the tool keeps it (one array of len*2048 records, moved down when
 * the read position passes its middle, :277-332), so that reads past the filled part of the buffer
 * near the end of a capture see the same stale records the tool would see.
 */
#include "raw28_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double alpha, prev; } onepole;          /* LowpassFilter :76-106 */

static void onepole_set(onepole *f, double rate, double hz)   /* setFilter :80-88 */
{
    const double timeInterval = 1.0 / rate;
    const double tau = 1 / (hz * 2 * M_PI);
    f->alpha = timeInterval / (tau + timeInterval);
}
static double onepole_lowpass(onepole *f, double sample)      /* lowpass :92-96 */
{
    const double stage1 = sample * f->alpha;
    const double stage2 = f->prev - (f->prev * f->alpha);
    return (f->prev = (stage1 + stage2));
}

typedef struct {            /* oneprocsamp :264-270 */
    uint8_t raw, hsync_dc_raw;
    int16_t luma, chroma, rawluma;
} samp;

struct raw28_oracle {
    raw28_opts o;
    double sample_rate, one_frame_time, one_scanline_time, one_scanline_width, one_scanline_width_err;
    unsigned len;                   /* one_scanline_raw_length */
    int width, height;
    /* front end :550-594 */
    onepole det[3];
    double level;                   /* hsync_dc_level */
    uint8_t *delay; size_t delay_n, delay_i;
    uint8_t thr;                    /* sync_threshhold */
    double blank, white;
    /* source + buffer :209-357 */
    const uint8_t *cap; size_t cap_n, cap_pos;
    int src_open;
    samp *buf; size_t buf_n;        /* input_samples */
    size_t rd, end;                 /* input_samples_read / _end as indices */
    unsigned long long byte_counter;
    /* scratch of composite_layer :258-261 (file-scope arrays in the tool: they keep their
     * contents from one scanline to the next) */
    int int_scanline[4096], int_chroma[4096], int_luma[4096];
    unsigned long long current;
};

static void front_init(raw28_oracle *d)           /* main() :936-946, globals :550-558 */
{
    d->delay_n = (size_t)((d->one_scanline_time * 0.075 * 0.75) * 0.5);
    d->delay = (uint8_t *)calloc(d->delay_n ? d->delay_n : 1, 1);
    d->delay_i = 0;
    for (size_t i = 0; i < 3; i++) {
        d->det[i].prev = 0;
        onepole_set(&d->det[i], d->sample_rate, d->sample_rate / (d->one_scanline_time * 0.075 * 0.75));
        for (size_t j = 0; j < d->one_frame_time; j++) onepole_lowpass(&d->det[i], 128);
    }
    d->level = 128.0;
    d->thr = (uint8_t)(192 * 0.25 * 0.5);
    d->blank = (uint8_t)0;
    d->white = (uint8_t)192;
}

static void hsync_dc_proc(raw28_oracle *d, samp *v)   /* :556-594 */
{
    double lv = v->raw;
    for (size_t i = 0; i < 3; i++) lv = onepole_lowpass(&d->det[i], lv);
    if (d->level > lv) {
        const double a = 1.0 / (d->one_scanline_time * 0.07 * 0.75);
        d->level = (d->level * (1.0 - a)) + (lv * a);
    } else {
        const double a = 1.0 / (d->one_frame_time * 0.6);
        d->level = (d->level * (1.0 - a)) + (lv * a);
    }
    if (d->delay_n) {
        const uint8_t ov = d->delay[d->delay_i];
        d->delay[d->delay_i] = v->raw;
        v->raw = ov;
        if (++d->delay_i >= d->delay_n) d->delay_i = 0;
    }
    {
        int x = (int)(lv - d->level);
        if (x < 0) x = 0;
        if (x > 255) x = 255;
        v->hsync_dc_raw = (uint8_t)x;
    }
    if (d->o.mark_sync && v->hsync_dc_raw < d->thr) v->raw = 255;
}

static void geometry(raw28_oracle *d, const raw28_opts *o)    /* :233-256, :395-402 */
{
    d->sample_rate = o->sample_rate > 0 ? o->sample_rate : ((315000000.00 * 8.0) / 88.00);
    d->one_frame_time = d->sample_rate / (30000.00 / 1001.00);
    d->one_scanline_time = d->one_frame_time / 525.00;
    d->len = (unsigned int)(d->one_scanline_time + 0.5);
    d->one_scanline_width = d->len;
    d->one_scanline_width_err = 0;
    d->height = 262;
    d->width = (int)((d->len + 1) & (~1u));
}

/* ---- buffer :277-357 */
static unsigned long long total_count_src(const raw28_oracle *d) { return d->byte_counter + d->rd; }
static size_t count_src(const raw28_oracle *d) { return d->end - d->rd; }
static void flush_src(raw28_oracle *d)            /* :290-303 */
{
    if (d->rd != 0) {
        d->byte_counter = total_count_src(d);
        const size_t move = d->rd, todo = d->end - d->rd;
        if (todo > 0) memmove(d->buf, d->buf + d->rd, todo * sizeof(samp));
        d->rd -= move;
        d->end -= move;
    }
}
static void refill_src(raw28_oracle *d)           /* :307-327 */
{
    if (!d->src_open) return;
    while (d->end < d->buf_n) {
        size_t todo = d->buf_n - d->end;
        if (todo > 4096) todo = 4096;
        size_t rdn = d->cap_n - d->cap_pos;
        if (rdn > todo) rdn = todo;
        if (rdn > 0) {
            for (size_t x = 0; x < rdn; x++) d->buf[d->end + x].raw = d->cap[d->cap_pos + x];
            d->cap_pos += rdn;
            for (size_t x = 0; x < rdn; x++) hsync_dc_proc(d, &d->buf[d->end + x]);   /* do_filter_new_input :596 */
            d->end += rdn;
        }
        if (rdn == 0) break;
    }
}
static void lazy_flush_src(raw28_oracle *d)       /* :329-332 */
{
    if (d->rd > d->buf_n / 2u) flush_src(d);
    refill_src(d);
}

raw28_oracle *raw28_oracle_open(const raw28_opts *o, const uint8_t *capture, size_t n)
{
    raw28_oracle *d = (raw28_oracle *)calloc(1, sizeof(*d));
    d->o = *o;
    geometry(d, o);
    front_init(d);
    d->cap = capture; d->cap_n = n; d->cap_pos = 0; d->src_open = 1;
    d->buf_n = (size_t)d->len * 2048;             /* open_src :353 */
    /* (slack after the array: the tool's searches can step a little past it at the very end of a
     * capture, which is undefined there; here those records read as zero) */
    d->buf = (samp *)calloc(d->buf_n + 8192, sizeof(samp));
    d->rd = d->end = 0;
    return d;
}
void raw28_oracle_close(raw28_oracle *d)
{
    if (!d) return;
    free(d->delay); free(d->buf); free(d);
}
void raw28_oracle_geometry(const raw28_oracle *d, int *width, int *height, int *scanline_samples)
{
    if (width) *width = d->width;
    if (height) *height = d->height;
    if (scanline_samples) *scanline_samples = (int)d->len;
}
void raw28_oracle_levels(const raw28_oracle *d, double *blank, double *white, uint64_t *read_pos)
{
    if (blank) *blank = d->blank;
    if (white) *white = d->white;
    if (read_pos) *read_pos = total_count_src(d);
}

/* composite_layer() :601-849 */
static void composite_layer(raw28_oracle *d, uint8_t *frame, int linesize)
{
    const unsigned len = d->len;
    const size_t E = d->end;
    samp *const B = d->buf;
    unsigned x, y;

    d->one_scanline_width_err = 0;
    lazy_flush_src(d);
    refill_src(d);

    if (!d->o.disable_sync) {                     /* :622-693 */
        size_t i = d->rd;
        int vsb_count = 0;
        const size_t E2 = d->end;
        while (i < E2) {
            while (i < E2 && B[i].hsync_dc_raw >= d->thr) i++;
            const size_t si = i;
            while (i < E2 && B[i].hsync_dc_raw < d->thr) i++;
            const size_t ei = i;
            const size_t synclen = ei - si;
            if (synclen >= (size_t)(int)(len * 0.3)) {
                i = si + (size_t)(int)(len * 0.3);
                if (i < ei) i = ei;
                vsb_count++;
            } else if (synclen >= (size_t)(int)(len * 0.06)) {
                if (vsb_count >= (3 * 3)) {
                    d->rd = si + (synclen / 2);
                    break;
                }
            } else if (synclen >= (size_t)(int)(len * 0.02)) {
                i = si + (size_t)(int)(len * 0.3);
                if (i < ei) i = ei;
                vsb_count++;
                {                                 /* :661-688 */
                    size_t j = si;
                    int mina = 0, mind = 0, maxa = 0, maxd = 0;
                    while (j < i) {
                        if (B[j].hsync_dc_raw >= d->thr) { maxa += B[j].raw; maxd++; }
                        else { mina += B[j].raw; mind++; }
                        j++;
                    }
                    if (mind > 0) mina /= mind;
                    if (maxd > 0) maxa /= maxd;
                    int t = (int)(maxa + ((maxa - mina) / (0.25 + 0.125)));
                    if (t < maxa + 1) t = maxa + 1;
                    if (t > 240) t = 240;
                    const int nwhite = (uint8_t)t;
                    const int nblack = maxa;
                    const double a = 1.0 / 8.0;
                    d->white = (d->white * (1.0 - a)) + (nwhite * a);
                    d->blank = (d->blank * (1.0 - a)) + (nblack * a);
                }
            }
        }
    }
    (void)E;
    {
        const size_t E3 = d->end;
        size_t scan = d->rd;
        const size_t start = d->rd;
        for (y = 0; y < (unsigned)d->height && (scan + (size_t)len * 2) < E3; y++) {   /* :700 */
            samp *s = B + scan;
            for (x = 0; x < len + 16; x++) { s[x].luma = s[x].raw; s[x].chroma = 0; }
            if (!d->o.disable_equalization) {     /* :706-712 */
                for (x = 0; x < len + 16; x++) {
                    int v = (int)((int)s[x].luma - d->blank);
                    if (!d->o.disable_wp_equ) v = (int)((v * 255) / (d->white - d->blank));
                    s[x].luma = (int16_t)v;
                }
            }
            for (x = 0; x < len + 16; x++) s[x].rawluma = s[x].luma;
            if (!d->o.disable_subcarrier) {       /* :719-755 */
                int *sc = d->int_scanline, *ch = d->int_chroma, *lu = d->int_luma;
                for (x = 0; x < len + 16; x++) sc[x] = s[x].luma;
                for (x = 0; x < len; x++) lu[x] = (sc[x] + sc[x + 4] + 1) / 2;
                for (x = 0; x < len; x++) ch[x] = sc[x] - lu[x];
                for (x = 0; x < len; x++) ch[x] = (ch[x] + ch[x + 8] - ch[x + 4] - ch[x + 12]);
                for (unsigned iter = 0; iter < 4; iter++)
                    for (x = 0; x < len; x++) ch[x] -= (ch[x] + ch[x + 4]) / 2;
                for (x = len - 1; (int)x >= 0; x--) ch[x + 8 + 8] = ch[x] / 4;
                for (x = 0; x < len; x++) lu[x] = sc[x] - ch[x];
                for (x = 0; x < len; x++) { s[x].luma = (int16_t)lu[x]; s[x].chroma = (int16_t)ch[x]; }
            }
            uint8_t *dst = frame + (size_t)linesize * y;                       /* :757-775 */
            for (x = 0; x < (unsigned)d->width; x++) {
                int Y = s[x].luma;
                if (d->o.show_subcarrier) Y = s[x].chroma + 128;
                if (Y < 0) Y = 0;
                if (Y > 255) Y = 255;
                dst[4 * x + 0] = (uint8_t)Y; dst[4 * x + 1] = (uint8_t)Y; dst[4 * x + 2] = (uint8_t)Y; dst[4 * x + 3] = 0;
            }
            {                                     /* :777-787 */
                unsigned adj = (unsigned)floor(d->one_scanline_width);
                d->one_scanline_width_err += d->one_scanline_width - adj;
                if (d->one_scanline_width_err >= 1.0) { d->one_scanline_width_err -= 1.0; adj++; }
                scan += adj;
                if (scan > E3) scan = E3;
            }
            if (!d->o.disable_sync) {             /* :789-830 */
                size_t i = scan;
                int vsb_count = 0;
                if (i > d->rd) {
                    size_t avail = i - d->rd;
                    if (avail >= (len * 0.1)) avail = (size_t)(len * 0.1);
                    i -= avail;
                }
                while (i < E3) {
                    while (i < E3 && B[i].hsync_dc_raw >= d->thr) i++;
                    const size_t si = i;
                    while (i < E3 && B[i].hsync_dc_raw < d->thr) i++;
                    const size_t ei = i;
                    const size_t synclen = ei - si;
                    if (synclen >= (size_t)(int)(len * 0.3)) {
                        i = si + (size_t)(int)(len * 0.3);
                        if (i < ei) i = ei;
                        vsb_count++;
                    } else if (synclen >= (size_t)(int)(len * 0.06)) {
                        scan = si + (synclen / 2);
                        break;
                    } else if (synclen >= (size_t)(int)(len * 0.02)) {
                        i = si + (size_t)(int)(len * 0.3);
                        if (i < ei) i = ei;
                        vsb_count++;
                    }
                    if (vsb_count >= (3 * 3)) { y = INT_MAX; break; }
                }
            }
        }
        if (d->o.disable_sync) d->rd = scan;      /* :833 */
        {                                         /* :836-845 */
            size_t should = start + ((size_t)len * 240);
            if (should > E3) should = E3;
            if (d->rd < should) d->rd = should;
        }
    }
}

int raw28_oracle_next_field(raw28_oracle *d, uint8_t *bgra, int linesize)   /* main() :1006-1038 */
{
    lazy_flush_src(d);
    refill_src(d);
    if (count_src(d) < ((size_t)d->len * 256)) {
        d->src_open = 0;                          /* close_src(); open_src() fails: one input */
        return 0;
    }
    memset(bgra, 0, (size_t)linesize * d->height);
    composite_layer(d, bgra, linesize);
    d->current++;
    return 1;
}

void raw28_oracle_front(const raw28_opts *o, const uint8_t *capture, size_t n, uint8_t *hsync_dc_raw,
                        uint8_t *raw_delayed)
{
    raw28_oracle *d = (raw28_oracle *)calloc(1, sizeof(*d));
    d->o = *o;
    geometry(d, o);
    front_init(d);
    for (size_t s = 0; s < n; s++) {
        samp v;
        memset(&v, 0, sizeof(v));
        v.raw = capture[s];
        hsync_dc_proc(d, &v);
        if (hsync_dc_raw) hsync_dc_raw[s] = v.hsync_dc_raw;
        if (raw_delayed) raw_delayed[s] = v.raw;
    }
    free(d->delay);
    free(d);
}

/* ---- synthetic capture (ours; no counterpart in the reference) ------------------------------ */
static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

size_t raw28_synth_capture(uint8_t *out, size_t cap, int fields, uint32_t seed, int noise_level)
{
    const int H = 1820, HH = 910;                 /* samples per line / half line at 8 x fsc */
    const double tip = 18, blank = 62, white = 205;
    uint32_t rs = seed * 2654435761u + 12345u;
    size_t t = 0;
    for (int f = 0; f < fields && t < cap; f++) {
        /* 525 half lines: 6 equalising, 6 broad (vertical sync), 6 equalising, then picture lines */
        for (int hl = 0; hl < 525 && t < cap;) {
            int kind, n;                          /* kind 0 eq, 1 broad, 2 picture line, 3 half picture line */
            if (hl < 6) { kind = 0; n = HH; }
            else if (hl < 12) { kind = 1; n = HH; }
            else if (hl < 18) { kind = 0; n = HH; }
            else if (hl + 2 <= 525) { kind = 2; n = H; }
            else { kind = 3; n = HH; }
            const int line_no = (hl - 18) / 2;
            for (int x = 0; x < n && t < cap; x++, t++) {
                double v = blank;
                const int sync_w = kind == 0 ? 73 : (kind == 1 ? 783 : 136);
                if (x < sync_w) v = tip;
                else if (kind >= 2) {
                    const double ph = 2.0 * M_PI * (double)(t & 7) / 8.0;
                    if (x >= 152 && x < 152 + 72) v = blank + 20.0 * sin(ph + M_PI);          /* burst */
                    else if (x >= 270 && x < n - 44) {
                        const int ax = x - 270, aw = H - 44 - 270;
                        const int bar = (ax * 8) / aw;
                        const double lum[8] = {0.95, 0.80, 0.65, 0.55, 0.40, 0.30, 0.15, 0.05};
                        const double sat[8] = {0.0, 0.28, 0.35, 0.33, 0.33, 0.35, 0.28, 0.0};
                        double y = lum[bar];
                        if (line_no > 160) y = (double)((ax + 7 * f) % aw) / aw;             /* moving ramp */
                        if (line_no > 220) y = ((ax / 3 + line_no) & 1) ? 0.85 : 0.15;     /* fine detail */
                        v = blank + (white - blank) * y + (white - blank) * sat[bar] * sin(ph + 0.6 * bar);
                    }
                }
                v += 5.0 * sin(2.0 * M_PI * (double)t / (3.3 * 477750.0));                   /* DC wander */
                if (noise_level > 0) v += (double)((int)(lcg(&rs) % (unsigned)(2 * noise_level + 1)) - noise_level);
                int iv = (int)floor(v + 0.5);
                out[t] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
            }
            hl += (kind == 2) ? 2 : 1;
        }
    }
    return t;
}
