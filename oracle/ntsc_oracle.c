/*
 * ntsc_oracle.c -- TEST INFRASTRUCTURE ONLY (see ntsc_oracle.h).
 *
 * Scalar CPU restatement of composite_layer() (ffmpeg_ntsc.cpp:1570-1921).  Written from the
 * behavioural spec in SURVEY.md Appendix A; every function cites the reference lines it follows.
 * It keeps the reference's arithmetic contract exactly:
 *   - planes are int32 holding Y/I/Q scaled by 256; only rows of the current field exist here
 *     (row k <-> frame row field+2k; the reference allocates all H rows and never touches the rest);
 *   - every filter is evaluated in IEEE fp64 in the reference's operation order, stored back with
 *     C truncation toward zero;  MUST be compiled with -ffp-contract=off (an FMA changes results);
 *   - integer `/` truncates toward zero, `>>` is an arithmetic shift;
 *   - all randomness comes from one glibc TYPE_3 rand() stream in the reference's call order.
 */
#include "ntsc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ glibc rand() clone ---- */
/* glibc stdlib/random_r.c: __srandom_r (TYPE_3: deg 31, sep 3) + __random_r.  SURVEY App. B. */

void ntsc_oracle_rng_seed(ntsc_oracle_rng *g, uint32_t seed)
{
    uint32_t s[344];
    int32_t word;
    int i;

    if (seed == 0) seed = 1;
    s[0] = seed;
    word = (int32_t)seed;
    for (i = 1; i < 31; i++) {
        /* 16807 * word mod (2^31 - 1) without overflow (Schrage) */
        int32_t hi = word / 127773;
        int32_t lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        s[i] = (uint32_t)word;
    }
    for (i = 31; i < 34; i++) s[i] = s[i - 31];
    for (i = 34; i < 344; i++) s[i] = s[i - 31] + s[i - 3];
    /* ring holds the last 31 words; the next draw is s[344] = s[313] + s[341] */
    for (i = 0; i < 31; i++) g->r[i] = s[313 + i];
    g->i = 0;
    g->count = 0;
}

uint32_t ntsc_oracle_rng_next(ntsc_oracle_rng *g)
{
    int i = g->i;
    int j = i + 28;
    uint32_t v;
    if (j >= 31) j -= 31;
    v = g->r[i] + g->r[j];
    g->r[i] = v;
    g->i = (i + 1 == 31) ? 0 : i + 1;
    g->count++;
    return v >> 1;
}

void ntsc_oracle_rng_discard(ntsc_oracle_rng *g, uint64_t n)
{
    while (n--) (void)ntsc_oracle_rng_next(g);
}

/* ------------------------------------------------------------------------- one-pole IIR ---- */
/* class LowpassFilter, ffmpeg_ntsc.cpp:74-106 */

#define NTSC_RATE ((315000000.00 * 4) / 88) /* every setFilter() call on the video path */

typedef struct { double alpha, prev; } onepole;

static void onepole_set(onepole *f, double rate, double hz, double reset)
{
    /* setFilter :78-86, resetFilter :87-89 */
    double timeInterval = 1.0 / rate;
    double tau = 1 / (hz * 2 * M_PI);
    f->alpha = timeInterval / (tau + timeInterval);
    f->prev = reset;
}

static double onepole_lp(onepole *f, double sample)
{
    /* lowpass :90-94 -- note prev - prev*alpha, not prev*(1-alpha) */
    double stage1 = sample * f->alpha;
    double stage2 = f->prev - (f->prev * f->alpha);
    f->prev = stage1 + stage2;
    return f->prev;
}

static double onepole_hp(onepole *f, double sample)
{
    /* highpass :95-99 */
    double stage1 = sample * f->alpha;
    double stage2 = f->prev - (f->prev * f->alpha);
    f->prev = stage1 + stage2;
    return sample - f->prev;
}

/* three cascaded low-passes, output delayed: the value for input index x lands at x-delay,
 * the last `delay` samples of the row keep their input (composite_lowpass :1429-1458,
 * composite_lowpass_tv :1399-1427, VHS chroma :1814-1836) */
static void row_lp3_delayed(int32_t *P, int W, double cutoff, double reset, int delay)
{
    onepole lp[3];
    int x, f;
    for (f = 0; f < 3; f++) onepole_set(&lp[f], NTSC_RATE, cutoff, reset);
    for (x = 0; x < W; x++) {
        double s = P[x];
        for (f = 0; f < 3; f++) s = onepole_lp(&lp[f], s);
        if (x >= delay) P[x - delay] = (int32_t)s;
    }
}

/* ------------------------------------------------------------------ colour conversion ------ */

static void rgb_to_yiq(int32_t *Y, int32_t *I, int32_t *Q, int r, int g, int b)
{
    /* RGB_to_YIQ :1375-1383 */
    double dY = (0.30 * r) + (0.59 * g) + (0.11 * b);
    *Y = (int32_t)(256 * dY);
    *I = (int32_t)(256 * ((-0.27 * (b - dY)) + (0.74 * (r - dY))));
    *Q = (int32_t)(256 * ((0.41 * (b - dY)) + (0.48 * (r - dY))));
}

static int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static uint32_t yiq_to_rgb_pixel(int32_t Y, int32_t I, int32_t Q)
{
    /* YIQ_to_RGB :1385-1396 and the pack at :1914 (alpha byte = 0) */
    int r = (int)(((1.000 * Y) + (0.956 * I) + (0.621 * Q)) / 256);
    int g = (int)(((1.000 * Y) + (-0.272 * I) + (-0.647 * Q)) / 256);
    int b = (int)(((1.000 * Y) + (-1.106 * I) + (1.703 * Q)) / 256);
    return ((uint32_t)clamp255(r) << 16) + ((uint32_t)clamp255(g) << 8) + (uint32_t)clamp255(b);
}

/* ------------------------------------------------------------- subcarrier mod / demod ------ */

static unsigned scanline_phase(const ntscsim_params *p, unsigned y, uint64_t fieldno)
{
    /* :1473-1480 and again :1529-1536 */
    unsigned off = (unsigned)p->video_scanline_phase_shift_offset;
    if (p->video_scanline_phase_shift == 90)
        return (unsigned)((fieldno + off + (y >> 1)) & 3);
    if (p->video_scanline_phase_shift == 180)
        return (unsigned)((((fieldno + y) & 2) + off) & 3);
    if (p->video_scanline_phase_shift == 270)
        return (unsigned)((fieldno + off - (y >> 1)) & 3);
    return off & 3;
}

static void row_chroma_into_luma(int32_t *Y, int32_t *I, int32_t *Q, int W, unsigned xi, int amp)
{
    /* chroma_into_luma :1460-1495 */
    static const int umult[4] = { 1, 0, -1, 0 };
    static const int vmult[4] = { 0, 1, 0, -1 };
    int x;
    for (x = 0; x < W; x++) {
        unsigned s = (xi + (unsigned)x) & 3;
        int chroma = I[x] * amp * umult[s];
        chroma += Q[x] * amp * vmult[s];
        Y[x] += chroma / 50;
        I[x] = 0;
        Q[x] = 0;
    }
}

static void row_chroma_from_luma(int32_t *Y, int32_t *I, int32_t *Q, int32_t *chroma, int W,
                                 unsigned xi, int amp)
{
    /* chroma_from_luma :1497-1567 */
    int32_t d0 = 0, d1 = 0, d2, d3, sum = 0, c;
    int x;

    /* 4-tap box, pre-charged with the first two samples :1507-1525 */
    d2 = Y[0]; sum += d2;
    d3 = Y[1]; sum += d3;
    for (x = 0; x < W; x++) {
        c = (x + 2 < W) ? Y[x + 2] : 0;
        sum -= d0;
        d0 = d1; d1 = d2; d2 = d3; d3 = c;
        sum += c;
        Y[x] = sum / 4;
        chroma[x] = c - Y[x];
    }

    /* undo the negative half-cycles :1539-1542 */
    for (x = (int)((4 - xi) & 3); x + 3 < W; x += 4) {
        chroma[x + 2] = -chroma[x + 2];
        chroma[x + 3] = -chroma[x + 3];
    }
    /* :1544-1546 */
    for (x = 0; x < W; x++) chroma[x] = (chroma[x] * 50) / amp;

    /* de-interleave :1549-1556 */
    for (x = 0; x + (int)xi + 1 < W; x += 2) {
        I[x] = -chroma[x + xi + 0];
        Q[x] = -chroma[x + xi + 1];
    }
    for (; x < W; x += 2) { I[x] = 0; Q[x] = 0; }
    /* fill odd samples :1557-1564 */
    for (x = 0; x + 2 < W; x += 2) {
        I[x + 1] = (I[x] + I[x + 2]) >> 1;
        Q[x + 1] = (Q[x] + Q[x + 2]) >> 1;
    }
    for (; x < W; x++) { I[x] = 0; Q[x] = 0; }
}

/* ------------------------------------------------------------------------------- taps ------ */

static void tap(int32_t *dst, const int32_t *src, size_t n)
{
    if (dst) memcpy(dst, src, n * sizeof(int32_t));
}

/* ------------------------------------------------------------------------- the field ------- */

int ntsc_oracle_field(const ntscsim_params *p, ntsc_oracle_rng *g,
                      const uint8_t *src, int src_linesize, int src_interlaced, int src_tff,
                      uint8_t *dst, int dst_linesize,
                      int W, int H, unsigned field, uint64_t fieldno,
                      const ntsc_oracle_taps *taps)
{
    static const ntsc_oracle_taps no_taps;
    const int output_ntsc = (p->tv_standard == NTSCSIM_TV_NTSC);
    unsigned opposite;
    int32_t *fY, *fI, *fQ, *chroma;
    int L, k, x;
    size_t n;

    /* guards :1578-1583 */
    if (!src || !dst) return -1;
    if (dst_linesize < W * 4 || src_linesize < W * 4) return -1;
    if (W <= 0 || H <= 0 || field > 1) return -1;
    if (!taps) taps = &no_taps;

    opposite = src_interlaced ? (src_tff ? 1u : 0u) : 0u;       /* :1585-1588 */
    L = ((unsigned)H > field) ? (int)((H - (int)field + 1) / 2) : 0;
    n = (size_t)L * (size_t)W;

    fY = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    fI = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    fQ = (int32_t *)malloc((n + 1) * sizeof(int32_t));
    chroma = (int32_t *)malloc(((size_t)W + 1) * sizeof(int32_t));
    if (!fY || !fI || !fQ || !chroma) { free(fY); free(fI); free(fQ); free(chroma); return -1; }

#define ROWY(k) (fY + (size_t)(k) * W)
#define ROWI(k) (fI + (size_t)(k) * W)
#define ROWQ(k) (fQ + (size_t)(k) * W)
#define FRAME_Y(k) ((unsigned)(field + 2u * (unsigned)(k)))

    /* RGB -> YIQ :1598-1606 */
    for (k = 0; k < L; k++) {
        unsigned y = FRAME_Y(k), sy = y + opposite;
        const uint8_t *srow;
        if (sy > (unsigned)H - 1u) sy = (unsigned)H - 1u;
        srow = src + (size_t)src_linesize * sy;
        for (x = 0; x < W; x++) {
            uint32_t px;
            memcpy(&px, srow + 4 * (size_t)x, 4);
            rgb_to_yiq(&ROWY(k)[x], &ROWI(k)[x], &ROWQ(k)[x],
                       (int)((px >> 16) & 0xFF), (int)((px >> 8) & 0xFF), (int)(px & 0xFF));
        }
    }

    /* input chroma low-pass :1608-1609 -> composite_lowpass :1429 */
    if (p->composite_in_chroma_lowpass) {
        for (k = 0; k < L; k++) row_lp3_delayed(ROWI(k), W, 1300000, 0, 2);
        for (k = 0; k < L; k++) row_lp3_delayed(ROWQ(k), W, 600000, 0, 4);
    }

    /* modulate :1611 */
    for (k = 0; k < L; k++)
        row_chroma_into_luma(ROWY(k), ROWI(k), ROWQ(k), W,
                             scanline_phase(p, FRAME_Y(k), fieldno), p->subcarrier_amplitude);

    /* composite pre-emphasis :1614-1629 */
    if (p->composite_preemphasis != 0 && p->composite_preemphasis_cut > 0) {
        for (k = 0; k < L; k++) {
            int32_t *Y = ROWY(k);
            onepole pre;
            onepole_set(&pre, NTSC_RATE, p->composite_preemphasis_cut, 16);
            for (x = 0; x < W; x++) {
                double s = Y[x];
                s += onepole_hp(&pre, s) * p->composite_preemphasis;
                Y[x] = (int32_t)s;
            }
        }
    }

    /* luma noise :1632-1644; the accumulator carries across rows */
    if (p->video_noise != 0) {
        int noise = 0;
        unsigned noise_mod = (unsigned)((p->video_noise * 2) + 1);
        for (k = 0; k < L; k++) {
            int32_t *Y = ROWY(k);
            for (x = 0; x < W; x++) {
                Y[x] += noise;
                noise += (int)(ntsc_oracle_rng_next(g) % noise_mod) - p->video_noise;
                noise /= 2;
            }
        }
    }
    tap(taps->composite_y, fY, n);

    /* EXTENSION (not in the reference; default off): multipath ghosting, see ntscsim.h */
    if (p->ghost_taps > 0) {
        int32_t *raw = (int32_t *)malloc(((size_t)W + 1) * sizeof(int32_t));
        int gk;
        for (k = 0; k < L && raw; k++) {
            int32_t *Y = ROWY(k);
            memcpy(raw, Y, (size_t)W * sizeof(int32_t));
            for (x = 0; x < W; x++) {
                int acc = 0;
                for (gk = 0; gk < p->ghost_taps; gk++)
                    if (x - p->ghost_delay[gk] >= 0) acc += p->ghost_gain[gk] * raw[x - p->ghost_delay[gk]];
                Y[x] = raw[x] + acc / 256;
            }
        }
        free(raw);
    }

    /* VHS head switching :1647-1713 */
    if (p->vhs_head_switching) {
        unsigned twidth = (unsigned)W + ((unsigned)W / 10u);
        unsigned tx, hx, pp, x2, shy = 0;
        double noise = 0, t;
        int shif, ishif, y;
        int32_t *tmp = (int32_t *)malloc((size_t)twidth * sizeof(int32_t));

        if (p->vhs_head_switching_phase_noise != 0) {
            unsigned u = ntsc_oracle_rng_next(g);
            u *= ntsc_oracle_rng_next(g);
            u *= ntsc_oracle_rng_next(g);
            u *= ntsc_oracle_rng_next(g);
            u %= 2000000000U;
            noise = ((double)u / 1000000000U) - 1.0;
            noise *= p->vhs_head_switching_phase_noise;
        }

        t = output_ntsc ? twidth * 262.5 : twidth * 312.5;

        pp = (unsigned)(fmod(p->vhs_head_switching_point + noise, 1.0) * t);
        y = (int)((pp / twidth) * 2u) + (int)field;
        pp = (unsigned)(fmod(p->vhs_head_switching_phase + noise, 1.0) * t);
        hx = pp % twidth;

        y -= output_ntsc ? (262 - 240) * 2 : (312 - 288) * 2;

        tx = hx;
        ishif = (hx >= twidth / 2) ? (int)(hx - twidth) : (int)hx;

        shif = 0;
        while (y < H) {
            if (y >= 0 && shif != 0 && tmp) {
                /* y has the parity of `field`, so it is one of our rows */
                int32_t *Y = ROWY((y - (int)field) / 2);
                unsigned xx;
                x2 = (tx + twidth + (unsigned)shif) % twidth;
                memset(tmp, 0, (size_t)twidth * sizeof(int32_t));
                memcpy(tmp, Y, (size_t)W * sizeof(int32_t));
                for (xx = tx; xx < (unsigned)W; xx++) {
                    Y[xx] = tmp[x2];
                    if (++x2 == twidth) x2 = 0;
                }
            }
            shif = (shy == 0) ? ishif : (shif * 7) / 8;
            tx = 0;
            y += 2;
            shy++;
        }
        free(tmp);
    }
    tap(taps->headswitch_y, fY, n);

    /* demodulate :1715-1716 */
    if (!p->nocolor_subcarrier)
        for (k = 0; k < L; k++)
            row_chroma_from_luma(ROWY(k), ROWI(k), ROWQ(k), chroma, W,
                                 scanline_phase(p, FRAME_Y(k), fieldno),
                                 p->subcarrier_amplitude_back);
    tap(taps->demod_y, fY, n); tap(taps->demod_i, fI, n); tap(taps->demod_q, fQ, n);

    /* chroma noise :1719-1735 */
    if (p->video_chroma_noise != 0) {
        int noiseU = 0, noiseV = 0;
        unsigned m = (unsigned)((p->video_chroma_noise * 2) + 1);
        for (k = 0; k < L; k++) {
            int32_t *U = ROWI(k), *V = ROWQ(k);
            for (x = 0; x < W; x++) {
                U[x] += noiseU;
                V[x] += noiseV;
                noiseU += (int)(ntsc_oracle_rng_next(g) % m) - p->video_chroma_noise;
                noiseU /= 2;
                noiseV += (int)(ntsc_oracle_rng_next(g) % m) - p->video_chroma_noise;
                noiseV /= 2;
            }
        }
    }
    /* chroma phase noise :1736-1764 */
    if (p->video_chroma_phase_noise != 0) {
        int noise = 0;
        unsigned m = (unsigned)((p->video_chroma_phase_noise * 2) + 1);
        for (k = 0; k < L; k++) {
            int32_t *U = ROWI(k), *V = ROWQ(k);
            double pi, sinpi, cospi;
            noise += (int)(ntsc_oracle_rng_next(g) % m) - p->video_chroma_phase_noise;
            noise /= 2;
            pi = ((double)noise * M_PI) / 100;
            sinpi = sin(pi);
            cospi = cos(pi);
            for (x = 0; x < W; x++) {
                double u = U[x], v = V[x];
                double u_ = (u * cospi) - (v * sinpi);
                double v_ = (u * sinpi) + (v * cospi);
                U[x] = (int32_t)u_;
                V[x] = (int32_t)v_;
            }
        }
    }
    tap(taps->noise_i, fI, n); tap(taps->noise_q, fQ, n);

    /* VHS block :1770-1889 */
    if (p->emulating_vhs) {
        double luma_cut, chroma_cut;
        int chroma_delay;
        switch (p->output_vhs_tape_speed) {          /* :1773-1791 */
        case NTSCSIM_VHS_LP: luma_cut = 1900000; chroma_cut = 300000; chroma_delay = 12; break;
        case NTSCSIM_VHS_EP: luma_cut = 1400000; chroma_cut = 280000; chroma_delay = 14; break;
        default:             luma_cut = 2400000; chroma_cut = 320000; chroma_delay = 9;  break;
        }

        /* luma low-pass + emphasis :1793-1812 */
        for (k = 0; k < L; k++) {
            int32_t *Y = ROWY(k);
            onepole lp[3], pre;
            int f;
            for (f = 0; f < 3; f++) onepole_set(&lp[f], NTSC_RATE, luma_cut, 16);
            onepole_set(&pre, NTSC_RATE, luma_cut, 16);
            for (x = 0; x < W; x++) {
                double s = Y[x];
                for (f = 0; f < 3; f++) s = onepole_lp(&lp[f], s);
                s += onepole_hp(&pre, s) * 1.6;
                Y[x] = (int32_t)s;
            }
        }

        /* chroma low-pass :1814-1836 (U and V filters are independent, order irrelevant) */
        for (k = 0; k < L; k++) {
            row_lp3_delayed(ROWI(k), W, chroma_cut, 0, chroma_delay);
            row_lp3_delayed(ROWQ(k), W, chroma_cut, 0, chroma_delay);
        }

        /* vertical chroma blend through a one-line delay :1843-1863 */
        if (p->vhs_chroma_vert_blend && output_ntsc) {
            int32_t *delayU = (int32_t *)calloc((size_t)W, sizeof(int32_t));
            int32_t *delayV = (int32_t *)calloc((size_t)W, sizeof(int32_t));
            if (delayU && delayV) {
                for (k = 1; k < L; k++) {
                    int32_t *U = ROWI(k), *V = ROWQ(k);
                    for (x = 0; x < W; x++) {
                        int32_t cU = U[x], cV = V[x];
                        U[x] = (delayU[x] + cU + 1) >> 1;
                        V[x] = (delayV[x] + cV + 1) >> 1;
                        delayU[x] = cU;
                        delayV[x] = cV;
                    }
                }
            }
            free(delayU); free(delayV);
        }

        /* playback sharpening :1866-1883 */
        for (k = 0; k < L; k++) {
            int32_t *Y = ROWY(k);
            onepole lp[3];
            int f;
            for (f = 0; f < 3; f++) onepole_set(&lp[f], NTSC_RATE, luma_cut * 4, 0);
            for (x = 0; x < W; x++) {
                double s, ts;
                s = ts = Y[x];
                for (f = 0; f < 3; f++) ts = onepole_lp(&lp[f], ts);
                Y[x] = (int32_t)(s + ((s - ts) * p->vhs_out_sharpen * 2));
            }
        }

        /* composite out of the VCR :1885-1888 */
        if (!p->vhs_svideo_out) {
            for (k = 0; k < L; k++) {
                unsigned xi = scanline_phase(p, FRAME_Y(k), fieldno);
                row_chroma_into_luma(ROWY(k), ROWI(k), ROWQ(k), W, xi, p->subcarrier_amplitude);
                row_chroma_from_luma(ROWY(k), ROWI(k), ROWQ(k), chroma, W, xi,
                                     p->subcarrier_amplitude);
            }
        }
    }
    tap(taps->vhs_y, fY, n); tap(taps->vhs_i, fI, n); tap(taps->vhs_q, fQ, n);

    /* chroma dropout :1891-1901 */
    if (p->video_chroma_loss != 0) {
        for (k = 0; k < L; k++) {
            if ((ntsc_oracle_rng_next(g) % 100000U) < (unsigned)p->video_chroma_loss) {
                memset(ROWI(k), 0, (size_t)W * sizeof(int32_t));
                memset(ROWQ(k), 0, (size_t)W * sizeof(int32_t));
            }
        }
    }

    /* output chroma low-pass :1903-1908 */
    if (p->composite_out_chroma_lowpass) {
        if (p->composite_out_chroma_lowpass_lite) {
            for (k = 0; k < L; k++) row_lp3_delayed(ROWI(k), W, 2600000, 0, 1);
            for (k = 0; k < L; k++) row_lp3_delayed(ROWQ(k), W, 2600000, 0, 1);
        } else {
            for (k = 0; k < L; k++) row_lp3_delayed(ROWI(k), W, 1300000, 0, 2);
            for (k = 0; k < L; k++) row_lp3_delayed(ROWQ(k), W, 600000, 0, 4);
        }
    }
    tap(taps->final_y, fY, n); tap(taps->final_i, fI, n); tap(taps->final_q, fQ, n);

    /* YIQ -> RGB into the rows of this field only :1910-1916 */
    for (k = 0; k < L; k++) {
        uint8_t *drow = dst + (size_t)dst_linesize * FRAME_Y(k);
        for (x = 0; x < W; x++) {
            uint32_t px = yiq_to_rgb_pixel(ROWY(k)[x], ROWI(k)[x], ROWQ(k)[x]);
            memcpy(drow + 4 * (size_t)x, &px, 4);
        }
    }

    free(fY); free(fI); free(fQ); free(chroma);
    return 0;
}

/* ---------------------------------------------------------------------------- bob ---------- */

void ntsc_oracle_bob(uint8_t *frame, int linesize, int W, int H, unsigned field)
{
    /* field loop, ffmpeg_ntsc.cpp:2233-2257.  field 1: every odd row is copied onto the row
     * above it; field 0: row y+1 is copied onto odd row y while y+1 < H (so with an even H the
     * last odd row is left alone). */
    int y;
    if (field) {
        for (y = 1; y < H; y += 2)
            memcpy(frame + (size_t)linesize * (y - 1), frame + (size_t)linesize * y, (size_t)W * 4);
    } else {
        for (y = 1; y + 1 < H; y += 2)
            memcpy(frame + (size_t)linesize * y, frame + (size_t)linesize * (y + 1), (size_t)W * 4);
    }
}

/* ------------------------------------------------------------------- helpers for tests ----- */

uint64_t ntsc_oracle_fnv1a(const void *buf, size_t n)
{
    const uint8_t *b = (const uint8_t *)buf;
    uint64_t h = 0xcbf29ce484222325ULL;
    size_t i;
    for (i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    return h;
}

void ntsc_oracle_make_bars(uint8_t *bgra, int linesize, int W, int H, int rot)
{
    /* colour(x) = table[floor(8x/W)], table rotated by `rot` pixels */
    static const uint32_t table[8] = { 0xC0C0C0, 0xC0C000, 0x00C0C0, 0x00C000,
                                       0xC000C0, 0xC00000, 0x0000C0, 0x000000 };
    int x, y;
    for (y = 0; y < H; y++) {
        uint8_t *row = bgra + (size_t)linesize * y;
        for (x = 0; x < W; x++) {
            int sx = (x + rot) % W;
            uint32_t px = table[(8 * sx) / W];
            memcpy(row + 4 * (size_t)x, &px, 4);
        }
    }
}

void ntsc_oracle_make_noise(uint8_t *bgra, int linesize, int W, int H, uint32_t seed)
{
    uint32_t s = seed ? seed : 0x1234567u;
    int x, y;
    for (y = 0; y < H; y++) {
        uint8_t *row = bgra + (size_t)linesize * y;
        for (x = 0; x < W; x++) {
            uint32_t px;
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            px = s & 0xFFFFFFu;
            memcpy(row + 4 * (size_t)x, &px, 4);
        }
    }
}

/* see ntsc_oracle.h: the product's own BT.601 definition (parity unpinned, no libswscale here) */
void ntsc_oracle_bgra_to_yuv(const uint8_t *bgra, int bgra_linesize, int width, int height,
                             uint8_t *y, int y_linesize, uint8_t *u, int u_linesize,
                             uint8_t *v, int v_linesize, int is420)
{
    const int RY = (int)(0.299 * 219 / 255 * 32768 + 0.5), GY = (int)(0.587 * 219 / 255 * 32768 + 0.5),
              BY = (int)(0.114 * 219 / 255 * 32768 + 0.5);
    const int RU = (int)(-0.169 * 224 / 255 * 32768 + 0.5), GU = (int)(-0.331 * 224 / 255 * 32768 + 0.5),
              BU = (int)(0.500 * 224 / 255 * 32768 + 0.5);
    const int RV = (int)(0.500 * 224 / 255 * 32768 + 0.5), GV = (int)(-0.419 * 224 / 255 * 32768 + 0.5),
              BV = (int)(-0.081 * 224 / 255 * 32768 + 0.5);
    for (int yy = 0; yy < height; yy++)
        for (int x = 0; x < width; x++) {
            const uint8_t *p = bgra + (size_t)bgra_linesize * yy + 4 * x;
            y[(size_t)y_linesize * yy + x] =
                (uint8_t)((RY * p[2] + GY * p[1] + BY * p[0] + (16 << 15) + (1 << 14)) >> 15);
        }
    const int step = is420 ? 2 : 1, lg = is420 ? 2 : 1, sh = 15 + lg;
    for (int y0 = 0, cy = 0; y0 < height; y0 += step, cy++) {
        const int y1 = (is420 && y0 + 1 < height) ? y0 + 1 : y0;
        for (int cx = 0; cx < width / 2; cx++) {
            int rs = 0, gs = 0, bs = 0;
            for (int r = 0; r < (is420 ? 2 : 1); r++)
                for (int k = 0; k < 2; k++) {
                    const uint8_t *p = bgra + (size_t)bgra_linesize * (r ? y1 : y0) + 4 * (2 * cx + k);
                    bs += p[0]; gs += p[1]; rs += p[2];
                }
            u[(size_t)u_linesize * cy + cx] =
                (uint8_t)((RU * rs + GU * gs + BU * bs + (128 << sh) + (1 << (sh - 1))) >> sh);
            v[(size_t)v_linesize * cy + cx] =
                (uint8_t)((RV * rs + GV * gs + BV * bs + (128 << sh) + (1 << (sh - 1))) >> sh);
        }
    }
}

/* ---- unit entry points for the stand-in-free pin (tests/test_oracle_pure_pins.py): the primitives above
 * over arrays, so that they can be compared one by one with the reference's own text compiled by
 * build_ref_pure.sh with libc headers alone. */
void ntsc_oracle_unit_filter(double rate, double hz, double reset, int highpass, const double *in, size_t n,
                             double *out, double *alpha)
{
    onepole f;
    size_t i;
    onepole_set(&f, rate, hz, reset);
    if (alpha) *alpha = f.alpha;
    for (i = 0; i < n; i++) out[i] = highpass ? onepole_hp(&f, in[i]) : onepole_lp(&f, in[i]);
}

void ntsc_oracle_unit_rgb_to_yiq(const int32_t *rgb, size_t n, int32_t *yiq)
{
    size_t i;
    for (i = 0; i < n; i++) rgb_to_yiq(&yiq[3 * i], &yiq[3 * i + 1], &yiq[3 * i + 2], rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
}

uint64_t ntsc_oracle_unit_rgb_to_yiq_cube(int32_t *yiq_or_null)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    size_t k = 0;
    int r, g, b, c, s;
    for (r = 0; r < 256; r++)
        for (g = 0; g < 256; g++)
            for (b = 0; b < 256; b++) {
                int32_t v[3];
                rgb_to_yiq(&v[0], &v[1], &v[2], r, g, b);
                for (c = 0; c < 3; c++) {
                    uint32_t w = (uint32_t)v[c];
                    for (s = 0; s < 4; s++) { h ^= (w >> (8 * s)) & 0xff; h *= 0x100000001b3ULL; }
                    if (yiq_or_null) yiq_or_null[k++] = v[c];
                }
            }
    return h;
}

void ntsc_oracle_unit_yiq_to_rgb(const int32_t *yiq, size_t n, int32_t *rgb)
{
    size_t i;
    for (i = 0; i < n; i++) {
        uint32_t px = yiq_to_rgb_pixel(yiq[3 * i], yiq[3 * i + 1], yiq[3 * i + 2]);
        rgb[3 * i] = (int32_t)((px >> 16) & 255); rgb[3 * i + 1] = (int32_t)((px >> 8) & 255); rgb[3 * i + 2] = (int32_t)(px & 255);
    }
}
