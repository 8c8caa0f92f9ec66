/*
 * tocomp_oracle.c -- TEST INFRASTRUCTURE ONLY (see tocomp_oracle.h).
 * Scalar restatement of ffmpeg_to_composite.cpp's per-field path; compile with -ffp-contract=off.
 */
#include "tocomp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RATE_LUMA   ((315000000.00 * 4) / 88)
#define RATE_CHROMA ((315000000.00 * 4) / (88 * 2))   /* 4:2:2: half the luma rate */

typedef struct { double alpha, prev; } onepole;

static void op_set(onepole *f, double rate, double hz, double reset)
{   /* LowpassFilter::setFilter / resetFilter, ffmpeg_to_composite.cpp:103-113 */
    double timeInterval = 1.0 / rate;
    double tau = 1 / (hz * 2 * M_PI);
    f->alpha = timeInterval / (tau + timeInterval);
    f->prev = reset;
}
static double op_lp(onepole *f, double s)
{   /* :114-118 */
    double s1 = s * f->alpha, s2 = f->prev - (f->prev * f->alpha);
    return f->prev = s1 + s2;
}
static double op_hp(onepole *f, double s)
{   /* :119-123 */
    double s1 = s * f->alpha, s2 = f->prev - (f->prev * f->alpha);
    f->prev = s1 + s2;
    return s - f->prev;
}
static int clampu8(int x) { return x > 255 ? 255 : (x < 0 ? 0 : x); }   /* :335-342 */

static unsigned phase_of(const ntscsim_params *p, unsigned y, uint64_t fieldno)
{   /* :449-460 and :508-522: phase 0 ignores the offset; PAL has its own rule */
    if (p->tv_standard == NTSCSIM_TV_NTSC) {
        unsigned off = (unsigned)p->video_scanline_phase_shift_offset;
        if (p->video_scanline_phase_shift == 90) return (unsigned)((fieldno + off + (y >> 1)) & 3);
        if (p->video_scanline_phase_shift == 180) return (unsigned)((((fieldno + y) & 2) + off) & 3);
        if (p->video_scanline_phase_shift == 270) return (unsigned)((fieldno + off - (y >> 1)) & 3);
        return 0;
    }
    return (unsigned)((fieldno + y) & 3);
}

/* composite_video_chroma_lowpass :353-393 */
static void chroma_lowpass_full(const ntscsim_params *p, tocomp_planes *d, unsigned field)
{
    int pl, W2 = d->width / 2;
    unsigned y;
    for (pl = 1; pl <= 2; pl++)
        for (y = field; y < (unsigned)d->height; y += 2) {
            uint8_t *P = d->data[pl] + (size_t)y * d->linesize[pl];
            onepole lp[3], hp;
            double cutoff;
            int delay, x, f;
            if (p->tv_standard == NTSCSIM_TV_NTSC) { cutoff = (pl == 1) ? 1300000 : 600000; delay = (pl == 1) ? 2 : 4; }
            else { cutoff = 1300000; delay = 2; }
            op_set(&hp, RATE_CHROMA, cutoff / 2, 128);
            for (f = 0; f < 3; f++) op_set(&lp[f], RATE_CHROMA, cutoff, 128);
            for (x = 0; x < W2; x++) {
                double s = P[x];
                s += op_hp(&hp, s);
                for (f = 0; f < 3; f++) s = op_lp(&lp[f], s);
                if (x >= delay) P[x - delay] = (uint8_t)clampu8((int)s);
            }
        }
}

/* composite_video_chroma_lowpass_lite :395-431 */
static void chroma_lowpass_lite(tocomp_planes *d, unsigned field)
{
    int pl, W2 = d->width / 2;
    unsigned y;
    for (pl = 1; pl <= 2; pl++)
        for (y = field; y < (unsigned)d->height; y += 2) {
            uint8_t *P = d->data[pl] + (size_t)y * d->linesize[pl];
            onepole lp[3];
            int x, f;
            for (f = 0; f < 3; f++) op_set(&lp[f], RATE_CHROMA, (315000000.00 * 4) / (88 * 2 * 4), 128);
            for (x = 0; x < W2; x++) {
                double s = P[x];
                for (f = 0; f < 3; f++) s = op_lp(&lp[f], s);
                if (x >= 1) P[x - 1] = (uint8_t)clampu8((int)s);
            }
        }
}

/* composite_video_yuv_to_ntsc :434-477 */
static void yuv_to_ntsc(const ntscsim_params *p, tocomp_planes *d, unsigned field, uint64_t fieldno, int amp)
{
    static const int umult[4] = { 1, 0, -1, 0 }, vmult[4] = { 0, 1, 0, -1 };
    unsigned y;
    for (y = field; y < (unsigned)d->height; y += 2) {
        uint8_t *Y = d->data[0] + (size_t)y * d->linesize[0];
        uint8_t *U = d->data[1] + (size_t)y * d->linesize[1];
        uint8_t *V = d->data[2] + (size_t)y * d->linesize[2];
        unsigned xi = phase_of(p, y, fieldno), x, sx;
        for (x = 0; x < (unsigned)d->width; x += 2, Y += 2, U++, V++) {
            for (sx = 0; sx < 2; sx++) {
                unsigned sxi = xi + x + sx;
                int chroma = ((int)U[0] - 128) * amp * umult[sxi & 3];
                chroma += ((int)V[0] - 128) * amp * vmult[sxi & 3];
                Y[sx] = (uint8_t)clampu8(Y[sx] + (chroma / 50));
            }
            if (p->nocolor_subcarrier) U[0] = V[0] = 128;
        }
    }
}

/* composite_ntsc_to_yuv :480-553 */
static void ntsc_to_yuv(const ntscsim_params *p, tocomp_planes *d, unsigned field, uint64_t fieldno,
                        int amp_back, int oob_mode)
{
    int W = d->width, x;
    unsigned y;
    uint8_t *chroma = (uint8_t *)malloc((size_t)W + 8);
    for (y = field; y < (unsigned)d->height; y += 2) {
        uint8_t *Y = d->data[0] + (size_t)y * d->linesize[0];
        uint8_t *U = d->data[1] + (size_t)y * d->linesize[1];
        uint8_t *V = d->data[2] + (size_t)y * d->linesize[2];
        uint8_t dl[4] = { 16, 16, 16, 16 };
        unsigned sum = 16 * (4 - 2);
        dl[2] = Y[0]; sum += dl[2];
        dl[3] = Y[1]; sum += dl[3];
        for (x = 0; x < W; x++) {
            uint8_t c;
            if (x + 2 < W || oob_mode == TOCOMP_OOB_MEMORY) c = Y[x + 2];   /* :496 reads past the row */
            else if (oob_mode == TOCOMP_OOB_PLANE &&
                     (size_t)y * (size_t)d->linesize[0] + (size_t)x + 2 < (size_t)d->linesize[0] * (size_t)d->height)
                c = Y[x + 2];                      /* ... but stays inside the luma plane: the caller's own bytes */
            else c = 16;
            sum -= dl[0];
            dl[0] = dl[1]; dl[1] = dl[2]; dl[2] = dl[3]; dl[3] = c;
            sum += c;
            Y[x] = (uint8_t)(sum / 4);
            chroma[x] = (uint8_t)clampu8(c + 128 - Y[x]);
            if (p->nocolor_subcarrier_after_yc_sep) {           /* :503-507 */
                Y[x] = chroma[x];
                U[x / 2] = V[x / 2] = 128;
            }
        }
        if (!p->nocolor_subcarrier_after_yc_sep) {
            unsigned xi = phase_of(p, y, fieldno);
            for (x = (int)((4 - xi) & 3); x < W; x += 4) {      /* :524-527 (writes past W dropped) */
                if (x + 2 < W) chroma[x + 2] = (uint8_t)(255 - chroma[x + 2]);
                if (x + 3 < W) chroma[x + 3] = (uint8_t)(255 - chroma[x + 3]);
            }
            for (x = 0; x < W; x++)
                chroma[x] = (uint8_t)clampu8(((((int)chroma[x] - 128) * 50) / amp_back) + 128);
            if (xi & 1) {
                for (x = 0; x < W / 2; x++) { U[x] = (uint8_t)(255 - chroma[x * 2 + 1]); V[x] = (uint8_t)(255 - chroma[x * 2 + 0]); }
            } else {
                for (x = 0; x < W / 2; x++) { U[x] = (uint8_t)(255 - chroma[x * 2 + 0]); V[x] = (uint8_t)(255 - chroma[x * 2 + 1]); }
            }
        }
    }
    free(chroma);
}

int tocomp_oracle_process(const ntscsim_params *p, ntsc_oracle_rng *g, tocomp_planes *d,
                          unsigned field, uint64_t fieldno, int oob_mode)
{
    const int ntsc = p->tv_standard == NTSCSIM_TV_NTSC;
    const int W = d->width, H = d->height, W2 = W / 2;
    unsigned y;
    int x, f, i;
    if (!d->data[0] || !d->data[1] || !d->data[2] || W < 4 || H < 1 || field > 1) return -1;

    if (p->composite_in_chroma_lowpass) chroma_lowpass_full(p, d, field);          /* :632 */
    yuv_to_ntsc(p, d, field, fieldno, p->subcarrier_amplitude);                    /* :633 */

    if (p->composite_preemphasis != 0 && p->composite_preemphasis_cut > 0) {       /* :636-651 */
        for (y = field; y < (unsigned)H; y += 2) {
            uint8_t *Y = d->data[0] + (size_t)y * d->linesize[0];
            onepole pre;
            op_set(&pre, RATE_LUMA, p->composite_preemphasis_cut, 16);
            for (x = 0; x < W; x++) {
                double s = Y[x];
                s += op_hp(&pre, s) * p->composite_preemphasis;
                Y[x] = (uint8_t)clampu8((int)s);
            }
        }
    }
    if (p->video_noise != 0) {                                                     /* :654-666 */
        int noise = 0;
        unsigned m = (unsigned)(p->video_noise * 2 + 1);
        for (y = field; y < (unsigned)H; y += 2) {
            uint8_t *Y = d->data[0] + (size_t)y * d->linesize[0];
            for (x = 0; x < W; x++) {
                Y[x] = (uint8_t)clampu8(Y[x] + noise);
                noise += (int)(ntsc_oracle_rng_next(g) % m) - p->video_noise;
                noise /= 2;
            }
        }
    }
    if (p->vhs_head_switching) {                                                   /* :669-732 */
        unsigned tw = (unsigned)W + ((unsigned)W / 10u), tx, hx, pp, x2, shy = 0, xx;
        double noise = 0, t;
        int shif, ishif, yy;
        uint8_t *tmp = (uint8_t *)malloc(tw);
        if (p->vhs_head_switching_phase_noise != 0) {
            unsigned u = ntsc_oracle_rng_next(g);
            u *= ntsc_oracle_rng_next(g); u *= ntsc_oracle_rng_next(g); u *= ntsc_oracle_rng_next(g);
            u %= 2000000000U;
            noise = ((double)u / 1000000000U) - 1.0;
            noise *= p->vhs_head_switching_phase_noise;
        }
        t = ntsc ? tw * 262.5 : tw * 312.5;
        pp = (unsigned)(fmod(p->vhs_head_switching_phase + noise, 1.0) * t);
        hx = pp % tw;
        yy = (int)((pp / tw) * 2u) + (int)field;
        yy -= ntsc ? (262 - 240) * 2 : (312 - 288) * 2;
        tx = hx;
        ishif = (hx >= tw / 2) ? (int)(hx - tw) : (int)hx;
        shif = 0;
        while (yy < H) {
            if (yy >= 0 && shif != 0) {
                uint8_t *Y = d->data[0] + (size_t)yy * d->linesize[0];
                x2 = (tx + tw + (unsigned)shif) % tw;
                memset(tmp, 16, tw);
                memcpy(tmp, Y, (size_t)W);
                for (xx = tx; xx < (unsigned)W; xx++) { Y[xx] = tmp[x2]; if (++x2 == tw) x2 = 0; }
            }
            shif = (shy == 0) ? ishif : (shif * 7) / 8;
            tx = 0; yy += 2; shy++;
        }
        free(tmp);
    }
    if (!p->nocolor_subcarrier) ntsc_to_yuv(p, d, field, fieldno, p->subcarrier_amplitude_back, oob_mode);  /* :734 */

    if (p->video_chroma_noise != 0) {                                              /* :738-754 */
        int nU = 0, nV = 0;
        unsigned m = (unsigned)(p->video_chroma_noise * 2 + 1);
        for (y = field; y < (unsigned)H; y += 2) {
            uint8_t *U = d->data[1] + (size_t)y * d->linesize[1], *V = d->data[2] + (size_t)y * d->linesize[2];
            for (x = 0; x < W2; x++) {
                U[x] = (uint8_t)clampu8(U[x] + nU);
                V[x] = (uint8_t)clampu8(V[x] + nV);
                nU += (int)(ntsc_oracle_rng_next(g) % m) - p->video_chroma_noise; nU /= 2;
                nV += (int)(ntsc_oracle_rng_next(g) % m) - p->video_chroma_noise; nV /= 2;
            }
        }
    }
    if (p->video_chroma_phase_noise != 0) {                                        /* :755-781: not a rotation */
        int noise = 0;
        unsigned m = (unsigned)(p->video_chroma_phase_noise * 2 + 1);
        for (y = field; y < (unsigned)H; y += 2) {
            uint8_t *U = d->data[1] + (size_t)y * d->linesize[1], *V = d->data[2] + (size_t)y * d->linesize[2];
            double pi;
            noise += (int)(ntsc_oracle_rng_next(g) % m) - p->video_chroma_phase_noise;
            noise /= 2;
            pi = ((double)noise * M_PI) / 100;
            for (x = 0; x < W2; x++) {
                double u = (int)U[x] - 128, v = (int)V[x] - 128;
                double u_ = (u * cos(pi)) - (u * sin(pi));
                double v_ = (v * cos(pi)) + (v * sin(pi));
                U[x] = (uint8_t)clampu8((int)(u_ + 128));
                V[x] = (uint8_t)clampu8((int)(v_ + 128));
            }
        }
    }

    if (p->emulating_vhs) {                                                        /* :786-930 */
        double luma_cut = 2400000, chroma_cut = 320000;
        int cdelay = 4;
        if (p->output_vhs_tape_speed == NTSCSIM_VHS_LP) { luma_cut = 1900000; chroma_cut = 300000; cdelay = 5; }
        if (p->output_vhs_tape_speed == NTSCSIM_VHS_EP) { luma_cut = 1400000; chroma_cut = 280000; cdelay = 6; }
        for (y = field; y < (unsigned)H; y += 2) {                                 /* luma LP :812-831 */
            uint8_t *Y = d->data[0] + (size_t)y * d->linesize[0];
            onepole lp[3], pre;
            for (f = 0; f < 3; f++) op_set(&lp[f], RATE_LUMA, luma_cut, 16);
            op_set(&pre, RATE_LUMA, luma_cut, 16);
            for (x = 0; x < W; x++) {
                double s = Y[x];
                for (f = 0; f < 3; f++) s = op_lp(&lp[f], s);
                s += op_hp(&pre, s) * 1.6;
                Y[x] = (uint8_t)clampu8((int)s);
            }
        }
        for (y = field; y < (unsigned)H; y += 2) {                                 /* chroma LP :834-855 */
            uint8_t *U = d->data[1] + (size_t)y * d->linesize[1], *V = d->data[2] + (size_t)y * d->linesize[2];
            onepole lU[3], lV[3];
            for (f = 0; f < 3; f++) { op_set(&lU[f], RATE_CHROMA, chroma_cut, 128); op_set(&lV[f], RATE_CHROMA, chroma_cut, 128); }
            for (x = 0; x < W2; x++) {
                double s = U[x];
                for (f = 0; f < 3; f++) s = op_lp(&lU[f], s);
                if (x >= cdelay) U[x - cdelay] = (uint8_t)clampu8((int)s);
                s = V[x];
                for (f = 0; f < 3; f++) s = op_lp(&lV[f], s);
                if (x >= cdelay) V[x - cdelay] = (uint8_t)clampu8((int)s);
            }
        }
        if (p->vhs_chroma_vert_blend && ntsc) {                                    /* :862-882 */
            uint8_t *dU = (uint8_t *)malloc((size_t)W2 + 1), *dV = (uint8_t *)malloc((size_t)W2 + 1);
            memset(dU, 128, (size_t)W2); memset(dV, 128, (size_t)W2);
            for (y = field + 2; y < (unsigned)H; y += 2) {
                uint8_t *U = d->data[1] + (size_t)y * d->linesize[1], *V = d->data[2] + (size_t)y * d->linesize[2];
                for (x = 0; x < W2; x++) {
                    uint8_t cU = U[x], cV = V[x];
                    U[x] = (uint8_t)((dU[x] + cU + 1) >> 1);
                    V[x] = (uint8_t)((dV[x] + cV + 1) >> 1);
                    dU[x] = cU; dV[x] = cV;
                }
            }
            free(dU); free(dV);
        }
        for (y = field; y < (unsigned)H; y += 2) {                                 /* luma sharpen :887-901 */
            uint8_t *Y = d->data[0] + (size_t)y * d->linesize[0];
            onepole lp[3];
            for (f = 0; f < 3; f++) op_set(&lp[f], RATE_LUMA, luma_cut * 2, 16);
            for (x = 0; x < W; x++) {
                double s, ts;
                s = ts = Y[x];
                for (f = 0; f < 3; f++) ts = op_lp(&lp[f], ts);
                Y[x] = (uint8_t)clampu8((int)(s + ((s - ts) * p->vhs_out_sharpen)));
            }
        }
        for (y = field; y < (unsigned)H; y += 2) {                                 /* chroma sharpen :904-924 */
            uint8_t *U = d->data[1] + (size_t)y * d->linesize[1], *V = d->data[2] + (size_t)y * d->linesize[2];
            onepole lU[3], lV[3];
            for (f = 0; f < 3; f++) { op_set(&lU[f], RATE_CHROMA, chroma_cut * 2, 128); op_set(&lV[f], RATE_CHROMA, chroma_cut * 2, 128); }
            for (x = 0; x < W2; x++) {
                double s, ts;
                s = ts = U[x];
                for (f = 0; f < 3; f++) ts = op_lp(&lU[f], ts);
                U[x] = (uint8_t)clampu8((int)(s + ((s - ts) * p->vhs_out_sharpen_chroma)));
                s = ts = V[x];
                for (f = 0; f < 3; f++) ts = op_lp(&lV[f], ts);
                V[x] = (uint8_t)clampu8((int)(s + ((s - ts) * p->vhs_out_sharpen_chroma)));
            }
        }
        if (!p->vhs_svideo_out) {                                                  /* :926-929 */
            yuv_to_ntsc(p, d, field, fieldno, p->subcarrier_amplitude);
            ntsc_to_yuv(p, d, field, fieldno, p->subcarrier_amplitude, oob_mode);
        }
    }
    if (p->video_chroma_loss != 0) {                                               /* :932-942 */
        for (y = field; y < (unsigned)H; y += 2)
            if ((ntsc_oracle_rng_next(g) % 100000U) < (unsigned)p->video_chroma_loss) {
                memset(d->data[1] + (size_t)y * d->linesize[1], 128, (size_t)W2);
                memset(d->data[2] + (size_t)y * d->linesize[2], 128, (size_t)W2);
            }
    }
    for (i = 0; i < p->video_yc_recombine; i++) {                                  /* :943-946 */
        yuv_to_ntsc(p, d, field, fieldno, p->subcarrier_amplitude);
        ntsc_to_yuv(p, d, field, fieldno, p->subcarrier_amplitude, oob_mode);
    }
    if (p->composite_out_chroma_lowpass) chroma_lowpass_full(p, d, field);         /* :948-951 */
    else if (p->composite_out_chroma_lowpass_lite) chroma_lowpass_lite(d, field);
    return 0;
}

/* black_key / black_key_feedback :954-999 */
static void black_key(int level, uint8_t *dY, uint8_t *dU, uint8_t *dV, uint8_t *fY, uint8_t *fU, uint8_t *fV, int wchroma)
{
    int dLuma = *dY - (16 + level);
    int dChroma = abs(((int)(*dU)) + ((int)(*dV)) - 256) - level;
    if (dLuma + dChroma <= 0) { *dY = *fY; if (wchroma) { *dU = *fU; *dV = *fV; } }
    *fY = *dY;
    if (wchroma) { *fU = *dU; *fV = *dV; }
}

void tocomp_oracle_black_key_feedback(tocomp_planes *d, tocomp_planes *f, unsigned field, int level)
{
    unsigned y;
    int x;
    for (y = field; y < (unsigned)d->height; y += 2) {
        uint8_t *dY = d->data[0] + (size_t)y * d->linesize[0], *dU = d->data[1] + (size_t)y * d->linesize[1], *dV = d->data[2] + (size_t)y * d->linesize[2];
        uint8_t *fY = f->data[0] + (size_t)y * f->linesize[0], *fU = f->data[1] + (size_t)y * f->linesize[1], *fV = f->data[2] + (size_t)y * f->linesize[2];
        for (x = 0; x < d->width; x += 2) {
            black_key(level, dY + 0, dU, dV, fY + 0, fU, fV, 1);
            black_key(level, dY + 1, dU, dV, fY + 1, fU, fV, 0);
            dY += 2; dU++; dV++; fY += 2; fU++; fV++;
        }
    }
}

/* render_field :1001-1129.  The reference copies src->linesize[p] bytes per row (padding
 * included); only the first width (luma) / width/2 (chroma) bytes are pixels, and only those are
 * produced here. */
void tocomp_oracle_render_field(tocomp_planes *d, const tocomp_planes *s, int is420, int interlaced,
                                int tff, int second_field, unsigned field)
{
    unsigned y, sy, sy2, syf, csy, csy2, csyf;
    unsigned chroma_height = is420 ? (unsigned)s->height >> 1 : (unsigned)s->height;
    int pl, x;
    for (y = field; y < (unsigned)d->height; y += 2) {
        sy = (y * 0x100 * (unsigned)s->height) / (unsigned)d->height;
        syf = sy & 0xFF;
        sy >>= 8;
        csy = sy; csyf = syf;
        if (is420) { if (!(csy & 1)) csyf = 0; csy >>= 1; }
        if (interlaced) {
            unsigned which = tff ? 0u : 1u;
            if (second_field) which ^= 1;
            if (which == 0) { sy++; if (!(sy & 1U)) syf = 0; else sy--; }
            else if (!(sy & 1U)) { syf = 0; sy++; }
            if (which == 0) { csy++; if (!(csy & 1U)) csyf = 0; else csy--; }
            else if (!(csy & 1U)) { csyf = 0; csy++; }
            if (sy >= (unsigned)(s->height - 2)) { sy = (unsigned)s->height - 2; syf = 0; }
            sy2 = sy + 2;
            if (csy >= chroma_height - 2) { csy = chroma_height - 2; csyf = 0; }
            csy2 = csy + 1;
        } else {
            if (sy >= (unsigned)(s->height - 1)) { sy = (unsigned)s->height - 1; syf = 0; }
            sy2 = sy + 1;
            if (csy >= chroma_height - 1) { csy = chroma_height - 1; csyf = 0; }
            csy2 = csy + 1;
        }
        for (pl = 0; pl < 3; pl++) {
            /* 4:2:0 sources: chroma planes use the chroma row pair; 4:2:2: all planes use sy */
            unsigned r1 = (is420 && pl > 0) ? csy : sy, r2 = (is420 && pl > 0) ? csy2 : sy2;
            unsigned fr = (is420 && pl > 0) ? csyf : syf;
            const uint8_t *s1 = s->data[pl] + (size_t)s->linesize[pl] * r1;
            const uint8_t *s2 = s->data[pl] + (size_t)s->linesize[pl] * r2;
            uint8_t *o = d->data[pl] + (size_t)d->linesize[pl] * y;
            int n = pl == 0 ? d->width : d->width / 2;
            if (fr == 0) memcpy(o, s1, (size_t)n);
            else for (x = 0; x < n; x++)
                o[x] = (uint8_t)(s1[x] + ((uint8_t)((((int)s2[x] - (int)s1[x]) * (int)fr) >> 8)));
        }
    }
}

/* output_frame() ffmpeg_to_composite.cpp:1177-1236 -- the row copies only (everything else in
 * that function is encoder plumbing). */
void tocomp_oracle_output_frame(tocomp_planes *bob, const tocomp_planes *frame, unsigned field,
                                int mode)
{
    const unsigned H = (unsigned)frame->height;
    const size_t W = (size_t)frame->width;
    for (unsigned y = 0; y < H; y++) {
        unsigned sy;
        if (mode == 2) sy = y;                                  /* :1202-1203 */
        else if (field) sy = y | 1u;                            /* :1181-1182, :1204-1205 */
        else sy = (y + 1u) & ~1u;                               /* :1183-1184, :1206-1207 */
        if (sy >= H) sy -= 2;                                   /* :1186-1187, :1209-1210 */
        memcpy(bob->data[0] + (size_t)bob->linesize[0] * y,
               frame->data[0] + (size_t)frame->linesize[0] * sy, W);
        int chroma = 1;
        unsigned cy = y;
        if (mode == 1) { chroma = (y & 1u) == 0; cy = y >> 1; }                              /* :1225-1226 */
        else if (mode == 2) { chroma = (y & 2u) == 0; cy = (y & 1u) + ((y & ~3u) >> 1); }    /* :1215-1216 */
        if (chroma)
            for (int p = 1; p <= 2; p++)
                memcpy(bob->data[p] + (size_t)bob->linesize[p] * cy,
                       frame->data[p] + (size_t)frame->linesize[p] * sy, W / 2);
    }
}

/* ---- unit entry points for the stand-in-free pin (tests/test_oracle_pure_pins.py) */
void tocomp_oracle_unit_filter(double rate, double hz, double reset, int highpass, const double *in, size_t n,
                               double *out, double *alpha)
{
    onepole f;
    size_t i;
    op_set(&f, rate, hz, reset);
    if (alpha) *alpha = f.alpha;
    for (i = 0; i < n; i++) out[i] = highpass ? op_hp(&f, in[i]) : op_lp(&f, in[i]);
}

void tocomp_oracle_unit_clampu8(const int32_t *x, size_t n, int32_t *out)
{
    size_t i;
    for (i = 0; i < n; i++) out[i] = clampu8(x[i]);
}

void tocomp_oracle_unit_black_key(int level, int wchroma, uint8_t *d, uint8_t *f, size_t n)
{
    size_t i;
    for (i = 0; i < n; i++)
        black_key(level, d + 3 * i, d + 3 * i + 1, d + 3 * i + 2, f + 3 * i, f + 3 * i + 1, f + 3 * i + 2, wchroma);
}
