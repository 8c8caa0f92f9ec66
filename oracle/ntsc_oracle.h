/*
 * ntsc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, single-threaded) of the reference's per-field function
 * composite_layer() (ffmpeg_ntsc.cpp:1570-1921) and the helpers it calls.  It exists to CHECK the
 * HIP path; nothing in the product (composite-video-simulator_amd/, include/) may link, import or
 * call it.  Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * Pinning status: PINNED against outputs of the reference's own hot-path functions run in the
 * build container (see oracle/README.md): (1) the FNV-1a hashes recorded in SURVEY.md Appendix C,
 * (2) a line-range extract of ffmpeg_ntsc.cpp compiled by oracle/build_ref.sh (never copied into
 * this repo), compared bit-for-bit over the flag matrix in tests/test_oracle_vs_ref.py, and
 * (3) the committed fixtures under tests/golden/ that (2) generated.
 */
#ifndef NTSC_ORACLE_H
#define NTSC_ORACLE_H

#include <stdint.h>
#include <stddef.h>
#include "ntscsim.h" /* shares the ntscsim_params POD so tests feed identical params to both */

#ifdef __cplusplus
extern "C" {
#endif

/* glibc TYPE_3 additive-feedback generator (glibc stdlib/random_r.c), default seed 1. */
typedef struct ntsc_oracle_rng {
    uint32_t r[34];
    int      i;      /* index of the oldest word in the ring */
    uint64_t count;  /* draws made so far */
} ntsc_oracle_rng;

void     ntsc_oracle_rng_seed(ntsc_oracle_rng *g, uint32_t seed);
uint32_t ntsc_oracle_rng_next(ntsc_oracle_rng *g);             /* == rand()            */
void     ntsc_oracle_rng_discard(ntsc_oracle_rng *g, uint64_t n);

/* Optional per-stage taps: each is NULL or an int32 buffer of L*W elements (rows of this
 * field only, row k = frame row field+2k), filled with the plane after the named stage. */
typedef struct ntsc_oracle_taps {
    int32_t *composite_y;     /* Y after modulate + pre-emphasis + luma noise (:1611-1644)        */
    int32_t *headswitch_y;    /* Y after head switching (:1647-1713)                              */
    int32_t *demod_y, *demod_i, *demod_q;   /* after chroma_from_luma #1 (:1716)                  */
    int32_t *noise_i, *noise_q;             /* after chroma noise + phase noise (:1719-1764)      */
    int32_t *vhs_y, *vhs_i, *vhs_q;         /* after the whole VHS block (:1770-1889)             */
    int32_t *final_y, *final_i, *final_q;   /* just before YIQ->RGB (:1910)                       */
} ntsc_oracle_taps;

/*
 * One call == one composite_layer() call.  `g` is the process-wide rand() stream; it is
 * advanced by exactly the draws the reference makes.  Returns 0, or -1 where the reference
 * returns silently (:1578-1583).
 */
int ntsc_oracle_field(const ntscsim_params *p, ntsc_oracle_rng *g,
                      const uint8_t *src_bgra, int src_linesize, int src_interlaced, int src_tff,
                      uint8_t *dst_bgra, int dst_linesize,
                      int width, int height, unsigned field, uint64_t fieldno,
                      const ntsc_oracle_taps *taps);

/* Bob line doubling done by the field loop after the call (ffmpeg_ntsc.cpp:2233-2257). */
void ntsc_oracle_bob(uint8_t *frame_bgra, int linesize, int width, int height, unsigned field);

/* 64-bit FNV-1a, the hash SURVEY.md Appendix C records reference outputs with. */
uint64_t ntsc_oracle_fnv1a(const void *buf, size_t n);

/* Synthetic inputs of SURVEY.md 8(d): 8-bar 75% colour bars rotated by `rot` pixels, and
 * xorshift32 noise frames.  BGRA, alpha 0. */
void ntsc_oracle_make_bars(uint8_t *bgra, int linesize, int width, int height, int rot);
void ntsc_oracle_make_noise(uint8_t *bgra, int linesize, int width, int height, uint32_t seed);

/* Encoder-side colour conversion (output side; the tool calls sws_scale at
 * ffmpeg_ntsc.cpp:2266).  PARITY UNPINNED: libswscale is not in the reference tree; this is the
 * plain-C statement of the product's own definition (include/ntscsim.h: BT.601 limited range,
 * 15-bit fixed point, 2x1 / 2x2 block chroma), used to check the HIP kernel bit for bit. */
void ntsc_oracle_bgra_to_yuv(const uint8_t *bgra, int bgra_linesize, int width, int height,
                             uint8_t *y, int y_linesize, uint8_t *u, int u_linesize,
                             uint8_t *v, int v_linesize, int is420);

/* unit entry points for the stand-in-free pin (oracle/build_ref_pure.sh, tests/test_oracle_pure_pins.py) */
void ntsc_oracle_unit_filter(double rate, double hz, double reset, int highpass, const double *in, size_t n,
                             double *out, double *alpha);
void ntsc_oracle_unit_rgb_to_yiq(const int32_t *rgb, size_t n, int32_t *yiq);
uint64_t ntsc_oracle_unit_rgb_to_yiq_cube(int32_t *yiq_or_null);   /* FNV-1a over all 2^24 triples */
void ntsc_oracle_unit_yiq_to_rgb(const int32_t *yiq, size_t n, int32_t *rgb);

#ifdef __cplusplus
}
#endif
#endif
