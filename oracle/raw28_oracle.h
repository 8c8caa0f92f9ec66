/*
 * raw28_oracle.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the raw-composite decoder ffmpeg_raw28ntsc.cpp:
 * the sample front end hsync_dc_proc() :556-594 (three one-pole low-passes, a two-rate envelope
 * follower, a raw-sample delay line), the buffer window open_src/flush_src/refill_src/lazy_flush_src
 * :277-357, composite_layer() :601-849 (vertical-sync search, black / white level calibration on the
 * equalisation pulses, per-line equalisation, delay-4 comb Y/C split, grey-scale rendering, per-line
 * horizontal re-sync) and the field loop of main() :1006-1038.  One oracle object = one run of the
 * tool on one input file.
 */
#ifndef RAW28_ORACLE_H
#define RAW28_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct raw28_opts {
    double  sample_rate;            /* -s: 0 = "ntsc28" (315e6*8/88), else samples per second      */
    int32_t mark_sync;              /* -marksig  */
    int32_t disable_sync;           /* -nosig    */
    int32_t disable_wp_equ;         /* -nowequ   */
    int32_t show_subcarrier;        /* -showsc   */
    int32_t disable_subcarrier;     /* -nosc     */
    int32_t disable_equalization;   /* -noequ    */
} raw28_opts;

typedef struct raw28_oracle raw28_oracle;

/* `capture` must stay valid until close */
raw28_oracle *raw28_oracle_open(const raw28_opts *o, const uint8_t *capture, size_t n);
void raw28_oracle_close(raw28_oracle *d);
/* output frame size (preset_NTSC :395-402) and samples per scanline (compute_NTSC :253) */
void raw28_oracle_geometry(const raw28_oracle *d, int *width, int *height, int *scanline_samples);
/* one iteration of the field loop :1006-1038: returns 1 and fills the BGRA frame (memset 0 +
 * composite_layer), or 0 when the tool would stop (fewer than 256 scanlines left) */
int raw28_oracle_next_field(raw28_oracle *d, uint8_t *bgra, int linesize);
/* decoder state after the last field (tests compare it with the product's) */
void raw28_oracle_levels(const raw28_oracle *d, double *blank, double *white, uint64_t *read_pos);

/* the front end alone, over a whole capture: hsync_dc_raw and the delayed (and, with mark_sync,
 * marked) raw value of every sample, in stream order */
void raw28_oracle_front(const raw28_opts *o, const uint8_t *capture, size_t n, uint8_t *hsync_dc_raw,
                        uint8_t *raw_delayed);

/* synthetic capture for tests and the bench: `fields` NTSC fields sampled at 8 x fsc, 8 bit,
 * cxadc-like levels (sync tip low), colour bars + a moving luma ramp with subcarrier, seeded
 * noise, slow DC wander.  Returns the number of samples written (<= cap). */
size_t raw28_synth_capture(uint8_t *out, size_t cap, int fields, uint32_t seed, int noise_level);

#ifdef __cplusplus
}
#endif
#endif
