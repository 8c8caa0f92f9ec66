/*
 * tocomp_oracle.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the 8-bit YUV422P variant of the hot path, ffmpeg_to_composite.cpp:
 * composite_video_process() :629-952 with its helpers (:353-553), render_field() :1001-1129 and
 * black_key_feedback() :954-999.  Shares the glibc rand() clone of ntsc_oracle.h.
 *
 * The reference's Y/C separator reads two bytes past the end of every luma row (`c = Y[x+2]`,
 * :496) and writes three bytes past its scratch array (:529-532).  `oob_mode` selects what the
 * out-of-row read returns:
 *   TOCOMP_OOB_DEFINED (0): the value 16 (black), the box filter's own pre-charge value -- this is
 *                           the semantics of the product and of all GPU parity tests;
 *   TOCOMP_OOB_MEMORY  (1): whatever follows the row in the caller's buffer, like the reference --
 *                           used to pin this restatement against the reference extract;
 *   TOCOMP_OOB_PLANE   (2): the product's documented contract (include/ntscsim.h): the caller's bytes where the read
 *                           stays inside the luma plane (height * linesize bytes: padding, or the next row's first
 *                           pixels), 16 where it would leave it (the last row of a plane with linesize < width + 2,
 *                           where the reference reads memory it does not own).  == MEMORY wherever both are defined.
 * The out-of-array writes are dropped in both modes (they never feed a read).
 */
#ifndef TOCOMP_ORACLE_H
#define TOCOMP_ORACLE_H
#include "ntsc_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { TOCOMP_OOB_DEFINED = 0, TOCOMP_OOB_MEMORY = 1, TOCOMP_OOB_PLANE = 2 };

typedef struct tocomp_planes {
    uint8_t *data[3];       /* Y, U, V  (4:2:2: U,V are width/2 wide, full height) */
    int      linesize[3];
    int      width, height;
} tocomp_planes;

/* composite_video_process(dst, field, fieldno), in place on a YUV422P frame */
int tocomp_oracle_process(const ntscsim_params *p, ntsc_oracle_rng *g, tocomp_planes *dst,
                          unsigned field, uint64_t fieldno, int oob_mode);

/* render_field(dst, src, field, ...): src_is_420 = the decoder's format is YUV420P (:1005);
 * second_field = (field_number - src_pts >= ticks_per_frame/2) (:1035), only read for
 * interlaced sources */
void tocomp_oracle_render_field(tocomp_planes *dst, const tocomp_planes *src, int src_is_420,
                                int src_interlaced, int src_tff, int second_field, unsigned field);

/* black_key_feedback(dst, flt, field, ...) with black_key_level_feedback = level */
void tocomp_oracle_black_key_feedback(tocomp_planes *dst, tocomp_planes *flt, unsigned field,
                                      int level);

/* the bob copy of output_frame() :1177-1236.  mode: 0 = 4:2:2 field-rate bob (:1177-1196),
 * 1 = 4:2:0 field-rate bob, 2 = 4:2:0 interlaced (:1197-1236).  `bob` is YUV422P (mode 0) or
 * YUV420P (modes 1, 2: chroma planes (height+1)/2 rows); bob->width/height are not read. */
void tocomp_oracle_output_frame(tocomp_planes *bob, const tocomp_planes *frame, unsigned field,
                                int mode);

/* unit entry points for the stand-in-free pin (oracle/build_ref_pure.sh, tests/test_oracle_pure_pins.py) */
void tocomp_oracle_unit_filter(double rate, double hz, double reset, int highpass, const double *in, size_t n,
                               double *out, double *alpha);
void tocomp_oracle_unit_clampu8(const int32_t *x, size_t n, int32_t *out);
void tocomp_oracle_unit_black_key(int level, int wchroma, uint8_t *d, uint8_t *f, size_t n);

#ifdef __cplusplus
}
#endif
#endif
