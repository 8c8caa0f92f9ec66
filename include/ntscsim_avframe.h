/*
 * ntscsim_avframe.h -- header-only AVFrame adapter for the drop-in call of include/ntscsim.h.
 *
 * The reference's hot path takes FFmpeg frames:
 *     composite_layer(AVFrame *dstframe, AVFrame *srcframe, InputFile &, unsigned field,
 *                     unsigned long long fieldno)                     ffmpeg_ntsc.cpp:1570
 * and reads exactly six members of them: data[0], linesize[0], width, height (:1578-1583, :1599,
 * :1911) and, of the source only, interlaced_frame / top_field_first (:1585-1588).  This header
 * maps those members onto ntscsim_field(), so that the call site ffmpeg_ntsc.cpp:2229
 *     composite_layer(ring[idx], (*i).input_avstream_video_frame_rgb, *i, (current & 1) ^ 1, current);
 * becomes
 *     ntscsim_field_avframe(sim, ring[idx], (*i).input_avstream_video_frame_rgb, (current & 1) ^ 1, current);
 * (INTEGRATION.md shows the whole patch).  The library itself never sees an FFmpeg type.
 *
 * With FFmpeg's headers on the include path this file includes <libavutil/frame.h>.  Without them
 * (this repository's image has no libav*), define NTSCSIM_AVFRAME_T to any struct type with the six
 * members above before including it -- tests/test_params_capi.py compiles it that way.
 */
#ifndef NTSCSIM_AVFRAME_H
#define NTSCSIM_AVFRAME_H

#include "ntscsim.h"

#ifndef NTSCSIM_AVFRAME_T
#include <libavutil/frame.h>
#define NTSCSIM_AVFRAME_T AVFrame
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* composite_layer() on AVFrames.  Returns NTSCSIM_OK, or the error code that stands for the
 * reference's silent `return` (:1578-1583): NULL frames / planes -> NTSCSIM_E_ARG, linesize below
 * 4 * width or mismatching sizes -> NTSCSIM_E_SIZE.  BGRA ("ARGB" in the reference's comments:
 * byte order B, G, R, A in memory), host memory, synchronous, rows of `field` only. */
static inline int ntscsim_field_avframe(ntscsim_ctx *ctx, NTSCSIM_AVFRAME_T *dstframe,
                                        const NTSCSIM_AVFRAME_T *srcframe, unsigned field,
                                        uint64_t fieldno)
{
    if (dstframe == 0 || srcframe == 0) return NTSCSIM_E_ARG;
    if (dstframe->data[0] == 0 || srcframe->data[0] == 0) return NTSCSIM_E_ARG;
    if (dstframe->linesize[0] < dstframe->width * 4) return NTSCSIM_E_SIZE;
    if (srcframe->linesize[0] < srcframe->width * 4) return NTSCSIM_E_SIZE;
    if (dstframe->width != srcframe->width || dstframe->height != srcframe->height) return NTSCSIM_E_SIZE;
    return ntscsim_field(ctx, srcframe->data[0], srcframe->linesize[0], srcframe->interlaced_frame,
                         srcframe->top_field_first, dstframe->data[0], dstframe->linesize[0],
                         dstframe->width, dstframe->height, field, fieldno);
}

/* The same call, asynchronously (ntscsim_submit(), include/ntscsim.h): returns at once with a ticket; the rows
 * are in dstframe after ntscsim_wait(ctx, *ticket).  srcframe is snapshotted by the call (the loop may
 * sws_scale the next decoded frame into it right away, ffmpeg_ntsc.cpp:603); dstframe must not be touched
 * until the wait.  `flags`: NTSCSIM_DESC_BOB (line doubling :2233-2257 done on the GPU as well) |
 * NTSCSIM_SUBMIT_SAME_SRC (srcframe still holds the frame of the previous submit: second field of a frame). */
static inline int ntscsim_submit_avframe(ntscsim_ctx *ctx, NTSCSIM_AVFRAME_T *dstframe,
                                         const NTSCSIM_AVFRAME_T *srcframe, unsigned field,
                                         uint64_t fieldno, uint32_t flags, uint64_t *ticket)
{
    if (dstframe == 0 || srcframe == 0) return NTSCSIM_E_ARG;
    if (dstframe->data[0] == 0 || srcframe->data[0] == 0) return NTSCSIM_E_ARG;
    if (dstframe->linesize[0] < dstframe->width * 4) return NTSCSIM_E_SIZE;
    if (srcframe->linesize[0] < srcframe->width * 4) return NTSCSIM_E_SIZE;
    if (dstframe->width != srcframe->width || dstframe->height != srcframe->height) return NTSCSIM_E_SIZE;
    return ntscsim_submit(ctx, srcframe->data[0], srcframe->linesize[0], srcframe->interlaced_frame,
                          srcframe->top_field_first, dstframe->data[0], dstframe->linesize[0],
                          dstframe->width, dstframe->height, field, fieldno, flags, ticket);
}

#ifdef __cplusplus
}
#endif
#endif /* NTSCSIM_AVFRAME_H */
