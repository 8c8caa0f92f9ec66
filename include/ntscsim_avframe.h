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
#include <libavutil/buffer.h>
#include <libavutil/imgutils.h>
#define NTSCSIM_AVFRAME_T AVFrame
#define NTSCSIM_AVFRAME_HAVE_LIBAV 1
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

/* ---- the YUV422P tool, ffmpeg_to_composite.cpp:1783-1800 on its own AVFrames -------------------------------
 * One loop iteration -- render_field :1784, black_key_feedback :1787, composite_video_process :1790 (signature
 * :629: AVFrame *dst, unsigned field, unsigned long long fieldno) and the copy loops of output_frame :1793-1796
 * (:1177-1236) -- as ONE call.  Arguments are the tool's own objects:
 *   frame        output_avstream_video_frame (YUV422P)
 *   input_frame  output_avstream_video_input_frame (what sws_scale wrote, :1770-1778; 4:2:0 when the decoder's
 *                format is, :1707-1719 -- pass src_is_420 accordingly) or NULL for "no render"
 *   filter_frame output_avstream_video_filter_frame, or NULL
 *   enc_frame    the frame output_frame() fills for its encoder (output_avstream_video_bob_frame), or NULL
 *   second       field_number - src_pts >= ticks_per_frame / 2 (:1035): second field of the source frame
 *   out_mode / out_field: NTSCSIM_OUT422_* and output_frame()'s `field` argument
 * ntscsim_loop422_from_avframes() fills the POD; ntscsim_field422_avframe() / ntscsim_submit422_avframe() call it. */
static inline int ntscsim_loop422_from_avframes(ntscsim_loop422 *it, NTSCSIM_AVFRAME_T *frame,
                                                const NTSCSIM_AVFRAME_T *input_frame, int src_is_420, int second,
                                                NTSCSIM_AVFRAME_T *filter_frame, NTSCSIM_AVFRAME_T *enc_frame,
                                                uint32_t out_mode, unsigned out_field, int nocomp,
                                                unsigned field, uint64_t fieldno)
{
    int k;
    if (it == 0 || frame == 0) return NTSCSIM_E_ARG;
    for (k = 0; k < (int)sizeof(*it); k++) ((unsigned char *)it)[k] = 0;
    it->struct_size = (uint32_t)sizeof(*it);
    it->width = frame->width; it->height = frame->height;
    for (k = 0; k < 3; k++) { it->frame.data[k] = frame->data[k]; it->frame.linesize[k] = frame->linesize[k]; }
    if (input_frame != 0) {
        if (input_frame->width != frame->width) return NTSCSIM_E_SIZE;
        it->src_height = input_frame->height;
        for (k = 0; k < 3; k++) { it->src.data[k] = input_frame->data[k]; it->src.linesize[k] = input_frame->linesize[k]; }
        it->flags |= (input_frame->interlaced_frame ? NTSCSIM_422_INTERLACED : 0u) |
                     (input_frame->top_field_first ? NTSCSIM_422_TFF : 0u) | (src_is_420 ? NTSCSIM_422_SRC420 : 0u) |
                     (second ? NTSCSIM_422_SECOND : 0u);
    }
    if (nocomp) it->flags |= NTSCSIM_422_NOCOMP;
    if (filter_frame != 0)
        for (k = 0; k < 3; k++) { it->filter.data[k] = filter_frame->data[k]; it->filter.linesize[k] = filter_frame->linesize[k]; }
    if (enc_frame != 0) {
        if (enc_frame->width != frame->width || enc_frame->height != frame->height) return NTSCSIM_E_SIZE;
        for (k = 0; k < 3; k++) { it->out.data[k] = enc_frame->data[k]; it->out.linesize[k] = enc_frame->linesize[k]; }
    }
    it->out_mode = out_mode; it->out_field = out_field;
    it->field = field; it->fieldno = fieldno;
    return NTSCSIM_OK;
}

static inline int ntscsim_field422_avframe(ntscsim_ctx *ctx, NTSCSIM_AVFRAME_T *frame,
                                           const NTSCSIM_AVFRAME_T *input_frame, int src_is_420, int second,
                                           NTSCSIM_AVFRAME_T *filter_frame, NTSCSIM_AVFRAME_T *enc_frame,
                                           uint32_t out_mode, unsigned out_field, int nocomp,
                                           unsigned field, uint64_t fieldno)
{
    ntscsim_loop422 it;
    const int rc = ntscsim_loop422_from_avframes(&it, frame, input_frame, src_is_420, second, filter_frame, enc_frame,
                                                 out_mode, out_field, nocomp, field, fieldno);
    return rc != NTSCSIM_OK ? rc : ntscsim_field422(ctx, &it);
}

static inline int ntscsim_submit422_avframe(ntscsim_ctx *ctx, NTSCSIM_AVFRAME_T *frame,
                                            const NTSCSIM_AVFRAME_T *input_frame, int src_is_420, int second,
                                            NTSCSIM_AVFRAME_T *filter_frame, NTSCSIM_AVFRAME_T *enc_frame,
                                            uint32_t out_mode, unsigned out_field, int nocomp,
                                            unsigned field, uint64_t fieldno, uint32_t submit_flags, uint64_t *ticket)
{
    ntscsim_loop422 it;
    const int rc = ntscsim_loop422_from_avframes(&it, frame, input_frame, src_is_420, second, filter_frame, enc_frame,
                                                 out_mode, out_field, nocomp, field, fieldno);
    return rc != NTSCSIM_OK ? rc : ntscsim_submit422(ctx, &it, submit_flags, ticket);
}

/* ---- frames the engine can serve without a host copy --------------------------------------------------------
 * The tools allocate their frames with av_frame_get_buffer(frame, 64) (ffmpeg_ntsc.cpp:351 in.rgb, :2082 the
 * frame-delay ring; ffmpeg_to_composite.cpp: output_avstream_video_frame / _input_frame / _bob_frame) -- av_malloc
 * blocks, which ntscsim_submit*() must stage (include/ntscsim.h "Host buffers").  ntscsim_av_frame_get_buffer() is
 * the one-token replacement: same arguments, same result (width / height / format must be set; data[], linesize[],
 * buf[0], extended_data are filled; av_frame_free() / av_frame_unref() release it), but the planes live in ONE block
 * of pinned host memory (ntscsim_host_alloc) wrapped by av_buffer_create(): DMA reads it, the GPU writes the field
 * rows straight into it.  Layout as libavutil's get_video_buffer(): linesizes from av_image_fill_linesizes() with the width
 * padded until linesize[0] is a multiple of `align`, each rounded up to `align`, height padded to a multiple of 32 rows, planes placed by
 * av_image_fill_pointers(), `align` spare bytes in front (none needed: the block is page aligned) and 64 behind.
 * Returns 0 or a negative AVERROR, like av_frame_get_buffer().
 *
 * Compiled only against FFmpeg's headers; a unit that defines NTSCSIM_AVFRAME_T itself may define
 * NTSCSIM_AVFRAME_HAVE_LIBAV as well if it provides av_image_fill_linesizes / av_image_fill_pointers /
 * av_buffer_create / FFALIGN / AVERROR with libavutil's signatures (tests/test_params_capi.py does, to compile
 * and run this function without libav). */
#ifdef NTSCSIM_AVFRAME_HAVE_LIBAV
static void ntscsim_av_buffer_free_(void *opaque, uint8_t *data)
{
    (void)opaque;
    ntscsim_host_free(data);
}

static inline int ntscsim_av_frame_get_buffer(NTSCSIM_AVFRAME_T *frame, int align)
{
    int i, ret, padded_height, total;
    uint8_t *block;
    if (frame == 0 || frame->width <= 0 || frame->height <= 0 || frame->format < 0) return AVERROR(EINVAL);
    if (align <= 0) align = 32;
    /* (libavutil's search: the smallest width alignment that makes linesize[0] a multiple of `align`) */
    for (i = 1; i <= align; i += i) {
        ret = av_image_fill_linesizes(frame->linesize, (enum AVPixelFormat)frame->format, FFALIGN(frame->width, i));
        if (ret < 0) return ret;
        if (!(frame->linesize[0] & (align - 1))) break;
    }
    for (i = 0; i < 4 && frame->linesize[i]; i++) frame->linesize[i] = FFALIGN(frame->linesize[i], align);
    padded_height = FFALIGN(frame->height, 32);
    total = av_image_fill_pointers(frame->data, (enum AVPixelFormat)frame->format, padded_height, 0, frame->linesize);
    if (total < 0) return total;
    block = (uint8_t *)ntscsim_host_alloc((size_t)total + 64);
    if (block == 0) return AVERROR(ENOMEM);
    frame->buf[0] = av_buffer_create(block, total + 64, ntscsim_av_buffer_free_, 0, 0);
    if (frame->buf[0] == 0) { ntscsim_host_free(block); return AVERROR(ENOMEM); }
    ret = av_image_fill_pointers(frame->data, (enum AVPixelFormat)frame->format, padded_height, block, frame->linesize);
    if (ret < 0) { av_buffer_unref(&frame->buf[0]); return ret; }
    frame->extended_data = frame->data;
    return 0;
}
#endif /* NTSCSIM_AVFRAME_HAVE_LIBAV */

#ifdef __cplusplus
}
#endif
#endif /* NTSCSIM_AVFRAME_H */
