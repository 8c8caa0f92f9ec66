/*
 * ntscsim.h -- C ABI of the MI355X-native NTSC composite / VHS field simulator.
 *
 * This is the drop-in boundary for ONE hot path of joncampbell123/composite-video-simulator:
 * the per-field function
 *
 *     void composite_layer(AVFrame *dst, AVFrame *src, InputFile&, unsigned field,
 *                          unsigned long long fieldno)          ffmpeg_ntsc.cpp:1570-1921
 *
 * called once per output field from the field loop at ffmpeg_ntsc.cpp:2229, together with the
 * ~35 process-wide globals it reads (ffmpeg_ntsc.cpp:205-214, :756-809) and the libc rand()
 * stream it consumes.  Everything here is plain C: pointers, sizes, PODs.  No torch / HIP types
 * appear in any signature; a HIP stream is passed as an opaque void*.
 *
 * The reference has no plugin API -- the "interface" is the function signature above plus the
 * globals -- so the mapping is:
 *
 *   reference                                   | this header
 *   --------------------------------------------+------------------------------------------
 *   globals set by parse_argv() :972-1282       | struct ntscsim_params + ntscsim_params_parse_argv()
 *   preset_NTSC()/preset_PAL() :815-831         | ntscsim_params_init() / "-tvstd" flag
 *   composite_layer(dst,src,_,field,fieldno)    | ntscsim_field()            (host AVFrame planes)
 *   N calls of composite_layer in the loop :2202 | ntscsim_fields_device()   (batched, HBM resident)
 *   process-wide rand() state (never seeded)    | explicit 64-bit stream position per field
 *   bob line doubling :2233-2257                | NTSCSIM_DESC_BOB flag of ntscsim_field_desc
 *   silent `return` on bad frames :1578-1583    | negative error codes
 */
#ifndef NTSCSIM_H
#define NTSCSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTSCSIM_ABI_VERSION 1

/* ---- error codes (the reference returns silently, ffmpeg_ntsc.cpp:1578-1583) ---- */
enum {
    NTSCSIM_OK          = 0,
    NTSCSIM_E_ARG       = -1,  /* NULL pointer / bad argument                              */
    NTSCSIM_E_SIZE      = -2,  /* linesize < 4*width, size mismatch, width/height too small */
    NTSCSIM_E_NODEV     = -3,  /* no HIP device / HIP runtime unavailable                   */
    NTSCSIM_E_HIP       = -4,  /* a HIP call failed (see ntscsim_last_error)                */
    NTSCSIM_E_NOMEM     = -5,
    NTSCSIM_E_PARAM     = -6,  /* params outside the supported domain                       */
    NTSCSIM_E_FLAG      = -7,  /* unknown switch / bad value (reference: "return 1")        */
    NTSCSIM_E_HELP      = -8,  /* -h / -help was given                                      */
    NTSCSIM_E_INTERNAL  = -9
};

enum { NTSCSIM_TV_NTSC = 0, NTSCSIM_TV_PAL = 1 };
enum { NTSCSIM_VHS_SP = 0, NTSCSIM_VHS_LP = 1, NTSCSIM_VHS_EP = 2 }; /* ffmpeg_ntsc.cpp:801-805 */

/*
 * Immutable snapshot of the globals composite_layer() reads.  Field names follow the
 * reference's global names; the comment gives file:line of the global and its default.
 */
typedef struct ntscsim_params {
    uint32_t struct_size;                 /* = sizeof(ntscsim_params); versions the struct      */
    int32_t  tv_standard;                 /* output_ntsc/output_pal :209-210; NTSCSIM_TV_NTSC    */
    int32_t  output_width;                /* :207  720                                          */
    int32_t  output_height;               /* :208  480 (576 PAL)                                */
    int32_t  video_scanline_phase_shift;        /* :213  180 (0|90|180|270)                     */
    int32_t  video_scanline_phase_shift_offset; /* :214  0                                      */
    double   composite_preemphasis;       /* :756  0                                            */
    double   composite_preemphasis_cut;   /* :757  1000000                                      */
    double   vhs_out_sharpen;             /* :759  1.5                                          */
    int32_t  vhs_head_switching;          /* :761  false (true with -vhs)                       */
    int32_t  _pad0;
    double   vhs_head_switching_point;    /* :762  1 - 4.51/262.5                               */
    double   vhs_head_switching_phase;    /* :763  0.99/262.5                                   */
    double   vhs_head_switching_phase_noise; /* :764 (1/500)/262.5                              */
    int32_t  composite_in_chroma_lowpass;       /* :766  true                                   */
    int32_t  composite_out_chroma_lowpass;      /* :767  true                                   */
    int32_t  composite_out_chroma_lowpass_lite; /* :768  true                                   */
    int32_t  video_yc_recombine;          /* :770  0  (parsed, unused by ffmpeg_ntsc L1)        */
    int32_t  video_chroma_noise;          /* :772  0                                            */
    int32_t  video_chroma_phase_noise;    /* :773  0                                            */
    int32_t  video_chroma_loss;           /* :774  0                                            */
    int32_t  video_noise;                 /* :775  2                                            */
    int32_t  subcarrier_amplitude;        /* :776  50                                           */
    int32_t  subcarrier_amplitude_back;   /* :777  50                                           */
    int32_t  emulating_vhs;               /* :791  false                                        */
    int32_t  nocolor_subcarrier;          /* :794  false                                        */
    int32_t  nocolor_subcarrier_after_yc_sep; /* :795 false (parsed, unused by ffmpeg_ntsc L1)  */
    int32_t  vhs_chroma_vert_blend;       /* :796  true                                         */
    int32_t  vhs_svideo_out;              /* :797  false                                        */
    int32_t  enable_composite_emulation;  /* :798  true (-nocomp clears it; L1 never tests it)  */
    int32_t  output_vhs_tape_speed;       /* :809  NTSCSIM_VHS_SP                               */
    int32_t  black_key_level_feedback;    /* ffmpeg_to_composite.cpp:322  -1 (variant only)      */
    double   vhs_out_sharpen_chroma;      /* ffmpeg_to_composite.cpp:271  0.85 (variant only)    */
    /* EXTENSION, not in the reference (SURVEY 0.3 / 8(f) "ext"): multipath ghosting on the
     * composite signal, between the luma-noise stage (:1644) and head switching (:1647):
     *   Y'[x] = Y[x] + ( sum_k ghost_gain[k] * Y[x - ghost_delay[k]] ) / 256      (Y[<0] = 0)
     * taps read the un-ghosted signal (FIR), C integer arithmetic.  ghost_taps = 0 (default)
     * disables it and leaves the output bit-identical to the reference.  BGRA path only.
     * (Speed: while every delay is at most 63 samples the taps are folded into the encoder
     * kernel; a longer delay costs a pass over the composite plane) */
    int32_t  ghost_taps;                  /* 0..NTSCSIM_MAX_GHOST_TAPS                           */
    int32_t  ghost_delay[4];              /* samples, 1..4096                                    */
    int32_t  ghost_gain[4];               /* 1/256 units, -256..256                              */
    int32_t  _pad1;
} ntscsim_params;
#define NTSCSIM_MAX_GHOST_TAPS 4

/* Host-side (L4/L2) settings parse_argv() also fills; not read by the field DSP. */
#define NTSCSIM_MAX_INPUTS 16
typedef struct ntscsim_cli {
    const char *input_paths[NTSCSIM_MAX_INPUTS]; /* -i (repeatable) :1012                       */
    int32_t     n_inputs;
    int32_t     frame_delay;                     /* -d :1003, 1..256, default 1                 */
    const char *output_path;                     /* -o :1017                                    */
    int32_t     use_422_colorspace;              /* -422 / -420 :1022-1027                      */
    /* audio-only flags are accepted for CLI compatibility and recorded, never used here */
    int32_t     emulating_preemphasis, emulating_deemphasis, output_vhs_hifi;
    double      output_audio_hiss_db, output_audio_linear_buzz, vhs_linear_high_boost;
    int32_t     output_video_as_interlaced;      /* ffmpeg_to_composite -vi / -vp :1399-1404 (default 0: bob) */
    int32_t     _pad;
} ntscsim_cli;

/* preset_NTSC() + the global initialisers (ffmpeg_ntsc.cpp:205-214, :756-809, :824-831). */
void ntscsim_params_init(ntscsim_params *p);
void ntscsim_cli_init(ntscsim_cli *c);

/*
 * Mirror of parse_argv() (ffmpeg_ntsc.cpp:972-1282): same switch names (any number of leading
 * '-'), same defaults, same preset side effects (-vhs :1141, -vhs-speed :1160, -comp-catv* :1077)
 * and the same post-parse derivation of subcarrier_amplitude_back (:1264-1265).
 * argv[0] is the program name, as in main().  `cli` may be NULL.  If `require_io` is non-zero the
 * "No output file / No input files specified" checks (:1271-1278) apply.
 * Returns NTSCSIM_OK, NTSCSIM_E_HELP (-h), or NTSCSIM_E_FLAG (reference: return 1).
 */
int ntscsim_params_parse_argv(ntscsim_params *p, ntscsim_cli *cli, int argc,
                              const char *const *argv, int require_io);

/*
 * The 8-bit YUV422P sibling tool, ffmpeg_to_composite.cpp: its global initialisers (:267-333 --
 * note the different head-switch defaults: one `phase` parameter = 1-4.51/262.5, noise 1/300 line)
 * and its parse_argv() (:1325-1639: no -d / -comp-catv4 / -vhs-head-switching-phase; adds
 * -bkey-feedback, -ss/-se/-t, -a/-v/-an/-vn, -vi/-vp; -vhs-head-switching-point sets the phase;
 * -comp-catv* = 1.5/2.5/4 at fsc/2; amplitude_back += 50*pre/4 (:1627)).
 */
void ntscsim_params_init_to_composite(ntscsim_params *p);
int  ntscsim_params_parse_argv_to_composite(ntscsim_params *p, ntscsim_cli *cli, int argc,
                                            const char *const *argv, int require_io);

/* Rejects parameter values for which the reference's own behaviour is undefined
 * (negative noise levels, subcarrier amplitude 0 -> division by zero at :1545, ...). */
int ntscsim_params_validate(const ntscsim_params *p);

/*
 * Number of libc rand() draws one composite_layer() call makes:
 *   [W*L if video_noise] + [4 if head switching with phase noise] + [2*W*L if chroma noise]
 *   + [L if chroma phase noise] + [L if chroma loss],   L = rows of this field.
 */
uint64_t ntscsim_rng_calls_per_field(const ntscsim_params *p, int width, int height,
                                     unsigned field);

/* glibc TYPE_3 rand() clone (stdlib/random_r.c, seed 1): draw `n` values starting at stream
 * position `pos` (0 = first rand() of the process) using O(log pos) jump-ahead. */
void ntscsim_rng_draw(uint64_t pos, size_t n, uint32_t *out);

/* ------------------------------------------------------------------------------------------ */

typedef struct ntscsim_ctx ntscsim_ctx;

/* Bind to HIP device `device` (ordinal), snapshot `p`.  Fails with NTSCSIM_E_NODEV when no
 * GPU is present: there is NO CPU fallback in this library.
 * Threading: a ctx (and the batches created from it) is NOT re-entrant -- one thread at a time
 * per ctx, like the reference's globals + process-wide rand(); use one ctx per thread / stream.
 * Work enqueued through one ctx is ordered by the caller's stream; its scratch is reused. */
int  ntscsim_create(const ntscsim_params *p, int device, ntscsim_ctx **out);
void ntscsim_destroy(ntscsim_ctx *ctx);
const char *ntscsim_strerror(int code);
const char *ntscsim_last_error(const ntscsim_ctx *ctx); /* text of the last HIP failure */

/*
 * Arithmetic mode of the BGRA path.
 *   NTSCSIM_MODE_EXACT  (default): fp64 filters in the reference's operation order, no FMA --
 *                       output bit-identical to composite_layer().
 *   NTSCSIM_MODE_FAST32: the same pipeline, integer stages and rand() stream, with the filters,
 *                       the colour matrices and the phase rotation in fp32 (FMA allowed).  Output
 *                       differs from the reference by at most 1 LSB per 8-bit channel on a small
 *                       fraction of pixels (bounds in tests/test_gpu_fast_mode.py).
 *   NTSCSIM_MODE_FLOAT:  the tolerance mode as a pipeline of its own (csrc/ntsc_float.hip): Y / I / Q and the composite
 *                       sample are fp32 from RGB -> YIQ to YIQ -> RGB -- no inter-stage (int), no int <-> float
 *                       conversion of the signal, FMA contraction on -- while the rand() stream, the noise accumulators,
 *                       head-switch geometry, phase-noise table and dropout stay the exact mode's integers.  Same stated
 *                       tolerance as FAST32 (at most 1 LSB per 8-bit channel; bounds and the measured share of exact
 *                       pixels in tests/test_gpu_fast_mode.py).  Default preset and the -vhs family with its standard
 *                       switches; other switch sets -- and the short launches of ntscsim_field() / ntscsim_submit() -- run
 *                       the FAST32 forms in this mode.
 */
enum { NTSCSIM_MODE_EXACT = 0, NTSCSIM_MODE_FAST32 = 1, NTSCSIM_MODE_FLOAT = 2 };
int ntscsim_set_mode(ntscsim_ctx *ctx, int mode);

/* Stream position of the next field processed through ntscsim_field() (starts at 0, advances
 * by ntscsim_rng_calls_per_field per call -- exactly the reference's process-wide rand()). */
uint64_t ntscsim_get_rng_pos(const ntscsim_ctx *ctx);
void     ntscsim_set_rng_pos(ntscsim_ctx *ctx, uint64_t pos);

/*
 * Drop-in for composite_layer() (ffmpeg_ntsc.cpp:1570).  HOST buffers, as an AVFrame holds them:
 *   src_bgra/src_linesize/src_interlaced/src_tff = srcframe->data[0], ->linesize[0],
 *                                                  ->interlaced_frame, ->top_field_first
 *   dst_bgra/dst_linesize/width/height           = dstframe->data[0], ->linesize[0], ->width, ->height
 * Only rows y = field, field+2, ... of dst are written (alpha byte 0, :1914); other rows are
 * untouched.  Synchronous.  Uses and advances the ctx's rand() stream position.
 * A dst frame the GPU can address -- memory declared with ntscsim_host_pin(), or pinned memory (ntscsim_host_frame_alloc /
 * ntscsim_av_frame_get_buffer of ntscsim_avframe.h / hipHostMalloc; the runtime is asked at every call, nothing is
 * registered or remembered) -- whose rows are 16-byte aligned is written by the kernels themselves; any other frame goes
 * through a device copy and a download.  Same bytes either way.  (The call runs encoder and decoder as wavefront ROLES of
 * one workgroup per 63 rows for the -vhs family and the default preset: DESIGN.md 1c; 4.8-5.8k calls per second at 720x486.)
 */
int ntscsim_field(ntscsim_ctx *ctx,
                  const uint8_t *src_bgra, int src_linesize, int src_interlaced, int src_tff,
                  uint8_t *dst_bgra, int dst_linesize,
                  int width, int height, unsigned field, uint64_t fieldno);

/*
 * ASYNCHRONOUS drop-in for composite_layer() (ntscsim_submit / ntscsim_wait).
 *
 * The reference's loop calls composite_layer(ring[idx], in.rgb, in, (current&1)^1, current) at
 * ffmpeg_ntsc.cpp:2229, then line-doubles (:2233-2257), converts (:2266) and encodes the frame before it
 * composites the next field.  One synchronous field per call leaves the GPU 98 % idle (a field is four
 * workgroups of work).  ntscsim_submit() takes the same arguments, returns at once with a ticket, and the
 * ctx coalesces submitted fields into launches of `depth` fields; ntscsim_wait(ticket) returns when that
 * field's rows are in the caller's dst frame.  The caller moves everything it does with the frame after
 * composite_layer() (bob, sws_scale, output_frame :2233-2280) behind a wait issued `depth` fields later
 * (INTEGRATION.md section 1b shows the patch of the loop :2202-2282).
 *
 * Contract (what makes a sequence of submits + waits byte-identical to the same sequence of
 * ntscsim_field() calls on the same pointers, rand() position included):
 *   - src is SNAPSHOTTED by the call: when ntscsim_submit() returns the caller may overwrite src (the
 *     reference rewrites in.rgb with the next decoded frame, InputFile::frame_copy_scale :544-613).
 *     NTSCSIM_SUBMIT_SAME_SRC: src still holds the frame of the previous submit on this ctx (same
 *     pointer, geometry and content -- e.g. the second field of a frame): the device copy is reused and
 *     nothing is uploaded.  NTSCSIM_SUBMIT_SRC_STABLE: the caller promises to leave src alone until the
 *     ticket has been waited for; the call then does not wait for its upload either.
 *   - dst is written some time between submit and the return of ntscsim_wait(ticket): exactly the rows
 *     composite_layer() writes (rows field, field+2, ... :1910-1916; alpha 0), or with NTSCSIM_DESC_BOB
 *     in `flags` additionally the rows the loop's line doubling writes (:2233-2257: every row except, for
 *     field 0 and an even height or field 1 and an odd height, the last one).  Other rows are not touched.
 *     The caller must not read or write dst between submit and wait.  Fields in flight that share a dst
 *     frame are delivered in submit order.
 *   - tickets are issued in increasing order starting at 1; waiting for a ticket also completes every
 *     earlier one.  NTSCSIM_TICKET_ALL waits for everything submitted so far.
 *   - the rand() stream advances at SUBMIT time, in submit order (ntscsim_get_rng_pos() reflects it), so
 *     the noise is the one the synchronous loop draws.  ntscsim_field() / ntscsim_sync() on a ctx with
 *     fields in flight first wait for them; the other entry points must not be mixed in without
 *     ntscsim_wait(ctx, NTSCSIM_TICKET_ALL).
 *   - a launch happens when `depth` fields are pending, on ntscsim_flush(), or when a wait needs it; a
 *     change of width/height flushes first.
 * Host buffers (the same rules for ntscsim_submit() and ntscsim_submit422()): frames in PINNED memory are served
 * without a host memcpy -- DMA uploads, and the GPU writes the field rows straight into the caller's frame.  What
 * counts as pinned is deterministic (pin policy 1, the default):
 *   - memory the caller DECLARED with ntscsim_host_pin(ctx, base, len);
 *   - memory that is pinned already: ntscsim_host_alloc() / ntscsim_host_frame_alloc() (what the
 *     ntscsim_av_frame_get_buffer() helper of ntscsim_avframe.h backs an AVFrame with), hipHostMalloc, or the
 *     caller's own hipHostRegister -- the HIP runtime is asked (hipPointerGetAttributes), nothing is guessed;
 *   - buffers of at least min_pin_bytes (never below 64 KiB) that START ON A PAGE BOUNDARY and lie above the program
 *     break: registered in place on first sight (hipHostRegister, whole pages, cached per ctx).
 * Everything else -- ordinary malloc / av_malloc blocks, whatever the allocator (glibc, jemalloc, tcmalloc, a pool) --
 * goes through pinned staging rings: one memcpy each way, the delivery side on the engine's own copy threads.  Same
 * bytes either way.  Policy 2 (opt-in, glibc only: pin_caller_buffers = 2 / ntscsim_set_pin_policy) additionally
 * registers blocks that carry glibc's header of a chunk with a mapping of its own (IS_MMAPPED and a chunk size that
 * covers the frame) -- what av_frame_get_buffer() planes are while glibc's mmap threshold is below their size; it reads
 * the allocator's private word in front of the pointer and must not be used with another allocator.  Policy 0 stages
 * everything.
 * A pinned buffer must stay allocated until ntscsim_host_unpin() / ntscsim_destroy(): free()ing registered
 * memory while the registration lives is undefined (HIP).
 */
typedef struct ntscsim_submit_opts {
    uint32_t struct_size;          /* sizeof(ntscsim_submit_opts)                                       */
    int32_t  depth;                /* fields per launch, 1..4096; default 32                              */
    int32_t  slots;                /* fields that may be in flight (device frame ring), >= 2*depth;
                                      default 8*depth (0: the same).  ntscsim_submit() blocks (waits for the
                                      oldest launch) when the ring is full.  Waiting 4*depth fields behind
                                      the submits keeps uploads, kernels and deliveries of neighbouring
                                      launches overlapped (2*depth: ~10 % slower)                           */
    int32_t  lanes;                /* launches in flight side by side (own stream + scratch), 1..8; def. 3 */
    int32_t  pin_caller_buffers;   /* pin policy 0 | 1 | 2 (above); default 1                             */
    int32_t  _pad;
    size_t   min_pin_bytes;        /* default 256 KiB                                                     */
} ntscsim_submit_opts;
void ntscsim_submit_opts_init(ntscsim_submit_opts *o);
/* (Re)configure the engine; waits for everything in flight first.  Optional: the defaults apply otherwise. */
int  ntscsim_submit_configure(ntscsim_ctx *ctx, const ntscsim_submit_opts *o);

#define NTSCSIM_SUBMIT_SAME_SRC   0x10000u
#define NTSCSIM_SUBMIT_SRC_STABLE 0x20000u
#define NTSCSIM_TICKET_ALL        UINT64_MAX

/* Arguments as ntscsim_field(); `flags`: NTSCSIM_DESC_BOB | NTSCSIM_SUBMIT_*.  *ticket (may be NULL)
 * receives the ticket.  Errors are those of ntscsim_field(); a field that was refused consumes no ticket
 * and no rand() draws. */
int ntscsim_submit(ntscsim_ctx *ctx,
                   const uint8_t *src_bgra, int src_linesize, int src_interlaced, int src_tff,
                   uint8_t *dst_bgra, int dst_linesize,
                   int width, int height, unsigned field, uint64_t fieldno,
                   uint32_t flags, uint64_t *ticket);
/* Launch what is pending without waiting for it. */
int ntscsim_flush(ntscsim_ctx *ctx);
/* Block until `ticket` (and every earlier one) has been delivered.  Returns the first error of the
 * launches it had to complete (their fields are then lost), NTSCSIM_E_ARG for a ticket never issued. */
int ntscsim_wait(ntscsim_ctx *ctx, uint64_t ticket);
/* Declare [base, base + len) as the caller's own memory to pin: the pages it touches are registered with the GPU now
 * (hipHostRegister; memory that is pinned already is just noted) and every frame inside the range takes the
 * no-copy path from then on, wherever it starts.  The caller vouches that those pages hold nothing that is freed or
 * recycled while the declaration lives (an allocation of its own: a mmap, a large malloc block incl. its header page,
 * a frame pool).  Memory of the brk heap -- small malloc blocks, whose pages the allocator trims and recycles -- is refused
 * (NTSCSIM_E_ARG): a registration there makes the GPU fault sooner or later.
 * NTSCSIM_OK, NTSCSIM_E_ARG, NTSCSIM_E_HIP (registration refused; nothing changed). */
int ntscsim_host_pin(ntscsim_ctx *ctx, const void *base, size_t len);
/* Drop the registration that covers `base` (NULL: all of them) -- declared or made by the engine on its own --
 * after waiting for everything in flight: call before free()ing a frame buffer the engine has seen while the ctx
 * lives on. */
int ntscsim_host_unpin(ntscsim_ctx *ctx, const void *base);
/* Pinned host memory (hipHostMalloc, visible to every GPU of the process), page-aligned; NULL when it cannot be had.
 * No ctx needed: usable as the allocator behind av_buffer_create() (ntscsim_avframe.h). */
void *ntscsim_host_alloc(size_t bytes);
void  ntscsim_host_free(void *p);
/* The planes of one frame, laid out like av_frame_get_buffer(frame, align) does (linesize[k] = row_bytes[k] rounded
 * up to `align`, a power of two; every plane 64-byte aligned with 64 spare bytes), in ONE ntscsim_host_alloc()
 * block: data[k] / linesize[k] receive the planes, *base the block to hand to ntscsim_host_free(), *bytes (may be
 * NULL) its size. */
int   ntscsim_host_frame_alloc(int n_planes, const int *row_bytes, const int *rows, int align,
                               uint8_t **data, int *linesize, void **base, size_t *bytes);
/* Pin policy of this ctx's host-frame engines (0 | 1 | 2, see "Host buffers" above; default 1). */
int   ntscsim_set_pin_policy(ntscsim_ctx *ctx, int policy);
/* Counters since ntscsim_create(): [0] fields submitted, [1] launches, [2] source uploads, [3] uploads that
 * went through the staging ring, [4] fields delivered by the GPU into pinned caller frames, [5] fields
 * delivered through the staging ring, [6] live registrations, [7] submits that blocked on a full ring. */
void ntscsim_submit_stats(const ntscsim_ctx *ctx, uint64_t out[8]);

/*
 * The field loop (ffmpeg_ntsc.cpp:2202-2282) for a run of frames held in HOST memory, pipelined:
 * source frame j (j = 0..n_frames-1, BGRA, at src + j*src_frame_stride) produces output fields
 * first_fieldno + 2j and + 2j+1 (field parity (cur&1)^1, :2229), each into its own host frame at
 * dst + k*dst_frame_stride (k = 0..2*n_frames-1).  Destination frames start zeroed on the device;
 * with NTSCSIM_DESC_BOB in `flags` every row is then written except the one row bob leaves alone
 * (the frame-delay ring's, :2248).  Chunks of `chunk_frames` frames (<= 0: default 32) flow
 * through H2D copy | kernels | D2H copy on three HIP streams; the caller's buffers are pinned for
 * the duration of the call.  Synchronous; continues the ctx's rand() stream.
 */
int ntscsim_frames_host(ntscsim_ctx *ctx, const uint8_t *src, size_t src_frame_stride,
                        int src_linesize, int n_frames, uint8_t *dst, size_t dst_frame_stride,
                        int dst_linesize, int width, int height, uint64_t first_fieldno,
                        uint32_t flags, int chunk_frames);
/* `flags` bits for ntscsim_frames_host() in addition to NTSCSIM_DESC_*: deliver the output frames as
 * planar YUV instead of BGRA (each output frame at dst + k*dst_frame_stride holds the Y plane,
 * height rows of dst_linesize bytes, then U, then V, chroma rows of dst_linesize/2 bytes;
 * dst_linesize >= width, even). */
#define NTSCSIM_HOST_YUV420P 0x1000u
#define NTSCSIM_HOST_YUV422P 0x2000u

/*
 * Source scaling / pixel-format conversion before the field loop: what the tool does with
 * libswscale (sws_getContext(any format, any size -> BGRA W x H, SWS_BILINEAR), ffmpeg_ntsc.cpp
 * :573-583, sws_scale :603; ffmpeg_to_composite.cpp:1773).  libswscale is third-party code that
 * is not part of the reference tree: this is NOT a bit-clone of it (parity unpinned) but the
 * definition in csrc/ntsc_scale.hip -- pixel-centre-aligned bilinear resampling with 8-bit weights,
 * BT.601 limited-range YUV -> RGB, alpha 255 -- which tests/_libs.py restates and the GPU tests
 * compare bit for bit.
 */
#define NTSCSIM_SRC_BGRA    0
#define NTSCSIM_SRC_YUV420P 1
#define NTSCSIM_SRC_YUV422P 2
typedef struct ntscsim_scale_desc {
    const void *src_dev[3];     /* device pointers: BGRA uses [0]; planar YUV: Y, U, V            */
    void       *bgra_dev;       /* device pointer, BGRA frame of the call's width x height        */
    int32_t     src_linesize[3];
    int32_t     bgra_linesize;  /* >= 4*width, multiple of 4                                      */
    int32_t     src_width, src_height;   /* chroma planes: (w+1)/2 wide, (h+1)/2 (4:2:0) or h rows */
    int32_t     src_format;     /* NTSCSIM_SRC_*                                                  */
    int32_t     _pad;
} ntscsim_scale_desc;
int ntscsim_scale_to_bgra_device(ntscsim_ctx *ctx, const ntscsim_scale_desc *descs, int n,
                                 int width, int height, void *hip_stream);

/* Layout of one SOURCE frame in host memory for ntscsim_frames_host_scaled(): plane p starts
 * `plane_offset[p]` bytes into the frame and has rows of `linesize[p]` bytes. */
typedef struct ntscsim_host_source {
    int32_t format;             /* NTSCSIM_SRC_*                                                  */
    int32_t width, height;
    int32_t linesize[3];
    size_t  plane_offset[3];
    size_t  frame_bytes;        /* bytes of one frame that must travel to the device              */
} ntscsim_host_source;
/* ntscsim_frames_host() for sources of any size in BGRA / YUV420P / YUV422P: every chunk is uploaded
 * in its own format, converted to BGRA width x height on the GPU (ntscsim_scale_to_bgra_device) and
 * fed to the field loop; everything else as ntscsim_frames_host(). */
int ntscsim_frames_host_scaled(ntscsim_ctx *ctx, const ntscsim_host_source *source, const uint8_t *src,
                               size_t src_frame_stride, int n_frames, uint8_t *dst,
                               size_t dst_frame_stride, int dst_linesize, int width, int height,
                               uint64_t first_fieldno, uint32_t flags, int chunk_frames);

/* ---- several GPUs: a pool of contexts (BASELINE north_star: "partition the input stream
 * frame-round-robin") ------------------------------------------------------------------------------------
 * The loop :2202-2282 carries nothing from field to field but the position of the process-wide rand() stream,
 * and the draws per composite_layer() call do not depend on the pixels (ntscsim_rng_calls_per_field), so every
 * field's position is a closed form.  ntscsim_pool_frames_host() is ntscsim_frames_host() dealt over N
 * contexts: block b of `block` frames (default 32; ntscsim_pool_set_block) goes to context b mod N, every context
 * runs its own upload | kernels | download pipeline on its blocks from a host thread of its own, no data moves
 * between GPUs, and the frames are byte-identical to one context processing the whole run.  `devices` lists the
 * HIP ordinals, one context each (an ordinal may repeat: several contexts on one GPU -- how the tests run on a
 * one-GPU box); devices = NULL: the first n_devices visible GPUs (n_devices = 0: all of them).  The pool keeps its
 * own rand() position, advanced by every successful call like the ctx's.  One thread at a time per pool. */
typedef struct ntscsim_pool ntscsim_pool;
int  ntscsim_pool_create(const ntscsim_params *p, const int *devices, int n_devices, ntscsim_pool **out);
void ntscsim_pool_destroy(ntscsim_pool *pool);
int  ntscsim_pool_size(const ntscsim_pool *pool);
ntscsim_ctx *ntscsim_pool_ctx(ntscsim_pool *pool, int i);       /* context i, e.g. for ntscsim_set_mode() */
int  ntscsim_pool_set_block(ntscsim_pool *pool, int block_frames);
uint64_t ntscsim_pool_get_rng_pos(const ntscsim_pool *pool);
void ntscsim_pool_set_rng_pos(ntscsim_pool *pool, uint64_t pos);
const char *ntscsim_pool_last_error(const ntscsim_pool *pool);
int  ntscsim_pool_frames_host(ntscsim_pool *pool, const uint8_t *src, size_t src_frame_stride, int src_linesize,
                              int n_frames, uint8_t *dst, size_t dst_frame_stride, int dst_linesize,
                              int width, int height, uint64_t first_fieldno, uint32_t flags, int chunk_frames);

/* ---- batched, device-resident form (what the field loop :2202-2282 becomes) -------------- */

#define NTSCSIM_RNG_AUTO  UINT64_MAX   /* rng_pos: continue after the previous descriptor      */

typedef struct ntscsim_field_desc {
    const void *src_dev;      /* device pointer, BGRA frame, width x height                    */
    void       *dst_dev;      /* device pointer, BGRA frame, width x height                    */
    int32_t     src_linesize; /* bytes, >= 4*width, multiple of 4                              */
    int32_t     dst_linesize;
    uint32_t    field;        /* 0 | 1 : rows field, field+2, ...                              */
    uint32_t    flags;        /* NTSCSIM_DESC_* : bit0 src interlaced_frame, bit1 src
                                 top_field_first, bit8 bob line doubling (:2233-2257)          */
    uint64_t    fieldno;      /* `current` in the reference's loop                             */
    uint64_t    rng_pos;      /* rand() stream position at entry, or NTSCSIM_RNG_AUTO          */
} ntscsim_field_desc;

#define NTSCSIM_DESC_INTERLACED 1u
#define NTSCSIM_DESC_TFF        2u
#define NTSCSIM_DESC_BOB        0x100u   /* bob writes the OTHER field's rows of dst_dev: give every
                                            descriptor of a batch its own destination frame */

/*
 * Process `n` independent fields of one geometry.  All pointers in `descs` are DEVICE pointers
 * on the ctx's device; the descriptor array itself is host memory.  Work is enqueued on
 * `hip_stream` (a hipStream_t passed as void*, NULL = the ctx's own stream) and the call returns
 * without synchronising; scratch owned by the ctx is reused by the next call on the same ctx,
 * which is stream-ordered after this one.  descs[0].rng_pos == NTSCSIM_RNG_AUTO continues from
 * the ctx position; after a successful call the ctx position is the end of the last descriptor
 * (a call that fails leaves it where it was).
 * The descriptors of one call run concurrently: two of them may share a destination frame only as
 * its two fields (different `field`, neither with NTSCSIM_DESC_BOB); any other sharing is refused
 * with NTSCSIM_E_ARG.
 */
int ntscsim_fields_device(ntscsim_ctx *ctx, const ntscsim_field_desc *descs, int n,
                          int width, int height, void *hip_stream);

/*
 * Prepared batch: the same work as ntscsim_fields_device(), split into "validate the descriptors,
 * derive each field's rand() window, upload the records" (once) and "enqueue the kernel chain"
 * (any number of times, e.g. a clip that is re-rendered, or a ring of frame buffers that is
 * refilled in place).  The descriptors' pointers must stay valid.  A batch uses its ctx's scratch:
 * runs on one ctx are stream-ordered by the caller; use one ctx per stream to overlap batches.
 * ntscsim_batch_create() does not move the ctx's rand() position; every successful
 * ntscsim_batch_run() leaves it at the end of the batch's last descriptor.
 */
typedef struct ntscsim_batch ntscsim_batch;
int  ntscsim_batch_create(ntscsim_ctx *ctx, const ntscsim_field_desc *descs, int n,
                          int width, int height, ntscsim_batch **out);
int  ntscsim_batch_run(ntscsim_batch *batch, void *hip_stream);
void ntscsim_batch_destroy(ntscsim_batch *batch);

/* ---- the 8-bit YUV422P sibling: ffmpeg_to_composite.cpp --------------------------------------
 * One descriptor == one iteration of the loop at ffmpeg_to_composite.cpp:1783-1800:
 *   render_field(dst, src, field, ...) :1784   (skipped when src_dev[0] == NULL)
 *   black_key_feedback(dst, flt, ...)  :1787   (when params.black_key_level_feedback >= 0 and
 *                                               flt_dev[0] != NULL)
 *   composite_video_process(dst, field, fieldno) :1790, IN PLACE on the rows of `field` of the
 *                                               YUV422P frame dst (skipped with NTSCSIM_422_NOCOMP,
 *                                               the tool's -nocomp :1789)
 * Create the ctx from ntscsim_params_init_to_composite()/..._parse_argv_to_composite() parameters.
 * The reference's Y/C separator reads two bytes past the end of each luma row (`Y[x+2]`, :496).  This
 * implementation has the same memory semantics: for the last two positions of a row it reads the
 * caller's bytes dst[0][y*linesize + width], [.. + width + 1] when they lie inside the luma plane
 * (height * linesize bytes) -- with linesize < width + 2 those are the first bytes of the NEXT row,
 * which belongs to the other field -- and the value 16 otherwise (the frame's last row).
 * Consequences for batching: (1) fields that share a feedback frame must not be in the same batch
 * (frame-to-frame recurrence); (2) when dst_linesize[0] < width + 2, the two fields of one destination
 * frame must not be in the same batch either: one field's kernel would read bytes the other rewrites
 * in place, whereas the tool processes them one after the other (:1783-1800).  Such a batch is refused
 * with NTSCSIM_E_ARG; submit the fields in separate, stream-ordered calls (or pad the rows).
 */
#define NTSCSIM_422_INTERLACED 1u   /* src->interlaced_frame                                     */
#define NTSCSIM_422_TFF        2u   /* src->top_field_first                                      */
#define NTSCSIM_422_SRC420     4u   /* decoder format is YUV420P (:1005): src chroma is H/2 tall */
#define NTSCSIM_422_SECOND     8u   /* field_number - src_pts >= ticks_per_frame/2 (:1035)       */
#define NTSCSIM_422_NOCOMP     16u  /* -nocomp: render only                                      */

typedef struct ntscsim_field422_desc {
    void       *dst_dev[3];         /* Y, U, V of the YUV422P output frame (device pointers)     */
    const void *src_dev[3];         /* source frame for render_field, or NULL                    */
    void       *flt_dev[3];         /* black-key feedback frame, or NULL                         */
    int32_t     dst_linesize[3], src_linesize[3], flt_linesize[3];
    int32_t     src_height;         /* source frame height (its width is the output width)       */
    uint32_t    field;
    uint32_t    flags;              /* NTSCSIM_422_*                                             */
    uint32_t    _pad;
    uint64_t    fieldno;
    uint64_t    rng_pos;            /* or NTSCSIM_RNG_AUTO                                       */
} ntscsim_field422_desc;

int ntscsim_fields422_device(ntscsim_ctx *ctx, const ntscsim_field422_desc *descs, int n,
                             int width, int height, void *hip_stream);

/* Prepared batch of the same call (like ntscsim_batch_create/run/destroy above): descriptors are
 * validated, their rand() windows derived and everything uploaded once; every ntscsim_batch422_run()
 * is then only the kernel launches.  The descriptors' frame pointers must stay valid; runs on one ctx
 * are stream-ordered by the caller (one ctx per stream to overlap batches). */
typedef struct ntscsim_batch422 ntscsim_batch422;
int  ntscsim_batch422_create(ntscsim_ctx *ctx, const ntscsim_field422_desc *descs, int n,
                             int width, int height, ntscsim_batch422 **out);
int  ntscsim_batch422_run(ntscsim_batch422 *batch, void *hip_stream);
void ntscsim_batch422_destroy(ntscsim_batch422 *batch);

/*
 * The same loop iteration on HOST frames -- the drop-in for the tool's own call sequence
 *     render_field(output_avstream_video_frame, output_avstream_video_input_frame, field, video_field, tgt_pts);  :1784
 *     if (black_key_level_feedback >= 0) black_key_feedback(frame, output_avstream_video_filter_frame, ...);      :1787
 *     if (enable_composite_emulation) composite_video_process(frame, field, video_field);   :1790 (signature :629)
 *     output_frame(frame, video_field, field);          :1793 / :1796 -- its pixel work, the copy loops :1177-1236
 * of ffmpeg_to_composite.cpp:1783-1800, on the planes an AVFrame holds (data[0..2], linesize[0..2]).
 * ntscsim_field422() is synchronous (0.31-0.34 ms per 720x480 iteration: the streamed kernels of the -vhs family run as
 * four wavefront roles of one workgroup per 63 rows for such short launches); ntscsim_submit422() returns at once with a ticket and ntscsim_wait(ctx,
 * ticket) / ntscsim_flush(ctx) (above) complete / launch it -- a ctx serves ONE of the two tools, its tickets
 * come from one sequence.  include/ntscsim_avframe.h wraps both for real AVFrames; INTEGRATION.md section 5
 * has the patch of the loop and host/field_loop422.cpp is that loop in C++.
 *
 * Contract (what makes submits + waits byte-identical to the same sequence of synchronous calls, and those to
 * the tool's own calls on the same buffers; rand() position included):
 *   - `frame` is the tool's ONE persistent YUV422P frame, processed in place: the rows of `field` are written
 *     (render, key, composite emulation), the separator's read of two bytes behind each luma row (:496) sees
 *     the caller's own bytes as in ntscsim_fields422_device().  The engine keeps device copies of the frames it
 *     has seen: between calls the caller must not modify `frame` / `filter` -- or pass NTSCSIM_SUBMIT422_DIRTY
 *     with the next call (everything in flight is then delivered first and the frames are read again).
 *   - `src` (render_field's source: output_width wide, src_height rows, 4:2:2 or with NTSCSIM_422_SRC420 4:2:0)
 *     is SNAPSHOTTED by the call; data[0] == NULL: no render_field (the frame is processed as it stands).
 *     NTSCSIM_SUBMIT_SAME_SRC: src still holds the frame of the previous submit422 (second field of a frame).
 *   - `filter`: the black-key feedback frame, used when params.black_key_level_feedback >= 0 and data[0] != NULL;
 *     its rows of `field` are updated like the tool's.
 *   - `out` (data[0] == NULL: no output_frame call): the frame output_frame() hands to its encoder -- YUV422P
 *     (NTSCSIM_OUT422_BOB422) or YUV420P, chroma planes (height + 1) / 2 rows -- written per `out_mode` from
 *     the rows of `out_field` (the tool passes video_field's parity, or with -vi the previous field's, :1792-1796).
 *     Rows inside the planes only: the interlaced 4:2:0 repack's one chroma row past the plane for a height of
 *     2 mod 4 (:1215-1223, lands in the AVFrame's padding in the tool) is not written.
 *   - results are in the caller's memory when ntscsim_wait(ticket) returns, delivered in submit order; frame,
 *     filter and out must not be touched between submit and wait.  With one persistent `frame` and several
 *     fields in flight the frame holds the LAST delivered version of each row -- consume `out` (a ring of
 *     encoder frames) rather than `frame`, as the tool does.
 *   - iterations that render from a source into rows padded by >= 2 bytes, without black-key feedback, run
 *     `depth` at a time (default 32); everything else (no source, linesize[0] < width + 2, feedback: a
 *     frame-to-frame recurrence, an interlaced repack whose other field was not submitted right before it)
 *     runs one at a time, in order -- exact, and as slow as the synchronous call.
 *   - ntscsim_sync() delivers everything in flight first; the device-pointer entry points
 *     (ntscsim_fields422_device() ...) must not be mixed in without ntscsim_wait(ctx, NTSCSIM_TICKET_ALL).
 *     ntscsim_host_unpin(ctx, base) drops the engine's registration of a frame the caller is about to free.
 * Host buffers: the rules of ntscsim_submit() above ("Host buffers"; the floor for pinning in place is 64 KiB per
 * plane here).  Results are written into pinned planes by the delivery kernels (the source is snapshotted by one memcpy
 * into a pinned ring: cheaper for the calling thread than waiting for a DMA out of its planes); planes that are not
 * pinned are filled from the staging ring by the engine's copy threads before ntscsim_wait() returns.  Same bytes
 * either way.
 * Errors: as ntscsim_fields422_device(); a refused call consumes no ticket and no rand() draws.
 */
typedef struct ntscsim_frame422 {    /* AVFrame::data[0..2] / linesize[0..2] of a planar YUV frame in host memory */
    uint8_t *data[3];
    int32_t  linesize[3];
    int32_t  _pad;
} ntscsim_frame422;

typedef struct ntscsim_loop422 {
    uint32_t struct_size;            /* sizeof(ntscsim_loop422)                                          */
    int32_t  width, height;          /* output_width x output_height: the size of frame, filter, out     */
    int32_t  src_height;             /* rows of src (its width is `width`)                               */
    ntscsim_frame422 frame;          /* output_avstream_video_frame                                      */
    ntscsim_frame422 src;            /* output_avstream_video_input_frame, or data[0] == NULL            */
    ntscsim_frame422 filter;         /* output_avstream_video_filter_frame, or data[0] == NULL           */
    ntscsim_frame422 out;            /* output_avstream_video_bob_frame / the encoder frame, or NULL     */
    uint32_t field;                  /* (video_field & 1) ^ 1                                            */
    uint32_t flags;                  /* NTSCSIM_422_*                                                    */
    uint32_t out_mode;               /* NTSCSIM_OUT422_* (below)                                         */
    uint32_t out_field;              /* output_frame()'s `field` argument                                */
    uint64_t fieldno;                /* video_field                                                      */
} ntscsim_loop422;

#define NTSCSIM_SUBMIT422_DIRTY 0x40000u   /* the caller changed frame / filter since the engine last saw them */

int ntscsim_field422(ntscsim_ctx *ctx, const ntscsim_loop422 *it);
int ntscsim_submit422(ntscsim_ctx *ctx, const ntscsim_loop422 *it, uint32_t submit_flags, uint64_t *ticket);
/* Optional: iterations per launch (1..4096, default 32) and iterations that may be in flight (0: 4 * depth;
 * >= 2 * depth + 2).  Waits for everything in flight first. */
int ntscsim_submit422_configure(ntscsim_ctx *ctx, int depth, int slots);
/* [0] submitted [1] launches [2] source uploads [3] batched iterations [4] one-at-a-time iterations
 * [5] frame (re)uploads [6] iterations delivered by kernels into pinned caller planes [7] submits that blocked on a full ring */
void ntscsim_submit422_stats(const ntscsim_ctx *ctx, uint64_t out[8]);


/*
 * The pixel work of output_frame() ffmpeg_to_composite.cpp:1131, lines :1177-1236: line-double
 * ("bob") the rows of `field` of a processed YUV422P frame into the frame handed to the encoder.
 *   NTSCSIM_OUT422_BOB422         use_422_colorspace, field rate (:1177-1196): bob is YUV422P
 *   NTSCSIM_OUT422_BOB420         4:2:0, field rate (:1197-1236): bob is YUV420P; chroma row y/2 is
 *                                 the frame's chroma row sy(y) for even y (decimation by copy)
 *   NTSCSIM_OUT422_INTERLACED420  4:2:0 with -interlaced (:1202, :1215-1223): luma copied 1:1,
 *                                 chroma rows interleaved per field
 * (4:2:2 interlaced output encodes the frame as is, :1158 -- nothing to do.)  The bob frame's
 * chroma planes need (height+1)/2 rows in the 4:2:0 modes -- and ONE MORE in NTSCSIM_OUT422_INTERLACED420 when
 * height is 2 mod 4: the tool's repack writes chroma row (y & 1) + ((y & ~3) >> 1) for y = height - 1 then, one
 * row past the plane (in the tool that lands in the frame's padding); the same row is written here.
 * Stream-ordered like the calls above.
 */
#define NTSCSIM_OUT422_BOB422        0u
#define NTSCSIM_OUT422_BOB420        1u
#define NTSCSIM_OUT422_INTERLACED420 2u
#define NTSCSIM_OUT422_FRAME         3u   /* 4:2:2 with -interlaced: the tool encodes the processed frame itself (:1158);
                                             a 1:1 copy, for callers that keep several fields in flight (ntscsim_submit422) */

typedef struct ntscsim_out422_desc {
    const void *frame_dev[3];       /* processed YUV422P frame (the dst of ntscsim_fields422_device) */
    void       *bob_dev[3];         /* encoder frame: YUV422P or YUV420P                         */
    int32_t     frame_linesize[3], bob_linesize[3];
    uint32_t    field;              /* output_frame()'s `field` argument (:1793, :1796)          */
    uint32_t    mode;               /* NTSCSIM_OUT422_*                                          */
} ntscsim_out422_desc;

int ntscsim_output422_device(ntscsim_ctx *ctx, const ntscsim_out422_desc *descs, int n,
                             int width, int height, void *hip_stream);

/* ---- encoder-side colour conversion (output side) ------------------------
 * What the tool does with sws_scale(BGRA -> codec pix_fmt) at ffmpeg_ntsc.cpp:2266 (pix_fmt chosen
 * at :1998: YUV420P, or YUV422P with -422).  libswscale is a third-party dependency that is not
 * in the reference tree, so this is NOT a bit-clone of it ("parity unpinned"): BT.601 limited
 * range, 15-bit fixed point, Y = (8414 R + 16519 G + 3208 B + (16<<15) + (1<<14)) >> 15; Cb/Cr from
 * the channel sums of each 2x1 (4:2:2) or 2x2 (4:2:0; an odd last row counts twice) block with
 * coefficients (-4864,-9527,14392) / (14392,-12060,-2331), +128, rounded.  Width must be even.
 * Converting on the GPU shrinks the device-to-host payload from 4 to 1.5 / 2 bytes per pixel. */
#define NTSCSIM_PIX_YUV420P 0
#define NTSCSIM_PIX_YUV422P 1
typedef struct ntscsim_yuv_desc {
    const void *bgra_dev;           /* BGRA frame (e.g. a bob frame written by ntscsim_fields_device) */
    void       *yuv_dev[3];         /* Y, U, V planes; chroma (height+1)/2 rows for 4:2:0          */
    int32_t     bgra_linesize;
    int32_t     yuv_linesize[3];
} ntscsim_yuv_desc;
int ntscsim_bgra_to_yuv_device(ntscsim_ctx *ctx, const ntscsim_yuv_desc *descs, int n,
                               int width, int height, int pix_fmt, void *hip_stream);
/* draws of one composite_video_process() call (chroma noise runs at width/2 samples per row) */
uint64_t ntscsim_rng_calls_per_field_422(const ntscsim_params *p, int width, int height,
                                         unsigned field);

/* Block until everything enqueued by this ctx has finished. */
int ntscsim_sync(ntscsim_ctx *ctx);

/* Which kernels a launch of DEVICE-RESIDENT fields takes (ntscsim_fields_device / ntscsim_fields422_device; the host-frame
 * entry points -- ntscsim_field, ntscsim_submit, ntscsim_field422, ntscsim_submit422 -- decide for themselves):
 *   NTSCSIM_FORM_THROUGHPUT (default)  the one-launch chain: one wavefront per 63 rows walks the whole pipeline -- what long
 *                                      batches want (1.3 us per field at 600 fields, 0.42 ms for a launch of ANY length up to 32);
 *   NTSCSIM_FORM_LATENCY               launches of up to 64 fields run the chain as wavefront ROLES of one workgroup per 63 rows
 *                                      (csrc/ntsc_pipe.hip, k422_pipe): 0.17 ms for a launch of up to 32 fields -- a frame at a
 *                                      time between a decoder and an encoder on the same GPU.  Same bytes either way; switch
 *                                      sets the role kernels do not cover, and longer launches, keep the throughput form.
 * NTSCSIM_OK, NTSCSIM_E_ARG. */
enum { NTSCSIM_FORM_THROUGHPUT = 0, NTSCSIM_FORM_LATENCY = 1 };
int ntscsim_set_launch_form(ntscsim_ctx *ctx, int form);

/* Kernel timing from hipEvents recorded on the launch stream around each stage of every
 * ntscsim_fields_device() call made while profiling is on.  ntscsim_get_timings_ms() waits for
 * those calls, returns the SUMS since the previous query (index 0 setup kernels, 1 encode
 * kernel, 2 decode kernel, 3 whole call incl. descriptor upload and bob) and the number of calls. */
void ntscsim_set_profiling(ntscsim_ctx *ctx, int on);
int  ntscsim_get_timings_ms(ntscsim_ctx *ctx, float out_ms[4], int *n_calls);

/* Debug tap used by the stage-level parity tests: copy the composite-signal plane (int32, one
 * value per pixel, row-major [n][L][W]) that the last ntscsim_fields_device() call left in
 * scratch -- the signal after chroma_into_luma + pre-emphasis + luma noise (ffmpeg_ntsc.cpp:1611-1644),
 * before head switching -- into host memory. */
int ntscsim_debug_read_composite(ntscsim_ctx *ctx, int32_t *out, size_t out_elems);

/* Test hook: shorten the noise-accumulator warm-up of k_row_states (default 64 / 128 draws) so
 * that the exact serial-replay fallback is exercised.  Results must not change. */
void ntscsim_debug_set_warmup(ntscsim_ctx *ctx, int luma_draws, int chroma_draws);

/* Test hook: always launch the GENERIC kernels (options read at run time) instead of the PRESET
 * specialisations chosen for the default / -vhs parameter sets.  Results must not change. */
void ntscsim_debug_force_generic(ntscsim_ctx *ctx, int on);

/* Test hook: bit 0 keeps the template-specialised PRESET kernels (k_encode / k_decode) where the
 * hand-tuned ones (csrc/ntsc_encode_fast.hip, csrc/ntsc_decode_fast.hip) would run; bit 1 runs
 * the hand-tuned VHS decoder as two launches (VCR half -> second composite plane -> TV half)
 * instead of one; for the YUV422P tool bit 0 selects the twelve-sweep kernel, bit 1 the run-time-switch
 * form of the four-sweep kernel, bit 2 the four-sweep form of the preset kernel (instead of sweep A + one
 * streamed pass).  Results must not change. */
void ntscsim_debug_no_fast_decode(ntscsim_ctx *ctx, int on);

/* Which kernel forms the last ntscsim_fields_device() / ntscsim_fields422_device() / batch run on this
 * ctx enqueued, in launch order, as a ';'-separated list of names as they appear in a rocprofv3 kernel
 * trace without the argument list (e.g. "k_field_setup;k_row_states;k_encode_fast<double>;
 * k_decode_fast<true,double>").  Returns the length of the full list (it is truncated to cap-1
 * characters), or NTSCSIM_E_ARG.  The parity tests assert on it so that a specialised form cannot
 * silently stop being the one that runs. */
int ntscsim_debug_last_kernels(const ntscsim_ctx *ctx, char *out, size_t cap);

/* Test hook (pure host arithmetic, no ctx): 1 when a batch of n_fields fields of width x height may take
 * the hand-tuned kernels as far as the size of its composite plane is concerned -- they address the
 * plane with 32-bit buffer offsets, and the displaced index of the head switch must not wrap around
 * 2^32 back into the plane.  head_switching: 0 = off, 1 = displacement of at most width/10 samples,
 * 2 = any displacement (the wrap-around form: a margin of a whole 1.1-width window either way).
 * 0 = the generic kernels run. */
int ntscsim_debug_fast_plane_ok(int n_fields, int width, int height, int head_switching);

/* Test hook: what the synchronous call ntscsim_field() did since the ctx was created -- [0] calls, [1] calls whose SOURCE
 * frame was read where it is (pinned memory the GPU can address, 16-byte aligned rows: no upload), [2] calls whose
 * DESTINATION frame was written where it is (no download), [3] calls (of either tool's synchronous entry point) that
 * found their setup kernel already run, launched ahead of time by the call before. */
void ntscsim_debug_field_stats(const ntscsim_ctx *ctx, uint64_t out[4]);

/* ---- the raw-composite decoder: ffmpeg_raw28ntsc.cpp ---------------
 * The tool reads 8-bit composite video sampled at 8 x fsc (28.636 MHz, e.g. a cxadc capture) and
 * renders one grey-scale BGRA frame of (scanline_samples + 1 & ~1) x 262 per field:
 *   hsync_dc_proc() :556-594 on every sample (sync-tip tracking, low-passed sync detector);
 *   composite_layer() :601-849 per field: vertical-sync search, black / white calibration on the
 *   equalising pulses, then per scanline: level equalisation, delay-4 comb luma / chroma split,
 *   rendering, horizontal re-sync;  field loop main() :1006-1038.
 * A decoder object is not re-entrant (one thread at a time); calls are synchronous.
 * One ntscsim_raw28_decode*() call = one run of the tool on one input file: the decoder state
 * starts from the tool's initial state every call (an empty capture gives 0 fields).  Long captures and
 * pipes: ntscsim_raw28_stream_*() below.  Results are bit-identical to the tool's, including
 * where its calibration sums run past the buffered part of the capture into stale or never-filled
 * records of its sample buffer (:655-676); only reads past the END of that array (undefined in the
 * tool) are defined here, as zero records.
 */
typedef struct ntscsim_raw28_opts {
    uint32_t struct_size;            /* sizeof(ntscsim_raw28_opts)                               */
    uint32_t _pad0;
    double   sample_rate;            /* -s: 0 = "ntsc28" = 315e6*8/88; "40mhz" = 40e6; else Hz   */
    int32_t  mark_sync;              /* -marksig :458                                            */
    int32_t  disable_sync;           /* -nosig   :467                                            */
    int32_t  disable_wp_equ;         /* -nowequ  :464                                            */
    int32_t  show_subcarrier;        /* -showsc  :473                                            */
    int32_t  disable_subcarrier;     /* -nosc    :470                                            */
    int32_t  disable_equalization;   /* -noequ   :461                                            */
} ntscsim_raw28_opts;
typedef struct ntscsim_raw28 ntscsim_raw28;

void ntscsim_raw28_opts_init(ntscsim_raw28_opts *o);
/* mirror of parse_argv() :442-520 (-marksig -noequ -nowequ -nosig -nosc -showsc -s <rate>; -width,
 * -i, -o, -422, -420, -inntsc are accepted and have no effect on the decoder: preset_NTSC() :395
 * overrides the width after parsing).  Returns NTSCSIM_OK, NTSCSIM_E_HELP or NTSCSIM_E_FLAG (the tool's
 * "return 1"). */
int  ntscsim_raw28_parse_argv(ntscsim_raw28_opts *o, int argc, const char *const *argv, int start);
int  ntscsim_raw28_geometry(const ntscsim_raw28_opts *o, int *width, int *height, int *scanline_samples);
int  ntscsim_raw28_create(const ntscsim_raw28_opts *o, int device, ntscsim_raw28 **out);
void ntscsim_raw28_destroy(ntscsim_raw28 *dec);
const char *ntscsim_raw28_last_error(const ntscsim_raw28 *dec);
/* Decode a whole capture.  `capture` is host memory (ntscsim_raw28_decode) or device memory
 * (ntscsim_raw28_decode_device), n_samples < 2^32.  frames_dev: device memory for max_fields BGRA
 * frames, frame k at frames_dev + k * frame_stride, rows of `linesize` bytes (>= 4 * width).  Rows
 * the tool does not render are zero, alpha is 0 (:757-775).  Synchronous.  *n_fields = fields the
 * tool would have produced (its loop stops when fewer than 256 scanlines remain), capped at
 * max_fields. */
int  ntscsim_raw28_decode(ntscsim_raw28 *dec, const uint8_t *capture_host, size_t n_samples,
                          void *frames_dev, size_t frame_stride, int linesize, int max_fields, int *n_fields);
int  ntscsim_raw28_decode_device(ntscsim_raw28 *dec, const void *capture_dev, size_t n_samples,
                                 void *frames_dev, size_t frame_stride, int linesize, int max_fields, int *n_fields);
/* The same decoder on a stream of any length, push by push (a capture file read in pieces, a pipe): the
 * tool itself keeps a window of 2048 scanlines (:353) and never needs the whole capture.
 *   ntscsim_raw28_stream_reset()  start a new stream (the state of a fresh run of the tool);
 *   ntscsim_raw28_stream_push()   `n` more samples (host memory, or device memory with on_device != 0;
 *       n = 0 is allowed), `final` != 0 when the stream ends with them.  Decodes, into frames_dev as in
 *       ntscsim_raw28_decode(), every field whose 2048-scanline window is complete -- the tool blocks in
 *       read() until its buffer is full, so a field is produced exactly when the tool could produce it --
 *       up to max_fields of them; fields held back by max_fields come out of the following pushes.  After
 *       the final push: the fields the tool produces before its "fewer than 256 scanlines left" stop.
 *       *n_fields = fields written by this call.  The sequence of all fields is bit-identical to one
 *       ntscsim_raw28_decode() of the concatenated samples, whatever the push sizes.
 * The decoder keeps only what it can still need (about two windows of samples on the device), so the
 * stream may be far longer than 2^32 samples; one push must be shorter than that.  A push that fails after it has
 * taken its samples (NTSCSIM_E_HIP, NTSCSIM_E_NOMEM, NTSCSIM_E_INTERNAL) leaves the stream half advanced: it is
 * then refused (NTSCSIM_E_ARG) until ntscsim_raw28_stream_reset(); argument errors leave the stream as it was.
 * ntscsim_raw28_decode*() are a reset followed by one final push. */
int  ntscsim_raw28_stream_reset(ntscsim_raw28 *dec);
int  ntscsim_raw28_stream_push(ntscsim_raw28 *dec, const void *samples, size_t n, int on_device, int final,
                               void *frames_dev, size_t frame_stride, int linesize, int max_fields, int *n_fields);
/* blank_level, white_level (:553-554) and the stream position (total_count_src) after the last call */
int  ntscsim_raw28_get_levels(const ntscsim_raw28 *dec, double *blank, double *white, uint64_t *read_pos);
/* Test hooks.  warm-up scanlines of the speculative front end (default 112; 0 forces every chunk
 * through the exact repair rounds) and chunk size in samples (default 4096); results must not
 * change.  stats: [0] front-end repair rounds, [1] chunks repaired, [2] comb-tail rounds (after the serial first guess: 1 when it was the fixed point),
 * [3] sync runs, [4] rendered scanlines, [5] calibration pulses of the last call; [6..11] wall-clock
 * microseconds of its phases: front end, run extraction, sync walk (level calibration, comb tails and rendering of
 * the groups of fields it has finished run on the GPU behind it), queuing the same for the last two groups, waiting
 * for the GPU to finish them, redoing the comb tails by rounds + a second rendering (0 unless a group's first guess
 * of the tails did not settle); [12] calibration pulses whose sums ran past the buffered
 * stream, [13] never-filled records among them; [14] compactions of the device buffer (streams),
 * [15] the most samples the device buffer ever held.  Counters accumulate over the pushes of a stream.
 * (A decoder whose front end leaves more than one chunk link in a hundred open after its speculative pass -- a noisy
 * source -- lengthens the warm-up of its later passes by 16 scanlines, up to 64: speed only.)
 * Environment (read by ntscsim_raw28_create, developer / test switches; results never depend on them):
 * NTSCSIM_RAW28_SEG = samples the front end takes per segment (default 2^29: 8 bytes of scratch per sample),
 * NTSCSIM_RAW28_CHUNKS = chunk count of its second sweep on long streams, NTSCSIM_RAW28_EXACT = scanlines at the end of the second sweep's warm-up that are walked sample
 * by sample (default 30; the ones before are taken in closed form where that is known to be safe; >= 112: all),
 * NTSCSIM_RAW28_LANES = chunks per wavefront of that sweep (default 16), NTSCSIM_RAW28_GROUP = fields per group of the
 * back half's pipeline behind the sync walk (default 160), NTSCSIM_RAW28_TAILROUNDS = 1: test hook, take the path of
 * comb tails whose first guess did not settle. */
void ntscsim_raw28_debug_set_speculation(ntscsim_raw28 *dec, int warm_lines, int chunk_samples);
void ntscsim_raw28_debug_stats(const ntscsim_raw28 *dec, int64_t out[16]);
/* Debug tap (host only, no GPU): the chunk length the front end's second sweep picks for a stream of `target_samples` x
 * 16,384 samples -- `subchunks` pieces of whole 64-sample blocks near a whole number of scanlines */
void ntscsim_raw28_debug_pick_chunk(double scanline_samples, double target_samples, int *chunk, int *subchunks);
/* Debug tap: the front end's hsync_dc_raw of every sample of the last call, to host memory */
int  ntscsim_raw28_debug_read_front(ntscsim_raw28 *dec, uint8_t *hsync_dc_raw, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* NTSCSIM_H */
