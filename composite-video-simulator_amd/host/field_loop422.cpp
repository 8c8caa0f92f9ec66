// field_loop422.cpp -- the YUV422P tool's loop (ffmpeg_to_composite.cpp:1783-1800) with its four calls replaced by ONE,
// on AVFrame-shaped pageable host buffers; what a maintainer gets from INTEGRATION.md section 5 (synchronous:
// ntscsim_field422_avframe) and section 5b (asynchronous: ntscsim_submit422_avframe + ntscsim_wait `lag` fields later).
//
//   field_loop422 [ffmpeg_to_composite switches] [--mode sync|submit] [--fields N] [--depth K] [--lag G] [--warmup N]
//                 [--hash 0|1] [--height H] [--align A] [--alloc malloc|pinned|pool|mmap] [--pin-policy 0|1|2]
//                 [--mmap-threshold BYTES]
//
// The loop owns, like the tool: ONE decoded-and-scaled input frame (output_avstream_video_input_frame, rewritten by a
// memcpy per source frame -- the stand-in for sws_scale :1770-1778), ONE persistent processing frame
// (output_avstream_video_frame), the filter frame when -bkey-feedback is given, and -- where the tool has ONE bob
// frame that it encodes synchronously inside output_frame() -- a ring of `lag + 1` encoder frames, consumed `lag`
// fields behind the submits (--hash 1: FNV-1a over each, the stand-in for the encoder).  Frames are allocated like
// av_frame_get_buffer(f, A) does: linesize = width rounded up to A (default 32), so 720 -> 736 (padded rows: the
// batched path) and 704 -> 704 (tight rows: one iteration at a time).  -vi / -422 as in the tool (:1792-1797, :1158).
// Where the plane memory comes from decides how the pixels travel (include/ntscsim.h "Host buffers"; nothing here is a
// guess about the allocator):
//   --alloc malloc   posix_memalign like av_malloc's (the tool unpatched): ordinary heap blocks -> the engine's pinned staging
//                    rings, one memcpy each way, the delivery side on the engine's copy threads.  The default.
//   --alloc pinned   ntscsim_host_frame_alloc() -- what ntscsim_av_frame_get_buffer() (ntscsim_avframe.h) backs an AVFrame
//                    with: pinned memory, DMA uploads, the GPU writes the results straight into the frames.
//   --alloc pool     ONE mmap'ed pool the loop carves all its planes from (64-byte aligned, NOT page aligned), declared
//                    with ntscsim_host_pin(pool, len): the explicit contract for callers with their own allocator.
//   --alloc mmap     every plane a mapping of its own (page-aligned buffers are pinned in place on first sight).
//   --alloc fakehdr  test: the pool NOT declared, every block behind a word that reads like glibc's IS_MMAPPED header.
// --pin-policy 2 with --alloc malloc and --mmap-threshold 65536 is round 5's arrangement (glibc chunk-header peek +
// mallopt); kept for A/B, no longer what INTEGRATION.md recommends.
// Prints one JSON line: fields/s over the timed fields, the FNV of all consumed frames (equal between the two modes =
// byte-identical frames in the same order), the rand() position, the engine's counters.
#include <malloc.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Frame {           // the AVFrame members the four calls read (ffmpeg_to_composite.cpp:1001-1129, :629, :1131-1236)
    uint8_t *data[8];
    int linesize[8];
    int width, height;
    int format;
    int interlaced_frame, top_field_first;
};
#define NTSCSIM_AVFRAME_T Frame
#include "ntscsim_avframe.h"

namespace {

enum { ALLOC_MALLOC = 0, ALLOC_PINNED, ALLOC_POOL, ALLOC_MMAP, ALLOC_FAKEHDR };
int g_alloc = ALLOC_MALLOC;
uint8_t *g_pool = nullptr;          // --alloc pool
size_t g_pool_len = 0, g_pool_used = 0;

Frame *frame_alloc(int W, int H, bool c420, int align, int fill)      // av_frame_alloc + av_frame_get_buffer(f, align)
{
    Frame *f = new Frame();
    std::memset(f, 0, sizeof(*f));
    f->width = W; f->height = H;
    if (g_alloc == ALLOC_PINNED) {        // av_frame_get_buffer's layout in ONE pinned block
        int rb[3], rows[3];
        for (int k = 0; k < 3; k++) { rb[k] = k ? W / 2 : W; rows[k] = (k && c420) ? (H + 1) / 2 : H; }
        void *base = nullptr;
        if (ntscsim_host_frame_alloc(3, rb, rows, align, f->data, f->linesize, &base, nullptr) != NTSCSIM_OK) return nullptr;
        for (int k = 0; k < 3; k++) std::memset(f->data[k], k ? 128 : fill, (size_t)f->linesize[k] * rows[k] + 64);
        return f;
    }
    for (int k = 0; k < 3; k++) {
        const int w = k ? W / 2 : W, rows = (k && c420) ? (H + 1) / 2 : H;
        f->linesize[k] = ((w + align - 1) / align) * align;
        void *p = nullptr;
        const size_t bytes = (size_t)f->linesize[k] * rows + 64;
        if (g_alloc == ALLOC_MMAP) {
            p = mmap(nullptr, (bytes + 4095) / 4096 * 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (p == MAP_FAILED) return nullptr;
        } else if (g_alloc == ALLOC_POOL || g_alloc == ALLOC_FAKEHDR) {
            const size_t need = (bytes + 128 + 63) / 64 * 64;         // 64 bytes of "allocator header" between blocks
            if (g_pool_used + need > g_pool_len) return nullptr;
            p = g_pool + g_pool_used + 64;
            if (!((uintptr_t)p & 4095)) p = (uint8_t *)p + 64;        // (never on a page boundary: that is --alloc mmap's case)
            g_pool_used += need;
            // a foreign allocator's header: the word in front of the block reads like glibc's "mmapped chunk, huge size"
            if (g_alloc == ALLOC_FAKEHDR) { const size_t hdr = ~(size_t)0 - 5; std::memcpy((uint8_t *)p - sizeof(size_t), &hdr, sizeof(hdr)); }
        } else if (posix_memalign(&p, 64, bytes) != 0) return nullptr;
        f->data[k] = (uint8_t *)p;
        std::memset(p, k ? 128 : fill, (size_t)f->linesize[k] * rows + 64);
    }
    return f;
}

void make_bars(Frame *f, long rot)          // 75 % colour bars in BT.601 limited-range YUV 4:2:2, rotated by `rot`
{
    static const uint8_t Y[8] = {180, 162, 131, 112, 84, 65, 35, 16}, U[8] = {128, 44, 156, 72, 184, 100, 212, 128},
                         V[8] = {128, 142, 44, 58, 198, 212, 114, 128};
    const int W = f->width;
    for (int x = 0; x < W; x++) {
        const int b = (8 * (int)((x + rot) % W)) / W;
        f->data[0][x] = Y[b];
        if (!(x & 1)) { f->data[1][x / 2] = U[b]; f->data[2][x / 2] = V[b]; }
    }
    for (int y = 1; y < f->height; y++) {
        std::memcpy(f->data[0] + (size_t)y * f->linesize[0], f->data[0], (size_t)W);
        std::memcpy(f->data[1] + (size_t)y * f->linesize[1], f->data[1], (size_t)W / 2);
        std::memcpy(f->data[2] + (size_t)y * f->linesize[2], f->data[2], (size_t)W / 2);
    }
}

uint64_t fnv1a(const Frame *f, bool c420, uint64_t h)
{
    for (int k = 0; k < 3; k++) {
        const int w = k ? f->width / 2 : f->width, rows = (k && c420) ? (f->height + 1) / 2 : f->height;
        for (int y = 0; y < rows; y++) {
            const uint8_t *p = f->data[k] + (size_t)y * f->linesize[k];
            for (int i = 0; i < w; i++) { h ^= p[i]; h *= 0x100000001B3ull; }
        }
    }
    return h;
}

} // namespace

int main(int argc, char **argv)
{
    std::string mode = "submit";
    long fields = 2000, warmup = 200;
    int depth = 32, lag = -1, do_hash = 0, height = 0, align = 32;
    long mmap_threshold = 0;
    int pin_policy = -1;
    std::vector<const char *> av;
    av.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        auto opt = [&](const char *name) { return !std::strcmp(argv[i], name) && i + 1 < argc; };
        if (opt("--mode")) { mode = argv[++i]; continue; }
        if (opt("--fields")) { fields = std::atol(argv[++i]); continue; }
        if (opt("--warmup")) { warmup = std::atol(argv[++i]); continue; }
        if (opt("--depth")) { depth = std::atoi(argv[++i]); continue; }
        if (opt("--lag")) { lag = std::atoi(argv[++i]); continue; }
        if (opt("--hash")) { do_hash = std::atoi(argv[++i]); continue; }
        if (opt("--height")) { height = std::atoi(argv[++i]); continue; }
        if (opt("--align")) { align = std::atoi(argv[++i]); continue; }
        if (opt("--page-frames")) { if (std::atoi(argv[++i]) != 0) g_alloc = ALLOC_MMAP; continue; }
        if (opt("--alloc")) {
            const std::string a = argv[++i];
            g_alloc = a == "pinned" ? ALLOC_PINNED : a == "pool" ? ALLOC_POOL : a == "mmap" ? ALLOC_MMAP : a == "fakehdr" ? ALLOC_FAKEHDR : ALLOC_MALLOC;
            continue;
        }
        if (opt("--pin-policy")) { pin_policy = std::atoi(argv[++i]); continue; }
        if (opt("--mmap-threshold")) { mmap_threshold = std::atol(argv[++i]); continue; }
        av.push_back(argv[i]);
    }
    if (mmap_threshold > 0) mallopt(M_MMAP_THRESHOLD, (int)mmap_threshold);     // (round 5's arrangement, with --pin-policy 2)
    av.push_back("-i"); av.push_back("unused"); av.push_back("-o"); av.push_back("unused");   // (the parser insists, :1634)
    const bool async = mode == "submit";
    if (!async && mode != "sync") { std::fprintf(stderr, "--mode sync|submit\n"); return 1; }
    if (lag < 0) lag = async ? 2 * depth : 0;
    if (!async) lag = 0;
    if (align < 1) align = 1;
    ntscsim_params prm;
    ntscsim_cli cli;
    ntscsim_params_init_to_composite(&prm);
    ntscsim_cli_init(&cli);
    int rc = ntscsim_params_parse_argv_to_composite(&prm, &cli, (int)av.size(), av.data(), 1);
    if (rc != NTSCSIM_OK) return 1;
    if (height > 0) prm.output_height = height;
    const int W = prm.output_width, H = prm.output_height;
    const bool interlaced_out = cli.output_video_as_interlaced != 0, out422 = cli.use_422_colorspace != 0;
    const bool feedback = prm.black_key_level_feedback >= 0, nocomp = prm.enable_composite_emulation == 0;
    // -vi -422: the tool encodes the processed frame itself (:1158); a loop with fields in flight takes a copy of it
    // as it stood after the pair (NTSCSIM_OUT422_FRAME) -- the persistent frame has moved on by the time it is consumed
    const uint32_t out_mode = (interlaced_out && out422) ? NTSCSIM_OUT422_FRAME : out422 ? NTSCSIM_OUT422_BOB422
                              : (interlaced_out ? NTSCSIM_OUT422_INTERLACED420 : NTSCSIM_OUT422_BOB420);

    ntscsim_ctx *sim = nullptr;
    rc = ntscsim_create(&prm, 0, &sim);
    if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_create: %s\n", ntscsim_strerror(rc)); return 1; }
    if (async) {
        rc = ntscsim_submit422_configure(sim, depth, lag + 2 * depth + 2 > 4 * depth ? lag + 2 * depth + 2 : 4 * depth);
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_submit422_configure: %s\n", ntscsim_strerror(rc)); return 1; }
    }
    if (pin_policy >= 0 && ntscsim_set_pin_policy(sim, pin_policy) != NTSCSIM_OK) { std::fprintf(stderr, "--pin-policy 0|1|2\n"); return 1; }
    if (g_alloc == ALLOC_POOL || g_alloc == ALLOC_FAKEHDR) {
        // the caller's own allocator: one mapping for every plane of the run, declared once
        const size_t per_frame = 2 * ((size_t)(W + align) * H + 256) + 1024;
        g_pool_len = (per_frame * (size_t)(8 + 3 + lag + 1) + 4095) / 4096 * 4096;
        void *pm = mmap(nullptr, g_pool_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (pm == MAP_FAILED) { std::fprintf(stderr, "mmap of the frame pool failed\n"); return 1; }
        g_pool = (uint8_t *)pm;
        // (--alloc fakehdr: the same pool NOT declared, every block behind a header word that looks like glibc's
        //  IS_MMAPPED chunk -- the engine must not take that for permission: tests/test_host422.py)
        rc = g_alloc == ALLOC_POOL ? ntscsim_host_pin(sim, g_pool, g_pool_len) : NTSCSIM_OK;
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_host_pin: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
    }
    // "decoded" frames, the tool's three frames, the encoder frame ring
    std::vector<Frame *> decoded;
    for (int k = 0; k < 8; k++) { decoded.push_back(frame_alloc(W, H, false, align, 16)); make_bars(decoded.back(), 3 * k); }
    Frame *input_frame = frame_alloc(W, H, false, align, 16);
    Frame *frame = frame_alloc(W, H, false, align, 16);
    Frame *filter = feedback ? frame_alloc(W, H, false, align, 16) : nullptr;
    const int ring = lag + 1;
    std::vector<Frame *> enc_ring;
    for (int k = 0; k < ring; k++) enc_ring.push_back(frame_alloc(W, H, !out422, align, 0));
    std::vector<uint64_t> tickets((size_t)ring, 0);
    std::vector<char> has_out((size_t)ring, 0);

    const long total = warmup + fields;
    uint64_t hash = 0xcbf29ce484222325ull;
    long consumed = 0;
    double us_new = 0, us_same = 0, us_wait = 0;        // host time inside the calls (timed part only)
    long n_new = 0, n_same = 0, n_wait = 0;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    bool timing = false;
    auto consume = [&](long k) {                      // what output_frame() does after its copy loops: encode (:1237-1250)
        const size_t ri = (size_t)(k % ring);
        if (async) {
            const double ta = now_us();
            const int r = ntscsim_wait(sim, tickets[ri]);
            if (timing) { us_wait += now_us() - ta; n_wait++; }
            if (r != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_wait: %s (%s)\n", ntscsim_strerror(r), ntscsim_last_error(sim)); std::exit(1); }
        }
        if (do_hash && has_out[ri]) hash = fnv1a(enc_ring[ri], !out422, hash);
        consumed++;
    };
    std::chrono::steady_clock::time_point t0;
    for (long video_field = 0; video_field < total; video_field++) {
        if (video_field == warmup) {
            for (long k = consumed; k < video_field; k++) consume(k);
            t0 = std::chrono::steady_clock::now();
            timing = true;
        }
        const unsigned field = (unsigned)((video_field & 1) ^ 1);              // :1784
        const bool new_frame = (video_field & 1) == 0;
        if (new_frame) {                                                       // sws_scale :1770-1778
            const Frame *d = decoded[(size_t)((video_field / 2) % 8)];
            for (int k = 0; k < 3; k++) std::memcpy(input_frame->data[k], d->data[k], (size_t)d->linesize[k] * H);
        }
        while (consumed + ring <= video_field) consume(consumed);
        const size_t ri = (size_t)(video_field % ring);
        // output_frame :1792-1797: -vi -> after the pair, with the previous field's parity
        const bool emit = interlaced_out ? (video_field & 1) != 0 : true;
        const unsigned out_field = interlaced_out ? (unsigned)(((video_field - 1) & 1) ^ 1) : field;
        Frame *enc = emit ? enc_ring[ri] : nullptr;
        has_out[ri] = emit ? 1 : 0;
        const double ts = now_us();
        if (async)
            rc = ntscsim_submit422_avframe(sim, frame, input_frame, 0, new_frame ? 0 : 1, filter, enc, out_mode, out_field, nocomp,
                                           field, (uint64_t)video_field, new_frame ? 0u : NTSCSIM_SUBMIT_SAME_SRC, &tickets[ri]);
        else
            rc = ntscsim_field422_avframe(sim, frame, input_frame, 0, new_frame ? 0 : 1, filter, enc, out_mode, out_field, nocomp,
                                          field, (uint64_t)video_field);
        if (timing) { if (new_frame) { us_new += now_us() - ts; n_new++; } else { us_same += now_us() - ts; n_same++; } }
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "field %ld: %s (%s)\n", video_field, ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
        while (consumed + lag <= video_field) consume(consumed);               // `lag` fields behind
    }
    for (long k = consumed; k < total; k++) consume(k);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t st[8];
    ntscsim_submit422_stats(sim, st);
    std::printf("{\"mode\": \"%s\", \"fields\": %ld, \"seconds\": %.6f, \"fields_per_s\": %.1f, \"width\": %d, \"height\": %d, "
                "\"linesize\": %d, \"depth\": %d, \"lag\": %d, \"out_mode\": %u, \"interlaced_out\": %d, \"alloc\": \"%s\", \"mmap_threshold\": %ld, "
                "\"host_us_per_call\": {\"new_frame\": %.1f, \"same_frame\": %.1f, \"wait\": %.1f}, "
                "\"fnv1a\": \"%016llx\", \"rng_pos\": %llu, \"stats\": {\"submitted\": %llu, \"launches\": %llu, \"uploads\": %llu, "
                "\"batched\": %llu, \"one_at_a_time\": %llu, \"frame_uploads\": %llu, \"delivered_direct\": %llu, \"ring_full_waits\": %llu}}\n",
                mode.c_str(), fields, dt, dt > 0 ? fields / dt : 0.0, W, H, frame->linesize[0], depth, lag, out_mode, interlaced_out ? 1 : 0,
                g_alloc == ALLOC_PINNED ? "pinned" : g_alloc == ALLOC_POOL ? "pool" : g_alloc == ALLOC_MMAP ? "mmap" : g_alloc == ALLOC_FAKEHDR ? "fakehdr" : "malloc", mmap_threshold,
                n_new ? us_new / n_new : 0.0, n_same ? us_same / n_same : 0.0, n_wait ? us_wait / n_wait : 0.0,
                (unsigned long long)(do_hash ? hash : 0), (unsigned long long)ntscsim_get_rng_pos(sim),
                (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3],
                (unsigned long long)st[4], (unsigned long long)st[5], (unsigned long long)st[6], (unsigned long long)st[7]);
    ntscsim_destroy(sim);
    return 0;
}
