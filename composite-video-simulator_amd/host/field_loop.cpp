// field_loop.cpp -- the reference's field loop (ffmpeg_ntsc.cpp:2202-2282) with the ONE call at :2229 replaced,
// on AVFrame-shaped pageable host buffers; what a maintainer gets from INTEGRATION.md section 1 (synchronous:
// ntscsim_field_avframe) and section 1b (asynchronous: ntscsim_submit_avframe + ntscsim_wait `lag` fields later).
//
//   field_loop [reference switches] [--mode sync|submit] [--fields N] [--depth K] [--lanes L] [--lag G]
//              [--ring R] [--bob 0|1] [--pin 0|1|2] [--alloc malloc|pinned|pool] [--rewrite-src 0|1] [--src-stable 0|1]
//              [--warmup N] [--hash 0|1] [--height H]
//
// The loop owns, like the tool: ONE source frame per input (in.rgb, av_frame_get_buffer(..., 64) :556: linesize
// = width*4 rounded up to 64, posix_memalign'ed), a ring of output frames (:2070-2092; `--ring`, at least lag+1 so
// that a frame is not handed out again before its field came back), `current` (:2144).  Per field: every other
// field a "new decoded frame" is put into in.rgb (--rewrite-src 1: a memcpy from a pre-generated frame, the stand-in
// for sws_scale :603; 0: in.rgb is re-pointed at one of 8 pre-generated frames, no host copy), composite_layer
// (:2229) -> the GPU, then -- `lag` fields later in submit mode -- the frame is consumed (--hash 1: FNV-1a over the
// frame, the stand-in for bob / sws_scale / output_frame :2233-2280; with --bob 1 the line doubling is the GPU's).
// Frame memory (include/ntscsim.h "Host buffers"): --alloc malloc = posix_memalign like av_malloc's (the tool unpatched:
// heap blocks -> the engine's staging rings and copy threads); --alloc pinned = ntscsim_host_frame_alloc(), what
// ntscsim_av_frame_get_buffer() backs an AVFrame with (DMA uploads, the GPU writes the rows into the frame); --alloc pool
// = one mmap'ed pool of the caller's, 64-byte aligned blocks, declared with ntscsim_host_pin().  --pin = the pin policy
// (0 stage everything, 1 default, 2 + glibc chunk-header peek: round 5's behaviour for malloc'ed frames).
// Prints one JSON line: fields/s over the timed fields, the FNV of all consumed frames (equal between the two modes
// = byte-identical frames in the same order), the engine's counters.
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct Frame {           // the six AVFrame members the hot path reads (ffmpeg_ntsc.cpp:1578-1588, :1599, :1911)
    uint8_t *data[8];
    int linesize[8];
    int width, height;
    int interlaced_frame, top_field_first;
};
#define NTSCSIM_AVFRAME_T Frame
#include "ntscsim_avframe.h"

namespace {

enum { ALLOC_MALLOC = 0, ALLOC_PINNED, ALLOC_POOL };
int g_alloc = ALLOC_MALLOC;
uint8_t *g_pool = nullptr;
size_t g_pool_len = 0, g_pool_used = 0;

Frame *frame_alloc(int W, int H)            // av_frame_alloc + av_frame_get_buffer(f, 64)
{
    Frame *f = new Frame();
    std::memset(f, 0, sizeof(*f));
    f->width = W; f->height = H;
    f->linesize[0] = ((W * 4 + 63) / 64) * 64;
    void *p = nullptr;
    if (g_alloc == ALLOC_PINNED) {
        const int rb = W * 4;
        void *base = nullptr;
        if (ntscsim_host_frame_alloc(1, &rb, &H, 64, f->data, f->linesize, &base, nullptr) != NTSCSIM_OK) return nullptr;
        p = f->data[0];
    } else if (g_alloc == ALLOC_POOL) {
        const size_t need = ((size_t)f->linesize[0] * H + 128 + 63) / 64 * 64;
        if (g_pool_used + need > g_pool_len) return nullptr;
        p = g_pool + g_pool_used + 64;          // 64 bytes of "allocator header" in front of every block
        g_pool_used += need;
    } else if (posix_memalign(&p, 64, (size_t)f->linesize[0] * H + 64) != 0) return nullptr;
    f->data[0] = (uint8_t *)p;
    std::memset(p, 0, (size_t)f->linesize[0] * H);
    return f;
}

void make_bars(Frame *f, long rot)          // 75 % colour bars rotated by `rot` pixels
{
    static const uint32_t table[8] = {0xC0C0C0, 0xC0C000, 0x00C0C0, 0x00C000, 0xC000C0, 0xC00000, 0x0000C0, 0x000000};
    const int W = f->width;
    uint32_t *row0 = reinterpret_cast<uint32_t *>(f->data[0]);
    for (int x = 0; x < W; x++) row0[x] = table[(8 * (int)((x + rot) % W)) / W];
    for (int y = 1; y < f->height; y++) std::memcpy(f->data[0] + (size_t)y * f->linesize[0], f->data[0], (size_t)W * 4);
}

uint64_t fnv1a(const Frame *f, uint64_t h)
{
    for (int y = 0; y < f->height; y++) {
        const uint8_t *p = f->data[0] + (size_t)y * f->linesize[0];
        for (int i = 0; i < f->width * 4; i++) { h ^= p[i]; h *= 0x100000001B3ull; }
    }
    return h;
}

void bob(Frame *f, unsigned field)          // the "field deinterlace" block of the loop, :2233-2257
{
    const size_t rb = (size_t)f->width * 4;
    if (field) {
        for (int y = (int)field; y < f->height; y += 2)
            std::memcpy(f->data[0] + (size_t)f->linesize[0] * (y - 1), f->data[0] + (size_t)f->linesize[0] * y, rb);
    } else {
        for (int y = 1; (y + 1) < f->height; y += 2)
            std::memcpy(f->data[0] + (size_t)f->linesize[0] * y, f->data[0] + (size_t)f->linesize[0] * (y + 1), rb);
    }
}

} // namespace

int main(int argc, char **argv)
{
    std::string mode = "submit";
    long fields = 2000, warmup = 200;
    int depth = 32, lanes = 3, lag = -1, ring = -1, do_bob = 0, pin = 1, rewrite = 0, do_hash = 0, height = 486, src_stable = 0;
    std::vector<const char *> av;
    av.push_back(argv[0]);
    for (int i = 1; i < argc; i++) {
        auto opt = [&](const char *name) { return !std::strcmp(argv[i], name) && i + 1 < argc; };
        if (opt("--mode")) { mode = argv[++i]; continue; }
        if (opt("--fields")) { fields = std::atol(argv[++i]); continue; }
        if (opt("--warmup")) { warmup = std::atol(argv[++i]); continue; }
        if (opt("--depth")) { depth = std::atoi(argv[++i]); continue; }
        if (opt("--lanes")) { lanes = std::atoi(argv[++i]); continue; }
        if (opt("--lag")) { lag = std::atoi(argv[++i]); continue; }
        if (opt("--ring")) { ring = std::atoi(argv[++i]); continue; }
        if (opt("--bob")) { do_bob = std::atoi(argv[++i]); continue; }
        if (opt("--pin")) { pin = std::atoi(argv[++i]); continue; }
        if (opt("--alloc")) { const std::string a = argv[++i]; g_alloc = a == "pinned" ? ALLOC_PINNED : a == "pool" ? ALLOC_POOL : ALLOC_MALLOC; continue; }
        if (opt("--rewrite-src")) { rewrite = std::atoi(argv[++i]); continue; }
        if (opt("--src-stable")) { src_stable = std::atoi(argv[++i]); continue; }
        if (opt("--hash")) { do_hash = std::atoi(argv[++i]); continue; }
        if (opt("--height")) { height = std::atoi(argv[++i]); continue; }
        av.push_back(argv[i]);
    }
    const bool async = mode == "submit";
    if (!async && mode != "sync") { std::fprintf(stderr, "--mode sync|submit\n"); return 1; }
    if (lag < 0) lag = async ? 4 * depth : 0;      // (2 * depth leaves the link idle between launches: ~10 % slower)
    if (!async) lag = 0;
    if (ring < lag + 1) ring = lag + 1;
    ntscsim_params prm;
    ntscsim_params_init(&prm);
    prm.output_height = height;
    int rc = ntscsim_params_parse_argv(&prm, nullptr, (int)av.size(), av.data(), 0);
    if (rc != NTSCSIM_OK) return 1;
    const int W = prm.output_width, H = prm.output_height;

    ntscsim_ctx *sim = nullptr;
    rc = ntscsim_create(&prm, 0, &sim);
    if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_create: %s\n", ntscsim_strerror(rc)); return 1; }
    if (async) {
        ntscsim_submit_opts so;
        ntscsim_submit_opts_init(&so);
        so.depth = depth; so.lanes = lanes; so.pin_caller_buffers = pin;
        so.slots = lag + 2 * depth > 4 * depth ? lag + 2 * depth : 4 * depth;
        rc = ntscsim_submit_configure(sim, &so);
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_submit_configure: %s\n", ntscsim_strerror(rc)); return 1; }
    }

    if (!async && ntscsim_set_pin_policy(sim, pin) != NTSCSIM_OK) return 1;
    if (g_alloc == ALLOC_POOL) {
        g_pool_len = ((((size_t)W * 4 + 64) * H + 256) * (size_t)(8 + 1 + ring) + 4095) / 4096 * 4096;
        void *pm = mmap(nullptr, g_pool_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (pm == MAP_FAILED) { std::fprintf(stderr, "mmap of the frame pool failed\n"); return 1; }
        g_pool = (uint8_t *)pm;
        rc = ntscsim_host_pin(sim, g_pool, g_pool_len);
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_host_pin: %s (%s)\n", ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
    }
    // "decoded" frames, the tool's in.rgb, the output frame ring
    std::vector<Frame *> decoded;
    for (int k = 0; k < 8; k++) { decoded.push_back(frame_alloc(W, H)); make_bars(decoded.back(), k); }
    Frame *in_rgb = frame_alloc(W, H);
    Frame in_view = *in_rgb;
    std::vector<Frame *> out_ring;
    for (int k = 0; k < ring; k++) out_ring.push_back(frame_alloc(W, H));
    std::vector<uint64_t> tickets((size_t)ring, 0);
    std::vector<unsigned> ring_field((size_t)ring, 0);

    const long total = warmup + fields;
    double us_new = 0, us_same = 0, us_wait = 0;        // host time inside the calls (timed part only)
    long n_new = 0, n_same = 0, n_wait = 0;
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    uint64_t hash = 0xcbf29ce484222325ull;
    long consumed = 0;
    auto consume = [&](long k) {                      // what the loop does with the frame after composite_layer()
        Frame *f = out_ring[(size_t)(k % ring)];
        if (async) {
            const double ta = now_us();
            const int r = ntscsim_wait(sim, tickets[(size_t)(k % ring)]);
            us_wait += now_us() - ta; n_wait++;
            if (r != NTSCSIM_OK) { std::fprintf(stderr, "ntscsim_wait: %s (%s)\n", ntscsim_strerror(r), ntscsim_last_error(sim)); std::exit(1); }
        }
        if (!do_bob) { /* the caller's own line doubling would go here (:2233-2257) */ }
        else if (!async) bob(f, ring_field[(size_t)(k % ring)]);
        if (do_hash) hash = fnv1a(f, hash);
        consumed++;
    };
    std::chrono::steady_clock::time_point t0;
    size_t ring_idx = 0;
    for (long current = 0; current < total; current++) {
        if (current == warmup) {
            // drain, then start the clock
            for (long k = consumed; k < current; k++) consume(k);
            t0 = std::chrono::steady_clock::now();
        }
        const unsigned field = (unsigned)((current & 1) ^ 1);                  // :2229
        const bool new_frame = (current & 1) == 0;
        if (new_frame) {
            Frame *d = decoded[(size_t)((current / 2) % 8)];
            if (rewrite) { std::memcpy(in_rgb->data[0], d->data[0], (size_t)d->linesize[0] * H); in_view = *in_rgb; }   // sws_scale :603
            else in_view = *d;
        }
        // a ring frame comes around again: its previous field must have been consumed
        while (consumed + ring <= current) consume(consumed);
        Frame *dst = out_ring[ring_idx];
        ring_field[ring_idx] = field;
        const double ts = now_us();
        if (async) {
            rc = ntscsim_submit_avframe(sim, dst, &in_view, field, (uint64_t)current,
                                        (do_bob ? NTSCSIM_DESC_BOB : 0u) | (new_frame ? 0u : NTSCSIM_SUBMIT_SAME_SRC) |
                                            ((src_stable && !rewrite) ? NTSCSIM_SUBMIT_SRC_STABLE : 0u),
                                        &tickets[ring_idx]);
        } else {
            rc = ntscsim_field_avframe(sim, dst, &in_view, field, (uint64_t)current);
        }
        if (current >= warmup) { if (new_frame) { us_new += now_us() - ts; n_new++; } else { us_same += now_us() - ts; n_same++; } }
        if (rc != NTSCSIM_OK) { std::fprintf(stderr, "field %ld: %s (%s)\n", current, ntscsim_strerror(rc), ntscsim_last_error(sim)); return 1; }
        ring_idx = (ring_idx + 1) % (size_t)ring;                                  // :2277
        while (consumed + lag <= current) consume(consumed);                       // `lag` fields behind
    }
    for (long k = consumed; k < total; k++) consume(k);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t st[8];
    ntscsim_submit_stats(sim, st);
    std::printf("{\"mode\": \"%s\", \"fields\": %ld, \"seconds\": %.6f, \"fields_per_s\": %.1f, \"width\": %d, \"height\": %d, "
                "\"depth\": %d, \"lanes\": %d, \"lag\": %d, \"ring\": %d, \"bob\": %d, \"pin\": %d, \"alloc\": \"%s\", \"rewrite_src\": %d, \"src_stable\": %d, "
                "\"host_us_per_call\": {\"new_frame\": %.1f, \"same_frame\": %.1f, \"wait\": %.1f}, "
                "\"fnv1a\": \"%016llx\", \"rng_pos\": %llu, \"stats\": {\"submitted\": %llu, \"launches\": %llu, \"uploads\": %llu, "
                "\"uploads_staged\": %llu, \"delivered_direct\": %llu, \"delivered_staged\": %llu, \"registrations\": %llu, "
                "\"ring_full_waits\": %llu}}\n",
                mode.c_str(), fields, dt, dt > 0 ? fields / dt : 0.0, W, H, depth, lanes, lag, ring, do_bob, pin,
                g_alloc == ALLOC_PINNED ? "pinned" : g_alloc == ALLOC_POOL ? "pool" : "malloc", rewrite, src_stable && !rewrite,
                n_new ? us_new / n_new : 0.0, n_same ? us_same / n_same : 0.0, n_wait ? us_wait / n_wait : 0.0,
                (unsigned long long)(do_hash ? hash : 0), (unsigned long long)ntscsim_get_rng_pos(sim),
                (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3],
                (unsigned long long)st[4], (unsigned long long)st[5], (unsigned long long)st[6], (unsigned long long)st[7]);
    ntscsim_destroy(sim);       // (drops the engine's registrations before the frames are freed)
    return 0;
}
