// ntscsim_pool.hip -- a pool of contexts, one per GPU: ntscsim_pool_*() (include/ntscsim.h).
// Included by ntscsim_hip.hip (one translation unit).
//
// The field loop of the reference (ffmpeg_ntsc.cpp:2202-2282) carries nothing from field to field except the
// position of the process-wide rand() stream, and the number of draws per composite_layer() call does not depend
// on the pixels, so the position of any field is a closed form.  The pool therefore deals a
// run of frames block-cyclically ("frame-round-robin": block b goes to context b mod N), every context runs the
// ordinary pipelined host-frame loop (upload | kernels | download on its own three streams, frames_host_impl) on its
// blocks from its own host thread, and the union is byte-identical to one context doing the whole run.  No data
// moves between GPUs; the host buffers are pinned once, portable across devices.
#include <thread>

struct ntscsim_pool {
    std::vector<ntscsim_ctx *> ctx;
    ntscsim_params prm;
    uint64_t rng_pos = 0;
    int blk_frames = 32;
    std::string err;
};

extern "C" int ntscsim_pool_create(const ntscsim_params *p, const int *devices, int n_devices, ntscsim_pool **out)
{
    if (!p || !out || n_devices < 0 || n_devices > 64) return NTSCSIM_E_ARG;
    *out = nullptr;
    std::vector<int> devs;
    if (n_devices == 0 || !devices) {           // every visible GPU
        int nd = 0;
        if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return NTSCSIM_E_NODEV;
        const int want = n_devices > 0 ? n_devices : nd;
        for (int i = 0; i < want; i++) devs.push_back(i % nd);
    } else devs.assign(devices, devices + n_devices);
    ntscsim_pool *pl = new (std::nothrow) ntscsim_pool();
    if (!pl) return NTSCSIM_E_NOMEM;
    pl->prm = *p;
    for (int d : devs) {
        ntscsim_ctx *c = nullptr;
        const int rc = ntscsim_create(p, d, &c);
        if (rc != NTSCSIM_OK) {
            for (ntscsim_ctx *x : pl->ctx) ntscsim_destroy(x);
            delete pl;
            return rc;
        }
        pl->ctx.push_back(c);
    }
    *out = pl;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_pool_destroy(ntscsim_pool *pl)
{
    if (!pl) return;
    for (ntscsim_ctx *c : pl->ctx) ntscsim_destroy(c);
    delete pl;
}

extern "C" int ntscsim_pool_size(const ntscsim_pool *pl) { return pl ? (int)pl->ctx.size() : 0; }
extern "C" ntscsim_ctx *ntscsim_pool_ctx(ntscsim_pool *pl, int i)
{
    return (pl && i >= 0 && i < (int)pl->ctx.size()) ? pl->ctx[(size_t)i] : nullptr;
}
extern "C" uint64_t ntscsim_pool_get_rng_pos(const ntscsim_pool *pl) { return pl ? pl->rng_pos : 0; }
extern "C" void ntscsim_pool_set_rng_pos(ntscsim_pool *pl, uint64_t pos) { if (pl) pl->rng_pos = pos; }
extern "C" const char *ntscsim_pool_last_error(const ntscsim_pool *pl) { return pl ? pl->err.c_str() : ""; }
extern "C" int ntscsim_pool_set_block(ntscsim_pool *pl, int block_frames)
{
    if (!pl || block_frames < 1 || block_frames > 16384) return NTSCSIM_E_ARG;
    pl->blk_frames = block_frames;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_pool_frames_host(ntscsim_pool *pl, const uint8_t *src, size_t src_frame_stride, int src_ls,
                                        int n_frames, uint8_t *dst, size_t dst_frame_stride, int dst_ls, int W, int H,
                                        uint64_t first_fieldno, uint32_t flags, int chunk_frames)
{
    if (!pl || pl->ctx.empty()) return NTSCSIM_E_ARG;
    if (!src || !dst || n_frames < 0) return NTSCSIM_E_ARG;
    if (n_frames == 0) return NTSCSIM_OK;
    if (W < 16 || H < 2 || src_ls < 4 * W) return NTSCSIM_E_SIZE;
    const int N = (int)pl->ctx.size();
    if (chunk_frames <= 0) chunk_frames = 32;
    const int blk = pl->blk_frames;
    // the spans the shares will touch: pinned once for every device (each share's own registration would collide
    // with its neighbours' on the same pages)
    const uint32_t yuv_bits = flags & (NTSCSIM_HOST_YUV420P | NTSCSIM_HOST_YUV422P);
    const size_t crows = yuv_bits == NTSCSIM_HOST_YUV420P ? ((size_t)H + 1) / 2 : (size_t)H;
    const size_t obytes = yuv_bits ? (size_t)dst_ls * H + 2 * (size_t)(dst_ls / 2) * crows : (size_t)dst_ls * H;
    const size_t src_span = src_frame_stride * (size_t)(n_frames - 1) + (size_t)src_ls * H;
    const size_t dst_span = dst_frame_stride * (size_t)(2 * n_frames - 1) + obytes;
    HostPins pins;
    (void)hipSetDevice(pl->ctx[0]->device);
    pins.pin(src, src_span, dst, dst_span, hipHostRegisterPortable);
    std::vector<int> rcs((size_t)N, NTSCSIM_OK);
    std::vector<std::thread> th;
    for (int i = 0; i < N; i++) {
        ntscsim_ctx *c = pl->ctx[(size_t)i];
        c->rng_pos = pl->rng_pos;
        th.emplace_back([=, &rcs] {
            rcs[(size_t)i] = frames_host_impl(c, nullptr, src, src_frame_stride, src_ls, n_frames, dst, dst_frame_stride,
                                              dst_ls, W, H, first_fieldno, flags, chunk_frames, blk, i, N, /*pin*/ false);
        });
    }
    for (auto &t : th) t.join();
    pins.unpin();
    int rc = NTSCSIM_OK;
    for (int i = 0; i < N; i++)
        if (rcs[(size_t)i] != NTSCSIM_OK && rc == NTSCSIM_OK) {
            rc = rcs[(size_t)i];
            pl->err = "context " + std::to_string(i) + " (device " + std::to_string(pl->ctx[(size_t)i]->device) + "): " +
                      pl->ctx[(size_t)i]->err;
        }
    if (rc == NTSCSIM_OK) {
        const uint64_t frame_draws = ntscsim_rng_calls_per_field(&pl->prm, W, H, 0) + ntscsim_rng_calls_per_field(&pl->prm, W, H, 1);
        pl->rng_pos += (uint64_t)n_frames * frame_draws;
    }
    return rc;
}
