// ntscsim_hip.hip -- context, scratch management and kernel launches behind include/ntscsim.h.
// Replaces the call site `composite_layer(dst, src, input, field, fieldno)` of the reference's
// field loop (ffmpeg_ntsc.cpp:2229) and the per-call heap planes it allocates (:1590-1592).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unistd.h>
#include <new>
#include <string>
#include <vector>

#include "ntscsim.h"
#include "ntsc_device.hpp"

// single translation unit: the kernels are compiled together with their launcher
#include "ntsc_kernels.hip"
#include "ntsc_decode_fast.hip"
#include "ntsc_encode_fast.hip"
#include "ntsc_pipe.hip"
#include "ntsc422_kernels.hip"
#include "ntsc422_fused.hip"
#include "ntsc_scale.hip"
#include "ntsc_float.hpp"       // NTSCSIM_MODE_FLOAT: its kernels are a translation unit of their own

using namespace ntscsim;

namespace {

// LowpassFilter::setFilter, ffmpeg_ntsc.cpp:78-86, at the video sample rate used by every call
double alpha_for(double hz)
{
    const double rate = (315000000.00 * 4) / 88;
    const double timeInterval = 1.0 / rate;
    const double tau = 1 / (hz * 2 * M_PI);
    return timeInterval / (tau + timeInterval);
}

// Host-to-device upload of one of the library's own small tables.  The source is ordinary heap memory, and a
// hipMemcpy from pageable memory breaks ("invalid argument") when its range straddles the edge of a hipHostRegister'ed
// region -- which can happen once the library has pinned caller frames in place (ntscsim_frames_host,
// ntscsim_submit): a heap block that starts in the last page of a registered frame is pinned at its head and pageable
// behind it.  So every such upload goes through a pinned bounce buffer of the library's own.
hipError_t upload_table(void *dst_dev, const void *src_host, size_t bytes)
{
    static std::mutex mu;
    static unsigned char *bounce = nullptr;
    static const size_t CAP = 1u << 20;
    std::lock_guard<std::mutex> lk(mu);
    if (!bounce) {
        const hipError_t e = hipHostMalloc((void **)&bounce, CAP, hipHostMallocPortable);
        if (e != hipSuccess) { bounce = nullptr; return e; }
    }
    const unsigned char *s = static_cast<const unsigned char *>(src_host);
    unsigned char *d = static_cast<unsigned char *>(dst_dev);
    for (size_t off = 0; off < bytes; off += CAP) {
        const size_t n = bytes - off < CAP ? bytes - off : CAP;
        std::memcpy(bounce, s + off, n);
        const hipError_t e = hipMemcpy(d + off, bounce, n, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Geometry {
    int W = 0, H = 0, variant = 0;
    bool valid = false;
    DevBuf<uint32_t> lskip, pskip, jrow, sstart;
    DevBuf<int32_t> jwarm;
    uint64_t calls[2] = {0, 0};
};

} // namespace

struct SubmitEngine;      // ntscsim_submit.hip (included at the end of this file)
struct Host422Engine;     // ntscsim_host422.hip (likewise)
struct PinCache;          // ntscsim_submit.hip: registrations of caller memory

struct ntscsim_ctx {
    ntscsim_params prm;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    uint64_t rng_pos = 0;

    // rand() position cache for O(1) advance between consecutive fields
    bool have_state = false;
    uint64_t state_pos = 0;
    RandState state;
    struct Delta { uint64_t n; RandPoly p; };
    std::vector<Delta> deltas;

    // Per-geometry jump tables: one IMMUTABLE set of device buffers per (W, H, tool) ever used on
    // this ctx.  A table is uploaded once into buffers nothing else references and never rewritten,
    // so launches queued for geometry A are not disturbed by a later call for geometry B (the
    // header's promise that scratch is stream-ordered).  `geom` = the entry of the current call.
    std::vector<Geometry *> geoms;
    Geometry *geom_cur = nullptr;
    DevBuf<double> ptab;
    bool ptab_ready = false;

    // per-batch scratch
    DevBuf<FieldDev> fields;
    DevBuf<int> hs_shift, pn_noise, dropout, n0_luma, n0_u, n0_v, comp, comp_ghost, comp_vcr, tails;
    DevBuf<Field422Dev> fields422;
    DevBuf<uint8_t> recs422;           // ntscsim_fields422_device: [n] FieldDev + [n] Field422Dev, one upload
    DevBuf<uint32_t> scratch422;
    DevBuf<uint8_t> halo422;         // k422_halo: the input rows the halo lanes read (ntsc422_kernels.hip: halo_redirect)
    std::vector<Out422Dev> host_out422;
    DevBuf<Out422Dev> out422;
    std::vector<YuvDev> host_yuv;
    DevBuf<YuvDev> yuv;
    std::vector<ScaleDev> host_scale;
    DevBuf<ScaleDev> scale;
    DevBuf<uint32_t> rs_luma, rs_chroma;
    FieldDev *stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    bool stage_used[2] = {false, false};
    int stage_idx = 0;
    // same for the YUV422P tool: n FieldDev records followed by n Field422Dev records
    unsigned char *stage422[2] = {nullptr, nullptr};
    size_t stage422_cap[2] = {0, 0};
    hipEvent_t stage422_ev[2] = {nullptr, nullptr};
    bool stage422_used[2] = {false, false};
    int stage422_idx = 0;

    // kernel forms enqueued by the last launch (ntscsim_debug_last_kernels)
    std::string kernels;
    // last batch (debug tap)
    int last_n = 0, last_W = 0, last_H = 0, last_Rpad = 0, last_Lslot = 0;

    // host-frame path
    DevBuf<uint8_t> fsrc, fdst;
    // ntscsim_frames_host(): two chunk slots, copy streams and events, kept between calls
    struct HostSlot { DevBuf<uint8_t> dsrc, ddst, dyuv, draw; DevBuf<YuvDev> yrec; DevBuf<ScaleDev> srec;
                      hipEvent_t up = nullptr, done = nullptr, down = nullptr; };
    HostSlot hslot[2];
    hipStream_t s_up = nullptr, s_dn = nullptr;
    // ntscsim_field(): the source rows go up on s_up while the setup kernel (which does not read pixels) runs on the ctx's
    // stream; the encoder waits for `ev_src` (launch_records)
    hipEvent_t ev_src = nullptr;
    bool src_pending = false;
    // The SYNCHRONOUS calls (ntscsim_field / ntscsim_field422) run their NEXT call's setup kernel ahead of time: behind the
    // event the caller waits for, while the host is on its way back into the library (speculate_setup).  What the kernel reads
    // is the key; launch_setup skips its launch when the call that comes has exactly that key, and launches as ever otherwise.
    struct SetupSpec {
        bool have_last = false;          // D / G / field of the last latency-form launch of one field
        DevParams D;
        GeomDev G;
        unsigned field = 0, field_before = 0;
        bool armed = false, dry = false; // key below: launched ahead | only predicted (after misses)
        hipStream_t st = nullptr;
        const void *tab[8] = {};
        unsigned kfield = 0;
        uint32_t krng[61];
        FieldDev *rec = nullptr;         // pinned, TWO records used in turn: the early kernel of the call before may still be
                                         // reading its own while the host writes the next (it has finished once the call
                                         // after it has been waited for -- the one before this one)
        unsigned rec_idx = 0;
        int misses = 0;
        uint64_t hits = 0, launched = 0, seen = 0;
    } spec;
    // (the early kernel runs BESIDE the call's own kernels, on a stream of its own, into the OTHER of two sets of setup tables:
    //  speculate_setup swaps the sets, so the next launch reads what the early kernel wrote while this call's kernels still
    //  read theirs)
    struct AltTables { DevBuf<int> hs_shift, pn_noise, dropout, n0_luma, n0_u, n0_v; DevBuf<uint32_t> rs_luma, rs_chroma; } alt;
    hipStream_t s_spec = nullptr;
    hipEvent_t ev_spec = nullptr;
    const FieldDev *setup_host_rec = nullptr;   // host copy of record 0 of the launch being made (set by the entry points)
    hipEvent_t ev_done = nullptr;               // ntscsim_field(): what the call waits for
    uint64_t field_stats[3] = {0, 0, 0};        // ntscsim_debug_field_stats
    // ntscsim_field422() / ntscsim_submit422(): the same for the YUV422P tool -- the engine's upload event (not owned);
    // launch422 runs the per-field / per-row draws first and waits for it in front of the first kernel that reads pixels
    hipEvent_t wait422_ev = nullptr;
    hipEvent_t ev_role[3] = {nullptr, nullptr, nullptr};      // NTSCSIM_ROLE_PROBE (timing probe, see launch_records)

    // profiling: five events per call (start | setup done | encode done | decode done | end),
    // recorded on the launch stream; summed and recycled by ntscsim_get_timings_ms()
    bool profiling = false;
    struct EvSet { hipEvent_t e[5]; };
    std::vector<EvSet> ev_live, ev_free;
    int warm_override[2] = {0, 0};
    bool force_generic = false;
    bool split_vhs = false;          // debug / A-B: VCR half + TV half in two launches instead of k_decode_fast<true>
    bool no_fast_decode = false;     // debug: keep the PRESET template kernels (A/B against k_decode_fast)
    bool no_stream422 = false;       // debug: the YUV422P preset kernel as four sweeps instead of A + one streamed pass
    bool no_ghost_fuse = false;      // debug / A-B: the ghosting extension as its own pass (k_ghost) for every delay
    bool no_setup_merge = false;     // debug / A-B: k_field_setup and k_row_states as two launches for short batches too
    int mode = NTSCSIM_MODE_EXACT;
    SubmitEngine *sub = nullptr;     // ntscsim_submit() / ntscsim_wait(): created on first use
    Host422Engine *h422 = nullptr;   // ntscsim_field422() / ntscsim_submit422(): created on first use
    PinCache *declared = nullptr;    // ntscsim_host_pin(): memory the caller declared its own to pin
    int pin_policy = 1;              // ntscsim_set_pin_policy()
    // the host-frame entry points (ntscsim_field(), the submit engine's lanes) ask for the LATENCY form of short launches
    // (k_field_pipe, ntsc_pipe.hip); the device-pointer entry points keep the kernels their callers (and tests) name
    bool latency_form = false;
    bool latency_pref = false;       // ntscsim_set_launch_form(): the device-pointer entry points' short launches take it too
    unsigned *pipe_fault = nullptr;  // pinned word the role kernels raise when a hand-off timed out (ntsc_pipe.hip); checked behind launches
};
static void declared_pins_destroy(ntscsim_ctx *c);
static uint8_t *pinned_device_ptr(ntscsim_ctx *c, const void *p, size_t span);      // ntscsim_submit.hip
static void submit_engine_destroy(ntscsim_ctx *c);
static int sub_wait_ticket(ntscsim_ctx *c, uint64_t ticket);
static void host422_engine_destroy(ntscsim_ctx *c);
static int h422_wait_ticket(ntscsim_ctx *c, uint64_t ticket);
static int h422_launch(ntscsim_ctx *c);

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess) {                                                       \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);           \
            return NTSCSIM_E_HIP;                                                      \
        }                                                                              \
    } while (0)

// which kernel form a launch site chose (read back by ntscsim_debug_last_kernels; the parity tests
// assert on it, so that a specialised form cannot silently stop being the one that runs)
// launches of at most this many fields from the host-frame entry points take the role kernels (ntsc_pipe.hip, k422_pipe);
// NTSCSIM_PIPE_MAX: developer switch for the crossover measurement (tools/sync_trace.sh)
static int pipe_max_fields()
{
    static const int v = std::getenv("NTSCSIM_PIPE_MAX") ? std::atoi(std::getenv("NTSCSIM_PIPE_MAX")) : NTSC_PIPE_MAX_FIELDS;
    return v;
}

static void note_kernel(ntscsim_ctx *c, const char *name)
{
    if (!c->kernels.empty()) c->kernels += ';';
    c->kernels += name;
}

static RandState ctx_state_at(ntscsim_ctx *c, uint64_t pos)
{
    if (c->have_state) {
        if (pos == c->state_pos) return c->state;
        if (pos > c->state_pos) {
            const uint64_t d = pos - c->state_pos;
            for (auto &e : c->deltas)
                if (e.n == d) {
                    c->state = rand_state_apply(e.p, c->state);
                    c->state_pos = pos;
                    return c->state;
                }
            if (c->deltas.size() < 8) {
                c->deltas.push_back({d, rand_poly_pow(d)});
                c->state = rand_state_apply(c->deltas.back().p, c->state);
                c->state_pos = pos;
                return c->state;
            }
        }
    }
    c->state = rand_state_at(pos);
    c->state_pos = pos;
    c->have_state = true;
    return c->state;
}

static void fill_dev_params(const ntscsim_params &p, DevParams &D)
{
    std::memset(&D, 0, sizeof(D));
    D.ntsc = p.tv_standard == NTSCSIM_TV_NTSC;
    D.phase_mode = p.video_scanline_phase_shift;
    D.phase_off = p.video_scanline_phase_shift_offset;
    D.in_lp = p.composite_in_chroma_lowpass != 0;
    D.out_lp = p.composite_out_chroma_lowpass ? (p.composite_out_chroma_lowpass_lite ? 1 : 2) : 0;
    D.amp = p.subcarrier_amplitude;
    D.amp_back = p.subcarrier_amplitude_back;
    D.m_amp = magic31((uint32_t)D.amp);
    D.m_amp_back = magic31((uint32_t)D.amp_back);
    D.noise_k = p.video_noise;
    D.m_noise = magic31((uint32_t)(2 * p.video_noise + 1));
    D.cnoise_k = p.video_chroma_noise;
    D.m_cnoise = magic31((uint32_t)(2 * p.video_chroma_noise + 1));
    D.pnoise_k = p.video_chroma_phase_noise;
    D.m_pnoise = magic31((uint32_t)(2 * p.video_chroma_phase_noise + 1));
    D.loss = p.video_chroma_loss;
    D.hs = p.vhs_head_switching != 0;
    D.hs_noise_on = p.vhs_head_switching_phase_noise != 0;
    D.hs_point = p.vhs_head_switching_point;
    D.hs_phase = p.vhs_head_switching_phase;
    D.hs_pn = p.vhs_head_switching_phase_noise;
    D.nocolor = p.nocolor_subcarrier != 0;
    D.vhs = p.emulating_vhs != 0;
    D.vblend = p.vhs_chroma_vert_blend != 0;
    D.svideo = p.vhs_svideo_out != 0;
    double luma_cut = 2400000, chroma_cut = 320000;          // :1773-1791
    D.cdelay = 9;
    if (p.output_vhs_tape_speed == NTSCSIM_VHS_LP) { luma_cut = 1900000; chroma_cut = 300000; D.cdelay = 12; }
    if (p.output_vhs_tape_speed == NTSCSIM_VHS_EP) { luma_cut = 1400000; chroma_cut = 280000; D.cdelay = 14; }
    D.pre_on = (p.composite_preemphasis != 0 && p.composite_preemphasis_cut > 0);
    D.pre_gain = p.composite_preemphasis;
    D.a_pre = D.pre_on ? alpha_for(p.composite_preemphasis_cut) : 0;
    D.a_in_i = alpha_for(1300000);
    D.a_in_q = alpha_for(600000);
    D.a_tv = alpha_for(2600000);
    D.a_vl = alpha_for(luma_cut);
    D.a_vc = alpha_for(chroma_cut);
    D.a_sh = alpha_for(luma_cut * 4);
    D.sharpen = p.vhs_out_sharpen;
    D.warm_luma = 64;
    D.warm_chroma = 128;
    D.ghost_taps = p.ghost_taps;
    for (int k = 0; k < 4; k++) { D.ghost_delay[k] = p.ghost_delay[k]; D.ghost_gain[k] = p.ghost_gain[k]; }
}

// stream layout of one composite_layer() call
static uint64_t chroma_stream_offset(const ntscsim_params &p, int W, int L)
{
    uint64_t o = 0;
    if (p.video_noise != 0) o += (uint64_t)W * L;
    if (p.vhs_head_switching && p.vhs_head_switching_phase_noise != 0) o += 4;
    return o;
}

static int build_geometry(ntscsim_ctx *c, int W, int H, const DevParams &D)
{
    for (Geometry *e : c->geoms)
        if (e->valid && e->W == W && e->H == H && e->variant == D.variant) { c->geom_cur = e; return NTSCSIM_OK; }
    if (c->geoms.size() >= 16) {
        // a context that cycles through more than 16 geometries: drain the device, then start over
        HIPCHK(c, hipDeviceSynchronize());
        for (Geometry *e : c->geoms) {
            e->lskip.release(); e->pskip.release(); e->jrow.release(); e->sstart.release(); e->jwarm.release();
            delete e;
        }
        c->geoms.clear();
    }
    Geometry *gp = new (std::nothrow) Geometry();
    if (!gp) return NTSCSIM_E_NOMEM;
    c->geoms.push_back(gp);
    c->geom_cur = gp;
    Geometry &g = *gp;
    g.valid = false;
    // chroma-noise draws per row: 2 per pixel (BGRA path) / 2 per chroma sample (YUV422P path)
    const uint64_t cdraws = D.variant ? 2ull * (uint64_t)(W / 2) : 2ull * (uint64_t)W;
    const int Lslot = (H + 1) / 2;
    const int Lp[2] = {(H + 1) / 2, H / 2};
    std::vector<uint32_t> lskip(2 * 31), pskip(4 * 31), sstart(4 * 31), jrow((size_t)4 * Lslot * 31);
    std::vector<int32_t> jwarm((size_t)4 * Lslot);
    const RandPoly xW = rand_poly_pow((uint64_t)W), x2W = rand_poly_pow(cdraws);
    for (int par = 0; par < 2; par++) {
        // draws before the 4 head-switch draws, and before the per-row phase-noise draws
        const uint64_t off_hs = c->prm.video_noise != 0 ? (uint64_t)W * Lp[par] : 0;
        const uint64_t off_pn = chroma_stream_offset(c->prm, W, Lp[par]) +
                                (c->prm.video_chroma_noise != 0 ? cdraws * Lp[par] : 0);
        const RandPoly a = rand_poly_pow(off_hs);
        const RandPoly b = rand_poly_pow(off_pn);
        std::memcpy(&lskip[par * 31], a.c, sizeof(a.c));
        std::memcpy(&pskip[par * 31], b.c, sizeof(b.c));
        // ... and before the per-row dropout draws, which follow the phase-noise ones (k_field_row_setup walks the two in parallel)
        const RandPoly bd = rand_poly_pow(off_pn + (c->prm.video_chroma_phase_noise != 0 ? (uint64_t)Lp[par] : 0));
        std::memcpy(&pskip[(2 + par) * 31], bd.c, sizeof(bd.c));
        g.calls[par] = D.variant ? ntscsim_rng_calls_per_field_422(&c->prm, W, H, (unsigned)par)
                                 : ntscsim_rng_calls_per_field(&c->prm, W, H, (unsigned)par);
        for (int s = 0; s < 2; s++) {
            const uint64_t off = s == 0 ? 0 : chroma_stream_offset(c->prm, W, Lp[par]);
            const RandPoly so = rand_poly_pow(off);
            std::memcpy(&sstart[(size_t)(s * 2 + par) * 31], so.c, sizeof(so.c));
            const uint64_t warm_max = s == 0 ? (uint64_t)D.warm_luma : (uint64_t)D.warm_chroma;
            const uint64_t per_row = s == 0 ? (uint64_t)W : cdraws;
            const RandPoly &step = s == 0 ? xW : x2W;
            RandPoly cur = so;
            uint64_t cur_e = off;
            for (int k = 0; k < Lslot; k++) {
                const uint64_t start = per_row * (uint64_t)k;
                const uint64_t warm = start < warm_max ? start : warm_max;
                const uint64_t e = off + start - warm;
                if (e != cur_e) {
                    cur = (e - cur_e == per_row) ? rand_poly_mul(cur, step) : rand_poly_pow(e);
                    cur_e = e;
                }
                const size_t idx = (size_t)(s * 2 + par) * Lslot + k;
                std::memcpy(&jrow[idx * 31], cur.c, sizeof(cur.c));
                jwarm[idx] = (int32_t)warm;
            }
        }
    }
    HIPCHK(c, g.lskip.ensure(lskip.size()));
    HIPCHK(c, g.pskip.ensure(pskip.size()));
    HIPCHK(c, g.sstart.ensure(sstart.size()));
    HIPCHK(c, g.jrow.ensure(jrow.size()));
    HIPCHK(c, g.jwarm.ensure(jwarm.size()));
    HIPCHK(c, upload_table(g.lskip.p, lskip.data(), lskip.size() * 4));
    HIPCHK(c, upload_table(g.pskip.p, pskip.data(), pskip.size() * 4));
    HIPCHK(c, upload_table(g.sstart.p, sstart.data(), sstart.size() * 4));
    HIPCHK(c, upload_table(g.jrow.p, jrow.data(), jrow.size() * 4));
    HIPCHK(c, upload_table(g.jwarm.p, jwarm.data(), jwarm.size() * 4));
    g.W = W; g.H = H; g.variant = D.variant; g.valid = true;
    return NTSCSIM_OK;
}

static int build_ptab(ntscsim_ctx *c)
{
    if (c->ptab_ready) return NTSCSIM_OK;
    const int K = c->prm.video_chroma_phase_noise;
    std::vector<double> t((size_t)(2 * K + 1) * 2);
    for (int n = -K; n <= K; n++) {
        const double pi = ((double)n * M_PI) / 100;        // ffmpeg_ntsc.cpp:1746
        t[(size_t)(n + K) * 2] = std::cos(pi);              // host libm, as the reference
        t[(size_t)(n + K) * 2 + 1] = std::sin(pi);
    }
    HIPCHK(c, c->ptab.ensure(t.size()));
    HIPCHK(c, upload_table(c->ptab.p, t.data(), t.size() * sizeof(double)));
    c->ptab_ready = true;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_create(const ntscsim_params *p, int device, ntscsim_ctx **out)
{
    if (!p || !out) return NTSCSIM_E_ARG;
    *out = nullptr;
    int rc = ntscsim_params_validate(p);
    if (rc != NTSCSIM_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NTSCSIM_E_NODEV;
    if (device < 0 || device >= ndev) return NTSCSIM_E_NODEV;
    ntscsim_ctx *c = new (std::nothrow) ntscsim_ctx();
    if (!c) return NTSCSIM_E_NOMEM;
    c->prm = *p;
    c->device = device;
    // developer A/B switch (same as ntscsim_debug_no_fast_decode): NTSCSIM_DEBUG_DECODE=1|2|3
    if (const char *e = std::getenv("NTSCSIM_DEBUG_DECODE")) {
        const int v = std::atoi(e);
        c->no_fast_decode = (v & 1) != 0; c->split_vhs = (v & 2) != 0; c->no_stream422 = (v & 4) != 0;
        c->no_ghost_fuse = (v & 8) != 0; c->no_setup_merge = (v & 16) != 0;
    }
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return NTSCSIM_E_HIP;
    }
    for (int i = 0; i < 2; i++) (void)hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming);
    for (int i = 0; i < 2; i++) (void)hipEventCreateWithFlags(&c->stage422_ev[i], hipEventDisableTiming);
    *out = c;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_destroy(ntscsim_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    submit_engine_destroy(c);
    host422_engine_destroy(c);
    declared_pins_destroy(c);
    for (Geometry *e : c->geoms) {
        e->lskip.release(); e->pskip.release(); e->jrow.release(); e->sstart.release(); e->jwarm.release();
        delete e;
    }
    c->geoms.clear();
    c->ptab.release(); c->fields.release(); c->hs_shift.release(); c->pn_noise.release();
    c->dropout.release(); c->n0_luma.release(); c->n0_u.release(); c->n0_v.release();
    c->comp.release(); c->comp_ghost.release(); c->comp_vcr.release(); c->tails.release(); c->fields422.release(); c->recs422.release(); c->out422.release(); c->yuv.release(); c->scale.release(); c->scratch422.release(); c->halo422.release(); c->rs_luma.release(); c->rs_chroma.release();
    c->fsrc.release(); c->fdst.release();
    for (auto &h : c->hslot) {
        h.dsrc.release(); h.ddst.release(); h.dyuv.release(); h.yrec.release(); h.draw.release(); h.srec.release();
        if (h.up) (void)hipEventDestroy(h.up);
        if (h.done) (void)hipEventDestroy(h.done);
        if (h.down) (void)hipEventDestroy(h.down);
    }
    if (c->ev_src) (void)hipEventDestroy(c->ev_src);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->spec.rec) (void)hipHostFree(c->spec.rec);
    c->alt.hs_shift.release(); c->alt.pn_noise.release(); c->alt.dropout.release(); c->alt.n0_luma.release();
    c->alt.n0_u.release(); c->alt.n0_v.release(); c->alt.rs_luma.release(); c->alt.rs_chroma.release();
    if (c->ev_spec) (void)hipEventDestroy(c->ev_spec);
    if (c->s_spec) (void)hipStreamDestroy(c->s_spec);
    if (std::getenv("NTSCSIM_SETUP_AHEAD_STATS") && c->spec.seen)       // developer probe
        std::fprintf(stderr, "setup ahead: %llu single-field launches, %llu early kernels, %llu used\n",
                     (unsigned long long)c->spec.seen, (unsigned long long)c->spec.launched, (unsigned long long)c->spec.hits);
    for (auto &e_ : c->ev_role) if (e_) (void)hipEventDestroy(e_);
    if (c->s_up) (void)hipStreamDestroy(c->s_up);
    if (c->s_dn) (void)hipStreamDestroy(c->s_dn);
    for (int i = 0; i < 2; i++) {
        if (c->stage[i]) (void)hipHostFree(c->stage[i]);
        if (i == 0 && c->pipe_fault) { (void)hipHostFree(c->pipe_fault); c->pipe_fault = nullptr; }
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
        if (c->stage422[i]) (void)hipHostFree(c->stage422[i]);
        if (c->stage422_ev[i]) (void)hipEventDestroy(c->stage422_ev[i]);
    }
    for (auto &s : c->ev_live) for (int i = 0; i < 5; i++) (void)hipEventDestroy(s.e[i]);
    for (auto &s : c->ev_free) for (int i = 0; i < 5; i++) (void)hipEventDestroy(s.e[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *ntscsim_last_error(const ntscsim_ctx *c) { return c ? c->err.c_str() : ""; }
extern "C" uint64_t ntscsim_get_rng_pos(const ntscsim_ctx *c) { return c ? c->rng_pos : 0; }
extern "C" void ntscsim_set_rng_pos(ntscsim_ctx *c, uint64_t pos) { if (c) c->rng_pos = pos; }
extern "C" void ntscsim_set_profiling(ntscsim_ctx *c, int on) { if (c) c->profiling = on != 0; }

extern "C" void ntscsim_debug_field_stats(const ntscsim_ctx *c, uint64_t out[4])
{
    if (!out) return;
    for (int i = 0; i < 3; i++) out[i] = c ? c->field_stats[i] : 0;
    out[3] = c ? c->spec.hits : 0;
}

extern "C" int ntscsim_set_launch_form(ntscsim_ctx *c, int form)
{
    if (!c || (form != NTSCSIM_FORM_THROUGHPUT && form != NTSCSIM_FORM_LATENCY)) return NTSCSIM_E_ARG;
    c->latency_pref = form == NTSCSIM_FORM_LATENCY;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_sync(ntscsim_ctx *c)
{
    if (!c) return NTSCSIM_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = c->sub ? sub_wait_ticket(c, NTSCSIM_TICKET_ALL) : NTSCSIM_OK;         // fields in flight: deliver them
    if (c->h422) { const int r = h422_wait_ticket(c, NTSCSIM_TICKET_ALL); if (rc == NTSCSIM_OK) rc = r; }
    HIPCHK(c, hipDeviceSynchronize());
    return rc;
}

extern "C" int ntscsim_get_timings_ms(ntscsim_ctx *c, float out_ms[4], int *n_calls)
{
    if (!c || !out_ms) return NTSCSIM_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    out_ms[0] = out_ms[1] = out_ms[2] = out_ms[3] = 0.f;
    int n = 0;
    for (auto &s : c->ev_live) {
        float t[4];
        HIPCHK(c, hipEventSynchronize(s.e[4]));
        HIPCHK(c, hipEventElapsedTime(&t[0], s.e[0], s.e[1]));
        HIPCHK(c, hipEventElapsedTime(&t[1], s.e[1], s.e[2]));
        HIPCHK(c, hipEventElapsedTime(&t[2], s.e[2], s.e[3]));
        HIPCHK(c, hipEventElapsedTime(&t[3], s.e[0], s.e[4]));
        for (int i = 0; i < 4; i++) out_ms[i] += t[i];
        c->ev_free.push_back(s);
        n++;
    }
    c->ev_live.clear();
    if (n_calls) *n_calls = n;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_set_mode(ntscsim_ctx *c, int mode)
{
    if (!c || (mode != NTSCSIM_MODE_EXACT && mode != NTSCSIM_MODE_FAST32 && mode != NTSCSIM_MODE_FLOAT)) return NTSCSIM_E_ARG;
    c->mode = mode;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_debug_force_generic(ntscsim_ctx *c, int on)
{
    if (c) c->force_generic = on != 0;
}

extern "C" void ntscsim_debug_no_fast_decode(ntscsim_ctx *c, int on)
{
    if (c) { c->no_fast_decode = (on & 1) != 0; c->split_vhs = (on & 2) != 0; c->no_stream422 = (on & 4) != 0; c->no_ghost_fuse = (on & 8) != 0; c->no_setup_merge = (on & 16) != 0; }
}

extern "C" int ntscsim_debug_last_kernels(const ntscsim_ctx *c, char *out, size_t cap)
{
    if (!c || !out || cap == 0) return NTSCSIM_E_ARG;
    const size_t n = std::min(cap - 1, c->kernels.size());
    std::memcpy(out, c->kernels.data(), n);
    out[n] = 0;
    return (int)c->kernels.size();
}

extern "C" void ntscsim_debug_set_warmup(ntscsim_ctx *c, int luma_draws, int chroma_draws)
{
    if (!c) return;
    c->warm_override[0] = luma_draws;
    c->warm_override[1] = chroma_draws & ~1;
    // (test hook only) the tables depend on the warm-up lengths: drain and forget them
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (Geometry *e : c->geoms) e->valid = false;
}

// The draws that are not per-pixel (k_field_setup) and the rand() / noise state of every row start (k_row_states):
// two launches for the long batches, one (k_field_row_setup: the field setup overlaps the row states) for short ones.
#ifndef NTSC_SETUP_MERGE_MAX
#define NTSC_SETUP_MERGE_MAX 64      /* fields; 0 = always two launches (A/B) */
#endif
static void launch_setup_kernels(ntscsim_ctx *c, const DevParams &D, const GeomDev &G, const FieldDev *fields_dev, int n, hipStream_t st,
                                 bool note, bool launch)
{
    const bool fs = D.hs || D.pnoise_k || D.loss, rs = D.noise_k || D.cnoise_k;
    if (fs && rs && n <= NTSC_SETUP_MERGE_MAX && !c->no_setup_merge) {
        const int nfs = 3 * n, nrs = (D.R + 63) / 64;      // (one block per field and part: head switch | phase noise | dropout)
        if (note) note_kernel(c, "k_field_row_setup");
        if (launch)
        hipLaunchKernelGGL(k_field_row_setup, dim3((unsigned)(nfs + 2 * nrs)), dim3(64), 0, st, D, G, fields_dev,
                           c->hs_shift.p, c->pn_noise.p, c->dropout.p, c->rs_luma.p, c->n0_luma.p, c->rs_chroma.p,
                           c->n0_u.p, c->n0_v.p, nfs, nrs);
        return;
    }
    if (fs) {
        if (note) note_kernel(c, "k_field_setup");
        if (launch)
        hipLaunchKernelGGL(k_field_setup, dim3((n + 63) / 64), dim3(64), 0, st, D, G, fields_dev,
                           c->hs_shift.p, c->pn_noise.p, c->dropout.p);
    }
    if (rs) {
        if (note) note_kernel(c, "k_row_states");
        if (launch)
        hipLaunchKernelGGL(k_row_states, dim3((D.R + 63) / 64, 2), dim3(64), 0, st, D, G,
                           fields_dev, c->rs_luma.p, c->n0_luma.p, c->rs_chroma.p, c->n0_u.p,
                           c->n0_v.p);
    }
}

static void setup_tables(const ntscsim_ctx *c, const void *t[8])
{
    t[0] = c->hs_shift.p; t[1] = c->pn_noise.p; t[2] = c->dropout.p; t[3] = c->rs_luma.p;
    t[4] = c->n0_luma.p; t[5] = c->rs_chroma.p; t[6] = c->n0_u.p; t[7] = c->n0_v.p;
}

static void launch_setup(ntscsim_ctx *c, const DevParams &D, const GeomDev &G, const FieldDev *fields_dev, int n, hipStream_t st)
{
    ntscsim_ctx::SetupSpec &sp = c->spec;
    const FieldDev *hr = c->setup_host_rec;
    c->setup_host_rec = nullptr;
    bool skip = false;
    if (sp.armed || sp.dry) {
        // the early launch's key against this call's: everything the setup kernels read, and where they write
        const void *t[8];
        setup_tables(c, t);
        const bool same = hr && n == 1 && c->latency_form && st == sp.st && !std::memcmp(&D, &sp.D, sizeof(D)) &&
                          !std::memcmp(&G, &sp.G, sizeof(G)) && !std::memcmp(t, sp.tab, sizeof(t)) &&
                          (hr->field & 1u) == sp.kfield && !std::memcmp(hr->rng, sp.krng, sizeof(sp.krng));
        if (same) { sp.misses = 0; if (sp.armed) { skip = true; sp.hits++; } }
        else if (c->latency_form && n == 1) sp.misses++;
        // (the early kernel ran on its own stream: whatever reads or rewrites its tables comes behind it)
        if (sp.armed) (void)hipStreamWaitEvent(st, c->ev_spec, 0);
        sp.armed = sp.dry = false;           // (whatever runs now rewrites the tables)
    }
    launch_setup_kernels(c, D, G, fields_dev, n, st, true, !skip);
    sp.have_last = false;
    if (hr && n == 1 && c->latency_form) {
        sp.have_last = true;
        sp.D = D; sp.G = G; sp.st = st;
        sp.field_before = sp.field;
        sp.field = hr->field & 1u;
        sp.seen++;
    }
}

// Behind a synchronous call's last operation (and the event its caller waits for): the setup kernel of the call that is
// expected next -- same switches and geometry, the field parity continuing the pattern of the last two calls, the rand()
// stream where this call left it.  NTSCSIM_SETUP_AHEAD=0: A/B switch.
static void speculate_setup(ntscsim_ctx *c, hipStream_t st)
{
    static const bool on = !(std::getenv("NTSCSIM_SETUP_AHEAD") && std::getenv("NTSCSIM_SETUP_AHEAD")[0] == '0');
    ntscsim_ctx::SetupSpec &sp = c->spec;
    if (!on || !sp.have_last || st != sp.st || sp.armed) return;
    sp.have_last = false;
    if (!sp.rec && hipHostMalloc((void **)&sp.rec, 2 * sizeof(FieldDev), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); sp.rec = nullptr; return; }
    if (!c->s_spec && hipStreamCreateWithFlags(&c->s_spec, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->s_spec = nullptr; return; }
    if (!c->ev_spec && hipEventCreateWithFlags(&c->ev_spec, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->ev_spec = nullptr; return; }
    FieldDev r;
    std::memset(&r, 0, sizeof(r));
    // the parity alternates if the last two calls' did (or there is one call only), repeats otherwise
    r.field = (sp.seen < 2 || sp.field != sp.field_before) ? sp.field ^ 1u : sp.field;
    const RandState s = ctx_state_at(c, c->rng_pos);
    std::memcpy(r.rng, s.w, sizeof(s.w));
    for (int j = 31; j < 61; j++) r.rng[j] = r.rng[j - 31] + r.rng[j - 3];
    sp.kfield = r.field & 1u;
    std::memcpy(sp.krng, r.rng, sizeof(sp.krng));
    if (sp.misses >= 2) { setup_tables(c, sp.tab); sp.dry = true; return; }      // (mispredicted twice: predict only, until a prediction holds again)
    // the other set of tables, as large as the set in use (the kernels that read that one may still be running)
    // (exactly as large: ensure() rounds up, and two sets that leapfrog each other would grow at every call)
    auto match = [](auto &a, const auto &m) {
        if (a.cap >= m.cap) return true;
        a.release();
        if (hipMalloc((void **)&a.p, m.cap * sizeof(*a.p)) != hipSuccess) { a.p = nullptr; return false; }
        a.cap = m.cap;
        return true;
    };
    if (!match(c->alt.hs_shift, c->hs_shift) || !match(c->alt.pn_noise, c->pn_noise) || !match(c->alt.dropout, c->dropout) ||
        !match(c->alt.n0_luma, c->n0_luma) || !match(c->alt.n0_u, c->n0_u) || !match(c->alt.n0_v, c->n0_v) ||
        !match(c->alt.rs_luma, c->rs_luma) || !match(c->alt.rs_chroma, c->rs_chroma)) {
        (void)hipGetLastError();
        return;
    }
    std::swap(c->hs_shift, c->alt.hs_shift); std::swap(c->pn_noise, c->alt.pn_noise); std::swap(c->dropout, c->alt.dropout);
    std::swap(c->n0_luma, c->alt.n0_luma); std::swap(c->n0_u, c->alt.n0_u); std::swap(c->n0_v, c->alt.n0_v);
    std::swap(c->rs_luma, c->alt.rs_luma); std::swap(c->rs_chroma, c->alt.rs_chroma);
    setup_tables(c, sp.tab);
    FieldDev *const slot = sp.rec + (sp.rec_idx ^= 1u);
    *slot = r;
    launch_setup_kernels(c, sp.D, sp.G, slot, 1, c->s_spec, false, true);
    (void)hipEventRecord(c->ev_spec, c->s_spec);
    (void)hipGetLastError();
    sp.armed = true;
    sp.launched++;
}

// ---- step 1: descriptors -> device records.  The rand() window of every field is computed on
// the host (one 31x31 multiply-accumulate per consecutive field).
static int prepare_records(ntscsim_ctx *c, const ntscsim_field_desc *descs, int n, int W, int H,
                           DevParams &D, FieldDev *fh, bool &any_bob, uint64_t &rng_end)
{
    if (W < 16 || H < 2 || W > 16384 || H > 16384) return NTSCSIM_E_SIZE;
    fill_dev_params(c->prm, D);
    if (c->warm_override[0] > 0) D.warm_luma = c->warm_override[0];
    if (c->warm_override[1] > 0) D.warm_chroma = c->warm_override[1];
    D.W = W; D.H = H;
    D.Lslot = (H + 1) / 2;
    D.nfields = n;
    const long long R = (long long)n * D.Lslot;
    if (R > (1ll << 30)) return NTSCSIM_E_SIZE;
    D.R = (int)R;
    D.Rpad = (int)(((R + 63) / 64) * 64 + 64);
    if ((long long)D.Rpad * W > (1ll << 31)) return NTSCSIM_E_SIZE;   // keep comp under 8 GiB

    int rc = build_geometry(c, W, H, D);
    if (rc != NTSCSIM_OK) return rc;
    if (D.pnoise_k) { rc = build_ptab(c); if (rc != NTSCSIM_OK) return rc; }

    bool al_src = true, al_dst = true;
    any_bob = false;
    uint64_t pos = c->rng_pos;
    for (int i = 0; i < n; i++) {
        const ntscsim_field_desc &d = descs[i];
        if (!d.src_dev || !d.dst_dev) return NTSCSIM_E_ARG;
        if (d.src_linesize < 4 * W || d.dst_linesize < 4 * W) return NTSCSIM_E_SIZE;  // :1580-1581
        if ((d.src_linesize & 3) || (d.dst_linesize & 3)) return NTSCSIM_E_SIZE;
        if (((uintptr_t)d.src_dev & 3) || ((uintptr_t)d.dst_dev & 3)) return NTSCSIM_E_ARG;
        if (d.field > 1) return NTSCSIM_E_ARG;
        if (d.rng_pos != NTSCSIM_RNG_AUTO) pos = d.rng_pos;
        FieldDev &o = fh[i];
        o.src = (const uint8_t *)d.src_dev;
        o.dst = (uint8_t *)d.dst_dev;
        o.src_ls = d.src_linesize; o.dst_ls = d.dst_linesize;
        o.field = d.field; o.flags = d.flags; o.fieldno = d.fieldno; o._pad = 0;
        const RandState s = ctx_state_at(c, pos);
        std::memcpy(o.rng, s.w, sizeof(s.w));
        for (int j = 31; j < 61; j++) o.rng[j] = o.rng[j - 31] + o.rng[j - 3];
        pos += c->geom_cur->calls[d.field & 1];
        al_src = al_src && !(((uintptr_t)d.src_dev | (uintptr_t)d.src_linesize) & 15);
        al_dst = al_dst && !(((uintptr_t)d.dst_dev | (uintptr_t)d.dst_linesize) & 15);
        any_bob = any_bob || (d.flags & NTSCSIM_DESC_BOB);
    }
    if (any_bob && n > 65535) return NTSCSIM_E_SIZE;      // k_bob: one grid row per field
    // Descriptors of one batch run concurrently: two of them may share a destination frame only
    // as its two fields (different parity, no bob) -- anything else is a write-write race.
    if (n > 1) {
        std::vector<std::pair<uintptr_t, unsigned>> keys((size_t)n);
        for (int i = 0; i < n; i++)
            keys[(size_t)i] = {(uintptr_t)descs[i].dst_dev,
                               (descs[i].flags & NTSCSIM_DESC_BOB) ? 2u : (descs[i].field & 1u)};
        std::sort(keys.begin(), keys.end());
        for (int i = 1; i < n; i++)
            if (keys[(size_t)i].first == keys[(size_t)i - 1].first &&
                (keys[(size_t)i].second == keys[(size_t)i - 1].second || keys[(size_t)i].second == 2u)) {
                c->err = "descriptors of one batch share a destination frame (same field parity, or bob)";
                return NTSCSIM_E_ARG;
            }
    }
    rng_end = pos;          // committed by the caller once the launch has succeeded
    D.src_al16 = al_src; D.dst_al16 = al_dst;
    return NTSCSIM_OK;
}

// Largest head-switch displacement any field can get (k_field_setup, ffmpeg_ntsc.cpp:1647-1684)
// is at most W/10 samples: then "row[(x + shift) mod twidth] with zero fill" is "row[x + shift]
// inside the row, 0 elsewhere", which k_decode_fast gets from a buffer bounds check.
static bool head_switch_is_small(const DevParams &D, int W)
{
    if (!D.hs) return true;
    const unsigned tw = (unsigned)W + (unsigned)W / 10u;
    const double t = D.ntsc ? tw * 262.5 : tw * 312.5;
    const double pn = D.hs_noise_on ? std::fabs(D.hs_pn) : 0.0;
    const double lo = D.hs_phase - pn, hi = D.hs_phase + pn;
    if (lo < 0 || std::trunc(lo) != std::trunc(hi)) return false;
    const unsigned plo = (unsigned)((lo - std::trunc(lo)) * t), phi = (unsigned)((hi - std::trunc(hi)) * t);
    if (plo / tw != phi / tw) return false;
    const unsigned hlo = plo % tw, hh = phi % tw;
    if ((hlo >= tw / 2) != (hh >= tw / 2)) return false;
    const int a = hlo >= tw / 2 ? (int)(hlo - tw) : (int)hlo, b = hh >= tw / 2 ? (int)(hh - tw) : (int)hh;
    const int m = std::max(std::abs(a), std::abs(b));
    return m <= (int)(tw - (unsigned)W);
}

// The hand-tuned kernels address the transposed composite plane through a raw buffer descriptor with a
// 32-bit byte offset: offset = row*4 + (x + head-switch displacement) * Rpad*4, compared (unsigned)
// with num_records = W * Rpad*4, and out-of-range reads return the reference's fill value 0.  That only
// works while the displaced offset cannot wrap around 2^32 back INTO the plane: the displacement lies in
// [-W/10, +W/10] (head_switch_is_small), so both x + shift >= W and x + shift < 0 stay outside iff
// (W + W/10 + 1) * Rpad*4 fits in 32 bits.  When the displacement can wrap around the tw = 1.1 W window
// (|shift| up to tw/2: k_decode_fast's WR form) the offset is first formed for x + shift, which reaches
// from -tw/2 to W - 1 + tw/2, and then moved by -+ tw samples: (W + tw + 1) * Rpad*4 is required to fit,
// a margin that covers every intermediate value.  Larger planes take the generic kernels (64-bit
// addressing).  hs_mode: 0 no head switching, 1 displacement <= W/10, 2 any displacement.
static bool fast_plane_ok(size_t Rpad, int W, int hs_mode)
{
    const size_t tw = (size_t)W + (size_t)W / 10u;
    const size_t span = (size_t)W + (hs_mode == 0 ? 0u : (hs_mode == 1 ? (size_t)W / 10u + 1u : tw + 1u));
    return span * Rpad * 4u < 0xFFF00000ull;
}

extern "C" int ntscsim_debug_fast_plane_ok(int n_fields, int W, int H, int head_switching)
{
    if (n_fields <= 0 || W <= 0 || H <= 0 || head_switching < 0 || head_switching > 2) return 0;
    const long long R = (long long)n_fields * ((H + 1) / 2);
    const size_t Rpad = (size_t)(((R + 63) / 64) * 64 + 64);         // as prepare_records
    return fast_plane_ok(Rpad, W, head_switching) ? 1 : 0;
}

// ---- step 2: scratch + the kernel chain over device-resident records
static int launch_records(ntscsim_ctx *c, const DevParams &D, const FieldDev *fields_dev,
                          bool any_bob, hipStream_t st, const ntscsim_ctx::EvSet *evs)
{
    const int n = D.nfields, W = D.W, H = D.H;
    // (no-ops unless another geometry was used on this ctx since the records were prepared)
    int grc = build_geometry(c, W, H, D);
    if (grc != NTSCSIM_OK) return grc;
    if (D.pnoise_k) { grc = build_ptab(c); if (grc != NTSCSIM_OK) return grc; }
    HIPCHK(c, c->comp.ensure((size_t)D.Rpad * W));
    if (D.hs) HIPCHK(c, c->hs_shift.ensure((size_t)D.R));
    if (D.pnoise_k) HIPCHK(c, c->pn_noise.ensure((size_t)D.R));
    if (D.loss) HIPCHK(c, c->dropout.ensure((size_t)D.R));
    if (D.noise_k) { HIPCHK(c, c->rs_luma.ensure((size_t)31 * D.Rpad)); HIPCHK(c, c->n0_luma.ensure((size_t)D.Rpad)); }
    if (D.cnoise_k) {
        HIPCHK(c, c->rs_chroma.ensure((size_t)31 * D.Rpad));
        HIPCHK(c, c->n0_u.ensure((size_t)D.Rpad));
        HIPCHK(c, c->n0_v.ensure((size_t)D.Rpad));
    }
    const dim3 dgrid((D.R + 62) / 63);
    if (D.vhs) HIPCHK(c, c->tails.ensure((size_t)32 * 64 * dgrid.x));

    GeomDev G;
    G.lskip = c->geom_cur->lskip.p; G.pskip = c->geom_cur->pskip.p; G.jrow = c->geom_cur->jrow.p;
    G.jwarm = c->geom_cur->jwarm.p; G.sstart = c->geom_cur->sstart.p; G.ptab = c->ptab.p;

    c->kernels.clear();
    launch_setup(c, D, G, fields_dev, n, st);
    if (c->src_pending) {          // ntscsim_field(): the pixels were uploaded beside the setup kernel
        c->src_pending = false;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_src, 0));
    }
    if (evs) HIPCHK(c, hipEventRecord(evs->e[1], st));
    // NTSCSIM_ROLE_PROBE=1 (developer timing probe, WRONG pixels): with the two-launch VHS form selected
    // (ntscsim_debug_no_fast_decode bit 1) the three roles of a field -- encoder, VCR half, TV half -- are launched side
    // by side on three streams with no dependency between them: the time of a call is then what a perfectly pipelined
    // three-role workgroup could reach at best (tools/role_probe.py, profiles/r06_role_probe.txt)
    static const bool role_probe = std::getenv("NTSCSIM_ROLE_PROBE") && std::getenv("NTSCSIM_ROLE_PROBE")[0] == '1';
    // PRESET kernels (options folded at compile time) when the parameters match the default
    // preset or the full -vhs preset exactly; otherwise the GENERIC kernels.  Same results.
    const bool enc_preset = !c->force_generic && D.in_lp && !D.pre_on && D.noise_k != 0 && D.amp == 50;
    const bool fast = c->mode != NTSCSIM_MODE_EXACT;          // float filter states (FAST32; FLOAT's fallback for other switch sets)
#define NTSC_LAUNCH_ENCODE(F, RT)                                                               \
    do { note_kernel(c, ("k_encode<" + std::to_string((unsigned)(F)) + "u," #RT ">").c_str());  \
    hipLaunchKernelGGL((k_encode<F, RT>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,             \
                       fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p); } while (0)
    const bool even_phase = (D.phase_mode == 180 || (D.phase_mode != 90 && D.phase_mode != 270)) && !(D.phase_off & 1);
    const bool hs_small = head_switch_is_small(D, W);
    const int hs_mode = !D.hs ? 0 : (hs_small ? 1 : 2);
    const bool small_plane = fast_plane_ok((size_t)D.Rpad, W, hs_mode);
    // the same with composite pre-emphasis (the -comp-catv* presets): k_encode_fast_pre<RT>
    const bool enc_preset_pre = !c->force_generic && D.in_lp && D.pre_on && D.noise_k != 0 && D.amp == 50;
    // extension: ghosting.  Folded into the hand-tuned encoder when every delay fits its 64-sample ring
    // (k_encode_fast_gh), a pass of its own between encoder and decoder otherwise (k_ghost).
    bool ghost_fused = D.ghost_taps > 0 && !c->no_ghost_fuse;
    for (int k = 0; k < D.ghost_taps; k++) ghost_fused = ghost_fused && D.ghost_delay[k] < NTSC_GHOST_RING;
    ghost_fused = ghost_fused && enc_preset && !c->no_fast_decode && even_phase && small_plane && D.src_al16 &&
                  D.noise_k <= (1 << 20);
    // NTSCSIM_MODE_FLOAT: the all-float pipeline (ntsc_float.hip) for the default preset and the -vhs family with its
    // standard switches -- encoder AND decoder (the plane between them holds floats); anything else runs the FAST32 forms
    const bool fp = c->mode == NTSCSIM_MODE_FLOAT && enc_preset && !c->no_fast_decode && even_phase && small_plane && D.src_al16 &&
                    D.ghost_taps == 0 && !D.nocolor && D.out_lp == 1 && D.amp == 50 && D.amp_back == 50 && D.dst_al16 && hs_small &&
                    !c->split_vhs &&
                    ((D.vhs && !D.svideo && D.cnoise_k && D.pnoise_k) || (!D.vhs && !D.cnoise_k && !D.pnoise_k));
    // (NTSCSIM_MODE_FLOAT: its short host-frame launches take the role kernels with float filter states -- the FAST32
    //  arithmetic, inside the mode's stated tolerance like every other FAST32 fallback of that mode -- because the float
    //  pipeline's own two-role decoder makes 2.3k calls per second where these make 4.8k)
    // the latency form: the chain as five wavefronts (roles) of one workgroup (ntsc_pipe.hip) -- short launches
    // from the host-frame entry points, the -vhs preset family of the hand-tuned kernels.  NTSCSIM_PIPE=0: A/B switch.
    static const bool pipe_env = !(std::getenv("NTSCSIM_PIPE") && std::getenv("NTSCSIM_PIPE")[0] == '0');
    // (odd scanline phases -- -comp-phase 90 / 270, odd offsets --: the XA instantiation, wrap-around loads whatever the displacement)
    const bool pipe_xa = !even_phase && !D.svideo && fast_plane_ok((size_t)D.Rpad, W, D.hs ? 2 : 0);
    // (the pre-emphasis presets -- composite pre-emphasis on, subcarrier_amplitude_back raised --: the CATV instantiation)
    const bool pipe_catv = enc_preset_pre && D.amp_back != 50 && D.amp_back >= 2 && even_phase && !D.svideo &&
                           fast_plane_ok((size_t)D.Rpad, W, D.hs ? 2 : 0);
    // NTSCSIM_PIPE_ALWAYS=1 (developer A/B switch): device-resident batches of any length take the role kernels too
    static const bool pipe_always = std::getenv("NTSCSIM_PIPE_ALWAYS") && std::getenv("NTSCSIM_PIPE_ALWAYS")[0] == '1';
    const bool pipe_form = (c->latency_form || pipe_always) && pipe_env && (pipe_always || n <= pipe_max_fields()) && !c->no_fast_decode && D.src_al16 && D.ghost_taps == 0 &&
                           !D.nocolor && D.out_lp == 1 && D.amp == 50 && D.dst_al16 && !c->split_vhs && D.vhs && D.cnoise_k && D.pnoise_k &&
                           (pipe_catv || (enc_preset && D.amp_back == 50 && (even_phase ? small_plane : pipe_xa)));
                           // (S-Video out of the VCR: the SV instantiation)
    // (head-switch displacements beyond W/10 -- PAL's default switching point -- take its wrap-around form; small_plane
    //  has been computed for the displacement range at hand)
    const bool pipe_wr = pipe_form && !hs_small;
    // ... and the default preset (no VCR) as three roles: encoder | TV front | TV back
    const bool pipe_tv = (c->latency_form || pipe_always) && pipe_env && (pipe_always || n <= pipe_max_fields()) &&
                         enc_preset && !c->no_fast_decode && even_phase && small_plane && D.src_al16 && D.ghost_taps == 0 &&
                         !D.nocolor && D.out_lp == 1 && D.amp == 50 && D.amp_back == 50 && D.dst_al16 && hs_small && !c->force_generic &&
                         !D.vhs && !D.cnoise_k && !D.pnoise_k;
    // ... and the pre-emphasis presets without the VCR (-comp-catv ... -comp-catv4; chroma phase noise on or off)
    const bool pipe_tv_catv = c->latency_form && pipe_env && n <= pipe_max_fields() &&
                              enc_preset_pre && !c->no_fast_decode && even_phase && small_plane && D.src_al16 && D.ghost_taps == 0 &&
                              !D.nocolor && D.out_lp == 1 && D.amp == 50 && D.amp_back != 50 && D.amp_back >= 2 && D.dst_al16 && hs_small &&
                              !D.vhs && !D.cnoise_k;
    if ((pipe_tv || pipe_tv_catv || pipe_form) && !c->pipe_fault) {
        HIPCHK(c, hipHostMalloc((void **)&c->pipe_fault, 64, hipHostMallocDefault));
        *c->pipe_fault = 0u;
    }
    if (pipe_tv || pipe_tv_catv) {
        note_kernel(c, pipe_tv_catv ? (fast ? "k_field_pipe_tv_catv<float>" : "k_field_pipe_tv_catv<double>")
                                    : (fast ? "k_field_pipe_tv<float>" : "k_field_pipe_tv<double>"));
#define NTSC_LAUNCH_PIPE_TV(RT, CATV)                                                                                            \
        hipLaunchKernelGGL((k_field_pipe_tv<RT, CATV>), dgrid, dim3(192), 0, st, D, fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p, \
                           c->hs_shift.p, c->dropout.p, c->pn_noise.p, G.ptab, c->pipe_fault)
        if (pipe_tv_catv) { if (fast) NTSC_LAUNCH_PIPE_TV(float, true); else NTSC_LAUNCH_PIPE_TV(double, true); }
        else { if (fast) NTSC_LAUNCH_PIPE_TV(float, false); else NTSC_LAUNCH_PIPE_TV(double, false); }
#undef NTSC_LAUNCH_PIPE_TV
    } else if (pipe_form) {
        // (developer switch: which wavefront of the workgroup takes which role, one hex digit per wavefront -- ntsc_pipe.hip)
        static const unsigned order = std::getenv("NTSCSIM_PIPE_ORDER") ? (unsigned)std::strtoul(std::getenv("NTSCSIM_PIPE_ORDER"), nullptr, 16)
                                                                        : NTSC_PIPE_ORDER;
        // NTSCSIM_PIPE_TIMING=1 (developer probe): every role's start / end / polling time of the first workgroups, on stderr
        static const bool pipe_timing = std::getenv("NTSCSIM_PIPE_TIMING") != nullptr;
        static unsigned long long *dbg = nullptr;
        if (pipe_timing && !dbg) { HIPCHK(c, hipMalloc((void **)&dbg, (size_t)4096 * 15 * sizeof(unsigned long long))); }
        const bool pipe_sv = D.svideo != 0;
        note_kernel(c, pipe_catv ? (fast ? "k_field_pipe_catv<float>" : "k_field_pipe_catv<double>")
                       : !even_phase ? (fast ? "k_field_pipe_xi<float>" : "k_field_pipe_xi<double>")
                       : pipe_sv ? (fast ? "k_field_pipe_sv<float>" : "k_field_pipe_sv<double>")
                               : pipe_wr ? (fast ? "k_field_pipe<float,true>" : "k_field_pipe<double,true>") : (fast ? "k_field_pipe<float>" : "k_field_pipe<double>"));
#define NTSC_LAUNCH_PIPE(RT, ...)                                                                                                \
        hipLaunchKernelGGL((k_field_pipe<RT, __VA_ARGS__>), dgrid, dim3(320), 0, st, D, G, fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p, \
                           c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p, c->dropout.p, c->tails.p, order, dbg, c->pipe_fault)
        if (pipe_catv) { if (fast) NTSC_LAUNCH_PIPE(float, true, false, false, true); else NTSC_LAUNCH_PIPE(double, true, false, false, true); }
        else if (!even_phase) { if (fast) NTSC_LAUNCH_PIPE(float, true, false, true); else NTSC_LAUNCH_PIPE(double, true, false, true); }
        else if (pipe_sv && pipe_wr) { if (fast) NTSC_LAUNCH_PIPE(float, true, true); else NTSC_LAUNCH_PIPE(double, true, true); }
        else if (pipe_sv) { if (fast) NTSC_LAUNCH_PIPE(float, false, true); else NTSC_LAUNCH_PIPE(double, false, true); }
        else if (pipe_wr) { if (fast) NTSC_LAUNCH_PIPE(float, true); else NTSC_LAUNCH_PIPE(double, true); }
        else { if (fast) NTSC_LAUNCH_PIPE(float, false); else NTSC_LAUNCH_PIPE(double, false); }
#undef NTSC_LAUNCH_PIPE
        if (pipe_timing && dgrid.x <= 4096) {
            static int shown = 0;
            std::vector<unsigned long long> h((size_t)dgrid.x * 15);
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, hipMemcpy(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            if (shown++ % 100 == 20) {
                static const char *const names[5] = {"ENC", "SEP", "CHR", "LUM", "OUT"};
                unsigned long long t0 = ~0ull;
                for (size_t i = 0; i < h.size(); i += 3) t0 = h[i] < t0 ? h[i] : t0;
                for (unsigned b = 0; b < dgrid.x && b < 4; b++)
                    for (int r = 0; r < 5; r++) {
                        const unsigned long long *o = &h[((size_t)b * 5 + r) * 3];
                        const unsigned hw = (unsigned)(o[2] >> 32);
                        std::fprintf(stderr, "pipe_timing wg %u %s start %.2f us end %.2f us polling %.2f us  simd %u cu %u se %u\n", b, names[r],
                                     (double)(o[0] - t0) / 100.0, (double)(o[1] - t0) / 100.0, (double)(o[2] & 0xffffffffu) / 100.0,
                                     (hw >> 4) & 3u, (hw >> 8) & 15u, (hw >> 13) & 7u);
                    }
            }
        }
    } else if (fp) {
        note_kernel(c, "k_encode_fp");
        launch_encode_fp(st, D, fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
    } else if (ghost_fused) {
        const bool two = D.ghost_taps <= 2;
        note_kernel(c, (std::string("k_encode_fast_gh<") + (fast ? "float," : "double,") + (two ? "2>" : "4>")).c_str());
#define NTSC_LAUNCH_GH(RT, GT)                                                                  \
        hipLaunchKernelGGL((k_encode_fast_gh<RT, GT>), dim3((D.R + 63) / 64), dim3(64), 0, st, D, \
                           fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p)
        if (fast) { if (two) NTSC_LAUNCH_GH(float, 2); else NTSC_LAUNCH_GH(float, 4); }
        else { if (two) NTSC_LAUNCH_GH(double, 2); else NTSC_LAUNCH_GH(double, 4); }
#undef NTSC_LAUNCH_GH
    } else if (enc_preset && !c->no_fast_decode && even_phase && small_plane && D.src_al16) {
        // hand-tuned encoder of the presets (ntsc_encode_fast.hip)
        note_kernel(c, fast ? "k_encode_fast<float>" : "k_encode_fast<double>");
        if (fast) hipLaunchKernelGGL((k_encode_fast<float>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,
                                     fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
        else hipLaunchKernelGGL((k_encode_fast<double>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,
                                fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
    } else if (enc_preset && !c->no_fast_decode && !even_phase && small_plane && D.src_al16) {
        // ... with scanline phases of either parity
        note_kernel(c, fast ? "k_encode_fast_xi<float>" : "k_encode_fast_xi<double>");
        if (fast) hipLaunchKernelGGL((k_encode_fast_xi<float>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,
                                     fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
        else hipLaunchKernelGGL((k_encode_fast_xi<double>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,
                                fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
    } else if (enc_preset_pre && !c->no_fast_decode && even_phase && small_plane && D.src_al16) {
        note_kernel(c, fast ? "k_encode_fast_pre<float>" : "k_encode_fast_pre<double>");
        if (fast) hipLaunchKernelGGL((k_encode_fast_pre<float>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,
                                     fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
        else hipLaunchKernelGGL((k_encode_fast_pre<double>), dim3((D.R + 63) / 64), dim3(64), 0, st, D,
                                fields_dev, c->rs_luma.p, c->n0_luma.p, c->comp.p);
    } else if (enc_preset) { if (fast) NTSC_LAUNCH_ENCODE(F_LNOISE, float); else NTSC_LAUNCH_ENCODE(F_LNOISE, double); }
    else { if (fast) NTSC_LAUNCH_ENCODE(F_GENERIC, float); else NTSC_LAUNCH_ENCODE(F_GENERIC, double); }
#undef NTSC_LAUNCH_ENCODE
    // extension: ghosting between encoder and decoder (reads the raw plane, writes a second one)
    const int *dec_in = c->comp.p;
    if (D.ghost_taps > 0 && !ghost_fused) {
        note_kernel(c, "k_ghost");
        HIPCHK(c, c->comp_ghost.ensure((size_t)D.Rpad * W));
        hipLaunchKernelGGL(k_ghost, dim3((D.R + 1023) / 1024, (unsigned)(W + NTSC_GHOST_XT - 1) / NTSC_GHOST_XT), dim3(256), 0, st, D,
                           c->comp.p, c->comp_ghost.p);
        dec_in = c->comp_ghost.p;
    }
    if (evs) HIPCHK(c, hipEventRecord(evs->e[2], st));
    // (the hand-tuned decoder also exists for a subcarrier_amplitude_back other than 50 -- what the pre-emphasis
    // presets set -- as its BK forms; the template forms of the generic kernel fold (c * 50) / 50 away)
    const bool back50 = D.amp_back == 50;
    const bool dec_base = !c->force_generic && !D.nocolor && D.out_lp == 1 && D.amp == 50;
    const bool dec_common = dec_base && back50;
#define NTSC_LAUNCH_DECODE_RT(VHS, CO, F, RT)                                                   \
    do { note_kernel(c, ("k_decode<" #VHS "," #CO "," + std::to_string((unsigned)(F)) + "u," #RT ">").c_str()); \
    hipLaunchKernelGGL((k_decode<VHS, CO, F, RT>), dgrid, dim3(64), 0, st, D, G, fields_dev,     \
                       dec_in, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,           \
                       c->pn_noise.p, c->dropout.p, c->tails.p); } while (0)
#define NTSC_LAUNCH_DECODE(VHS, CO, F)                                                          \
    do { if (fast) NTSC_LAUNCH_DECODE_RT(VHS, CO, F, float);                                     \
         else NTSC_LAUNCH_DECODE_RT(VHS, CO, F, double); } while (0)
    // hand-tuned decoder of the two presets (ntsc_decode_fast.hip) when its preconditions hold
    // (the one-launch VHS form also exists with wrap-around head-switch loads; every other fast form
    // needs the displacement to stay within W/10 samples)
    const bool dec_fast = dec_base && (back50 || D.amp_back >= 2) && !c->no_fast_decode && D.dst_al16 && even_phase && small_plane;
    // the same family with scanline phases of either parity: its own form (any head-switch displacement)
    const bool dec_fast_xi = dec_base && back50 && !c->no_fast_decode && D.dst_al16 && !even_phase && D.vhs && !D.svideo &&
                             D.cnoise_k && D.pnoise_k && fast_plane_ok((size_t)D.Rpad, W, D.hs ? 2 : 0);
#define NTSC_LAUNCH_FAST(VHS, RT)                                                                \
    do { note_kernel(c, "k_decode_fast<" #VHS "," #RT ">");                                      \
    hipLaunchKernelGGL((k_decode_fast<VHS, RT>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in, \
                       c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,       \
                       c->dropout.p, c->tails.p); } while (0)
#define NTSC_LAUNCH_FAST_BK(VHS, RT)                                                             \
    do { note_kernel(c, "k_decode_fast_bk<" #VHS "," #RT ">");                                   \
    hipLaunchKernelGGL((k_decode_fast_bk<VHS, RT>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in, \
                       c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,       \
                       c->dropout.p, c->tails.p); } while (0)
#define NTSC_LAUNCH_FAST_WR(RT)                                                                  \
    do { note_kernel(c, "k_decode_fast<true," #RT ",true>");                                     \
    hipLaunchKernelGGL((k_decode_fast<true, RT, true>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in, \
                       c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,       \
                       c->dropout.p, c->tails.p); } while (0)
    if (pipe_form || pipe_tv || pipe_tv_catv) {
        // (encoder and decoder ran as one launch above)
    } else if (fp) {
        static const int fpv_env = std::getenv("NTSCSIM_FP_VARIANT") ? std::atoi(std::getenv("NTSCSIM_FP_VARIANT")) : -1;    // developer A/B switch
        const int fpv = fpv_env >= 0 ? fpv_env : (n <= 128 ? 0 : 10);      // (launch_decode_fp's own rule, for the kernel's name)
        note_kernel(c, D.vhs ? (fpv < 10 ? "k_decode_fp2" : "k_decode_fp<true>") : "k_decode_fp<false>");
        launch_decode_fp(st, D, G, fields_dev, dec_in, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                         c->dropout.p, c->tails.p, fpv);
    } else if (dec_fast && back50 && hs_small && D.vhs && !D.svideo && D.cnoise_k && D.pnoise_k && c->split_vhs) {
        // VCR half -> second composite plane -> TV half (= the non-VHS decoder without head switching)
        HIPCHK(c, c->comp_vcr.ensure((size_t)D.Rpad * W));
        note_kernel(c, fast ? "k_vcr_front<float>" : "k_vcr_front<double>");
        note_kernel(c, fast ? "k_decode_fast<false,float>" : "k_decode_fast<false,double>");
        hipStream_t sv = st, stv = st;
        if (role_probe) {
            // the encoder is on `st` already; the two decoder halves start when the setup kernel has finished, beside it
            if (!c->s_up) HIPCHK(c, hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
            if (!c->s_dn) HIPCHK(c, hipStreamCreateWithFlags(&c->s_dn, hipStreamNonBlocking));
            for (auto &e_ : c->ev_role) if (!e_) HIPCHK(c, hipEventCreateWithFlags(&e_, hipEventDisableTiming));
            sv = c->s_up; stv = c->s_dn;
            if (evs) { HIPCHK(c, hipStreamWaitEvent(sv, evs->e[1], 0)); HIPCHK(c, hipStreamWaitEvent(stv, evs->e[1], 0)); }
        }
        if (fast) hipLaunchKernelGGL((k_vcr_front<float>), dgrid, dim3(64), 0, sv, D, G, fields_dev, dec_in,
                                     c->comp_vcr.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                                     c->pn_noise.p, c->tails.p);
        else hipLaunchKernelGGL((k_vcr_front<double>), dgrid, dim3(64), 0, sv, D, G, fields_dev, dec_in,
                                c->comp_vcr.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                                c->pn_noise.p, c->tails.p);
        DevParams D2 = D;
        D2.hs = 0;
        const int *tv_in = c->comp_vcr.p;
        if (fast) hipLaunchKernelGGL((k_decode_fast<false, float>), dgrid, dim3(64), 0, stv, D2, G, fields_dev,
                                     tv_in, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                                     c->pn_noise.p, c->dropout.p, c->tails.p);
        else hipLaunchKernelGGL((k_decode_fast<false, double>), dgrid, dim3(64), 0, stv, D2, G, fields_dev,
                                tv_in, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                                c->pn_noise.p, c->dropout.p, c->tails.p);
        if (role_probe) {
            HIPCHK(c, hipEventRecord(c->ev_role[0], sv)); HIPCHK(c, hipEventRecord(c->ev_role[1], stv));
            HIPCHK(c, hipStreamWaitEvent(st, c->ev_role[0], 0)); HIPCHK(c, hipStreamWaitEvent(st, c->ev_role[1], 0));
        }
    } else if (!c->force_generic && !D.nocolor && D.out_lp == 2 && D.amp == 50 && back50 && !c->no_fast_decode && D.dst_al16 &&
               even_phase && D.vhs && !D.svideo && D.cnoise_k && D.pnoise_k && fast_plane_ok((size_t)D.Rpad, W, D.hs ? 2 : 0)) {
        // the -vhs family with the FULL output chroma low-pass
        note_kernel(c, fast ? "k_decode_fast_fo<float>" : "k_decode_fast_fo<double>");
        if (fast) hipLaunchKernelGGL((k_decode_fast_fo<float>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in,
                                     c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                                     c->dropout.p, c->tails.p);
        else hipLaunchKernelGGL((k_decode_fast_fo<double>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in,
                                c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                                c->dropout.p, c->tails.p);
    } else if (dec_fast_xi) {
        note_kernel(c, fast ? "k_decode_fast_xi<float>" : "k_decode_fast_xi<double>");
        if (fast) hipLaunchKernelGGL((k_decode_fast_xi<float>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in,
                                     c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                                     c->dropout.p, c->tails.p);
        else hipLaunchKernelGGL((k_decode_fast_xi<double>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in,
                                c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                                c->dropout.p, c->tails.p);
    } else if (dec_fast && back50 && hs_small && D.vhs && !D.svideo && D.cnoise_k && D.pnoise_k) {
        if (fast) NTSC_LAUNCH_FAST(true, float); else NTSC_LAUNCH_FAST(true, double);
    } else if (dec_fast && back50 && D.vhs && !D.svideo && D.cnoise_k && D.pnoise_k) {
        if (fast) NTSC_LAUNCH_FAST_WR(float); else NTSC_LAUNCH_FAST_WR(double);
    } else if (dec_fast && !back50 && D.vhs && !D.svideo && D.cnoise_k && D.pnoise_k &&
               fast_plane_ok((size_t)D.Rpad, W, D.hs ? 2 : 0)) {
        // (one form for both displacement ranges: the wrap-around loads cost a sign test per load)
        if (fast) NTSC_LAUNCH_FAST_BK(true, float); else NTSC_LAUNCH_FAST_BK(true, double);
    } else if (dec_fast && back50 && D.vhs && D.svideo && D.cnoise_k && D.pnoise_k &&
               fast_plane_ok((size_t)D.Rpad, W, D.hs ? 2 : 0)) {
        note_kernel(c, fast ? "k_decode_fast_sv<float>" : "k_decode_fast_sv<double>");
        if (fast) hipLaunchKernelGGL((k_decode_fast_sv<float>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in,
                                     c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                                     c->dropout.p, c->tails.p);
        else hipLaunchKernelGGL((k_decode_fast_sv<double>), dgrid, dim3(64), 0, st, D, G, fields_dev, dec_in,
                                c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p,
                                c->dropout.p, c->tails.p);
    } else if (dec_fast && back50 && hs_small && !D.vhs && !D.cnoise_k && !D.pnoise_k) {
        if (fast) NTSC_LAUNCH_FAST(false, float); else NTSC_LAUNCH_FAST(false, double);
    } else if (dec_fast && !back50 && hs_small && !D.vhs && !D.cnoise_k && !D.pnoise_k) {
        if (fast) NTSC_LAUNCH_FAST_BK(false, float); else NTSC_LAUNCH_FAST_BK(false, double);
    } else
#undef NTSC_LAUNCH_FAST
#undef NTSC_LAUNCH_FAST_WR
#undef NTSC_LAUNCH_FAST_BK
    if (!D.vhs) {
        if (dec_common && !D.cnoise_k && !D.pnoise_k) NTSC_LAUNCH_DECODE(false, false, 0u);
        else NTSC_LAUNCH_DECODE(false, false, F_GENERIC);
    } else if (D.svideo) {
        NTSC_LAUNCH_DECODE(true, false, F_GENERIC);
    } else {
        if (dec_common && D.cnoise_k && D.pnoise_k) NTSC_LAUNCH_DECODE(true, true, (F_CNOISE | F_PNOISE));
        else NTSC_LAUNCH_DECODE(true, true, F_GENERIC);
    }
#undef NTSC_LAUNCH_DECODE
#undef NTSC_LAUNCH_DECODE_RT
    if (evs) HIPCHK(c, hipEventRecord(evs->e[3], st));
    if (any_bob) {
        note_kernel(c, "k_bob");
        hipLaunchKernelGGL(k_bob, dim3((H + 1) / 2, n), dim3(256), 0, st, D, fields_dev);
    }
    if (evs) HIPCHK(c, hipEventRecord(evs->e[4], st));
    HIPCHK(c, hipGetLastError());
    c->last_n = n; c->last_W = W; c->last_H = H; c->last_Rpad = D.Rpad; c->last_Lslot = D.Lslot;
    return NTSCSIM_OK;
}

static int take_events(ntscsim_ctx *c, ntscsim_ctx::EvSet &evs)
{
    if (!c->ev_free.empty()) { evs = c->ev_free.back(); c->ev_free.pop_back(); }
    else for (int i = 0; i < 5; i++) HIPCHK(c, hipEventCreate(&evs.e[i]));
    return NTSCSIM_OK;
}

extern "C" int ntscsim_fields_device(ntscsim_ctx *c, const ntscsim_field_desc *descs, int n,
                                     int W, int H, void *hip_stream)
{
    if (!c || (n > 0 && !descs)) return NTSCSIM_E_ARG;
    if (n == 0) return NTSCSIM_OK;
    if (n < 0) return NTSCSIM_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;

    if (c->pipe_fault && *c->pipe_fault) {      // raised by an earlier launch of this ctx (ntsc_pipe.hip: a hand-off timed out)
        c->err = "k_field_pipe: hand-off timed out in workgroup " + std::to_string(*c->pipe_fault - 1u);
        *c->pipe_fault = 0u;
        return NTSCSIM_E_HIP;
    }
    // pinned staging for the records, double-buffered against the asynchronous upload
    const int si = c->stage_idx;
    c->stage_idx ^= 1;
    if (c->stage_used[si]) HIPCHK(c, hipEventSynchronize(c->stage_ev[si]));
    if (c->stage_cap[si] < (size_t)n) {
        if (c->stage[si]) (void)hipHostFree(c->stage[si]);
        c->stage[si] = nullptr; c->stage_cap[si] = 0;
        const size_t want = (size_t)n + (size_t)n / 4 + 16;
        HIPCHK(c, hipHostMalloc((void **)&c->stage[si], want * sizeof(FieldDev), hipHostMallocDefault));
        c->stage_cap[si] = want;
    }
    DevParams D;
    bool any_bob = false;
    uint64_t rng_end = c->rng_pos;
    int rc = prepare_records(c, descs, n, W, H, D, c->stage[si], any_bob, rng_end);
    if (rc != NTSCSIM_OK) return rc;
    HIPCHK(c, c->fields.ensure((size_t)n));

    ntscsim_ctx::EvSet evs;
    const bool prof = c->profiling;
    if (prof) {
        rc = take_events(c, evs);
        if (rc != NTSCSIM_OK) return rc;
        HIPCHK(c, hipEventRecord(evs.e[0], st));
    }
    // Short launches of the host-frame entry points read their records where the host wrote them: the pinned staging
    // buffer is the GPU's to address, a handful of 288-byte records is not worth a copy kernel and the dependency behind it
    // (the staging slot is then busy until the kernels have run: the event follows them).  NTSCSIM_RECORDS_INPLACE=0: A/B.
    static const bool inplace_env = !(std::getenv("NTSCSIM_RECORDS_INPLACE") && std::getenv("NTSCSIM_RECORDS_INPLACE")[0] == '0');
    // (ntscsim_set_launch_form(NTSCSIM_FORM_LATENCY): a caller of this entry point asked for the role kernels)
    struct FormGuard {      // (every return below restores the flag)
        ntscsim_ctx *c; bool on;
        FormGuard(ntscsim_ctx *c_, bool on_) : c(c_), on(on_) { if (on) c->latency_form = true; }
        ~FormGuard() { if (on) c->latency_form = false; }
    } form_guard(c, c->latency_pref && !c->latency_form && n <= pipe_max_fields());
    const bool inplace = inplace_env && c->latency_form && n <= pipe_max_fields();
    if (!inplace) {
        HIPCHK(c, hipMemcpyAsync(c->fields.p, c->stage[si], (size_t)n * sizeof(FieldDev),
                                 hipMemcpyHostToDevice, st));
        HIPCHK(c, hipEventRecord(c->stage_ev[si], st));
    }
    c->stage_used[si] = true;
    c->setup_host_rec = c->stage[si];
    rc = launch_records(c, D, inplace ? c->stage[si] : c->fields.p, any_bob, st, prof ? &evs : nullptr);
    c->setup_host_rec = nullptr;
    if (inplace) HIPCHK(c, hipEventRecord(c->stage_ev[si], st));
    if (rc != NTSCSIM_OK) return rc;
    if (prof) c->ev_live.push_back(evs);
    c->rng_pos = rng_end;
    return NTSCSIM_OK;
}

// ---- prepared batches: validate + derive + upload once, launch many times ------------------
struct ntscsim_batch {
    ntscsim_ctx *ctx;
    DevParams D;
    DevBuf<FieldDev> records;
    bool any_bob;
    uint64_t rng_end;
};

extern "C" int ntscsim_batch_create(ntscsim_ctx *c, const ntscsim_field_desc *descs, int n, int W,
                                    int H, ntscsim_batch **out)
{
    if (!c || !descs || !out || n <= 0) return NTSCSIM_E_ARG;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<FieldDev> host((size_t)n);
    ntscsim_batch *b = new (std::nothrow) ntscsim_batch();
    if (!b) return NTSCSIM_E_NOMEM;
    b->ctx = c;
    b->rng_end = c->rng_pos;
    int rc = prepare_records(c, descs, n, W, H, b->D, host.data(), b->any_bob, b->rng_end);
    if (rc != NTSCSIM_OK) { delete b; return rc; }
    if (b->records.ensure((size_t)n) != hipSuccess ||
        upload_table(b->records.p, host.data(), (size_t)n * sizeof(FieldDev)) !=
            hipSuccess) {
        b->records.release();
        delete b;
        c->err = "ntscsim_batch_create: device allocation / upload failed";
        return NTSCSIM_E_HIP;
    }
    *out = b;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_batch_run(ntscsim_batch *b, void *hip_stream)
{
    if (!b) return NTSCSIM_E_ARG;
    ntscsim_ctx *c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    ntscsim_ctx::EvSet evs;
    const bool prof = c->profiling;
    if (prof) {
        int rc = take_events(c, evs);
        if (rc != NTSCSIM_OK) return rc;
        HIPCHK(c, hipEventRecord(evs.e[0], st));
    }
    int rc = launch_records(c, b->D, b->records.p, b->any_bob, st, prof ? &evs : nullptr);
    if (rc != NTSCSIM_OK) return rc;
    if (prof) c->ev_live.push_back(evs);
    c->rng_pos = b->rng_end;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_batch_destroy(ntscsim_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipDeviceSynchronize();
    b->records.release();
    delete b;
}

// ---- the 8-bit YUV422P sibling (ffmpeg_to_composite.cpp) ------------------------------------
static double alpha_rate(double rate, double hz)
{
    const double timeInterval = 1.0 / rate;
    const double tau = 1 / (hz * 2 * M_PI);
    return timeInterval / (tau + timeInterval);
}

// One ntscsim_fields422_device() call in two halves, so that a prepared batch (ntscsim_batch422_*) can
// repeat the second: everything derived from the descriptors ...
struct Prep422 {
    DevParams D;
    bool any_render = false, any_flt = false;
    double a_hp_i = 0, a_hp_q = 0, a_sh_c = 0;
    uint64_t rng_end = 0;          // in: position of the first descriptor with NTSCSIM_RNG_AUTO; out: after the last
    int n = 0, W = 0, H = 0;
    int max_ls = 0;                // largest destination linesize of the batch (the pitch of the halo-row copies)
};
static int prepare422(ntscsim_ctx *c, const ntscsim_field422_desc *descs, int n, int W, int H, Prep422 &P,
                      FieldDev *host_fields, Field422Dev *host_fields422)
{
    if (W < 16 || (W & 1) || H < 2 || W > 16384 || H > 16384) return NTSCSIM_E_SIZE;   // 4:2:2
    if (n > 65535) return NTSCSIM_E_SIZE;                 // render / black-key grids: one plane per field
    P.n = n; P.W = W; P.H = H;
    const ntscsim_params &p = c->prm;

    DevParams &D = P.D;
    fill_dev_params(p, D);
    if (c->warm_override[0] > 0) D.warm_luma = c->warm_override[0];
    if (c->warm_override[1] > 0) D.warm_chroma = c->warm_override[1];
    D.variant = 1;
    D.W = W; D.H = H; D.Lslot = (H + 1) / 2; D.nfields = n;
    const long long R = (long long)n * D.Lslot;
    if (R > (1ll << 30)) return NTSCSIM_E_SIZE;
    D.R = (int)R;
    D.Rpad = (int)(((R + 63) / 64) * 64 + 64);
    // filter constants of the 8-bit tool: chroma runs at half the luma sample rate
    const double rl = (315000000.00 * 4) / 88, rc2 = (315000000.00 * 4) / (88 * 2);
    double luma_cut = 2400000, chroma_cut = 320000;                     // :793-808
    D.cdelay = 4;
    if (p.output_vhs_tape_speed == NTSCSIM_VHS_LP) { luma_cut = 1900000; chroma_cut = 300000; D.cdelay = 5; }
    if (p.output_vhs_tape_speed == NTSCSIM_VHS_EP) { luma_cut = 1400000; chroma_cut = 280000; D.cdelay = 6; }
    D.a_in_i = alpha_rate(rc2, 1300000);                                // :366-381
    D.a_in_q = alpha_rate(rc2, 600000);
    P.a_hp_i = alpha_rate(rc2, 1300000 / 2.0); P.a_hp_q = alpha_rate(rc2, 600000 / 2.0);
    D.a_tv = alpha_rate(rc2, (315000000.00 * 4) / (88 * 2 * 4));        // lite :408
    D.a_vl = alpha_rate(rl, luma_cut);
    D.a_vc = alpha_rate(rc2, chroma_cut);
    D.a_sh = alpha_rate(rl, luma_cut * 2);                              // :893
    P.a_sh_c = alpha_rate(rc2, chroma_cut * 2);                          // :911
    D.sharpen = p.vhs_out_sharpen;
    D.out_lp = p.composite_out_chroma_lowpass ? 2 : (p.composite_out_chroma_lowpass_lite ? 1 : 0);  // :948-951
    if (p.video_yc_recombine < 0 || p.video_yc_recombine > 64) return NTSCSIM_E_PARAM;

    int rc = build_geometry(c, W, H, D);
    if (rc != NTSCSIM_OK) return rc;
    if (D.pnoise_k) { rc = build_ptab(c); if (rc != NTSCSIM_OK) return rc; }

    uint64_t pos = P.rng_end;
    bool any_render = false, any_flt = false;
    bool al_y16 = true, al_c8 = true;      // vector copies between frame rows and scratch words
    for (int i = 0; i < n; i++) {
        const ntscsim_field422_desc &d = descs[i];
        if (d.field > 1) return NTSCSIM_E_ARG;
        Field422Dev &o = host_fields422[(size_t)i];
        std::memset(&o, 0, sizeof(o));
        for (int k = 0; k < 3; k++) {
            if (!d.dst_dev[k]) return NTSCSIM_E_ARG;
            const int need = k == 0 ? W : W / 2;
            if (d.dst_linesize[k] < need) return NTSCSIM_E_SIZE;
            o.dst[k] = (uint8_t *)d.dst_dev[k]; o.dst_ls[k] = d.dst_linesize[k];
            if (d.dst_linesize[k] > P.max_ls) P.max_ls = d.dst_linesize[k];
            if (d.src_dev[0]) {
                if (!d.src_dev[k] || d.src_linesize[k] < need) return NTSCSIM_E_SIZE;
                o.src[k] = (const uint8_t *)d.src_dev[k]; o.src_ls[k] = d.src_linesize[k];
            }
            if (d.flt_dev[0] && p.black_key_level_feedback >= 0) {
                if (!d.flt_dev[k] || d.flt_linesize[k] < need) return NTSCSIM_E_SIZE;
                o.flt[k] = (uint8_t *)d.flt_dev[k]; o.flt_ls[k] = d.flt_linesize[k];
            }
        }
        if (d.src_dev[0]) {
            const int minh = (d.flags & NTSCSIM_422_INTERLACED) ? 4 : 2;
            if (d.src_height < minh) return NTSCSIM_E_SIZE;
            any_render = true;
        }
        any_flt = any_flt || o.flt[0] != nullptr;
        al_y16 = al_y16 && !(((uintptr_t)d.dst_dev[0] | (uintptr_t)d.dst_linesize[0]) & 15);
        al_c8 = al_c8 && !(((uintptr_t)d.dst_dev[1] | (uintptr_t)d.dst_linesize[1] |
                            (uintptr_t)d.dst_dev[2] | (uintptr_t)d.dst_linesize[2]) & 7);
        o.src_height = d.src_height;
        o.field = d.field; o.flags = d.flags; o.fieldno = d.fieldno;
        if (d.rng_pos != NTSCSIM_RNG_AUTO) pos = d.rng_pos;
        FieldDev &fo = host_fields[(size_t)i];
        std::memset(&fo, 0, sizeof(fo));
        fo.field = d.field; fo.fieldno = d.fieldno;
        const RandState s = ctx_state_at(c, pos);
        std::memcpy(fo.rng, s.w, sizeof(s.w));
        for (int j = 31; j < 61; j++) fo.rng[j] = fo.rng[j - 31] + fo.rng[j - 3];
        if (!(d.flags & NTSCSIM_422_NOCOMP)) pos += c->geom_cur->calls[d.field & 1];
    }
    // Descriptors of one call run concurrently and in place.  Two of them on the same destination frame
    // are a write-write race when they have the same field parity, and a read-write race when the luma
    // rows are tighter than width + 2: the Y/C separator reads two bytes past its row (:496), i.e. the
    // first bytes of the OTHER field's row, which the tool has / has not processed yet depending on the
    // order of its sequential loop (:1783-1800).  Refuse both; separate calls are stream-ordered.
    // (frames are compared by the byte range of their luma plane, not by base pointer: sub-frame views of one
    // allocation and frames whose planes overlap partly race in the same way)
    if (n > 1) {
        struct Key { uintptr_t lo, hi; unsigned field; int ls; unsigned nocomp; };
        std::vector<Key> keys((size_t)n);
        for (int i = 0; i < n; i++) {
            const uintptr_t lo = (uintptr_t)descs[i].dst_dev[0];
            keys[(size_t)i] = {lo, lo + (uintptr_t)descs[i].dst_linesize[0] * (uintptr_t)(H - 1) + (uintptr_t)W,
                               descs[i].field & 1u, descs[i].dst_linesize[0],
                               (descs[i].flags & NTSCSIM_422_NOCOMP) ? 1u : 0u};
        }
        std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
            return a.lo != b.lo ? a.lo < b.lo : a.field < b.field; });
        // sweep: every descriptor against the ones whose range is still open
        size_t open0 = 0;
        for (size_t i = 1; i < keys.size(); i++) {
            while (open0 < i && keys[open0].hi <= keys[i].lo) {
                // (ranges sorted by start: one that ended before this start can still be followed by longer ones --
                //  only skip a prefix of closed ranges, the inner loop tests the rest)
                open0++;
            }
            for (size_t j = open0; j < i; j++) {
                const Key &a = keys[j], &b = keys[i];
                if (a.hi <= b.lo) continue;
                if (a.lo != b.lo || a.ls != b.ls) {
                    // views of one allocation with the same linesize whose pixel columns (plus the separator's two bytes
                    // behind each row) never meet -- side-by-side tiles, row-interleaved frames with a doubled linesize --
                    // write disjoint bytes: no race
                    if (a.ls == b.ls && a.ls > 0) {
                        const uintptr_t d = (b.lo - a.lo) % (uintptr_t)a.ls;       // b's first column inside a's rows
                        const uintptr_t span = (uintptr_t)W + 2u;
                        if (d >= span && d + span <= (uintptr_t)a.ls) continue;
                    }
                    c->err = "descriptors of one batch write overlapping destination frames that are not the same frame";
                    return NTSCSIM_E_ARG;
                }
                if (a.field == b.field) {
                    c->err = "descriptors of one batch share a destination frame and field";
                    return NTSCSIM_E_ARG;
                }
                if ((a.ls < W + 2 || b.ls < W + 2) && !(a.nocomp && b.nocomp)) {
                    c->err = "both fields of one destination frame in one batch need dst_linesize[0] >= width + 2 "
                             "(the Y/C separator reads two bytes past each luma row); use separate calls";
                    return NTSCSIM_E_ARG;
                }
            }
        }
    }
    D.src_al16 = al_y16;      // (422 path: luma rows 16-byte aligned)
    D.dst_al16 = al_c8;       // (422 path: chroma rows 8-byte aligned)

    P.any_render = any_render; P.any_flt = any_flt;
    P.rng_end = pos;
    return NTSCSIM_OK;
}

// ... and the launches, on records that are already in device memory
static int launch422(ntscsim_ctx *c, const Prep422 &P, const FieldDev *fields_dev, const Field422Dev *fields422_dev,
                     hipStream_t st, ntscsim_ctx::EvSet *evs)
{
    const ntscsim_params &p = c->prm;
    DevParams D = P.D;
    const int n = P.n, W = P.W, H = P.H;
    const bool any_render = P.any_render, any_flt = P.any_flt;
    const double a_hp_i = P.a_hp_i, a_hp_q = P.a_hp_q, a_sh_c = P.a_sh_c;
    int rc = build_geometry(c, W, H, D);           // (cached per geometry; selects this call's tables)
    if (rc != NTSCSIM_OK) return rc;
    if (D.pnoise_k) { rc = build_ptab(c); if (rc != NTSCSIM_OK) return rc; }
    const dim3 pgrid((D.R + 62) / 63);
    const size_t S = (size_t)pgrid.x * 64;
    const size_t W2 = (size_t)W / 2;
    const size_t Wq = ((size_t)W + 3) / 4 + 2, W2q = (W2 + 3) / 4 + 2;   // words per row (+slack)
    HIPCHK(c, c->scratch422.ensure(S * (2 * Wq + 2 * W2q) + 256));
    if (D.hs) HIPCHK(c, c->hs_shift.ensure((size_t)D.R));
    if (D.pnoise_k) HIPCHK(c, c->pn_noise.ensure((size_t)D.R));
    if (D.loss) HIPCHK(c, c->dropout.ensure((size_t)D.R));
    if (D.noise_k) { HIPCHK(c, c->rs_luma.ensure((size_t)31 * D.Rpad)); HIPCHK(c, c->n0_luma.ensure((size_t)D.Rpad)); }
    if (D.cnoise_k) {
        HIPCHK(c, c->rs_chroma.ensure((size_t)31 * D.Rpad));
        HIPCHK(c, c->n0_u.ensure((size_t)D.Rpad));
        HIPCHK(c, c->n0_v.ensure((size_t)D.Rpad));
    }
    GeomDev G;
    G.lskip = c->geom_cur->lskip.p; G.pskip = c->geom_cur->pskip.p; G.jrow = c->geom_cur->jrow.p;
    G.jwarm = c->geom_cur->jwarm.p; G.sstart = c->geom_cur->sstart.p; G.ptab = c->ptab.p;
    Scratch422 Sc;
    Sc.S = S;
    Sc.Y = c->scratch422.p;
    Sc.T = Sc.Y + S * Wq;
    Sc.U = Sc.T + S * Wq;
    Sc.V = Sc.U + S * W2q;
    // the rows above the workgroups, copied aside before the in-place kernel starts (halo_redirect)
    Sc.halo_pitch = (uint32_t)((((size_t)P.max_ls + 2 + 63) / 64) * 64);
    Sc.halo = nullptr;
    if (pgrid.x > 1) {
        HIPCHK(c, c->halo422.ensure((size_t)pgrid.x * 3 * Sc.halo_pitch + 256));
        Sc.halo = c->halo422.p;
    }

    c->kernels.clear();
    if (any_render) note_kernel(c, "k422_render");
    if (any_flt) note_kernel(c, "k422_bkey");
    // the draws read no pixels: they go first, beside the host-frame engine's upload of the source (ntscsim_host422.hip)
    launch_setup(c, D, G, fields_dev, n, st);
    if (c->wait422_ev) {
        HIPCHK(c, hipStreamWaitEvent(st, c->wait422_ev, 0));
        c->wait422_ev = nullptr;
    }
    if (any_render)
        hipLaunchKernelGGL(k422_render, dim3((unsigned)((2 * W + 255) / 256), (unsigned)D.Lslot, (unsigned)n),
                           dim3(256), 0, st, D, fields422_dev);
    if (any_flt)
        hipLaunchKernelGGL(k422_bkey, dim3((unsigned)((W2 + 255) / 256), (unsigned)D.Lslot, (unsigned)n),
                           dim3(256), 0, st, D, fields422_dev, p.black_key_level_feedback);
    if (Sc.halo)
        hipLaunchKernelGGL(k422_halo, dim3(pgrid.x - 1), dim3(256), 0, st, D, fields422_dev, Sc);
    // (profiling slots: "setup" = render, black key and the per-field / per-row draws, "encode" is
    // empty, "decode" = the one kernel that does composite_video_process)
    if (evs) { HIPCHK(c, hipEventRecord(evs->e[1], st)); HIPCHK(c, hipEventRecord(evs->e[2], st)); }
    // four-sweep form (ntsc422_fused.hip) for the VHS family of option sets, twelve-sweep form otherwise
    const bool family = !c->no_fast_decode && !D.nocolor && D.in_lp &&
                        !p.nocolor_subcarrier_after_yc_sep && p.video_yc_recombine == 0;
    const bool fused = family && D.vhs && !D.svideo;
    // the two families beside it (round 5): no VCR at all -- the tool's default preset -- in two sweeps (k422_short), the
    // VCR with S-Video out as the streamed pass without its re-modulation (k422_fused<false,true,D,true>)
    const bool direct = family && !D.vhs, fused_sv = family && D.vhs && D.svideo;
    // the '-vhs' preset's switch set has its own instantiation (debug bit 1 keeps the general one)
    // = what `ffmpeg_to_composite -vhs` runs: NTSC, SP, no pre-emphasis, all three noises and the FULL
    // output chroma low-pass (ffmpeg_to_composite.cpp:278 default true, selection :948-951)
    const bool spec = !c->split_vhs && D.ntsc && !D.pre_on && D.noise_k && D.cnoise_k && D.pnoise_k && D.out_lp == 2 &&
                      D.cdelay == 4 && D.src_al16 && D.dst_al16;
    // ... and runs its B sweeps as one streamed pass (debug bit 2 keeps the four-sweep preset form)
    // (its steady loop uses identities of an even scanline phase -- NTSC mode 180 with an even offset, or mode 0 --
    // and of subcarrier amplitude 50 both ways: scan_phase422, ntsc422_fused.hip StreamB::iter_fast)
    const bool even422 = D.phase_mode == 180 ? !(D.phase_off & 1) : (D.phase_mode != 90 && D.phase_mode != 270);
    const bool stream = spec && even422 && D.amp == 50 && D.amp_back == 50 && !c->no_stream422;
    // every other switch set of the family gets the same streamed pass with the switches read at run time
    // (aligned frame rows: the 64-byte frame bursts), per chroma delay of the tape speed
    const bool stream_gen = fused && !stream && !c->no_stream422 && !c->split_vhs && D.src_al16 && D.dst_al16 &&
                            D.cdelay >= 4 && D.cdelay <= 6;
    // sweep A of the no-VCR form in the streamed preset's shape where its identities hold (the default preset does)
    const bool fasta = direct && !c->split_vhs && D.ntsc && !D.pre_on && D.noise_k && even422 && D.amp == 50 &&
                       D.src_al16 && D.dst_al16;
    // the VCR with S-Video out: the streamed pass without its re-modulation (aligned rows), else the twelve sweeps
    const bool stream_sv = fused_sv && !c->no_stream422 && !c->split_vhs && D.src_al16 && D.dst_al16 && D.cdelay >= 4 && D.cdelay <= 6;
    static const bool pipe422_env = !(std::getenv("NTSCSIM_PIPE") && std::getenv("NTSCSIM_PIPE")[0] == '0');
    const bool pipe422 = c->latency_form && pipe422_env && n <= pipe_max_fields() &&
                         ((fused && stream) || stream_gen || stream_sv);
    const bool pipe422_direct = c->latency_form && pipe422_env && n <= pipe_max_fields() && direct && fasta;
    if (pipe422_direct) note_kernel(c, "k422_direct_pipe");
    else if (pipe422)
        note_kernel(c, (fused && stream) ? "k422_pipe<true,4>"
                       : stream_sv ? (D.cdelay == 4 ? "k422_pipe_sv<4>" : D.cdelay == 5 ? "k422_pipe_sv<5>" : "k422_pipe_sv<6>")
                                   : (D.cdelay == 4 ? "k422_pipe<false,4>" : D.cdelay == 5 ? "k422_pipe<false,5>" : "k422_pipe<false,6>"));
    else
    note_kernel(c, direct ? (fasta ? "k422_direct_fast" : "k422_direct")
                          : stream_sv ? (D.cdelay == 4 ? "k422_fused_sv<4>" : D.cdelay == 5 ? "k422_fused_sv<5>" : "k422_fused_sv<6>")
                          : !fused ? "k422_process"
                          : stream ? "k422_fused<true,true,4>"
                          : stream_gen ? (D.cdelay == 4 ? "k422_fused<false,true,4>" : D.cdelay == 5 ? "k422_fused<false,true,5>" : "k422_fused<false,true,6>")
                          : spec ? "k422_fused<true,false,4>" : "k422_fused<false,false,4>");
#define NTSC_LAUNCH_422(...)                                                                                  \
    hipLaunchKernelGGL((k422_fused<__VA_ARGS__>), pgrid, dim3(64), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p, \
                       c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,                      \
                       c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma)
    // the latency form of the streamed kernels: sweep A | head-switch gather | front and back of the streamed pass as four wavefronts of one
    // workgroup (k422_pipe) -- short launches of the host-frame engine (ntscsim_field422 / the lanes of ntscsim_submit422)
#define NTSC_LAUNCH_422_PIPE(...)                                                                             \
    hipLaunchKernelGGL((k422_pipe<__VA_ARGS__>), pgrid, dim3(256), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p, \
                       c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,                      \
                       c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma, c->pipe_fault)
    if ((pipe422 || pipe422_direct) && !c->pipe_fault) { HIPCHK(c, hipHostMalloc((void **)&c->pipe_fault, 64, hipHostMallocDefault)); *c->pipe_fault = 0u; }
    if (pipe422_direct) {
        hipLaunchKernelGGL(k422_short_pipe, pgrid, dim3(192), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p, c->n0_luma.p, c->rs_chroma.p,
                           c->n0_u.p, c->n0_v.p, c->hs_shift.p, c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, c->pipe_fault);
    } else if (pipe422) {
        if (fused && stream) NTSC_LAUNCH_422_PIPE(true, 4);
        else if (stream_sv && D.cdelay == 4) NTSC_LAUNCH_422_PIPE(false, 4, true);
        else if (stream_sv && D.cdelay == 5) NTSC_LAUNCH_422_PIPE(false, 5, true);
        else if (stream_sv) NTSC_LAUNCH_422_PIPE(false, 6, true);
        else if (D.cdelay == 4) NTSC_LAUNCH_422_PIPE(false, 4);
        else if (D.cdelay == 5) NTSC_LAUNCH_422_PIPE(false, 5);
        else NTSC_LAUNCH_422_PIPE(false, 6);
    } else
#undef NTSC_LAUNCH_422_PIPE
    if (direct) {
        if (fasta)
            hipLaunchKernelGGL(k422_short<true>, pgrid, dim3(64), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p,
                               c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                               c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma);
        else
            hipLaunchKernelGGL(k422_short<false>, pgrid, dim3(64), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p,
                               c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                               c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma);
    }
    else if (stream_sv && D.cdelay == 4) NTSC_LAUNCH_422(false, true, 4, true);
    else if (stream_sv && D.cdelay == 5) NTSC_LAUNCH_422(false, true, 5, true);
    else if (stream_sv) NTSC_LAUNCH_422(false, true, 6, true);
    else if (fused && stream) NTSC_LAUNCH_422(true, true, 4);
    else if (stream_gen && D.cdelay == 4) NTSC_LAUNCH_422(false, true, 4);
    else if (stream_gen && D.cdelay == 5) NTSC_LAUNCH_422(false, true, 5);
    else if (stream_gen) NTSC_LAUNCH_422(false, true, 6);
#undef NTSC_LAUNCH_422
    else if (fused && spec)
        hipLaunchKernelGGL(k422_fused<true>, pgrid, dim3(64), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p,
                           c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                           c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma);
    else if (fused)
        hipLaunchKernelGGL(k422_fused<false>, pgrid, dim3(64), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p,
                           c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                           c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma);
    else
    hipLaunchKernelGGL(k422_process, pgrid, dim3(64), 0, st, D, G, fields422_dev, Sc, c->rs_luma.p,
                       c->n0_luma.p, c->rs_chroma.p, c->n0_u.p, c->n0_v.p, c->hs_shift.p,
                       c->pn_noise.p, c->dropout.p, a_hp_i, a_hp_q, a_sh_c, p.vhs_out_sharpen_chroma,
                       p.video_yc_recombine, p.nocolor_subcarrier_after_yc_sep);
    HIPCHK(c, hipGetLastError());
    if (evs) {
        HIPCHK(c, hipEventRecord(evs->e[3], st));
        HIPCHK(c, hipEventRecord(evs->e[4], st));
    }
    return NTSCSIM_OK;
}

extern "C" int ntscsim_fields422_device(ntscsim_ctx *c, const ntscsim_field422_desc *descs, int n,
                                        int W, int H, void *hip_stream)
{
    if (!c || (n > 0 && !descs)) return NTSCSIM_E_ARG;
    if (n == 0) return NTSCSIM_OK;
    if (n < 0) return NTSCSIM_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (c->pipe_fault && *c->pipe_fault) {      // raised by an earlier launch of this ctx (k422_pipe: a hand-off timed out)
        c->err = "k422_pipe: hand-off timed out in workgroup " + std::to_string(*c->pipe_fault - 1u);
        *c->pipe_fault = 0u;
        return NTSCSIM_E_HIP;
    }
    // pinned staging for the records, double-buffered against the asynchronous upload
    const int si = c->stage422_idx;
    c->stage422_idx ^= 1;
    if (c->stage422_used[si]) HIPCHK(c, hipEventSynchronize(c->stage422_ev[si]));
    const size_t rec_bytes = sizeof(FieldDev) + sizeof(Field422Dev);
    if (c->stage422_cap[si] < (size_t)n) {
        if (c->stage422[si]) (void)hipHostFree(c->stage422[si]);
        c->stage422[si] = nullptr; c->stage422_cap[si] = 0;
        const size_t want = (size_t)n + (size_t)n / 4 + 16;
        HIPCHK(c, hipHostMalloc((void **)&c->stage422[si], want * rec_bytes, hipHostMallocDefault));
        c->stage422_cap[si] = want;
    }
    FieldDev *const host_fields = (FieldDev *)c->stage422[si];
    Field422Dev *const host_fields422 = (Field422Dev *)(c->stage422[si] + (size_t)n * sizeof(FieldDev));
    Prep422 P;
    P.rng_end = c->rng_pos;
    int rc = prepare422(c, descs, n, W, H, P, host_fields, host_fields422);
    if (rc != NTSCSIM_OK) return rc;
    // (both record arrays lie back to back in the staging buffer: one upload into one device buffer of the same layout)
    static_assert(sizeof(FieldDev) % alignof(Field422Dev) == 0, "records back to back");
    HIPCHK(c, c->recs422.ensure((size_t)n * rec_bytes));
    FieldDev *const dev_fields = reinterpret_cast<FieldDev *>(c->recs422.p);
    Field422Dev *const dev_fields422 = reinterpret_cast<Field422Dev *>(c->recs422.p + (size_t)n * sizeof(FieldDev));
    ntscsim_ctx::EvSet evs;
    const bool prof = c->profiling;
    if (prof) {
        rc = take_events(c, evs);
        if (rc != NTSCSIM_OK) return rc;
        HIPCHK(c, hipEventRecord(evs.e[0], st));
    }
    HIPCHK(c, hipMemcpyAsync(c->recs422.p, c->stage422[si], (size_t)n * rec_bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(c->stage422_ev[si], st));
    c->stage422_used[si] = true;
    // (reading the records in the pinned staging buffer, as ntscsim_fields_device does for short launches, was measured
    //  SLOWER here -- 3.21k against 3.39k calls/s: seven kernels read them, not two)

    c->setup_host_rec = host_fields;
    const bool pref = c->latency_pref && !c->latency_form && n <= pipe_max_fields();      // (ntscsim_set_launch_form)
    if (pref) c->latency_form = true;
    rc = launch422(c, P, dev_fields, dev_fields422, st, prof ? &evs : nullptr);
    if (pref) c->latency_form = false;
    c->setup_host_rec = nullptr;
    if (rc != NTSCSIM_OK) return rc;
    if (prof) c->ev_live.push_back(evs);
    c->rng_pos = P.rng_end;
    return NTSCSIM_OK;
}

// ---- prepared batch of the YUV422P tool: validate + derive + upload once, launch many times
struct ntscsim_batch422 {
    ntscsim_ctx *ctx;
    Prep422 P;
    DevBuf<FieldDev> fields;
    DevBuf<Field422Dev> fields422;
};

extern "C" int ntscsim_batch422_create(ntscsim_ctx *c, const ntscsim_field422_desc *descs, int n, int W, int H,
                                       ntscsim_batch422 **out)
{
    if (!c || !descs || !out || n <= 0) return NTSCSIM_E_ARG;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<FieldDev> hf((size_t)n);
    std::vector<Field422Dev> hf422((size_t)n);
    ntscsim_batch422 *b = new (std::nothrow) ntscsim_batch422();
    if (!b) return NTSCSIM_E_NOMEM;
    b->ctx = c;
    b->P.rng_end = c->rng_pos;
    const int rc = prepare422(c, descs, n, W, H, b->P, hf.data(), hf422.data());
    if (rc != NTSCSIM_OK) { delete b; return rc; }
    if (b->fields.ensure((size_t)n) != hipSuccess || b->fields422.ensure((size_t)n) != hipSuccess ||
        upload_table(b->fields.p, hf.data(), (size_t)n * sizeof(FieldDev)) != hipSuccess ||
        upload_table(b->fields422.p, hf422.data(), (size_t)n * sizeof(Field422Dev)) != hipSuccess) {
        b->fields.release(); b->fields422.release();
        delete b;
        c->err = "ntscsim_batch422_create: device allocation / upload failed";
        return NTSCSIM_E_HIP;
    }
    *out = b;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_batch422_run(ntscsim_batch422 *b, void *hip_stream)
{
    if (!b) return NTSCSIM_E_ARG;
    ntscsim_ctx *c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    ntscsim_ctx::EvSet evs;
    const bool prof = c->profiling;
    if (prof) {
        const int rc = take_events(c, evs);
        if (rc != NTSCSIM_OK) return rc;
        HIPCHK(c, hipEventRecord(evs.e[0], st));
    }
    const int rc = launch422(c, b->P, b->fields.p, b->fields422.p, st, prof ? &evs : nullptr);
    if (rc != NTSCSIM_OK) return rc;
    if (prof) c->ev_live.push_back(evs);
    c->rng_pos = b->P.rng_end;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_batch422_destroy(ntscsim_batch422 *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipDeviceSynchronize();
    b->fields.release(); b->fields422.release();
    delete b;
}

extern "C" int ntscsim_output422_device(ntscsim_ctx *c, const ntscsim_out422_desc *descs, int n,
                                        int W, int H, void *hip_stream)
{
    if (!c || (n > 0 && !descs)) return NTSCSIM_E_ARG;
    if (n == 0) return NTSCSIM_OK;
    if (n < 0) return NTSCSIM_E_ARG;
    if (W < 16 || (W & 1) || H < 2 || W > 16384 || H > 16384) return NTSCSIM_E_SIZE;
    if (n > 65535) return NTSCSIM_E_SIZE;                 // one grid row per descriptor
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->host_out422.resize((size_t)n);
    bool al4 = true;
    for (int i = 0; i < n; i++) {
        const ntscsim_out422_desc &d = descs[i];
        if (d.field > 1 || d.mode > NTSCSIM_OUT422_FRAME) return NTSCSIM_E_ARG;
        Out422Dev &o = c->host_out422[(size_t)i];
        std::memset(&o, 0, sizeof(o));
        for (int k = 0; k < 3; k++) {
            if (!d.frame_dev[k] || !d.bob_dev[k]) return NTSCSIM_E_ARG;
            const int need = k == 0 ? W : W / 2;
            if (d.frame_linesize[k] < need || d.bob_linesize[k] < need) return NTSCSIM_E_SIZE;
            o.frame[k] = (const uint8_t *)d.frame_dev[k]; o.frame_ls[k] = d.frame_linesize[k];
            o.bob[k] = (uint8_t *)d.bob_dev[k]; o.bob_ls[k] = d.bob_linesize[k];
            al4 = al4 && !(((uintptr_t)d.frame_dev[k] | (uintptr_t)d.frame_linesize[k] |
                            (uintptr_t)d.bob_dev[k] | (uintptr_t)d.bob_linesize[k]) & 3);
        }
        o.field = d.field; o.mode = d.mode;
    }
    DevParams D;
    std::memset(&D, 0, sizeof(D));
    D.W = W; D.H = H; D.variant = 1;
    HIPCHK(c, c->out422.ensure((size_t)n));
    // (pageable staging: synchronous with respect to the host vector, like ntscsim_fields422_device)
    HIPCHK(c, hipStreamSynchronize(st));      // (the record buffer may still be read by the previous call's kernel)
    HIPCHK(c, upload_table(c->out422.p, c->host_out422.data(), (size_t)n * sizeof(Out422Dev)));
    hipLaunchKernelGGL(k422_output, dim3((unsigned)H, (unsigned)n), dim3(256), 0, st, D, c->out422.p,
                       al4 ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return NTSCSIM_OK;
}

extern "C" int ntscsim_bgra_to_yuv_device(ntscsim_ctx *c, const ntscsim_yuv_desc *descs, int n,
                                          int W, int H, int pix_fmt, void *hip_stream)
{
    if (!c || (n > 0 && !descs)) return NTSCSIM_E_ARG;
    if (n == 0) return NTSCSIM_OK;
    if (n < 0 || (pix_fmt != NTSCSIM_PIX_YUV420P && pix_fmt != NTSCSIM_PIX_YUV422P)) return NTSCSIM_E_ARG;
    if (W < 2 || (W & 1) || H < 1 || W > 16384 || H > 16384) return NTSCSIM_E_SIZE;
    if (n > 65535) return NTSCSIM_E_SIZE;                 // one grid plane per frame
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->host_yuv.resize((size_t)n);
    bool vec = (W & 7) == 0;
    for (int i = 0; i < n; i++) {
        const ntscsim_yuv_desc &d = descs[i];
        if (!d.bgra_dev || !d.yuv_dev[0] || !d.yuv_dev[1] || !d.yuv_dev[2]) return NTSCSIM_E_ARG;
        if (d.bgra_linesize < 4 * W || (d.bgra_linesize & 3) || ((uintptr_t)d.bgra_dev & 3)) return NTSCSIM_E_SIZE;
        if (d.yuv_linesize[0] < W || d.yuv_linesize[1] < W / 2 || d.yuv_linesize[2] < W / 2) return NTSCSIM_E_SIZE;
        YuvDev &o = c->host_yuv[(size_t)i];
        o.bgra = (const uint8_t *)d.bgra_dev; o.bgra_ls = d.bgra_linesize;
        o.y = (uint8_t *)d.yuv_dev[0]; o.u = (uint8_t *)d.yuv_dev[1]; o.v = (uint8_t *)d.yuv_dev[2];
        o.y_ls = d.yuv_linesize[0]; o.u_ls = d.yuv_linesize[1]; o.v_ls = d.yuv_linesize[2];
        vec = vec && !(((uintptr_t)o.bgra | (uintptr_t)o.bgra_ls) & 15) &&
              !(((uintptr_t)o.y | (uintptr_t)o.y_ls) & 7) &&
              !(((uintptr_t)o.u | (uintptr_t)o.u_ls | (uintptr_t)o.v | (uintptr_t)o.v_ls) & 3);
    }
    DevParams D;
    std::memset(&D, 0, sizeof(D));
    D.W = W; D.H = H;
    HIPCHK(c, c->yuv.ensure((size_t)n));
    HIPCHK(c, hipStreamSynchronize(st));      // (the record buffer may still be read by the previous call's kernel)
    HIPCHK(c, upload_table(c->yuv.p, c->host_yuv.data(), (size_t)n * sizeof(YuvDev)));
    const int v420 = pix_fmt == NTSCSIM_PIX_YUV420P;
    const unsigned rows = v420 ? (unsigned)(H + 1) / 2 : (unsigned)H;
    hipLaunchKernelGGL(k_bgra_to_yuv, dim3((unsigned)((W / 8 + 1 + 127) / 128), rows, (unsigned)n),
                       dim3(128), 0, st, D, c->yuv.p, v420, vec ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return NTSCSIM_OK;
}


static int fill_scale_dev(const ntscsim_scale_desc &d, int W, ScaleDev &o)
{
    if (d.src_format < NTSCSIM_SRC_BGRA || d.src_format > NTSCSIM_SRC_YUV422P) return NTSCSIM_E_ARG;
    if (!d.src_dev[0] || !d.bgra_dev) return NTSCSIM_E_ARG;
    if (d.src_width < 1 || d.src_height < 1 || d.src_width > 16384 || d.src_height > 16384) return NTSCSIM_E_SIZE;
    if (d.bgra_linesize < 4 * W || (d.bgra_linesize & 3) || ((uintptr_t)d.bgra_dev & 3)) return NTSCSIM_E_SIZE;
    std::memset(&o, 0, sizeof(o));
    if (d.src_format == NTSCSIM_SRC_BGRA) {
        if (d.src_linesize[0] < 4 * d.src_width) return NTSCSIM_E_SIZE;
    } else {
        const int cw = (d.src_width + 1) / 2;
        if (!d.src_dev[1] || !d.src_dev[2]) return NTSCSIM_E_ARG;
        if (d.src_linesize[0] < d.src_width || d.src_linesize[1] < cw || d.src_linesize[2] < cw) return NTSCSIM_E_SIZE;
    }
    for (int k = 0; k < 3; k++) { o.src[k] = (const uint8_t *)d.src_dev[k]; o.src_ls[k] = d.src_linesize[k]; }
    o.dst = (uint8_t *)d.bgra_dev; o.dst_ls = d.bgra_linesize;
    o.sw = d.src_width; o.sh = d.src_height; o.fmt = d.src_format;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_scale_to_bgra_device(ntscsim_ctx *c, const ntscsim_scale_desc *descs, int n,
                                            int W, int H, void *hip_stream)
{
    if (!c || (n > 0 && !descs)) return NTSCSIM_E_ARG;
    if (n == 0) return NTSCSIM_OK;
    if (n < 0) return NTSCSIM_E_ARG;
    if (W < 1 || H < 1 || W > 16384 || H > 16384 || n > 65535) return NTSCSIM_E_SIZE;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    c->host_scale.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const int rc = fill_scale_dev(descs[i], W, c->host_scale[(size_t)i]);
        if (rc != NTSCSIM_OK) return rc;
    }
    HIPCHK(c, c->scale.ensure((size_t)n));
    HIPCHK(c, hipStreamSynchronize(st));      // (the record buffer may still be read by the previous call's kernel)
    HIPCHK(c, upload_table(c->scale.p, c->host_scale.data(), (size_t)n * sizeof(ScaleDev)));
    hipLaunchKernelGGL(k_scale_to_bgra, dim3((unsigned)((W + 127) / 128), (unsigned)H, (unsigned)n), dim3(128), 0, st,
                       c->scale.p, W, H);
    HIPCHK(c, hipGetLastError());
    return NTSCSIM_OK;
}

extern "C" int ntscsim_field(ntscsim_ctx *c, const uint8_t *src, int src_ls, int src_interlaced,
                             int src_tff, uint8_t *dst, int dst_ls, int W, int H, unsigned field,
                             uint64_t fieldno)
{
    if (!c || !src || !dst) return NTSCSIM_E_ARG;          // :1578-1579
    if (src_ls < 4 * W || dst_ls < 4 * W) return NTSCSIM_E_SIZE;   // :1580-1581
    if (field > 1) return NTSCSIM_E_ARG;
    if (W < 16 || H < 2) return NTSCSIM_E_SIZE;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->sub) {     // submitted fields come first: same order of dst writes as the synchronous sequence
        const int rc = sub_wait_ticket(c, NTSCSIM_TICKET_ALL);
        if (rc != NTSCSIM_OK) return rc;
    }
    const size_t pitch = (((size_t)W * 4 + 255) / 256) * 256;
    HIPCHK(c, c->fsrc.ensure(pitch * H));
    // A destination frame the GPU can address -- memory the caller declared (ntscsim_host_pin) or that is pinned already
    // (ntscsim_host_frame_alloc / ntscsim_av_frame_get_buffer / hipHostMalloc; asked of the runtime at every call, nothing
    // is registered or remembered here) -- with 16-byte aligned rows is written by the kernels themselves: no device copy
    // of the frame, no download behind the chain.  NTSCSIM_FIELD_DIRECT=0: A/B switch.
    static const bool direct_env = !(std::getenv("NTSCSIM_FIELD_DIRECT") && std::getenv("NTSCSIM_FIELD_DIRECT")[0] == '0');
    uint8_t *dst_dev = nullptr;
    if (direct_env && c->pin_policy > 0 && !(((uintptr_t)dst | (uintptr_t)dst_ls) & 15u))
        dst_dev = pinned_device_ptr(c, dst, (size_t)dst_ls * (size_t)(H - 1) + (size_t)W * 4);
    if (dst_dev && ((uintptr_t)dst_dev & 15u)) dst_dev = nullptr;
    if (!dst_dev) HIPCHK(c, c->fdst.ensure(pitch * H));
    // Only the source rows this field reads go up: row min(y + opposite, H - 1) for y = field, field + 2, ...
    // (ffmpeg_ntsc.cpp:1585-1588, :1599) -- every second row from `first`, plus row H - 1 when the last one is clamped
    // onto it.  They land at their own place in the device frame; the rows in between are never read.
    static const bool full_upload = std::getenv("NTSCSIM_FIELD_FULL_UPLOAD") != nullptr;      // A/B
    const int opposite = src_interlaced ? (src_tff ? 1 : 0) : 0;
    const int first = (int)field + opposite, Lf = (H - (int)field + 1) / 2;
    const int direct = first <= H - 1 ? (H - 1 - first) / 2 + 1 : 0;      // rows first, first + 2, ... <= H - 1
    // The rows travel on a copy stream of their own: the first kernel of the chain (rand() states, head-switch geometry)
    // does not read pixels and runs beside the upload; the encoder waits for `ev_src` (launch_records).
    if (!c->s_up) HIPCHK(c, hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
    if (!c->ev_src) HIPCHK(c, hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming));
    const hipStream_t su = c->s_up;
    // A SOURCE frame the GPU can address (pinned like the destination above) with 16-byte aligned rows is not uploaded at
    // all: the call returns when the field is done, so the encoder role reads the rows where they are -- its loads are a
    // chunk (2 us of its own work) ahead of their use, which covers the link's latency, and the pixels of its guarded steps
    // are requested at the start (encoder_role).  NTSCSIM_FIELD_SRC_DIRECT=0: A/B switch.
    static const bool src_direct_env = !(std::getenv("NTSCSIM_FIELD_SRC_DIRECT") && std::getenv("NTSCSIM_FIELD_SRC_DIRECT")[0] == '0');
    const uint8_t *src_dev = nullptr;
    // (never a source that shares bytes with the destination: the upload is the snapshot such a caller relies on)
    const uintptr_t s0_ = (uintptr_t)src, s1_ = s0_ + (size_t)src_ls * (size_t)(H - 1) + (size_t)W * 4;
    const uintptr_t d0_ = (uintptr_t)dst, d1_ = d0_ + (size_t)dst_ls * (size_t)(H - 1) + (size_t)W * 4;
    const bool apart = s1_ <= d0_ || d1_ <= s0_;
    if (src_direct_env && apart && c->pin_policy > 0 && !(((uintptr_t)src | (uintptr_t)src_ls) & 15u))
        src_dev = pinned_device_ptr(c, src, (size_t)src_ls * (size_t)(H - 1) + (size_t)W * 4);
    if (src_dev && ((uintptr_t)src_dev & 15u)) src_dev = nullptr;
    c->field_stats[0]++;
    if (src_dev) c->field_stats[1]++;
    if (dst_dev) c->field_stats[2]++;
    if (src_dev) {
        // (nothing to wait for)
    } else
    if (full_upload || Lf <= 0) {
        HIPCHK(c, hipMemcpy2DAsync(c->fsrc.p, pitch, src, (size_t)src_ls, (size_t)W * 4, (size_t)H,
                                   hipMemcpyHostToDevice, su));
    } else {
        if (direct > 0)
            HIPCHK(c, hipMemcpy2DAsync(c->fsrc.p + pitch * first, pitch * 2, src + (size_t)src_ls * first, (size_t)src_ls * 2,
                                       (size_t)W * 4, (size_t)(direct < Lf ? direct : Lf), hipMemcpyHostToDevice, su));
        if (direct < Lf)      // the clamped last row
            HIPCHK(c, hipMemcpyAsync(c->fsrc.p + pitch * (H - 1), src + (size_t)src_ls * (H - 1), (size_t)W * 4,
                                     hipMemcpyHostToDevice, su));
    }
    if (!src_dev) {
        HIPCHK(c, hipEventRecord(c->ev_src, su));
        c->src_pending = true;
    }
    ntscsim_field_desc d;
    std::memset(&d, 0, sizeof(d));
    d.src_dev = src_dev ? (void *)src_dev : (void *)c->fsrc.p; d.dst_dev = dst_dev ? dst_dev : c->fdst.p;
    d.src_linesize = src_dev ? src_ls : (int)pitch; d.dst_linesize = dst_dev ? dst_ls : (int)pitch;
    d.field = field;
    d.flags = (src_interlaced ? NTSCSIM_DESC_INTERLACED : 0u) | (src_tff ? NTSCSIM_DESC_TFF : 0u);
    d.fieldno = fieldno;
    d.rng_pos = NTSCSIM_RNG_AUTO;
    c->latency_form = true;
    int rc = ntscsim_fields_device(c, &d, 1, W, H, c->stream);
    c->latency_form = false;
    if (rc != NTSCSIM_OK) { c->src_pending = false; (void)hipStreamSynchronize(su); return rc; }
    // only the rows of this field are written back (:1910-1916)
    const int L = (H - (int)field + 1) / 2;
    if (L > 0 && !dst_dev)
        HIPCHK(c, hipMemcpy2DAsync(dst + (size_t)dst_ls * field, (size_t)dst_ls * 2,
                                   c->fdst.p + pitch * field, pitch * 2, (size_t)W * 4, (size_t)L,
                                   hipMemcpyDeviceToHost, c->stream));
    // (the call waits for its own work only: the next call's setup kernel goes behind the event)
    if (!c->ev_done) HIPCHK(c, hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_done, c->stream));
    speculate_setup(c, c->stream);
    HIPCHK(c, hipEventSynchronize(c->ev_done));
    if (c->pipe_fault && *c->pipe_fault) {
        c->err = "k_field_pipe: hand-off timed out in workgroup " + std::to_string(*c->pipe_fault - 1u);
        *c->pipe_fault = 0u;
        return NTSCSIM_E_HIP;
    }
    return NTSCSIM_OK;
}

// ---- host-frame streaming: the field loop for a run of frames in HOST memory ---------------
// Chunks of frames flow through H2D copy -> kernel chain -> D2H copy on three streams with two
// chunk slots, so the PCIe transfers of neighbouring chunks overlap the kernels (and each other:
// the link is full duplex).  The caller's buffers are pinned in place (hipHostRegister) for the
// duration of the call; if that fails the copies still work, just synchronously.
// Pin two caller buffers in place for the duration of a call (whole pages, explicitly).  Small buffers are not
// worth a registration, and two registrations must never share a page: small heap allocations often do, and
// unpinning the first one then pulls the page from under the second (seen as sporadic aborts inside later,
// unrelated hipMemcpy calls of the process).
struct HostPins {
    uintptr_t s0 = 0, d0 = 0;
    bool pin_src = false, pin_dst = false;
    void pin(const void *src, size_t src_span, void *dst, size_t dst_span, unsigned flags)
    {
        const uintptr_t PG = 4096, MIN_PIN = 1u << 20;
        uintptr_t s1 = ((uintptr_t)src + src_span + PG - 1) & ~(PG - 1);
        uintptr_t d1 = ((uintptr_t)dst + dst_span + PG - 1) & ~(PG - 1);
        s0 = (uintptr_t)src & ~(PG - 1);
        d0 = (uintptr_t)dst & ~(PG - 1);
        bool want_src = src_span >= MIN_PIN, want_dst = dst_span >= MIN_PIN;
        // (never memory of the brk heap: its pages are trimmed and recycled by the allocator)
        const uintptr_t brk = (uintptr_t)sbrk(0);
        want_src = want_src && (uintptr_t)src >= brk;
        want_dst = want_dst && (uintptr_t)dst >= brk;
        if (want_src && want_dst && s0 < d1 && d0 < s1) {          // page ranges touch: one registration
            s0 = s0 < d0 ? s0 : d0; s1 = s1 > d1 ? s1 : d1;
            want_dst = false;
        } else if (s0 < d1 && d0 < s1) {
            want_src = want_dst = false;                            // a small buffer inside the other's pages
        }
        pin_src = want_src && hipHostRegister((void *)s0, s1 - s0, flags) == hipSuccess;
        pin_dst = want_dst && hipHostRegister((void *)d0, d1 - d0, flags) == hipSuccess;
        (void)hipGetLastError();
    }
    void unpin()
    {
        if (pin_src) (void)hipHostUnregister((void *)s0);
        if (pin_dst) (void)hipHostUnregister((void *)d0);
        pin_src = pin_dst = false;
    }
};

// Deal: the call works on the frames of blocks blk_index, blk_index + blk_count, ... (blocks of blk_frames frames) of
// the run only -- the share of one context of a pool (ntscsim_pool_frames_host); blk_count = 1: all of them.  Every
// block starts from its closed-form rand() position (the draws per field do not depend on the data), so the union of
// the shares is the sequential run.  pin = false: the caller has pinned the buffers already.
static int frames_host_impl(ntscsim_ctx *c, const ntscsim_host_source *S, const uint8_t *src, size_t src_frame_stride,
                            int src_ls, int n_frames, uint8_t *dst, size_t dst_frame_stride,
                            int dst_ls, int W, int H, uint64_t first_fieldno, uint32_t flags,
                            int chunk_frames, int blk_frames = 0, int blk_index = 0, int blk_count = 1,
                            bool pin = true)
{
    if (!c || !src || !dst || n_frames < 0) return NTSCSIM_E_ARG;
    if (n_frames == 0) return NTSCSIM_OK;
    if (S) {
        if (S->format < NTSCSIM_SRC_BGRA || S->format > NTSCSIM_SRC_YUV422P) return NTSCSIM_E_ARG;
        if (S->width < 1 || S->height < 1 || S->width > 16384 || S->height > 16384) return NTSCSIM_E_SIZE;
        if (S->frame_bytes == 0 || src_frame_stride < S->frame_bytes) return NTSCSIM_E_SIZE;
        const int np = S->format == NTSCSIM_SRC_BGRA ? 1 : 3;
        for (int p = 0; p < np; p++) {
            const int need = S->format == NTSCSIM_SRC_BGRA ? 4 * S->width : (p == 0 ? S->width : (S->width + 1) / 2);
            const size_t rows = (p == 0 || S->format != NTSCSIM_SRC_YUV420P) ? (size_t)S->height : ((size_t)S->height + 1) / 2;
            if (S->linesize[p] < need || S->plane_offset[p] + (size_t)S->linesize[p] * rows > S->frame_bytes) return NTSCSIM_E_SIZE;
        }
        src_ls = 4 * W;       // (the BGRA frames the field loop reads are made on the device)
    }
    const uint32_t yuv_bits = flags & (NTSCSIM_HOST_YUV420P | NTSCSIM_HOST_YUV422P);
    if (yuv_bits == (NTSCSIM_HOST_YUV420P | NTSCSIM_HOST_YUV422P)) return NTSCSIM_E_ARG;
    const bool yuv = yuv_bits != 0, v420 = yuv_bits == NTSCSIM_HOST_YUV420P;
    const uint32_t desc_flags = flags & 0xFFFu;
    if (src_ls < 4 * W) return NTSCSIM_E_SIZE;
    if (yuv ? (dst_ls < W || (dst_ls & 1) || (W & 1)) : dst_ls < 4 * W) return NTSCSIM_E_SIZE;
    if (W < 16 || H < 2) return NTSCSIM_E_SIZE;
    HIPCHK(c, hipSetDevice(c->device));
    if (chunk_frames <= 0) chunk_frames = 32;
    if (chunk_frames > n_frames) chunk_frames = n_frames;
    if (chunk_frames > 16384) chunk_frames = 16384;
    // device frames are packed at a 16-byte row pitch (the kernels' vector path); when the host
    // layout is the same, a whole chunk moves as ONE linear copy instead of per-frame 2-D copies
    const size_t pitch = (((size_t)W * 4 + 15) / 16) * 16;
    const size_t fbytes = pitch * H;
    // planar output frame (host and device): Y rows, then U rows, then V rows
    const size_t crows = v420 ? ((size_t)H + 1) / 2 : (size_t)H;
    const size_t ypitch_d = (((size_t)W + 15) / 16) * 16, cpitch_d = ypitch_d / 2;
    const size_t ybytes_d = ypitch_d * H, cbytes_d = cpitch_d * crows;
    const size_t obytes_d = yuv ? ybytes_d + 2 * cbytes_d : fbytes;       // multiple of 8
    const size_t ybytes_h = (size_t)dst_ls * H, cbytes_h = (size_t)(dst_ls / 2) * crows;
    const size_t obytes_h = yuv ? ybytes_h + 2 * cbytes_h : (size_t)dst_ls * H;
    if (dst_frame_stride < obytes_h) return NTSCSIM_E_SIZE;
    const size_t rawbytes = S ? ((S->frame_bytes + 15) / 16) * 16 : 0;    // device copy of one source frame
    const bool lin_src = S ? src_frame_stride == rawbytes : ((size_t)src_ls == pitch && src_frame_stride == fbytes);
    const bool lin_dst = yuv ? ((size_t)dst_ls == ypitch_d && dst_frame_stride == obytes_d)
                             : ((size_t)dst_ls == pitch && dst_frame_stride == fbytes);
    const size_t src_span = src_frame_stride * (size_t)(n_frames - 1) + (S ? S->frame_bytes : (size_t)src_ls * H);
    const size_t dst_span = dst_frame_stride * (size_t)(2 * n_frames - 1) + obytes_h;
    HostPins pins;
    if (pin) pins.pin(src, src_span, dst, dst_span, hipHostRegisterDefault);

    struct Slot { uint8_t *dsrc = nullptr, *ddst = nullptr, *dyuv = nullptr, *draw = nullptr; YuvDev *yrec = nullptr; ScaleDev *srec = nullptr;
                  hipEvent_t up = nullptr, done = nullptr, down = nullptr; bool used = false; };
    Slot slot[2];
    int rc = NTSCSIM_OK;
    auto fail = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == NTSCSIM_OK) { c->err = std::string(what) + ": " + hipGetErrorString(e); rc = NTSCSIM_E_HIP; }
        return e != hipSuccess;
    };
    if (!c->s_up) fail(hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking), "hipStreamCreate");
    if (!c->s_dn) fail(hipStreamCreateWithFlags(&c->s_dn, hipStreamNonBlocking), "hipStreamCreate");
    const hipStream_t s_up = c->s_up, s_dn = c->s_dn;
    std::vector<YuvDev> yrec_host;
    for (int i = 0; i < 2 && rc == NTSCSIM_OK; i++) {
        ntscsim_ctx::HostSlot &h = c->hslot[i];
        fail(h.dsrc.ensure(fbytes * chunk_frames), "hipMalloc");
        fail(h.ddst.ensure(fbytes * chunk_frames * 2), "hipMalloc");
        if (!h.up) fail(hipEventCreateWithFlags(&h.up, hipEventDisableTiming), "hipEventCreate");
        if (!h.done) fail(hipEventCreateWithFlags(&h.done, hipEventDisableTiming), "hipEventCreate");
        if (!h.down) fail(hipEventCreateWithFlags(&h.down, hipEventDisableTiming), "hipEventCreate");
        slot[i].dsrc = h.dsrc.p; slot[i].ddst = h.ddst.p;
        slot[i].up = h.up; slot[i].done = h.done; slot[i].down = h.down;
        if (S && rc == NTSCSIM_OK) {
            // raw source frames of this slot + their conversion records (fixed addresses: uploaded once)
            fail(h.draw.ensure(rawbytes * chunk_frames), "hipMalloc");
            fail(h.srec.ensure((size_t)chunk_frames), "hipMalloc");
            if (rc != NTSCSIM_OK) break;
            slot[i].draw = h.draw.p; slot[i].srec = h.srec.p;
            std::vector<ScaleDev> recs((size_t)chunk_frames);
            for (int k = 0; k < chunk_frames; k++) {
                ScaleDev &o = recs[(size_t)k];
                std::memset(&o, 0, sizeof(o));
                for (int p = 0; p < 3; p++) { o.src[p] = h.draw.p + rawbytes * (size_t)k + S->plane_offset[p]; o.src_ls[p] = S->linesize[p]; }
                o.dst = h.dsrc.p + fbytes * (size_t)k; o.dst_ls = (int32_t)pitch;
                o.sw = S->width; o.sh = S->height; o.fmt = S->format;
            }
            fail(upload_table(h.srec.p, recs.data(), recs.size() * sizeof(ScaleDev)), "hipMemcpy");
        }
        if (yuv && rc == NTSCSIM_OK) {
            // conversion records of this slot: fixed addresses, uploaded once per call
            fail(h.dyuv.ensure(obytes_d * chunk_frames * 2), "hipMalloc");
            fail(h.yrec.ensure((size_t)chunk_frames * 2), "hipMalloc");
            if (rc != NTSCSIM_OK) break;
            slot[i].dyuv = h.dyuv.p; slot[i].yrec = h.yrec.p;
            yrec_host.resize((size_t)chunk_frames * 2);
            for (int k = 0; k < 2 * chunk_frames; k++) {
                YuvDev &o = yrec_host[(size_t)k];
                o.bgra = h.ddst.p + fbytes * (size_t)k; o.bgra_ls = (int32_t)pitch;
                o.y = h.dyuv.p + obytes_d * (size_t)k; o.u = o.y + ybytes_d; o.v = o.u + cbytes_d;
                o.y_ls = (int32_t)ypitch_d; o.u_ls = o.v_ls = (int32_t)cpitch_d;
            }
            fail(upload_table(h.yrec.p, yrec_host.data(), yrec_host.size() * sizeof(YuvDev)), "hipMemcpy");
        }
    }
    DevParams Dyuv;
    std::memset(&Dyuv, 0, sizeof(Dyuv));
    Dyuv.W = W; Dyuv.H = H;
    std::vector<ntscsim_field_desc> descs((size_t)chunk_frames * 2);
    // the chunks of this call: all frames, or this context's blocks of a dealt run
    std::vector<std::pair<int, int>> chunks;        // (first frame, frames)
    const bool dealt = blk_count > 1 && blk_frames > 0;
    if (!dealt)
        for (int f0 = 0; f0 < n_frames; f0 += chunk_frames)
            chunks.push_back({f0, (n_frames - f0 < chunk_frames) ? n_frames - f0 : chunk_frames});
    else
        for (int b0 = blk_index * blk_frames; b0 < n_frames; b0 += blk_count * blk_frames) {
            const int bend = b0 + blk_frames < n_frames ? b0 + blk_frames : n_frames;
            for (int f0 = b0; f0 < bend; f0 += chunk_frames)
                chunks.push_back({f0, (bend - f0 < chunk_frames) ? bend - f0 : chunk_frames});
        }
    // rand() draws of one frame (two consecutive fields, whatever their order)
    const uint64_t frame_draws = ntscsim_rng_calls_per_field(&c->prm, W, H, 0) + ntscsim_rng_calls_per_field(&c->prm, W, H, 1);
    const uint64_t rng_base = c->rng_pos;
    int chunk_no = 0;
    for (size_t ci = 0; ci < chunks.size() && rc == NTSCSIM_OK; ci++, chunk_no++) {
        Slot &sl = slot[chunk_no & 1];
        const int f0 = chunks[ci].first, nf = chunks[ci].second;
        uint64_t cur = first_fieldno + 2ull * (uint64_t)f0;
        if (dealt) c->rng_pos = rng_base + (uint64_t)f0 * frame_draws;
        // the slot's previous download must have left the device buffers
        if (sl.used) { if (fail(hipEventSynchronize(sl.down), "hipEventSynchronize")) break; }
        // H2D: nf frames, row by row into the device pitch (or, for a scaled source, as they are)
        if (S && lin_src)      // (the last frame's rounding bytes are not the caller's: stop at frame_bytes)
            fail(hipMemcpyAsync(sl.draw, src + rawbytes * (size_t)f0, rawbytes * (size_t)(nf - 1) + S->frame_bytes,
                                hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D");
        else if (S)
            for (int j = 0; j < nf; j++)
                if (fail(hipMemcpyAsync(sl.draw + rawbytes * (size_t)j, src + src_frame_stride * (size_t)(f0 + j),
                                        S->frame_bytes, hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D")) break;
        if (S) { /* uploaded above */ }
        else if (lin_src)
            fail(hipMemcpyAsync(sl.dsrc, src + fbytes * (size_t)f0, fbytes * (size_t)nf,
                                hipMemcpyHostToDevice, s_up), "hipMemcpyAsync H2D");
        else
            for (int j = 0; j < nf; j++)
                if (fail(hipMemcpy2DAsync(sl.dsrc + fbytes * j, pitch, src + src_frame_stride * (size_t)(f0 + j),
                                          (size_t)src_ls, (size_t)W * 4, (size_t)H, hipMemcpyHostToDevice, s_up),
                         "hipMemcpy2DAsync H2D")) break;
        if (rc != NTSCSIM_OK) break;
        fail(hipEventRecord(sl.up, s_up), "hipEventRecord");
        // kernels on the ctx stream, after the upload
        fail(hipStreamWaitEvent(c->stream, sl.up, 0), "hipStreamWaitEvent");
        fail(hipMemsetAsync(sl.ddst, 0, fbytes * nf * 2, c->stream), "hipMemsetAsync");
        if (S) {
            hipLaunchKernelGGL(k_scale_to_bgra, dim3((unsigned)((W + 127) / 128), (unsigned)H, (unsigned)nf), dim3(128), 0,
                               c->stream, sl.srec, W, H);
            fail(hipGetLastError(), "k_scale_to_bgra");
        }
        for (int k = 0; k < 2 * nf; k++) {
            ntscsim_field_desc &d = descs[(size_t)k];
            std::memset(&d, 0, sizeof(d));
            d.src_dev = sl.dsrc + fbytes * (size_t)(k / 2);
            d.dst_dev = sl.ddst + fbytes * (size_t)k;
            d.src_linesize = (int)pitch; d.dst_linesize = (int)pitch;
            d.field = (uint32_t)((cur & 1) ^ 1);              // ffmpeg_ntsc.cpp:2229
            d.flags = desc_flags;
            d.fieldno = cur++;
            d.rng_pos = NTSCSIM_RNG_AUTO;
        }
        if (rc == NTSCSIM_OK) {
            const int r2 = ntscsim_fields_device(c, descs.data(), 2 * nf, W, H, c->stream);
            if (r2 != NTSCSIM_OK) { rc = r2; break; }
        }
        if (yuv) {
            // the encoder's pixel format, made on the GPU: the download shrinks 4 -> 1.5 / 2 B per pixel
            const unsigned rows = (unsigned)crows;
            hipLaunchKernelGGL(k_bgra_to_yuv, dim3((unsigned)((W / 8 + 1 + 127) / 128), rows, (unsigned)(2 * nf)),
                               dim3(128), 0, c->stream, Dyuv, sl.yrec, v420 ? 1 : 0, (W & 7) == 0 ? 1 : 0);
            fail(hipGetLastError(), "k_bgra_to_yuv");
        }
        fail(hipEventRecord(sl.done, c->stream), "hipEventRecord");
        // D2H on its own stream, after the kernels
        fail(hipStreamWaitEvent(s_dn, sl.done, 0), "hipStreamWaitEvent");
        const uint8_t *dout = yuv ? sl.dyuv : sl.ddst;
        if (lin_dst)
            fail(hipMemcpyAsync(dst + obytes_d * (size_t)(2 * f0), dout, obytes_d * (size_t)(2 * nf),
                                hipMemcpyDeviceToHost, s_dn), "hipMemcpyAsync D2H");
        else if (!yuv)
            for (int k = 0; k < 2 * nf && rc == NTSCSIM_OK; k++)
                fail(hipMemcpy2DAsync(dst + dst_frame_stride * (size_t)(2 * f0 + k), (size_t)dst_ls,
                                      sl.ddst + fbytes * (size_t)k, pitch, (size_t)W * 4, (size_t)H,
                                      hipMemcpyDeviceToHost, s_dn), "hipMemcpy2DAsync D2H");
        else
            for (int k = 0; k < 2 * nf && rc == NTSCSIM_OK; k++) {
                uint8_t *hf = dst + dst_frame_stride * (size_t)(2 * f0 + k);
                const uint8_t *df = sl.dyuv + obytes_d * (size_t)k;
                fail(hipMemcpy2DAsync(hf, (size_t)dst_ls, df, ypitch_d, (size_t)W, (size_t)H,
                                      hipMemcpyDeviceToHost, s_dn), "hipMemcpy2DAsync D2H");
                fail(hipMemcpy2DAsync(hf + ybytes_h, (size_t)dst_ls / 2, df + ybytes_d, cpitch_d,
                                      (size_t)W / 2, crows, hipMemcpyDeviceToHost, s_dn), "hipMemcpy2DAsync D2H");
                fail(hipMemcpy2DAsync(hf + ybytes_h + cbytes_h, (size_t)dst_ls / 2, df + ybytes_d + cbytes_d,
                                      cpitch_d, (size_t)W / 2, crows, hipMemcpyDeviceToHost, s_dn),
                     "hipMemcpy2DAsync D2H");
            }
        fail(hipEventRecord(sl.down, s_dn), "hipEventRecord");
        sl.used = true;
    }
    (void)hipStreamSynchronize(s_up);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(s_dn);
    if (dealt && rc == NTSCSIM_OK) c->rng_pos = rng_base + (uint64_t)n_frames * frame_draws;   // the whole run's end
    pins.unpin();
    return rc;
}

extern "C" int ntscsim_frames_host(ntscsim_ctx *c, const uint8_t *src, size_t src_frame_stride,
                                   int src_ls, int n_frames, uint8_t *dst, size_t dst_frame_stride,
                                   int dst_ls, int W, int H, uint64_t first_fieldno, uint32_t flags,
                                   int chunk_frames)
{
    return frames_host_impl(c, nullptr, src, src_frame_stride, src_ls, n_frames, dst, dst_frame_stride, dst_ls, W, H,
                            first_fieldno, flags, chunk_frames);
}

extern "C" int ntscsim_frames_host_scaled(ntscsim_ctx *c, const ntscsim_host_source *source, const uint8_t *src,
                                          size_t src_frame_stride, int n_frames, uint8_t *dst,
                                          size_t dst_frame_stride, int dst_ls, int W, int H,
                                          uint64_t first_fieldno, uint32_t flags, int chunk_frames)
{
    if (!source) return NTSCSIM_E_ARG;
    return frames_host_impl(c, source, src, src_frame_stride, 4 * W, n_frames, dst, dst_frame_stride, dst_ls, W, H,
                            first_fieldno, flags, chunk_frames);
}

extern "C" int ntscsim_debug_read_composite(ntscsim_ctx *c, int32_t *out, size_t out_elems)
{
    if (!c || !out) return NTSCSIM_E_ARG;
    if (c->last_n <= 0) return NTSCSIM_E_ARG;
    const size_t W = (size_t)c->last_W, Rpad = (size_t)c->last_Rpad;
    const size_t R = (size_t)c->last_n * c->last_Lslot;
    if (out_elems < R * W) return NTSCSIM_E_SIZE;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    std::vector<int32_t> tmp(W * Rpad);
    HIPCHK(c, hipMemcpy(tmp.data(), c->comp.p, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < R; r++)
        for (size_t x = 0; x < W; x++) out[r * W + x] = tmp[x * Rpad + r];
    return NTSCSIM_OK;
}

// ---- asynchronous host-frame drop-in: ntscsim_submit() / ntscsim_wait()
#include "ntscsim_submit.hip"
#include "ntscsim_host422.hip"

// ---- a pool of contexts over several GPUs: ntscsim_pool_*()
#include "ntscsim_pool.hip"
