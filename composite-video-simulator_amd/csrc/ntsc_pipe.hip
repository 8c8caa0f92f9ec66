// ntsc_pipe.hip -- the LATENCY form of the -vhs chain for short batches: one workgroup of three
// wavefronts per 63 scanlines (+ the halo row above), each wavefront one ROLE of the same rows, running side by side:
//
//     wave 0  encoder    BGRA -> composite samples            (the loop of encode_fast_body, ntsc_encode_fast.hip)
//     wave 1  VCR half   composite -> the VCR's composite out  (the loop of k_vcr_front, ntsc_decode_fast.hip)
//     wave 2  TV half    the VCR's output -> BGRA              (the loop of decode_fast_body<false>)
//
// A field is four lone wavefronts whichever way it is cut, and a lone wavefront issues one VALU instruction per ~5
// cycles: the one-launch chain walks a row's 235 instructions per pixel one after the other (0.43 ms of kernels per
// field), three roles walk 58 / 110 / 65 of them concurrently.  tools/role_probe.py measured the ceiling of this form
// BEFORE it was built (the three roles as independent launches: 386 us per call against 573 us); this kernel is that
// arrangement with the hand-offs in place.  It is only worth it while every wavefront has a SIMD to itself, so only the
// synchronous call and launches of at most NTSC_PIPE_MAX_FIELDS fields take it; long batches fill the chip with the
// one-wave-per-63-rows kernels, which issue less.
//
// The ARITHMETIC is not restated here: every sample goes through the same step functions as in the two-launch form
// (fastenc::step / edge_step, fastdec::vcr_step / vcr_edge, fastdec::step<false> / edge_step<false>), so the bits are
// the same by construction; this file is the three loops around them plus the flow control.
//
// Hand-offs.  The samples travel through the same transposed planes in global memory as in the two-launch form
// (comp[x][row]: encoder -> VCR half, comp_vcr[x][row]: VCR half -> TV half) -- the head switch needs random access in x
// and the luma path re-reads a sample 5 + d positions later -- and every role writes the columns its own lanes' rows read
// (a lane of the next workgroup's halo re-computes the same values: identical stores).  Flow control is two monotonic
// counters in LDS: `sync[0]` = composite columns the encoder has made visible, `sync[1]` = columns of the VCR's output.
// A producer publishes a column only when its stores have been acknowledged by the L2 (s_waitcnt vmcnt(N) with N = the
// memory operations it has issued since: one group behind, so it never stalls on its newest stores) behind a
// workgroup-scope release; a consumer waits (s_sleep poll, acquire) until the highest column it is about to request has
// been published, and requests composite samples with streaming (nt) loads, which bypass the CU's L1: what it reads is
// what the L2 holds.
#pragma clang fp contract(off)

#ifndef NTSC_PIPE_MAX_FIELDS
#define NTSC_PIPE_MAX_FIELDS 64
#endif

namespace ntscsim {
namespace pipe {

using namespace fastdec;

DEV int lds_peek(const volatile uint32_t *p) { return __builtin_amdgcn_readfirstlane((int)*p); }
DEV void wait_ge(const volatile uint32_t *p, int v, int &cached)
{
    if (cached >= v) return;
    int seen = lds_peek(p);
    while (seen < v) { __builtin_amdgcn_s_sleep(2); seen = lds_peek(p); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    cached = seen;
}
DEV void publish(volatile uint32_t *p, int v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *p = (uint32_t)v;
}
// s_waitcnt vmcnt(N), everything else untouched (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], expcnt imm[6:4], lgkmcnt imm[11:8])
#define NTSC_PIPE_VMCNT(N) __builtin_amdgcn_s_waitcnt((((N) >> 4) << 14) | 0x0F70 | ((N) & 15))

// what the three roles share about their lane's row
struct Row {
    int lane, rc, k;
    unsigned field, y;
    bool rowok, is_out;
    const FieldDev *fd;
};

// ------------------------------------------------------------------------------------------------ wave 0: encoder
template <class RT>
DEV void encoder_role(const DevParams &P, const Row &R, const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                      int *__restrict__ comp, uint32_t *ring, uint32_t *ltile, volatile uint32_t *sync)
{
    using namespace fastenc;
    const int lane = R.lane, W = P.W;
    const FieldDev &fd = *R.fd;
    const unsigned opposite = (fd.flags & 1u) ? ((fd.flags & 2u) ? 1u : 0u) : 0u;      // :1585-1588, :1599
    unsigned sy = R.y + opposite;
    if (sy > (unsigned)P.H - 1u) sy = (unsigned)P.H - 1u;
    const uint8_t *srow = fd.src + (size_t)fd.src_ls * sy;

    EConst<RT> C;
    C.xi = scan_phase(P, R.y, fd.fieldno);
    C.W = W;
    C.lane = lane;
    C.mL = opaque_v((C.xi & 2u) ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    C.odd = false;
#pragma unroll
    for (int j = 0; j < 4; j++) C.ms[j] = 0;
    C.a_i = (RT)P.a_in_i; C.a_q = (RT)P.a_in_q;
    C.a_pre = (RT)P.a_pre; C.pre_gain = (RT)P.pre_gain;
    C.rowbytes = P.Rpad * 4;
    C.vcol = R.rc * 4;                 // the column of this lane's row (a halo / padding lane re-computes a neighbour's: same values)
    C.comp = __builtin_amdgcn_make_buffer_rsrc(comp, 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    EState<RT> S;
    S.rng.init(ring, rs_luma + R.rc, P.Rpad, lane, 1);
    S.noise = n0_luma[R.rc];
    S.lpI.reset(0, C.a_i); S.lpQ.reset(0, C.a_q);
    S.pre.p = 16;
#pragma unroll
    for (int q = 0; q < 4; q++) { S.Yd[q] = 0; S.Ir[q] = 0; S.Qr[q] = 0; }
    S.fI[0] = S.fI[1] = 0;

    int t = 0;
    for (; t < 4; t++) edge_step<RT, false, 0>(P, S, C, ring, reinterpret_cast<const uint32_t *>(srow), t);
    if (t + 16 <= W) {
        CoopLoader L;
        L.begin(srow, ltile, lane);
        uint32_t cur[16];
        v4u nq[4];
        L.request(t, nq);
        L.deliver(nq, cur);
        int Y0 = S.Yd[0], Y1 = S.Yd[1], Y2 = S.Yd[2], Y3 = S.Yd[3];
        int I0 = S.fI[0], I1 = S.fI[1];
        RT IdT[4], QdT[4];
        int sbase = S.rng.pos;
        for (; t + 16 <= W; t += 16) {
            const bool more = t + 32 <= W;
            uint32_t *const rb = ring + sbase * 64 + lane;
            const bool rb0 = sbase == 0;
            sbase = (sbase + 16) & 31;
            if (more) L.request(t + 16, nq);
            unsigned soff = (unsigned)(t - 4) * (unsigned)C.rowbytes;
            int Yn[16], F[16];
#define NTSC_PIPE_ENC_STEP(J, YX, IX)                                                             \
            {                                                                                     \
                RT dY, Id_, Qd_;                                                                  \
                rgb_to_yiq256<RT>(cur[J], dY, Id_, Qd_);                                          \
                Yn[J] = (int)dY;                                                                  \
                if (J >= 12) { IdT[J & 3] = Id_; QdT[J & 3] = Qd_; }                              \
                const int Y = step<J, RT, false, false>(P, S, C, rb, rb0, Id_, Qd_, YX, IX, F[J]); \
                __builtin_amdgcn_raw_buffer_store_b32(Y, C.comp, C.vcol, (int)soff, 0);           \
                soff += (unsigned)C.rowbytes;                                                     \
            }
            NTSC_PIPE_ENC_STEP(0, Y0, I0)
            NTSC_PIPE_ENC_STEP(1, Y1, I1)
            NTSC_PIPE_ENC_STEP(2, Y2, F[0])
            NTSC_PIPE_ENC_STEP(3, Y3, F[1])
            NTSC_PIPE_ENC_STEP(4, Yn[0], F[2])
            NTSC_PIPE_ENC_STEP(5, Yn[1], F[3])
            NTSC_PIPE_ENC_STEP(6, Yn[2], F[4])
            NTSC_PIPE_ENC_STEP(7, Yn[3], F[5])
            NTSC_PIPE_ENC_STEP(8, Yn[4], F[6])
            NTSC_PIPE_ENC_STEP(9, Yn[5], F[7])
            NTSC_PIPE_ENC_STEP(10, Yn[6], F[8])
            NTSC_PIPE_ENC_STEP(11, Yn[7], F[9])
            NTSC_PIPE_ENC_STEP(12, Yn[8], F[10])
            NTSC_PIPE_ENC_STEP(13, Yn[9], F[11])
            NTSC_PIPE_ENC_STEP(14, Yn[10], F[12])
            NTSC_PIPE_ENC_STEP(15, Yn[11], F[13])
#undef NTSC_PIPE_ENC_STEP
            Y0 = Yn[12]; Y1 = Yn[13]; Y2 = Yn[14]; Y3 = Yn[15];
            I0 = F[14]; I1 = F[15];
            // the chunk before this one is in the L2 once at most THIS chunk's memory operations are still in flight: its 16
            // stores, and the 4 row requests when there is a next chunk (vmcnt counts in issue order)
            if (more) NTSC_PIPE_VMCNT(20); else NTSC_PIPE_VMCNT(16);
            publish(sync, t - 4);                  // columns < t - 4: everything the previous chunks stored
            if (more) L.deliver(nq, cur);
        }
        S.rng.pos = sbase;
        S.Yd[0] = Y0; S.Yd[1] = Y1; S.Yd[2] = Y2; S.Yd[3] = Y3;
        S.fI[0] = I0; S.fI[1] = I1;
#pragma unroll
        for (int q = 0; q < 4; q++) { S.Ir[q] = (int)IdT[q]; S.Qr[q] = (int)QdT[q]; }
    }
    for (; t < W + 4; t++) edge_step<RT, false, 0>(P, S, C, ring, reinterpret_cast<const uint32_t *>(srow), t);
    NTSC_PIPE_VMCNT(0);
    publish(sync, W);
}

// ------------------------------------------------------------------------------------------------ wave 1: VCR half
// columns of the composite plane a step at stream position t may request: its own loads reach t + 7 (two iterations of
// look-ahead minus one) and a head-switched lane reads up to W/10 + 1 columns further (k_field_setup; the launcher takes
// this form only for displacements within W/10)
template <class RT>
DEV void vcr_role(const DevParams &P, const GeomDev &G, const Row &R, const int *__restrict__ comp, int *__restrict__ comp_out,
                  const uint32_t *__restrict__ rs_chroma, const int *__restrict__ n0_u, const int *__restrict__ n0_v,
                  const int *__restrict__ hs_shift, const int *__restrict__ pn_noise, int *__restrict__ tails, uint32_t *ring,
                  volatile uint32_t *sync)
{
    const int lane = R.lane, W = P.W;
    const FieldDev &fd = *R.fd;
    typedef Const<RT, false> CT;
    CT C;
    C.wrapoff = 0; C.wrapA = 0x3FFFFFFF; C.wrapS = 0;
    C.bmul = 0; C.bshift = 0; C.odd = false; C.mo = 0;
    C.xi = scan_phase(P, R.y, fd.fieldno);
    C.hi = (C.xi & 2u) != 0;
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = lane;
    C.d = P.cdelay;
    C.SKT = 7 + C.d;
    C.LOFF = 5 + C.d;
    C.mL = opaque_v(C.hi ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    const bool vb = P.vblend && P.ntsc;
    C.bA = opaque_v((vb && R.k >= 2) ? -1 : 0);
    C.bC = opaque_v((vb && R.k >= 1) ? 1 : 0);
    C.dm = -1;
    {
        int n = (R.rowok ? pn_noise[R.rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (RT)G.ptab[2 * n]; C.sinv = (RT)G.ptab[2 * n + 1];
    }
    C.a_vc = (RT)P.a_vc; C.a_vl = (RT)P.a_vl; C.a_sh = (RT)P.a_sh; C.a_tv = (RT)P.a_tv;
    C.sharp2 = (RT)(P.sharpen * 2);
    C.tailU = tails + (size_t)blockIdx.x * 64 + lane;
    C.rstride = (size_t)gridDim.x * 64;
    C.rowbytes = P.Rpad * 4;
    const int hs = P.hs ? hs_shift[R.rc] : 0;
    C.vbase = (int)((unsigned)R.rc * 4u + (unsigned)hs * (unsigned)C.rowbytes);
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp), 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(comp_out, 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);
    // The column this lane's TV role reads -- for the lanes that own a row.  The halo lane and the lanes past the last row
    // compute a NEIGHBOUR's row with another row above it (the vertical blend), i.e. other values: like k_vcr_front they
    // write to padding columns (R <= column < Rpad), and their TV lanes, whose pixels nobody stores, read whatever the
    // row's owner wrote.
    const int gidx = (int)blockIdx.x * 63 + lane - 1;
    const int vout = (lane >= 1 && gidx < P.R ? gidx : P.R + lane) * 4;

    State<true, RT> S;
    S.D1.init(); S.D2.init();
    S.l0 = S.l1 = S.l2 = S.lsum = 0;
    S.vl.reset(16, C.a_vl); S.vpre.reset(16, C.a_vl); S.vcU.reset(0, C.a_vc); S.vcV.reset(0, C.a_vc);
    S.sh.reset(0, C.a_sh); S.oU.reset(0, C.a_tv); S.oV.reset(0, C.a_tv);
    S.Yprev = S.Uraw = S.Vraw = 0;
    S.Uf2[0] = S.Uf2[1] = 0;
    S.rng.init(ring, rs_chroma + R.rc, P.Rpad, lane, (-(31 + 2 * (C.SKT - 7))) & 7);
    S.nU = n0_u[R.rc]; S.nV = n0_v[R.rc];

    const int SK1 = C.SKT, LOFF = C.LOFF;
    const int total = W + SK1;
    const unsigned rb = (unsigned)C.rowbytes;
    const int reach = W / 10 + 2;          // head-switch displacement of the farthest lane, + 1
    int enc_seen = 0;
    int t = 0;
    int yv_, uv_, vv_;
    // a guarded step at t loads column t (+ displacement) and column t - LOFF
    auto need_edge = [&](int tt) { const int c = tt + reach; wait_ge(sync, c < W ? c : W, enc_seen); };
    for (; t < SK1 && t < total; t++) { need_edge(t); (void)vcr_edge<RT, CT>(P, S, C, ring, t, yv_, uv_, vv_); }
    {
        const int t_end = W - (C.d > 7 ? C.d - 7 : 0);
        if (t + 4 <= t_end && !(S.rng.pos & 7)) {
            Steady T;
            T.D1.from(S.D1, (C.d & 1) != 0);
            T.D2.from(S.D2, true);
            T.lc1 = S.l2; T.lpA = S.l2 + S.l1; T.lpB = S.l1 + S.l0;
            int sbase = S.rng.pos;
            int pc[4], pl[4];
            { const int c = t + 4 + reach; wait_ge(sync, c < W ? c : W, enc_seen); }
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j] = cs_load<2>(C, t + j); pl[j] = cs_load<2>(C, t + j - LOFF); }
            unsigned soff = (unsigned)(t - SK1) * rb;
#define NTSC_PIPE_VCR_STEP(DPV, J)                                                                \
            {                                                                                     \
                const int c2 = vcr_step<DPV, J, RT, CT>(P, S, T, C, pc[J], pl[J], yv_, uv_, vv_); \
                pc[J] = cs_load<2>(C, t + 4 + J);                                                 \
                pl[J] = cs_load<2>(C, t + 4 + J - LOFF);                                          \
                __builtin_amdgcn_raw_buffer_store_b32(c2, out, vout, (int)soff, 0);               \
                soff += rb;                                                                       \
                NTSC_STEP_SCHED_BARRIER();                                                        \
            }
#define NTSC_PIPE_VCR_ITER(DPV)                                                                   \
            for (; t + 4 <= t_end; t += 4) {                                                      \
                { const int c = t + 8 + reach; wait_ge(sync, c < W ? c : W, enc_seen); }          \
                T.rb = ring + sbase * 64 + lane; T.rb0 = sbase == 0; sbase = (sbase + 8) & 31;    \
                NTSC_PIPE_VCR_STEP(DPV, 0) NTSC_PIPE_VCR_STEP(DPV, 1) NTSC_PIPE_VCR_STEP(DPV, 2) NTSC_PIPE_VCR_STEP(DPV, 3) \
                /* the iteration before this one is in the L2 once at most this one's 8 loads + 4 stores are in flight */ \
                NTSC_PIPE_VMCNT(12);                                                              \
                publish(sync + 1, t - SK1);                                                       \
            }
            switch (C.d & 3) {
                case 0: NTSC_PIPE_VCR_ITER(0) break;
                case 1: NTSC_PIPE_VCR_ITER(1) break;
                case 2: NTSC_PIPE_VCR_ITER(2) break;
                default: NTSC_PIPE_VCR_ITER(3) break;
            }
#undef NTSC_PIPE_VCR_ITER
#undef NTSC_PIPE_VCR_STEP
            T.D1.to(S.D1, (C.d & 1) != 0);
            S.l2 = T.lc1; S.l1 = T.lpA - T.lc1; S.l0 = T.lpB - S.l1; S.lsum = S.l0 + S.l1 + S.l2;
            S.rng.pos = sbase;
        }
    }
    for (; t < total; t++) {
        need_edge(t);
        const int c2 = vcr_edge<RT, CT>(P, S, C, ring, t, yv_, uv_, vv_);
        const int x2 = t - SK1;
        if (x2 >= 0) __builtin_amdgcn_raw_buffer_store_b32(c2, out, vout, (int)((unsigned)x2 * rb), 0);
    }
    NTSC_PIPE_VMCNT(0);
    publish(sync + 1, W);
}

// ------------------------------------------------------------------------------------------------ wave 2: TV half
template <class RT>
DEV void tv_role(const DevParams &P, const Row &R, const int *__restrict__ comp_vcr, const int *__restrict__ dropout,
                 uint32_t *ostage, const unsigned long long *orow, uint32_t *drow, volatile uint32_t *sync)
{
    const int lane = R.lane, W = P.W;
    const FieldDev &fd = *R.fd;
    typedef Const<RT, false> CT;
    CT C;
    C.wrapoff = 0; C.wrapA = 0x3FFFFFFF; C.wrapS = 0;
    C.bmul = P.m_amp_back.mul; C.bshift = P.m_amp_back.shift; C.odd = false; C.mo = 0;
    C.xi = scan_phase(P, R.y, fd.fieldno);
    C.hi = (C.xi & 2u) != 0;
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = lane;
    C.d = 0;
    C.SKT = 8;
    C.LOFF = 5;
    C.mL = opaque_v(C.hi ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    C.bA = 0; C.bC = 0;
    C.dm = opaque_v((P.loss && dropout[R.rc] != 0) ? 0 : -1);
    C.cosv = 1; C.sinv = 0;
    C.a_vc = (RT)P.a_vc; C.a_vl = (RT)P.a_vl; C.a_sh = (RT)P.a_sh; C.a_tv = (RT)P.a_tv;
    C.a_oi = (RT)P.a_in_i; C.a_oq = (RT)P.a_in_q;
    C.sharp2 = (RT)(P.sharpen * 2);
    C.tailU = nullptr; C.rstride = 0;
    C.xs = nullptr;
    C.rowbytes = P.Rpad * 4;
    C.vbase = (int)((unsigned)R.rc * 4u);                  // the VCR's output carries the head switch already
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp_vcr), 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    State<false, RT> S;
    S.D1.init(); S.D2.init();
    S.l0 = S.l1 = S.l2 = S.lsum = 0;
    S.vl.reset(16, C.a_vl); S.vpre.reset(16, C.a_vl); S.vcU.reset(0, C.a_vc); S.vcV.reset(0, C.a_vc);
    S.sh.reset(0, C.a_sh);
    S.oU.reset(0, C.a_tv); S.oV.reset(0, C.a_tv);
    S.Yprev = S.Uraw = S.Vraw = 0;
    S.Uf2[0] = S.Uf2[1] = 0;
    S.nU = S.nV = 0;

    const int SKT = C.SKT, total = W + SKT;
    int vcr_seen = 0;
    int t = 0;
    auto need = [&](int cols) { wait_ge(sync + 1, cols < W ? cols : W, vcr_seen); };
    // (the guarded steps of the one-separator decoder load with cs_load's plain policy; the columns they ask for were
    //  never in this CU's L1 before -- each is requested exactly once, after it was published)
    for (; t < SKT && t < total; t++) {
        uint32_t px; int xo;
        need(t + 1);
        (void)edge_step<false, RT, CT>(P, S, C, nullptr, t, px, xo);
    }
    // steady: 4 positions per iteration, the next iteration's samples requested at the top of the current one
    {
        const int t_end = W;
        if (t + 4 <= t_end) {
            Steady T;
            T.D1.from(S.D1, true);
            T.D2.from(S.D2, true);
            T.lc1 = S.l2; T.lpA = S.l2 + S.l1; T.lpB = S.l1 + S.l0;
            T.D1.ieP &= C.dm; T.D1.qeP &= C.dm; T.D1.ieN &= C.dm; T.D1.qeN &= C.dm;
            int pc[4];
            need(t + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) pc[j] = cs_load<2>(C, t + j);
            int pend_x = -1;
            for (; t + 4 <= t_end; t += 4) {
                uint32_t o[4];
                int nc[4] = {0, 0, 0, 0};
                need(t + 8);
#pragma unroll
                for (int j = 0; j < 4; j++) nc[j] = cs_load<2>(C, t + 4 + j);      // (past the row end: the buffer's bounds check, 0)
                o[0] = step<false, 0, 0, RT, CT>(P, S, T, C, pc[0], 0);
                o[1] = step<false, 0, 1, RT, CT>(P, S, T, C, pc[1], 0);
                o[2] = step<false, 0, 2, RT, CT>(P, S, T, C, pc[2], 0);
                o[3] = step<false, 0, 3, RT, CT>(P, S, T, C, pc[3], 0);
#pragma unroll
                for (int j = 0; j < 4; j++) pc[j] = nc[j];
                const int xo0 = t - SKT;
                const int sub = (xo0 >> 2) & 3;
                *reinterpret_cast<uint4 *>(&ostage[lane * 20 + sub * 4]) = make_uint4(o[0], o[1], o[2], o[3]);
                if (sub == 3) {
                    pend_x = xo0 - 12;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int r = 16 * k + (lane >> 2);
                        const unsigned long long rp = orow[r];
                        const uint4 v = *reinterpret_cast<const uint4 *>(&ostage[r * 20 + (lane & 3) * 4]);
                        if (rp) NTSC_OUT_STORE((g_v4u_ptr)(rp + 4ull * (unsigned)(pend_x + (lane & 3) * 4)), to_v4u(v));
                    }
                }
            }
            T.D1.to(S.D1, true);
            S.Uraw = 0; S.Vraw = 0;
        }
    }
    for (; t < total; t++) {
        uint32_t px; int xo;
        need(t + 1);
        if (!edge_step<false, RT, CT>(P, S, C, nullptr, t, px, xo)) continue;
        ostage[lane * 20 + (xo & 15)] = px;
        if ((xo & 15) == 15) {
            if (R.is_out) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(&ostage[lane * 20]);
                g_v4u_ptr dp = (g_v4u_ptr)(drow + (xo - 15));
                const uint4 a = sp[0], b = sp[1], c4 = sp[2], d4 = sp[3];
                dp[0] = to_v4u(a); dp[1] = to_v4u(b); dp[2] = to_v4u(c4); dp[3] = to_v4u(d4);
            }
        } else if (xo == W - 1 && R.is_out) {
            const int xb = xo & ~15;
            for (int q = xb; q <= xo; q++) ((g_u32_ptr)drow)[q] = ostage[lane * 20 + (q - xb)];
        }
    }
}

} // namespace pipe

// One workgroup = 63 rows + the halo row above, three wavefronts = three roles (see the head of this file).
// Preconditions (launcher): the -vhs preset family of the hand-tuned kernels (input chroma low-pass on, no pre-emphasis, luma /
// chroma / phase noise on, amplitudes 50 / 50, even scanline phase, output low-pass "lite", composite out), head-switch
// displacement within W/10, 16-byte aligned rows, planes below 4 GiB, no ghosting.
template <class RT>
__global__ __launch_bounds__(192) void k_field_pipe(DevParams P, GeomDev G, const FieldDev *__restrict__ fields,
                                                    const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                                                    int *__restrict__ comp, int *__restrict__ comp_vcr,
                                                    const uint32_t *__restrict__ rs_chroma, const int *__restrict__ n0_u,
                                                    const int *__restrict__ n0_v, const int *__restrict__ hs_shift,
                                                    const int *__restrict__ pn_noise, const int *__restrict__ dropout,
                                                    int *__restrict__ tails)
{
    using namespace pipe;
    __shared__ uint32_t ring_e[33 * 64];                                   // the encoder's rand() ring
    __shared__ __attribute__((aligned(16))) uint32_t ltile[64 * 20];       // its cooperative row loads
    __shared__ uint32_t ring_v[33 * 64];                                   // the VCR half's rand() ring
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];      // the TV half's pixel staging
    __shared__ unsigned long long orow[64];
    __shared__ uint32_t sync[2];

    const int role = threadIdx.x >> 6;
    Row R;
    R.lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 63 + R.lane - 1;          // lane 0 = halo (row above)
    R.rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = R.rc / P.Lslot;
    R.k = R.rc - f * P.Lslot;
    R.fd = &fields[f];
    R.field = R.fd->field & 1u;
    R.rowok = (int)(R.field + 2u * R.k) < P.H;
    R.is_out = R.lane >= 1 && gidx < P.R && R.rowok;
    R.y = R.rowok ? R.field + 2u * (unsigned)R.k : R.field;
    uint32_t *drow = reinterpret_cast<uint32_t *>(R.fd->dst + (size_t)R.fd->dst_ls * R.y);
    if (role == 2) orow[R.lane] = R.is_out ? (unsigned long long)drow : 0ull;
    if (threadIdx.x < 2) sync[threadIdx.x] = 0u;
    __syncthreads();
    if (role == 0) encoder_role<RT>(P, R, rs_luma, n0_luma, comp, ring_e, ltile, sync);
    else if (role == 1) vcr_role<RT>(P, G, R, comp, comp_vcr, rs_chroma, n0_u, n0_v, hs_shift, pn_noise, tails, ring_v, sync);
    else tv_role<RT>(P, R, comp_vcr, dropout, ostage, orow, drow, sync);
}

} // namespace ntscsim
