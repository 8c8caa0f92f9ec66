// ntsc_pipe.hip -- the LATENCY form of the -vhs chain for short batches: one workgroup of FIVE
// wavefronts per 63 scanlines (+ the halo row above), each wavefront one ROLE of the same rows, running side by side:
//
//     ENC   encoder              BGRA -> composite samples               (the loop of encode_fast_body, ntsc_encode_fast.hip)
//     SEP   VCR, chroma front    composite -> raw chroma + chroma noise  (first separator, :1497-1567, :1719-1735)
//     CHR   VCR, chroma back     phase noise, VHS chroma low-pass, vertical blend, carrier sign (:1736-1762, :1814-1863)
//     LUM   VCR luma + TV front  VHS luma low-pass / emphasis / sharpen, re-modulation, second separator
//                                (:1793-1812, :1866-1888, :1497-1567, dropout :1891-1901)
//     OUT   TV back              TV chroma low-pass, YIQ -> RGB, pixel stores  (:1399-1427, :1385-1396)
//
// A field is four lone wavefronts whichever way it is cut, and a lone wavefront gets one instruction of any kind through
// per ~5.3 cycles: the one-launch chain walks a row's ~235 VALU instructions per pixel one after the other (encoder 97 us,
// VCR half 209 us, TV half 107 us as lone kernels on one field); here five wavefronts walk 63 / 58 / 52 / 70 / 65
// instructions per pixel concurrently on the four SIMDs of one CU (SEP and CHR share one): 146 us, the roles' clocks in
// profiles/r06_sync_pipe.txt, DESIGN.md 1c.
// It is only worth it while every workgroup has a CU to itself, so only the synchronous call and launches of at most
// NTSC_PIPE_MAX_FIELDS fields from the host-frame entry points take it; long batches fill the chip with the
// one-wave-per-63-rows kernels, which issue less in total.
//
// The ARITHMETIC of a position is the one of the one-launch kernels, cut at three places where only integers cross:
// every role runs the same stream positions t = 0 .. W + 15 + d as k_decode_fast<true> (fill: guarded steps, steady:
// 4 positions per iteration, drain: guarded steps) with ITS part of fastdec::vcr_step / step / vcr_edge / edge_step,
// written out below next to the statement of the original it is (same operations on the same operands: same bits;
// tests/test_gpu_parity.py::test_synchronous_call_takes_the_pipelined_form_and_equals_the_oracle).
//
// Hand-offs.
//   ENC -> SEP, LUM: the composite samples travel through the transposed plane in global memory (comp[x][row]) -- the
//       head switch needs random access in x.  ENC publishes a column count (flag F_ENC) once its stores have been
//       acknowledged by the L2 (s_waitcnt vmcnt(N), N = the memory operations issued since: one chunk behind, so it never
//       stalls on its newest stores); the readers request samples with streaming (nt) loads, which always miss the
//       CU's L1: what they read is what the L2 holds (two workgroups of one launch may share a CU, and a 128-byte line
//       of a column holds rows of two workgroups).
//   SEP -> CHR -> LUM -> OUT: rings of RING stream positions in LDS (2 / 1 / 3 integers per lane and position), each
//       with a `produced` and a `consumed` position count.  A producer writes a position only when the slot's previous
//       tenant has been consumed and publishes it behind a workgroup-scope release (s_waitcnt lgkmcnt(0)), in the
//       steady loop one step late, when the stores have long landed; a consumer polls (s_sleep) only when the count it
//       remembers is not enough.
#pragma clang fp contract(off)

#ifndef NTSC_PIPE_MAX_FIELDS
#define NTSC_PIPE_MAX_FIELDS 64
#endif
// wavefront w of the workgroup takes role (NTSC_PIPE_ORDER >> 4w) & 15: 0 ENC, 1 SEP, 2 CHR, 3 LUM, 4 OUT.  Wavefronts 0 and 4
// land on the same SIMD (round-robin placement): the two lightest roles.
#ifndef NTSC_PIPE_ORDER
#define NTSC_PIPE_ORDER 0x24301u
#endif

namespace ntscsim {
namespace pipe {

using namespace fastdec;

constexpr int RING = 16;          // stream positions per LDS ring (four steady iterations)
enum { F_ENC = 0, F_AB_P, F_AB_C, F_BC_P, F_BC_C, F_CD_P, F_CD_C, F_COUNT = 8 };

typedef __attribute__((address_space(3))) volatile uint32_t *lds_flag;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) uint32_t *lds_x1;
typedef __attribute__((address_space(3))) u32x2 *lds_x2;
typedef __attribute__((address_space(3))) u32x4 *lds_x4;

__shared__ unsigned g_waited[8];
// A hand-off that never comes (a count two roles disagree about -- a bug, not a data condition) must not hang the GPU: a poll
// gives up after NTSC_PIPE_SPIN_LIMIT rounds (~0.3 s, five orders of magnitude above any real wait), raises g_fault in LDS --
// every later wait of the workgroup then falls through at once -- and the kernel reports it in a pinned word the host
// checks behind the launch (NTSCSIM_E_HIP "k_field_pipe: hand-off timed out"; pixels of that launch are garbage).
#ifndef NTSC_PIPE_SPIN_LIMIT
#define NTSC_PIPE_SPIN_LIMIT (1 << 22)
#endif
__shared__ unsigned g_fault;
DEV int flag_peek(lds_flag p) { return __builtin_amdgcn_readfirstlane((int)*p); }
// wait until the count at p is at least v (wave-uniform); `cached`: the last value this wavefront saw
DEV void wait_ge(lds_flag p, int v, int &cached)
{
    if (cached >= v) return;
    int seen = flag_peek(p);
    if (seen < v) {
        // (developer timing, NTSCSIM_PIPE_TIMING: 100 MHz ticks this wavefront spent polling, summed per wavefront behind the flags)
        const unsigned t0 = (unsigned)wall_clock64();
        int spins = 0;
        do {
            __builtin_amdgcn_s_sleep(1);
            seen = flag_peek(p);
            if (++spins > NTSC_PIPE_SPIN_LIMIT || (spins > 64 && *(lds_flag)&g_fault)) { *(lds_flag)&g_fault = 1u; seen = 0x3FFFFFFF; }
        } while (seen < v);
        g_waited[threadIdx.x >> 6] += (unsigned)wall_clock64() - t0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    cached = seen;
}
DEV void publish(lds_flag p, int v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *p = (uint32_t)v;
}
// s_waitcnt vmcnt(N), everything else untouched (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], expcnt imm[6:4], lgkmcnt imm[11:8])
#define NTSC_PIPE_VMCNT(N) __builtin_amdgcn_s_waitcnt((((N) >> 4) << 14) | 0x0F70 | ((N) & 15))

// what the roles share about their lane's row
struct Row {
    int lane, rc, k;
    unsigned field, y;
    bool rowok, is_out;
    const FieldDev *fd;
};

// slot of stream position t in a ring: the steady loop's first position (t = SKT) sits on slot 0, so the four
// positions of an iteration are four consecutive slots
DEV int slot_of(int t, int SKT) { return (t - SKT) & (RING - 1); }

// ------------------------------------------------------------------------------------------------ ENC: encoder
template <class RT, bool XA = false, bool PRE = false>
DEV void encoder_role(const DevParams &P, const Row &R, const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                      int *__restrict__ comp, uint32_t *ring, uint32_t *ltile, uint32_t *etail, lds_flag fl)
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
    C.odd = XA && (C.xi & 1u) != 0;
#pragma unroll
    for (int j = 0; j < 4; j++) C.ms[j] = XA ? opaque_v(((C.xi + (unsigned)j) & 2u) ? -1 : 0) : 0;
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

    // The source frame may be the caller's own, in pinned HOST memory (ntscsim_field() on a frame the GPU can address: no
    // upload): a load then takes a round trip over the link, so everything the row's guarded steps need -- its first four
    // pixels, the up to 15 behind the last whole chunk -- and the first chunk are requested HERE, together; the pixels of the
    // row's end wait in LDS (etail: [16][64]).
    const fastdec::g_cu32_ptr gpx = (fastdec::g_cu32_ptr) reinterpret_cast<const uint32_t *>(srow);
    const int t_e = 4 + 16 * ((W - 4) / 16);          // where the chunks end (W >= 16)
    uint32_t hpx[4], tpx[16];
#pragma unroll
    for (int j = 0; j < 4; j++) hpx[j] = gpx[j];
#pragma unroll
    for (int j = 0; j < 16; j++) tpx[j] = t_e + j < W ? gpx[t_e + j] : 0u;
    CoopLoader L;
    v4u nq[4];
    const bool chunks = 4 + 16 <= W;
    if (chunks) {
        L.begin(srow, ltile, lane);
        L.request(4, nq);
    }
    int t = 0;
#pragma unroll
    for (int j = 0; j < 4; j++, t++) edge_step_px<RT, PRE, 0>(P, S, C, ring, hpx[j], j);
#pragma unroll
    for (int j = 0; j < 16; j++) etail[j * 64 + lane] = tpx[j];
    if (chunks) {
        uint32_t cur[16];
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
                const int Y = step<J, RT, PRE, XA>(P, S, C, rb, rb0, Id_, Qd_, YX, IX, F[J]);      \
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
            // the chunk before this one is in the L2 once at most this chunk's memory operations are still in flight (its
            // 8 stores so far, and the 4 row requests when there is a next chunk; vmcnt counts in issue order): published
            // in the middle of the chunk, half a chunk earlier than at its end
            if (more) NTSC_PIPE_VMCNT(12); else NTSC_PIPE_VMCNT(8);
            publish(fl + F_ENC, t - 4);            // columns < t - 4: everything the previous chunks stored
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
            if (more) L.deliver(nq, cur);
        }
        S.rng.pos = sbase;
        S.Yd[0] = Y0; S.Yd[1] = Y1; S.Yd[2] = Y2; S.Yd[3] = Y3;
        S.fI[0] = I0; S.fI[1] = I1;
#pragma unroll
        for (int q = 0; q < 4; q++) { S.Ir[q] = (int)IdT[q]; S.Qr[q] = (int)QdT[q]; }
        NTSC_PIPE_VMCNT(0);
        publish(fl + F_ENC, t - 4);
    }
    for (; t < W + 4; t++) edge_step_px<RT, PRE, 0>(P, S, C, ring, t < W ? etail[(t - t_e) * 64 + lane] : 0u, t);
    NTSC_PIPE_VMCNT(0);
    publish(fl + F_ENC, W);
}

// ------------------------------------------------------------------------------------------------ the decoder roles
// per-lane constants of the four decoder roles (every role fills what it reads; the rest is dead code)
template <class RT, bool WR = false, bool XA = false, bool BK = false>
DEV void dec_const(Const<RT, WR, BK, false, XA> &C, const DevParams &P, const Row &R, const int *comp, const int *hs_shift, bool sv = false)
{
    const int W = P.W;
    C.wrapoff = 0; C.wrapA = 0x3FFFFFFF; C.wrapS = 0;
    C.bmul = BK ? P.m_amp_back.mul : 0; C.bshift = BK ? P.m_amp_back.shift : 0; C.odd = false; C.mo = 0;
    C.xi = scan_phase(P, R.y, R.fd->fieldno);
    C.hi = (C.xi & 2u) != 0;
    if constexpr (XA) {        // scanline phases of either parity (-comp-phase 90 / 270, odd offsets): per-lane picks and signs
        C.odd = (C.xi & 1u) != 0;
        C.mo = opaque_v(C.odd ? -1 : 0);
#pragma unroll
        for (int j = 0; j < 4; j++) C.ms[j] = opaque_v(((C.xi + (unsigned)j) & 2u) ? -1 : 0);
    }
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = R.lane;
    C.d = P.cdelay;
    C.SKT = (sv ? 8 : 15) + C.d;       // the depth of the whole decoder: every role runs k_decode_fast<true>'s positions (S-Video out
                                       // of the VCR, k_decode_fast_sv: no re-modulation / second separation, 7 stages fewer)
    C.LOFF = 5 + C.d;
    C.mL = opaque_v(C.hi ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    const bool vb = P.vblend && P.ntsc;
    C.bA = opaque_v((vb && R.k >= 2) ? -1 : 0);
    C.bC = opaque_v((vb && R.k >= 1) ? 1 : 0);
    C.dm = -1;
    C.cosv = 1; C.sinv = 0;
    C.a_vc = (RT)P.a_vc; C.a_vl = (RT)P.a_vl; C.a_sh = (RT)P.a_sh; C.a_tv = (RT)P.a_tv;
    C.a_oi = (RT)P.a_in_i; C.a_oq = (RT)P.a_in_q;
    C.sharp2 = (RT)(P.sharpen * 2);
    C.tailU = nullptr; C.rstride = 0; C.xs = nullptr;
    C.rowbytes = P.Rpad * 4;
    const int hs = (P.hs && hs_shift) ? hs_shift[R.rc] : 0;
    C.vbase = (int)((unsigned)R.rc * 4u + (unsigned)hs * (unsigned)C.rowbytes);
    if constexpr (WR) {       // displacements that wrap around the 1.1 W window (decode_fast_body, cs_load)
        const int tw = W + W / 10;
        C.wrapoff = (int)((unsigned)(hs > 0 ? -tw : tw) * (unsigned)C.rowbytes);
        C.wrapA = hs > 0 ? tw - hs - 1 : (hs < 0 ? -hs - 1 : 0x3FFFFFFF);
        C.wrapS = opaque_v(hs < 0 ? -1 : 0);
    }
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp), 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);
}
// the steady loop exists iff one iteration fits (every role decides alike: the positions are the same for all)
DEV bool has_steady(int W, int d, int SKT) { return SKT + 4 <= W - (d > 7 ? d - 7 : 0); }
DEV bool has_steady(int W, int d) { return has_steady(W, d, 15 + d); }
// How far ahead of its own position a reader of the composite plane may ask: the head-switch displacement of its farthest
// lane, + 2.  Displacements within W/10 (the launcher's head_switch_is_small): W/10 + 2 covers them.  WR (any displacement,
// e.g. PAL's default switching point: the sample is row[(x + shift) mod 1.1 W]): the workgroup's largest forward
// displacement -- and the WHOLE row when a lane's backward displacement exceeds W/10, because the row's first samples
// then come from its end (that workgroup, the one that holds the switched rows, runs its encoder first and the rest behind).
template <bool WR>
DEV int wg_reach(const DevParams &P, const int *hs_shift, int rc)
{
    const int W = P.W;
    if (!WR || !P.hs) return W / 10 + 2;
    int hs = hs_shift[rc], mx = hs, mn = hs;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int a = __shfl_xor(mx, o), b = __shfl_xor(mn, o);
        mx = a > mx ? a : mx; mn = b < mn ? b : mn;
    }
    mx = __builtin_amdgcn_readfirstlane(mx); mn = __builtin_amdgcn_readfirstlane(mn);
    if (mn < -(W / 10)) return W;
    return (mx > 0 ? mx : 0) + 2;
}

// ------------------------------------------------------------------------------------------------ SEP: VCR, chroma front
// vcr_step / vcr_edge up to the chroma noise: first separator at x1 = t - 7 (no luma out), U / V += noise, two draws
// BK: the separated chroma is scaled by 50 / subcarrier_amplitude_back (the pre-emphasis presets raise it: k_decode_fast_bk)
template <class RT, bool WR, bool SV = false, bool XA = false, bool BK = false>
DEV void sep_role(const DevParams &P, const Row &R, const int *__restrict__ comp, const uint32_t *__restrict__ rs_chroma,
                  const int *__restrict__ n0_u, const int *__restrict__ n0_v, const int *__restrict__ hs_shift,
                  uint32_t *ring, lds_x2 ab, lds_flag fl)
{
    typedef Const<RT, WR, BK, false, XA> CT;
    CT C;
    dec_const<RT, WR, XA, BK>(C, P, R, comp, hs_shift, SV);
    const int lane = R.lane, W = P.W, SKT = C.SKT, total = W + SKT;
    DemodR D1;
    D1.init();
    LaneRand32 rng;
    rng.init(ring, rs_chroma + R.rc, P.Rpad, lane, (-(31 + 2 * (SKT - 7))) & 7);      // (as decode_fast_body)
    int nU = n0_u[R.rc], nV = n0_v[R.rc];
    // columns of the composite plane a step at stream position t may request: its own column t and, on a head-switched
    // lane, up to W/10 + 1 columns further (k_field_setup; the launcher takes this form only for displacements within W/10)
    const int reach = wg_reach<WR>(P, hs_shift, R.rc);
    int enc_seen = 0, cons_seen = 0;
    auto need_enc = [&](int c) { wait_ge(fl + F_ENC, c < W ? c : W, enc_seen); };
    auto edge = [&](int t) {
        need_enc(t + reach);
        wait_ge(fl + F_AB_C, t + 1 - RING, cons_seen);
        const int pc = t < W ? cs_load<2>(C, t) : 0;
        int Y, U, V;
        D1.template push_edge<BK, XA>(pc, t, C.xi, C.hi, W, C.xe, Y, U, V, C.bmul, C.bshift);
        const int x1 = t - 7;
        if (x1 >= 0 && x1 < W) {
            U += nU; V += nV;                                                       // chroma noise :1719-1735
            nU = sdiv2(nU + (int)umod31(rng.next(ring, lane), P.m_cnoise) - P.cnoise_k);
            nV = sdiv2(nV + (int)umod31(rng.next(ring, lane), P.m_cnoise) - P.cnoise_k);
        }
        ab[slot_of(t, SKT) * 64 + lane] = u32x2{(uint32_t)U, (uint32_t)V};
        publish(fl + F_AB_P, t + 1);
    };
    int t = 0;
    for (; t < SKT && t < total; t++) edge(t);
    const int t_end = W - (C.d > 7 ? C.d - 7 : 0);
    if (has_steady(W, C.d, SKT) && !(rng.pos & 7)) {
        DemodS S1;
        S1.from(D1, ((SKT - 7) & 1) != 0);        // x1 = t - 7 = SKT - 7 (mod 4) at the loop's first position
        int sbase = rng.pos;
        int pc[4];
        need_enc(t + 4 + reach);
#pragma unroll
        for (int j = 0; j < 4; j++) pc[j] = cs_load<2>(C, t + j);
#define NTSC_PIPE_SEP_STEP(DPH, J, PRE)                                                           \
        {                                                                                         \
            constexpr bool pick1 = (((DPH) + (J)) & 1) != 0, neg1 = (((DPH) + (J)) & 3) == 3;     \
            int Yd, U, V;                                                                         \
            S1.template push<pick1, neg1, false, BK, false, XA>(pc[J], C.hi, -1, Yd, U, V, C.bmul, C.bshift, C.odd, C.mo); \
            U += nU; V += nV;                                                                     \
            nU = sdiv2(nU + (int)umod31(rng.template draw<2 * (J)>(rb, rb0), P.m_cnoise) - P.cnoise_k);     \
            nV = sdiv2(nV + (int)umod31(rng.template draw<2 * (J) + 1>(rb, rb0), P.m_cnoise) - P.cnoise_k); \
            PRE;                                                                                  \
            o[(J) * 64] = u32x2{(uint32_t)U, (uint32_t)V};                                        \
            NTSC_STEP_SCHED_BARRIER();                                                            \
        }
#define NTSC_PIPE_SEP_ITER(DPH)                                                                   \
        for (; t + 4 <= t_end; t += 4) {                                                          \
            need_enc(t + 8 + reach);                                                              \
            wait_ge(fl + F_AB_C, t + 4 - RING, cons_seen);                                        \
            uint32_t *const rb = ring + sbase * 64 + lane;                                        \
            const bool rb0 = sbase == 0;                                                          \
            sbase = (sbase + 8) & 31;                                                             \
            const lds_x2 o = ab + slot_of(t, SKT) * 64 + lane;                                    \
            /* the next iteration's samples are requested here and taken over at the bottom: a whole iteration of       \
               arithmetic (~0.5 us) covers the L2 round trip (past the row end: the buffer's bounds check, 0) */         \
            int nc[4];                                                                            \
            _Pragma("unroll") for (int j = 0; j < 4; j++) nc[j] = cs_load<2>(C, t + 4 + j);       \
            /* the iteration before this one is published between the arithmetic of step 0 and its store: the release's    \
               s_waitcnt lgkmcnt(0) then finds the earlier stores long landed and nothing new in flight */                 \
            NTSC_PIPE_SEP_STEP(DPH, 0, publish(fl + F_AB_P, t))                                   \
            NTSC_PIPE_SEP_STEP(DPH, 1, (void)0) NTSC_PIPE_SEP_STEP(DPH, 2, (void)0) NTSC_PIPE_SEP_STEP(DPH, 3, (void)0) \
            _Pragma("unroll") for (int j = 0; j < 4; j++) pc[j] = nc[j];                          \
        }
        switch ((SKT - 7) & 3) {
            case 0: NTSC_PIPE_SEP_ITER(0) break;
            case 1: NTSC_PIPE_SEP_ITER(1) break;
            case 2: NTSC_PIPE_SEP_ITER(2) break;
            default: NTSC_PIPE_SEP_ITER(3) break;
        }
#undef NTSC_PIPE_SEP_ITER
#undef NTSC_PIPE_SEP_STEP
        publish(fl + F_AB_P, t);
        S1.to(D1, ((SKT - 7) & 1) != 0);
        rng.pos = sbase;
    }
    for (; t < total; t++) edge(t);
}

// ------------------------------------------------------------------------------------------------ CHR: VCR, chroma back
// vcr_step / vcr_edge from the phase noise to the sign of the re-modulated chroma: what comes out is the term the
// composite sample at x2 = t - 7 - d adds to its luma
// SV (S-Video out of the VCR): no carrier -- both blended components go on, as they are (vcr_step's Uv, Vv)
// XA: the carrier's U / V role swaps on lanes with an odd scanline phase, its sign is per lane (vcr_step, CT::anyxi)
template <class RT, bool SV = false, bool XA = false>
DEV void chroma_role(const DevParams &P, const GeomDev &G, const Row &R, const int *__restrict__ pn_noise,
                     int *__restrict__ tails, lds_x2 ab, lds_x2 bc, lds_flag fl)
{
    typedef Const<RT, false, false, false, XA> CT;
    CT C;
    dec_const<RT, false, XA>(C, P, R, nullptr, nullptr, SV);
    const int lane = R.lane, W = P.W, SKT = C.SKT, total = W + SKT;
    {
        int n = (R.rowok ? pn_noise[R.rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (RT)G.ptab[2 * n]; C.sinv = (RT)G.ptab[2 * n + 1];
    }
    int *const tailU = tails + (size_t)blockIdx.x * 64 + lane;
    const size_t rstride = (size_t)gridDim.x * 64;
    Casc3<RT> vcU, vcV;
    vcU.reset(0, C.a_vc); vcV.reset(0, C.a_vc);
    int in_seen = 0, cons_seen = 0;
    auto edge = [&](int t) {
        wait_ge(fl + F_AB_P, t + 1, in_seen);
        wait_ge(fl + F_BC_C, t + 1 - RING, cons_seen);
        const u32x2 uv = ab[slot_of(t, SKT) * 64 + lane];
        const int x1 = t - 7;
        int fU = 0, fV = 0;
        if (x1 >= 0 && x1 < W) {
            const RT u = (RT)(int)uv.x, v = (RT)(int)uv.y;
            const int U = (int)((u * C.cosv) - (v * C.sinv));                       // chroma phase noise :1748-1762
            const int V = (int)((u * C.sinv) + (v * C.cosv));
            fU = (int)vcU.push((RT)U, C.a_vc);                                      // VHS chroma low-pass :1814-1836
            fV = (int)vcV.push((RT)V, C.a_vc);
            if (x1 >= W - C.d) {                  // raw tail of the chroma low-pass :1830
                tailU[(size_t)(x1 & 15) * rstride] = U;
                tailU[(size_t)(16 + (x1 & 15)) * rstride] = V;
            }
        }
        const int x2 = x1 - C.d;
        const bool in2 = x2 >= 0 && x2 < W;
        if (in2 && x2 >= W - C.d) {
            fU = tailU[(size_t)(x2 & 15) * rstride];
            fV = tailU[(size_t)(16 + (x2 & 15)) * rstride];
        }
        const int U = ((wave_up(fU) & C.bA) + fU + C.bC) >> C.bC;                   // vertical blend :1843-1863
        const int V = ((wave_up(fV) & C.bA) + fV + C.bC) >> C.bC;
        int ch = 0;
        if (in2) {
            const unsigned s = (C.xi + (unsigned)x2) & 3u;
            ch = (s & 1u) ? V : U;
            if (s & 2u) ch = -ch;
        }
        bc[slot_of(t, SKT) * 64 + lane] = SV ? u32x2{(uint32_t)(in2 ? U : 0), (uint32_t)(in2 ? V : 0)} : u32x2{(uint32_t)ch, 0u};
        publish(fl + F_BC_P, t + 1);
        *(fl + F_AB_C) = (uint32_t)(t + 1);
    };
    int t = 0;
    for (; t < SKT && t < total; t++) edge(t);
    const int t_end = W - (C.d > 7 ? C.d - 7 : 0);
    if (has_steady(W, C.d, SKT)) {
#define NTSC_PIPE_CHR_STEP(J, PRE)                                                                \
        {                                                                                         \
            const RT u = (RT)(int)in[J].x, v = (RT)(int)in[J].y;                                  \
            const RT Ud = rtrunc<RT>((u * C.cosv) - (v * C.sinv));                                \
            const RT Vd = rtrunc<RT>((u * C.sinv) + (v * C.cosv));                                \
            const int fU = (int)vcU.push(Ud, C.a_vc);                                             \
            const int fV = (int)vcV.push(Vd, C.a_vc);                                             \
            /* x2 = J (mod 4): U for even J, sign by J & 2 (vcr_step); only the component that is modulated is blended */ \
            u32x2 outv;                                                                           \
            if constexpr (SV) {                                                                   \
                outv = u32x2{(uint32_t)(((wave_up(fU) & C.bA) + fU + C.bC) >> C.bC),              \
                             (uint32_t)(((wave_up(fV) & C.bA) + fV + C.bC) >> C.bC)};             \
            } else {                                                                              \
                int chroma;                                                                       \
                if constexpr (XA) {       /* (rows of one wavefront differ in parity: blend both, then pick per lane) */ \
                    const int bU = ((wave_up(fU) & C.bA) + fU + C.bC) >> C.bC;                    \
                    const int bV = ((wave_up(fV) & C.bA) + fV + C.bC) >> C.bC;                    \
                    chroma = ((J) & 1) ? (C.odd ? bU : bV) : (C.odd ? bV : bU);                   \
                } else {                                                                          \
                    const int f = ((J) & 1) ? fV : fU;                                            \
                    chroma = ((wave_up(f) & C.bA) + f + C.bC) >> C.bC;                            \
                }                                                                                 \
                const int mm = XA ? C.ms[(J) & 3] : (((J) & 2) ? C.mNL : C.mL);                   \
                outv = u32x2{(uint32_t)((chroma ^ mm) - mm), 0u};                                 \
            }                                                                                     \
            PRE;                                                                                  \
            o[(J) * 64] = outv;                                                                   \
            NTSC_STEP_SCHED_BARRIER();                                                            \
        }
        for (; t + 4 <= t_end; t += 4) {
            wait_ge(fl + F_AB_P, t + 4, in_seen);
            wait_ge(fl + F_BC_C, t + 4 - RING, cons_seen);
            const lds_x2 ip = ab + slot_of(t, SKT) * 64 + lane;
            const lds_x2 o = bc + slot_of(t, SKT) * 64 + lane;
            u32x2 in[4];
#pragma unroll
            for (int j = 0; j < 4; j++) in[j] = ip[j * 64];
            // (the iteration before this one: written, and its inputs read -- published as in SEP)
            NTSC_PIPE_CHR_STEP(0, (publish(fl + F_BC_P, t), (void)(*(fl + F_AB_C) = (uint32_t)t)))
            NTSC_PIPE_CHR_STEP(1, (void)0) NTSC_PIPE_CHR_STEP(2, (void)0) NTSC_PIPE_CHR_STEP(3, (void)0)
        }
#undef NTSC_PIPE_CHR_STEP
        publish(fl + F_BC_P, t);
        *(fl + F_AB_C) = (uint32_t)t;
    }
    for (; t < total; t++) edge(t);
}

// ------------------------------------------------------------------------------------------------ LUM: VCR luma + TV front
// the luma path of vcr_step / vcr_edge (box at x2, VHS low-pass + emphasis, sharpen), the VCR's composite sample
// c2 = Y + chroma term, and the TV's separator on it (step<true> / edge_step<true>: x3 = x2 - 7, dropout as the and-mask)
// SV: the VCR's luma and the blended chroma go to the TV's output stage as they are (x3 = x2: step<true> / edge_step<true>
// with CT::svideo), dropout as the and-mask
template <class RT, bool WR, bool SV = false, bool XA = false>
DEV void luma_role(const DevParams &P, const Row &R, const int *__restrict__ comp, const int *__restrict__ hs_shift,
                   const int *__restrict__ dropout, lds_x2 bc, lds_x4 cd, lds_flag fl)
{
    typedef Const<RT, WR, false, false, XA> CT;
    CT C;
    dec_const<RT, WR, XA>(C, P, R, comp, hs_shift, SV);
    C.dm = opaque_v((P.loss && dropout[R.rc] != 0) ? 0 : -1);
    const int lane = R.lane, W = P.W, SKT = C.SKT, total = W + SKT, LOFF = C.LOFF;
    DemodR D2;
    D2.init();
    int l0 = 0, l1 = 0, l2 = 0;
    Casc3<RT> vl, sh;
    PoleHp<RT> vpre;
    vl.reset(16, C.a_vl); vpre.reset(16, C.a_vl); sh.reset(0, C.a_sh);
    const int reach = wg_reach<WR>(P, hs_shift, R.rc);
    int enc_seen = 0, in_seen = 0, cons_seen = 0;
    auto need_enc = [&](int c) { wait_ge(fl + F_ENC, c < W ? c : W, enc_seen); };
    auto edge = [&](int t) {
        wait_ge(fl + F_BC_P, t + 1, in_seen);
        wait_ge(fl + F_CD_C, t + 1 - RING, cons_seen);
        const int xl = t - LOFF;
        if (xl >= 0) need_enc(xl + reach);
        const int pl = (xl >= 0 && xl < W) ? cs_load<2>(C, xl) : 0;
        const u32x2 chv = bc[slot_of(t, SKT) * 64 + lane];
        const int ch = (int)chv.x;
        const int yb = sdiv4(l0 + l1 + l2 + pl);
        l0 = l1; l1 = l2; l2 = pl;
        const int x2 = t - 7 - C.d;
        int c2 = 0, Ysv = 0;
        if (x2 >= 0 && x2 < W) {
            RT m2;
            RT s = vl.push((RT)yb, C.a_vl, m2);                                     // :1793-1812
            s += vpre.hp(s, m2, C.a_vl) * RT(1.6);
            const RT s0 = rtrunc<RT>(s);
            const RT ts = sh.push(s0, C.a_sh);                                      // :1866-1883
            const int Y = (int)(s0 + ((s0 - ts) * C.sharp2));
            c2 = Y + ch;                                                            // :1885-1888
            Ysv = Y;
        }
        int Y, U, V;
        if constexpr (SV) {
            // (vcr_edge hands Yv / Uv / Vv on, zero outside the row; the chroma role has zeroed its part)
            Y = Ysv; U = (int)chv.x; V = (int)chv.y;
            if (x2 >= W) { U = 0; V = 0; Y = 0; }
        } else {
        D2.template push_edge<false, XA>(c2, x2, C.xi, C.hi, W, C.xe, Y, U, V);
        if (x2 - 7 >= W) { U = 0; V = 0; Y = 0; }
        }
        U &= C.dm; V &= C.dm;                                                       // :1891-1901
        cd[slot_of(t, SKT) * 64 + lane] = u32x4{(uint32_t)Y, (uint32_t)U, (uint32_t)V, 0u};
        publish(fl + F_CD_P, t + 1);
        *(fl + F_BC_C) = (uint32_t)(t + 1);
    };
    int t = 0;
    for (; t < SKT && t < total; t++) edge(t);
    const int t_end = W - (C.d > 7 ? C.d - 7 : 0);
    if (has_steady(W, C.d, SKT)) {
        DemodS S2;
        S2.from(D2, true);                        // x3 = 1 (mod 4) at the loop's first position: a pick
        S2.ieP &= C.dm; S2.qeP &= C.dm; S2.ieN &= C.dm; S2.qeN &= C.dm;     // (as steady(): the guarded steps mask their outputs)
        int lc1 = l2, lpA = l2 + l1, lpB = l1 + l0;
        int pl[4];
        need_enc(t + 4 - LOFF + reach);
#pragma unroll
        for (int j = 0; j < 4; j++) pl[j] = cs_load<2>(C, t + j - LOFF);
#define NTSC_PIPE_LUM_STEP(J, PRE)                                                                \
        {                                                                                         \
            const int lp = pl[J] + lc1;                                                           \
            const int yb = sdiv4s(lp + lpB);                                                      \
            lc1 = pl[J]; lpB = lpA; lpA = lp;                                                     \
            RT m2;                                                                                \
            RT s = vl.push((RT)yb, C.a_vl, m2);                                                   \
            s += vpre.hp(s, m2, C.a_vl) * RT(1.6);                                                \
            const RT s0 = rtrunc<RT>(s);                                                          \
            const RT ts = sh.push(s0, C.a_sh);                                                    \
            const int Yl = (int)(s0 + ((s0 - ts) * C.sharp2));                                    \
            int Y, U, V;                                                                          \
            if constexpr (SV) { Y = Yl; U = (int)in[J].x & C.dm; V = (int)in[J].y & C.dm; }       \
            else {                                                                                \
                const int c2 = Yl + (int)in[J].x;                                                 \
                constexpr bool pick3 = (((J) + 1) & 1) != 0, neg3 = (((J) + 1) & 3) == 3;         \
                S2.template push<pick3, neg3, true, false, true, XA>(c2, C.hi, C.dm, Y, U, V, 0, 0, C.odd, C.mo); \
            }                                                                                     \
            PRE;                                                                                  \
            o[(J) * 64] = u32x4{(uint32_t)Y, (uint32_t)U, (uint32_t)V, 0u};                       \
            NTSC_STEP_SCHED_BARRIER();                                                            \
        }
        for (; t + 4 <= t_end; t += 4) {
            need_enc(t + 8 - LOFF + reach);
            wait_ge(fl + F_BC_P, t + 4, in_seen);
            wait_ge(fl + F_CD_C, t + 4 - RING, cons_seen);
            const lds_x2 ip = bc + slot_of(t, SKT) * 64 + lane;
            const lds_x4 o = cd + slot_of(t, SKT) * 64 + lane;
            u32x2 in[4];
            int nl[4];                                // (the next iteration's samples: as in SEP)
#pragma unroll
            for (int j = 0; j < 4; j++) nl[j] = cs_load<2>(C, t + 4 + j - LOFF);
#pragma unroll
            for (int j = 0; j < 4; j++) in[j] = ip[j * 64];
            NTSC_PIPE_LUM_STEP(0, (publish(fl + F_CD_P, t), (void)(*(fl + F_BC_C) = (uint32_t)t)))
            NTSC_PIPE_LUM_STEP(1, (void)0) NTSC_PIPE_LUM_STEP(2, (void)0) NTSC_PIPE_LUM_STEP(3, (void)0)
#pragma unroll
            for (int j = 0; j < 4; j++) pl[j] = nl[j];
        }
#undef NTSC_PIPE_LUM_STEP
        publish(fl + F_CD_P, t);
        *(fl + F_BC_C) = (uint32_t)t;
        S2.to(D2, true);
        l2 = lc1; l1 = lpA - lc1; l0 = lpB - l1;
    }
    for (; t < total; t++) edge(t);
}

// ------------------------------------------------------------------------------------------------ TVF: TV front, default preset
// the decoder of the default preset (no VCR) up to its separator: step<false> / edge_step<false> of k_decode_fast<false>
// without the output stage -- composite sample (head switching, if switched on without the VCR) -> Y, U, V at x3 = t - 7
// CATV (the pre-emphasis presets without the VCR: k_encode_fast_pre in front): the separated chroma is scaled by
// 50 / subcarrier_amplitude_back, then rotated by the row's chroma phase noise (:1736-1764; the presets switch it on) --
// a rotation by cos = 1, sin = 0 when the noise is off: (int)((u * 1.0) - (v * 0.0)) = u
template <class RT, bool CATV = false>
DEV void tvfront_role(const DevParams &P, const Row &R, const int *__restrict__ comp, const int *__restrict__ hs_shift,
                      const int *__restrict__ dropout, const int *__restrict__ pn_noise, const double *__restrict__ ptab,
                      lds_x4 cd, lds_flag fl)
{
    typedef Const<RT, false, CATV> CT;
    CT C;
    dec_const<RT, false, false, CATV>(C, P, R, comp, hs_shift);
    C.d = 0; C.SKT = 8;
    C.dm = opaque_v((P.loss && dropout[R.rc] != 0) ? 0 : -1);
    if (CATV && P.pnoise_k) {
        int n = (R.rowok ? pn_noise[R.rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (RT)ptab[2 * n]; C.sinv = (RT)ptab[2 * n + 1];
    }
    const int lane = R.lane, W = P.W, SKT = 8, total = W + SKT;
    DemodR D1;
    D1.init();
    const int reach = P.hs ? W / 10 + 2 : 1;
    int enc_seen = 0, cons_seen = 0;
    auto need_enc = [&](int c) { wait_ge(fl + F_ENC, c < W ? c : W, enc_seen); };
    auto edge = [&](int t) {
        need_enc(t + reach);
        wait_ge(fl + F_CD_C, t + 1 - RING, cons_seen);
        const int pc = t < W ? cs_load<2>(C, t) : 0;
        int Y, U, V;
        D1.template push_edge<CATV, false>(pc, t, C.xi, C.hi, W, C.xe, Y, U, V, C.bmul, C.bshift);
        if (t - 7 >= W) { U = 0; V = 0; Y = 0; }
        if constexpr (CATV) {                                                       // chroma phase noise :1748-1762
            const RT u = (RT)U, v = (RT)V;
            U = (int)((u * C.cosv) - (v * C.sinv));
            V = (int)((u * C.sinv) + (v * C.cosv));
        }
        U &= C.dm; V &= C.dm;                                                       // :1891-1901
        cd[slot_of(t, SKT) * 64 + lane] = u32x4{(uint32_t)Y, (uint32_t)U, (uint32_t)V, 0u};
        publish(fl + F_CD_P, t + 1);
    };
    int t = 0;
    for (; t < SKT && t < total; t++) edge(t);
    if (has_steady(W, 0, SKT)) {
        DemodS S1;
        S1.from(D1, true);                        // x3 = 1 (mod 4) at the loop's first position: a pick
        S1.ieP &= C.dm; S1.qeP &= C.dm; S1.ieN &= C.dm; S1.qeN &= C.dm;
        int pc[4];
        need_enc(t + 4 + reach);
#pragma unroll
        for (int j = 0; j < 4; j++) pc[j] = cs_load<2>(C, t + j);
#define NTSC_PIPE_TVF_STEP(J, PRE)                                                                \
        {                                                                                         \
            constexpr bool pick3 = (((J) + 1) & 1) != 0, neg3 = (((J) + 1) & 3) == 3;             \
            int Y, U, V;                                                                          \
            S1.template push<pick3, neg3, true, CATV, true, false>(pc[J], C.hi, C.dm, Y, U, V, C.bmul, C.bshift); \
            if constexpr (CATV) {                                                                 \
                const RT u = (RT)U, v = (RT)V;                                                    \
                U = (int)((u * C.cosv) - (v * C.sinv));                                           \
                V = (int)((u * C.sinv) + (v * C.cosv));                                           \
            }                                                                                     \
            PRE;                                                                                  \
            o[(J) * 64] = u32x4{(uint32_t)Y, (uint32_t)U, (uint32_t)V, 0u};                       \
            NTSC_STEP_SCHED_BARRIER();                                                            \
        }
        for (; t + 4 <= W; t += 4) {
            need_enc(t + 8 + reach);
            wait_ge(fl + F_CD_C, t + 4 - RING, cons_seen);
            const lds_x4 o = cd + slot_of(t, SKT) * 64 + lane;
            int nc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) nc[j] = cs_load<2>(C, t + 4 + j);      // (past the row end: the buffer's bounds check, 0)
            NTSC_PIPE_TVF_STEP(0, publish(fl + F_CD_P, t))
            NTSC_PIPE_TVF_STEP(1, (void)0) NTSC_PIPE_TVF_STEP(2, (void)0) NTSC_PIPE_TVF_STEP(3, (void)0)
#pragma unroll
            for (int j = 0; j < 4; j++) pc[j] = nc[j];
        }
#undef NTSC_PIPE_TVF_STEP
        publish(fl + F_CD_P, t);
        S1.to(D1, true);
    }
    for (; t < total; t++) edge(t);
}

// ------------------------------------------------------------------------------------------------ OUT: TV back
// the output stage of step<true> / edge_step<true>: composite_lowpass_tv (delay 1), YIQ -> RGB for the previous position,
// 16 pixels of 64 rows staged in LDS and stored as 64-byte bursts (steady()'s cooperative flush)
template <class RT, bool VHS = true, bool SV = false>
DEV void output_role(const DevParams &P, const Row &R, uint32_t *ostage, const unsigned long long *orow, uint32_t *drow,
                     lds_x4 cd, lds_flag fl)
{
    // (the positions of the decoder in front: k_decode_fast<true> -- SKT = 15 + d -- or, default preset, k_decode_fast<false>: 8)
    const int lane = R.lane, W = P.W, d = VHS ? P.cdelay : 0, SKT = VHS ? (SV ? 8 : 15) + d : 8, total = W + SKT;
    const RT a_tv = (RT)P.a_tv;
    Casc3<RT> oU, oV;
    oU.reset(0, a_tv); oV.reset(0, a_tv);
    int Yprev = 0, Uraw = 0, Vraw = 0;
    int in_seen = 0;
    auto edge = [&](int t) {
        wait_ge(fl + F_CD_P, t + 1, in_seen);
        const u32x4 yuv = cd[slot_of(t, SKT) * 64 + lane];
        publish(fl + F_CD_C, t + 1);              // (behind the release: the slot has been read)
        const int x3 = t - (SKT - 1);
        if (x3 < 0 || x3 > W) return;
        const int Y = (int)yuv.x, U = (int)yuv.y, V = (int)yuv.z;
        RT fUd = 0, fVd = 0;
        if (x3 < W) {
            fUd = rtrunc<RT>(oU.push((RT)U, a_tv));
            fVd = rtrunc<RT>(oV.push((RT)V, a_tv));
        }
        const int xo = x3 - 1;
        const int Yo = Yprev, Ur = Uraw, Vr = Vraw;
        Yprev = Y; Uraw = U; Vraw = V;
        if (xo < 0) return;
        if (xo >= W - 1) { fUd = (RT)Ur; fVd = (RT)Vr; }      // last sample keeps its input :1419-1424
        const uint32_t px = yiq_to_bgra<RT>(Yo, fUd, fVd);
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
    };
    int t = 0;
    for (; t < SKT && t < total; t++) edge(t);
    const int t_end = W - (d > 7 ? d - 7 : 0);
    if (has_steady(W, d, SKT)) {
        for (; t + 4 <= t_end; t += 4) {
            wait_ge(fl + F_CD_P, t + 4, in_seen);
            const lds_x4 ip = cd + slot_of(t, SKT) * 64 + lane;
            u32x4 in[4];
#pragma unroll
            for (int j = 0; j < 4; j++) in[j] = ip[j * 64];
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const RT fUd = rtrunc<RT>(oU.push((RT)(int)in[j].y, a_tv));
                const RT fVd = rtrunc<RT>(oV.push((RT)(int)in[j].z, a_tv));
                o[j] = yiq_to_bgra<RT>(Yprev, fUd, fVd);
                Yprev = (int)in[j].x;
                if (j == 0) publish(fl + F_CD_C, t + 4);      // the four slots are in registers
                NTSC_STEP_SCHED_BARRIER();
            }
            const int xo0 = t - SKT;                   // multiple of 4
            const int sub = (xo0 >> 2) & 3;
            *reinterpret_cast<uint4 *>(&ostage[lane * 20 + sub * 4]) = make_uint4(o[0], o[1], o[2], o[3]);
            if (sub == 3) {
                const int pend_x = xo0 - 12;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = 16 * k + (lane >> 2);
                    const unsigned long long rp = orow[r];
                    const uint4 v = *reinterpret_cast<const uint4 *>(&ostage[r * 20 + (lane & 3) * 4]);
                    if (rp) NTSC_OUT_STORE((g_v4u_ptr)(rp + 4ull * (unsigned)(pend_x + (lane & 3) * 4)), to_v4u(v));
                }
            }
        }
        Uraw = 0; Vraw = 0;        // (only the row's last sample reads them, every guarded step rewrites them)
    }
    for (; t < total; t++) edge(t);
}

} // namespace pipe

// One workgroup = 63 rows + the halo row above, five wavefronts = five roles (see the head of this file).
// Preconditions (launcher): the -vhs preset family of the hand-tuned kernels (input chroma low-pass on, no pre-emphasis, luma /
// chroma / phase noise on, amplitudes 50 / 50, even scanline phase, output low-pass "lite", composite out), head-switch
// displacement within W/10 (WR = false) or any (WR = true), 16-byte aligned rows, planes below 4 GiB, no ghosting.
// WR: head-switch displacements of any size (wrap-around loads, wg_reach) -- e.g. PAL with its default switching point.
// SV: the -vhs preset with S-Video out of the VCR (-vhs-svideo 1): k_decode_fast_sv's positions (8 + d deep).
// XA: scanline phases of either parity (-comp-phase 90 / 270, odd -comp-phase-offset): k_encode_fast_xi / k_decode_fast_xi's forms.
// CATV: the pre-emphasis presets (-comp-catv, -comp-catv2/3/4): composite pre-emphasis in the encoder role, the first
// separator's 50 / subcarrier_amplitude_back -- k_encode_fast_pre / k_decode_fast_bk's forms.
template <class RT, bool WR = false, bool SV = false, bool XA = false, bool CATV = false>
__global__ __launch_bounds__(320) void k_field_pipe(DevParams P, GeomDev G, const FieldDev *__restrict__ fields,
                                                    const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                                                    int *__restrict__ comp,
                                                    const uint32_t *__restrict__ rs_chroma, const int *__restrict__ n0_u,
                                                    const int *__restrict__ n0_v, const int *__restrict__ hs_shift,
                                                    const int *__restrict__ pn_noise, const int *__restrict__ dropout,
                                                    int *__restrict__ tails, unsigned order, unsigned long long *dbg,
                                                    unsigned *__restrict__ fault)
{
    using namespace pipe;
    __shared__ uint32_t ring_e[33 * 64];                                   // the encoder's rand() ring
    __shared__ __attribute__((aligned(16))) uint32_t ltile[64 * 20];       // its cooperative row loads
    __shared__ uint32_t etail[16 * 64];                                    // the pixels of its rows' ends, requested at the start
    __shared__ uint32_t ring_v[33 * 64];                                   // the chroma noise's rand() ring
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];      // the pixel staging
    __shared__ unsigned long long orow[64];
    __shared__ __attribute__((aligned(16))) uint32_t ring_ab[RING * 64 * 2];
    __shared__ __attribute__((aligned(16))) uint32_t ring_bc[RING * 64 * 2];
    __shared__ __attribute__((aligned(16))) uint32_t ring_cd[RING * 64 * 4];
    __shared__ uint32_t flags[F_COUNT];

    const int role = (int)((order >> (4 * (threadIdx.x >> 6))) & 15u);
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
    if (threadIdx.x < 64) orow[R.lane] = R.is_out ? (unsigned long long)drow : 0ull;
    if (threadIdx.x < F_COUNT) { flags[threadIdx.x] = 0u; g_waited[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) g_fault = 0u;
    __syncthreads();
    const unsigned long long t_start = dbg ? wall_clock64() : 0ull;
    const lds_flag fl = (lds_flag)flags;
    const lds_x2 ab = (lds_x2)ring_ab;
    const lds_x2 bc = (lds_x2)ring_bc;
    const lds_x4 cd = (lds_x4)ring_cd;
    if (role == 0) encoder_role<RT, XA, CATV>(P, R, rs_luma, n0_luma, comp, ring_e, ltile, etail, fl);
    else if (role == 1) sep_role<RT, WR, SV, XA, CATV>(P, R, comp, rs_chroma, n0_u, n0_v, hs_shift, ring_v, ab, fl);
    else if (role == 2) chroma_role<RT, SV, XA>(P, G, R, pn_noise, tails, ab, bc, fl);
    else if (role == 3) luma_role<RT, WR, SV, XA>(P, R, comp, hs_shift, dropout, bc, cd, fl);
    else output_role<RT, true, SV>(P, R, ostage, orow, drow, cd, fl);
    if (R.lane == 0 && *(lds_flag)&g_fault) *fault = 1u + blockIdx.x;
    if (dbg && R.lane == 0) {      // NTSCSIM_PIPE_TIMING: start, end, ticks spent polling -- per workgroup and role
        unsigned long long *o = dbg + ((size_t)blockIdx.x * 5 + role) * 3;
        o[0] = t_start; o[1] = wall_clock64();
        o[2] = g_waited[threadIdx.x >> 6] | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32);      // HW_ID: SIMD, CU
    }
}

// The default preset (no VCR) as three roles: ENC | TVF | OUT (same workgroup shape, same hand-offs; the launcher's
// preconditions are those of k_encode_fast / k_decode_fast<false>).
// CATV: the pre-emphasis presets without the VCR (-comp-catv ... -comp-catv4): ENC with k_encode_fast_pre's steps, TVF
// with 50 / subcarrier_amplitude_back and the chroma phase noise of those presets.
template <class RT, bool CATV = false>
__global__ __launch_bounds__(192) void k_field_pipe_tv(DevParams P, const FieldDev *__restrict__ fields,
                                                       const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                                                       int *__restrict__ comp, const int *__restrict__ hs_shift,
                                                       const int *__restrict__ dropout, const int *__restrict__ pn_noise,
                                                       const double *__restrict__ ptab, unsigned *__restrict__ fault)
{
    using namespace pipe;
    __shared__ uint32_t ring_e[33 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t ltile[64 * 20];
    __shared__ uint32_t etail[16 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];
    __shared__ unsigned long long orow[64];
    __shared__ __attribute__((aligned(16))) uint32_t ring_cd[RING * 64 * 4];
    __shared__ uint32_t flags[F_COUNT];

    const int role = (int)(threadIdx.x >> 6);
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
    if (threadIdx.x < 64) orow[R.lane] = R.is_out ? (unsigned long long)drow : 0ull;
    if (threadIdx.x < F_COUNT) { flags[threadIdx.x] = 0u; g_waited[threadIdx.x] = 0u; }
    if (threadIdx.x == 0) g_fault = 0u;
    __syncthreads();
    const lds_flag fl = (lds_flag)flags;
    const lds_x4 cd = (lds_x4)ring_cd;
    if (role == 0) encoder_role<RT, false, CATV>(P, R, rs_luma, n0_luma, comp, ring_e, ltile, etail, fl);
    else if (role == 1) tvfront_role<RT, CATV>(P, R, comp, hs_shift, dropout, pn_noise, ptab, cd, fl);
    else output_role<RT, false>(P, R, ostage, orow, drow, cd, fl);
    if (R.lane == 0 && *(lds_flag)&g_fault) *fault = 1u + blockIdx.x;
}

} // namespace ntscsim
