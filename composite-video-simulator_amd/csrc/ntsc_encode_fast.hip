// ntsc_encode_fast.hip -- the encoder of the default / -vhs presets (BGRA -> composite signal),
// trimmed for the gfx950 VALU the same way as k_decode_fast (ntsc_decode_fast.hip): same results
// as k_encode (ntsc_kernels.hip), fewer and cheaper instructions.
//   * RGB -> YIQ (ffmpeg_ntsc.cpp:1375-1383) is evaluated on 256 * {r, g, b}: scaling by a power
//     of two commutes with every rounding, so (int)(256 * expr(r, g, b)) == (int)expr(256r, 256g,
//     256b) bit for bit, the three `256 *` multiplications disappear and the byte extraction is
//     full-rate and / shift work;
//   * the two input chroma low-passes (composite_lowpass :1429-1458) run in carry form (3 fp64
//     instructions per pole), fed with trunc() instead of an int round trip;
//   * 16 pixels per iteration, fully unrolled: every delay line is register renaming, the
//     subcarrier phase of each position is a compile-time constant (even scanline phase) and the
//     sign of the modulated chroma is mask arithmetic on full-rate opcodes;
//   * stores through a buffer descriptor (uniform row offset in an SGPR), unconditional: every
//     lane owns a column of the transposed plane, also the ones past the last row.
// Preconditions (launcher): input chroma low-pass on, no pre-emphasis, luma noise on, subcarrier
// amplitude 50, even scanline phase, 16-byte aligned source rows, composite plane below 4 GiB.
#pragma clang fp contract(off)

// Cache hints (round 4, tools/nt_probe.sh): the composite plane is written here and read back by another kernel long
// after the L2 has forgotten it, so its stores are streaming (nt) ones -- they no longer push the source rows of the
// cooperative loader out of the L2 before their second half is read (FETCH_SIZE of this kernel -12 %).
#ifndef NTSC_COMP_STORE_AUX
#define NTSC_COMP_STORE_AUX 2      /* 0 = plain stores (A/B) */
#endif
#ifdef NTSC_ENC_LOAD_NT            /* A/B: streaming loads of the source pixels -- measured WORSE (fetch x 2, kernel 0.26 -> 0.33 ms):
                                      each 128-byte source line is read as two 64-byte halves a chunk apart */
#define NTSC_ENC_LOAD(p) __builtin_nontemporal_load(p)
#else
#define NTSC_ENC_LOAD(p) (*(p))
#endif
namespace ntscsim {
namespace fastenc {

using fastdec::Casc3;
using fastdec::opaque_v;

template <class RT>
struct EState {
    Casc3<RT> lpI, lpQ;
    int Yd[4];            // 256 * luma of pixels t-4 .. t-1
    int Ir[4], Qr[4];     // raw I, Q of pixels t-4 .. t-1 (row tail :1447-1455), edge steps only
    int fI[2];            // filtered I pushed at t-2, t-1
    fastdec::LaneRand32 rng;   // 32 slots + a copy of slot 0: the 16 draws of a chunk are immediate LDS offsets from one address
    int noise;
    OnePoleT<RT> pre;     // PRE: the pre-emphasis high-pass :1610-1623
};

template <class RT>
struct EConst {
    unsigned xi;
    int W, lane;
    int mL, mNL;          // -1 / 0 masks of (xi == 2) and its complement
    bool odd;             // XA (scanline phases of either parity): xi is odd
    int ms[4];            // XA: sign of the modulated chroma at unrolled position J: -1 where (xi + J) & 2
    RT a_i, a_q;
    RT a_pre, pre_gain;   // PRE
    __amdgpu_buffer_rsrc_t comp;
    int vcol;             // byte offset of this lane's column
    int rowbytes;
};

// 256 * (Y, I, Q) of one BGRA pixel as reals; I and Q already truncated like the reference's (int)
template <class RT>
DEV void rgb_to_yiq256(uint32_t px, RT &dY, RT &Id, RT &Qd)
{
#ifdef NTSC_ENC_SHIFT_EXTRACT
    const RT r = (RT)((px >> 8) & 0xFF00u), g = (RT)(px & 0xFF00u), b = (RT)((px << 8) & 0xFF00u);
#else
    // 256 * channel = the channel's byte moved to byte 1 of an otherwise zero word: one v_perm_b32 each for R (byte 2) and
    // B (byte 0) instead of shift + mask (selector 0x0c = constant zero), G is in place already
    const RT r = (RT)__builtin_amdgcn_perm(0u, px, 0x0c0c020cu), g = (RT)(px & 0xFF00u),
             b = (RT)__builtin_amdgcn_perm(0u, px, 0x0c0c000cu);
#endif
    dY = ((RT(0.30) * r) + (RT(0.59) * g)) + (RT(0.11) * b);
    const RT bd = b - dY, rd = r - dY;
    Id = rtrunc<RT>((RT(-0.27) * bd) + (RT(0.74) * rd));
    Qd = rtrunc<RT>((RT(0.41) * bd) + (RT(0.48) * rd));
}

// One steady-state step at unrolled position J of a 16-pixel chunk starting at t0 = 0 (mod 4):
// consumes pixel t = t0 + J, emits composite sample x = t - 4 = J (mod 4).
//   Yx = 256 * luma of pixel x, I2 = filtered I pushed two steps ago (index x).
// composite pre-emphasis :1610-1623 (the -comp-catv* presets), as in enc_step of ntsc_kernels.hip
template <class RT>
DEV int preemphasis(EState<RT> &S, const EConst<RT> &C, int Y)
{
    RT sd = (RT)Y;
    sd += S.pre.hp(sd, C.a_pre) * C.pre_gain;
    return (int)sd;
}

template <int J, class RT, bool PRE, bool XA = false>
DEV int step(const DevParams &P, EState<RT> &S, const EConst<RT> &C, uint32_t *rb, bool rb0,
             RT Id, RT Qd, int Yx, int I2, int &fI_out)
{
    fI_out = (int)S.lpI.push(Id, C.a_i);            // lands at index t - 2
    const int fQ = (int)S.lpQ.push(Qd, C.a_q);      // lands at index t - 4 = x
    // chroma_into_luma :1460-1495, phase (xi + x) & 3 with xi in {0, 2}: I for even x, sign by
    // (x & 2) ^ (xi & 2)
    // XA: xi of either parity, per lane (-comp-phase 90 / 270, odd offsets): I / Q swap roles on odd lanes
    const int chroma = XA ? ((J & 1) ? (C.odd ? I2 : fQ) : (C.odd ? fQ : I2)) : ((J & 1) ? fQ : I2);
    const int mm = XA ? C.ms[J & 3] : ((J & 2) ? C.mNL : C.mL);
    int Y = Yx + ((chroma ^ mm) - mm);
    if (PRE) Y = preemphasis<RT>(S, C, Y);
    // luma noise :1632-1644
    Y += S.noise;
    S.noise = sdiv2(S.noise + (int)umod31(S.rng.template draw<J>(rb, rb0), P.m_noise) - P.noise_k);
    return Y;
}

// EXTENSION (ghosting, see ntscsim.h) folded into the encoder: k_ghost (ntsc_kernels.hip) reads the raw plane 1 + taps
// times and writes a second one, a pass longer than the encoder itself.  Here every lane keeps the last 64 raw samples of
// its scanline in an LDS ring (slot x & 63, one column per lane) and stores the ghosted sample only:
//   out[x] = raw[x] + (sum_k gain_k * raw[x - delay_k]) / 256,  raw[< 0] = 0
// -- the same integer expression (|raw| < 2^16 + 2^17 + noise_k <= 1.25 M, so four products of at most 256 * that stay
// below 2^31: no overflow, the order of the sum does not matter).  The ring starts
// zeroed, so a tap that reaches before the row start reads a slot not yet written (delay <= 63 < 64).  GT = taps of the
// form (2 or 4; unused ones run with gain 0, delay 1: no branch in the step), 0 = no ghosting.  The products are 24-bit
// ones (full-rate v_mad_i32_i24 instead of the quarter-rate 32-bit multiply): |gain| <= 256 by ntscsim_params_validate,
// |raw| < 2^23 because 256 * luma < 2^16, the low-passed chroma stays inside its input range (< 2^17) and the luma noise
// inside +-noise_k.  Launcher: every delay <= NTSC_GHOST_RING - 1, noise_k <= 2^20.
#define NTSC_GHOST_RING 64
template <int GT>
struct GhostRing {
    uint32_t *col;            // this lane's column of the ring: slot s at col[64 * s]
    int delay[GT > 0 ? GT : 1], gain[GT > 0 ? GT : 1];
    DEV void begin(const DevParams &P, uint32_t *lds, int lane)
    {
        if (GT == 0) return;
        col = lds + lane;
#pragma unroll
        for (int k = 0; k < GT; k++) {
            const bool on = k < P.ghost_taps;
            delay[k] = on ? P.ghost_delay[k] : 1;
            gain[k] = on ? P.ghost_gain[k] : 0;
        }
#pragma unroll 8
        for (int s = 0; s < NTSC_GHOST_RING; s++) col[64 * s] = 0u;
    }
    // raw sample of position x (wave-uniform) -> the sample the decoder sees
    DEV int emit(int x, int Y) const
    {
        if (GT == 0) return Y;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < GT; k++)
            acc += __mul24(gain[k], (int)col[64 * ((x - delay[k]) & (NTSC_GHOST_RING - 1))]);
        col[64 * (x & (NTSC_GHOST_RING - 1))] = (uint32_t)Y;
        return Y + acc / 256;
    }
};

// One guarded step at any stream position t (wave-uniform): row start, row end, filter tails.
// (edge_step_px: the pixel of position t -- 0 behind the row's end -- is handed in; edge_step loads it from the frame)
template <class RT, bool PRE, int GH = 0>
DEV void edge_step_px(const DevParams &P, EState<RT> &S, const EConst<RT> &C, uint32_t *ring,
                      const uint32_t px, int t, const GhostRing<GH> &gh = GhostRing<GH>())
{
    const int W = C.W;
    RT dY, Id, Qd;
    rgb_to_yiq256<RT>(px, dY, Id, Qd);
    const int Yx = S.Yd[0], Ix = S.Ir[0], Qx = S.Qr[0];          // pixel t - 4
#pragma unroll
    for (int q = 0; q < 3; q++) { S.Yd[q] = S.Yd[q + 1]; S.Ir[q] = S.Ir[q + 1]; S.Qr[q] = S.Qr[q + 1]; }
    S.Yd[3] = (int)dY; S.Ir[3] = (int)Id; S.Qr[3] = (int)Qd;
    const int I2 = S.fI[0];
    S.fI[0] = S.fI[1];
    S.fI[1] = (int)S.lpI.push(Id, C.a_i);
    const int fQ = (int)S.lpQ.push(Qd, C.a_q);
    const int x = t - 4;
    if (x < 0) return;
    const int I1 = x < W - 2 ? I2 : Ix;                           // the last `delay` samples keep their input
    const int Q1 = x < W - 4 ? fQ : Qx;
    const unsigned s = (C.xi + (unsigned)x) & 3u;
    int chroma = (s & 1u) ? Q1 : I1;
    if (s & 2u) chroma = -chroma;
    int Y = Yx + chroma;
    if (PRE) Y = preemphasis<RT>(S, C, Y);
    Y += S.noise;
    S.noise = sdiv2(S.noise + (int)umod31(S.rng.next(ring, C.lane), P.m_noise) - P.noise_k);
    Y = gh.emit(x, Y);
    __builtin_amdgcn_raw_buffer_store_b32(Y, C.comp, C.vcol, (int)((unsigned)x * (unsigned)C.rowbytes), NTSC_COMP_STORE_AUX);
}

template <class RT, bool PRE, int GH = 0>
DEV void edge_step(const DevParams &P, EState<RT> &S, const EConst<RT> &C, uint32_t *ring,
                   const uint32_t *srow, int t, const GhostRing<GH> &gh = GhostRing<GH>())
{
    const uint32_t px = t < C.W ? ((fastdec::g_cu32_ptr)srow)[t] : 0u;
    edge_step_px<RT, PRE, GH>(P, S, C, ring, px, t, gh);
}

DEV void load_chunk(const uint8_t *srow, int t0, uint32_t (&px)[16])
{
#ifdef NTSC_AB_ENC_NOLOAD   // timing-only A/B build (WRONG pixels): no frame loads
    for (int j = 0; j < 16; j++) px[j] = 0x00406080u + (uint32_t)(t0 + j) * 0x010101u;
    return;
#endif
    fastdec::g_cv4u_ptr p = (fastdec::g_cv4u_ptr)(srow + 4 * (size_t)t0);          // global, not FLAT
    const fastdec::v4u a = p[0], b = p[1], c = p[2], d = p[3];
    px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w;
    px[4] = b.x; px[5] = b.y; px[6] = b.z; px[7] = b.w;
    px[8] = c.x; px[9] = c.y; px[10] = c.z; px[11] = c.w;
    px[12] = d.x; px[13] = d.y; px[14] = d.z; px[15] = d.w;
}

// The same 16 pixels of 64 rows, loaded COOPERATIVELY: a lane's own 64 bytes are one of 64 scattered pieces per
// load instruction (one per row), which costs the memory pipeline more than the bytes are worth (stubbing the
// loads out made the whole path 5 % faster).  Here four consecutive lanes fetch one row's 64 contiguous bytes,
// 16 rows per instruction; the pieces go through an LDS tile (row stride 20 words: conflict-free for the b128
// reads) and every lane reads its own row back.  ptr[i] = address of this lane's piece in row group i at t = 0.
struct CoopLoader {
    const uint8_t *ptr[4];
    uint32_t *tile;            // [64][20] words, wave-private
    int wr, rd;                // word index of this lane's piece / of this lane's row
    DEV void begin(const uint8_t *srow, uint32_t *lds_tile, int lane)
    {
        tile = lds_tile;
        const unsigned lo = (unsigned)(uintptr_t)srow, hi = (unsigned)((uintptr_t)srow >> 32);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int from = 16 * i + (lane >> 2);
            const unsigned l2 = (unsigned)__shfl((int)lo, from), h2 = (unsigned)__shfl((int)hi, from);
            ptr[i] = (const uint8_t *)(((uintptr_t)h2 << 32) | l2) + 16 * (lane & 3);
        }
        wr = (lane >> 2) * 20 + (lane & 3) * 4;
        rd = lane * 20;
    }
    // request the pieces of pixels t0 .. t0+15 of all 64 rows (this lane's four of them)
    DEV void request(int t0, fastdec::v4u (&q)[4]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = NTSC_ENC_LOAD((fastdec::g_cv4u_ptr)(ptr[i] + 4 * (size_t)t0));
    }
    // pieces -> tile -> this lane's 16 pixels
    DEV void deliver(const fastdec::v4u (&q)[4], uint32_t (&px)[16]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(&tile[16 * 20 * i + wr]) = make_uint4(q[i].x, q[i].y, q[i].z, q[i].w);
        // (the wave runs in lock-step and the tile is its own: LDS accesses of one wave complete in order)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&tile[rd + 4 * i]);
            px[4 * i] = v.x; px[4 * i + 1] = v.y; px[4 * i + 2] = v.z; px[4 * i + 3] = v.w;
        }
    }
};

} // namespace fastenc

template <class RT, bool PRE, bool XA = false, int GH = 0>
DEV void encode_fast_body(const DevParams &P, const FieldDev *__restrict__ fields, const uint32_t *__restrict__ rs_luma,
                          const int *__restrict__ n0_luma, int *__restrict__ comp)
{
    using namespace fastenc;
    __shared__ uint32_t ring[33 * 64];            // LaneRand32
#ifndef NTSC_ENC_NOCOOP
    __shared__ __attribute__((aligned(16))) uint32_t ltile[64 * 20];
#endif
    __shared__ uint32_t gring[GH ? NTSC_GHOST_RING * 64 : 1];
    const int lane = threadIdx.x;
    const int rho = blockIdx.x * 64 + lane;
    const int rc = rho < P.R ? rho : P.R - 1;
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool valid = rho < P.R && (int)(field + 2u * k) < P.H;
    const unsigned y = valid ? field + 2u * (unsigned)k : field;
    // source row: min(y + opposite, H-1), ffmpeg_ntsc.cpp:1585-1588, :1599
    const unsigned opposite = (fd.flags & 1u) ? ((fd.flags & 2u) ? 1u : 0u) : 0u;
    unsigned sy = y + opposite;
    if (sy > (unsigned)P.H - 1u) sy = (unsigned)P.H - 1u;
    const uint8_t *srow = fd.src + (size_t)fd.src_ls * sy;
    const int W = P.W;

    EConst<RT> C;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.W = W;
    C.lane = lane;
    C.mL = opaque_v((C.xi & 2u) ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    C.odd = (C.xi & 1u) != 0;
#pragma unroll
    for (int j = 0; j < 4; j++) C.ms[j] = fastdec::opaque_v(((C.xi + (unsigned)j) & 2u) ? -1 : 0);
    C.a_i = (RT)P.a_in_i; C.a_q = (RT)P.a_in_q;
    C.a_pre = (RT)P.a_pre; C.pre_gain = (RT)P.pre_gain;
    C.rowbytes = P.Rpad * 4;
    C.vcol = rho * 4;                  // rho < Rpad: lanes past the last row own padding columns
    C.comp = __builtin_amdgcn_make_buffer_rsrc(comp, 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    EState<RT> S;
    // (no draw before the steady loop -- the row's first four steps only fill the look-ahead -- so the window is placed
    //  with its first draw on slot 0: the loop's 16 draws per chunk start on a multiple of 16)
    S.rng.init(ring, rs_luma + rc, P.Rpad, lane, 1);
    S.noise = n0_luma[rc];
    S.lpI.reset(0, C.a_i); S.lpQ.reset(0, C.a_q);
    S.pre.p = 16;
#pragma unroll
    for (int q = 0; q < 4; q++) { S.Yd[q] = 0; S.Ir[q] = 0; S.Qr[q] = 0; }
    S.fI[0] = S.fI[1] = 0;
    GhostRing<GH> gh;
    gh.begin(P, gring, lane);

    // ---------------- row start: pixels 0..3 fill the 4-sample look-ahead of the Q low-pass
    int t = 0;
    for (; t < 4; t++) edge_step<RT, PRE, GH>(P, S, C, ring, reinterpret_cast<const uint32_t *>(srow), t, gh);
    // ---------------- steady state: 16-pixel chunks strictly inside the row
    if (t + 16 <= W) {
#ifndef NTSC_ENC_NOCOOP
        CoopLoader L;
        L.begin(srow, ltile, lane);
        uint32_t cur[16];
        fastdec::v4u nq[4];
        L.request(t, nq);
        L.deliver(nq, cur);
#else
        uint32_t cur[16], nxt[16];
        load_chunk(srow, t, cur);
#endif
        int Y0 = S.Yd[0], Y1 = S.Yd[1], Y2 = S.Yd[2], Y3 = S.Yd[3];   // luma of pixels t-4 .. t-1
        int I0 = S.fI[0], I1 = S.fI[1];                               // filtered I of indices t-4, t-3
        RT IdT[4], QdT[4];     // raw I, Q of the chunk's last four pixels (row tail)
        int sbase = S.rng.pos;          // 0: see init
        for (; t + 16 <= W; t += 16) {
            const bool more = t + 32 <= W;
            uint32_t *const rb = ring + sbase * 64 + lane;
            const bool rb0 = sbase == 0;
            sbase = (sbase + 16) & 31;
#ifndef NTSC_ENC_NOCOOP
            if (more) L.request(t + 16, nq);
#else
            if (more) load_chunk(srow, t + 16, nxt);
#endif
            unsigned soff = (unsigned)(t - 4) * (unsigned)C.rowbytes;
            int Yn[16], F[16];
#define NTSC_ENC_STEP(J, YX, IX)                                                                  \
            {                                                                                     \
                RT dY, Id_, Qd_;                                                                  \
                rgb_to_yiq256<RT>(cur[J], dY, Id_, Qd_);                                          \
                Yn[J] = (int)dY;                                                                  \
                if (J >= 12) { IdT[J & 3] = Id_; QdT[J & 3] = Qd_; }                              \
                const int Y = gh.emit(t - 4 + J, step<J, RT, PRE, XA>(P, S, C, rb, rb0, Id_, Qd_, YX, IX, F[J])); \
                __builtin_amdgcn_raw_buffer_store_b32(Y, C.comp, C.vcol, (int)soff, NTSC_COMP_STORE_AUX);          \
                soff += (unsigned)C.rowbytes;                                                     \
            }
            NTSC_ENC_STEP(0, Y0, I0)
            NTSC_ENC_STEP(1, Y1, I1)
            NTSC_ENC_STEP(2, Y2, F[0])
            NTSC_ENC_STEP(3, Y3, F[1])
            NTSC_ENC_STEP(4, Yn[0], F[2])
            NTSC_ENC_STEP(5, Yn[1], F[3])
            NTSC_ENC_STEP(6, Yn[2], F[4])
            NTSC_ENC_STEP(7, Yn[3], F[5])
            NTSC_ENC_STEP(8, Yn[4], F[6])
            NTSC_ENC_STEP(9, Yn[5], F[7])
            NTSC_ENC_STEP(10, Yn[6], F[8])
            NTSC_ENC_STEP(11, Yn[7], F[9])
            NTSC_ENC_STEP(12, Yn[8], F[10])
            NTSC_ENC_STEP(13, Yn[9], F[11])
            NTSC_ENC_STEP(14, Yn[10], F[12])
            NTSC_ENC_STEP(15, Yn[11], F[13])
#undef NTSC_ENC_STEP
            Y0 = Yn[12]; Y1 = Yn[13]; Y2 = Yn[14]; Y3 = Yn[15];
            I0 = F[14]; I1 = F[15];
#ifndef NTSC_ENC_NOCOOP
            if (more) L.deliver(nq, cur);
#else
            if (more) {
#pragma unroll
                for (int j = 0; j < 16; j++) cur[j] = nxt[j];
            }
#endif
        }
        S.rng.pos = sbase;
        // hand the delay lines back to the guarded steps
        S.Yd[0] = Y0; S.Yd[1] = Y1; S.Yd[2] = Y2; S.Yd[3] = Y3;
        S.fI[0] = I0; S.fI[1] = I1;
#pragma unroll
        for (int q = 0; q < 4; q++) { S.Ir[q] = (int)IdT[q]; S.Qr[q] = (int)QdT[q]; }
    }
    // ---------------- row end + drain
    for (; t < W + 4; t++) edge_step<RT, PRE, GH>(P, S, C, ring, reinterpret_cast<const uint32_t *>(srow), t, gh);
}

template <class RT>
__global__ __launch_bounds__(64) void k_encode_fast(DevParams P, const FieldDev *__restrict__ fields,
                                                    const uint32_t *__restrict__ rs_luma,
                                                    const int *__restrict__ n0_luma,
                                                    int *__restrict__ comp)
{
    encode_fast_body<RT, false>(P, fields, rs_luma, n0_luma, comp);
}

// the same with the ghosting extension folded in (GT = 2 or 4 taps, delays <= 63): stores the ghosted plane, no k_ghost pass
template <class RT, int GT>
__global__ __launch_bounds__(64) void k_encode_fast_gh(DevParams P, const FieldDev *__restrict__ fields,
                                                       const uint32_t *__restrict__ rs_luma,
                                                       const int *__restrict__ n0_luma,
                                                       int *__restrict__ comp)
{
    encode_fast_body<RT, false, false, GT>(P, fields, rs_luma, n0_luma, comp);
}

// the same for scanline phases of either parity (-comp-phase 90 / 270, odd -comp-phase-offset)
template <class RT>
__global__ __launch_bounds__(64) void k_encode_fast_xi(DevParams P, const FieldDev *__restrict__ fields,
                                                       const uint32_t *__restrict__ rs_luma,
                                                       const int *__restrict__ n0_luma,
                                                       int *__restrict__ comp)
{
    encode_fast_body<RT, false, true>(P, fields, rs_luma, n0_luma, comp);
}

// the same with composite pre-emphasis (ffmpeg_ntsc.cpp:1614-1629; the -comp-catv* presets)
template <class RT>
__global__ __launch_bounds__(64) void k_encode_fast_pre(DevParams P, const FieldDev *__restrict__ fields,
                                                        const uint32_t *__restrict__ rs_luma,
                                                        const int *__restrict__ n0_luma,
                                                        int *__restrict__ comp)
{
    encode_fast_body<RT, true>(P, fields, rs_luma, n0_luma, comp);
}

} // namespace ntscsim
