// ntsc_float.hip -- NTSCSIM_MODE_FLOAT: the tolerance mode as a pipeline of its own.
//
// BASELINE north_star: "bit-exact output vs the reference for integer pixel paths and within a stated fp32 tolerance
// for the filtered signal".  NTSCSIM_MODE_FAST32 is the exact kernels with float filter states: it still carries every
// inter-stage `(int)` of the reference (ffmpeg_ntsc.cpp:1422, :1453, :1810, :1830, :1880), i.e. a v_cvt / v_trunc pair
// between any two stages, and the separators' integer boxes, shifts and masks.  Here the signal -- Y, I, Q and the
// composite sample, all scaled by 256 like the reference's int planes -- is fp32 from RGB -> YIQ (:1375) to YIQ -> RGB
// (:1385):
//   * no inter-stage truncation and no int <-> float conversion of the signal; FMA contraction on; one-pole filters
//     as p += a * (s - p) (2 instructions per pole, 25 poles per pixel on the -vhs path);
//   * the Y/C separators' boxes, averages, sign flips, the vertical blend, the dropout mask and the (re-)modulation
//     are float multiplies / FMAs (x * 0.25, x * 0.5, x * +-1, x * {0, 1});
//   * the composite plane between encoder and decoder holds floats;
//   * what stays integer: the glibc rand() stream and the three noise accumulators (:1632-1644, :1719-1735) -- the
//     noise IS its integer recurrence -- converted once where it is added; head-switch geometry, phase-noise table
//     and dropout come from the same setup kernels as in the exact mode;
//   * dropping a truncation of a positive signal moves its mean by +1/2 (of 1/256 of an output step): the luma
//     path's five dropped truncations are compensated by ONE constant in the output matrix (NTSC_FP_LUMA_BIAS).
// Same execution model as the exact kernels (ntsc_kernels.hip): one lane = one scanline, everything streamed in x,
// the transposed composite plane comp[x][row] between encoder and decoder (head switching needs random access in x).
// Same row-start / row-end semantics (filter resets, raw tails of the delayed filters, zero fill) as the reference:
// the guarded steps below mirror ntsc_decode_fast.hip / ntsc_encode_fast.hip position by position.
// Forms: default preset and the -vhs family with its standard switches (input chroma low-pass on, no pre-emphasis,
// amplitudes 50 / 50, even scanline phase, output low-pass "lite", composite out, head-switch displacement <= W/10);
// every other switch set runs the FAST32 forms in this mode (same tolerance, tests/test_gpu_fast_mode.py).
// This file is a translation unit of its own (csrc/Makefile: ntsc_float.o, -ffp-contract=fast -fno-slp-vectorize): the
// exact kernels' TU is compiled with -ffp-contract=off, and the SLP vectoriser -- which pairs the U and V chains into
// half-rate v_pk_*_f32 here -- costs this pipeline 30-36 VGPRs and a v_mov per packed operand for nothing.  The small
// device helpers it shares with the exact kernels are restated below (same definitions as ntsc_kernels.hip /
// ntsc_decode_fast.hip / ntsc_encode_fast.hip); ntscsim_hip.hip calls launch_encode_fp() / launch_decode_fp().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ntsc_device.hpp"
#include "ntsc_float.hpp"

#pragma clang fp contract(fast)

#ifndef NTSC_FP_LUMA_BIAS
// five truncations of the luma path dropped (encoder :1381, first box :1517, VHS luma :1810, sharpen :1880, second
// box :1517), each worth 1/2 (the two boxes 3/8): measured against the oracle (tools/float_err.py: mean signed
// difference per channel within +-0.001 on noise frames)
#define NTSC_FP_LUMA_BIAS_VHS (-2.25f)
#define NTSC_FP_LUMA_BIAS_DEF (-0.875f)
#endif

namespace ntscsim {

#define DEV __device__ __forceinline__
#define NTSC_COMP_STORE_AUX 2            /* the plane is read back long after the L2 has forgotten it: streaming stores */
#define NTSC_COMP_LOAD2_AUX 2            /* nt on the luma path's second, trailing read of every sample */

namespace fpipe {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4u *g_v4u_ptr;
typedef __attribute__((address_space(1))) const v4u *g_cv4u_ptr;
typedef __attribute__((address_space(1))) uint32_t *g_u32_ptr;
typedef __attribute__((address_space(1))) const uint32_t *g_cu32_ptr;
DEV v4u to_v4u(const uint4 &a) { return v4u{a.x, a.y, a.z, a.w}; }

DEV int sdiv2(int n) { return (n + (int)((unsigned)n >> 31)) >> 1; }   // C `/ 2` (truncating)
DEV unsigned umod31(unsigned n, const Magic31 &m) { return n - (__umulhi(n, m.mul) >> m.shift) * m.div; }
DEV unsigned scan_phase(const DevParams &P, unsigned y, uint64_t fieldno)      // ffmpeg_ntsc.cpp:1473-1480
{
    const unsigned off = (unsigned)P.phase_off;
    if (P.phase_mode == 90)  return (unsigned)((fieldno + off + (y >> 1)) & 3);
    if (P.phase_mode == 180) return (unsigned)((((fieldno + y) & 2) + off) & 3);
    if (P.phase_mode == 270) return (unsigned)((fieldno + off - (y >> 1)) & 3);
    return off & 3;
}
DEV uint32_t cvt_sat_u32(float x) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
DEV uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// rand() ring with static LDS offsets (ntsc_decode_fast.hip: LaneRand32): 32 slots + a copy of slot 0
struct LaneRand32 {
    uint32_t p3, p2, p1;
    int pos;
    DEV void init(uint32_t *ring, const uint32_t *state, int stride, int lane, int o)
    {
        for (int j = 0; j < 31; j++) {
            const uint32_t w = state[(size_t)j * stride];
            const int sl = (o + j) & 31;
            ring[sl * 64 + lane] = w;
            if (sl == 0) ring[32 * 64 + lane] = w;
            if (j == 28) p3 = w;
            if (j == 29) p2 = w;
            if (j == 30) p1 = w;
        }
        pos = (o + 31) & 31;
    }
    DEV uint32_t next(uint32_t *ring, int lane)
    {
        const uint32_t v = ring[(pos + 1) * 64 + lane] + p3;
        ring[pos * 64 + lane] = v;
        if (pos == 0) ring[32 * 64 + lane] = v;
        p3 = p2; p2 = p1; p1 = v;
        pos = (pos + 1) & 31;
        return v >> 1;
    }
    template <int K>
    DEV uint32_t draw(uint32_t *rb, bool first_slot_is_zero)
    {
        const uint32_t v = rb[(K + 1) * 64] + p3;
        rb[K * 64] = v;
        if (K == 0 && first_slot_is_zero) rb[32 * 64] = v;
        p3 = p2; p2 = p1; p1 = v;
        return v >> 1;
    }
};

// three one-pole low-passes with one alpha (LowpassFilter x3, :74-106) as p += a * (s - p)
struct Casc3 {
    float p0, p1, p2;
    DEV void reset(float v) { p0 = p1 = p2 = v; }
    DEV float push(float s, float a)
    {
        p0 = __builtin_fmaf(a, s - p0, p0);
        p1 = __builtin_fmaf(a, p0 - p1, p1);
        p2 = __builtin_fmaf(a, p1 - p2, p2);
        return p2;
    }
};
struct PoleHp {             // one more pole behind a cascade, used as a high-pass (s - lowpass(s))
    float p;
    DEV void reset(float v) { p = v; }
    DEV float hp(float s, float a) { p = __builtin_fmaf(a, s - p, p); return s - p; }
};

// 16 pixels of 64 rows loaded cooperatively (ntsc_encode_fast.hip: CoopLoader): four consecutive lanes fetch one row's 64
// contiguous bytes, the pieces go through an LDS tile and every lane reads its own row back
struct CoopLoader {
    const uint8_t *ptr[4];
    uint32_t *tile;
    int wr, rd;
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
    DEV void request(int t0, v4u (&q)[4]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = *((g_cv4u_ptr)(ptr[i] + 4 * (size_t)t0));
    }
    DEV void deliver(const v4u (&q)[4], uint32_t (&px)[16]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(&tile[16 * 20 * i + wr]) = make_uint4(q[i].x, q[i].y, q[i].z, q[i].w);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&tile[rd + 4 * i]);
            px[4 * i] = v.x; px[4 * i + 1] = v.y; px[4 * i + 2] = v.z; px[4 * i + 3] = v.w;
        }
    }
};

DEV float as_f(int v) { return __int_as_float(v); }
DEV int as_i(float v) { return __float_as_int(v); }
DEV float wave_up_f(float v) { return as_f(__builtin_amdgcn_update_dpp(0, as_i(v), 0x138, 0xf, 0xf, false)); }   // wave_shr:1

// ================================================================================================ encoder
struct EState {
    Casc3 lpI, lpQ;
    float Yd[4];           // 256 * luma of pixels t-4 .. t-1
    float Ir[4], Qr[4];    // raw I, Q of pixels t-4 .. t-1 (row tail :1447-1455), guarded steps only
    float fI[2];           // filtered I pushed at t-2, t-1
    LaneRand32 rng;
    int noise;
};

struct EConst {
    unsigned xi;
    int W, lane;
    float sg0, sg2;        // sign of the modulated chroma at positions x = 0, 1 / 2, 3 (mod 4): -1 where (xi + x) & 2
    float a_i, a_q;
    __amdgpu_buffer_rsrc_t comp;
    int vcol, rowbytes;
};

// 256 * (Y, I, Q) of one BGRA pixel, RGB_to_YIQ :1375-1383 without the (int)
DEV void rgb_to_yiq256(uint32_t px, float &Y, float &I, float &Q)
{
    const float r = (float)((px >> 16) & 0xFFu), g = (float)((px >> 8) & 0xFFu), b = (float)(px & 0xFFu);   // v_cvt_f32_ubyteN
    const float dY = __builtin_fmaf(0.11f, b, __builtin_fmaf(0.59f, g, 0.30f * r));
    const float bd = b - dY, rd = r - dY;
    Y = dY * 256.0f;
    I = __builtin_fmaf(0.74f * 256.0f, rd, (-0.27f * 256.0f) * bd);
    Q = __builtin_fmaf(0.48f * 256.0f, rd, (0.41f * 256.0f) * bd);
}

// steady step at unrolled position J of a 16-pixel chunk starting at t0 = 0 (mod 4): consumes pixel t = t0 + J, emits
// composite sample x = t - 4 = J (mod 4)
template <int J>
DEV float enc_step(const DevParams &P, EState &S, const EConst &C, uint32_t *rb, bool rb0, float Id, float Qd, float Yx,
                   float I2, float &fI_out)
{
    fI_out = S.lpI.push(Id, C.a_i);                 // lands at index t - 2
    const float fQ = S.lpQ.push(Qd, C.a_q);         // lands at index t - 4 = x
    // chroma_into_luma :1460-1495 at phase (xi + x) & 3, xi in {0, 2}: I for even x, Q for odd, sign by (xi + x) & 2
    const float chroma = (J & 1) ? fQ : I2;
    float Y = __builtin_fmaf(chroma, (J & 2) ? C.sg2 : C.sg0, Yx);
    // luma noise :1632-1644 (the integer recurrence, added as it stands)
    Y += (float)S.noise;
    S.noise = sdiv2(S.noise + (int)umod31(S.rng.template draw<J>(rb, rb0), P.m_noise) - P.noise_k);
    return Y;
}

// guarded step at any stream position t (wave-uniform): row start, row end, filter tails
DEV void enc_edge(const DevParams &P, EState &S, const EConst &C, uint32_t *ring, const uint32_t *srow, int t)
{
    const int W = C.W;
    const uint32_t px = t < W ? ((g_cu32_ptr)srow)[t] : 0u;
    float dY, Id, Qd;
    rgb_to_yiq256(px, dY, Id, Qd);
    const float Yx = S.Yd[0], Ix = S.Ir[0], Qx = S.Qr[0];            // pixel t - 4
#pragma unroll
    for (int q = 0; q < 3; q++) { S.Yd[q] = S.Yd[q + 1]; S.Ir[q] = S.Ir[q + 1]; S.Qr[q] = S.Qr[q + 1]; }
    S.Yd[3] = dY; S.Ir[3] = Id; S.Qr[3] = Qd;
    const float I2 = S.fI[0];
    S.fI[0] = S.fI[1];
    S.fI[1] = S.lpI.push(Id, C.a_i);
    const float fQ = S.lpQ.push(Qd, C.a_q);
    const int x = t - 4;
    if (x < 0) return;
    const float I1 = x < W - 2 ? I2 : Ix;                             // the last `delay` samples keep their input :1448-1453
    const float Q1 = x < W - 4 ? fQ : Qx;
    const unsigned s = (C.xi + (unsigned)x) & 3u;
    float chroma = (s & 1u) ? Q1 : I1;
    if (s & 2u) chroma = -chroma;
    float Y = Yx + chroma;
    Y += (float)S.noise;
    S.noise = sdiv2(S.noise + (int)umod31(S.rng.next(ring, C.lane), P.m_noise) - P.noise_k);
    __builtin_amdgcn_raw_buffer_store_b32(as_i(Y), C.comp, C.vcol, (int)((unsigned)x * (unsigned)C.rowbytes), NTSC_COMP_STORE_AUX);
}

} // namespace fpipe

__global__ __launch_bounds__(64) void k_encode_fp(DevParams P, const FieldDev *__restrict__ fields,
                                                  const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                                                  int *__restrict__ comp)
{
    using namespace fpipe;
    __shared__ uint32_t ring[33 * 64];            // LaneRand32
    __shared__ __attribute__((aligned(16))) uint32_t ltile[64 * 20];
    const int lane = threadIdx.x;
    const int rho = blockIdx.x * 64 + lane;
    const int rc = rho < P.R ? rho : P.R - 1;
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool valid = rho < P.R && (int)(field + 2u * k) < P.H;
    const unsigned y = valid ? field + 2u * (unsigned)k : field;
    const unsigned opposite = (fd.flags & 1u) ? ((fd.flags & 2u) ? 1u : 0u) : 0u;      // :1585-1588, :1599
    unsigned sy = y + opposite;
    if (sy > (unsigned)P.H - 1u) sy = (unsigned)P.H - 1u;
    const uint8_t *srow = fd.src + (size_t)fd.src_ls * sy;
    const int W = P.W;

    EConst C;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.W = W;
    C.lane = lane;
    C.sg0 = (C.xi & 2u) ? -1.0f : 1.0f;
    C.sg2 = -C.sg0;
    C.a_i = (float)P.a_in_i; C.a_q = (float)P.a_in_q;
    C.rowbytes = P.Rpad * 4;
    C.vcol = rho * 4;
    C.comp = __builtin_amdgcn_make_buffer_rsrc(comp, 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    EState S;
    S.rng.init(ring, rs_luma + rc, P.Rpad, lane, 1);
    S.noise = n0_luma[rc];
    S.lpI.reset(0); S.lpQ.reset(0);
#pragma unroll
    for (int q = 0; q < 4; q++) { S.Yd[q] = 0; S.Ir[q] = 0; S.Qr[q] = 0; }
    S.fI[0] = S.fI[1] = 0;

    int t = 0;
    for (; t < 4; t++) enc_edge(P, S, C, ring, reinterpret_cast<const uint32_t *>(srow), t);
    if (t + 16 <= W) {
        CoopLoader L;
        L.begin(srow, ltile, lane);
        uint32_t cur[16];
        v4u nq[4];
        L.request(t, nq);
        L.deliver(nq, cur);
        float Y0 = S.Yd[0], Y1 = S.Yd[1], Y2 = S.Yd[2], Y3 = S.Yd[3];
        float I0 = S.fI[0], I1 = S.fI[1];
        float IdT[4], QdT[4];
        int sbase = S.rng.pos;
        for (; t + 16 <= W; t += 16) {
            const bool more = t + 32 <= W;
            uint32_t *const rb = ring + sbase * 64 + lane;
            const bool rb0 = sbase == 0;
            sbase = (sbase + 16) & 31;
            if (more) L.request(t + 16, nq);
            unsigned soff = (unsigned)(t - 4) * (unsigned)C.rowbytes;
            float Yn[16], F[16];
#define NTSC_FP_ENC_STEP(J, YX, IX)                                                               \
            {                                                                                     \
                float dY, Id_, Qd_;                                                               \
                rgb_to_yiq256(cur[J], dY, Id_, Qd_);                                              \
                Yn[J] = dY;                                                                       \
                if (J >= 12) { IdT[J & 3] = Id_; QdT[J & 3] = Qd_; }                              \
                const float Y = enc_step<J>(P, S, C, rb, rb0, Id_, Qd_, YX, IX, F[J]);            \
                __builtin_amdgcn_raw_buffer_store_b32(as_i(Y), C.comp, C.vcol, (int)soff, NTSC_COMP_STORE_AUX); \
                soff += (unsigned)C.rowbytes;                                                     \
            }
            NTSC_FP_ENC_STEP(0, Y0, I0)
            NTSC_FP_ENC_STEP(1, Y1, I1)
            NTSC_FP_ENC_STEP(2, Y2, F[0])
            NTSC_FP_ENC_STEP(3, Y3, F[1])
            NTSC_FP_ENC_STEP(4, Yn[0], F[2])
            NTSC_FP_ENC_STEP(5, Yn[1], F[3])
            NTSC_FP_ENC_STEP(6, Yn[2], F[4])
            NTSC_FP_ENC_STEP(7, Yn[3], F[5])
            NTSC_FP_ENC_STEP(8, Yn[4], F[6])
            NTSC_FP_ENC_STEP(9, Yn[5], F[7])
            NTSC_FP_ENC_STEP(10, Yn[6], F[8])
            NTSC_FP_ENC_STEP(11, Yn[7], F[9])
            NTSC_FP_ENC_STEP(12, Yn[8], F[10])
            NTSC_FP_ENC_STEP(13, Yn[9], F[11])
            NTSC_FP_ENC_STEP(14, Yn[10], F[12])
            NTSC_FP_ENC_STEP(15, Yn[11], F[13])
#undef NTSC_FP_ENC_STEP
            Y0 = Yn[12]; Y1 = Yn[13]; Y2 = Yn[14]; Y3 = Yn[15];
            I0 = F[14]; I1 = F[15];
            if (more) L.deliver(nq, cur);
        }
        S.rng.pos = sbase;
        S.Yd[0] = Y0; S.Yd[1] = Y1; S.Yd[2] = Y2; S.Yd[3] = Y3;
        S.fI[0] = I0; S.fI[1] = I1;
#pragma unroll
        for (int q = 0; q < 4; q++) { S.Ir[q] = IdT[q]; S.Qr[q] = QdT[q]; }
    }
    for (; t < W + 4; t++) enc_edge(P, S, C, ring, reinterpret_cast<const uint32_t *>(srow), t);
}

// ================================================================================================ decoder
namespace fpipe {

// chroma_from_luma :1497-1567, guarded form (any position): the layout of fastdec::DemodR with float samples
struct DemodR {
    float c0, c1, c2;
    float w0, w1, w2, w3, w4, w5;   // raw chroma at q-5 .. q   (q = t-2)
    float y0, y1, y2, y3, y4;       // box-filtered luma at q-5 .. q-1
    float ieP, qeP, ieN, qeN;
    DEV void init()
    {
        c0 = c1 = c2 = 0;
        w0 = w1 = w2 = w3 = w4 = w5 = 0;
        y0 = y1 = y2 = y3 = y4 = 0;
        ieP = qeP = ieN = qeN = 0;
    }
    DEV void push_edge(float ct, int t, bool hi, int W, int xe, float &Yo, float &Io, float &Qo)
    {
        const float yb = (((c0 + c1) + c2) + ct) * 0.25f;
        const float ch = ct - yb;
        c0 = c1; c1 = c2; c2 = ct;
        w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = ch;
        Yo = y0; y0 = y1; y1 = y2; y2 = y3; y3 = y4; y4 = yb;
        const int x = t - 7;
        float I, Q;
        if (x & 1) {
            const int xiv = hi ? 2 : 0;
            const bool m = (x + 1 + xiv + 1) < W;                        // :1550
            const float a = hi ? w3 : w1, b = hi ? w4 : w2;
            const bool pos = (x & 3) == 1;                               // flipped :1539-1542, then negated :1550-1553
            ieN = m ? (pos ? a : -a) : 0.0f;
            qeN = m ? (pos ? b : -b) : 0.0f;
            I = (ieP + ieN) * 0.5f;
            Q = (qeP + qeN) * 0.5f;
        } else {
            I = ieN; Q = qeN;
            ieP = ieN; qeP = qeN;
        }
        if (x >= xe) { I = 0; Q = 0; }                                   // :1553-1556, :1562-1565
        Io = I; Qo = Q;
    }
};

// the same separator in the form the steady loop runs (fastdec::DemodS): pair-sum box, chroma window split by parity
struct DemodS {
    float c1, pA, pB;
    float e1, e2, o1, o2;
    float y0, y1, y2, y3, y4;
    float ieP, qeP, ieN, qeN;
    DEV void from(const DemodR &D, bool pick_next)
    {
        c1 = D.c2; pA = D.c2 + D.c1; pB = D.c1 + D.c0;
        if (pick_next) { e1 = D.w4; e2 = D.w2; o1 = D.w5; o2 = D.w3; }
        else           { e1 = D.w5; e2 = D.w3; o1 = D.w4; o2 = D.w2; }
        y0 = D.y0; y1 = D.y1; y2 = D.y2; y3 = D.y3; y4 = D.y4;
        ieP = D.ieP; qeP = D.qeP; ieN = D.ieN; qeN = D.qeN;
    }
    DEV void to(DemodR &D, bool pick_next, float cm1, float cm2) const
    {
        D.c2 = c1; D.c1 = cm1; D.c0 = cm2;          // (the last three samples are handed over as such: no float subtraction of sums)
        D.w0 = 0; D.w1 = 0;
        if (pick_next) { D.w4 = e1; D.w2 = e2; D.w5 = o1; D.w3 = o2; }
        else           { D.w5 = e1; D.w3 = e2; D.w4 = o1; D.w2 = o2; }
        D.y0 = y0; D.y1 = y1; D.y2 = y2; D.y3 = y3; D.y4 = y4;
        D.ieP = ieP; D.qeP = qeP; D.ieN = ieN; D.qeN = qeN;
    }
    // PICK = x is odd; NEG = the picked pair is negated (x = 3 mod 4); hi = lane mask of xi == 2; dmf = the dropout
    // factor (0 / 1) on the picked pair where MASK
    template <bool PICK, bool NEG, bool LUMA, bool MASK>
    DEV void push(float ct, bool hi, float dmf, float &Yo, float &Io, float &Qo)
    {
        const float p = ct + c1;
        const float yb = (p + pB) * 0.25f;
        const float ch = ct - yb;
        c1 = ct; pB = pA; pA = p;
        if (LUMA) { Yo = y0; y0 = y1; y1 = y2; y2 = y3; y3 = y4; y4 = yb; }
        if (PICK) {
            float a = hi ? e1 : e2, b = hi ? o1 : o2;
            if (NEG) { a = -a; b = -b; }
            e2 = e1; e1 = ch;
            if (MASK) { a *= dmf; b *= dmf; }
            ieN = a; qeN = b;
            Io = (ieP + ieN) * 0.5f;
            Qo = (qeP + qeN) * 0.5f;
        } else {
            o2 = o1; o1 = ch;
            Io = ieN; Qo = qeN;
            ieP = ieN; qeP = qeN;
        }
    }
};

template <bool VHS>
struct State {
    DemodR D1, D2;
    float l0, l1, l2;                         // luma stream window (VHS)
    Casc3 vl, vcU, vcV, sh, oU, oV;
    PoleHp vpre;
    float Yprev, Uraw, Vraw;
    LaneRand32 rng;
    int nU, nV;
};

struct Const {
    unsigned xi;
    bool hi;
    int W, xe, lane;
    int d, SKT, LOFF;
    float sg0, sg2;           // sign of the re-modulated chroma at x2 = 0, 1 / 2, 3 (mod 4)
    float bA, bS;             // vertical blend: weight of the row above (0 / 1), scale (1 / 0.5)
    float dmf;                // dropout factor (0 = this row's chroma is dropped)
    float cosv, sinv;
    float a_vc, a_vl, a_sh, a_tv, sharp2;
    float ybias;              // NTSC_FP_LUMA_BIAS_* / 256
    int *tailU;
    size_t rstride;
    __amdgpu_buffer_rsrc_t comp;
    int vbase, rowbytes;
};

template <int AUX = 0>
DEV float cs_load(const Const &C, int x)
{
    const unsigned off = (unsigned)C.vbase + (unsigned)x * (unsigned)C.rowbytes;
    return as_f(__builtin_amdgcn_raw_buffer_load_b32(C.comp, (int)off, 0, AUX));
}

// YIQ_to_RGB :1385-1396 on the 256-scaled signal: (int)(x / 256) clamped to 0..255; alpha 0 (:1914)
DEV uint32_t yiq_to_bgra(const Const &C, float Yo, float fU, float fV)
{
    const float y = __builtin_fmaf(Yo, 1.0f / 256.0f, C.ybias);
    const float rf = __builtin_fmaf(0.621f / 256.0f, fV, __builtin_fmaf(0.956f / 256.0f, fU, y));
    const float gf = __builtin_fmaf(-0.647f / 256.0f, fV, __builtin_fmaf(-0.272f / 256.0f, fU, y));
    const float bf = __builtin_fmaf(1.703f / 256.0f, fV, __builtin_fmaf(-1.106f / 256.0f, fU, y));
    // v_cvt_u32_f32 truncates and saturates (negative -> 0): the lower clamp is the conversion's own
    const uint32_t r = umin32(cvt_sat_u32(rf), 255u);
    const uint32_t g = umin32(cvt_sat_u32(gf), 255u);
    const uint32_t b = umin32(cvt_sat_u32(bf), 255u);
    return ((r << 16) | b) | (g << 8);
}

struct Steady {
    DemodS D1, D2;
    float lc1, lpA, lpB;
    uint32_t *rb;
    bool rb0;
};

// The VCR half of a steady step: first separator at x1 = t - 7 = DPH + J (mod 4), chroma noise, phase noise, VHS
// chroma / luma filters, vertical blend, re-modulation; returns the VCR's composite sample at x2 = x1 - d
template <int DPH, int J>
DEV float vcr_step(const DevParams &P, State<true> &S, Steady &T, const Const &C, float pc, float pl)
{
    constexpr bool pick1 = ((DPH + J) & 1) != 0;
    constexpr bool neg1 = ((DPH + J) & 3) == 3;
    float Yd, U, V;
    T.D1.template push<pick1, neg1, false, false>(pc, C.hi, 1.0f, Yd, U, V);
    // chroma noise :1719-1735
    U += (float)S.nU; V += (float)S.nV;
    S.nU = sdiv2(S.nU + (int)umod31(S.rng.template draw<2 * J>(T.rb, T.rb0), P.m_cnoise) - P.cnoise_k);
    S.nV = sdiv2(S.nV + (int)umod31(S.rng.template draw<2 * J + 1>(T.rb, T.rb0), P.m_cnoise) - P.cnoise_k);
    // chroma phase noise :1748-1762
    const float Ud = __builtin_fmaf(U, C.cosv, -(V * C.sinv));
    const float Vd = __builtin_fmaf(U, C.sinv, V * C.cosv);
    // VHS chroma low-pass :1814-1836 (value for input x1 lands at x2 = x1 - d)
    const float fU = S.vcU.push(Ud, C.a_vc);
    const float fV = S.vcV.push(Vd, C.a_vc);
    // luma at x2: box -> low-pass + emphasis :1793-1812 -> sharpen :1866-1883
    const float lp = pl + T.lc1;
    const float yb = (lp + T.lpB) * 0.25f;
    T.lc1 = pl; T.lpB = T.lpA; T.lpA = lp;
    float s = S.vl.push(yb, C.a_vl);
    s = __builtin_fmaf(S.vpre.hp(s, C.a_vl), 1.6f, s);
    const float ts = S.sh.push(s, C.a_sh);
    const float Y = __builtin_fmaf(s - ts, C.sharp2, s);
    // vertical chroma blend :1843-1863: (above + cur + 1) >> 1, above = 0 for the field's second row, untouched for
    // its first row / blend off
    U = __builtin_fmaf(wave_up_f(fU), C.bA, fU) * C.bS;
    V = __builtin_fmaf(wave_up_f(fV), C.bA, fV) * C.bS;
    // composite out of the VCR :1885-1888: modulate at x2 = J (mod 4) (amplitude 50: (v*50)/50 == v)
    const float chroma = (J & 1) ? V : U;
    return __builtin_fmaf(chroma, (J & 2) ? C.sg2 : C.sg0, Y);
}

// One steady step at unrolled position J (t = SKT + 4n + J):
//   second separator / output   x3 = t - 14 - d = 4n + J + 1
//   re-modulation               x2 = x3 + 7     = 4n + J (mod 4)
//   first separator             x1 = t - 7      = DPH + J (mod 4)
// Non-VHS form: x3 = x1 = t - 7 = 4n + J + 1 (SKT = 8), one separator.
template <bool VHS, int DPH, int J>
DEV uint32_t step(const DevParams &P, State<VHS> &S, Steady &T, const Const &C, float pc, float pl)
{
    float Y, U, V;
    constexpr bool pick3 = ((J + 1) & 1) != 0;
    constexpr bool neg3 = ((J + 1) & 3) == 3;
    if constexpr (!VHS) {
        T.D1.template push<pick3, neg3, true, true>(pc, C.hi, C.dmf, Y, U, V);
    } else {
        const float c2 = vcr_step<DPH, J>(P, S, T, C, pc, pl);
        T.D2.template push<pick3, neg3, true, true>(c2, C.hi, C.dmf, Y, U, V);      // dropout :1891-1901 on the picked pair
    }
    // composite_lowpass_tv :1399-1427 (delay 1) and YIQ -> RGB for the previous position
    const float fUd = S.oU.push(U, C.a_tv);
    const float fVd = S.oV.push(V, C.a_tv);
    const float Yo = S.Yprev;
    S.Yprev = Y;
    return yiq_to_bgra(C, Yo, fUd, fVd);
}

// The VCR half of a guarded step (any position): the VCR's composite sample at x2 = t - 7 - d (0 outside the row)
DEV float vcr_edge(const DevParams &P, State<true> &S, const Const &C, uint32_t *ring, int t)
{
    const int W = C.W;
    const float pc = t < W ? cs_load(C, t) : 0.0f;
    float Y, U, V;
    S.D1.push_edge(pc, t, C.hi, W, C.xe, Y, U, V);
    const int x1 = t - 7;
    const bool in1 = x1 >= 0 && x1 < W;
    float fU = 0, fV = 0;
    if (in1) {
        U += (float)S.nU; V += (float)S.nV;
        S.nU = sdiv2(S.nU + (int)umod31(S.rng.next(ring, C.lane), P.m_cnoise) - P.cnoise_k);
        S.nV = sdiv2(S.nV + (int)umod31(S.rng.next(ring, C.lane), P.m_cnoise) - P.cnoise_k);
        const float Ud = __builtin_fmaf(U, C.cosv, -(V * C.sinv));
        const float Vd = __builtin_fmaf(U, C.sinv, V * C.cosv);
        fU = S.vcU.push(Ud, C.a_vc);
        fV = S.vcV.push(Vd, C.a_vc);
        if (x1 >= W - C.d) {                  // raw tail of the chroma low-pass :1830
            C.tailU[(size_t)(x1 & 15) * C.rstride] = as_i(Ud);
            C.tailU[(size_t)(16 + (x1 & 15)) * C.rstride] = as_i(Vd);
        }
    }
    const int x2 = x1 - C.d;
    const int xl = t - C.LOFF;
    const float pl = (xl >= 0 && xl < W) ? cs_load(C, xl) : 0.0f;
    const float yb = (((S.l0 + S.l1) + S.l2) + pl) * 0.25f;
    S.l0 = S.l1; S.l1 = S.l2; S.l2 = pl;
    const bool in2 = x2 >= 0 && x2 < W;
    Y = 0;
    if (in2) {
        if (x2 >= W - C.d) {
            fU = as_f(C.tailU[(size_t)(x2 & 15) * C.rstride]);
            fV = as_f(C.tailU[(size_t)(16 + (x2 & 15)) * C.rstride]);
        }
        float s = S.vl.push(yb, C.a_vl);
        s = __builtin_fmaf(S.vpre.hp(s, C.a_vl), 1.6f, s);
        const float ts = S.sh.push(s, C.a_sh);
        Y = __builtin_fmaf(s - ts, C.sharp2, s);
    }
    U = __builtin_fmaf(wave_up_f(fU), C.bA, fU) * C.bS;
    V = __builtin_fmaf(wave_up_f(fV), C.bA, fV) * C.bS;
    float c2 = 0;
    if (in2) {
        const unsigned s = (C.xi + (unsigned)x2) & 3u;
        float chroma = (s & 1u) ? V : U;
        if (s & 2u) chroma = -chroma;
        c2 = Y + chroma;
    }
    return c2;
}

template <bool VHS>
DEV bool edge_step(const DevParams &P, State<VHS> &S, const Const &C, uint32_t *ring, int t, uint32_t &px, int &xo_out)
{
    const int W = C.W;
    float Y, U, V;
    int x3 = t - 7;
    if constexpr (VHS) {
        const float c2 = vcr_edge(P, S, C, ring, t);
        const int x2 = t - 7 - C.d;
        S.D2.push_edge(c2, x2, C.hi, W, C.xe, Y, U, V);
        x3 = x2 - 7;
    } else {
        const float pc = t < W ? cs_load(C, t) : 0.0f;
        S.D1.push_edge(pc, t, C.hi, W, C.xe, Y, U, V);
    }
    if (x3 < 0 || x3 > W) return false;
    const bool in3 = x3 < W;
    if (!in3) { U = 0; V = 0; Y = 0; }
    U *= C.dmf; V *= C.dmf;
    float fUd = 0, fVd = 0;
    if (in3) {
        fUd = S.oU.push(U, C.a_tv);
        fVd = S.oV.push(V, C.a_tv);
    }
    const int xo = x3 - 1;
    const float Yo = S.Yprev;
    const float Ur = S.Uraw, Vr = S.Vraw;
    S.Yprev = Y; S.Uraw = U; S.Vraw = V;
    if (xo < 0) return false;
    if (xo >= W - 1) { fUd = Ur; fVd = Vr; }      // last sample keeps its input :1419-1424
    px = yiq_to_bgra(C, Yo, fUd, fVd);
    xo_out = xo;
    return true;
}

// Steady-state loop: every stage strictly inside the row; 4 pixels per iteration.  Composite samples are requested TWO
// iterations ahead (the float steps are short: one iteration no longer covers an L2 / HBM round trip -- with the exact
// kernels' reload-after-use the waves sat 37 % of their cycles in s_waitcnt vmcnt(0) at the top of the loop).
template <bool VHS, int DPH, int VAR>
DEV int steady(const DevParams &P, State<VHS> &S, const Const &C, uint32_t *ring, uint32_t *ostage,
               const unsigned long long *orow, int t)
{
    const int t_end = C.W - (C.d > 7 ? C.d - 7 : 0);
    const int SKT = C.SKT, LOFF = C.LOFF, lane = C.lane;
    if (t + 4 > t_end) return t;
    if (VHS && (S.rng.pos & 7)) return t;
    Steady T;
    constexpr bool pick3_next = true;          // x3 = 1 (mod 4) at J = 0
    T.D1.from(S.D1, VHS ? (DPH & 1) != 0 : pick3_next);
    T.D2.from(S.D2, pick3_next);
    T.lc1 = S.l2; T.lpA = S.l2 + S.l1; T.lpB = S.l1 + S.l0;
    // (the separator in front of the TV stages carries the dropout factor on everything it has picked: the guarded
    //  steps apply it to their outputs instead, so what they left behind is scaled here)
    DemodS &Dout = VHS ? T.D2 : T.D1;
    Dout.ieP *= C.dmf; Dout.qeP *= C.dmf; Dout.ieN *= C.dmf; Dout.qeN *= C.dmf;
    int sbase = VHS ? S.rng.pos : 0;
    float pc[4], pl[4], nc[4], nl[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        pc[j] = cs_load(C, t + j); pl[j] = VHS ? cs_load<NTSC_COMP_LOAD2_AUX>(C, t + j - LOFF) : 0.0f;
        nc[j] = cs_load(C, t + 4 + j); nl[j] = VHS ? cs_load<NTSC_COMP_LOAD2_AUX>(C, t + 4 + j - LOFF) : 0.0f;
    }
    int pend_x = -1;
    // cooperative stores: burst k covers rows 16k .. 16k+15, four lanes per row, so that each row's 64 bytes leave as one
    // contiguous request; the eight LDS reads of a flush are issued together (one wait), rows without an output (halo,
    // padding) are skipped by the store's own predicate
#define NTSC_FP_FLUSH()                                                                           \
    if (pend_x >= 0) {                                                                            \
        unsigned long long rp[4];                                                                 \
        uint4 fv[4];                                                                              \
        _Pragma("unroll")                                                                         \
        for (int k = 0; k < 4; k++) {                                                             \
            const int r = 16 * k + (lane >> 2);                                                   \
            rp[k] = orow[r];                                                                      \
            fv[k] = *reinterpret_cast<const uint4 *>(&ostage[r * 20 + (lane & 3) * 4]);           \
        }                                                                                         \
        _Pragma("unroll")                                                                         \
        for (int k = 0; k < 4; k++)                                                               \
            if (rp[k]) __builtin_nontemporal_store(to_v4u(fv[k]), (g_v4u_ptr)(rp[k] + 4ull * (unsigned)(pend_x + (lane & 3) * 4))); \
        pend_x = -1;                                                                              \
    }
    for (; t + 4 <= t_end; t += 4) {
        uint32_t o[4];
        float fc[4], fl[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { fc[j] = cs_load(C, t + 8 + j); fl[j] = VHS ? cs_load<NTSC_COMP_LOAD2_AUX>(C, t + 8 + j - LOFF) : 0.0f; }
        T.rb = ring + sbase * 64 + lane;
        T.rb0 = sbase == 0;
        sbase = (sbase + 8) & 31;
#define NTSC_FP_STEP(J)                                                                           \
        o[J] = step<VHS, DPH, J>(P, S, T, C, pc[J], pl[J]);                                       \
        if (!(VAR & 1)) __builtin_amdgcn_sched_barrier(0);
        NTSC_FP_STEP(0)
        NTSC_FP_STEP(1)
        NTSC_FP_STEP(2)
        NTSC_FP_STEP(3)
#undef NTSC_FP_STEP
#pragma unroll
        for (int j = 0; j < 4; j++) { pc[j] = nc[j]; pl[j] = nl[j]; nc[j] = fc[j]; nl[j] = fl[j]; }
        const int xo0 = t - SKT;                   // multiple of 4
        const int sub = (xo0 >> 2) & 3;
        *reinterpret_cast<uint4 *>(&ostage[lane * 20 + sub * 4]) = make_uint4(o[0], o[1], o[2], o[3]);
        if (sub == 3) pend_x = xo0 - 12;
        NTSC_FP_FLUSH()
    }
    NTSC_FP_FLUSH()
#undef NTSC_FP_FLUSH
    // back to the guarded steps' layout: a separator's last three inputs out of its pair sums (values of ~2^17 in a
    // 24-bit mantissa: the subtraction is off by at most 2^-6 of 1/256 of an output step)
    {
        const float d1c1 = T.D1.pA - T.D1.c1, d2c1 = T.D2.pA - T.D2.c1, lc2 = T.lpA - T.lc1;
        T.D1.to(S.D1, VHS ? (DPH & 1) != 0 : pick3_next, d1c1, T.D1.pB - d1c1);
        T.D2.to(S.D2, pick3_next, d2c1, T.D2.pB - d2c1);
        S.l2 = T.lc1; S.l1 = lc2; S.l0 = T.lpB - lc2;
    }
    S.Uraw = 0; S.Vraw = 0;
    if (VHS) S.rng.pos = sbase;
    return t;
}

} // namespace fpipe

template <bool VHS, int VAR = 0, int WAVES = 2>
__global__ __launch_bounds__(64, VHS ? WAVES : 4) void k_decode_fp(DevParams P, GeomDev G, const FieldDev *__restrict__ fields,
                                                                const int *__restrict__ comp,
                                                                const uint32_t *__restrict__ rs_chroma,
                                                                const int *__restrict__ n0_u, const int *__restrict__ n0_v,
                                                                const int *__restrict__ hs_shift,
                                                                const int *__restrict__ pn_noise,
                                                                const int *__restrict__ dropout, int *__restrict__ tails)
{
    using namespace fpipe;
    __shared__ uint32_t ring[33 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];
    __shared__ unsigned long long orow[64];

    const int lane = threadIdx.x;
    const int gidx = blockIdx.x * 63 + lane - 1;          // lane 0 = halo (row above)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok;
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const int W = P.W;
    uint32_t *drow = reinterpret_cast<uint32_t *>(fd.dst + (size_t)fd.dst_ls * y);
    orow[lane] = is_out ? (unsigned long long)drow : 0ull;
    const size_t tcol = (size_t)blockIdx.x * 64 + lane;
    const size_t tstride = (size_t)gridDim.x * 64;

    Const C;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.hi = (C.xi & 2u) != 0;
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = lane;
    C.d = VHS ? P.cdelay : 0;
    C.SKT = VHS ? 15 + C.d : 8;
    C.LOFF = 5 + C.d;
    C.sg0 = C.hi ? -1.0f : 1.0f;
    C.sg2 = -C.sg0;
    const bool vb = VHS && P.vblend && P.ntsc;
    C.bA = (vb && k >= 2) ? 1.0f : 0.0f;
    C.bS = (vb && k >= 1) ? 0.5f : 1.0f;
    C.dmf = (P.loss && dropout[rc] != 0) ? 0.0f : 1.0f;
    C.cosv = 1; C.sinv = 0;
    if (VHS) {
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (float)G.ptab[2 * n]; C.sinv = (float)G.ptab[2 * n + 1];
    }
    C.a_vc = (float)P.a_vc; C.a_vl = (float)P.a_vl; C.a_sh = (float)P.a_sh; C.a_tv = (float)P.a_tv;
    C.sharp2 = (float)(P.sharpen * 2);
    // the blend's (a + b + 1) >> 1 rounds half up where (a + b) / 2 does not: +1/4 on I and Q -- not compensated (it
    // would take two more constants per channel for a quarter of 1/256 of a step); luma: one constant
    C.ybias = (VHS ? NTSC_FP_LUMA_BIAS_VHS : NTSC_FP_LUMA_BIAS_DEF) * (1.0f / 256.0f);
    C.tailU = tails + tcol;
    C.rstride = tstride;
    C.rowbytes = P.Rpad * 4;
    const int hs = P.hs ? hs_shift[rc] : 0;
    C.vbase = (int)((unsigned)rc * 4u + (unsigned)hs * (unsigned)C.rowbytes);
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp), 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    State<VHS> S;
    S.D1.init(); S.D2.init();
    S.l0 = S.l1 = S.l2 = 0;
    S.vl.reset(16); S.vpre.reset(16); S.vcU.reset(0); S.vcV.reset(0);
    S.sh.reset(0);
    S.oU.reset(0); S.oV.reset(0);
    S.Yprev = S.Uraw = S.Vraw = 0;
    S.nU = S.nV = 0;
    if (VHS) {
        const int fill_draws = 2 * (C.SKT - 7);
        S.rng.init(ring, rs_chroma + rc, P.Rpad, lane, (-(31 + fill_draws)) & 7);
        S.nU = n0_u[rc]; S.nV = n0_v[rc];
    }

    const int SKT = C.SKT;
    const int total = W + SKT;
    int t = 0;
    for (; t < SKT && t < total; t++) {
        uint32_t px; int xo;
        (void)edge_step<VHS>(P, S, C, ring, t, px, xo);
    }
    if constexpr (!VHS) t = steady<VHS, 0, VAR>(P, S, C, ring, ostage, orow, t);
    else switch ((C.SKT - 7) & 3) {
        case 0: t = steady<VHS, 0, VAR>(P, S, C, ring, ostage, orow, t); break;
        case 1: t = steady<VHS, 1, VAR>(P, S, C, ring, ostage, orow, t); break;
        case 2: t = steady<VHS, 2, VAR>(P, S, C, ring, ostage, orow, t); break;
        default: t = steady<VHS, 3, VAR>(P, S, C, ring, ostage, orow, t); break;
    }
    for (; t < total; t++) {
        uint32_t px; int xo;
        if (!edge_step<VHS>(P, S, C, ring, t, px, xo)) continue;
        ostage[lane * 20 + (xo & 15)] = px;
        if ((xo & 15) == 15) {
            if (is_out) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(&ostage[lane * 20]);
                g_v4u_ptr dp = (g_v4u_ptr)(drow + (xo - 15));
                const uint4 a = sp[0], b = sp[1], c4 = sp[2], d4 = sp[3];
                dp[0] = to_v4u(a); dp[1] = to_v4u(b); dp[2] = to_v4u(c4); dp[3] = to_v4u(d4);
            }
        } else if (xo == W - 1 && is_out) {
            const int xb = xo & ~15;
            for (int q = xb; q <= xo; q++) ((g_u32_ptr)drow)[q] = ostage[lane * 20 + (q - xb)];
        }
    }
}

// ================================================================================================ k_decode_fp2
// The -vhs decoder as a TWO-ROLE workgroup: wave 0 runs the VCR half of its 63 rows (+ halo), wave 1 the TV half of the
// same rows, and the VCR's composite output travels through an LDS ring instead of living in one wave's registers.
// Why: the one-wave form needs ~196 VGPRs (two waves per SIMD) and its waves spend most of their cycles stalled on their
// own dependency chains (profiles/r06_float_pmc.txt); the two roles need ~half the registers each, so twice as many
// waves fit a SIMD, and a field's rows are worked on by two instruction streams instead of one.
//   ring slot of sample x2: (x2 & (NSLOT - 1)), one column per lane;
//   sync[0] = samples published by the VCR role, sync[1] = samples consumed by the TV role (both monotonic);
//   the VCR role publishes after every group of samples behind a workgroup-scope release (LDS operations of one wave
//   are performed in order; the fence keeps the compiler from moving the counter store in front of the samples) and
//   waits while the ring is full; the TV role waits until what it is about to read has been published.
namespace fpipe {

constexpr int NSLOT = 32;

DEV int lds_peek(const volatile uint32_t *p) { return __builtin_amdgcn_readfirstlane((int)*p); }
DEV void lds_wait_ge(const volatile uint32_t *p, int v, int &cached)
{
    if (cached >= v) return;
    int seen = lds_peek(p);
    while (seen < v) { __builtin_amdgcn_s_sleep(2); seen = lds_peek(p); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    cached = seen;
}
DEV void lds_publish(volatile uint32_t *p, int v)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *p = (uint32_t)v;
}

// ---- wave 0: the VCR half (vcr_edge / vcr_step as in the one-wave form), output x2 = t - (7 + d) into the ring
template <int DPH>
DEV int vcr_steady(const DevParams &P, State<true> &S, const Const &C, uint32_t *ring, uint32_t *c2col, volatile uint32_t *sync,
                   int &cons_seen, int t, int SK1)
{
    const int t_end = C.W - (C.d > 7 ? C.d - 7 : 0);
    const int LOFF = C.LOFF, lane = C.lane;
    if (t + 4 > t_end || (S.rng.pos & 7)) return t;
    Steady T;
    T.D1.from(S.D1, (DPH & 1) != 0);
    T.D2.from(S.D2, true);          // (unused by this role)
    T.lc1 = S.l2; T.lpA = S.l2 + S.l1; T.lpB = S.l1 + S.l0;
    int sbase = S.rng.pos;
    float pc[4], pl[4], nc[4], nl[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        pc[j] = cs_load(C, t + j); pl[j] = cs_load<NTSC_COMP_LOAD2_AUX>(C, t + j - LOFF);
        nc[j] = cs_load(C, t + 4 + j); nl[j] = cs_load<NTSC_COMP_LOAD2_AUX>(C, t + 4 + j - LOFF);
    }
    for (; t + 4 <= t_end; t += 4) {
        float fc[4], fl[4], c2[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { fc[j] = cs_load(C, t + 8 + j); fl[j] = cs_load<NTSC_COMP_LOAD2_AUX>(C, t + 8 + j - LOFF); }
        T.rb = ring + sbase * 64 + lane;
        T.rb0 = sbase == 0;
        sbase = (sbase + 8) & 31;
        c2[0] = vcr_step<DPH, 0>(P, S, T, C, pc[0], pl[0]); __builtin_amdgcn_sched_barrier(0);
        c2[1] = vcr_step<DPH, 1>(P, S, T, C, pc[1], pl[1]); __builtin_amdgcn_sched_barrier(0);
        c2[2] = vcr_step<DPH, 2>(P, S, T, C, pc[2], pl[2]); __builtin_amdgcn_sched_barrier(0);
        c2[3] = vcr_step<DPH, 3>(P, S, T, C, pc[3], pl[3]); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) { pc[j] = nc[j]; pl[j] = nl[j]; nc[j] = fc[j]; nl[j] = fl[j]; }
        const int x2 = t - SK1;                    // multiple of 4 (the steady loop starts on x2 = 0 (mod 4), see the kernel)
        lds_wait_ge(sync + 1, x2 + 4 - NSLOT, cons_seen);
        const int sl = x2 & (NSLOT - 1);
#pragma unroll
        for (int j = 0; j < 4; j++) c2col[(sl + j) * 64] = (uint32_t)as_i(c2[j]);
        lds_publish(sync, x2 + 4);
    }
    {
        const float d1c1 = T.D1.pA - T.D1.c1, lc2 = T.lpA - T.lc1;
        T.D1.to(S.D1, (DPH & 1) != 0, d1c1, T.D1.pB - d1c1);
        S.l2 = T.lc1; S.l1 = lc2; S.l0 = T.lpB - lc2;
    }
    S.rng.pos = sbase;
    return t;
}

DEV void vcr_role(const DevParams &P, State<true> &S, const Const &C, uint32_t *ring, uint32_t *c2ring, volatile uint32_t *sync)
{
    const int W = C.W, SK1 = 7 + C.d, total = W + SK1;
    uint32_t *c2col = c2ring + C.lane;
    int cons_seen = 0;
    int t = 0;
    for (; t < SK1 && t < total; t++) (void)vcr_edge(P, S, C, ring, t);          // pipeline fill: x2 < 0
    switch (C.d & 3) {
        case 0: t = vcr_steady<0>(P, S, C, ring, c2col, sync, cons_seen, t, SK1); break;
        case 1: t = vcr_steady<1>(P, S, C, ring, c2col, sync, cons_seen, t, SK1); break;
        case 2: t = vcr_steady<2>(P, S, C, ring, c2col, sync, cons_seen, t, SK1); break;
        default: t = vcr_steady<3>(P, S, C, ring, c2col, sync, cons_seen, t, SK1); break;
    }
    for (; t < total; t++) {
        const float c2 = vcr_edge(P, S, C, ring, t);
        const int x2 = t - SK1;
        lds_wait_ge(sync + 1, x2 + 1 - NSLOT, cons_seen);
        c2col[(x2 & (NSLOT - 1)) * 64] = (uint32_t)as_i(c2);
        lds_publish(sync, x2 + 1);
    }
}

// ---- wave 1: the TV half = the one-separator decoder on the samples of the ring (position u = the VCR's x2)
DEV void tv_role(const DevParams &P, State<false> &S, const Const &C, uint32_t *c2ring, volatile uint32_t *sync, uint32_t *ostage,
                 const unsigned long long *orow, uint32_t *drow, bool is_out)
{
    const int W = C.W, lane = C.lane, SKT = 8, total = W + SKT;
    const uint32_t *c2col = c2ring + lane;
    int prod_seen = 0;
    uint32_t *nouse = nullptr;
    int u = 0;
    auto guarded = [&](int uu, uint32_t &px, int &xo) -> bool {
        float Y, U, V;
        float pc = 0.0f;
        if (uu < W) {
            lds_wait_ge(sync, uu + 1, prod_seen);
            pc = as_f((int)c2col[(uu & (NSLOT - 1)) * 64]);
            lds_publish(sync + 1, uu + 1);
        }
        S.D1.push_edge(pc, uu, C.hi, W, C.xe, Y, U, V);
        const int x3 = uu - 7;
        if (x3 < 0 || x3 > W) return false;
        const bool in3 = x3 < W;
        if (!in3) { U = 0; V = 0; Y = 0; }
        U *= C.dmf; V *= C.dmf;
        float fUd = 0, fVd = 0;
        if (in3) { fUd = S.oU.push(U, C.a_tv); fVd = S.oV.push(V, C.a_tv); }
        xo = x3 - 1;
        const float Yo = S.Yprev, Ur = S.Uraw, Vr = S.Vraw;
        S.Yprev = Y; S.Uraw = U; S.Vraw = V;
        if (xo < 0) return false;
        if (xo >= W - 1) { fUd = Ur; fVd = Vr; }
        px = yiq_to_bgra(C, Yo, fUd, fVd);
        return true;
    };
    (void)nouse;
    for (; u < SKT && u < total; u++) { uint32_t px; int xo; (void)guarded(u, px, xo); }
    // steady: every sample inside the row, 4 per iteration, the next iteration's samples read one iteration ahead
    if (u + 4 <= W) {
        Steady T;
        T.D1.from(S.D1, true);
        T.D2.from(S.D2, true);
        T.D1.ieP *= C.dmf; T.D1.qeP *= C.dmf; T.D1.ieN *= C.dmf; T.D1.qeN *= C.dmf;
        float pc[4], nc[4];
        lds_wait_ge(sync, u + 4, prod_seen);
#pragma unroll
        for (int j = 0; j < 4; j++) pc[j] = as_f((int)c2col[((u + j) & (NSLOT - 1)) * 64]);
        int pend_x = -1;
        for (; u + 4 <= W; u += 4) {
            const bool more = u + 8 <= W;
            if (more) {
                lds_wait_ge(sync, u + 8, prod_seen);
#pragma unroll
                for (int j = 0; j < 4; j++) nc[j] = as_f((int)c2col[((u + 4 + j) & (NSLOT - 1)) * 64]);
            }
            // (everything up to u + 4, or u + 8, is in registers: the VCR role may reuse those slots)
            lds_publish(sync + 1, more ? u + 8 : u + 4);
            uint32_t o[4];
            State<false> &Sx = S;
            o[0] = step<false, 0, 0>(P, Sx, T, C, pc[0], 0.0f);
            o[1] = step<false, 0, 1>(P, Sx, T, C, pc[1], 0.0f);
            o[2] = step<false, 0, 2>(P, Sx, T, C, pc[2], 0.0f);
            o[3] = step<false, 0, 3>(P, Sx, T, C, pc[3], 0.0f);
#pragma unroll
            for (int j = 0; j < 4; j++) pc[j] = nc[j];
            const int xo0 = u - SKT;
            const int sub = (xo0 >> 2) & 3;
            *reinterpret_cast<uint4 *>(&ostage[lane * 20 + sub * 4]) = make_uint4(o[0], o[1], o[2], o[3]);
            if (sub == 3) {
                pend_x = xo0 - 12;
                unsigned long long rp[4];
                uint4 fv[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int r = 16 * k + (lane >> 2);
                    rp[k] = orow[r];
                    fv[k] = *reinterpret_cast<const uint4 *>(&ostage[r * 20 + (lane & 3) * 4]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (rp[k]) __builtin_nontemporal_store(to_v4u(fv[k]), (g_v4u_ptr)(rp[k] + 4ull * (unsigned)(pend_x + (lane & 3) * 4)));
            }
        }
        const float d1c1 = T.D1.pA - T.D1.c1;
        T.D1.to(S.D1, true, d1c1, T.D1.pB - d1c1);
        S.Uraw = 0; S.Vraw = 0;
    }
    for (; u < total; u++) {
        uint32_t px; int xo;
        if (!guarded(u, px, xo)) continue;
        ostage[lane * 20 + (xo & 15)] = px;
        if ((xo & 15) == 15) {
            if (is_out) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(&ostage[lane * 20]);
                g_v4u_ptr dp = (g_v4u_ptr)(drow + (xo - 15));
                const uint4 a = sp[0], b = sp[1], c4 = sp[2], d4 = sp[3];
                dp[0] = to_v4u(a); dp[1] = to_v4u(b); dp[2] = to_v4u(c4); dp[3] = to_v4u(d4);
            }
        } else if (xo == W - 1 && is_out) {
            const int xb = xo & ~15;
            for (int q = xb; q <= xo; q++) ((g_u32_ptr)drow)[q] = ostage[lane * 20 + (q - xb)];
        }
    }
}

} // namespace fpipe

template <int WAVES>
__global__ __launch_bounds__(128, WAVES) void k_decode_fp2(DevParams P, GeomDev G, const FieldDev *__restrict__ fields,
                                                           const int *__restrict__ comp, const uint32_t *__restrict__ rs_chroma,
                                                           const int *__restrict__ n0_u, const int *__restrict__ n0_v,
                                                           const int *__restrict__ hs_shift, const int *__restrict__ pn_noise,
                                                           const int *__restrict__ dropout, int *__restrict__ tails)
{
    using namespace fpipe;
    __shared__ uint32_t ring[33 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];
    __shared__ unsigned long long orow[64];
    __shared__ uint32_t c2ring[NSLOT * 64];
    __shared__ uint32_t sync[2];

    const int role = threadIdx.x >> 6;            // 0: VCR half, 1: TV half
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 63 + lane - 1;          // lane 0 = halo (row above)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok;
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const int W = P.W;
    uint32_t *drow = reinterpret_cast<uint32_t *>(fd.dst + (size_t)fd.dst_ls * y);
    if (role == 1) orow[lane] = is_out ? (unsigned long long)drow : 0ull;
    if (threadIdx.x < 2) sync[threadIdx.x] = 0u;
    __syncthreads();

    Const C;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.hi = (C.xi & 2u) != 0;
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = lane;
    C.d = P.cdelay;
    C.SKT = 15 + C.d;
    C.LOFF = 5 + C.d;
    C.sg0 = C.hi ? -1.0f : 1.0f;
    C.sg2 = -C.sg0;
    const bool vb = P.vblend && P.ntsc;
    C.bA = (vb && k >= 2) ? 1.0f : 0.0f;
    C.bS = (vb && k >= 1) ? 0.5f : 1.0f;
    C.dmf = (P.loss && dropout[rc] != 0) ? 0.0f : 1.0f;
    {
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (float)G.ptab[2 * n]; C.sinv = (float)G.ptab[2 * n + 1];
    }
    C.a_vc = (float)P.a_vc; C.a_vl = (float)P.a_vl; C.a_sh = (float)P.a_sh; C.a_tv = (float)P.a_tv;
    C.sharp2 = (float)(P.sharpen * 2);
    C.ybias = NTSC_FP_LUMA_BIAS_VHS * (1.0f / 256.0f);
    C.tailU = tails + (size_t)blockIdx.x * 64 + lane;
    C.rstride = (size_t)gridDim.x * 64;
    C.rowbytes = P.Rpad * 4;
    const int hs = P.hs ? hs_shift[rc] : 0;
    C.vbase = (int)((unsigned)rc * 4u + (unsigned)hs * (unsigned)C.rowbytes);
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp), 0, (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    if (role == 0) {
        State<true> S;
        S.D1.init(); S.D2.init();
        S.l0 = S.l1 = S.l2 = 0;
        S.vl.reset(16); S.vpre.reset(16); S.vcU.reset(0); S.vcV.reset(0);
        S.sh.reset(0);
        S.oU.reset(0); S.oV.reset(0);
        S.Yprev = S.Uraw = S.Vraw = 0;
        // the fill draws twice per step from x1 = 0 on (t = 7 .. SK1 - 1 = 6 + d): the steady loop's first draw on a slot
        // that is a multiple of 8
        S.rng.init(ring, rs_chroma + rc, P.Rpad, lane, (-(31 + 2 * C.d)) & 7);
        S.nU = n0_u[rc]; S.nV = n0_v[rc];
        vcr_role(P, S, C, ring, c2ring, sync);
    } else {
        State<false> S;
        S.D1.init(); S.D2.init();
        S.l0 = S.l1 = S.l2 = 0;
        S.vl.reset(0); S.vpre.reset(0); S.vcU.reset(0); S.vcV.reset(0); S.sh.reset(0);
        S.oU.reset(0); S.oV.reset(0);
        S.Yprev = S.Uraw = S.Vraw = 0;
        S.nU = S.nV = 0;
        tv_role(P, S, C, c2ring, sync, ostage, orow, drow, is_out);
    }
}

// ---------------------------------------------------------------------------------------------- host side
void launch_encode_fp(hipStream_t st, const DevParams &D, const FieldDev *fields, const uint32_t *rs_luma, const int *n0_luma,
                      int *comp)
{
    hipLaunchKernelGGL(k_encode_fp, dim3((D.R + 63) / 64), dim3(64), 0, st, D, fields, rs_luma, n0_luma, comp);
}

void launch_decode_fp(hipStream_t st, const DevParams &D, const GeomDev &G, const FieldDev *fields, const int *comp,
                      const uint32_t *rs_chroma, const int *n0_u, const int *n0_v, const int *hs_shift, const int *pn_noise,
                      const int *dropout, int *tails, int variant)
{
    const dim3 grid((D.R + 62) / 63);
#define NTSC_FPV(VHS, VAR, WV) hipLaunchKernelGGL((k_decode_fp<VHS, VAR, WV>), grid, dim3(64), 0, st, D, G, fields, comp, rs_chroma, \
                                                  n0_u, n0_v, hs_shift, pn_noise, dropout, tails)
    if (!D.vhs) { NTSC_FPV(false, 0, 2); return; }
    // Two forms of the -vhs decoder.  One wave per 63 rows (k_decode_fp<true>): the form of long batches -- with tens of
    // thousands of rows in flight the chip's VALU issue is what bounds the step (measured 0.70 of it), and this form
    // issues least.  Two-role workgroup (k_decode_fp2): a field's rows are worked on by two instruction streams, so a
    // SHORT batch -- the synchronous one-field call, a launch of the submit engine -- finishes sooner (isolated 600-field
    // launch 0.48 against 0.52 ms; tools/fp_probe.sh).  variant: -1 = by batch size; 0 / 1 = two roles (4 / 3 waves per
    // SIMD); 10.. = one wave (developer A/B, NTSCSIM_FP_VARIANT).
    if (variant < 0) variant = D.nfields <= 128 ? 0 : 10;
    if (variant < 10) {
        if (variant == 1) hipLaunchKernelGGL((k_decode_fp2<3>), grid, dim3(128), 0, st, D, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift,
                                             pn_noise, dropout, tails);
        else hipLaunchKernelGGL((k_decode_fp2<4>), grid, dim3(128), 0, st, D, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift,
                                pn_noise, dropout, tails);
        return;
    }
    switch (variant - 10) {
        case 1: NTSC_FPV(true, 1, 2); break;
        default: NTSC_FPV(true, 0, 2); break;
    }
#undef NTSC_FPV
}

} // namespace ntscsim
