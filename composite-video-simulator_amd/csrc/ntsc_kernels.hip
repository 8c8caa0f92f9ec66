// ntsc_kernels.hip -- CDNA4 (gfx950) kernels for the per-field NTSC composite / VHS chain
// (reference: composite_layer(), ffmpeg_ntsc.cpp:1570-1921).
//
// Execution model.  Every filter in the reference is a SERIAL fp64 recurrence along a scanline
// whose result is truncated to int between stages, so bit-exactness forbids any re-association
// inside a row.  Parallelism therefore comes from scanlines: ONE LANE = ONE SCANLINE, a wavefront
// carries 64 scanlines in lock-step along x, and a batch of fields supplies tens of thousands of
// independent rows.  All lanes of a wave are always at the same x, which gives
//   * coalesced access to the transposed composite plane comp[x][row] (256 B per wave access),
//   * the VHS vertical chroma blend as a one-lane wave shift (__shfl_up), no line buffer,
//   * wave-uniform control flow for every pipeline guard.
// The whole stage chain is streamed in x with bounded look-ahead, so no intermediate plane ever
// touches HBM except the composite signal itself (the encoder|decoder interface, which the head
// switching stage needs random access to).
//
//   k_field_setup : 1 lane / field   head-switch geometry, per-row phase noise + dropout draws
//   k_row_states  : 1 lane / (row, noise stream)  rand() state + noise accumulator at row start
//   k_encode      : 1 lane / row     BGRA -> YIQ -> chroma LP -> QAM -> pre-emphasis -> luma noise
//   k_decode      : 1 lane / row     head switch -> Y/C split -> noise -> VHS -> TV LP -> BGRA
//
// fp contract: this file MUST be compiled with -ffp-contract=off (an FMA changes the results).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ntsc_device.hpp"

#pragma clang fp contract(off)

namespace ntscsim {

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------ integer helpers
DEV int sdiv2(int n) { return (n + (int)((unsigned)n >> 31)) >> 1; }   // C `/ 2` (truncating)
DEV int sdiv4(int n) { return (n + ((n >> 31) & 3)) >> 2; }             // C `/ 4`
DEV unsigned udiv31(unsigned n, const Magic31 &m) { return __umulhi(n, m.mul) >> m.shift; }
DEV unsigned umod31(unsigned n, const Magic31 &m) { return n - udiv31(n, m) * m.div; }
DEV int sdivm(int n, const Magic31 &m)                                   // C `/ d`, d > 0
{
    if (m.mul == 0) return n;                  // d == 1 has no 32-bit magic (wave-uniform branch)
    const unsigned a = (unsigned)(n < 0 ? -n : n);
    const int q = (int)udiv31(a, m);
    return n < 0 ? -q : q;
}

// scanline subcarrier phase, ffmpeg_ntsc.cpp:1473-1480 / :1529-1536
DEV unsigned scan_phase(const DevParams &P, unsigned y, uint64_t fieldno)
{
    const unsigned off = (unsigned)P.phase_off;
    if (P.phase_mode == 90)  return (unsigned)((fieldno + off + (y >> 1)) & 3);
    if (P.phase_mode == 180) return (unsigned)((((fieldno + y) & 2) + off) & 3);
    if (P.phase_mode == 270) return (unsigned)((fieldno + off - (y >> 1)) & 3);
    return off & 3;
}

// ------------------------------------------------------------------------------ option policy
// The hot kernels are instantiated twice: a GENERIC form that reads every option from DevParams
// (wave-uniform branches inside the loop), and a PRESET form in which the options of the two
// configurations that matter for throughput -- the default preset and the full `-vhs` preset --
// are compile-time constants, so the steady-state loop is one straight-line block.  The launcher
// picks the PRESET form only when the parameters match it exactly; results are identical.
enum : unsigned {
    F_GENERIC = 1u,     // read options at run time
    F_CNOISE  = 2u,     // chroma noise on
    F_PNOISE  = 4u,     // chroma phase noise on
    F_LNOISE  = 8u,     // luma noise on
};
template <unsigned F>
struct Opt {
    static constexpr bool generic = (F & F_GENERIC) != 0;
    DEV static bool cnoise(const DevParams &P) { return generic ? P.cnoise_k != 0 : (F & F_CNOISE) != 0; }
    DEV static bool pnoise(const DevParams &P) { return generic ? P.pnoise_k != 0 : (F & F_PNOISE) != 0; }
    DEV static bool lnoise(const DevParams &P) { return generic ? P.noise_k != 0 : (F & F_LNOISE) != 0; }
    DEV static bool nocolor(const DevParams &P) { return generic ? P.nocolor != 0 : false; }
    DEV static int  outlp(const DevParams &P) { return generic ? P.out_lp : 1; }      // preset: lite
    DEV static bool inlp(const DevParams &P) { return generic ? P.in_lp != 0 : true; }
    DEV static bool pre(const DevParams &P) { return generic ? P.pre_on != 0 : false; }
    // subcarrier amplitude 50 both ways: (c*50)/50 == c and (v*50)/50 == v exactly
    DEV static int scale_back(const DevParams &P, int c, const Magic31 &m)
    { return generic ? sdivm(c * 50, m) : c; }
    DEV static int modulate(const DevParams &P, int v) { return generic ? (v * P.amp) / 50 : v; }
};

// ------------------------------------------------------------------------------ one-pole IIR
// LowpassFilter::lowpass / highpass, ffmpeg_ntsc.cpp:90-99 (operation order is the contract)
// RT = double: the EXACT mode (bit-identical to the reference).  RT = float: the FAST mode the
// north star allows for "the filtered signal" (stated tolerance, tests/test_gpu_fast_mode.py):
// same pipeline, same integer stages and rand() stream, filters in fp32 in the algebraically
// equal form p += a*(s - p) (one subtract + one FMA per pole instead of two multiplies and two adds).
template <class RT>
struct OnePoleT {
    RT p;
    DEV RT lp(RT s, RT a)
    {
        const RT s1 = s * a;
        const RT s2 = p - (p * a);
        p = s1 + s2;
        return p;
    }
    DEV RT hp(RT s, RT a) { return s - lp(s, a); }
};
template <>
struct OnePoleT<float> {
    float p;
    DEV float lp(float s, float a) { p = __builtin_fmaf(a, s - p, p); return p; }
    DEV float hp(float s, float a) { return s - lp(s, a); }
};
template <class RT>
struct Lp3T {
    OnePoleT<RT> f0, f1, f2;
    DEV void reset(RT v) { f0.p = v; f1.p = v; f2.p = v; }
    DEV RT push(RT s, RT a) { return f2.lp(f1.lp(f0.lp(s, a), a), a); }
};
using OnePole = OnePoleT<double>;
using Lp3 = Lp3T<double>;
template <class RT> DEV RT rtrunc(RT v);
template <> DEV double rtrunc<double>(double v) { return trunc(v); }
template <> DEV float rtrunc<float>(float v) { return truncf(v); }

// ------------------------------------------------------------------------------ rand() in LDS
// Per-lane glibc TYPE_3 generator: the 31-word window lives in LDS as ring[slot][lane]
// (conflict-free: bank = lane), the three newest words in registers.
struct LaneRand {
    uint32_t p3, p2, p1;   // s[i-3], s[i-2], s[i-1]
    int slot;              // wave-uniform
    DEV void init(uint32_t *ring, const uint32_t *state, int stride, int lane)
    {
        for (int j = 0; j < 31; j++) ring[j * 64 + lane] = state[(size_t)j * stride];
        p3 = ring[28 * 64 + lane];
        p2 = ring[29 * 64 + lane];
        p1 = ring[30 * 64 + lane];
        slot = 0;
    }
    DEV uint32_t next(uint32_t *ring, int lane)
    {
#ifdef NTSC_AB_NORAND      // timing-only A/B build (WRONG pixels): no LDS ring
        (void)ring; (void)lane;
        p3 = p3 * 1664525u + 1013904223u;
        return p3 >> 1;
#endif
        const uint32_t v = ring[slot * 64 + lane] + p3;   // s[i-31] + s[i-3]
        ring[slot * 64 + lane] = v;
        p3 = p2; p2 = p1; p1 = v;
        slot = (slot == 30) ? 0 : slot + 1;
        return v >> 1;
    }
};

// Jump-ahead: state advanced by the polynomial c (x^n mod x^31 - x^28 - 1) given the 61-word
// extension w of the starting window:  out[j] = sum_k c[k] * w[j+k].  Fully unrolled so that
// everything stays in registers (private arrays with dynamic indices would live in scratch).
DEV void jump61(const uint32_t *__restrict__ c, const uint32_t *__restrict__ w, uint32_t (&o)[31])
{
    uint32_t cc[31], ww[61];
#pragma unroll
    for (int k = 0; k < 31; k++) cc[k] = c[k];
#pragma unroll
    for (int i = 0; i < 61; i++) ww[i] = w[i];
#pragma unroll
    for (int j = 0; j < 31; j++) {
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 31; k++) acc += cc[k] * ww[j + k];
        o[j] = acc;
    }
}

// Per-lane generator for the setup kernels: same LDS ring as LaneRand but the slot is a per-lane
// value (lanes run different numbers of draws).
struct SetupRand {
    uint32_t p3, p2, p1;
    int slot;
    DEV void init(uint32_t *ring, const uint32_t (&st)[31], int lane)
    {
#pragma unroll
        for (int j = 0; j < 31; j++) ring[j * 64 + lane] = st[j];
        p3 = st[28]; p2 = st[29]; p1 = st[30];
        slot = 0;
    }
    DEV uint32_t next(uint32_t *ring, int lane)
    {
        const uint32_t v = ring[slot * 64 + lane] + p3;
        ring[slot * 64 + lane] = v;
        p3 = p2; p2 = p1; p1 = v;
        slot = (slot == 30) ? 0 : slot + 1;
        return v >> 1;
    }
};

DEV int field_rows(const DevParams &P, unsigned field) { return (P.H - (int)field + 1) / 2; }

// =============================================================================== k_field_setup
// Per field: the draws that are not per-pixel.  Order of draws inside one composite_layer call
// (SURVEY A.10): [W*L luma] [4 head switch] [2*W*L chroma] [L phase noise] [L dropout].
DEV void field_setup_body(const DevParams &P, const GeomDev &G, const FieldDev *__restrict__ fields,
                          int *__restrict__ hs_shift, int *__restrict__ pn_noise, int *__restrict__ dropout,
                          uint32_t *ring, int block)
{
    const int lane = threadIdx.x;
    const int f = block * 64 + lane;
    if (f >= P.nfields) return;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const int L = field_rows(P, field);
    int *hs_row = hs_shift + (size_t)f * P.Lslot;
    uint32_t st[31];
    SetupRand g;

    // VHS head switching geometry, ffmpeg_ntsc.cpp:1647-1713
    if (P.hs) {
        // (every row of the field's slot, also the ones the switch does not reach: no fill of the plane before this kernel)
        for (int k = 0; k < P.Lslot; k++) hs_row[k] = 0;
        const unsigned twidth = (unsigned)P.W + ((unsigned)P.W / 10u);
        double noise = 0;
        if (P.hs_noise_on) {
            jump61(G.lskip + field * 31, fd.rng, st);      // skip the luma-noise draws
            g.init(ring, st, lane);
            unsigned u = g.next(ring, lane);
            u *= g.next(ring, lane); u *= g.next(ring, lane); u *= g.next(ring, lane);
            u %= 2000000000U;
            noise = ((double)u / 1000000000U) - 1.0;
            noise *= P.hs_pn;
        }
        const double t = P.ntsc ? twidth * 262.5 : twidth * 312.5;
        // fmod(a, 1.0) == a - trunc(a) exactly (params are validated non-negative)
        // ffmpeg_ntsc: row from `point`, column from `phase` (:1666-1670);
        // ffmpeg_to_composite: both from the single `phase` parameter (:691-693)
        double a = (P.variant ? P.hs_phase : P.hs_point) + noise;
        unsigned pp = (unsigned)((a - trunc(a)) * t);
        int y = (int)((pp / twidth) * 2u) + (int)field;
        a = P.hs_phase + noise;
        pp = (unsigned)((a - trunc(a)) * t);
        const unsigned hx = pp % twidth;
        y -= P.ntsc ? (262 - 240) * 2 : (312 - 288) * 2;
        const int ishif = (hx >= twidth / 2) ? (int)(hx - twidth) : (int)hx;
        // the first row (shif = 0) is a no-op; later rows always start at tx = 0
        int shif = 0;
        unsigned shy = 0;
        while (y < P.H) {
            if (y >= 0 && shif != 0) hs_row[(y - (int)field) >> 1] = shif;
            shif = (shy == 0) ? ishif : (shif * 7) / 8;
            y += 2;
            shy++;
        }
    }

    if (P.pnoise_k || P.loss) {
        jump61(G.pskip + field * 31, fd.rng, st);          // skip luma + head switch + chroma
        g.init(ring, st, lane);
        // chroma phase noise accumulator, one draw per row, carried down the field (:1736-1746)
        if (P.pnoise_k) {
            int n = 0;
            for (int k = 0; k < L; k++) {
                n += (int)umod31(g.next(ring, lane), P.m_pnoise) - P.pnoise_k;
                n = sdiv2(n);
                pn_noise[(size_t)f * P.Lslot + k] = n;
            }
        }
        // chroma dropout, one draw per row (:1891-1901)
        if (P.loss) {
            for (int k = 0; k < L; k++)
                dropout[(size_t)f * P.Lslot + k] = (g.next(ring, lane) % 100000U) < (unsigned)P.loss;
        }
    }
}

__global__ __launch_bounds__(64) void k_field_setup(DevParams P, GeomDev G,
                                                    const FieldDev *__restrict__ fields,
                                                    int *__restrict__ hs_shift,
                                                    int *__restrict__ pn_noise,
                                                    int *__restrict__ dropout)
{
    __shared__ uint32_t ring[31 * 64];
    field_setup_body(P, G, fields, hs_shift, pn_noise, dropout, ring, (int)blockIdx.x);
}

// =============================================================================== k_row_states
// rand() state and noise accumulator(s) at the first pixel of every scanline.
// The accumulators are carried across rows in the reference (noise = (noise + d - k) / 2, C
// truncation), but the map is monotone in `noise` and halves the distance, so running it from
// both extremes (-k and +k) over a short warm-up pins the exact value as soon as the two
// trajectories meet.  If they have not met after the warm-up (probability ~2^-warm) the lane
// recomputes serially from the start of the field -- exact by construction either way.
DEV void row_states_body(const DevParams &P, const GeomDev &G, const FieldDev *__restrict__ fields,
                         uint32_t *__restrict__ rs_luma, int *__restrict__ n0_luma,
                         uint32_t *__restrict__ rs_chroma, int *__restrict__ n0_u, int *__restrict__ n0_v,
                         uint32_t *ring, int block, int stream)
{
    const int lane = threadIdx.x;
    const int rho = block * 64 + lane;             // stream: 0 luma, 1 chroma
    if (rho >= P.R) return;
    if (stream == 0 ? !P.noise_k : !P.cnoise_k) return;
    const int f = rho / P.Lslot, k = rho - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned par = fd.field & 1u;
    if (k >= field_rows(P, par)) return;

    const size_t jidx = ((size_t)(stream * 2 + par) * P.Lslot + k);
    const int warm = G.jwarm[jidx];
    const int K = stream == 0 ? P.noise_k : P.cnoise_k;
    const Magic31 M = stream == 0 ? P.m_noise : P.m_cnoise;
    // draws made before this row: luma 1/pixel; chroma 2/pixel (BGRA path) or 2/chroma sample
    const long long cpr = P.variant ? 2ll * (P.W / 2) : 2ll * P.W;
    const long long start = stream == 0 ? (long long)k * P.W : cpr * k;
    const bool exact = (long long)warm == start;   // warm-up reaches the start of the stream

    uint32_t st[31];
    SetupRand g;
    jump61(G.jrow + jidx * 31, fd.rng, st);
    g.init(ring, st, lane);

    int lo0 = exact ? 0 : -K, hi0 = exact ? 0 : K;   // luma / U
    int lo1 = stream == 0 ? 0 : lo0, hi1 = stream == 0 ? 0 : hi0;   // V (chroma stream only)
    if (stream == 0) {
        for (int i = 0; i < warm; i++) {
            const int d = (int)umod31(g.next(ring, lane), M) - K;
            lo0 = sdiv2(lo0 + d); hi0 = sdiv2(hi0 + d);
        }
    } else {
        for (int i = 0; i < warm; i += 2) {
            int d = (int)umod31(g.next(ring, lane), M) - K;
            lo0 = sdiv2(lo0 + d); hi0 = sdiv2(hi0 + d);
            d = (int)umod31(g.next(ring, lane), M) - K;
            lo1 = sdiv2(lo1 + d); hi1 = sdiv2(hi1 + d);
        }
    }
    if (lo0 != hi0 || lo1 != hi1) {
        // not pinned: serial replay from the first draw of this stream in this field
        jump61(G.sstart + (size_t)(stream * 2 + par) * 31, fd.rng, st);
        g.init(ring, st, lane);
        lo0 = lo1 = 0;
        if (stream == 0) {
            for (long long i = 0; i < start; i++)
                lo0 = sdiv2(lo0 + (int)umod31(g.next(ring, lane), M) - K);
        } else {
            for (long long i = 0; i < start; i += 2) {
                lo0 = sdiv2(lo0 + (int)umod31(g.next(ring, lane), M) - K);
                lo1 = sdiv2(lo1 + (int)umod31(g.next(ring, lane), M) - K);
            }
        }
    }

    uint32_t *rs = stream == 0 ? rs_luma : rs_chroma;
    int q = g.slot;
    for (int j = 0; j < 31; j++) {
        rs[(size_t)j * P.Rpad + rho] = ring[q * 64 + lane];
        q = (q == 30) ? 0 : q + 1;
    }
    if (stream == 0) n0_luma[rho] = lo0;
    else { n0_u[rho] = lo0; n0_v[rho] = lo1; }
}

__global__ __launch_bounds__(64) void k_row_states(DevParams P, GeomDev G,
                                                   const FieldDev *__restrict__ fields,
                                                   uint32_t *__restrict__ rs_luma,
                                                   int *__restrict__ n0_luma,
                                                   uint32_t *__restrict__ rs_chroma,
                                                   int *__restrict__ n0_u, int *__restrict__ n0_v)
{
    __shared__ uint32_t ring[31 * 64];
    row_states_body(P, G, fields, rs_luma, n0_luma, rs_chroma, n0_u, n0_v, ring, (int)blockIdx.x, (int)blockIdx.y);
}

// One PART of a field's setup by one block (the short batches of the host-frame calls: k_field_row_setup): part 0 the head
// switch, 1 the phase-noise walk, 2 the dropout walk.  field_setup_body does the three one after the other in one lane (a
// lane per field: right for hundreds of fields, 36 us for one); here the three walks of a field run side by side, and
// the jump to a walk's first draw -- 31 dot products of 31 terms, ~8 us in one lane -- is one dot product per lane.
DEV void field_part_body(const DevParams &P, const GeomDev &G, const FieldDev *__restrict__ fields,
                         int *__restrict__ hs_shift, int *__restrict__ pn_noise, int *__restrict__ dropout,
                         uint32_t *ring, int f, int part)
{
    const int lane = threadIdx.x;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const int L = field_rows(P, field);
    const bool on = part == 0 ? P.hs != 0 : (part == 1 ? P.pnoise_k != 0 : P.loss != 0);
    if (!on) return;
    int *hs_row = hs_shift + (size_t)f * P.Lslot;
    if (part == 0)      // (every row of the field's slot, also the ones the switch does not reach)
        for (int k = lane; k < P.Lslot; k += 64) hs_row[k] = 0;
    if (part == 0) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the zeros have landed before lane 0 writes its rows
    if (part != 0 || P.hs_noise_on) {
        // part 2 starts behind the phase-noise draws: the second half of the pskip table (build_geometry)
        const uint32_t *c = (part == 0 ? G.lskip : G.pskip + (part == 2 ? 62 : 0)) + field * 31;
        if (lane < 31) {
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 31; k++) acc += c[k] * fd.rng[lane + k];
            ring[lane * 64] = acc;
        }
    }
    __syncthreads();
    if (lane != 0) return;
    SetupRand g;
    g.p3 = ring[28 * 64]; g.p2 = ring[29 * 64]; g.p1 = ring[30 * 64];
    g.slot = 0;
    if (part == 0) {
        // VHS head switching geometry, ffmpeg_ntsc.cpp:1647-1713 (as field_setup_body)
        const unsigned twidth = (unsigned)P.W + ((unsigned)P.W / 10u);
        double noise = 0;
        if (P.hs_noise_on) {
            unsigned u = g.next(ring, 0);
            u *= g.next(ring, 0); u *= g.next(ring, 0); u *= g.next(ring, 0);
            u %= 2000000000U;
            noise = ((double)u / 1000000000U) - 1.0;
            noise *= P.hs_pn;
        }
        const double t = P.ntsc ? twidth * 262.5 : twidth * 312.5;
        double a = (P.variant ? P.hs_phase : P.hs_point) + noise;
        unsigned pp = (unsigned)((a - trunc(a)) * t);
        int y = (int)((pp / twidth) * 2u) + (int)field;
        a = P.hs_phase + noise;
        pp = (unsigned)((a - trunc(a)) * t);
        const unsigned hx = pp % twidth;
        y -= P.ntsc ? (262 - 240) * 2 : (312 - 288) * 2;
        const int ishif = (hx >= twidth / 2) ? (int)(hx - twidth) : (int)hx;
        int shif = 0;
        unsigned shy = 0;
        while (y < P.H) {
            if (y >= 0 && shif != 0) hs_row[(y - (int)field) >> 1] = shif;
            shif = (shy == 0) ? ishif : (shif * 7) / 8;
            y += 2;
            shy++;
        }
    } else if (part == 1) {
        // chroma phase noise accumulator, one draw per row, carried down the field (:1736-1746)
        int n = 0;
        for (int k = 0; k < L; k++) {
            n += (int)umod31(g.next(ring, 0), P.m_pnoise) - P.pnoise_k;
            n = sdiv2(n);
            pn_noise[(size_t)f * P.Lslot + k] = n;
        }
    } else {
        // chroma dropout, one draw per row (:1891-1901)
        for (int k = 0; k < L; k++)
            dropout[(size_t)f * P.Lslot + k] = (g.next(ring, 0) % 100000U) < (unsigned)P.loss;
    }
}

// Both in ONE launch, for the short batches of the host-frame calls (ntscsim_field(): one field; a submit lane: `depth`
// fields): the two are independent of each other, and with a handful of fields k_field_setup is a single wavefront
// walking its rows one draw at a time (36 us for one field) -- as blocks of the same grid that walk overlaps the row
// states instead of preceding them.  Blocks [0, nfs) do the field setup, one block per field and part (first: they run
// longest), the rest the row states.
__global__ __launch_bounds__(64) void k_field_row_setup(DevParams P, GeomDev G, const FieldDev *__restrict__ fields,
                                                        int *__restrict__ hs_shift, int *__restrict__ pn_noise,
                                                        int *__restrict__ dropout,
                                                        uint32_t *__restrict__ rs_luma, int *__restrict__ n0_luma,
                                                        uint32_t *__restrict__ rs_chroma,
                                                        int *__restrict__ n0_u, int *__restrict__ n0_v, int nfs, int nrs)
{
    __shared__ uint32_t ring[31 * 64];
    const int b = (int)blockIdx.x;
    if (b < nfs) { field_part_body(P, G, fields, hs_shift, pn_noise, dropout, ring, b / 3, b % 3); return; }
    const int q = b - nfs;
    row_states_body(P, G, fields, rs_luma, n0_luma, rs_chroma, n0_u, n0_v, ring, q % nrs, q / nrs);
}

// =============================================================================== k_encode
// BGRA row -> composite signal (int32, Y*256 with the chroma subcarrier riding on it).
// Stream position t reads pixel t; output sample x = t - 4 (the Q low-pass looks 4 ahead).

DEV void load_px16(const uint8_t *srow, int x0, int W, bool al16, uint32_t (&px)[16])
{
    if (al16 && x0 + 16 <= W) {
        const uint4 *p = reinterpret_cast<const uint4 *>(srow + 4 * (size_t)x0);
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        px[0] = a.x; px[1] = a.y; px[2] = a.z; px[3] = a.w;
        px[4] = b.x; px[5] = b.y; px[6] = b.z; px[7] = b.w;
        px[8] = c.x; px[9] = c.y; px[10] = c.z; px[11] = c.w;
        px[12] = d.x; px[13] = d.y; px[14] = d.z; px[15] = d.w;
    } else {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(srow);
#pragma unroll
        for (int j = 0; j < 16; j++) px[j] = (x0 + j < W) ? p[x0 + j] : 0u;
    }
}

template <class RT>
struct EncState {
    Lp3T<RT> lpI, lpQ;
    OnePoleT<RT> pre;
    // delay windows: element 0 is the oldest (sample t-4), element 4 the newest (sample t)
    int Yw[5], Iw[5], Qw[5];
    int fI[3];
    LaneRand rng;
    int noise;
};

// One encoder step: consumes pixel t, emits composite sample x = t - 4.  EDGE=false is the
// steady state (4 <= t < W): no row-boundary predicate survives.
template <bool EDGE, class O, class RT>
DEV void enc_step(const DevParams &P, EncState<RT> &S, unsigned xi, int W, uint32_t *ring, int lane,
                  int t, uint32_t px_in, int *cdst, bool valid)
{
    // ---- RGB -> YIQ, ffmpeg_ntsc.cpp:1375-1383 (pixels past the row end feed zeros into
    //      filters whose outputs are never used)
    const uint32_t px = (!EDGE || t < W) ? px_in : 0u;
    const int r = (int)((px >> 16) & 0xFF), g = (int)((px >> 8) & 0xFF), b = (int)(px & 0xFF);
    const RT dY = (RT(0.30) * r) + (RT(0.59) * g) + (RT(0.11) * b);
    const int Yn = (int)(256 * dY);
    const int In = (int)(256 * ((RT(-0.27) * (b - dY)) + (RT(0.74) * (r - dY))));
    const int Qn = (int)(256 * ((RT(0.41) * (b - dY)) + (RT(0.48) * (r - dY))));
#pragma unroll
    for (int q = 0; q < 4; q++) { S.Yw[q] = S.Yw[q + 1]; S.Iw[q] = S.Iw[q + 1]; S.Qw[q] = S.Qw[q + 1]; }
    S.Yw[4] = Yn; S.Iw[4] = In; S.Qw[4] = Qn;
    // ---- input chroma low-pass, composite_lowpass :1429-1458 (I: 1.3 MHz delay 2,
    //      Q: 0.6 MHz delay 4; the last `delay` samples keep their input)
    int fQ = 0;
    if (O::inlp(P)) {
        S.fI[0] = S.fI[1]; S.fI[1] = S.fI[2];
        S.fI[2] = (int)S.lpI.push((RT)In, (RT)P.a_in_i);
        fQ = (int)S.lpQ.push((RT)Qn, (RT)P.a_in_q);
    }
    const int x = t - 4;
    if (EDGE && x < 0) return;
    int I1 = S.Iw[0], Q1 = S.Qw[0];
    if (O::inlp(P)) {
        if (!EDGE || x < W - 2) I1 = S.fI[0];
        if (!EDGE || x < W - 4) Q1 = fQ;
    }
    // ---- chroma_into_luma :1460-1495
    const unsigned s = (xi + (unsigned)x) & 3u;
    int chroma = O::modulate(P, (s & 1u) ? Q1 : I1);
    if (s & 2u) chroma = -chroma;
    int Y = S.Yw[0] + chroma;
    // ---- composite pre-emphasis :1614-1629
    if (O::pre(P)) {
        RT sd = Y;
        sd += S.pre.hp(sd, (RT)P.a_pre) * (RT)P.pre_gain;
        Y = (int)sd;
    }
    // ---- luma noise :1632-1644
    if (O::lnoise(P)) {
        Y += S.noise;
        S.noise += (int)umod31(S.rng.next(ring, lane), P.m_noise) - P.noise_k;
        S.noise = sdiv2(S.noise);
    }
    if (valid) cdst[(size_t)x * P.Rpad] = Y;
}

template <unsigned F, class RT>
__global__ __launch_bounds__(64) void k_encode(DevParams P, const FieldDev *__restrict__ fields,
                                               const uint32_t *__restrict__ rs_luma,
                                               const int *__restrict__ n0_luma,
                                               int *__restrict__ comp)
{
    using O = Opt<F>;
    __shared__ uint32_t ring[31 * 64];
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
    const unsigned xi = scan_phase(P, y, fd.fieldno);
    const int W = P.W;
    const bool al = P.src_al16 != 0;

    EncState<RT> S;
    S.noise = 0;
    if (O::lnoise(P)) {
        S.rng.init(ring, rs_luma + rc, P.Rpad, lane);
        S.noise = n0_luma[rc];
    }
    S.lpI.reset(0); S.lpQ.reset(0);
    S.pre.p = 16;
#pragma unroll
    for (int q = 0; q < 5; q++) { S.Yw[q] = 0; S.Iw[q] = 0; S.Qw[q] = 0; }
    S.fI[0] = S.fI[1] = S.fI[2] = 0;
    int *cdst = comp + rho;

    uint32_t cur[16], nxt[16];
    load_px16(srow, 0, W, al, cur);
    int t0 = 0;
    // first chunk: pipeline fill (guarded)
    {
        if (16 < W) load_px16(srow, 16, W, al, nxt);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < W + 4) enc_step<true, O, RT>(P, S, xi, W, ring, lane, j, cur[j], cdst, valid);
#pragma unroll
        for (int j = 0; j < 16; j++) cur[j] = nxt[j];
        t0 = 16;
    }
    // steady state: whole 16-pixel chunks strictly inside the row
    for (; t0 + 16 <= W; t0 += 16) {
        if (t0 + 16 < W) load_px16(srow, t0 + 16, W, al, nxt);
#pragma unroll
        for (int j = 0; j < 16; j++)
            enc_step<false, O, RT>(P, S, xi, W, ring, lane, t0 + j, cur[j], cdst, valid);
#pragma unroll
        for (int j = 0; j < 16; j++) cur[j] = nxt[j];
    }
    // row end + pipeline drain (guarded)
    for (; t0 < W + 4; t0 += 16) {
        if (t0 + 16 < W) load_px16(srow, t0 + 16, W, al, nxt);
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (t0 + j < W + 4) enc_step<true, O, RT>(P, S, xi, W, ring, lane, t0 + j, cur[j], cdst, valid);
#pragma unroll
        for (int j = 0; j < 16; j++) cur[j] = nxt[j];
    }
}

// =============================================================================== k_decode
// streaming chroma_from_luma (ffmpeg_ntsc.cpp:1497-1567).  push(t) takes composite sample t and
// returns Y/I/Q for x = t - 7.  EDGE=false is the steady-state specialisation: every row-boundary
// predicate is known (2 <= q, q+1 < W, x+xi+2 < W, x < xe), so they compile away.
struct Demod {
    int c0, c1, c2;               // cs(t-3), cs(t-2), cs(t-1)
    int w0, w1, w2, w3, w4, w5;   // scaled chroma at q-5 .. q   (q = t-2)
    int y0, y1, y2, y3, y4;       // box-filtered luma at q-5 .. q-1
    int ie_prev, qe_prev, ie_next, qe_next;
    DEV void init()
    {
        c0 = c1 = c2 = 0;
        w0 = w1 = w2 = w3 = w4 = w5 = 0;
        y0 = y1 = y2 = y3 = y4 = 0;
        ie_prev = qe_prev = ie_next = qe_next = 0;
    }
    DEV static int sel4(unsigned xi, int a0, int a1, int a2, int a3)
    {
        const int lo = (xi & 1u) ? a1 : a0;
        const int hi = (xi & 1u) ? a3 : a2;
        return (xi & 2u) ? hi : lo;
    }
    // PAR: parity of the output position x = t - 7 when it is known at compile time (0 even,
    // 1 odd), -1 to test it at run time.
    template <bool EDGE, class O, int PAR>
    DEV void push(const DevParams &P, int ct, int t, unsigned xi, int W, int xe, const Magic31 &mA,
                  bool nocolor, int &Yo, int &Io, int &Qo)
    {
        const int q = t - 2;
        // 4-tap box with zero extension (:1507-1525)
        const int yb = nocolor ? c1 : sdiv4(c0 + c1 + c2 + ct);
        int ch = ct - yb;
        // un-flip the negative half cycles (:1539-1542): positions x+2, x+3 for
        // x = (4-xi)&3 + 4m while x+3 < W
        const unsigned g = (unsigned)(q - 2 + (int)xi) & 3u;
        bool neg;
        if (EDGE) neg = (g == 0u && q >= 2 && q + 1 < W) || (g == 1u && q >= 3);
        else neg = g < 2u;
        if (neg) ch = -ch;
        ch = O::scale_back(P, ch, mA);                                   // :1544-1546
        w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = ch;
        c0 = c1; c1 = c2; c2 = ct;
        Yo = y0;
        y0 = y1; y1 = y2; y2 = y3; y3 = y4; y4 = yb;
        const int x = q - 5;
        int I, Q;
        const bool odd = PAR < 0 ? (x & 1) != 0 : PAR == 1;
        if (odd) {
            // odd x (and x = -1): fetch the even sample at x+1, interpolate (:1549-1561)
            const bool m = EDGE ? (x + 1 + (int)xi + 1) < W : true;
            ie_next = m ? -sel4(xi, w1, w2, w3, w4) : 0;
            qe_next = m ? -sel4(xi, w2, w3, w4, w5) : 0;
            I = (ie_prev + ie_next) >> 1;
            Q = (qe_prev + qe_next) >> 1;
        } else {
            I = ie_next; Q = qe_next;
            ie_prev = ie_next; qe_prev = qe_next;
        }
        if ((EDGE && x >= xe) || nocolor) { I = 0; Q = 0; }              // :1553-1556, :1562-1565
        Io = I; Qo = Q;
    }
};

// per-lane state of the decode pipeline
template <class RT>
struct DecState {
    Demod D1, D2;
    int l0, l1, l2;               // luma stream window (VHS)
    Lp3T<RT> vl, vcU, vcV, sh, oU, oV;
    OnePoleT<RT> vpre;
    int Yd[5];                    // luma delayed to the output position (0 oldest)
    int Ur[5], Vr[5];             // raw chroma at the output stage (row tails)
    int Uf[3];                    // filtered U waiting for V (full output low-pass only)
    LaneRand rng;
    int nU, nV;
};

// per-lane / per-launch constants of the decode pipeline
struct DecConst {
    unsigned xi;
    int W, xe, k, lane;
    int d, SKO, dI, dQ;
    bool vb, drop;
    double cosv, sinv;
    int *tailU, *tailV;           // [16][Rpad] global scratch, this lane's column
    size_t rstride;
};

// One pipeline step at stream position t.  Returns true when a pixel for x = *xo was produced.
template <bool VHS, bool COMPOUT, bool EDGE, class O, int PAR1, int PAR2, class RT>
DEV bool dec_step(const DevParams &P, DecState<RT> &S, const DecConst &C, uint32_t *ring, int t,
                  int pc, int pl, uint32_t &px, int &xo_out)
{
    const int W = C.W;
    const int lane = C.lane;
    // ================= Y/C separation #1 at x1 = t - 7 (:1716, amplitude_back)
    int Y, U, V;
    S.D1.template push<EDGE, O, PAR1>(P, pc, t, C.xi, W, C.xe, P.m_amp_back, O::nocolor(P), Y, U, V);
    const int x1 = t - 7;
    const bool in1 = EDGE ? (x1 >= 0 && x1 < W) : true;
    constexpr bool FAST = !EDGE && !O::generic;   // steady state of a PRESET kernel
    RT Ud = 0, Vd = 0;                             // chroma as reals (== U, V)
    if (in1) {
        // chroma noise :1719-1735
        if (O::cnoise(P)) {
            U += S.nU; V += S.nV;
            S.nU = sdiv2(S.nU + (int)umod31(S.rng.next(ring, lane), P.m_cnoise) - P.cnoise_k);
            S.nV = sdiv2(S.nV + (int)umod31(S.rng.next(ring, lane), P.m_cnoise) - P.cnoise_k);
        }
        // chroma phase noise :1748-1762
        if (O::pnoise(P)) {
            const RT u = U, v = V;
            const RT cosv = (RT)C.cosv, sinv = (RT)C.sinv;
            const RT u_ = (u * cosv) - (v * sinv);
            const RT v_ = (u * sinv) + (v * cosv);
            if (FAST && VHS) {
                // (double)(int)d == trunc(d) up to the sign of zero, which no later stage can
                // observe (filters start from +0/16, results are truncated to int): one v_trunc
                // instead of a convert pair.  The int copies are only needed for the row tail.
                Ud = rtrunc<RT>(u_); Vd = rtrunc<RT>(v_);
            } else { U = (int)u_; V = (int)v_; Ud = U; Vd = V; }
        } else { Ud = U; Vd = V; }
    }
    int x2 = x1;           // position after the VHS block
    if (VHS) {
        const int d = C.d;
        // ---- VHS chroma low-pass :1814-1836: value for input x1 lands at x1 - d
        int fU = 0, fV = 0;
        if (in1) {
            fU = (int)S.vcU.push(Ud, (RT)P.a_vc);
            fV = (int)S.vcV.push(Vd, (RT)P.a_vc);
            if (EDGE && x1 >= W - 16) {
                C.tailU[(size_t)(x1 & 15) * C.rstride] = U;
                C.tailV[(size_t)(x1 & 15) * C.rstride] = V;
            }
        }
        x2 = x1 - d;
        // ---- luma path at x2: box (or pass-through) -> low-pass + emphasis -> sharpen
        const int lc = pl;                 // cs(x2 + 2)
        const int yb = O::nocolor(P) ? S.l1 : sdiv4(S.l0 + S.l1 + S.l2 + lc);
        S.l0 = S.l1; S.l1 = S.l2; S.l2 = lc;
        const bool in2 = EDGE ? (x2 >= 0 && x2 < W) : true;
        if (in2) {
            if (EDGE && x2 >= W - d) {
                fU = C.tailU[(size_t)(x2 & 15) * C.rstride];
                fV = C.tailV[(size_t)(x2 & 15) * C.rstride];
            }
            U = fU; V = fV;
            // luma low-pass + emphasis :1793-1812
            RT s = yb;
            s = S.vl.push(s, (RT)P.a_vl);
            s += S.vpre.hp(s, (RT)P.a_vl) * RT(1.6);
            // Y = (int)s, then sharpen reads it back as a double :1866-1883
            {
                RT s0;
                if (FAST) s0 = rtrunc<RT>(s); else { Y = (int)s; s0 = Y; }
                const RT ts = S.sh.push(s0, (RT)P.a_sh);
                Y = (int)(s0 + ((s0 - ts) * (RT)P.sharpen * 2));
            }
        }
        // ---- vertical chroma blend :1843-1863 (wave shift: lane-1 is the row above)
        {
            const int upU = __shfl_up(U, 1), upV = __shfl_up(V, 1);
            if (C.vb && C.k >= 1) {
                const int pu = C.k >= 2 ? upU : 0, pv = C.k >= 2 ? upV : 0;
                U = (pu + U + 1) >> 1;
                V = (pv + V + 1) >> 1;
            }
        }
    }
    int x3 = x2;
    if (COMPOUT) {
        // ---- composite out of the VCR :1885-1888: modulate, then separate again
        int c2 = 0;
        if (!EDGE || (x2 >= 0 && x2 < W)) {
            const unsigned s = (C.xi + (unsigned)x2) & 3u;
            int chroma = O::modulate(P, (s & 1u) ? V : U);      // (v*amp)/50, sign applied after:
            if (s & 2u) chroma = -chroma;                        // truncation is symmetric
            c2 = Y + chroma;
        }
        S.D2.template push<EDGE, O, PAR2>(P, c2, x2, C.xi, W, C.xe, P.m_amp, false, Y, U, V);
        x3 = x2 - 7;
    }
    const int SKO = C.SKO;
    if (EDGE && (x3 < 0 || x3 >= W + SKO)) return false;
    // ================= dropout :1891-1901, output low-pass :1903-1908 at x3
    const bool in3 = EDGE ? x3 < W : true;
    if (C.drop || !in3) { U = 0; V = 0; }
    if (!in3) Y = 0;
    if (FAST) {
        // PRESET steady state (output low-pass = lite, delay 1): the next step only reads the
        // previous luma sample; the raw-chroma windows are refilled by the >=16 guarded steps of
        // the epilogue before anything reads them.  Filter outputs stay doubles (trunc, see above).
        const RT fUd = rtrunc<RT>(S.oU.push((RT)U, (RT)P.a_tv));
        const RT fVd = rtrunc<RT>(S.oV.push((RT)V, (RT)P.a_tv));
        const int Yo = S.Yd[4];                    // luma of the previous step == position xo
        S.Yd[3] = S.Yd[4]; S.Yd[4] = Y;
        int r = (int)(((RT(1.000) * Yo) + (RT(0.956) * fUd) + (RT(0.621) * fVd)) / 256);
        int g = (int)(((RT(1.000) * Yo) + (RT(-0.272) * fUd) + (RT(-0.647) * fVd)) / 256);
        int b = (int)(((RT(1.000) * Yo) + (RT(-1.106) * fUd) + (RT(1.703) * fVd)) / 256);
        r = r < 0 ? 0 : (r > 255 ? 255 : r);
        g = g < 0 ? 0 : (g > 255 ? 255 : g);
        b = b < 0 ? 0 : (b > 255 ? 255 : b);
        px = ((uint32_t)r << 16) + ((uint32_t)g << 8) + (uint32_t)b;
        xo_out = x3 - 1;
        return true;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) { S.Yd[q] = S.Yd[q + 1]; S.Ur[q] = S.Ur[q + 1]; S.Vr[q] = S.Vr[q + 1]; }
    S.Yd[4] = Y; S.Ur[4] = U; S.Vr[4] = V;
    int fU = 0, fV = 0;
    const int out_lp = O::outlp(P);
    if (out_lp && in3) {
        const RT a_u = (RT)(out_lp == 1 ? P.a_tv : P.a_in_i);
        const RT a_v = (RT)(out_lp == 1 ? P.a_tv : P.a_in_q);
        fU = (int)S.oU.push((RT)U, a_u);
        fV = (int)S.oV.push((RT)V, a_v);
    }
    S.Uf[0] = S.Uf[1]; S.Uf[1] = S.Uf[2]; S.Uf[2] = fU;
    const int xo = x3 - SKO;
    if (EDGE && xo < 0) return false;
    // (static selects instead of Yd[4 - SKO]: dynamic register indexing would go to scratch)
    const int Yo = SKO == 0 ? S.Yd[4] : (SKO == 1 ? S.Yd[3] : S.Yd[0]);
    int Uo, Vo;
    if (out_lp == 0) { Uo = U; Vo = V; }
    else {
        // U value for xo was produced dI steps after xo entered, V value dQ steps after
        const int Uraw = SKO == 1 ? S.Ur[3] : S.Ur[0];
        const int Vraw = SKO == 1 ? S.Vr[3] : S.Vr[0];
        const int Ufil = (C.dQ - C.dI) == 0 ? S.Uf[2] : S.Uf[0];
        Uo = (!EDGE || xo < W - C.dI) ? Ufil : Uraw;
        Vo = (!EDGE || xo < W - C.dQ) ? fV : Vraw;
    }
    // ================= YIQ -> RGB :1385-1396, pack :1914 (alpha = 0)
    int r = (int)(((RT(1.000) * Yo) + (RT(0.956) * Uo) + (RT(0.621) * Vo)) / 256);
    int g = (int)(((RT(1.000) * Yo) + (RT(-0.272) * Uo) + (RT(-0.647) * Vo)) / 256);
    int b = (int)(((RT(1.000) * Yo) + (RT(-1.106) * Uo) + (RT(1.703) * Vo)) / 256);
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    px = ((uint32_t)r << 16) + ((uint32_t)g << 8) + (uint32_t)b;
    xo_out = xo;
    return true;
}

// dec_step for unrolled iteration j of the steady loop (j is a compile-time constant after
// unrolling, so the switch folds away)
template <bool VHS, bool COMPOUT, class O, int TP, class RT>
DEV bool dec_step_j(const DevParams &P, DecState<RT> &S, const DecConst &C, uint32_t *ring, int t,
                    int j, int pc, int pl, uint32_t &px, int &xo)
{
    if (TP < 0) return dec_step<VHS, COMPOUT, false, O, -1, -1, RT>(P, S, C, ring, t + j, pc, pl, px, xo);
    // PAR1 = (TP + j + 1) & 1, PAR2 = (j + 1) & 1
    if (((TP + j + 1) & 1) == 0) {
        if (((j + 1) & 1) == 0) return dec_step<VHS, COMPOUT, false, O, 0, 0, RT>(P, S, C, ring, t + j, pc, pl, px, xo);
        return dec_step<VHS, COMPOUT, false, O, 0, 1, RT>(P, S, C, ring, t + j, pc, pl, px, xo);
    }
    if (((j + 1) & 1) == 0) return dec_step<VHS, COMPOUT, false, O, 1, 0, RT>(P, S, C, ring, t + j, pc, pl, px, xo);
    return dec_step<VHS, COMPOUT, false, O, 1, 1, RT>(P, S, C, ring, t + j, pc, pl, px, xo);
}

// run-time constants of one lane's decode
struct DecRun {
    const int *comp;      // composite plane (uniform) and this lane's column
    int rc;
    const int *cbase;
    uint32_t *drow;
    uint32_t *ostage;
    bool is_out, any_hs;
    int hs, tw, SKT, LOFF;
};

// Steady-state loop of k_decode.  TP = parity of the pipeline depth SKT (-1: unknown).  With the
// preset options (output low-pass delay 1) position parities are: first demodulator x1 = t - 7
// -> (TP + j + 1) & 1; second demodulator x = t - 14 - d with d = SKT - 15 -> (j + 1) & 1.
template <bool VHS, bool COMPOUT, class O, int TP, class RT>
DEV int dec_steady(const DevParams &P, DecState<RT> &S, const DecConst &C, const DecRun &Rn,
                   uint32_t *ring, int t)
{
    const int W = C.W;
    const int lane = C.lane;
    const int *cbase = Rn.cbase;
    uint32_t *drow = Rn.drow;
    uint32_t *ostage = Rn.ostage;
    const bool is_out = Rn.is_out, any_hs = Rn.any_hs;
    const int hs = Rn.hs, tw = Rn.tw, SKT = Rn.SKT, LOFF = Rn.LOFF;
    auto cs = [&](int x) -> int {
        if (x < 0 || x >= W) return 0;
        int idx = x;
        if (hs != 0) {
            idx = x + hs;
            if (idx < 0) idx += tw; else if (idx >= tw) idx -= tw;
            if (idx >= W) return 0;
        }
        return cbase[(size_t)idx * P.Rpad];
    };
    {
        int pc[4], pl[4];
        const int t_end = W - 16;
        if (t + 4 <= t_end) {
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j] = cs(t + j); pl[j] = VHS ? cs(t + j - LOFF) : 0; }
        }
        for (; t + 4 <= t_end; t += 4) {
            int nc[4], nl[4];
            if (any_hs) {
                // some row of this wave is displaced by the head switch: per-lane index remap,
                // branch-free (x is inside the row here; |hs| < tw/2 so one wrap suffices)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int i0 = t + 4 + j + hs;
                    i0 += (i0 >> 31) & tw;
                    i0 -= (i0 >= tw) ? tw : 0;
                    const int v0 = cbase[(size_t)(i0 < W ? i0 : W - 1) * P.Rpad];
                    nc[j] = i0 < W ? v0 : 0;
                    if (VHS) {
                        int i1 = t + 4 + j - LOFF + hs;
                        i1 += (i1 >> 31) & tw;
                        i1 -= (i1 >= tw) ? tw : 0;
                        const int v1 = cbase[(size_t)(i1 < W ? i1 : W - 1) * P.Rpad];
                        nl[j] = i1 < W ? v1 : 0;
                    } else nl[j] = 0;
                }
            } else {
                // t + 7 < W and t + 4 - LOFF >= 0: plain coalesced loads
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    // uniform row base + per-lane column: SGPR base / VGPR offset addressing
                    nc[j] = (Rn.comp + (size_t)(t + 4 + j) * P.Rpad)[Rn.rc];
                    nl[j] = VHS ? (Rn.comp + (size_t)(t + 4 + j - LOFF) * P.Rpad)[Rn.rc] : 0;
                }
            }
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int xo;
                (void)dec_step_j<VHS, COMPOUT, O, TP, RT>(P, S, C, ring, t, j, pc[j], pl[j], o[j], xo);
                // keep the scheduler from interleaving whole pipeline steps: one step already has
                // ~10 independent filter chains, and mixing four of them costs >70 extra VGPRs
#ifndef NTSC_NO_STEP_SCHED_BARRIER
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            // stage 4 pixels; every 4th iteration write the lane's 16 pixels as one 64-byte burst
            {
                const int xo0 = t - SKT;                   // multiple of 4
                const int sub = (xo0 >> 2) & 3;
#ifdef NTSC_DIRECT_STORE
                (void)sub; (void)ostage;
                if (P.dst_al16) {
                    if (is_out) *reinterpret_cast<uint4 *>(drow + xo0) = make_uint4(o[0], o[1], o[2], o[3]);
                } else if (is_out) {
                    drow[xo0] = o[0]; drow[xo0 + 1] = o[1]; drow[xo0 + 2] = o[2]; drow[xo0 + 3] = o[3];
                }
#else
                if (P.dst_al16) {
                    *reinterpret_cast<uint4 *>(&ostage[lane * 20 + sub * 4]) =
                        make_uint4(o[0], o[1], o[2], o[3]);
                    if (sub == 3 && is_out) {
                        const uint4 *sp = reinterpret_cast<const uint4 *>(&ostage[lane * 20]);
                        uint4 *dp = reinterpret_cast<uint4 *>(drow + (xo0 - 12));
                        const uint4 a = sp[0], b = sp[1], c4 = sp[2], d4 = sp[3];
                        dp[0] = a; dp[1] = b; dp[2] = c4; dp[3] = d4;
                    }
                } else if (is_out) {
                    drow[xo0] = o[0]; drow[xo0 + 1] = o[1]; drow[xo0 + 2] = o[2]; drow[xo0 + 3] = o[3];
                }
#endif
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j] = nc[j]; pl[j] = nl[j]; }
        }
        // (a partly filled 16-pixel group stays staged; the epilogue keeps filling it)
    }
    return t;
}

// Occupancy: the VHS forms with the composite re-encode (two demodulators, 19 filter states, run-time
// switches) do not fit the 168 registers of 3 waves per SIMD without spilling inside the loop (99 / 7
// spilled registers for the PRESET / GENERIC form, and a spill reload waits on vmcnt(0), i.e. on the
// prefetched samples too); with 2 waves per SIMD they need no scratch, and two waves saturate a SIMD's
// fp64 pipe (tools/chain_probe.hip) -- the same trade as k_decode_fast.
#ifndef NTSC_DEC_WAVES
#define NTSC_DEC_WAVES 3
#endif
#ifndef NTSC_DEC_WAVES_VHS
#define NTSC_DEC_WAVES_VHS 2
#endif
template <bool VHS, bool COMPOUT, unsigned F, class RT>
__global__ __launch_bounds__(64, (VHS && COMPOUT) ? NTSC_DEC_WAVES_VHS : NTSC_DEC_WAVES) void k_decode(DevParams P, GeomDev G,
                                               const FieldDev *__restrict__ fields,
                                               const int *__restrict__ comp,
                                               const uint32_t *__restrict__ rs_chroma,
                                               const int *__restrict__ n0_u,
                                               const int *__restrict__ n0_v,
                                               const int *__restrict__ hs_shift,
                                               const int *__restrict__ pn_noise,
                                               const int *__restrict__ dropout,
                                               int *__restrict__ tails)
{
    __shared__ uint32_t ring[31 * 64];
    // output staging: 16 pixels per lane, row stride 20 dwords (16-byte aligned, b128 accesses)
#ifdef NTSC_DIRECT_STORE
    uint32_t *ostage = nullptr;
#else
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];
#endif

    const int lane = threadIdx.x;
    // 63 output rows per wave; lane 0 recomputes the row above (halo for the vertical blend)
    const int gidx = blockIdx.x * 63 + lane - 1;
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok;
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const int W = P.W;
    const int tw = W + W / 10;
    const int hs = P.hs ? hs_shift[rc] : 0;
    uint32_t *drow = reinterpret_cast<uint32_t *>(fd.dst + (size_t)fd.dst_ls * y);
    const int *cbase = comp + rc;
    // every lane of a wave owns a distinct column of the tails scratch (halo lanes included)
    const size_t tcol = (size_t)blockIdx.x * 64 + lane;
    const size_t tstride = (size_t)gridDim.x * 64;

    DecConst C;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;      // first even x with x+2 >= W
    C.k = k;
    C.lane = lane;
    C.d = VHS ? P.cdelay : 0;            // VHS chroma delay (9/12/14)
    using O = Opt<F>;
    C.dI = O::outlp(P) == 2 ? 2 : (O::outlp(P) == 1 ? 1 : 0);
    C.dQ = O::outlp(P) == 2 ? 4 : (O::outlp(P) == 1 ? 1 : 0);
    C.SKO = C.dQ;                        // output low-pass look-ahead
    C.vb = VHS && P.vblend && P.ntsc;
    C.drop = P.loss ? dropout[rc] != 0 : false;
    C.cosv = 1; C.sinv = 0;
    if (O::pnoise(P)) {
        // (slots past the last row of an odd-height field are never written by k_field_setup)
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = G.ptab[2 * n]; C.sinv = G.ptab[2 * n + 1];
    }
    C.tailU = tails + tcol;
    C.tailV = tails + 16 * tstride + tcol;
    C.rstride = tstride;

    DecState<RT> S;
    S.D1.init(); S.D2.init();
    S.l0 = S.l1 = S.l2 = 0;
    S.vl.reset(16); S.vpre.p = 16; S.vcU.reset(0); S.vcV.reset(0); S.sh.reset(0);
    S.oU.reset(0); S.oV.reset(0);
#pragma unroll
    for (int q = 0; q < 5; q++) { S.Yd[q] = 0; S.Ur[q] = 0; S.Vr[q] = 0; }
    S.Uf[0] = S.Uf[1] = S.Uf[2] = 0;
    S.nU = S.nV = 0;
    if (O::cnoise(P)) {
        S.rng.init(ring, rs_chroma + rc, P.Rpad, lane);
        S.nU = n0_u[rc]; S.nV = n0_v[rc];
    }

    // composite sample after head switching (:1687-1697): Y[x] = tmp[(x + shif) mod twidth],
    // tmp = the row followed by zeros up to twidth; zero outside the row.
    auto cs = [&](int x) -> int {
        if (x < 0 || x >= W) return 0;
        int idx = x;
        if (hs != 0) {
            idx = x + hs;
            if (idx < 0) idx += tw; else if (idx >= tw) idx -= tw;
            if (idx >= W) return 0;
        }
        return cbase[(size_t)idx * P.Rpad];
    };
    const bool any_hs = __any(hs != 0);

    const int d = C.d;
    const int SKT = 7 + d + (COMPOUT ? 7 : 0) + C.SKO;   // pipeline depth: xo = t - SKT
    const int total = W + SKT;
    const int LOFF = 5 + d;                                // luma stream reads cs(t - LOFF)

    int t = 0;
    // ---------------- prologue: fill the pipeline (guarded steps)
    for (; t < SKT && t < total; t++) {
        uint32_t px; int xo;
        (void)dec_step<VHS, COMPOUT, true, O, -1, -1, RT>(P, S, C, ring, t, cs(t), VHS ? cs(t - LOFF) : 0, px, xo);
    }
    // ---------------- steady state: every stage is strictly inside the row, 4 pixels per
    // iteration; ends 16 samples before the row end.  The PRESET kernels know the parity of every
    // demodulator position at compile time (one loop per parity of the pipeline depth).
    {
        DecRun Rn;
        Rn.comp = comp; Rn.rc = rc;
        Rn.cbase = cbase; Rn.drow = drow; Rn.ostage = ostage; Rn.is_out = is_out; Rn.hs = hs;
        Rn.tw = tw; Rn.SKT = SKT; Rn.LOFF = LOFF; Rn.any_hs = any_hs;
        if (O::generic) t = dec_steady<VHS, COMPOUT, O, -1, RT>(P, S, C, Rn, ring, t);
        else if (SKT & 1) t = dec_steady<VHS, COMPOUT, O, 1, RT>(P, S, C, Rn, ring, t);
        else t = dec_steady<VHS, COMPOUT, O, 0, RT>(P, S, C, Rn, ring, t);
    }
    // ---------------- epilogue: row end, filter tails, pipeline drain (guarded steps)
    for (; t < total; t++) {
        uint32_t px; int xo;
        if (!dec_step<VHS, COMPOUT, true, O, -1, -1, RT>(P, S, C, ring, t, cs(t), VHS ? cs(t - LOFF) : 0, px, xo))
            continue;
#ifdef NTSC_DIRECT_STORE
        if (is_out) drow[xo] = px;
        continue;
#endif
        if (!P.dst_al16) { if (is_out) drow[xo] = px; continue; }
        // same 16-pixel staging as the steady loop: whole 64-byte bursts, then the row's tail
        ostage[lane * 20 + (xo & 15)] = px;
        if ((xo & 15) == 15) {
            if (is_out) {
                const uint4 *sp = reinterpret_cast<const uint4 *>(&ostage[lane * 20]);
                uint4 *dp = reinterpret_cast<uint4 *>(drow + (xo - 15));
                const uint4 a = sp[0], b = sp[1], c4 = sp[2], d4 = sp[3];
                dp[0] = a; dp[1] = b; dp[2] = c4; dp[3] = d4;
            }
        } else if (xo == W - 1 && is_out) {
            const int xb = xo & ~15;
            for (int q = xb; q <= xo; q++) drow[q] = ostage[lane * 20 + (q - xb)];
        }
    }
}

// =============================================================================== k_ghost
// EXTENSION (not in the reference, default off): multipath ghosting of the composite signal,
// out[x] = in[x] + (sum_k gain_k * in[x - delay_k]) / 256, as a pass of its own -- the form for a delay of 64 samples
// or more (shorter ones are folded into the encoder: k_encode_fast_gh, ntsc_encode_fast.hip).  Pointwise over the
// transposed plane: a thread owns four neighbouring rows (one 16-byte piece of every plane row, Rpad is a multiple
// of 64) and NTSC_GHOST_XT consecutive positions; both planes are streamed (nt): each is far larger than the L2
// and next touched by another kernel.  A tap's piece was this pass's own in[x] delay positions earlier -- for a
// delay of a few hundred samples still in the Infinity Cache.
#define NTSC_GHOST_XT 8
__global__ __launch_bounds__(256) void k_ghost(DevParams P, const int *__restrict__ in, int *__restrict__ out)
{
    const int rho = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (rho >= P.R) return;
    const int x0 = blockIdx.y * NTSC_GHOST_XT;
    const int taps = P.ghost_taps;
    typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < NTSC_GHOST_XT; i++) {
        const int x = x0 + i;
        if (x >= P.W) break;
        v4i acc = {0, 0, 0, 0};
        for (int k = 0; k < taps; k++) {
            const int xs = x - P.ghost_delay[k];
            if (xs >= 0) acc += P.ghost_gain[k] * __builtin_nontemporal_load((const v4i *)(in + (size_t)xs * P.Rpad + rho));
        }
        const v4i v = __builtin_nontemporal_load((const v4i *)(in + (size_t)x * P.Rpad + rho));
        __builtin_nontemporal_store(v + acc / 256, (v4i *)(out + (size_t)x * P.Rpad + rho));
    }
}

// =============================================================================== k_bob
// Line doubling done by the field loop after composite_layer (ffmpeg_ntsc.cpp:2233-2257):
// field 1: row y (odd) is copied onto row y-1; field 0: row y+1 is copied onto odd row y while
// y+1 < H.  One workgroup per (field slot, destination row); 16-byte coalesced copies.
__global__ void k_bob(DevParams P, const FieldDev *__restrict__ fields)
{
    const int f = blockIdx.y;
    const FieldDev &fd = fields[f];
    if (!(fd.flags & 0x100u)) return;
    const unsigned field = fd.field & 1u;
    const int yo = 2 * blockIdx.x + 1;            // odd row index
    int ysrc, ydst;
    if (field) { ysrc = yo; ydst = yo - 1; if (ysrc >= P.H) return; }
    else       { ysrc = yo + 1; ydst = yo; if (ysrc >= P.H) return; }
    const uint32_t *s = reinterpret_cast<const uint32_t *>(fd.dst + (size_t)fd.dst_ls * ysrc);
    uint32_t *o = reinterpret_cast<uint32_t *>(fd.dst + (size_t)fd.dst_ls * ydst);
    if (P.dst_al16 && (P.W & 3) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
        uint4 *o4 = reinterpret_cast<uint4 *>(o);
        for (int i = threadIdx.x; i < P.W / 4; i += blockDim.x) o4[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < P.W; i += blockDim.x) o[i] = s[i];
    }
}

// =============================================================================== k_bgra_to_yuv
// SURVEY 8(f) row f2, the output side: BGRA -> planar YUV 4:2:0 / 4:2:2 for the encoder (what the
// tool does with sws_scale at ffmpeg_ntsc.cpp:2266).  libswscale is not part of the reference
// tree, so this is NOT a bit-clone of it: it is BT.601 limited range in the classic 15-bit
// fixed-point form (Y = 16 + 219/255 * (0.299 R + 0.587 G + 0.114 B), Cb/Cr = 128 + 224/255 * ...),
// chroma taken from the rounded mean of the 2x1 (4:2:2) or 2x2 (4:2:0) block.  Parity unpinned.
struct YuvDev {
    const uint8_t *bgra;
    uint8_t *y, *u, *v;
    int32_t bgra_ls, y_ls, u_ls, v_ls;
};
namespace yuvc {
constexpr int RY = 8414, GY = 16519, BY = 3208;        // (int)(c * 219 / 255 * 32768 + 0.5)
constexpr int RU = -4864, GU = -9527, BU = 14392;      // (int)(c * 224 / 255 * 32768 + 0.5)
constexpr int RV = 14392, GV = -12060, BV = -2331;
}
DEV uint32_t yuv_luma(uint32_t px)
{
    const int b = px & 255u, g = (px >> 8) & 255u, r = (px >> 16) & 255u;
    return (uint32_t)(yuvc::RY * r + yuvc::GY * g + yuvc::BY * b + (16 << 15) + (1 << 14)) >> 15;
}
// chroma of a block of n = 1 << lg pixels whose channel sums are rs, gs, bs
DEV void yuv_chroma(int rs, int gs, int bs, int lg, uint32_t &u, uint32_t &v)
{
    const int sh = 15 + lg;
    u = (uint32_t)(yuvc::RU * rs + yuvc::GU * gs + yuvc::BU * bs + (128 << sh) + (1 << (sh - 1))) >> sh;
    v = (uint32_t)(yuvc::RV * rs + yuvc::GV * gs + yuvc::BV * bs + (128 << sh) + (1 << (sh - 1))) >> sh;
}
DEV void yuv_acc(uint32_t px, int &rs, int &gs, int &bs)
{
    bs += px & 255u; gs += (px >> 8) & 255u; rs += (px >> 16) & 255u;
}

// One thread = 8 pixels of one row (4:2:2) or of a row pair (4:2:0); v420: 1 = 4:2:0.
__global__ void k_bgra_to_yuv(DevParams P, const YuvDev *__restrict__ frames, int v420, int vec)
{
    const YuvDev &f = frames[blockIdx.z];
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (x0 >= P.W) return;
    const int y0 = v420 ? 2 * (int)blockIdx.y : (int)blockIdx.y;
    const int y1 = (v420 && y0 + 1 < P.H) ? y0 + 1 : y0;          // odd height: last row twice
    const int nrows = v420 ? 2 : 1;
    const int npx = P.W - x0 < 8 ? P.W - x0 : 8;                  // W is even
    uint32_t px[2][8];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r >= nrows) break;
        const uint8_t *row = f.bgra + (size_t)f.bgra_ls * (r ? y1 : y0) + (size_t)x0 * 4;
        if (vec && npx == 8) {
            const uint4 a = reinterpret_cast<const uint4 *>(row)[0], b = reinterpret_cast<const uint4 *>(row)[1];
            px[r][0] = a.x; px[r][1] = a.y; px[r][2] = a.z; px[r][3] = a.w;
            px[r][4] = b.x; px[r][5] = b.y; px[r][6] = b.z; px[r][7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                px[r][i] = i < npx ? reinterpret_cast<const uint32_t *>(row)[i] : 0u;
        }
    }
    // luma
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (r >= nrows || (r == 1 && y1 == y0)) break;
        uint8_t *yrow = f.y + (size_t)f.y_ls * (r ? y1 : y0) + x0;
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { w0 |= yuv_luma(px[r][i]) << (8 * i); w1 |= yuv_luma(px[r][4 + i]) << (8 * i); }
        if (vec && npx == 8) *reinterpret_cast<uint2 *>(yrow) = make_uint2(w0, w1);
        else
            for (int i = 0; i < npx; i++) yrow[i] = (uint8_t)((i < 4 ? w0 >> (8 * i) : w1 >> (8 * (i - 4))) & 255u);
    }
    // chroma
    uint32_t uw = 0, vw = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int rs = 0, gs = 0, bs = 0;
        yuv_acc(px[0][2 * c], rs, gs, bs); yuv_acc(px[0][2 * c + 1], rs, gs, bs);
        if (v420) { yuv_acc(px[1][2 * c], rs, gs, bs); yuv_acc(px[1][2 * c + 1], rs, gs, bs); }
        uint32_t u, v;
        yuv_chroma(rs, gs, bs, v420 ? 2 : 1, u, v);
        uw |= u << (8 * c); vw |= v << (8 * c);
    }
    const int cy = blockIdx.y;
    uint8_t *urow = f.u + (size_t)f.u_ls * cy + x0 / 2, *vrow = f.v + (size_t)f.v_ls * cy + x0 / 2;
    if (vec && npx == 8) {
        *reinterpret_cast<uint32_t *>(urow) = uw;
        *reinterpret_cast<uint32_t *>(vrow) = vw;
    } else {
        for (int c = 0; c < npx / 2; c++) { urow[c] = (uint8_t)((uw >> (8 * c)) & 255u); vrow[c] = (uint8_t)((vw >> (8 * c)) & 255u); }
    }
}

} // namespace ntscsim
