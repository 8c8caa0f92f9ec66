// ntsc_decode_fast.hip -- the decoder of the two presets that matter for throughput (the default
// preset and the full `-vhs` family), written instruction by instruction for the gfx950 VALU.
// Same execution model and the same results as k_decode (ntsc_kernels.hip): one lane = one
// scanline, 63 rows + 1 halo row per wave, all stages streamed in x.  What differs is the cost:
// measured on MI355X (tools/valu_rate_probe.hip) every fp64 instruction and most integer opcodes
// (v_cndmask, v_lshl*, v_add3, v_med3, v_mul_*) occupy a SIMD for ~4.3 cycles per wave, only
// add/sub/and/xor/shift-right/mov run at ~2.7, and the decoder sits at ~75 % of that VALU roof --
// so the only way to go faster is to issue less.  This kernel
//   * carries each one-pole filter as r = p - p*a, the term the NEXT step adds to its input
//     product, and feeds pole k+1 (same alpha) with the product p_k*a that pole k needs anyway:
//     3 fp64 instructions per pole instead of 4 (same operations on the same operands, so the
//     same bits: ffmpeg_ntsc.cpp:90-99);
//   * keeps the demodulator windows un-negated and applies the sign when a sample is picked
//     (ffmpeg_ntsc.cpp:1539-1561: for an even scanline phase the picked sample is negated iff
//     x = 1 mod 4, which is a compile-time property of the unrolled loop position);
//   * replaces per-lane selects by mask arithmetic on the full-rate opcodes, the LDS
//     __shfl_up of the vertical blend by a DPP wave shift, 64-bit address arithmetic by buffer
//     loads whose bounds check also implements the zero fill of the head-switch displacement
//     (ffmpeg_ntsc.cpp:1687-1697) for free.
// Preconditions (checked by the launcher, otherwise k_decode runs): even scanline phase for every
// row (-comp-phase 180 with an even offset), subcarrier amplitude 50 both ways, output low-pass
// "lite", 16-byte aligned destination rows, composite plane (plus the head-switch displacement) within
// 32-bit buffer offsets; VHS form: chroma noise + phase noise on, composite (not s-video) out.  Head-switch
// displacements beyond W/10 samples (wrap-around inside the 1.1 W window) run the WR instantiation.
#pragma clang fp contract(off)

namespace ntscsim {
namespace fastdec {

// Frame pointers come out of the FieldDev records as generic pointers, for which hipcc emits FLAT
// loads / stores.  A FLAT access counts on lgkmcnt as well as vmcnt, so every LDS wait behind it
// (the rand() ring, the pixel staging) also waits for the HBM round trip.  Frames are global
// memory: say so, and the accesses become global_load / global_store (vmcnt only).
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
// Cache hints (round 4, tools/nt_probe.sh, profiles/r04_nt_probe.txt): the output rows leave as 64-byte halves of
// 128-byte lines, 16 pixels per iteration; as plain stores the halves wait in the L2 for each other, crowd out the
// composite samples the VCR's luma path re-reads 5 + d positions behind the chroma path, and a fifth of them is
// written back twice.  As streaming (nt) stores: WRITE_SIZE of k_decode_fast 1.28 -> 1.05 x the output, FETCH_SIZE
// -11 %, and -17 % with the re-read itself marked nt (its line is dead afterwards).
#ifdef NTSC_PLAIN_OUT_STORES        /* A/B: the round-3 stores */
#define NTSC_OUT_STORE(p, v) (*(p) = (v))
#else
#define NTSC_OUT_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif
typedef __attribute__((address_space(1))) v4u *g_v4u_ptr;
typedef __attribute__((address_space(1))) const v4u *g_cv4u_ptr;
DEV v4u to_v4u(const uint4 &a) { return v4u{a.x, a.y, a.z, a.w}; }
typedef __attribute__((address_space(1))) uint32_t *g_u32_ptr;
typedef __attribute__((address_space(1))) const uint32_t *g_cu32_ptr;

#ifdef NTSC_NO_STEP_SCHED_BARRIER
#define NTSC_STEP_SCHED_BARRIER() ((void)0)
#else
// keeps the scheduler from interleaving whole pipeline steps (costs registers, gains nothing)
#define NTSC_STEP_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

// ------------------------------------------------------------------ rand() ring with static LDS offsets
// LaneRand (ntsc_kernels.hip) keeps the 31-word window of glibc's r[i] = r[i-31] + r[i-3] in a 31-slot ring and needs
// one address (slot << 8 | lane) per draw.  Here the ring has 32 slots plus a copy of slot 0 behind slot 31: the word
// for draw i is written to slot i & 31, and r[i-31] sits in slot (i + 1) & 31 = the NEXT slot, or the copy when the
// write slot is 31 -- so read and write of a draw are two immediate offsets from one address, and the eight draws of an
// unrolled steady iteration are sixteen immediate offsets from the address of that iteration's first slot (a multiple
// of 8: init() places the window so that the pipeline fill ends on one).  One address per iteration instead of eight.
struct LaneRand32 {
    uint32_t p3, p2, p1;   // r[i-3], r[i-2], r[i-1]
    int pos;               // wave-uniform: slot of the next draw's word
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
    DEV uint32_t next(uint32_t *ring, int lane)               // any position
    {
        const uint32_t v = ring[(pos + 1) * 64 + lane] + p3;  // r[i-31] + r[i-3]
        ring[pos * 64 + lane] = v;
        if (pos == 0) ring[32 * 64 + lane] = v;
        p3 = p2; p2 = p1; p1 = v;
        pos = (pos + 1) & 31;
        return v >> 1;
    }
    // draw K of a steady iteration whose first slot is rb's (a multiple of 8 slots for 8 draws per iteration, of 16 for 16:
    // slot K + 1 <= 32 must exist)
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

// ------------------------------------------------------------------ filters
// three one-pole low-passes with ONE alpha (LowpassFilter x3, ffmpeg_ntsc.cpp:1399-1427 etc.)
template <class RT>
struct Casc3;
template <>
struct Casc3<double> {
    double r0, r1, r2;       // r_k = p_k - (p_k * a)
    DEV void reset(double v, double a) { const double r = v - (v * a); r0 = r1 = r2 = r; }
    // s = input sample.  Returns the output p2; m2 = p2 * a for a following same-alpha stage.
    DEV double push(double s, double a, double &m2)
    {
        const double p0 = (s * a) + r0;
        const double m0 = p0 * a;
        r0 = p0 - m0;
        const double p1 = m0 + r1;
        const double m1 = p1 * a;
        r1 = p1 - m1;
        const double p2 = m1 + r2;
        m2 = p2 * a;
        r2 = p2 - m2;
        return p2;
    }
    DEV double push(double s, double a) { double m; return push(s, a, m); }
};
template <>
struct Casc3<float> {       // FAST32 mode: p += a * (s - p), tests/test_gpu_fast_mode.py
    float p0, p1, p2;
    DEV void reset(float v, float) { p0 = p1 = p2 = v; }
    DEV float push(float s, float a, float &m2)
    {
        p0 = __builtin_fmaf(a, s - p0, p0);
        p1 = __builtin_fmaf(a, p0 - p1, p1);
        p2 = __builtin_fmaf(a, p1 - p2, p2);
        m2 = 0.f;
        return p2;
    }
    DEV float push(float s, float a) { float m; return push(s, a, m); }
};
// one more pole of the same alpha behind a cascade, used as a high-pass (s - lowpass(s))
template <class RT>
struct PoleHp;
template <>
struct PoleHp<double> {
    double r;
    DEV void reset(double v, double a) { r = v - (v * a); }
    DEV double hp(double s, double s_times_a, double a)
    {
        const double p = s_times_a + r;
        r = p - (p * a);
        return s - p;
    }
};
template <>
struct PoleHp<float> {
    float p;
    DEV void reset(float v, float) { p = v; }
    DEV float hp(float s, float, float a) { p = __builtin_fmaf(a, s - p, p); return s - p; }
};

// C `/ 4` (truncating) for |n| < 2^30: the two top bits of a negative n are 11
DEV int sdiv4s(int n) { return (n + (int)((unsigned)n >> 30)) >> 2; }

// (c * 50) / subcarrier_amplitude_back, ffmpeg_ntsc.cpp:1544-1546 (C division: truncating, odd in c), for an
// amplitude other than 50 (the pre-emphasis presets raise it): magnitude through the 31-bit magic multiplier,
// sign put back with a mask.  |c| * 50 < 2^31 for every composite sample the encoder can make.
DEV int scale_back50(int c, unsigned mul, unsigned shift)
{
    const int sg = c >> 31;
    const unsigned a = (unsigned)((c ^ sg) - sg) * 50u;
    const int q = (int)(__umulhi(a, mul) >> shift);
    return (q ^ sg) - sg;
}

// ------------------------------------------------------------------ Y/C separation, raw windows
// chroma_from_luma (ffmpeg_ntsc.cpp:1497-1567) with the half-cycle flip (:1539-1542) applied when
// a sample is picked instead of when it is stored.  Valid for EVEN scanline phase xi in {0, 2}:
// the I sample for odd x is raw(x+1+xi), the Q sample raw(x+2+xi), both negated iff x = 3 mod 4
// (the reference's double negation :1539 / :1552-1553 leaves +raw at x = 1 mod 4).
struct DemodR {
    int c0, c1, c2, csum;         // cs(t-3), cs(t-2), cs(t-1) and their sum
    int w0, w1, w2, w3, w4, w5;   // raw chroma at q-5 .. q   (q = t-2)
    int y0, y1, y2, y3, y4;       // box-filtered luma at q-5 .. q-1
    int ieP, qeP, ieN, qeN;
    DEV void init()
    {
        c0 = c1 = c2 = csum = 0;
        w0 = w1 = w2 = w3 = w4 = w5 = 0;
        y0 = y1 = y2 = y3 = y4 = 0;
        ieP = qeP = ieN = qeN = 0;
    }
    // steady state: every position inside the row.  ODD = x is odd; NEGS = 0 / -1 (x = 1 / 3 mod 4)
    // as a compile-time (NEG >= 0) or wave-uniform (sneg) value; hi = lane mask of xi == 2.
    // BK: the chroma sample is scaled by 50 / subcarrier_amplitude_back (bmul, bshift: its magic multiplier)
    template <bool ODD, int NEG, bool LUMA, bool BK = false>
    DEV void push(int ct, bool hi, int sneg, int &Yo, int &Io, int &Qo, unsigned bmul = 0, unsigned bshift = 0)
    {
        const int yb = sdiv4s(csum + ct);
        int ch = ct - yb;
        if (BK) ch = scale_back50(ch, bmul, bshift);
        csum = csum - c0 + ct;
        c0 = c1; c1 = c2; c2 = ct;
        w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = ch;
        if (LUMA) { Yo = y0; y0 = y1; y1 = y2; y2 = y3; y3 = y4; y4 = yb; }
        if (ODD) {
            const int a = hi ? w3 : w1, b = hi ? w4 : w2;
            if (NEG == 0) { ieN = a; qeN = b; }
            else if (NEG == 1) { ieN = -a; qeN = -b; }
            else { ieN = (a ^ sneg) - sneg; qeN = (b ^ sneg) - sneg; }
            Io = (ieP + ieN) >> 1;
            Qo = (qeP + qeN) >> 1;
        } else {
            Io = ieN; Qo = qeN;
            ieP = ieN; qeP = qeN;
        }
    }
    // any position (row ends, pipeline fill and drain); t is wave-uniform, xi per lane
    template <bool BK = false, bool ANY = false>
    DEV void push_edge(int ct, int t, unsigned xi, bool hi, int W, int xe, int &Yo, int &Io, int &Qo,
                       unsigned bmul = 0, unsigned bshift = 0)
    {
        const int yb = sdiv4(c0 + c1 + c2 + ct);
        int ch = ct - yb;
        if (BK) ch = scale_back50(ch, bmul, bshift);
        csum = c1 + c2 + ct;
        c0 = c1; c1 = c2; c2 = ct;
        w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = ch;
        Yo = y0; y0 = y1; y1 = y2; y2 = y3; y3 = y4; y4 = yb;
        const int x = t - 7;
        int I, Q;
        if (x & 1) {
            const bool m = (x + 1 + (int)xi + 1) < W;                    // :1550
            // (after the shift w5 = ch(q), ..., w1 = ch(q-4); the pick is ch(q - 4 + xi), ch(q - 3 + xi); ANY: xi may be odd)
            const bool od = ANY && (xi & 1u) != 0;
            const int a = hi ? (od ? w4 : w3) : (od ? w2 : w1), b = hi ? (od ? w5 : w4) : (od ? w3 : w2);
            // flipped (:1539-1542) and then negated (:1550-1553) = positive iff (xe + 2 xi) & 2, xe = x + 1 (even xi: iff
            // x = 1 mod 4) -- except the row's first pair with xi = 1: the flip loop starts at x = (4 - xi) & 3 = 3, so
            // chroma[1], chroma[2] keep their sign although the pattern would flip them
            const bool pos = ANY ? ((((x + 1 + 2 * (int)xi) & 2) != 0) && !(xi == 1u && x == -1)) : ((x & 3) == 1);
            ieN = m ? (pos ? a : -a) : 0;
            qeN = m ? (pos ? b : -b) : 0;
            I = (ieP + ieN) >> 1;
            Q = (qeP + qeN) >> 1;
        } else {
            I = ieN; Q = qeN;
            ieP = ieN; qeP = qeN;
        }
        if (x >= xe) { I = 0; Q = 0; }                                   // :1553-1556, :1562-1565
        Io = I; Qo = Q;
    }
};

// The same separator in the form the STEADY loop runs (round 4: the instruction diet, profiles/r04_decode_census.txt).
// Two things change, neither touches a result:
//   * the four-tap box sum is kept as pair sums p(t) = c(t) + c(t-1): box(t) = p(t) + p(t-2).  One add less per
//     sample than "running sum - oldest + newest", and the delay lines are 1 and 2 deep instead of 3 -- under the
//     4x unrolled loop a 2-deep line is pure register renaming, a 3-deep one costs three v_mov at every back edge;
//   * the raw chroma window is split by position parity.  Picks happen at every second position and take either
//     (ch(q-4), ch(q-3)) or (ch(q-2), ch(q-1)): one sample of the pick's own parity and one of the other, each 1 or 2
//     positions back IN ITS OWN PARITY STREAM.  Two 2-deep lines (e: the pick parity, o: the other) replace one
//     6-deep window -- again renaming instead of six moves per iteration.
// from(): the state the guarded steps (DemodR) left behind; `pick_next`: the next position is a pick (odd x).
// to(): back into DemodR's layout for the guarded steps of the row end.
struct DemodS {
    int c1, pA, pB;               // c(t-1); p(t-1) = c(t-1)+c(t-2); p(t-2)
    int e1, e2, o1, o2;           // raw chroma of the last two positions of the pick parity / of the other parity
    int y0, y1, y2, y3, y4;       // box-filtered luma at q-5 .. q-1
    int ieP, qeP, ieN, qeN;
    DEV void from(const DemodR &D, bool pick_next)
    {
        c1 = D.c2; pA = D.c2 + D.c1; pB = D.c1 + D.c0;
        // before the push at q: w5 = ch(q-1), w4 = ch(q-2), w3 = ch(q-3), w2 = ch(q-4)
        if (pick_next) { e1 = D.w4; e2 = D.w2; o1 = D.w5; o2 = D.w3; }
        else           { e1 = D.w5; e2 = D.w3; o1 = D.w4; o2 = D.w2; }
        y0 = D.y0; y1 = D.y1; y2 = D.y2; y3 = D.y3; y4 = D.y4;
        ieP = D.ieP; qeP = D.qeP; ieN = D.ieN; qeN = D.qeN;
    }
    DEV void to(DemodR &D, bool pick_next) const
    {
        D.c2 = c1; D.c1 = pA - c1; D.c0 = pB - D.c1; D.csum = D.c0 + D.c1 + D.c2;
        D.w0 = 0; D.w1 = 0;       // (never picked again: a pick reads at most five positions back)
        if (pick_next) { D.w4 = e1; D.w2 = e2; D.w5 = o1; D.w3 = o2; }
        else           { D.w5 = e1; D.w3 = e2; D.w4 = o1; D.w2 = o2; }
        D.y0 = y0; D.y1 = y1; D.y2 = y2; D.y3 = y3; D.y4 = y4;
        D.ieP = ieP; D.qeP = qeP; D.ieN = ieN; D.qeN = qeN;
    }
    // PICK = x is odd; NEG = the picked pair is negated (x = 3 mod 4), compile time; hi = lane mask of xi == 2;
    // dm = and-mask applied to the picked pair (dropout :1891-1901 folded into the pick: everything the separator
    // puts out afterwards is an average or a copy of picked values; -1 where there is no dropout stage behind it)
    // ANY: the scanline phase xi may be odd as well (-comp-phase 90 / 270, odd offsets; per lane).  The pick for the
    // even position xe = x + 1 is raw chroma (xe + xi, xe + xi + 1) = the samples pushed (4 - xi, 3 - xi) positions ago
    // -- for odd xi one position later than for xi - 1, i.e. out of the OTHER parity stream, the newest one being this
    // step's own sample -- and the half-cycle flip :1539-1542 leaves it positive iff (xe + 2 xi) & 2: the even-phase sign
    // (NEG) inverted on odd lanes (`mo`: -1 where xi is odd).
    template <bool PICK, bool NEG, bool LUMA, bool BK = false, bool MASK = false, bool ANY = false>
    DEV void push(int ct, bool hi, int dm, int &Yo, int &Io, int &Qo, unsigned bmul = 0, unsigned bshift = 0,
                  bool odd = false, int mo = 0)
    {
        const int p = ct + c1;
        const int yb = sdiv4s(p + pB);
        int ch = ct - yb;
        if (BK) ch = scale_back50(ch, bmul, bshift);
        c1 = ct; pB = pA; pA = p;
        if (LUMA) { Yo = y0; y0 = y1; y1 = y2; y2 = y3; y3 = y4; y4 = yb; }
        if (PICK) {
            int a, b;
            if (ANY) {
                const int aL = odd ? o2 : e2, aH = odd ? o1 : e1;
                const int bL = odd ? e1 : o2, bH = odd ? ch : o1;
                a = hi ? aH : aL; b = hi ? bH : bL;
                const int sg = NEG ? ~mo : mo;
                a = (a ^ sg) - sg; b = (b ^ sg) - sg;
            } else {
                a = hi ? e1 : e2; b = hi ? o1 : o2;
                if (NEG) { a = -a; b = -b; }
            }
            e2 = e1; e1 = ch;
            if (MASK) { a &= dm; b &= dm; }
            ieN = a; qeN = b;
            Io = (ieP + ieN) >> 1;
            Qo = (qeP + qeN) >> 1;
        } else {
            o2 = o1; o1 = ch;
            Io = ieN; Qo = qeN;
            ieP = ieN; qeP = qeN;
        }
    }
};

template <bool VHS, class RT>
struct State {
    DemodR D1, D2;
    int l0, l1, l2, lsum;                     // luma stream window (VHS)
    Casc3<RT> vl, vcU, vcV, sh, oU, oV;
    PoleHp<RT> vpre;
    int Yprev, Uraw, Vraw;                    // previous step's output-stage inputs
    // FO (full output low-pass), guarded steps: the filtered I of the last two positions (luma and raw chroma of the
    // last four live in LDS, Const::xs: registers are what this form has least of)
    RT Uf2[2];
    LaneRand32 rng;
    int nU, nV;
};

// per-lane / per-launch constants.  WR: the head-switch displacement may wrap around the 1.1 W window
// (address fix-up per load, see cs_load)
template <class RT, bool WR = false, bool BK = false, bool SV = false, bool XA = false, bool FO = false>
struct Const {
    // FO: the FULL output chroma low-pass (composite_lowpass :1429-1458 behind the VCR: -out-composite-lowpass-lite 0):
    // I at 1.3 MHz landing 2 samples back, Q at 0.6 MHz landing 4 back, instead of the TV filter (2.6 MHz, 1 back, both).
    // The pixel for position xo leaves at x3 = xo + 4 (3 steps later than with the TV filter): luma waits 4 steps, the
    // filtered I 2 -- both delay lines a divisor of the 4x unroll -- and every position phase of the TV half moves by
    // SH = 3 (the phases of the steady step are written in terms of it).
    static constexpr bool fullout = FO;
    static constexpr int SH = FO ? 3 : 0;
    RT a_oi, a_oq;            // FO: alphas of the I / Q output filters
    uint32_t *xs;             // FO: this lane's column of 12 LDS slots -- luma, raw U, raw V of position p at slot (p & 3) (+4, +8)
    static constexpr bool wraps = WR;
    static constexpr bool back = BK;
    static constexpr bool svideo = SV;     // VHS form with S-Video out: no re-modulation, no second separation
    static constexpr bool anyxi = XA;      // scanline phases of either parity (per lane): -comp-phase 90 / 270, odd offsets
    bool odd;                 // XA: xi is odd
    int mo;                   // XA: -1 where xi is odd
    int ms[4];                // XA: sign of the re-modulated chroma at unrolled position J: -1 where (xi + J) & 2
    unsigned bmul, bshift;    // BK: magic multiplier of subcarrier_amplitude_back
    int wrapoff;              // WR: byte offset of the wrapped index, -tw or +tw samples (sign of the shift)
    int wrapA, wrapS;         // WR: x wraps iff ((wrapA - x) ^ wrapS) < 0
    unsigned xi;
    bool hi;                  // xi == 2
    int W, xe, lane;
    int d, SKT, LOFF;
    int mL, mNL;              // -1 / 0 masks of hi and !hi (sign of the re-modulated chroma)
    int bA, bC;               // vertical blend: and-mask of the row above, carry/shift (0: blend off)
    int dm;                   // dropout and-mask (0 = this row's chroma is dropped)
    RT cosv, sinv;
    RT a_vc, a_vl, a_sh, a_tv, sharp2;
    int *tailU;               // this lane's column of the raw-chroma tail scratch: U at rows 0..15, V at rows 16..31
    size_t rstride;
    __amdgpu_buffer_rsrc_t comp;   // the whole composite plane; out-of-range reads return 0
    int vbase;                // byte offset of this lane's column (+ head-switch displacement)
    int rowbytes;             // bytes between consecutive x
};

// composite sample x of this lane's row after head switching, for 0 <= x < W (the caller's duty:
// the bounds check only covers the displaced index).  ffmpeg_ntsc.cpp:1687-1697: Y[x] = row[(x + shift)
// mod tw] where that index is below W, else 0, with tw = W + W/10 and |shift| <= tw/2.  Without wrap
// (|shift| <= W/10) the displaced index is x + shift and one bounds-checked load does it.  With wrap it
// is x + shift - tw from x = tw - shift on (shift > 0), x + shift + tw below x = -shift (shift < 0):
// one sign test per load, as mask arithmetic on full-rate opcodes, moves the offset by -+ tw samples.
#ifndef NTSC_COMP_LOAD2_AUX
#define NTSC_COMP_LOAD2_AUX 2       /* nt on the VCR luma path's (second, trailing) read of every sample; 0 = plain (A/B) */
#endif
template <int AUX = 0, class CT>
DEV int cs_load(const CT &C, int x)
{
#ifdef NTSC_AB_NOLOAD      // timing-only A/B build (WRONG pixels): no composite loads
    return C.vbase + x;
#endif
    unsigned off = (unsigned)C.vbase + (unsigned)x * (unsigned)C.rowbytes;
    if constexpr (CT::wraps) off += (unsigned)((((C.wrapA - x) ^ C.wrapS) >> 31) & C.wrapoff);
    return __builtin_amdgcn_raw_buffer_load_b32(C.comp, (int)off, 0, AUX);
}

// Masks built from per-lane / wave-uniform booleans are laundered through an empty asm so that the
// compiler cannot turn the full-rate and / xor / sub back into half-rate v_cndmask selects.
DEV int opaque_v(int v) { asm volatile("" : "+v"(v)); return v; }
DEV int opaque_s(int v) { return __builtin_amdgcn_readfirstlane(v); }   // (also proves uniformity)

DEV int wave_up(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1

DEV uint32_t cvt_sat_u32(double x) { uint32_t r; asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(x)); return r; }
DEV uint32_t cvt_sat_u32(float x) { uint32_t r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
DEV uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

template <class RT>
DEV uint32_t yiq_to_bgra(int Yo, RT fU, RT fV)
{
    // YIQ_to_RGB :1385-1396: (int)(x / 256) then clamp; for the clamp's sake (int)x >> 8 is the
    // same (a negative quotient clamps to 0 whichever way it was rounded)
    const RT y = (RT)Yo;
#ifdef NTSC_OLD_CLAMP
    int r = (int)((y + (RT(0.956) * fU)) + (RT(0.621) * fV)) >> 8;
    int g = (int)((y + (RT(-0.272) * fU)) + (RT(-0.647) * fV)) >> 8;
    int b = (int)((y + (RT(-1.106) * fU)) + (RT(1.703) * fV)) >> 8;
    r = r < 0 ? 0 : (r > 255 ? 255 : r);
    g = g < 0 ? 0 : (g > 255 ? 255 : g);
    b = b < 0 ? 0 : (b > 255 ? 255 : b);
    return (((uint32_t)r << 16) | (uint32_t)b) | ((uint32_t)g << 8);   // two v_lshl_or_b32
#else
    // the lower clamp is the conversion's own: v_cvt_u32 saturates (negative -> 0, too large -> 2^32 - 1), so
    // clamp((int)x >> 8, 0, 255) == min((unsigned)x >> 8, 255) for every x -- a full-rate v_min_u32 per channel in
    // place of a half-rate v_med3_i32 (C's conversion of a negative double to unsigned is undefined, hence the asm)
    const uint32_t r = umin32(cvt_sat_u32((y + (RT(0.956) * fU)) + (RT(0.621) * fV)) >> 8, 255u);
    const uint32_t g = umin32(cvt_sat_u32((y + (RT(-0.272) * fU)) + (RT(-0.647) * fV)) >> 8, 255u);
    const uint32_t b = umin32(cvt_sat_u32((y + (RT(-1.106) * fU)) + (RT(1.703) * fV)) >> 8, 255u);
    return ((r << 16) | b) | (g << 8);
#endif
}

// One steady-state pipeline step at unrolled position J (t = SKT + 4n + J).  Position phases:
//   second demodulator / output   x3 = t - 14 - d = 4n + J + 1     -> odd for even J, sign by J
//   re-modulation                 x2 = x3 + 7     = 4n + J (mod 4) -> U for even J, sign by J & 2
//   first demodulator             x1 = t - 7      = d + J (mod 4)  -> parity by DP = d & 1 (template),
//                                                                     sign wave-uniform (sneg1)
// Non-VHS form: x3 = x1 = t - 7 = 4n + J + 1 (SKT = 8), one demodulator.
// The VCR half of a steady step (VHS form): first demodulator at x1 = t - 7, chroma noise, phase
// noise, VHS chroma / luma filters, vertical blend, re-modulation.  Returns the composite sample
// the VCR puts out at x2 = x1 - d (ffmpeg_ntsc.cpp:1716-1888).
// (Yv, Uv, Vv: the same signal as components, what the VCR's S-Video connector carries.)
// What the steady loop keeps beside State: the separators and the luma box in their steady form (DemodS), and the LDS
// address of the iteration's first rand() slot.
struct Steady {
    DemodS D1, D2;
    int lc1, lpA, lpB;          // VHS luma stream: c(t-1) and the pair sums (as DemodS)
    uint32_t *rb;               // this lane's column of the iteration's first ring slot
    bool rb0;                   // that slot is slot 0 (its copy behind slot 31 is written too)
    int yd[4];                  // FO: luma of the last four positions, slot J rewritten at unrolled position J
    int ud[2];                  // FO: filtered I of the last two positions (already truncated: kept as integers)
};

// DPH = the first separator's position phase: x1 = t - 7 = DPH + J (mod 4) -- odd positions pick, x1 = 3 (mod 4)
// negates (a compile-time property of the unrolled position since round 4: one instantiation per chroma delay mod 4).
template <int DPH, int J, class RT, class CT>
DEV int vcr_step(const DevParams &P, State<true, RT> &S, Steady &T, const CT &C,
                 int pc, int pl, int &Yv, int &Uv, int &Vv)
{
    constexpr bool pick1 = ((DPH + J) & 1) != 0;
    constexpr bool neg1 = ((DPH + J) & 3) == 3;
    int Yd, U, V;
    T.D1.template push<pick1, neg1, false, CT::back, false, CT::anyxi>(pc, C.hi, -1, Yd, U, V, C.bmul, C.bshift, C.odd, C.mo);
    // chroma noise :1719-1735
    U += S.nU; V += S.nV;
    S.nU = sdiv2(S.nU + (int)umod31(S.rng.template draw<2 * J>(T.rb, T.rb0), P.m_cnoise) - P.cnoise_k);
    S.nV = sdiv2(S.nV + (int)umod31(S.rng.template draw<2 * J + 1>(T.rb, T.rb0), P.m_cnoise) - P.cnoise_k);
    // chroma phase noise :1748-1762; (double)(int)d == trunc(d) up to the sign of zero, which
    // no later stage can observe
    const RT u = (RT)U, v = (RT)V;
    const RT Ud = rtrunc<RT>((u * C.cosv) - (v * C.sinv));
    const RT Vd = rtrunc<RT>((u * C.sinv) + (v * C.cosv));
    // VHS chroma low-pass :1814-1836 (value for input x1 lands at x2 = x1 - d)
    const int fU = (int)S.vcU.push(Ud, C.a_vc);
    const int fV = (int)S.vcV.push(Vd, C.a_vc);
    // luma at x2: box -> low-pass + emphasis :1793-1812 -> sharpen :1866-1883
    const int lp = pl + T.lc1;
    const int yb = sdiv4s(lp + T.lpB);
    T.lc1 = pl; T.lpB = T.lpA; T.lpA = lp;
    RT m2;
    RT s = S.vl.push((RT)yb, C.a_vl, m2);
    s += S.vpre.hp(s, m2, C.a_vl) * RT(1.6);
    const RT s0 = rtrunc<RT>(s);
    const RT ts = S.sh.push(s0, C.a_sh);
    const int Y = (int)(s0 + ((s0 - ts) * C.sharp2));
    // vertical chroma blend :1843-1863: (above + cur + 1) >> 1, above = 0 for the field's
    // second row, untouched for its first row / blend off (mask, carry and shift all 0)
    U = ((wave_up(fU) & C.bA) + fU + C.bC) >> C.bC;
    V = ((wave_up(fV) & C.bA) + fV + C.bC) >> C.bC;
    Yv = Y; Uv = U; Vv = V;
    // composite out of the VCR :1885-1888: modulate at x2 (amplitude 50: (v*50)/50 == v)
    constexpr int J2 = (J + CT::SH) & 3;      // x2 = x3 + 7 = J + SH (mod 4)
    if constexpr (CT::anyxi) {
        // U4 / V4 at (xi + x2) & 3: the carrier's U / V role swaps on odd lanes, the sign is per lane
        const int chroma = (J2 & 1) ? (C.odd ? U : V) : (C.odd ? V : U);
        const int mm = C.ms[J2];
        return Y + ((chroma ^ mm) - mm);
    }
    const int chroma = (J2 & 1) ? V : U;
    const int mm = (J2 & 2) ? C.mNL : C.mL;
    return Y + ((chroma ^ mm) - mm);
}

// One steady-state pipeline step at unrolled position J (t = SKT + 4n + J).  Position phases:
//   second demodulator / output   x3 = t - 14 - d = 4n + J + 1     -> odd for even J, negated for J = 2
//   re-modulation                 x2 = x3 + 7     = 4n + J (mod 4) -> U for even J, sign by J & 2
//   first demodulator             x1 = t - 7      = DPH + J (mod 4)
// Non-VHS form: x3 = x1 = t - 7 = 4n + J + 1 (SKT = 8), one demodulator.
template <bool VHS, int DPH, int J, class RT, class CT>
DEV uint32_t step(const DevParams &P, State<VHS, RT> &S, Steady &T, const CT &C, int pc, int pl)
{
    int Y, U, V;
    // the TV half's separator sits at x3 = J + 1 + SH (mod 4): odd positions pick, 3 (mod 4) negates
    constexpr bool pick3 = ((J + 1 + CT::SH) & 1) != 0;
    constexpr bool neg3 = ((J + 1 + CT::SH) & 3) == 3;
    if constexpr (!VHS) {
        T.D1.template push<pick3, neg3, true, CT::back, true, CT::anyxi>(pc, C.hi, C.dm, Y, U, V, C.bmul, C.bshift, C.odd, C.mo);
    } else {
        int Yv, Uv, Vv;
        const int c2 = vcr_step<DPH, J, RT, CT>(P, S, T, C, pc, pl, Yv, Uv, Vv);
        if constexpr (CT::svideo) {
            Y = Yv; U = Uv & C.dm; V = Vv & C.dm;  // -vhs-svideo: the components go on as they are :1885; dropout :1891
        } else {
            // ... and separate again at x3 (dropout :1891-1901 is the and-mask on the picked pair)
            T.D2.template push<pick3, neg3, true, false, true, CT::anyxi>(c2, C.hi, C.dm, Y, U, V, 0, 0, C.odd, C.mo);
        }
    }
    if constexpr (CT::fullout) {
        // composite_lowpass :1429-1458 as the output filter: I lands 2 positions back, Q 4; the pixel of position
        // x3 - 4 takes the luma of four steps ago, the filtered I of two steps ago and this step's filtered Q
        const int fUn = (int)S.oU.push((RT)U, C.a_oi);
        const RT fVd = rtrunc<RT>(S.oV.push((RT)V, C.a_oq));
        const int Yo = T.yd[J];
        T.yd[J] = Y;
        const RT fUd = (RT)T.ud[J & 1];
        T.ud[J & 1] = fUn;
        return yiq_to_bgra<RT>(Yo, fUd, fVd);
    } else {
        // composite_lowpass_tv :1399-1427 (delay 1) and YIQ -> RGB for the previous position
        const RT fUd = rtrunc<RT>(S.oU.push((RT)U, C.a_tv));
        const RT fVd = rtrunc<RT>(S.oV.push((RT)V, C.a_tv));
        const int Yo = S.Yprev;
        S.Yprev = Y;
        return yiq_to_bgra<RT>(Yo, fUd, fVd);
    }
}

// One guarded step at any stream position t (wave-uniform): pipeline fill, row end, filter
// tails, drain.  Same state, same results as `step` where both apply.
// The VCR half of a guarded step: returns the VCR's composite sample at x2 = t - 7 - d (0 outside
// the row).
template <class RT, class CT>
DEV int vcr_edge(const DevParams &P, State<true, RT> &S, const CT &C, uint32_t *ring, int t, int &Yv, int &Uv, int &Vv)
{
    const int W = C.W;
    const int pc = t < W ? cs_load(C, t) : 0;             // t is wave-uniform
    int Y, U, V;
    S.D1.template push_edge<CT::back, CT::anyxi>(pc, t, C.xi, C.hi, W, C.xe, Y, U, V, C.bmul, C.bshift);
    const int x1 = t - 7;
    const bool in1 = x1 >= 0 && x1 < W;
    int fU = 0, fV = 0;
    if (in1) {
        U += S.nU; V += S.nV;
        S.nU = sdiv2(S.nU + (int)umod31(S.rng.next(ring, C.lane), P.m_cnoise) - P.cnoise_k);
        S.nV = sdiv2(S.nV + (int)umod31(S.rng.next(ring, C.lane), P.m_cnoise) - P.cnoise_k);
        const RT u = (RT)U, v = (RT)V;
        U = (int)((u * C.cosv) - (v * C.sinv));
        V = (int)((u * C.sinv) + (v * C.cosv));
        fU = (int)S.vcU.push((RT)U, C.a_vc);
        fV = (int)S.vcV.push((RT)V, C.a_vc);
        if (x1 >= W - C.d) {                  // raw tail of the chroma low-pass :1830
            C.tailU[(size_t)(x1 & 15) * C.rstride] = U;
            C.tailU[(size_t)(16 + (x1 & 15)) * C.rstride] = V;
        }
    }
    const int x2 = x1 - C.d;
    const int xl = t - C.LOFF;
    const int pl = (xl >= 0 && xl < W) ? cs_load(C, xl) : 0;
    const int yb = sdiv4(S.l0 + S.l1 + S.l2 + pl);
    S.lsum = S.l1 + S.l2 + pl;
    S.l0 = S.l1; S.l1 = S.l2; S.l2 = pl;
    const bool in2 = x2 >= 0 && x2 < W;
    if (in2) {
        if (x2 >= W - C.d) {
            fU = C.tailU[(size_t)(x2 & 15) * C.rstride];
            fV = C.tailU[(size_t)(16 + (x2 & 15)) * C.rstride];
        }
        RT m2;
        RT s = S.vl.push((RT)yb, C.a_vl, m2);
        s += S.vpre.hp(s, m2, C.a_vl) * RT(1.6);
        const RT s0 = rtrunc<RT>(s);
        const RT ts = S.sh.push(s0, C.a_sh);
        Y = (int)(s0 + ((s0 - ts) * C.sharp2));
    }
    U = ((wave_up(fU) & C.bA) + fU + C.bC) >> C.bC;
    V = ((wave_up(fV) & C.bA) + fV + C.bC) >> C.bC;
    int c2 = 0;
    Yv = 0; Uv = 0; Vv = 0;
    if (in2) {
        Yv = Y; Uv = U; Vv = V;
        const unsigned s = (C.xi + (unsigned)x2) & 3u;
        int chroma = (s & 1u) ? V : U;
        if (s & 2u) chroma = -chroma;
        c2 = Y + chroma;
    }
    return c2;
}

template <bool VHS, class RT, class CT>
DEV bool edge_step(const DevParams &P, State<VHS, RT> &S, const CT &C, uint32_t *ring, int t,
                   uint32_t &px, int &xo_out)
{
    const int W = C.W;
    int Y, U, V;
    int x3 = t - 7;
    if constexpr (VHS) {
        int Yv, Uv, Vv;
        const int c2 = vcr_edge<RT, CT>(P, S, C, ring, t, Yv, Uv, Vv);
        const int x2 = t - 7 - C.d;
        if constexpr (CT::svideo) {
            Y = Yv; U = Uv; V = Vv;
            x3 = x2;
        } else {
            S.D2.template push_edge<false, CT::anyxi>(c2, x2, C.xi, C.hi, W, C.xe, Y, U, V);
            x3 = x2 - 7;
        }
    } else {
        const int pc = t < W ? cs_load(C, t) : 0;         // t is wave-uniform
        S.D1.template push_edge<CT::back, CT::anyxi>(pc, t, C.xi, C.hi, W, C.xe, Y, U, V, C.bmul, C.bshift);
    }
    if constexpr (CT::fullout) {
        if (x3 < 0 || x3 >= W + 4) return false;
        const bool in3f = x3 < W;
        if (!in3f) { U = 0; V = 0; Y = 0; }
        U &= C.dm; V &= C.dm;
        RT fUn = 0, fVn = 0;
        if (in3f) {
            fUn = rtrunc<RT>(S.oU.push((RT)U, C.a_oi));
            fVn = rtrunc<RT>(S.oV.push((RT)V, C.a_oq));
        }
        const int k3 = x3 & 3;                                         // (wave-uniform)
        const int Yo4 = (int)C.xs[k3 * 64], Ur4 = (int)C.xs[(4 + k3) * 64], Vr4 = (int)C.xs[(8 + k3) * 64];   // position x3 - 4
        C.xs[k3 * 64] = (uint32_t)Y; C.xs[(4 + k3) * 64] = (uint32_t)U; C.xs[(8 + k3) * 64] = (uint32_t)V;
        const RT Uf = S.Uf2[0];                                        // filtered I pushed two steps ago
        S.Uf2[0] = S.Uf2[1]; S.Uf2[1] = fUn;
        const int xof = x3 - 4;
        if (xof < 0) return false;
        // the last `delay` samples of a row keep their input :1448-1453
        const RT Uo = xof < W - 2 ? Uf : (RT)Ur4;
        const RT Vo = xof < W - 4 ? fVn : (RT)Vr4;
        px = yiq_to_bgra<RT>(Yo4, Uo, Vo);
        xo_out = xof;
        return true;
    }
    if (x3 < 0 || x3 > W) return false;
    const bool in3 = x3 < W;
    if (!in3) { U = 0; V = 0; Y = 0; }
    U &= C.dm; V &= C.dm;
    RT fUd = 0, fVd = 0;
    if (in3) {
        fUd = rtrunc<RT>(S.oU.push((RT)U, C.a_tv));
        fVd = rtrunc<RT>(S.oV.push((RT)V, C.a_tv));
    }
    const int xo = x3 - 1;
    const int Yo = S.Yprev;
    const int Ur = S.Uraw, Vr = S.Vraw;
    S.Yprev = Y; S.Uraw = U; S.Vraw = V;
    if (xo < 0) return false;
    if (xo >= W - 1) { fUd = (RT)Ur; fVd = (RT)Vr; }      // last sample keeps its input :1419-1424
    px = yiq_to_bgra<RT>(Yo, fUd, fVd);
    xo_out = xo;
    return true;
}

// Steady-state loop: every stage strictly inside the row.  Starts at t = SKT (mod 4), 4 pixels per
// iteration, the next iteration's composite samples requested before the current ones are used.
template <bool VHS, int DPH, class RT, class CT>
DEV int steady(const DevParams &P, State<VHS, RT> &S, const CT &C, uint32_t *ring, uint32_t *lring,
               uint32_t *ostage, const unsigned long long *orow, uint32_t *drow, bool is_out, int t)
{
    // last steady position: every composite sample inside the row (t < W) and no raw chroma tail
    // needed yet (x1 = t - 7 < W - d)
    const int t_end = C.W - (C.d > 7 ? C.d - 7 : 0);
    const int SKT = C.SKT, LOFF = C.LOFF, lane = C.lane;
    if (t + 4 > t_end) return t;
    // the rand() ring's static offsets need the iteration's first slot on a multiple of 8 (LaneRand32::init arranges
    // that for a row wider than the pipeline is deep; otherwise the guarded steps do the whole row)
    if (VHS && (S.rng.pos & 7)) return t;
    // the separators and the luma box in their steady form: the next position is t; the first separator (x1 = t - 7 =
    // DPH mod 4) picks there iff DPH is odd, the second one / the only one of the non-VHS form (x3 = 4n + 1) always
    Steady T;
    constexpr bool pick3_next = ((1 + CT::SH) & 1) != 0;        // x3 = 1 + SH (mod 4) at J = 0
    T.D1.from(S.D1, VHS ? (DPH & 1) != 0 : pick3_next);
    T.D2.from(S.D2, pick3_next);
    T.lc1 = S.l2; T.lpA = S.l2 + S.l1; T.lpB = S.l1 + S.l0;
    if constexpr (CT::fullout) {
        // (x3 = 4 (mod 4) at J = 0: unrolled position J rewrites the slot of positions = J (mod 4); read before the
        //  loop)
#pragma unroll
        for (int q = 0; q < 4; q++) T.yd[q] = (int)C.xs[q * 64];
        T.ud[0] = (int)S.Uf2[0]; T.ud[1] = (int)S.Uf2[1];
    }
    // (the separator in front of the TV stages carries the dropout mask on everything it has picked: the guarded steps
    //  apply it to their outputs instead, so what they left behind is masked here)
    DemodS &Dout = (VHS && !CT::svideo) ? T.D2 : T.D1;
    if (!(VHS && CT::svideo)) { Dout.ieP &= C.dm; Dout.qeP &= C.dm; Dout.ieN &= C.dm; Dout.qeN &= C.dm; }
    int sbase = VHS ? S.rng.pos : 0;
    // samples in flight per stream: one unrolled iteration
    constexpr int PD = 4;
    int pc[PD], pl[PD];
#pragma unroll
    for (int j = 0; j < PD; j++) { pc[j] = cs_load(C, t + j); pl[j] = VHS ? cs_load<NTSC_COMP_LOAD2_AUX>(C, t + j - LOFF) : 0; }
    // (The VCR's luma path reads every composite sample a second time, LOFF = 5 + d positions behind the chroma path.
    //  Tried in round 4: an LDS ring of 20 + 3 slots that keeps each sample until the luma path wants it -- the second
    //  pass over the plane disappears from the L2's memory side, but the ring takes the workgroup from 14 to 20 KB of
    //  LDS, eight decoder workgroups then fill a CU's 160 KB, and the encoder waves of the neighbouring steps no longer
    //  fit beside them: kernel 0.766 -> 0.776 ms, four steps in flight 769k -> 755k fields/s.  Not shipped.)
    // Where the registers allow it (non-VHS form) the next iteration's samples are requested at the
    // top of the current one: a whole iteration of arithmetic hides the HBM latency.  The one-launch
    // VHS form has no registers to spare and reloads each sample right after its step consumed it
    // (its steps are long enough to cover most of the latency).
    constexpr bool PFTOP = !VHS;
    int pend_x = -1;          // wave-uniform: first pixel of a staged 16-pixel group that has not been stored yet
#ifdef NTSC_FAST_LANE_STORE     /* A/B: every lane stores its own 64 bytes as four 16-byte pieces */
#define NTSC_FAST_FLUSH()                                                                         \
    if (pend_x >= 0) {                                                                            \
        if (is_out) {                                                                             \
            const uint4 *sp = reinterpret_cast<const uint4 *>(&ostage[lane * 20]);                \
            g_v4u_ptr dp = (g_v4u_ptr)(drow + pend_x);                                            \
            const uint4 a = sp[0], b = sp[1], c4 = sp[2], d4 = sp[3];                             \
            dp[0] = to_v4u(a); dp[1] = to_v4u(b); dp[2] = to_v4u(c4); dp[3] = to_v4u(d4);         \
        }                                                                                         \
        pend_x = -1;                                                                              \
    }
#else
    // cooperative: store k covers rows 16k .. 16k+15, four lanes per row, so that each row's 64 bytes leave as
    // one contiguous request instead of four 16-byte ones from one lane (a quarter of the write requests)
#define NTSC_FAST_FLUSH()                                                                         \
    if (pend_x >= 0) {                                                                            \
        _Pragma("unroll")                                                                         \
        for (int k = 0; k < 4; k++) {                                                             \
            const int r = 16 * k + (lane >> 2);                                                   \
            const unsigned long long rp = orow[r];                                                \
            const uint4 v = *reinterpret_cast<const uint4 *>(&ostage[r * 20 + (lane & 3) * 4]);   \
            if (rp) NTSC_OUT_STORE((g_v4u_ptr)(rp + 4ull * (unsigned)(pend_x + (lane & 3) * 4)), to_v4u(v)); \
        }                                                                                         \
        pend_x = -1;                                                                              \
    }
#endif
    // Where the burst leaves: at the bottom of the iteration that completed it (measured best, 0.79 against
    // 0.82 ms), or -- in the wrap form, which has no registers left for that -- at the top of the next one.
#ifdef NTSC_FAST_STORE_AT_TOP
    constexpr bool FLUSH_TOP = true;
#else
    constexpr bool FLUSH_TOP = CT::wraps;
#endif
#ifndef NTSC_FAST_NO_ENTRY_WAIT
    // enter the loop with nothing in flight: the waits the compiler would otherwise place inside the loop for the
    // samples requested before it count the stores of every later iteration as well (vmcnt is one in-order counter)
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
#endif
    for (; t + 4 <= t_end; t += 4) {
        if constexpr (FLUSH_TOP) { NTSC_FAST_FLUSH() }
        uint32_t o[4];
        int nc[4] = {0, 0, 0, 0}, nl[4] = {0, 0, 0, 0};
        if (PFTOP) {
#pragma unroll
            for (int j = 0; j < 4; j++) { nc[j] = cs_load(C, t + 4 + j); nl[j] = VHS ? cs_load<NTSC_COMP_LOAD2_AUX>(C, t + 4 + j - LOFF) : 0; }
        }
        T.rb = ring + sbase * 64 + lane;
        T.rb0 = sbase == 0;
        sbase = (sbase + 8) & 31;
#define NTSC_FAST_STEP(J)                                                                         \
        o[J] = step<VHS, DPH, J, RT, CT>(P, S, T, C, pc[J % PD], pl[J % PD]);                     \
        if (!PFTOP) {                                                                             \
            pc[J % PD] = cs_load(C, t + PD + J);                                                  \
            if (VHS) pl[J % PD] = cs_load<NTSC_COMP_LOAD2_AUX>(C, t + PD + J - LOFF);                                  \
        }                                                                                         \
        NTSC_STEP_SCHED_BARRIER();
        NTSC_FAST_STEP(0)
        NTSC_FAST_STEP(1)
        NTSC_FAST_STEP(2)
        NTSC_FAST_STEP(3)
#undef NTSC_FAST_STEP
        if constexpr (PFTOP) {
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j] = nc[j]; pl[j] = nl[j]; }
        }
        // stage 4 pixels; every 4th iteration the wave's 64 x 16 pixels are complete and leave as 64-byte bursts
        const int xo0 = t - SKT;                   // multiple of 4
        const int sub = (xo0 >> 2) & 3;
        *reinterpret_cast<uint4 *>(&ostage[lane * 20 + sub * 4]) = make_uint4(o[0], o[1], o[2], o[3]);
        if (sub == 3) pend_x = xo0 - 12;
        if constexpr (!FLUSH_TOP) { NTSC_FAST_FLUSH() }
    }
    NTSC_FAST_FLUSH()
#undef NTSC_FAST_FLUSH
    // back to the guarded steps' layout (t has advanced by a multiple of 4: the same position phases as at the entry)
    T.D1.to(S.D1, VHS ? (DPH & 1) != 0 : pick3_next);
    T.D2.to(S.D2, pick3_next);
    if constexpr (CT::fullout) {
#pragma unroll
        for (int q = 0; q < 4; q++) C.xs[q * 64] = (uint32_t)T.yd[q];
        S.Uf2[0] = (RT)T.ud[0]; S.Uf2[1] = (RT)T.ud[1];
        // (raw chroma of the last four positions, slots 4..11: stale by now -- the >= 16 guarded steps that follow
        //  rewrite them long before the row's tail reads any)
    }
    S.Uraw = 0; S.Vraw = 0;        // (likewise: only the row's last sample reads them, every guarded step rewrites them)
    S.l2 = T.lc1; S.l1 = T.lpA - T.lc1; S.l0 = T.lpB - S.l1; S.lsum = S.l0 + S.l1 + S.l2;
    if (VHS) S.rng.pos = sbase;
    return t;
}

} // namespace fastdec

// =============================================================================== k_decode_fast
// Occupancy: the VHS form keeps 19 fp64 filter states, two demodulators and a dozen per-lane
// constants alive; squeezed into the 168 registers of 3 waves per SIMD it spills inside the loop
// (every spill reload is an s_waitcnt vmcnt(0) that also waits for the prefetched samples:
// 1.31 ms per 600 fields), with 2 waves per SIMD it needs no scratch at all (0.94 ms) -- and two
// waves already saturate a SIMD's fp64 pipe (tools/chain_probe.hip).
#ifndef NTSC_FAST_WAVES
#define NTSC_FAST_WAVES 2
#endif
// FAST32 (RT = float): its arithmetic is full-rate opcodes (2 cycles per wave64 instruction), which two waves per SIMD do not
// keep busy the way they do the 4-cycle fp64 pipe (a lone wave gets an instruction through per ~5 cycles), so a third wave
// would pay -- but at the 168 registers of three waves every VHS form except the S-Video one spills (58-147 registers,
// scratch in the kernel: -DNTSC_FAST_WAVES_F32=3, round 4), so the float forms stay at two as well.
#ifndef NTSC_FAST_WAVES_F32
#define NTSC_FAST_WAVES_F32 2
#endif
template <class RT> struct FastWaves { static constexpr int n = NTSC_FAST_WAVES; };
template <> struct FastWaves<float> { static constexpr int n = NTSC_FAST_WAVES_F32; };
// WR (VHS form only): head-switch displacements beyond W/10 samples, e.g. PAL's 312.5-line field with
// the default switching point (see cs_load).  BK: subcarrier_amplitude_back other than 50 (the pre-emphasis
// presets -comp-catv* raise it), see scale_back50.
template <bool VHS, class RT, bool WR, bool BK, bool SV = false, bool XA = false, bool FO = false>
DEV void decode_fast_body(const DevParams &P, const GeomDev &G, const FieldDev *__restrict__ fields,
                          const int *__restrict__ comp, const uint32_t *__restrict__ rs_chroma,
                          const int *__restrict__ n0_u, const int *__restrict__ n0_v,
                          const int *__restrict__ hs_shift, const int *__restrict__ pn_noise,
                          const int *__restrict__ dropout, int *__restrict__ tails)
{
    using namespace fastdec;
    __shared__ uint32_t ring[33 * 64];            // LaneRand32: 32 slots + the copy of slot 0
    __shared__ uint32_t lring[FO ? 12 * 64 : 64]; // FO: luma and raw chroma of the last four positions (Const::xs)
    __shared__ __attribute__((aligned(16))) uint32_t ostage[64 * 20];
    __shared__ unsigned long long orow[64];       // every lane's output row (0 = none), for the cooperative stores

    const int lane = threadIdx.x;
    const int gidx = blockIdx.x * 63 + lane - 1;          // lane 0 = halo (row above)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
#ifdef NTSC_AB_NOSTORE      // timing-only A/B build (WRONG pixels): no pixel stores
    const bool is_out = false;
#else
    const bool is_out = lane >= 1 && gidx < P.R && rowok;
#endif
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const int W = P.W;
    uint32_t *drow = reinterpret_cast<uint32_t *>(fd.dst + (size_t)fd.dst_ls * y);
    orow[lane] = is_out ? (unsigned long long)drow : 0ull;
    const size_t tcol = (size_t)blockIdx.x * 64 + lane;
    const size_t tstride = (size_t)gridDim.x * 64;

    typedef Const<RT, WR, BK, SV, XA, FO> CT;
    CT C;
    C.bmul = P.m_amp_back.mul; C.bshift = P.m_amp_back.shift;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.hi = (C.xi & 2u) != 0;
    C.odd = (C.xi & 1u) != 0;
    if constexpr (XA) {
        C.mo = opaque_v(C.odd ? -1 : 0);
#pragma unroll
        for (int j = 0; j < 4; j++) C.ms[j] = opaque_v(((C.xi + (unsigned)j) & 2u) ? -1 : 0);
    }
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = lane;
    C.d = VHS ? P.cdelay : 0;
    C.SKT = (VHS ? (CT::svideo ? 8 : 15) + C.d : 8) + CT::SH;
    C.LOFF = 5 + C.d;
    C.mL = opaque_v(C.hi ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    const bool vb = VHS && P.vblend && P.ntsc;
    C.bA = opaque_v((vb && k >= 2) ? -1 : 0);
    C.bC = opaque_v((vb && k >= 1) ? 1 : 0);
    C.dm = opaque_v((P.loss && dropout[rc] != 0) ? 0 : -1);
    C.cosv = 1; C.sinv = 0;
    if (VHS) {
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (RT)G.ptab[2 * n]; C.sinv = (RT)G.ptab[2 * n + 1];
    }
    C.a_vc = (RT)P.a_vc; C.a_vl = (RT)P.a_vl; C.a_sh = (RT)P.a_sh; C.a_tv = (RT)P.a_tv;
    C.a_oi = (RT)P.a_in_i; C.a_oq = (RT)P.a_in_q;     // (composite_lowpass :1429: the input filter's cutoffs)
    C.sharp2 = (RT)(P.sharpen * 2);            // (x * s) * 2 == x * (s * 2): scaling by 2 is exact
    C.tailU = tails + tcol;
    C.rstride = tstride;
    C.rowbytes = P.Rpad * 4;
    // head switching :1687-1697: a per-lane byte offset plus the buffer bounds check (cs_load)
    const int hs = P.hs ? hs_shift[rc] : 0;
    C.vbase = (int)((unsigned)rc * 4u + (unsigned)hs * (unsigned)C.rowbytes);
    {
        // shift > 0: wrap iff x >= tw - shift, i.e. (tw - shift - 1 - x) < 0; shift < 0: wrap iff x < -shift,
        // i.e. ~(-shift - 1 - x) < 0; shift == 0: never (A = INT_MAX / 2, no sign flip)
        const int tw = W + W / 10;
        C.wrapoff = (int)((unsigned)(hs > 0 ? -tw : tw) * (unsigned)C.rowbytes);
        C.wrapA = hs > 0 ? tw - hs - 1 : (hs < 0 ? -hs - 1 : 0x3FFFFFFF);
        C.wrapS = opaque_v(hs < 0 ? -1 : 0);
    }
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp), 0,
                                               (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);

    State<VHS, RT> S;
    S.D1.init(); S.D2.init();
    S.l0 = S.l1 = S.l2 = S.lsum = 0;
    S.vl.reset(16, C.a_vl); S.vpre.reset(16, C.a_vl); S.vcU.reset(0, C.a_vc); S.vcV.reset(0, C.a_vc);
    S.sh.reset(0, C.a_sh);
    if (CT::fullout) { S.oU.reset(0, C.a_oi); S.oV.reset(0, C.a_oq); }
    else { S.oU.reset(0, C.a_tv); S.oV.reset(0, C.a_tv); }
    S.Yprev = S.Uraw = S.Vraw = 0;
    C.xs = lring + lane;
    if (CT::fullout)
        for (int q = 0; q < 12; q++) C.xs[q * 64] = 0;
    S.Uf2[0] = S.Uf2[1] = 0;
    S.nU = S.nV = 0;
    if (VHS) {
        // the pipeline fill draws twice at every step from x1 = 0 on (t = 7 .. SKT - 1): place the window so that the
        // steady loop's first draw lands on a slot that is a multiple of 8
        const int fill_draws = 2 * (C.SKT - 7);
        S.rng.init(ring, rs_chroma + rc, P.Rpad, lane, (-(31 + fill_draws)) & 7);
        S.nU = n0_u[rc]; S.nV = n0_v[rc];
    }

    const int SKT = C.SKT;
    const int total = W + SKT;
    int t = 0;
    // ---------------- pipeline fill
#ifdef NTSC_EDGE_UNROLL
#pragma unroll NTSC_EDGE_UNROLL
#endif
    for (; t < SKT && t < total; t++) {
        uint32_t px; int xo;
        (void)edge_step<VHS, RT, CT>(P, S, C, ring, t, px, xo);
    }
    // ---------------- steady state: 4 pixels per iteration, ends 16 samples before the row end
    // (one loop per position phase of the first separator, x1 = t - 7 = SKT - 7 (mod 4) at the loop's first position;
    //  the non-VHS form has one separator at a fixed phase)
    if constexpr (!VHS) t = steady<VHS, 0, RT, CT>(P, S, C, ring, lring, ostage, orow, drow, is_out, t);
    else switch ((C.SKT - 7) & 3) {
        case 0: t = steady<VHS, 0, RT, CT>(P, S, C, ring, lring, ostage, orow, drow, is_out, t); break;
        case 1: t = steady<VHS, 1, RT, CT>(P, S, C, ring, lring, ostage, orow, drow, is_out, t); break;
        case 2: t = steady<VHS, 2, RT, CT>(P, S, C, ring, lring, ostage, orow, drow, is_out, t); break;
        default: t = steady<VHS, 3, RT, CT>(P, S, C, ring, lring, ostage, orow, drow, is_out, t); break;
    }
    // ---------------- row end, filter tails, pipeline drain
#ifdef NTSC_EDGE_UNROLL
#pragma unroll NTSC_EDGE_UNROLL
#endif
    for (; t < total; t++) {
        uint32_t px; int xo;
        if (!edge_step<VHS, RT, CT>(P, S, C, ring, t, px, xo)) continue;
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

template <bool VHS, class RT, bool WR = false>
__global__ __launch_bounds__(64, VHS ? FastWaves<RT>::n : 4) void k_decode_fast(DevParams P, GeomDev G,
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
    decode_fast_body<VHS, RT, WR, false>(P, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift, pn_noise, dropout, tails);
}

// the same for a subcarrier_amplitude_back other than 50 (one form per preset: the VHS one takes the wrap-around loads)
template <bool VHS, class RT>
__global__ __launch_bounds__(64, VHS ? FastWaves<RT>::n : 4) void k_decode_fast_bk(DevParams P, GeomDev G,
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
    decode_fast_body<VHS, RT, VHS, true>(P, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift, pn_noise, dropout, tails);
}

// the -vhs family with scanline phases of either parity (-comp-phase 90 / 270, or an odd -comp-phase-offset): the picks
// and the re-modulation take their per-lane forms (DemodS::push ANY, vcr_step); wrap-around loads (any displacement)
template <class RT>
__global__ __launch_bounds__(64, FastWaves<RT>::n) void k_decode_fast_xi(DevParams P, GeomDev G,
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
    decode_fast_body<true, RT, true, false, false, true>(P, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift, pn_noise, dropout, tails);
}

// the -vhs family with the FULL output chroma low-pass (-out-composite-lowpass-lite 0; wrap-around loads)
template <class RT>
__global__ __launch_bounds__(64, FastWaves<RT>::n) void k_decode_fast_fo(DevParams P, GeomDev G,
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
    decode_fast_body<true, RT, true, false, false, false, true>(P, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift, pn_noise, dropout, tails);
}

// the -vhs preset with S-Video out (-vhs-svideo 1): the VCR's components go to the TV stages directly, no
// re-modulation and no second separation; 7 pipeline stages fewer (wrap-around loads: any displacement)
template <class RT>
__global__ __launch_bounds__(64, FastWaves<RT>::n) void k_decode_fast_sv(DevParams P, GeomDev G,
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
    decode_fast_body<true, RT, true, false, true>(P, G, fields, comp, rs_chroma, n0_u, n0_v, hs_shift, pn_noise, dropout, tails);
}

// =============================================================================== k_vcr_front
// The VHS form as two launches (A/B alternative, ntscsim_debug_no_fast_decode bit 1): this kernel
// runs the VCR half of every row (vcr_step / vcr_edge) and writes the VCR's composite output to a
// second transposed plane; the TV half is then exactly the non-VHS decoder (k_decode_fast<false>)
// reading that plane.  Each half keeps about half of the filter states and needs no scratch at 3
// (VCR) / 6 (TV) waves per SIMD, but the composite signal makes one more pass through HBM and a
// 600-field launch cannot fill the extra wave slots: measured 0.86 ms for the pair against 0.80 ms
// for the one-launch form, which therefore stays the default.
#ifndef NTSC_FRONT_WAVES
#define NTSC_FRONT_WAVES 3
#endif
template <class RT>
__global__ __launch_bounds__(64, NTSC_FRONT_WAVES) void k_vcr_front(DevParams P, GeomDev G,
                                                      const FieldDev *__restrict__ fields,
                                                      const int *__restrict__ comp,
                                                      int *__restrict__ comp_out,
                                                      const uint32_t *__restrict__ rs_chroma,
                                                      const int *__restrict__ n0_u,
                                                      const int *__restrict__ n0_v,
                                                      const int *__restrict__ hs_shift,
                                                      const int *__restrict__ pn_noise,
                                                      int *__restrict__ tails)
{
    using namespace fastdec;
    __shared__ uint32_t ring[33 * 64];            // LaneRand32
    const int lane = threadIdx.x;
    const int gidx = blockIdx.x * 63 + lane - 1;          // lane 0 = halo (row above)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const FieldDev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R;
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const int W = P.W;
    const size_t tcol = (size_t)blockIdx.x * 64 + lane;
    const size_t tstride = (size_t)gridDim.x * 64;

    typedef Const<RT, false> CT;
    CT C;
    C.wrapoff = 0; C.wrapA = 0x3FFFFFFF; C.wrapS = 0;
    C.xi = scan_phase(P, y, fd.fieldno);
    C.hi = (C.xi & 2u) != 0;
    C.W = W;
    C.xe = (W & 1) ? W - 1 : W - 2;
    C.lane = lane;
    C.d = P.cdelay;
    C.SKT = 7 + C.d;                   // depth of this half: x2 = t - 7 - d
    C.LOFF = 5 + C.d;
    C.mL = opaque_v(C.hi ? -1 : 0);
    C.mNL = opaque_v(~C.mL);
    const bool vb = P.vblend && P.ntsc;
    C.bA = opaque_v((vb && k >= 2) ? -1 : 0);
    C.bC = opaque_v((vb && k >= 1) ? 1 : 0);
    C.dm = -1;
    {
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        C.cosv = (RT)G.ptab[2 * n]; C.sinv = (RT)G.ptab[2 * n + 1];
    }
    C.a_vc = (RT)P.a_vc; C.a_vl = (RT)P.a_vl; C.a_sh = (RT)P.a_sh; C.a_tv = (RT)P.a_tv;
    C.sharp2 = (RT)(P.sharpen * 2);
    C.tailU = tails + tcol;
    C.rstride = tstride;
    C.rowbytes = P.Rpad * 4;
    const int hs = P.hs ? hs_shift[rc] : 0;
    C.vbase = (int)((unsigned)rc * 4u + (unsigned)hs * (unsigned)C.rowbytes);
    C.comp = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(comp), 0,
                                               (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(comp_out, 0,
                                               (int)((unsigned)W * (unsigned)C.rowbytes), 0x00020000);
    // halo lanes and the lanes past the last row write to padding columns (R <= column < Rpad)
    const int vout = (is_out ? gidx : P.R + lane) * 4;

    State<true, RT> S;
    S.D1.init(); S.D2.init();
    S.l0 = S.l1 = S.l2 = S.lsum = 0;
    S.vl.reset(16, C.a_vl); S.vpre.reset(16, C.a_vl); S.vcU.reset(0, C.a_vc); S.vcV.reset(0, C.a_vc);
    S.sh.reset(0, C.a_sh); S.oU.reset(0, C.a_tv); S.oV.reset(0, C.a_tv);
    S.Yprev = S.Uraw = S.Vraw = 0;
    S.rng.init(ring, rs_chroma + rc, P.Rpad, lane, (-(31 + 2 * (C.SKT - 7))) & 7);     // (as decode_fast_body)
    S.nU = n0_u[rc]; S.nV = n0_v[rc];

    const int SK1 = C.SKT, LOFF = C.LOFF;
    const int total = W + SK1;
    const unsigned rb = (unsigned)C.rowbytes;
    int t = 0;
    // ---------------- pipeline fill (no output: x2 < 0)
    int yv_, uv_, vv_;                 // (components: only the S-Video form of the one-launch kernel uses them)
    for (; t < SK1 && t < total; t++) (void)vcr_edge<RT, CT>(P, S, C, ring, t, yv_, uv_, vv_);
    // ---------------- steady state
    {
        const int t_end = W - (C.d > 7 ? C.d - 7 : 0);
        if (t + 4 <= t_end && !(S.rng.pos & 7)) {
            Steady T;
            T.D1.from(S.D1, (C.d & 1) != 0);
            T.D2.from(S.D2, true);              // (unused by this half)
            T.lc1 = S.l2; T.lpA = S.l2 + S.l1; T.lpB = S.l1 + S.l0;
            int sbase = S.rng.pos;
            int pc[4], pl[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { pc[j] = cs_load(C, t + j); pl[j] = cs_load<NTSC_COMP_LOAD2_AUX>(C, t + j - LOFF); }
            unsigned soff = (unsigned)(t - SK1) * rb;
#define NTSC_VCR_STEP(DPV, J)                                                                     \
            {                                                                                     \
                const int c2 = vcr_step<DPV, J, RT, CT>(P, S, T, C, pc[J], pl[J], yv_, uv_, vv_); \
                pc[J] = cs_load(C, t + 4 + J);        /* reloaded right after its step consumed it */ \
                pl[J] = cs_load<NTSC_COMP_LOAD2_AUX>(C, t + 4 + J - LOFF);                                             \
                __builtin_amdgcn_raw_buffer_store_b32(c2, out, vout, (int)soff, 0);               \
                soff += rb;                                                                       \
                NTSC_STEP_SCHED_BARRIER();                                                        \
            }
#define NTSC_VCR_ITER(DPV)                                                                        \
            for (; t + 4 <= t_end; t += 4) {                                                      \
                T.rb = ring + sbase * 64 + lane; T.rb0 = sbase == 0; sbase = (sbase + 8) & 31;    \
                NTSC_VCR_STEP(DPV, 0) NTSC_VCR_STEP(DPV, 1) NTSC_VCR_STEP(DPV, 2) NTSC_VCR_STEP(DPV, 3)  \
            }
            switch (C.d & 3) {
                case 0: NTSC_VCR_ITER(0) break;
                case 1: NTSC_VCR_ITER(1) break;
                case 2: NTSC_VCR_ITER(2) break;
                default: NTSC_VCR_ITER(3) break;
            }
#undef NTSC_VCR_ITER
#undef NTSC_VCR_STEP
            T.D1.to(S.D1, (C.d & 1) != 0);
            S.l2 = T.lc1; S.l1 = T.lpA - T.lc1; S.l0 = T.lpB - S.l1; S.lsum = S.l0 + S.l1 + S.l2;
            S.rng.pos = sbase;
        }
    }
    // ---------------- row end, filter tails, drain
    for (; t < total; t++) {
        const int c2 = vcr_edge<RT, CT>(P, S, C, ring, t, yv_, uv_, vv_);
        const int x2 = t - SK1;
        if (x2 >= 0) __builtin_amdgcn_raw_buffer_store_b32(c2, out, vout, (int)((unsigned)x2 * rb), 0);
    }
}

} // namespace ntscsim
