// raw28_decode.hip -- the raw-composite decoder of ffmpeg_raw28ntsc.cpp on gfx950 (SURVEY.md
// section 8(f) row f4; C ABI in include/ntscsim.h: ntscsim_raw28_*).
//
// How the tool's three serial recurrences are run in parallel without changing a bit:
//
//  (1) hsync_dc_proc() :556-594 is one fp64 recurrence over the whole capture (three one-pole
//      low-passes and an envelope follower with a data-dependent coefficient).  It is a contraction:
//      at every sync pulse the follower is pulled to the sync tip with a 95-sample time constant,
//      so two trajectories that start apart halve their distance every scanline or faster and
//      become bit-identical after about a hundred.
//      The capture is cut into chunks; chunk c starts `warm` scanlines early from a guessed state
//      (follower at 255, i.e. above the truth) and records the state it reaches at its first own
//      sample and at its end.  The result is exact iff every chunk started from its predecessor's
//      end state -- which is CHECKED (bitwise) on the device; chunks that did not are recomputed
//      from their predecessor's end state, round after round, until every link holds.  At the
//      fixed point chunk 0 started from the tool's initial state and every later chunk from the
//      true state before it: the whole stream is the serial result.  (Worst case = as many rounds
//      as chunks, i.e. the serial algorithm.)  The low-passes (which forget their state within a
//      thousand samples) and the follower are run as two such sweeps, see "the front end as two
//      sweeps" below.
//  (2) The sync search of composite_layer() :622-693, :789-830 walks runs of "below threshold"
//      samples.  The runs are extracted on the GPU (count | scan | scatter); the walk itself is a
//      few hundred scalar steps per field over that run list and stays on the host, together with
//      the buffer-window arithmetic of :277-332 that bounds it.
//  (3) The comb filter's work arrays are file-scope in the tool (:258-261): the last 16 chroma
//      values of a scanline leak into the next one.  tail(y) = G(samples of y, tail(y-1)) is
//      iterated for all scanlines in parallel from tail = 0 until nothing changes (the map forgets
//      its input after a few lines: values are divided by 8 on the way through), again a checked
//      fixed point.
#include "ntscsim.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#pragma clang fp contract(off)

namespace {

template <class T>
struct Buf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    ~Buf() { if (p) (void)hipFree(p); }
};

// pinned host memory, grow-only (contents are not kept)
template <class T>
struct HostBuf {
    T *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const hipError_t e = hipHostMalloc((void **)&p, n * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = n;
        return e;
    }
    ~HostBuf() { if (p) (void)hipHostFree(p); }
};

struct FrontState { double p0, p1, p2, level; };
typedef double lvpair __attribute__((ext_vector_type(2)));     // lv of two consecutive samples

struct FrontConst {
    double alpha, a_fast, om_fast, a_slow, om_slow;   // a, 1.0 - a of the two follower rates (:563-570)
    double om_slow_sb;                                // om_slow ^ FOLLOW_SB: the cheap part of sweep 2's warm-up
    double slow_margin;                               // more than FOLLOW_SB slow steps can lift the level: 1.1 * 64 * a_slow * 255
    int thr;
};

// The follower's rate switch `if (hsync_dc_level > lv)` :563-570 without a compare / select pair (on
// gfx950 a v_cndmask fed by a VALU compare costs several times an fp64 operation, and this loop is one
// long dependent chain): level > lv  <=>  lv - level < 0, and the sign of a rounded fp64 difference is
// the sign of the exact one (x - x is +0; fp64 subnormals are kept), so the high word of (lv - level),
// shifted down arithmetically, is the all-ones / all-zeros mask of the FAST branch.  The two constants
// are then picked word by word with v_bfi (mask laundered so that it is not folded back into a select).
__device__ __forceinline__ double pick64(int m, double if_set, double if_clear)
{
    const unsigned long long a = (unsigned long long)__double_as_longlong(if_set),
                             b = (unsigned long long)__double_as_longlong(if_clear);
    const unsigned um = (unsigned)m;
    const unsigned lo = ((unsigned)a & um) | ((unsigned)b & ~um);
    const unsigned hi = ((unsigned)(a >> 32) & um) | ((unsigned)(b >> 32) & ~um);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double follow(double level, double lv, const FrontConst &K)
{
    int m = __double2hiint(lv - level) >> 31;
    asm volatile("" : "+v"(m));
    return (level * pick64(m, K.om_fast, K.om_slow)) + (lv * pick64(m, K.a_fast, K.a_slow));
}

// one sample of hsync_dc_proc() :556-594 (the raw delay line is a pure index shift and lives in the
// consumers: raw_delayed[s] = raw[s - D])
__device__ __forceinline__ int front_step(FrontState &s, const FrontConst &K, double lv)
{
    lv = (lv * K.alpha) + (s.p0 - (s.p0 * K.alpha)); s.p0 = lv;        // LowpassFilter::lowpass :92-96
    lv = (lv * K.alpha) + (s.p1 - (s.p1 * K.alpha)); s.p1 = lv;
    lv = (lv * K.alpha) + (s.p2 - (s.p2 * K.alpha)); s.p2 = lv;
    s.level = follow(s.level, lv, K);
    int x = (int)(lv - s.level);
    x = x < 0 ? 0 : (x > 255 ? 255 : x);
    return x;
}

// ---- the front end as two sweeps over the new samples [o0, o1) -------------------------------------------
// Grid: a0 = o0 rounded up to 16; chunk c = samples [a0 + c m, a0 + (c + 1) m), m a multiple of 16 Q; a chunk is
// cut into Q sub-chunks of ms = m / Q samples for sweep 1.  The (at most 15) samples before a0 are walked by one
// thread from the exact state before o0 (k_raw28_head), which leaves the exact state before a0 in st_a0.
//
// Sweep 1 (k_raw28_lp): the three low-passes, one lane per sub-chunk, warm-up of w1 samples from a guess
// (state = the first sample), the link of every sub-chunk to its predecessor checked bitwise and repaired as
// described at the top.  Its output, the fp64 lv of every sample, goes to a plane in HBM that is TRANSPOSED, as
// sample pairs: LV2[t / 2][c] = lv of the samples a0 + c m + t, t + 1.  All lanes of a wavefront (64 consecutive
// chunks, same sub-chunk number) are at the same t at the same time, so every store is one contiguous 1 KiB piece.
// Sweep 2 (k_raw28_follow): the follower alone, one lane per chunk, `warm` samples of warm-up from 255.  Lane
// c walks down its own column after the last rows of the columns before it; the lanes of a wavefront are
// again at the same row at the same time, in consecutive columns: every load is one contiguous 1 KiB piece,
// five blocks of 16 samples in flight per lane.  The follower's chain is 7 instructions per sample against
// the 23 of low-passes + follower, and the low-passes are no longer recomputed during the (46 times longer)
// follower warm-up: that is the whole gain (18.7 ms -> 6.7 ms), the run of one lane being
// a single dependent chain whatever is done.
__device__ __forceinline__ double lp3_step(double &p0, double &p1, double &p2, double alpha, double x)
{
    double lv = (x * alpha) + (p0 - (p0 * alpha)); p0 = lv;            // LowpassFilter::lowpass :92-96
    lv = (lv * alpha) + (p1 - (p1 * alpha)); p1 = lv;
    lv = (lv * alpha) + (p2 - (p2 * alpha)); p2 = lv;
    return lv;
}

__global__ void k_raw28_head(const uint8_t *__restrict__ raw, uint8_t *__restrict__ h, size_t o0, size_t e,
                             FrontConst K, FrontState init, FrontState *__restrict__ st_a0)
{
    if (blockIdx.x || threadIdx.x) return;
    FrontState st = init;
    for (size_t s = o0; s < e; s++) h[s] = (uint8_t)front_step(st, K, (double)raw[s]);
    *st_a0 = st;
}

// samples [s0, s1) through the three low-passes, s0 a multiple of 16 (raw is padded by 64 bytes); OUT: lv of
// the sample pair (s0 + 2u, s0 + 2u + 1) to out[u * stride]
// The fp64 plane between the two sweeps (2.3 GB for 600 fields) is written once and read long after the L2 has forgotten
// it: streaming (nt) stores and loads (round 4: k_raw28_lp 0.85 -> 0.75 ms, the front end 4.68 -> 4.54 ms).
#ifdef RAW28_PLAIN_PLANE            /* A/B: plain stores / loads */
#define RAW28_PLANE_STORE(p, v) (*(p) = (v))
#define RAW28_PLANE_LOAD(p) (*(p))
#else
#define RAW28_PLANE_STORE(p, v) __builtin_nontemporal_store((v), (p))
#define RAW28_PLANE_LOAD(p) __builtin_nontemporal_load(p)
#endif
// With `sum` (OUT only; s0 is then a multiple of FOLLOW_SB = 64 within its column): one record per 64 samples for the
// cheap part of sweep 2's warm-up, sum[u * stride] = (min lv, a_slow * SUM lv_i om_slow^(63 - i)) -- what 64 steps of
// the follower's slow branch add to om_slow^64 * level.  A GUESS feeds on it, nothing exact: fused multiply-add is fine.
template <bool OUT>
__device__ __forceinline__ void lp_span(double &p0, double &p1, double &p2, double alpha, const uint8_t *__restrict__ raw,
                                        size_t s0, size_t s1, lvpair *__restrict__ out, size_t stride,
                                        lvpair *__restrict__ sum = nullptr, double a_slow = 0.0, double om_slow = 0.0)
{
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const size_t n = s1 - s0, nb = n / 16;
    const v4 *rv = (const v4 *)(raw + s0);
    v4 cur = rv[0];
    double acc = 0.0, mn = 1e300;
    for (size_t b = 0; b < nb; b++) {
        const v4 nxt = rv[b + 1];                  // (at most 16 bytes past s1: inside the padding)
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            lvpair pr;
            pr.x = lp3_step(p0, p1, p2, alpha, (double)((w[j >> 2] >> (8 * (j & 3))) & 0xFFu));
            pr.y = lp3_step(p0, p1, p2, alpha, (double)((w[j >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu));
            if (OUT) RAW28_PLANE_STORE(&out[(b * 8 + (size_t)(j >> 1)) * stride], pr);
            if (OUT && sum) {
                acc = __builtin_fma(acc, om_slow, pr.x);
                acc = __builtin_fma(acc, om_slow, pr.y);
                mn = __builtin_fmin(mn, __builtin_fmin(pr.x, pr.y));
            }
        }
        if (OUT && sum && (b & 3) == 3) {
            lvpair rec; rec.x = mn; rec.y = acc * a_slow;
            sum[(b >> 2) * stride] = rec;
            acc = 0.0; mn = 1e300;
        }
        cur = nxt;
    }
    for (size_t t = nb * 16; t < n; t += 2) {      // (the end of the stream: the second half of an odd pair is never read)
        lvpair pr;
        pr.x = lp3_step(p0, p1, p2, alpha, (double)raw[s0 + t]);
        pr.y = t + 1 < n ? lp3_step(p0, p1, p2, alpha, (double)raw[s0 + t + 1]) : 0.0;
        if (OUT) out[(t >> 1) * stride] = pr;
    }
}

// Sub-chunk i = c Q + q (stream order) = samples [a0 + i ms, a0 + (i + 1) ms) cut at o1.  Block b covers
// the 64 chunks (b % nwc) * 64 .. of sub-chunk number q = b / nwc.  Round 0: flags == nullptr, all sub-chunks;
// repair rounds: the flagged ones, from prev_end[i - 1].
__global__ __launch_bounds__(64) void k_raw28_lp(const uint8_t *__restrict__ raw, size_t a0, size_t o1, int m, int ms, int Q,
                                                 int w1, int nchunks, int nsub, double alpha,
                                                 const FrontState *__restrict__ st_a0, lvpair *__restrict__ LV2,
                                                 FrontState *__restrict__ st_begin, FrontState *__restrict__ st_end,
                                                 const FrontState *__restrict__ prev_end, const int *__restrict__ flags,
                                                 lvpair *__restrict__ SUM, double a_slow, double om_slow)
{
    const int nwc = (nchunks + 63) / 64;
    const int q = blockIdx.x / nwc, c = (blockIdx.x - q * nwc) * 64 + threadIdx.x;
    const int i = c * Q + q;
    if (c >= nchunks || i >= nsub) return;
    if (flags && !flags[i]) return;
    const size_t g = a0 + (size_t)i * (size_t)ms;
    const size_t e = g + (size_t)ms < o1 ? g + (size_t)ms : o1;
    double p0, p1, p2;
    if (flags) {
        const FrontState s = prev_end[i - 1]; p0 = s.p0; p1 = s.p1; p2 = s.p2;
    } else if (g - a0 <= (size_t)w1) {             // (i == 0 included) close to a0: walk there from the exact state
        const FrontState s = *st_a0; p0 = s.p0; p1 = s.p1; p2 = s.p2;
        lp_span<false>(p0, p1, p2, alpha, raw, a0, g, nullptr, 0);
    } else {
        const size_t w0 = g - (size_t)w1;          // (w1 and ms are multiples of 16)
        p0 = p1 = p2 = (double)raw[w0];
        lp_span<false>(p0, p1, p2, alpha, raw, w0, g, nullptr, 0);
    }
    st_begin[i] = FrontState{p0, p1, p2, 0.0};
    lp_span<true>(p0, p1, p2, alpha, raw, g, e, LV2 + (size_t)q * (size_t)(ms / 2) * (size_t)nchunks + c, (size_t)nchunks,
                  SUM ? SUM + (size_t)q * (size_t)(ms / 64) * (size_t)nchunks + c : nullptr, a_slow, om_slow);
    st_end[i] = FrontState{p0, p1, p2, 0.0};
}

// hsync_dc_raw :588-593
__device__ __forceinline__ uint32_t dc_byte(double lv, double level)
{
    // clamp((int)(lv - level), 0, 255): the conversion to unsigned saturates at 0 and truncates like (int) above it
    // (both values are bounded by the 8-bit samples), which leaves one full-rate v_min_u32 of the clamp
    uint32_t u;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(u) : "v"(lv - level));
    return u < 255u ? u : 255u;
}

constexpr size_t FRONT_SEG = (size_t)1 << 29;     // samples per front-end segment (4 GiB of plane)
constexpr int LP_WARM = 2048;                      // sweep 1's warm-up: 0.942^t t^2 is below an ulp after ~900 samples (measured, round 4:
                                                   // 1280 / 1024 leave every link intact too and the sweep at 0.77 / 0.82 instead of 0.83 ms -- it is
                                                   // bound by the 2.3 GB of plane it writes, not by the walk; 896 needs a repair round)
#ifndef RAW28_FOLLOW_FORM
#define RAW28_FOLLOW_FORM 0       /* measured on the 600-field capture: form 0 7.1 ms, 1 7.6 ms, 2 7.8 ms (whole front end) */
#endif
constexpr int FOLLOW_BLK = 16;                     // samples per block of loads (8 loads of one sample pair)
constexpr int FOLLOW_CK = 1024;                    // samples between the checkpoints a repair round compares with
constexpr int FOLLOW_NB = 6;                       // blocks per lane in registers: five in flight while one is used
constexpr size_t FOLLOW_PIN_LDS = 80 * 1024;       // with the 16 KiB of `sums` > half of the CU's 160 KiB: one workgroup (= one wavefront) per CU
#ifndef RAW28_SLOW_BLOCKS
#define RAW28_SLOW_BLOCKS 0
#endif
constexpr int FOLLOW_SB = 64;                      // samples per record of the summary plane (the cheap part of the warm-up)
constexpr int FOLLOW_SG = 16;                      // records per batch: fetched one batch ahead, parked in LDS

// 16 follower steps over the samples of one block.  A lone wavefront gets one instruction through per ~5
// cycles whatever its kind, so the step is written for the fewest instructions, exactly the tool's own
// statement :563-570: compare, pick the rate a (two v_cndmask), 1.0 - a, two multiplies, one add.
template <bool OUT>
__device__ __forceinline__ void follow_block(double &level, const lvpair (&src)[FOLLOW_BLK / 2], const FrontConst &K, uint32_t (&o)[4])
{
#pragma unroll
    for (int j = 0; j < FOLLOW_BLK; j++) {
        const double lv = (j & 1) ? src[j >> 1].y : src[j >> 1].x;
#if RAW28_FOLLOW_FORM == 0       /* pick the rate, 7 instructions, 5 dependent */
        const double a = level > lv ? K.a_fast : K.a_slow;
        level = (level * (1.0 - a)) + (lv * a);
#elif RAW28_FOLLOW_FORM == 1     /* pick rate and 1 - rate, 8 instructions, 4 dependent */
        const bool fast = level > lv;
        const double a = fast ? K.a_fast : K.a_slow, om = fast ? K.om_fast : K.om_slow;
        level = (level * om) + (lv * a);
#else                            /* both branches, pick the result: 9 instructions, 3 dependent */
        const bool fast = level > lv;
        const double cf = (level * K.om_fast) + (lv * K.a_fast);
        const double cs = (level * K.om_slow) + (lv * K.a_slow);
        level = fast ? cf : cs;
#endif
        if (OUT) o[j >> 2] |= dc_byte(lv, level) << (8 * (j & 3));
    }
}

// 16 steps for a GUESS (the cheap part of sweep 2's warm-up, where a superblock is not all slow): the same update as
// level + (lv - level) a, the rate picked by the sign of the difference through a mask and one fused multiply-add --
// 5 instructions instead of 8 issue slots, a rounding or two away from the tool's expression per step (which is what
// the closed form of the slow superblocks costs as well).
__device__ __forceinline__ void guess_block(double &level, const lvpair (&src)[FOLLOW_BLK / 2], const FrontConst &K)
{
#pragma unroll
    for (int j = 0; j < FOLLOW_BLK; j++) {
        const double lv = (j & 1) ? src[j >> 1].y : src[j >> 1].x;
        const double dlt = lv - level;
        int m = __double2hiint(dlt) >> 31;         // all ones: level > lv, the fast rate
        asm volatile("" : "+v"(m));
        level = __builtin_fma(dlt, pick64(m, K.a_fast, K.a_slow), level);
    }
}

// 16 steps of the slow branch alone -- level = level (1.0 - a_slow) + lv a_slow, the tool's own expression :568-569 with
// 1.0 - a_slow the same rounded constant -- for a block of which the caller KNOWS that no step takes the fast branch:
// 3 instructions per sample instead of 7.
template <bool OUT>
__device__ __forceinline__ void slow_block(double &level, const lvpair (&src)[FOLLOW_BLK / 2], const FrontConst &K, uint32_t (&o)[4])
{
#pragma unroll
    for (int j = 0; j < FOLLOW_BLK; j++) {
        const double lv = (j & 1) ? src[j >> 1].y : src[j >> 1].x;
        level = (level * K.om_slow) + (lv * K.a_slow);
        if (OUT) o[j >> 2] |= dc_byte(lv, level) << (8 * (j & 3));
    }
}
// That knowledge: sweep 1's record of the 64-sample superblock around the block holds the smallest lv in it; a level
// more than 0.0625 below it stays below every lv of the block (16 slow steps lift it by less than 16 a_slow 255 =
// 0.015), so `hsync_dc_level > lv` :563 is false 16 times.  Taken by a wavefront only if it holds for all its lanes.
__device__ __forceinline__ bool block_is_slow(double level, const lvpair &rec, const FrontConst &K) { return level + K.slow_margin < rec.x; }

// Sweep 2.  One wavefront = 64 consecutive chunks.  With Kw = ceil(warm / m) and r0 = Kw m - warm, lane c
// starts at row r0 of column c - Kw and walks to the end of column c - 1 (its warm-up, `warm` samples), then
// its own column with output.  Columns before 0 do not exist: such a lane waits, with the exact level before
// a0, until its walk reaches column 0.  Repair rounds (flags): the own column only, from prev_end[c - 1].
// The plane holds sample PAIRS: LV2[t / 2][c] = (lv(t), lv(t + 1)), one 16-byte load per lane and pair.
//
// The warm-up only has to produce a GUESS of the level before the chunk (the link check is what makes the result
// exact), so its first `cheap` samples (round 4) are not walked sample by sample: where the level is below every
// lv of a 64-sample superblock by more than the slow branch can lift it in 64 steps (K.slow_margin > 64 a_slow 255), all 64
// steps take the slow branch, and they amount to level = om_slow^64 level + B with the B sweep 1 left in SUM -- one
// fused multiply-add and 16 bytes instead of 448 instructions and 512 bytes.  That holds for ~94 % of the superblocks
// (everything but the sync tips); the others are walked exactly.  The closed form differs from 64 rounded steps by a
// few ulp, so the guess is ~1e-12 off the true level (it would be 1e-14 with exact steps) and the last `warm - cheap`
// samples, walked exactly, have to close that: the median guess is bit-identical after 8 scanlines, one in a thousand
// needs more than 25 (tools/follow_guess_probe.c); with the default 30 a chunk or two of ten thousand are left to a
// repair round (measured, profiles/r04_raw28_sweep.txt: 12 scanlines 1,394 chunks, 16: 347, 20: 86, 24: 28, 30: 1).
// "All lanes" is the catch: a lane inside a field's broad sync pulses is fast for lines on end, lanes that are not a
// whole number of scanlines apart see their sync tips at different times.  Hence the host's choice of the chunk length
// (16 scanlines: the only multiple of 64 samples that is a whole number of 1820-sample lines) and of 16 chunks per
// wavefront, and the way out below when hardly any superblock turns out slow.
__global__ __launch_bounds__(64) void k_raw28_follow(const lvpair *__restrict__ LV2, const lvpair *__restrict__ SUM, size_t a0,
                                                     size_t o1, int m, int warm, int cheap,
                                                     int nchunks, FrontConst K, const FrontState *__restrict__ st_a0,
                                                     uint8_t *__restrict__ h, double *__restrict__ lv_begin,
                                                     double *__restrict__ lv_end, double *__restrict__ ckpt, int ncp,
                                                     const double *__restrict__ prev_end, const int *__restrict__ flags, int lpw)
{
    constexpr int NB = FOLLOW_NB, HB = FOLLOW_BLK / 2;
    // lpw chunks per wavefront: the other lanes mirror the first one (same addresses, no stores)
    const int c = blockIdx.x * lpw + threadIdx.x;
    const bool valid = (int)threadIdx.x < lpw && c < nchunks;
    const int cc = valid ? c : (int)blockIdx.x * lpw;
    const size_t ncols = (size_t)nchunks;
    const size_t rstride = ncols * sizeof(lvpair);     // bytes from one row pair to the next
    const int Kw = (warm + m - 1) / m;
    double level = 255.0;
    if ((size_t)cc * (size_t)m <= (size_t)warm) level = st_a0->level;
    const bool repair = flags != nullptr;
    const bool mine = valid && (!repair || flags[cc]);
    if (repair) {
        if (!__any(mine)) return;
        if (mine) level = prev_end[cc - 1];
    }
    lvpair R[NB][HB];
    lvpair Sr[NB];                                 // the superblock record that goes with each block (SUM != nullptr)
    // (measured: the per-block branch between slow_block and follow_block costs the loads their pipelining -- the
    // compiler waits for ALL of them at every block -- and the sweep 3.5 -> 4.4 ms: off until the loops are split)
    const bool have_sum = RAW28_SLOW_BLOCKS && SUM != nullptr;
    uint32_t o[4] = {0, 0, 0, 0};
    // a block of 8 row pairs: a uniform row pointer plus the lane's column (the loads take their base from
    // scalar registers)
    auto load_block = [&](lvpair (&dst)[HB], int row, int col) {
        const unsigned coff = (unsigned)col * (unsigned)sizeof(lvpair);
        const char *rowp = (const char *)(LV2 + (size_t)(row >> 1) * ncols);
#pragma unroll
        for (int j = 0; j < HB; j++) dst[j] = RAW28_PLANE_LOAD((const lvpair *)(rowp + (size_t)j * rstride + coff));
    };
    // ---- warm-up: columns cc - Kw .. cc - 1, the first one from row r0
    if (!repair) {
        int row = Kw * m - warm, k = -Kw;          // wave-uniform position of the next block to use ...
        if (cheap > 0) {                           // (the host: SUM exists; m, warm, cheap are multiples of 64)
            __shared__ lvpair sums[FOLLOW_SG][64];
            const int lane = (int)threadIdx.x;
            const int nsb = cheap / FOLLOW_SB;
            int frow = row, fk = k;                // where the next record to fetch lies
            lvpair tmp[FOLLOW_SG];
            auto fetch = [&]() {                   // (runs up to a batch past the cheap part: columns clamped, never used)
#pragma unroll
                for (int j = 0; j < FOLLOW_SG; j++) {
                    int col = cc + fk;
                    col = col < 0 ? 0 : (col >= nchunks ? nchunks - 1 : col);
                    tmp[j] = SUM[(size_t)(frow / FOLLOW_SB) * ncols + (size_t)col];
                    frow += FOLLOW_SB;
                    if (frow >= m) { frow -= m; fk++; }
                }
            };
            fetch();
            int walked = 0, done_sb = 0;           // superblocks that had to be walked exactly / superblocks behind us
            for (int sb = 0; sb < nsb; sb += FOLLOW_SG) {
                // lanes that are out of step (a capture whose scanlines are not the nominal length, a forced chunk
                // length): hardly any superblock is slow for all of them at once, and fetching the samples on demand
                // is slower than the pipelined exact walk below, which then takes over
                if (sb >= 16 * FOLLOW_SG && 2 * walked > sb) break;
#pragma unroll
                for (int j = 0; j < FOLLOW_SG; j++) sums[j][lane] = tmp[j];    // (every lane reads back its own records only)
                fetch();                           // the next batch is in flight while this one is walked
                const int nb = nsb - sb < FOLLOW_SG ? nsb - sb : FOLLOW_SG;
                lvpair Snext = sums[0][lane];
                for (int j = 0; j < nb; j++) {
                    const lvpair S = Snext;
                    Snext = sums[j + 1 < FOLLOW_SG ? j + 1 : j][lane];       // (the LDS read is off the level's chain)
                    const bool waiting = cc + k < 0;
                    if (__all(waiting || block_is_slow(level, S, K))) {
                        if (!waiting) level = __builtin_fma(level, K.om_slow_sb, S.y);
                    } else {
                        walked++;
                        const int col = waiting ? 0 : cc + k;
#pragma unroll
                        for (int i = 0; i < FOLLOW_SB / FOLLOW_BLK; i++) load_block(R[i], row + i * FOLLOW_BLK, col);
#pragma unroll
                        for (int i = 0; i < FOLLOW_SB / FOLLOW_BLK; i++)
                            if (!waiting) guess_block(level, R[i], K);
                    }
                    row += FOLLOW_SB;
                    if (row >= m) { row -= m; k++; }
                }
                done_sb += nb;
            }
            cheap = done_sb * FOLLOW_SB;           // (what is left of the warm-up is walked exactly)
        }
        int lrow = row, lk = k;                    // ... and of the next block to request
        const long long nblk = (long long)(warm - cheap) / FOLLOW_BLK;         // (warm and m are multiples of 16)
        auto request = [&](lvpair (&dst)[HB], lvpair &rec) {    // column clamped at 0 for the lanes that are still waiting
            const int col = cc + lk < 0 ? 0 : cc + lk;
            load_block(dst, lrow, col);
            if (have_sum) rec = SUM[(size_t)(lrow / FOLLOW_SB) * ncols + (size_t)col];
            lrow += FOLLOW_BLK;
            if (lrow >= m) { lrow -= m; lk++; }
        };
        auto use = [&](const lvpair (&src)[HB], const lvpair &rec) {
            const bool act = cc + k >= 0;
            if (have_sum && __all(!act || block_is_slow(level, rec, K))) {
                if (act) slow_block<false>(level, src, K, o);
            } else if (act) follow_block<false>(level, src, K, o);
            row += FOLLOW_BLK;
            if (row >= m) { row -= m; k++; }
        };
        // NB - 1 blocks in flight, no condition inside the loop (the compiler's wait counts stay exact)
        long long b = 0;
        if (nblk >= NB - 1) {
#pragma unroll
            for (int i = 0; i < NB - 1; i++) request(R[i], Sr[i]);
            for (; b + 2 * NB - 1 <= nblk; b += NB) {
#pragma unroll
                for (int i = 0; i < NB; i++) { request(R[(i + NB - 1) % NB], Sr[(i + NB - 1) % NB]); use(R[i], Sr[i]); }
            }
#pragma unroll
            for (int i = 0; i < NB - 1; i++) use(R[i], Sr[i]);
            b += NB - 1;
        }
        for (; b < nblk; b++) { request(R[0], Sr[0]); use(R[0], Sr[0]); }
        if (valid) lv_begin[c] = level;
    } else if (mine) {
        lv_begin[c] = level;
    }
    // ---- the own column: rows 0 .. len - 1, 16 bytes of hsync_dc_raw per store
    const size_t g = a0 + (size_t)cc * (size_t)m;
    const int len = !mine ? 0 : (g + (size_t)m <= o1 ? m : (int)(o1 - g));
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const int nbf = len / FOLLOW_BLK, nbm = m / FOLLOW_BLK;    // full blocks of this lane; blocks of a full column (uniform)
    {
        // (rows past `len` of the stream's last column hold nothing: they are loaded, never used)
        auto request = [&](lvpair (&dst)[HB], lvpair &rec, int blk) {
            const int r = (blk < nbm ? blk : nbm - 1) * FOLLOW_BLK;
            load_block(dst, r, cc);
            if (have_sum) rec = SUM[(size_t)(r / FOLLOW_SB) * ncols + (size_t)cc];
        };
        // Every FOLLOW_CK samples the level is left in ckpt[c][.].  A repair round compares instead: once the repaired
        // run meets the earlier one bitwise, everything after it (bytes, end level) is what the earlier run wrote.
        bool done = !mine;
        auto use = [&](const lvpair (&src)[HB], const lvpair &rec, int blk) {
            // (a record exists for whole superblocks only: the last, cut one of the stream has none)
            const bool act = blk < nbf && !done;
            const bool slow = have_sum && __all(!act || ((blk | (FOLLOW_SB / FOLLOW_BLK - 1)) < nbf && block_is_slow(level, rec, K)));
            if (act) {
                o[0] = o[1] = o[2] = o[3] = 0;
                if (slow) slow_block<true>(level, src, K, o);
                else follow_block<true>(level, src, K, o);
                *(v4 *)(h + g + (size_t)blk * FOLLOW_BLK) = v4{o[0], o[1], o[2], o[3]};
                if ((blk + 1) % (FOLLOW_CK / FOLLOW_BLK) == 0) {
                    double *cp = ckpt + (size_t)cc * (size_t)ncp + (size_t)((blk + 1) / (FOLLOW_CK / FOLLOW_BLK) - 1);
                    if (repair && __double_as_longlong(*cp) == __double_as_longlong(level)) done = true;
                    else *cp = level;
                }
            }
        };
#pragma unroll
        for (int i = 0; i < NB - 1; i++) request(R[i], Sr[i], i);
        for (int b = 0; b < nbm; b += NB) {
#pragma unroll
            for (int i = 0; i < NB; i++) { request(R[(i + NB - 1) % NB], Sr[(i + NB - 1) % NB], b + i + NB - 1); use(R[i], Sr[i], b + i); }
            if (repair && __all(done)) return;
        }
        if (done) return;                          // (lanes without a chunk; repair rounds: runs that met the earlier one)
    }
    for (int t = nbf * FOLLOW_BLK; t < len; t++) {
        const lvpair pr = LV2[(size_t)(t >> 1) * ncols + (size_t)cc];
        const double lv = (t & 1) ? pr.y : pr.x;
        level = follow(level, lv, K);
        h[g + t] = (uint8_t)dc_byte(lv, level);
    }
    if (mine) lv_end[c] = level;
}

__global__ void k_raw28_links1(const double *__restrict__ lv_begin, const double *__restrict__ lv_end, int nchunks,
                               int *__restrict__ flags, int *__restrict__ nbad)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const int bad = c > 0 && __double_as_longlong(lv_begin[c]) != __double_as_longlong(lv_end[c - 1]);
    flags[c] = bad;
    if (bad) atomicAdd(nbad, 1);
}

__global__ void k_raw28_links(const FrontState *__restrict__ st_begin, const FrontState *__restrict__ st_end,
                              int nchunks, int *__restrict__ flags, int *__restrict__ nbad)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    int bad = 0;
    if (c > 0) {
        const unsigned long long *a = (const unsigned long long *)&st_begin[c];
        const unsigned long long *b = (const unsigned long long *)&st_end[c - 1];
        bad = (a[0] != b[0]) | (a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3]);
    }
    flags[c] = bad;
    if (bad) atomicAdd(nbad, 1);
}

// ---- runs of h < thr.  A block looks at 4096 samples, 16 per thread as one 128-bit load turned into a
// bit mask; run starts / ends are the 0->1 / 1->0 transitions of that mask.  Pass 1 counts them per
// block (starts in the low word, ends in the high word), a device scan turns the counts into
// offsets, pass 2 writes the positions in stream order.
constexpr int RUN_T = 256, RUN_PER = 16, RUN_BLOCK = RUN_T * RUN_PER;
__device__ __forceinline__ void run_masks(const uint8_t *__restrict__ h, size_t N, int thr, size_t s0,
                                          uint32_t &starts, uint32_t &ends)
{
    starts = ends = 0;
    if (s0 >= N) return;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4 v = *(const v4 *)(h + s0);            // (h is allocated 16 bytes past N)
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t byte = (v[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        m |= (uint32_t)((int)byte < thr) << k;
    }
    const uint32_t nvalid = N - s0 >= 16 ? 16u : (uint32_t)(N - s0);
    const uint32_t valid = nvalid >= 16 ? 0xFFFFu : ((1u << nvalid) - 1u);
    m &= valid;
    const uint32_t prev = (s0 > 0 && h[s0 - 1] < thr) ? 1u : 0u;
    const uint32_t sh = ((m << 1) | prev);
    starts = m & ~sh & valid;
    ends = ~m & sh & valid;
    // a run still open at the end of the capture ends at N: bit `nvalid` (position N)
    if (s0 + nvalid == N && ((m >> (nvalid - 1)) & 1u)) ends |= 1u << nvalid;
}
// Prefix sums of the run counts (two 32-bit counters packed in one u64: starts low, ends high -- neither can carry into
// the other: at most one start and one end per sample, N < 2^32).  Own code, no library: an inclusive scan across the 64
// lanes of a wave by six shuffle-and-add steps, the waves of a block chained through LDS.
__device__ __forceinline__ unsigned long long wave_scan_incl_u64(unsigned long long v)
{
    const int lane = (int)(threadIdx.x & 63);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned lo = (unsigned)__shfl_up((int)(unsigned)(v & 0xFFFFFFFFull), d), hi = (unsigned)__shfl_up((int)(unsigned)(v >> 32), d);
        if (lane >= d) v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
// exclusive prefix of `mine` over the T threads of the block (T a multiple of 64, at most 1024); *total = the block's sum
template <int T>
__device__ __forceinline__ unsigned long long block_scan_excl_u64(unsigned long long mine, unsigned long long *wsum /* [T / 64] LDS */,
                                                                  unsigned long long *total)
{
    const int w = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const unsigned long long incl = wave_scan_incl_u64(mine);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < T / 64; k++) { const unsigned long long s = wsum[k]; if (k < w) base += s; tot += s; }
    __syncthreads();                 // (wsum may be reused by the caller's next round)
    if (total) *total = tot;
    return base + incl - mine;
}

__global__ __launch_bounds__(RUN_T) void k_raw28_run_count(const uint8_t *__restrict__ h, size_t N, int thr,
                                                           unsigned long long *__restrict__ cnt)
{
    __shared__ unsigned long long wsum[RUN_T / 64];
    uint32_t st, en;
    run_masks(h, N, thr, ((size_t)blockIdx.x * RUN_T + threadIdx.x) * RUN_PER, st, en);
    const unsigned long long mine = (unsigned long long)__popc(st) | ((unsigned long long)__popc(en) << 32);
    unsigned long long tot;
    (void)block_scan_excl_u64<RUN_T>(mine, wsum, &tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

// exclusive scan of n counters in place of a library's device-wide scan: n is the number of 4,096-sample blocks of the
// buffer (~70,000 for a ten-second capture), so ONE workgroup does it -- every thread sums its own contiguous slice,
// the slice sums are scanned across the block, every thread writes its slice's prefixes
constexpr int SCAN_T = 1024;
__global__ __launch_bounds__(SCAN_T) void k_raw28_scan_counts(const unsigned long long *__restrict__ in,
                                                              unsigned long long *__restrict__ out, size_t n)
{
    __shared__ unsigned long long wsum[SCAN_T / 64];
    const size_t per = (n + SCAN_T - 1) / SCAN_T;
    const size_t a = (size_t)threadIdx.x * per, b = a + per < n ? a + per : n;
    unsigned long long sum = 0;
    for (size_t i = a; i < b; i++) sum += in[i];
    unsigned long long run = block_scan_excl_u64<SCAN_T>(sum, wsum, nullptr);
    for (size_t i = a; i < b; i++) { const unsigned long long v = in[i]; out[i] = run; run += v; }
}
__global__ __launch_bounds__(RUN_T) void k_raw28_run_scatter(const uint8_t *__restrict__ h, size_t N, int thr,
                                                             const unsigned long long *__restrict__ off,
                                                             uint32_t *__restrict__ rstart, uint32_t *__restrict__ rend)
{
    __shared__ unsigned long long wsum[RUN_T / 64];
    const size_t s0 = ((size_t)blockIdx.x * RUN_T + threadIdx.x) * RUN_PER;
    uint32_t st, en;
    run_masks(h, N, thr, s0, st, en);
    const unsigned long long mine = (unsigned long long)__popc(st) | ((unsigned long long)__popc(en) << 32);
    unsigned long long before = block_scan_excl_u64<RUN_T>(mine, wsum, nullptr);
    before += off[blockIdx.x];
    uint32_t os = (uint32_t)(before & 0xFFFFFFFFull), oe = (uint32_t)(before >> 32);
    while (st) { const int k = __ffs(st) - 1; st &= st - 1; rstart[os++] = (uint32_t)(s0 + k); }
    while (en) { const int k = __ffs(en) - 1; en &= en - 1; rend[oe++] = (uint32_t)(s0 + k); }
}

// ---- calibration sums of the equalising pulses :661-676: one wave per range [si, i)
struct CalRange { uint32_t si, i, pulse, _pad; };   // samples [si, i) of the stream, summed into pulse `pulse`
struct CalSums { int mina, mind, maxa, maxd; };
// (base = stream position of sample 0 of the buffer: 0 for a whole capture; a stream keeps at least D
// samples in front of every position it can still ask for)
__device__ __forceinline__ int raw_delayed(const uint8_t *raw, const uint8_t *h, size_t s, int D, int thr, int mark,
                                           unsigned long long base)
{
    if (mark && h[s] < thr) return 255;            // :590-591
    return base + s >= (unsigned long long)D ? raw[s - D] : 0;   // :572-581 (the delay line starts zero-filled)
}
__global__ __launch_bounds__(64) void k_raw28_cal(const uint8_t *__restrict__ raw, const uint8_t *__restrict__ h,
                                                  const CalRange *__restrict__ rg, CalSums *__restrict__ out,
                                                  int D, int thr, int mark, unsigned long long base)
{
    const CalRange r = rg[blockIdx.x];
    int mina = 0, mind = 0, maxa = 0, maxd = 0;
    for (size_t s = (size_t)r.si + threadIdx.x; s < (size_t)r.i; s += 64) {
        const int v = raw_delayed(raw, h, s, D, thr, mark, base);
        if (h[s] >= thr) { maxa += v; maxd++; } else { mina += v; mind++; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        mina += __shfl_down(mina, o); mind += __shfl_down(mind, o);
        maxa += __shfl_down(maxa, o); maxd += __shfl_down(maxd, o);
    }
    if (threadIdx.x == 0) out[blockIdx.x] = CalSums{mina, mind, maxa, maxd};
}

// ---- per scanline: equalisation :701-716, comb split :719-755, rendering :757-775
struct LineRec {
    uint32_t pos;          // first sample of the scanline (absolute)
    int32_t field, row;    // destination frame and row
    double blank, white;   // levels in force while the field is rendered
};
struct RenderConst {
    int len, width, D, thr;
    int mark, no_equ, no_wequ, no_sc, show_sc;
    unsigned long long base;       // stream position of sample 0 of the buffer
};
// black / white level equalisation of one sample value v = 0 .. 255 (:708-710)
__device__ __forceinline__ int equalise_value(int v, const LineRec &L, const RenderConst &R)
{
    if (!R.no_equ) {
        v = (int)((double)v - L.blank);                                 // :708
        if (!R.no_wequ) v = (int)((double)(v * 255) / (L.white - L.blank));   // :709
        v = (int)(int16_t)v;                                            // :710 (int16 member)
    }
    return v;
}
__device__ __forceinline__ int equalised(const uint8_t *raw, const uint8_t *h, size_t N, const LineRec &L,
                                         const RenderConst &R, int x)
{
    const size_t s = (size_t)L.pos + (size_t)x;
    const int v = s < N ? raw_delayed(raw, h, s, R.D, R.thr, R.mark, R.base) : 0;     // int16 luma = raw :702
    return equalise_value(v, L, R);
}

// tail(y) from the last 32 positions of the scanline and tail(y-1): the 16 values the tool leaves in
// int_chroma[len .. len+15] (:744-745 applied to int_chroma[len-16 .. len-1])
// tail(y) = G(samples of scanline y, tail(y - 1)), :719-745 restricted to the 16 values that survive the scanline
__device__ __forceinline__ void raw28_tail_of(const uint8_t *__restrict__ raw, const uint8_t *__restrict__ h, size_t N,
                                              const LineRec &L, const RenderConst &R, const int (&prev)[16], int (&out)[16])
{
    const int len = R.len;
    int S[20], C[32];                              // S: positions len-16 .. len+3; C: len-16 .. len+15
#pragma unroll
    for (int k = 0; k < 20; k++) S[k] = equalised(raw, h, N, L, R, len - 16 + k);
#pragma unroll
    for (int k = 0; k < 16; k++) C[k] = S[k] - (S[k] + S[k + 4] + 1) / 2;     // :731-734
#pragma unroll
    for (int k = 0; k < 16; k++) C[16 + k] = prev[k];
#pragma unroll
    for (int k = 0; k < 16; k++) C[k] = C[k] + C[k + 8] - C[k + 4] - C[k + 12];   // :736-737 (ascending, in place)
#pragma unroll
    for (int it = 0; it < 4; it++)
#pragma unroll
        for (int k = 0; k < 16; k++) C[k] -= (C[k] + C[k + 4]) / 2;                // :739-742
#pragma unroll
    for (int k = 0; k < 16; k++) out[k] = C[k] / 4;                                // :744-745
}

// One round of the fixed-point iteration: tout(y) = G(y, tin(y - 1)) for every scanline at once; counts the
// scanlines whose value differs from the previous round's.  (Row -1 of the arrays: the tail carried in, 0 at a
// stream's start.)
__global__ void k_raw28_tails(const uint8_t *__restrict__ raw, const uint8_t *__restrict__ h, size_t N,
                              const LineRec *__restrict__ lines, int nlines, RenderConst R,
                              const int *__restrict__ tin, int *__restrict__ tout, int *__restrict__ nchanged)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= nlines) return;
    int prev[16], v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) prev[k] = tin[((ptrdiff_t)y - 1) * 16 + k];
    raw28_tail_of(raw, h, N, lines[y], R, prev, v);
    int ch = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        ch |= (tin[(size_t)y * 16 + k] != v[k]);   // against this scanline's value of the previous round
        tout[(size_t)y * 16 + k] = v[k];
    }
    if (ch) atomicAdd(nchanged, 1);
}

// A first guess that is almost always the fixed point already: every thread walks TAIL_WU scanlines of warm-up
// from a zero tail (the map divides what it inherits by 8 on the way through, so a tail forgets its
// predecessors within a few scanlines; a walk that reaches row -1 starts from the carried tail, exactly) and
// then its own TAIL_B scanlines serially.  2.5 rounds' worth of work in one launch instead of ~16 rounds; the
// rounds of k_raw28_tails that follow confirm it (a round that changes nothing) or finish the job.
// (Round 4: 24 + 16 scanlines per thread became 16 + 4 -- the launch lasts as long as ONE thread's serial walk whatever
// the number of scanlines, and it now runs once per group of fields behind the sync walk.  Measured against the
// capture's noise, tools/raw28_noise_probe.py: 8 scanlines of warm-up settle a clean capture and no longer one of noise
// level 6, 12 do up to 6, 16 at every level tried (24); where the guess does not settle the rounds over all scanlines and
// a second rendering cost 0.8-0.95 ms.)
#ifndef RAW28_TAIL_B
#define RAW28_TAIL_B 4
#endif
#ifndef RAW28_TAIL_WU
#define RAW28_TAIL_WU 16
#endif
constexpr int TAIL_B = RAW28_TAIL_B, TAIL_WU = RAW28_TAIL_WU;
__global__ void k_raw28_tails_scan(const uint8_t *__restrict__ raw, const uint8_t *__restrict__ h, size_t N,
                                   const LineRec *__restrict__ lines, int nlines, RenderConst R,
                                   const int *__restrict__ carried, int *__restrict__ ta, int *__restrict__ tb)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = j * TAIL_B;
    if (y0 >= nlines) return;
    const int ys = y0 - TAIL_WU;
    int T[16];
#pragma unroll
    for (int k = 0; k < 16; k++) T[k] = ys <= 0 ? carried[k] : 0;
    const int ye = y0 + TAIL_B < nlines ? y0 + TAIL_B : nlines;
    for (int y = ys < 0 ? 0 : ys; y < ye; y++) {
        int v[16];
        raw28_tail_of(raw, h, N, lines[y], R, T, v);
#pragma unroll
        for (int k = 0; k < 16; k++) T[k] = v[k];
        if (y >= y0) {
#pragma unroll
            for (int k = 0; k < 16; k++) { ta[(size_t)y * 16 + k] = v[k]; tb[(size_t)y * 16 + k] = v[k]; }
        }
    }
}

#ifdef RAW28_PLAIN_FRAME_STORES    /* A/B: the round-3 stores */
#define RAW28_FRAME_STORE(p, v) (*(p) = (v))
#else                              /* the frames are written once and not read again here: streaming stores */
#define RAW28_FRAME_STORE(p, v) __builtin_nontemporal_store((v), (p))
#endif
__global__ __launch_bounds__(256) void k_raw28_render(const uint8_t *__restrict__ raw, const uint8_t *__restrict__ h, size_t N,
                                                      const LineRec *__restrict__ lines, RenderConst R,
                                                      const int *__restrict__ tails, uint8_t *__restrict__ frames,
                                                      size_t frame_stride, int linesize)
{
    extern __shared__ int lds[];
    const int len = R.len, n = len + 16;
    int *S = lds, *A = lds + n;
    const int y = blockIdx.x;
    const LineRec L = lines[y];
    // a sample is one of 256 values and the levels are the scanline's: the equalisation (an fp64 division per
    // sample, :708-710) is evaluated once per VALUE, by the workgroup's 256 threads, and looked up per sample
    int *lut = lds + 2 * n;
    lut[threadIdx.x] = equalise_value((int)threadIdx.x, L, R);
    __syncthreads();
    // four samples per thread: one (unaligned) 32-bit load of the delayed raw bytes where all four exist, one 128-bit
    // store into the LDS (n = len + 16 is a multiple of 4 for the even line lengths the geometry allows; odd ones and the
    // ends of the stream take the sample-by-sample form)
    for (int x = 4 * (int)threadIdx.x; x < n; x += 4 * 256) {
        const size_t s = (size_t)L.pos + (size_t)x;
        if (x + 3 < n && s + 3 < N && R.base + s >= (unsigned long long)R.D) {
            uint32_t w, hw = 0xFFFFFFFFu;
            __builtin_memcpy(&w, raw + (s - (size_t)R.D), 4);
            if (R.mark) __builtin_memcpy(&hw, h + s, 4);
            int v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int b = (int)((w >> (8 * k)) & 0xFFu);
                if (R.mark && (int)((hw >> (8 * k)) & 0xFFu) < R.thr) b = 255;     // :590-591
                v[k] = lut[b];
            }
            *reinterpret_cast<int4 *>(&S[x]) = make_int4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4 && x + k < n; k++) {
                const size_t sk = s + (size_t)k;
                S[x + k] = lut[sk < N ? raw_delayed(raw, h, sk, R.D, R.thr, R.mark, R.base) : 0];
            }
        }
    }
    __syncthreads();
    uint32_t *dst = (uint32_t *)(frames + (size_t)L.field * frame_stride + (size_t)L.row * (size_t)linesize);
    if (R.no_sc) {
        for (int x = threadIdx.x; x < R.width; x += 256) {
            int Y = R.show_sc ? 128 : S[x];        // chroma stays 0 (:703)
            Y = Y < 0 ? 0 : (Y > 255 ? 255 : Y);
            dst[x] = (uint32_t)Y * 0x010101u;
        }
        return;
    }
    // The comb :731-742 only ever combines samples 4 apart, so the scanline is four interleaved sequences; a thread
    // takes eight consecutive positions of one of them (x = r + 4 j) through all six stages in registers, reading
    // the 16 samples that reach them from the LDS once -- two barriers per scanline instead of seven.
    //   A(x) = x < len ? S[x] - (S[x] + S[x + 4] + 1) / 2 : tail of the scanline before (x - len)        :731-734
    //   B(x) = x < len ? A(x) + A(x + 8) - A(x + 4) - A(x + 12) : A(x)                                  :736-737
    //   C(x) = x < len ? C(x) - (C(x) + C(x + 4)) / 2 : C(x), four times                                :739-742
    // (every read of a position below len stays below n = len + 16; what lies past n feeds nothing)
    {
        const int r = threadIdx.x & 3, q = threadIdx.x >> 2;
        const int *tl = tails + ((ptrdiff_t)y - 1) * 16;
        for (int j0 = q * 8; r + 4 * j0 < n; j0 += 64 * 8) {
            int v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) { const int x = r + 4 * (j0 + i); v[i] = x < n ? S[x] : 0; }
            int tv[16];                            // tail values where the window reaches past len
#pragma unroll
            for (int i = 0; i < 16; i++) { const int x = r + 4 * (j0 + i); tv[i] = (x >= len && x < n) ? tl[x - len] : 0; }
            // A: 15 values
#pragma unroll
            for (int i = 0; i < 15; i++) { const int x = r + 4 * (j0 + i); v[i] = x < len ? v[i] - (v[i] + v[i + 1] + 1) / 2 : tv[i]; }
            // B: 12 values
#pragma unroll
            for (int i = 0; i < 12; i++) { const int x = r + 4 * (j0 + i); v[i] = x < len ? v[i] + v[i + 2] - v[i + 1] - v[i + 3] : v[i]; }
            // C: 11, 10, 9, 8 values
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int i = 0; i < 11 - it; i++) { const int x = r + 4 * (j0 + i); v[i] = x < len ? v[i] - (v[i] + v[i + 1]) / 2 : v[i]; }
#pragma unroll
            for (int i = 0; i < 8; i++) { const int x = r + 4 * (j0 + i); if (x < n) A[x] = v[i]; }
        }
    }
    __syncthreads();
    const int *in = A;
    // `in` = int_chroma before the shift :744; final chroma[x] = x < 16 ? in[x] : in[x - 16] / 4
    for (int x = threadIdx.x; x < R.width; x += 256) {
        int Y;
        if (x < len) {
            const int chroma = x < 16 ? in[x] : in[x - 16] / 4;
            Y = R.show_sc ? (int)(int16_t)chroma + 128 : (int)(int16_t)(S[x] - chroma);     // :751-753, :760-763
        } else {
            Y = R.show_sc ? 128 : S[x];            // past the comb's range: luma as equalised, chroma 0
        }
        Y = Y < 0 ? 0 : (Y > 255 ? 255 : Y);
        RAW28_FRAME_STORE(dst + x, (uint32_t)Y * 0x010101u);       // RGBTRIPLET :366, alpha 0
    }
}

} // namespace

// ------------------------------------------------------------------------------------------ host
namespace {

// The walk over the sync runs: composite_layer() :622-693 and :789-830 with the buffer window of
// :277-332 reduced to its two numbers (begin, end of the buffered part of the stream).
struct RunWalk {
    const uint32_t *rs, *re;
    size_t nruns;
    mutable size_t k = 0;          // cursor: the searches move forward, or back by a fraction of a scanline
    // first run that ends after position i: [si, ei) clipped to i and E; si == ei == E when none
    void next(size_t i, size_t E, size_t &si, size_t &ei) const
    {
        while (k > 0 && (size_t)re[k - 1] > i) k--;
        while (k < nruns && (size_t)re[k] <= i) k++;
        if (k >= nruns || (size_t)rs[k] >= E) { si = ei = E; return; }
        si = std::max<size_t>(rs[k], i);
        ei = std::min<size_t>(re[k], E);
    }
};

// What the tool's sample buffer holds (:264-357): which stream position every record of the array was
// last filled from.  Only the calibration sums of :661-676 can run past the filled part (the search
// position jumps 0.3 scanlines ahead of a pulse, :655), and there they read whatever the records held
// before -- samples of an earlier window after the buffer has been moved down, zero-initialised
// records (hsync_dc_raw 0, raw 0) while the array has never been full.
struct BufMap {
    struct Seg { size_t k0, k1, abs0; };           // records [k0, k1) hold samples abs0 + (k - k0)
    std::vector<Seg> segs;                         // sorted, disjoint
    void assign(size_t k0, size_t k1, size_t abs0)
    {
        if (k0 >= k1) return;
        std::vector<Seg> out;
        for (const Seg &g : segs) {
            if (g.k1 <= k0 || g.k0 >= k1) { out.push_back(g); continue; }
            if (g.k0 < k0) out.push_back(Seg{g.k0, k0, g.abs0});
            if (g.k1 > k1) out.push_back(Seg{k1, g.k1, g.abs0 + (k1 - g.k0)});
        }
        out.push_back(Seg{k0, k1, abs0});
        std::sort(out.begin(), out.end(), [](const Seg &a, const Seg &b) { return a.k0 < b.k0; });
        segs.swap(out);
    }
    // records [k0, k1): stream pieces through `piece(abs_begin, abs_end)`, returns the number of records
    // that were never filled
    template <class F>
    size_t lookup(size_t k0, size_t k1, F piece) const
    {
        size_t zeros = 0, k = k0;
        for (const Seg &g : segs) {
            if (g.k1 <= k || g.k0 >= k1) continue;
            if (g.k0 > k) { zeros += g.k0 - k; k = g.k0; }
            const size_t e = std::min(g.k1, k1);
            piece(g.abs0 + (k - g.k0), g.abs0 + (e - g.k0));
            k = e;
        }
        if (k < k1) zeros += k1 - k;
        return zeros;
    }
};

} // namespace

struct ntscsim_raw28 {
    ntscsim_raw28_opts o;
    int device = 0;
    std::string err;
    // geometry / constants (compute_NTSC :247-256, preset_NTSC :395-402, main :936-946)
    double sample_rate, one_frame_time, one_scanline_time;
    unsigned len;
    int width, height, D;
    FrontConst K;
    FrontState init;
    bool chunk_forced = false;
    bool front_pin = true;         // one front-end workgroup per CU (NTSCSIM_RAW28_NOPIN=1: developer A/B switch)
    int warm_lines = 112, chunk = 4096;    // measured: a start 230 levels too high meets the truth after ~100 noisy scanlines
    int warm_boost = 0;            // scanlines of warm-up added after passes that left many links open (see the front end)
    bool warm_forced = false;      // the warm-up was set through ntscsim_raw28_debug_set_speculation(): no boost
    int exact_lines = 30;          // the last scanlines of that warm-up walked sample by sample (NTSCSIM_RAW28_EXACT; >= warm_lines: all)
    bool force_tail_rounds = false;    // NTSCSIM_RAW28_TAILROUNDS=1, test hook: take the path of a first guess that did not settle
    int group_fields = 160;        // fields per group of the back half's pipeline (NTSCSIM_RAW28_GROUP: developer A/B switch)
    int follow_lanes = 16;         // chunks per wavefront of sweep 2 (NTSCSIM_RAW28_LANES: 1..64)
    size_t front_seg = FRONT_SEG;  // samples per front-end segment (NTSCSIM_RAW28_SEG: test hook, the segment loop on small captures)
    int max_chunks = 16384;        // sweep 2: 256 wavefronts (NTSCSIM_RAW28_CHUNKS: developer A/B switch)
    // decoder state: levels (:553-554), stream position; kept from push to push of a stream
    double blank = 0, white = 192;
    uint64_t read_pos = 0;
    // stream state (ntscsim_raw28_stream_*; a whole-capture decode is a reset plus one final push).  The
    // device buffers raw / h hold the samples [base, base + cnt) of the stream; every other position below
    // is RELATIVE to base and moves down with it when the buffer is compacted.
    uint64_t base = 0;
    size_t cnt = 0;                // samples held
    size_t front_done = 0;         // samples the front end has processed
    FrontState front_state;        // its exact state after sample front_done - 1
    size_t Bw = 0, Rd = 0, Ew = 0; // the tool's buffer window: begin, read position, end
    BufMap bm;                     // what the tool's sample array holds (stale records included)
    int tail_carry[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // comb tail of the last scanline rendered
    bool eof = false;
    uint64_t fields_total = 0;
    int64_t stats[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t last_n = 0;
    bool mutated = false;        // the current push has changed the stream state
    bool broken = false;         // a push failed after it had changed the state: only a reset helps
    // device scratch
    Buf<uint8_t> raw, h, raw_alt, h_alt, tmp;
    Buf<FrontState> st_begin, st_end, st_prev, st_a0;
    Buf<double> lvplane, sumplane, lv_begin, lv_end, lv_prev, ckpt;
    Buf<int> flags, counters, tails_a, tails_b;
    Buf<unsigned long long> segcnt, segoff;
    Buf<uint32_t> rstart, rend;
    Buf<CalRange> cal_rg;
    Buf<CalSums> cal_out;
    Buf<LineRec> lines;
    hipEvent_t ev_cal = nullptr, ev_cal2 = nullptr;    // behind a group's calibration sums on their way back (two groups in flight)
    HostBuf<LineRec> lines_h;      // pinned: what the walk writes and the copies read while it goes on
    HostBuf<CalRange> cal_h;
    HostBuf<uint32_t> rs_h, re_h;  // the sync runs (pinned)
    HostBuf<CalSums> part_h;
    Buf<int> gcount;               // per group of fields: scanlines whose comb tail the confirming round changed
    ~ntscsim_raw28() { if (ev_cal) (void)hipEventDestroy(ev_cal); if (ev_cal2) (void)hipEventDestroy(ev_cal2); }
};

#define R28CHK(d, call)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            (d)->err = std::string(#call) + ": " + hipGetErrorString(e__);                 \
            return NTSCSIM_E_HIP;                                                          \
        }                                                                                  \
    } while (0)

// The chunk length of sweep 2 (see raw28_stream_push_impl): q sub-chunks of whole 64-sample superblocks, about 2048
// samples each, near `target` samples, as close to a whole number of (nominal) scanlines as such a length gets.
static void raw28_pick_chunk(double line_t, double target, int *chunk, int *q_out)
{
    target = std::max(target, 16.0 * line_t);
    double best = 1e300;
    *chunk = 0; *q_out = 0;
    for (int q = 1; q <= 16; q++)
        for (int sub = 2048; sub <= 4096; sub += FOLLOW_SB) {
            const double len_c = (double)q * (double)sub;
            if (len_c < 0.55 * target || len_c > 2.0 * target) continue;
            const double lines = len_c / line_t;
            const double drift = std::fabs(lines - std::nearbyint(lines)) * line_t;       // samples per lane
            const double cost = 64.0 * drift + 100.0 * std::fabs(std::log(len_c / target)) - q;
            if (cost < best) { best = cost; *chunk = q * sub; *q_out = q; }
        }
}
extern "C" void ntscsim_raw28_debug_pick_chunk(double scanline_samples, double target_samples, int *chunk, int *subchunks)
{
    if (chunk && subchunks) raw28_pick_chunk(scanline_samples, target_samples, chunk, subchunks);
}

static void raw28_geometry(const ntscsim_raw28_opts &o, double &rate, double &frame_t, double &line_t,
                           unsigned &len, int &width, int &height)
{
    rate = o.sample_rate > 0 ? o.sample_rate : ((315000000.00 * 8.0) / 88.00);   // :237-245
    frame_t = rate / (30000.00 / 1001.00);                                        // :249
    line_t = frame_t / 525.00;                                                    // :250
    len = (unsigned int)(line_t + 0.5);                                           // :251
    height = 262;                                                                 // :398
    width = (int)((len + 1) & (~1u));                                             // :399
}

extern "C" void ntscsim_raw28_opts_init(ntscsim_raw28_opts *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t)sizeof(*o);
}

extern "C" int ntscsim_raw28_parse_argv(ntscsim_raw28_opts *o, int argc, const char *const *argv, int start)
{
    if (!o || (argc > 0 && !argv)) return NTSCSIM_E_ARG;
    for (int i = start; i < argc;) {                       // parse_argv :442-520
        const char *a = argv[i++];
        if (*a != '-') return NTSCSIM_E_FLAG;              // "Unhandled arg"
        do { a++; } while (*a == '-');
        if (!std::strcmp(a, "h") || !std::strcmp(a, "help")) return NTSCSIM_E_HELP;
        else if (!std::strcmp(a, "marksig")) o->mark_sync = 1;
        else if (!std::strcmp(a, "noequ")) o->disable_equalization = 1;
        else if (!std::strcmp(a, "nowequ")) o->disable_wp_equ = 1;
        else if (!std::strcmp(a, "nosig")) o->disable_sync = 1;
        else if (!std::strcmp(a, "nosc")) o->disable_subcarrier = 1;
        else if (!std::strcmp(a, "showsc")) o->show_subcarrier = 1;
        else if (!std::strcmp(a, "s")) {
            if (i >= argc || !argv[i]) return NTSCSIM_E_FLAG;
            const char *v = argv[i++];
            if (!std::strcmp(v, "ntsc28")) o->sample_rate = 0;                    // main :918-928
            else if (!std::strcmp(v, "40mhz")) o->sample_rate = 40000000.00;
            else if (*v >= '0' && *v <= '9') o->sample_rate = std::atof(v);
            else o->sample_rate = 0;                       // "Unknown -s preset": falls back to ntsc28
        } else if (!std::strcmp(a, "width")) {
            if (i >= argc || !argv[i]) return NTSCSIM_E_FLAG;
            if ((int)std::strtoul(argv[i++], nullptr, 0) < 32) return NTSCSIM_E_FLAG;
        } else if (!std::strcmp(a, "i") || !std::strcmp(a, "o")) {
            if (i >= argc || !argv[i]) return NTSCSIM_E_FLAG;
            i++;
        } else if (!std::strcmp(a, "422") || !std::strcmp(a, "420") || !std::strcmp(a, "inntsc")) {
        } else return NTSCSIM_E_FLAG;                      // "Unknown switch"
    }
    return NTSCSIM_OK;
}

extern "C" int ntscsim_raw28_geometry(const ntscsim_raw28_opts *o, int *width, int *height, int *scanline_samples)
{
    if (!o) return NTSCSIM_E_ARG;
    double r, f, l; unsigned len; int w, h;
    raw28_geometry(*o, r, f, l, len, w, h);
    if (len < 64 || len > 4000) return NTSCSIM_E_PARAM;    // the tool's scratch arrays hold 4096 samples :258
    if (width) *width = w;
    if (height) *height = h;
    if (scanline_samples) *scanline_samples = (int)len;
    return NTSCSIM_OK;
}

extern "C" int ntscsim_raw28_create(const ntscsim_raw28_opts *o, int device, ntscsim_raw28 **out)
{
    if (!o || !out) return NTSCSIM_E_ARG;
    if (o->struct_size != sizeof(ntscsim_raw28_opts)) return NTSCSIM_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return NTSCSIM_E_NODEV;
    int w, h, sl;
    const int rc = ntscsim_raw28_geometry(o, &w, &h, &sl);
    if (rc != NTSCSIM_OK) return rc;
    ntscsim_raw28 *d = new ntscsim_raw28();
    d->o = *o;
    d->device = device;
    raw28_geometry(*o, d->sample_rate, d->one_frame_time, d->one_scanline_time, d->len, d->width, d->height);
    d->D = (int)(size_t)((d->one_scanline_time * 0.075 * 0.75) * 0.5);            // main :937
    if (d->D < 1) { delete d; return NTSCSIM_E_PARAM; }    // (the tool would index an empty delay line)
    // detector filters: setFilter :80-88 with main :941, then one frame time of lowpass(128) :942
    {
        const double rate = d->sample_rate, hz = d->sample_rate / (d->one_scanline_time * 0.075 * 0.75);
        const double timeInterval = 1.0 / rate;
        const double tau = 1 / (hz * 2 * M_PI);
        d->K.alpha = timeInterval / (tau + timeInterval);
        double prev = 0;
        for (size_t j = 0; j < d->one_frame_time; j++) {
            const double stage1 = 128 * d->K.alpha;
            const double stage2 = prev - (prev * d->K.alpha);
            prev = stage1 + stage2;
        }
        d->init.p0 = d->init.p1 = d->init.p2 = prev;
        d->init.level = 128.0;                                                    // :552
    }
    d->K.a_fast = 1.0 / (d->one_scanline_time * 0.07 * 0.75);                     // :564
    d->K.om_fast = 1.0 - d->K.a_fast;
    d->K.a_slow = 1.0 / (d->one_frame_time * 0.6);                                // :568
    d->K.om_slow = 1.0 - d->K.a_slow;
    d->K.om_slow_sb = std::pow(d->K.om_slow, (double)FOLLOW_SB);
    d->K.slow_margin = 1.1 * (double)FOLLOW_SB * d->K.a_slow * 255.0;            // (0.0626 at 8 x fsc)
    d->K.thr = (int)(uint8_t)(192 * 0.25 * 0.5);                                  // :553
    if (const char *e = std::getenv("NTSCSIM_RAW28_NOPIN")) d->front_pin = std::atoi(e) == 0;
    if (const char *e = std::getenv("NTSCSIM_RAW28_SEG")) { const long long v = std::atoll(e); if (v >= 4096) d->front_seg = (size_t)v; }
    if (const char *e = std::getenv("NTSCSIM_RAW28_CHUNKS")) { const int v = std::atoi(e); if (v >= 64) d->max_chunks = v; }
    if (const char *e = std::getenv("NTSCSIM_RAW28_EXACT")) { const int v = std::atoi(e); if (v >= 0) d->exact_lines = v; }
    if (const char *e = std::getenv("NTSCSIM_RAW28_TAILROUNDS")) d->force_tail_rounds = std::atoi(e) != 0;
    if (const char *e = std::getenv("NTSCSIM_RAW28_GROUP")) { const int v = std::atoi(e); if (v >= 1) d->group_fields = v; }
    if (const char *e = std::getenv("NTSCSIM_RAW28_LANES")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) d->follow_lanes = v; }
    *out = d;
    return NTSCSIM_OK;
}

extern "C" void ntscsim_raw28_destroy(ntscsim_raw28 *d) { delete d; }
extern "C" const char *ntscsim_raw28_last_error(const ntscsim_raw28 *d) { return d ? d->err.c_str() : ""; }
extern "C" int ntscsim_raw28_get_levels(const ntscsim_raw28 *d, double *blank, double *white, uint64_t *read_pos)
{
    if (!d) return NTSCSIM_E_ARG;
    if (blank) *blank = d->blank;
    if (white) *white = d->white;
    if (read_pos) *read_pos = d->read_pos;
    return NTSCSIM_OK;
}
extern "C" void ntscsim_raw28_debug_set_speculation(ntscsim_raw28 *d, int warm_lines, int chunk_samples)
{
    if (!d) return;
    if (warm_lines >= 0) { d->warm_lines = warm_lines; d->warm_forced = true; d->warm_boost = 0; }
    if (chunk_samples >= 64) { d->chunk = (chunk_samples + 15) & ~15; d->chunk_forced = true; }
}
extern "C" void ntscsim_raw28_debug_stats(const ntscsim_raw28 *d, int64_t out[16])
{
    if (d && out) std::memcpy(out, d->stats, sizeof(d->stats));
}
extern "C" int ntscsim_raw28_debug_read_front(ntscsim_raw28 *d, uint8_t *hs, size_t n)
{
    if (!d || !hs || n > d->last_n) return NTSCSIM_E_ARG;
    R28CHK(d, hipSetDevice(d->device));
    R28CHK(d, hipMemcpy(hs, d->h.p, n, hipMemcpyDeviceToHost));
    return NTSCSIM_OK;
}


// ---- a stream, push by push --------------------------------------------------------------------------
static void raw28_stream_reset(ntscsim_raw28 *d)
{
    d->base = 0; d->cnt = 0; d->front_done = 0;
    d->front_state = d->init;
    d->Bw = d->Rd = d->Ew = 0;
    d->bm.segs.clear();
    std::memset(d->tail_carry, 0, sizeof(d->tail_carry));
    d->eof = false;
    d->fields_total = 0;
    d->blank = (uint8_t)0; d->white = (uint8_t)192; d->read_pos = 0;              // :553-554
    std::memset(d->stats, 0, sizeof(d->stats));
    d->last_n = 0;
    d->broken = false;
}

// make room for `need` bytes in a / b (same capacity policy), keeping the first `keep` bytes of both
static int raw28_grow(ntscsim_raw28 *d, size_t keep, size_t need, hipStream_t st)
{
    if (need <= d->raw.cap && need <= d->h.cap) return NTSCSIM_OK;
    const size_t want = need + need / 4 + 4096;
    Buf<uint8_t> nr, nh;
    R28CHK(d, nr.ensure(want));
    R28CHK(d, nh.ensure(want));
    if (keep) {
        R28CHK(d, hipMemcpyAsync(nr.p, d->raw.p, keep, hipMemcpyDeviceToDevice, st));
        R28CHK(d, hipMemcpyAsync(nh.p, d->h.p, keep, hipMemcpyDeviceToDevice, st));
        R28CHK(d, hipStreamSynchronize(st));
    }
    std::swap(d->raw.p, nr.p); std::swap(d->raw.cap, nr.cap);
    std::swap(d->h.p, nh.p); std::swap(d->h.cap, nh.cap);
    return NTSCSIM_OK;            // (nr / nh free the old buffers)
}

// One push of a stream: `n` more samples (host or device memory), `final` = the stream ends with them.
// Decodes every field the tool would have produced so far whose buffer window (:307, 2048 scanlines) is
// complete -- at most max_fields of them; the rest come out of later pushes (n = 0 is allowed).
static int raw28_stream_push_impl(ntscsim_raw28 *d, const void *samples, bool on_device, size_t n, bool final, void *frames_dev,
                                  size_t frame_stride, int linesize, int max_fields, int *n_fields)
{
    if (!d || (n > 0 && !samples) || !n_fields || max_fields < 0 || (max_fields > 0 && !frames_dev)) return NTSCSIM_E_ARG;
    if (d->broken) {
        d->err = "an earlier push of this stream failed half way: ntscsim_raw28_stream_reset() starts a new one";
        return NTSCSIM_E_ARG;
    }
    if (d->eof && n > 0) { d->err = "the stream has ended: ntscsim_raw28_stream_reset() starts a new one"; return NTSCSIM_E_ARG; }
    if (d->cnt + n >= 0xFFFFFFF0ull) return NTSCSIM_E_SIZE;
    if (max_fields > 0 && (linesize < d->width * 4 || (linesize & 3) || frame_stride < (size_t)linesize * (size_t)d->height))
        return NTSCSIM_E_SIZE;
    *n_fields = 0;
    R28CHK(d, hipSetDevice(d->device));
    hipStream_t st = nullptr;
    const unsigned len = d->len;
    d->mutated = true;           // from here on an error leaves the stream half advanced (see the wrapper)
    if (final) d->eof = true;

    // wall-clock split of the call (every phase ends in a stream synchronisation), stats[6..11] in us
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](int slot) {
        const auto t = std::chrono::steady_clock::now();
        d->stats[slot] += (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(t - t_last).count();
        t_last = t;
    };
    // ---- (0) the new samples join the buffer (the kernels read whole 16-byte pieces: 64 zero bytes follow)
    {
        const int rc = raw28_grow(d, d->cnt, d->cnt + n + 64, st);
        if (rc != NTSCSIM_OK) return rc;
    }
    if (n) R28CHK(d, hipMemcpyAsync(d->raw.p + d->cnt, samples, n, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    d->cnt += n;
    R28CHK(d, hipMemsetAsync(d->raw.p + d->cnt, 0, 64, st));
    const size_t N = d->cnt;
    const uint8_t *raw = d->raw.p;
    d->last_n = N;
    R28CHK(d, d->counters.ensure(4));

    // ---- (1) front end over the new samples [front_done, N), at most FRONT_SEG of them at a time (the fp64
    // plane between the two sweeps is 8 bytes per sample)
    // (warm_boost: scanlines added by earlier front-end passes of this decoder that left many links open -- a noisy source)
    const int warm = (int)(((size_t)(d->warm_lines + d->warm_boost) * len + 63) & ~(size_t)63);
    const int warm_exact = (int)std::min((size_t)warm, ((size_t)d->exact_lines * len + 63) & ~(size_t)63);
    while (N > d->front_done) {
        const size_t o0 = d->front_done, o1 = std::min(N, o0 + d->front_seg), fresh = o1 - o0;
        const size_t a0 = (o0 + 15) & ~(size_t)15;
        R28CHK(d, d->st_a0.ensure(1));
        hipLaunchKernelGGL(k_raw28_head, dim3(1), dim3(64), 0, st, raw, d->h.p, o0, std::min(a0, o1), d->K, d->front_state, d->st_a0.p);
        if (a0 >= o1) {
            R28CHK(d, hipMemcpyAsync(&d->front_state, d->st_a0.p, sizeof(FrontState), hipMemcpyDeviceToHost, st));
            R28CHK(d, hipStreamSynchronize(st));
            d->front_done = o1;
            continue;
        }
        // one lane per chunk in sweep 2: a lane's run is warm-up + chunk samples long whatever the chunk count and
        // every lane reads warm-up + chunk values of the plane, so no more chunks than it takes to occupy the
        // CUs with one wavefront each
        size_t chunk = (size_t)d->chunk;
        int Q = 0;
        if (!d->chunk_forced) {
            // The cheap part of sweep 2's warm-up pays where all 64 lanes of a wavefront (64 consecutive chunks, at the
            // same row of their columns) are outside the sync tips together, i.e. where the chunk is a whole number of
            // scanlines long (nominal ones: a capture whose lines are longer or shorter drifts by that much per lane and
            // takes the exact walk more often, nothing else).  It also has to be Q sub-chunks of whole 64-sample
            // superblocks, about 2048 samples each, for sweep 1: the best such length near stream / max_chunks.
            int pm = 0, pq = 0;
            raw28_pick_chunk(d->one_scanline_time, (double)fresh / (double)d->max_chunks, &pm, &pq);
            if (pq > 0) { chunk = (size_t)pm; Q = pq; }
        }
        if (chunk > (size_t)INT_MAX / 2) { d->err = "front end: chunk too long"; return NTSCSIM_E_SIZE; }
        const int m = (int)chunk;
        const int nchunks = (int)((o1 - a0 + chunk - 1) / chunk);
        if (Q == 0) {                              // a forced chunk length: sub-chunks of sweep 1 of about 2048 samples each
            Q = 8;
            while (Q > 1 && (m % (16 * Q) != 0 || m / Q < 2048)) Q >>= 1;
        }
        const int ms = m / Q;
        const int nsub = (int)((o1 - a0 + (size_t)ms - 1) / (size_t)ms);
        const int w1 = std::min(warm, LP_WARM);
        R28CHK(d, d->lvplane.ensure((size_t)nchunks * (size_t)m));
        // the cheap part of sweep 2's warm-up needs whole 64-sample superblocks in every sub-chunk
        const bool have_sum = ms % FOLLOW_SB == 0;
        const int cheap = (have_sum && warm > warm_exact) ? warm - warm_exact : 0;
        if (have_sum) R28CHK(d, d->sumplane.ensure((size_t)nchunks * (size_t)(m / FOLLOW_SB) * 2));
        lvpair *const SUM = have_sum ? (lvpair *)d->sumplane.p : nullptr;
        R28CHK(d, d->st_begin.ensure((size_t)nsub));
        R28CHK(d, d->st_end.ensure((size_t)nsub));
        R28CHK(d, d->st_prev.ensure((size_t)nsub));
        R28CHK(d, d->flags.ensure((size_t)std::max(nsub, nchunks)));
        R28CHK(d, d->lv_begin.ensure((size_t)nchunks));
        R28CHK(d, d->lv_end.ensure((size_t)nchunks));
        R28CHK(d, d->lv_prev.ensure((size_t)nchunks));
        const int ncp = m / FOLLOW_CK + 1;
        R28CHK(d, d->ckpt.ensure((size_t)nchunks * (size_t)ncp));
        auto any_bad = [&](int &nbad) -> int {
            R28CHK(d, hipMemcpyAsync(&nbad, d->counters.p, sizeof(int), hipMemcpyDeviceToHost, st));
            R28CHK(d, hipStreamSynchronize(st));
            return NTSCSIM_OK;
        };
        // sweep 1: the low-passes
        const unsigned lp_blocks = (unsigned)(((nchunks + 63) / 64) * Q);
        hipLaunchKernelGGL(k_raw28_lp, dim3(lp_blocks), dim3(64), 0, st, raw, a0, o1, m, ms, Q, w1, nchunks, nsub, d->K.alpha,
                           (const FrontState *)d->st_a0.p, (lvpair *)d->lvplane.p, d->st_begin.p, d->st_end.p,
                           (const FrontState *)nullptr, (const int *)nullptr, SUM, d->K.a_slow, d->K.om_slow);
        for (;;) {
            R28CHK(d, hipMemsetAsync(d->counters.p, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_raw28_links, dim3((nsub + 255) / 256), dim3(256), 0, st, d->st_begin.p, d->st_end.p,
                               nsub, d->flags.p, d->counters.p);
            int nbad = 0;
            { const int rc = any_bad(nbad); if (rc != NTSCSIM_OK) return rc; }
            if (nbad == 0) break;
            d->stats[0]++; d->stats[1] += nbad;
            // repair round: flagged sub-chunks restart from their predecessor's end state as it is NOW
            R28CHK(d, hipMemcpyAsync(d->st_prev.p, d->st_end.p, (size_t)nsub * sizeof(FrontState), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(k_raw28_lp, dim3(lp_blocks), dim3(64), 0, st, raw, a0, o1, m, ms, Q, w1, nchunks, nsub, d->K.alpha,
                               (const FrontState *)d->st_a0.p, (lvpair *)d->lvplane.p, d->st_begin.p, d->st_end.p,
                               (const FrontState *)d->st_prev.p, (const int *)d->flags.p, SUM, d->K.a_slow, d->K.om_slow);
        }
        // sweep 2: the follower
        // Few chunks per wavefront where the warm-up has a cheap part: a superblock is taken in closed form only if ALL
        // lanes of the wavefront can, and a lane inside a field's broad sync pulses cannot for lines on end (1.1 % of
        // the scanlines: with 64 lanes 16 scanlines apart one of them nearly always is).  The walk is one dependent
        // chain per wavefront whatever its lane count, and there are SIMDs to spare: one wavefront per SIMD.
        const int lpw = have_sum ? d->follow_lanes : 64;
        const unsigned fw_blocks = (unsigned)((nchunks + lpw - 1) / lpw);
        const size_t pin = !d->front_pin ? 0 : fw_blocks <= 256 ? FOLLOW_PIN_LDS : fw_blocks <= 1024 ? 20 * 1024 : 0;
        auto launch_follow = [&](size_t lds, const double *prev, const int *fl) {
            hipLaunchKernelGGL(k_raw28_follow, dim3(fw_blocks), dim3(64), lds, st, (const lvpair *)d->lvplane.p, (const lvpair *)SUM,
                               a0, o1, m, warm, cheap, nchunks, d->K, (const FrontState *)d->st_a0.p, d->h.p, d->lv_begin.p,
                               d->lv_end.p, d->ckpt.p, ncp, prev, fl, lpw);
        };
        if (pin) R28CHK(d, hipFuncSetAttribute((const void *)k_raw28_follow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pin));
        launch_follow(pin, (const double *)nullptr, (const int *)nullptr);
        for (int round = 0;; round++) {
            R28CHK(d, hipMemsetAsync(d->counters.p, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_raw28_links1, dim3((nchunks + 255) / 256), dim3(256), 0, st, (const double *)d->lv_begin.p,
                               (const double *)d->lv_end.p, nchunks, d->flags.p, d->counters.p);
            int nbad = 0;
            { const int rc = any_bad(nbad); if (rc != NTSCSIM_OK) return rc; }
            // A warm-up that leaves more than one link in a hundred open is too short for this source (its noise slows the
            // follower's contraction: profiles/r04_raw28_noise.txt); its closed-form part costs 15 us per scanline, a
            // repaired chunk far more: the next passes of this decoder get 16 scanlines more, up to 64.  (Speed only.)
            if (round == 0 && cheap > 0 && !d->warm_forced && nchunks >= 256 && nbad * 100 > nchunks && d->warm_boost < 64)
                d->warm_boost += 16;
            if (nbad == 0) break;
            d->stats[0]++; d->stats[1] += nbad;
            R28CHK(d, hipMemcpyAsync(d->lv_prev.p, d->lv_end.p, (size_t)nchunks * sizeof(double), hipMemcpyDeviceToDevice, st));
            launch_follow(0, (const double *)d->lv_prev.p, (const int *)d->flags.p);
        }
        // the exact state after the last sample: where the next segment / push starts
        FrontState fin;
        double fin_level = 0;
        R28CHK(d, hipMemcpyAsync(&fin, d->st_end.p + (nsub - 1), sizeof(FrontState), hipMemcpyDeviceToHost, st));
        R28CHK(d, hipMemcpyAsync(&fin_level, d->lv_end.p + (nchunks - 1), sizeof(double), hipMemcpyDeviceToHost, st));
        R28CHK(d, hipStreamSynchronize(st));
        fin.level = fin_level;
        d->front_state = fin;
        d->front_done = o1;
    }

    lap(6);
    // ---- (2) runs of h < thr over the buffer
    const size_t nseg = (N + RUN_BLOCK - 1) / RUN_BLOCK;
    size_t nruns_h = 0;                            // the runs on the host: pinned, the walk reads them in place
    if (nseg > 0) {
        R28CHK(d, d->segcnt.ensure(nseg + 1));
        R28CHK(d, d->segoff.ensure(nseg + 1));
        hipLaunchKernelGGL(k_raw28_run_count, dim3((unsigned)nseg), dim3(RUN_T), 0, st, d->h.p, N, d->K.thr, d->segcnt.p);
        R28CHK(d, hipMemsetAsync(d->segcnt.p + nseg, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_raw28_scan_counts, dim3(1), dim3(SCAN_T), 0, st, d->segcnt.p, d->segoff.p, nseg + 1);
        unsigned long long total = 0;
        R28CHK(d, hipMemcpyAsync(&total, d->segoff.p + nseg, sizeof(total), hipMemcpyDeviceToHost, st));
        R28CHK(d, hipStreamSynchronize(st));
        const size_t nruns = (size_t)(total & 0xFFFFFFFFull);
        if ((size_t)(total >> 32) != nruns) { d->err = "run extraction: starts != ends"; return NTSCSIM_E_INTERNAL; }
        if (nruns > 0) {
            R28CHK(d, d->rstart.ensure(nruns));
            R28CHK(d, d->rend.ensure(nruns));
            hipLaunchKernelGGL(k_raw28_run_scatter, dim3((unsigned)nseg), dim3(RUN_T), 0, st, d->h.p, N, d->K.thr,
                               d->segoff.p, d->rstart.p, d->rend.p);
            R28CHK(d, d->rs_h.ensure(nruns));
            R28CHK(d, d->re_h.ensure(nruns));
            R28CHK(d, hipMemcpyAsync(d->rs_h.p, d->rstart.p, nruns * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            R28CHK(d, hipMemcpyAsync(d->re_h.p, d->rend.p, nruns * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            R28CHK(d, hipStreamSynchronize(st));
            nruns_h = nruns;
        }
        d->stats[3] += (int64_t)nruns;
    }

    lap(7);
    // ---- (3) the walk: field loop main() :1006-1019 around composite_layer()'s searches -- with (4) (5) and the rendering
    // of the fields it has finished running on the GPU BEHIND it (round 4): the walk is 1.2 ms of strictly serial host
    // work for 600 fields (7 ns per sync run) during which the GPU used to wait.  Every GROUP_FIELDS fields the
    // calibration ranges of the group go out (k_raw28_cal, sums back through pinned memory, an event); a group later --
    // the sums have long arrived -- the host does the group's level recurrence and queues its scanline records, the
    // clearing of its frames, its comb tails (the serial first guess + ONE confirming round) and its rendering.  Whether
    // every group's round did confirm is looked at once, at the end; if one did not (never seen), the tails are redone by
    // rounds over all scanlines and everything is rendered again.
    RunWalk W;
    W.rs = d->rs_h.p; W.re = d->re_h.p; W.nruns = nruns_h;
    const size_t CAP = (size_t)len * 2048;                                        // open_src :353
    const size_t L30 = (size_t)(int)(len * 0.3), L06 = (size_t)(int)(len * 0.06), L02 = (size_t)(int)(len * 0.02);
    size_t &Bw = d->Bw, &Rd = d->Rd, &Ew = d->Ew;   // buffer begin, read position, buffer end (relative to base)
    BufMap &bm = d->bm;
    const int GROUP_FIELDS = d->group_fields;
    RenderConst RC;
    RC.len = (int)len; RC.width = d->width; RC.D = d->D; RC.thr = d->K.thr;
    RC.mark = d->o.mark_sync ? 1 : 0; RC.no_equ = d->o.disable_equalization ? 1 : 0;
    RC.no_wequ = d->o.disable_wp_equ ? 1 : 0; RC.no_sc = d->o.disable_subcarrier ? 1 : 0;
    RC.show_sc = d->o.show_subcarrier ? 1 : 0;
    RC.base = (unsigned long long)d->base;
    const size_t render_lds = ((size_t)2 * (len + 16) + 256) * sizeof(int);
    // capacities known before the walk (nothing may move while copies are in flight): a field advances the read position
    // by at least 240 scanlines (:836-845) and holds at most `height` of them; every calibration range stems from a run
    const size_t nf_cap = std::min<size_t>((size_t)max_fields, (N > Rd ? N - Rd : 0) / ((size_t)len * 240) + 2);
    const size_t lines_cap = nf_cap * (size_t)d->height;
    R28CHK(d, d->lines_h.ensure(lines_cap + 1));
    R28CHK(d, d->lines.ensure(lines_cap + 1));
    LineRec *const lines_h = d->lines_h.p;
    size_t nlines = 0;
    size_t cal_cap = 2 * nruns_h + 4096, ncal = 0;
    R28CHK(d, d->cal_h.ensure(cal_cap));
    R28CHK(d, d->part_h.ensure(cal_cap));
    R28CHK(d, d->cal_rg.ensure(cal_cap));
    R28CHK(d, d->cal_out.ensure(cal_cap));
    std::vector<int> cal_zero;                     // per pulse: never-filled records inside its range
    std::vector<int> cal_field;                    // number of pulses seen before each field is rendered
    std::vector<CalSums> sums;                     // per pulse
    const size_t max_groups = nf_cap / (size_t)GROUP_FIELDS + 2;
    R28CHK(d, d->gcount.ensure(max_groups));
    R28CHK(d, hipMemsetAsync(d->gcount.p, 0, max_groups * sizeof(int), st));
    int *tails_a = nullptr, *tails_b = nullptr;    // row 0 = the tail carried in; row y + 1 = the tail of scanline y
    if (!RC.no_sc && lines_cap > 0) {
        const size_t trows = lines_cap + 1;
        R28CHK(d, d->tails_a.ensure(trows * 16));
        R28CHK(d, d->tails_b.ensure(trows * 16));
        tails_a = d->tails_a.p; tails_b = d->tails_b.p;
        R28CHK(d, hipMemsetAsync(tails_a, 0, trows * 16 * sizeof(int), st));
        R28CHK(d, hipMemsetAsync(tails_b, 0, trows * 16 * sizeof(int), st));
        R28CHK(d, hipMemcpyAsync(tails_a, d->tail_carry, sizeof(d->tail_carry), hipMemcpyHostToDevice, st));
        R28CHK(d, hipMemcpyAsync(tails_b, d->tail_carry, sizeof(d->tail_carry), hipMemcpyHostToDevice, st));
    }
    if (!d->ev_cal) R28CHK(d, hipEventCreateWithFlags(&d->ev_cal, hipEventDisableTiming));
    if (!d->ev_cal2) R28CHK(d, hipEventCreateWithFlags(&d->ev_cal2, hipEventDisableTiming));
    struct Group { int idx, f0, f1; size_t l0, l1, c0, c1, p0, p1; };
    Group grp{0, 0, 0, 0, 0, 0, 0, 0, 0}, pending{-1, 0, 0, 0, 0, 0, 0, 0, 0};
    size_t ci_done = 0;                            // pulses the level recurrence has consumed
    // a calibration range joins the list (the list grows only with nothing in flight)
    auto cal_push = [&](const CalRange &r) -> int {
        if (ncal == cal_cap) {
            R28CHK(d, hipStreamSynchronize(st));
            const size_t ncap = cal_cap * 2;
            std::vector<CalRange> keep(d->cal_h.p, d->cal_h.p + ncal);
            std::vector<CalSums> keepp(d->part_h.p, d->part_h.p + ncal);
            R28CHK(d, d->cal_h.ensure(ncap));
            R28CHK(d, d->part_h.ensure(ncap));
            R28CHK(d, d->cal_rg.ensure(ncap));
            R28CHK(d, d->cal_out.ensure(ncap));
            std::memcpy(d->cal_h.p, keep.data(), ncal * sizeof(CalRange));
            std::memcpy(d->part_h.p, keepp.data(), ncal * sizeof(CalSums));
            cal_cap = ncap;
        }
        d->cal_h.p[ncal++] = r;
        return NTSCSIM_OK;
    };
    // (4a) the sums of a group's calibration pulses :661-676 start on the GPU
    auto submit_levels = [&](const Group &g) -> int {
        if (g.c1 > g.c0) {
            const size_t n = g.c1 - g.c0;
            R28CHK(d, hipMemcpyAsync(d->cal_rg.p + g.c0, d->cal_h.p + g.c0, n * sizeof(CalRange), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_raw28_cal, dim3((unsigned)n), dim3(64), 0, st, raw, d->h.p, d->cal_rg.p + g.c0,
                               d->cal_out.p + g.c0, d->D, d->K.thr, d->o.mark_sync ? 1 : 0, (unsigned long long)d->base);
            R28CHK(d, hipMemcpyAsync(d->part_h.p + g.c0, d->cal_out.p + g.c0, n * sizeof(CalSums), hipMemcpyDeviceToHost, st));
        }
        R28CHK(d, hipEventRecord((g.idx & 1) ? d->ev_cal2 : d->ev_cal, st));
        return NTSCSIM_OK;
    };
    // (4b) (5) the eight-tap level recurrence :683-688 of a group on the host, then everything else of it on the GPU
    auto complete_group = [&](const Group &g) -> int {
        R28CHK(d, hipEventSynchronize((g.idx & 1) ? d->ev_cal2 : d->ev_cal));
        sums.resize(g.p1, CalSums{0, 0, 0, 0});
        for (size_t k = g.c0; k < g.c1; k++) {
            CalSums &t = sums[d->cal_h.p[k].pulse];
            const CalSums &q = d->part_h.p[k];
            t.mina += q.mina; t.mind += q.mind; t.maxa += q.maxa; t.maxd += q.maxd;
        }
        for (size_t k = g.p0; k < g.p1; k++) sums[k].mind += cal_zero[k];       // zero records: below threshold, raw 0
        size_t li = g.l0;
        for (int f = g.f0; f < g.f1; f++) {
            for (; ci_done < (size_t)cal_field[(size_t)f]; ci_done++) {
                int mina = sums[ci_done].mina, maxa = sums[ci_done].maxa;
                if (sums[ci_done].mind > 0) mina /= sums[ci_done].mind;
                if (sums[ci_done].maxd > 0) maxa /= sums[ci_done].maxd;
                int t = (int)(maxa + ((maxa - mina) / (0.25 + 0.125)));
                t = std::min(std::max(t, maxa + 1), 240);
                const int nwhite = (uint8_t)t, nblack = maxa;
                const double a = 1.0 / 8.0;
                d->white = (d->white * (1.0 - a)) + (nwhite * a);
                d->blank = (d->blank * (1.0 - a)) + (nblack * a);
            }
            for (; li < g.l1 && lines_h[li].field == f; li++) { lines_h[li].blank = d->blank; lines_h[li].white = d->white; }
        }
        // the tool's memset before every composite_layer() :1016, the group's frames at once
        R28CHK(d, hipMemset2DAsync((uint8_t *)frames_dev + (size_t)g.f0 * frame_stride, frame_stride, 0,
                                   (size_t)linesize * (size_t)d->height, (size_t)(g.f1 - g.f0), st));
        const int nl = (int)(g.l1 - g.l0);
        if (nl > 0) {
            R28CHK(d, hipMemcpyAsync(d->lines.p + g.l0, lines_h + g.l0, (size_t)nl * sizeof(LineRec), hipMemcpyHostToDevice, st));
            const int *tails = nullptr;
            if (tails_a) {
                // rows relative to the group: row -1 = the tail of the scanline before it (both arrays hold it once the
                // group before has settled); the serial first guess writes both arrays, the round reads a and writes b
                int *ta = tails_a + (g.l0 + 1) * 16, *tb = tails_b + (g.l0 + 1) * 16;
                hipLaunchKernelGGL(k_raw28_tails_scan, dim3(((nl + TAIL_B - 1) / TAIL_B + 63) / 64), dim3(64), 0, st, raw, d->h.p, N,
                                   d->lines.p + g.l0, nl, RC, (const int *)(tails_a + g.l0 * 16), ta, tb);
                hipLaunchKernelGGL(k_raw28_tails, dim3((nl + 127) / 128), dim3(128), 0, st, raw, d->h.p, N, d->lines.p + g.l0,
                                   nl, RC, (const int *)ta, tb, d->gcount.p + g.idx);
                tails = tb;
            }
            hipLaunchKernelGGL(k_raw28_render, dim3((unsigned)nl), dim3(256), render_lds, st, raw, d->h.p, N, d->lines.p + g.l0, RC,
                               tails, (uint8_t *)frames_dev, frame_stride, linesize);
        }
        return NTSCSIM_OK;
    };
    int nf = 0;
    // the fields walked since the last group: their sums start now, the group before them is completed
    auto close_group = [&]() -> int {
        grp.f1 = nf; grp.l1 = nlines; grp.c1 = ncal; grp.p1 = cal_zero.size();
        { const int rc = submit_levels(grp); if (rc != NTSCSIM_OK) return rc; }
        if (pending.idx >= 0) { const int rc = complete_group(pending); if (rc != NTSCSIM_OK) return rc; }
        pending = grp;
        grp = Group{grp.idx + 1, nf, nf, nlines, nlines, ncal, ncal, cal_zero.size(), cal_zero.size()};
        return NTSCSIM_OK;
    };
    while (nf < max_fields) {
        // the tool blocks in read() until its buffer is full; here a field whose window is not complete yet
        // waits for the next push (nothing is changed before that is known)
        if (!d->eof && ((Rd - Bw > CAP / 2u) ? Rd : Bw) + CAP > N) break;
        if (Rd - Bw > CAP / 2u) {                  // lazy_flush_src :329 -> flush_src :290 (twice per field, idempotent)
            bm.assign(0, Ew - Rd, Rd);             // memmove: the live records move to the front, the rest stays
            Bw = Rd;
        }
        const size_t E = std::min(N, Bw + CAP);    // refill_src :307
        if (E > Ew) { bm.assign(Ew - Bw, E - Bw, Ew); Ew = E; }
        if (E - Rd < (size_t)len * 256) break;     // main :1011
        if (!d->o.disable_sync) {                  // :622-693
            size_t i = Rd;
            int vsb = 0;
            while (i < E) {
                size_t si, ei;
                W.next(i, E, si, ei);
                i = ei;
                const size_t synclen = ei - si;
                if (synclen >= L30) { i = si + L30; if (i < ei) i = ei; vsb++; }
                else if (synclen >= L06) { if (vsb >= 9) { Rd = si + synclen / 2; break; } }
                else if (synclen >= L02) {
                    i = si + L30; if (i < ei) i = ei; vsb++;
                    const uint32_t pulse = (uint32_t)cal_zero.size();
                    { const int rc = cal_push(CalRange{(uint32_t)si, (uint32_t)std::min(i, E), pulse, 0}); if (rc != NTSCSIM_OK) return rc; }
                    size_t zeros = 0;
                    int lookup_rc = NTSCSIM_OK;
                    if (i > E) {                   // records past the buffered stream: what the array still holds
                        const size_t k1 = std::min(i - Bw, CAP);          // (past the array itself: undefined in the tool, zero here)
                        zeros = (i - Bw) - k1;
                        if (E - Bw < k1)
                            zeros += bm.lookup(E - Bw, k1, [&](size_t a, size_t b) {
                                if (lookup_rc == NTSCSIM_OK) lookup_rc = cal_push(CalRange{(uint32_t)a, (uint32_t)b, pulse, 0});
                            });
                    }
                    if (lookup_rc != NTSCSIM_OK) return lookup_rc;
                    if (i > E) { d->stats[12]++; d->stats[13] += (int64_t)zeros; }
                    cal_zero.push_back((int)zeros);
                }
            }
        }
        cal_field.push_back((int)cal_zero.size());
        size_t scan = Rd;
        const size_t start = Rd;
        for (unsigned y = 0; y < (unsigned)d->height && (scan + (size_t)len * 2) < E; y++) {      // :700
            LineRec L;
            L.pos = (uint32_t)scan; L.field = nf; L.row = (int)y; L.blank = 0; L.white = 0;
            if (nlines == lines_cap) { d->err = "sync walk: more scanlines than the stream can hold"; return NTSCSIM_E_INTERNAL; }
            lines_h[nlines++] = L;
            scan += len;                            // :777-787 (one_scanline_width is integral: err stays 0)
            if (scan > E) scan = E;
            if (!d->o.disable_sync) {              // :789-830
                size_t i = scan;
                int vsb = 0;
                if (i > Rd) {
                    size_t avail = i - Rd;
                    if ((double)avail >= (len * 0.1)) avail = (size_t)(len * 0.1);
                    i -= avail;
                }
                while (i < E) {
                    size_t si, ei;
                    W.next(i, E, si, ei);
                    i = ei;
                    const size_t synclen = ei - si;
                    if (synclen >= L30) { i = si + L30; if (i < ei) i = ei; vsb++; }
                    else if (synclen >= L06) { scan = si + synclen / 2; break; }
                    else if (synclen >= L02) { i = si + L30; if (i < ei) i = ei; vsb++; }
                    if (vsb >= 9) { y = INT_MAX; break; }
                }
            }
        }
        if (d->o.disable_sync) Rd = scan;          // :833
        {
            size_t should = start + (size_t)len * 240;                             // :836-845
            if (should > E) should = E;
            if (Rd < should) Rd = should;
        }
        nf++;
        if (nf - grp.f0 >= GROUP_FIELDS) {         // a group of fields is complete: its GPU work starts while the walk goes on
            const int rc = close_group();
            if (rc != NTSCSIM_OK) return rc;
        }
    }
    d->read_pos = d->base + Rd;
    d->stats[15] = (int64_t)std::max<size_t>((size_t)d->stats[15], N);      // most samples ever held at once
    d->stats[4] += (int64_t)nlines;
    d->stats[5] += (int64_t)cal_zero.size();
    d->fields_total += (uint64_t)nf;
    *n_fields = nf;

    lap(8);
    // ---- (4) (5) what the walk has left: the last, incomplete group and the one before it
    if (nf > grp.f0) { const int rc = close_group(); if (rc != NTSCSIM_OK) return rc; }
    if (pending.idx >= 0) { const int rc = complete_group(pending); if (rc != NTSCSIM_OK) return rc; }
    const int ngroups = grp.idx;
    lap(9);
    if (nlines > 0 && tails_a) {
        // did every group's round confirm its first guess?  (b holds the rounds' results, a the guesses)
        std::vector<int> nch((size_t)ngroups, 0);
        R28CHK(d, hipMemcpyAsync(nch.data(), d->gcount.p, (size_t)ngroups * sizeof(int), hipMemcpyDeviceToHost, st));
        R28CHK(d, hipMemcpyAsync(d->tail_carry, tails_b + nlines * 16, sizeof(d->tail_carry), hipMemcpyDeviceToHost, st));
        R28CHK(d, hipStreamSynchronize(st));
        d->stats[2]++;
        bool settled = true;
        for (int v : nch) settled = settled && v == 0;
        if (d->force_tail_rounds) settled = false;             // (test hook: the rounds over all scanlines + a second rendering)
        lap(10);
        if (!settled) {
            // round r: tout(y) = G(samples of y, tin(y-1)); it ends when tout == tin everywhere, i.e. tail(y) =
            // G(y, tail(y-1)) for every y with tail(-1) = the carried tail: the serial result.  From the guesses in a.
            const int nl = (int)nlines;
            int *tin = tails_a + 16, *tout = tails_b + 16;
            for (int round = 0;;) {
                for (int b4 = 0; b4 < 4; b4++, round++) {
                    R28CHK(d, hipMemsetAsync(d->counters.p, 0, sizeof(int), st));
                    hipLaunchKernelGGL(k_raw28_tails, dim3((nl + 127) / 128), dim3(128), 0, st, raw, d->h.p, N, d->lines.p,
                                       nl, RC, (const int *)tin, tout, d->counters.p);
                    d->stats[2]++;
                    std::swap(tin, tout);
                }
                int left = 0;
                R28CHK(d, hipMemcpyAsync(&left, d->counters.p, sizeof(int), hipMemcpyDeviceToHost, st));
                R28CHK(d, hipStreamSynchronize(st));
                if (left == 0) break;
                if (round > nl + 8) { d->err = "comb tails did not settle"; return NTSCSIM_E_INTERNAL; }
            }
            R28CHK(d, hipMemcpyAsync(d->tail_carry, tin + (size_t)(nl - 1) * 16, sizeof(d->tail_carry), hipMemcpyDeviceToHost, st));
            hipLaunchKernelGGL(k_raw28_render, dim3((unsigned)nl), dim3(256), render_lds, st, raw, d->h.p, N, d->lines.p, RC,
                               (const int *)tin, (uint8_t *)frames_dev, frame_stride, linesize);
        }
    } else lap(10);
    R28CHK(d, hipGetLastError());
    R28CHK(d, hipStreamSynchronize(st));
    lap(11);

    // ---- (6) a stream that goes on: drop the samples nothing can ask for any more.  Still needed: the
    // window [Bw, ...), whatever stale records of the tool's array refer to (bm), the front end's warm-up
    // before the next new sample, and D samples of delay line in front of all of it.
    if (!d->eof) {
        size_t low = std::min(Bw, Rd);
        for (const BufMap::Seg &g : bm.segs) low = std::min(low, g.abs0);
        low = std::min(low, N > (size_t)warm + 64 ? N - (size_t)warm - 64 : 0);
        const size_t margin = (size_t)d->D + 64;
        size_t keep = low > margin ? (low - margin) & ~(size_t)15 : 0;
        if (keep >= (4u << 20) && keep > N / 4) {          // (worth a copy)
            const size_t rest = N - keep;
            R28CHK(d, d->raw_alt.ensure(std::max(d->raw.cap, rest + 64)));
            R28CHK(d, d->h_alt.ensure(std::max(d->h.cap, rest + 64)));
            R28CHK(d, hipMemcpyAsync(d->raw_alt.p, d->raw.p + keep, rest, hipMemcpyDeviceToDevice, st));
            R28CHK(d, hipMemcpyAsync(d->h_alt.p, d->h.p + keep, rest, hipMemcpyDeviceToDevice, st));
            R28CHK(d, hipStreamSynchronize(st));
            std::swap(d->raw.p, d->raw_alt.p); std::swap(d->raw.cap, d->raw_alt.cap);
            std::swap(d->h.p, d->h_alt.p); std::swap(d->h.cap, d->h_alt.cap);
            d->stats[14]++;
            d->base += keep; d->cnt -= keep; d->front_done -= keep;
            d->last_n = d->cnt;      // (ntscsim_raw28_debug_read_front reads the compacted buffer)
            Bw -= keep; Rd -= keep; Ew -= keep;
            for (BufMap::Seg &g : bm.segs) g.abs0 -= keep;
        }
    }
    return NTSCSIM_OK;
}

// A push that fails after it has appended its samples (a HIP error, NTSCSIM_E_INTERNAL) leaves counters, the front-end
// state and the end-of-stream mark half advanced: retrying it would append the samples twice.  Such a stream is marked
// broken and refuses further pushes until ntscsim_raw28_stream_reset().
static int raw28_stream_push(ntscsim_raw28 *d, const void *samples, bool on_device, size_t n, bool final, void *frames_dev,
                             size_t frame_stride, int linesize, int max_fields, int *n_fields)
{
    if (d) d->mutated = false;
    const int rc = raw28_stream_push_impl(d, samples, on_device, n, final, frames_dev, frame_stride, linesize, max_fields, n_fields);
    if (d && rc != NTSCSIM_OK && d->mutated) d->broken = true;
    return rc;
}

extern "C" int ntscsim_raw28_stream_reset(ntscsim_raw28 *d)
{
    if (!d) return NTSCSIM_E_ARG;
    raw28_stream_reset(d);
    return NTSCSIM_OK;
}
extern "C" int ntscsim_raw28_stream_push(ntscsim_raw28 *d, const void *samples, size_t n, int on_device, int final,
                                         void *frames_dev, size_t frame_stride, int linesize, int max_fields, int *n_fields)
{
    return raw28_stream_push(d, samples, on_device != 0, n, final != 0, frames_dev, frame_stride, linesize, max_fields, n_fields);
}

// a whole capture = a stream of one push
extern "C" int ntscsim_raw28_decode(ntscsim_raw28 *d, const uint8_t *capture_host, size_t n, void *frames_dev,
                                    size_t frame_stride, int linesize, int max_fields, int *n_fields)
{
    if (!d || (n > 0 && !capture_host) || !frames_dev || !n_fields || max_fields < 0) return NTSCSIM_E_ARG;
    if (n >= 0xFFFFFFFFull) return NTSCSIM_E_SIZE;
    raw28_stream_reset(d);
    return raw28_stream_push(d, capture_host, false, n, true, frames_dev, frame_stride, linesize, max_fields, n_fields);
}
extern "C" int ntscsim_raw28_decode_device(ntscsim_raw28 *d, const void *capture_dev, size_t n, void *frames_dev,
                                           size_t frame_stride, int linesize, int max_fields, int *n_fields)
{
    if (!d || (n > 0 && !capture_dev) || !frames_dev || !n_fields || max_fields < 0) return NTSCSIM_E_ARG;
    if (n >= 0xFFFFFFFFull) return NTSCSIM_E_SIZE;
    raw28_stream_reset(d);
    return raw28_stream_push(d, capture_dev, true, n, true, frames_dev, frame_stride, linesize, max_fields, n_fields);
}
