// ntsc422_fused.hip -- composite_video_process() (ffmpeg_to_composite.cpp:629-952) for the VHS
// family of option sets in FOUR sweeps per scanline instead of the twelve of k422_process:
//
//   A   frame row -> input chroma low-pass :353-393 -> modulate :434-477 -> pre-emphasis :636-651
//       -> luma noise :654-666                                   -> composite bytes (scratch Y)
//       (head switching :669-732: waves that hold switched rows gather the displaced bytes into the
//       spare scratch plane and swap the two planes)
//   B1  Y/C separation :480-553 -> chroma noise :738-754 -> phase noise :755-781 (chroma)
//       -> VHS luma low-pass + emphasis :812-831 -> luma sharpen :887-901 (luma, same sweep)
//   B2  VHS chroma low-pass :834-855 -> vertical blend :862-882 -> chroma sharpen :904-924
//       -> re-modulate :926-928 onto the luma of B1               -> composite bytes (scratch Y)
//   B3  Y/C separation :929 -> chroma dropout :932-942 -> output chroma low-pass :948-951
//       -> the frame row
//
// Same execution model (one lane = one scanline, 63 rows + 1 halo row per wave, byte planes kept
// transposed and packed in HBM scratch between sweeps, every stage clamped to uint8 like the
// reference) and bit-identical results; what changes is the traffic: the frame is read and written
// directly (no copy sweeps), chroma planes that the next Y/C separation overwrites anyway are never
// stored, and each scratch plane makes one round trip per sweep -- ~12 byte-visits per luma sample
// instead of ~48.  Filters run in the carry form of ntsc_decode_fast.hip (3 fp64 instructions per pole).
//
// k422_fused<true> is the '-vhs' preset's own switch set (full output low-pass) with the switches as
// compile-time constants (and aligned frame rows); k422_fused<false> reads them at run time.
//
// Preconditions (launcher; otherwise k422_process): VHS emulation with composite output (not
// s-video), colour subcarrier on, input chroma low-pass on, no -nocolor-subcarrier-after-yc-sep,
// no extra -yc-recomb passes.  Everything else (tape speed, PAL, noise levels, pre-emphasis, head
// switching, blend, dropout, output low-pass mode, subcarrier amplitude, phase mode) is handled.
#pragma clang fp contract(off)

namespace ntscsim {
#ifdef F422_AB_TIMES   // developer A/B build: wave-clock time per sweep, summed over waves (tools/sweep_times.py)
__device__ unsigned long long g422_times[8];
extern "C" int ntscsim_debug_422_times(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g422_times), sizeof(g422_times)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g422_times), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define F422_STAMP(k) do { const unsigned long long t__ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g422_times[k], t__ - t_prev__); t_prev__ = t__; } while (0)
#else
#define F422_STAMP(k) do { } while (0)
#endif
namespace fused422 {

using fastdec::Casc3;

typedef __attribute__((address_space(1))) const uint8_t *g_cu8;
typedef __attribute__((address_space(1))) uint8_t *g_u8;
typedef __attribute__((address_space(1))) const uint32_t *g_cu32;
typedef __attribute__((address_space(1))) uint32_t *g_u32;

// sequential byte writer into a frame row: bytes are gathered into NW 32-bit words and leave as ONE
// 4*NW-byte store per lane (rows aligned to 4*NW bytes)
template <int NW>
struct RowWriter {
    typedef uint32_t vec __attribute__((ext_vector_type(NW)));
    typedef __attribute__((address_space(1))) vec *g_vec;
    g_u8 p;
    bool al, on;
    uint32_t acc, w[NW - 1];      // finished words of the current piece, oldest first
    DEV void begin(uint8_t *row, bool aligned, bool enabled)
    {
        p = (g_u8)row; al = aligned; on = enabled; acc = 0;
#pragma unroll
        for (int i = 0; i < NW - 1; i++) w[i] = 0;
    }
    DEV void put(int x, int v)    // x ascending, every index exactly once
    {
        if (!al) { if (on) p[x] = (uint8_t)v; return; }
        const uint32_t sh = 8u * (unsigned)(x & 3);
        acc = (x & 3) ? (acc | ((uint32_t)v << sh)) : (uint32_t)v;
        if ((x & 3) != 3) return;
        if (((x >> 2) & (NW - 1)) != NW - 1) {          // word done, piece not
#pragma unroll
            for (int i = 0; i < NW - 2; i++) w[i] = w[i + 1];
            w[NW - 2] = acc;
            return;
        }
        vec pc;
#pragma unroll
        for (int i = 0; i < NW - 1; i++) pc[i] = w[i];
        pc[NW - 1] = acc;
#ifdef F422_AB_NOSTORE    // timing-only A/B build (WRONG frames): the piece is folded into one word instead
        { w[0] ^= pc[0] ^ pc[NW - 1]; if (x < 0 && on) p[0] = (uint8_t)w[0]; return; }
#endif
        if (on) *(g_vec)(p + (x & ~(4 * NW - 1))) = pc;
    }
    DEV void finish(int n)        // n bytes were put: flush what an incomplete piece holds
    {
        if (!al || !on) return;
        const int w0 = n & ~(4 * NW - 1);                // start of the incomplete piece
        const int nwords = (n - w0) >> 2;
        for (int i = nwords; i < NW - 1; i++) {
#pragma unroll
            for (int k = 0; k < NW - 2; k++) w[k] = w[k + 1];
        }
#pragma unroll
        for (int i = 0; i < NW - 1; i++)
            if (i < nwords) *(g_u32)(p + w0 + 4 * i) = w[i];
        for (int i = 0; i < (n & 3); i++) p[(n & ~3) + i] = (uint8_t)((acc >> (8 * i)) & 0xFFu);
    }
};

// The same sequential byte writer with the finished words staged in LDS (16 words per lane) and sent as
// ONE 64-byte burst per lane: a frame row then reaches HBM in whole 64-byte pieces instead of 16- / 8-byte
// ones that the L2 has often evicted before the rest of their line arrives (measured: 2.4x the bytes
// written).  NB = words per store instruction (4: rows aligned to 16 bytes, 2: to 8); NW = words per burst
// (16: 64 bytes, 8: 32 bytes).  Aligned rows only.
#ifdef NTSC_422_NT_STORES          /* A/B: streaming stores of the finished bursts -- measured WORSE here (profiles/r04_nt_probe.txt:
                                      0.647 -> 0.660 ms per step, WRITE_SIZE + 42 %: the bursts are 64-byte halves of lines whose other half
                                      follows a chunk later, and the rows are rewritten in place) */
#define NTSC_422_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define NTSC_422_STORE(p, v) (*(p) = (v))
#endif
template <int NB, int NW>
struct BurstWriter {
    typedef uint32_t vec __attribute__((ext_vector_type(NB)));
    typedef __attribute__((address_space(1))) vec *g_vec;
    g_u8 p;
    bool on;
    uint32_t acc;
    uint32_t *st;                 // this lane's NW staged words (+ padding) in LDS
    DEV void begin(uint8_t *row, bool, bool enabled) { p = (g_u8)row; on = enabled; acc = 0; }
    DEV void put(int x, int v)    // x ascending, every index exactly once
    {
        const uint32_t sh = 8u * (unsigned)(x & 3);
        acc = (x & 3) ? (acc | ((uint32_t)v << sh)) : (uint32_t)v;
        if ((x & 3) != 3) return;
        const int wq = (x >> 2) & (NW - 1);
        st[wq] = acc;
        if (wq != NW - 1) return;
        if (on) {
            const vec *sp = reinterpret_cast<const vec *>(st);
            g_vec dp = (g_vec)(p + (x & ~(4 * NW - 1)));
            vec t[NW / NB];
#pragma unroll
            for (int i = 0; i < NW / NB; i++) t[i] = sp[i];
#pragma unroll
            for (int i = 0; i < NW / NB; i++) NTSC_422_STORE(dp + i, t[i]);
        }
    }
    DEV void finish(int n)        // n bytes were put: flush the incomplete burst
    {
        if (!on) return;
        const int w0 = n & ~(4 * NW - 1);
        const int nwords = (n - w0) >> 2;
        for (int i = 0; i < nwords; i++) *(g_u32)(p + w0 + 4 * i) = st[i];
        for (int i = 0; i < (n & 3); i++) p[(n & ~3) + i] = (uint8_t)((acc >> (8 * i)) & 0xFFu);
    }
};

// A scratch plane read in blocks of 4*NWB samples: body(x, j, byte) for x = 0 .. N-1 with j = x mod
// the block size a compile-time constant.  Whole blocks run without any bounds test; the ragged
// last block is a second copy of the body.  The next block's words are requested before the
// current block is worked on.
// NEED (the sweep as a ROLE of k422_direct_pipe): need(bytes) returns once the plane's first `bytes` samples may be read;
// such a reader loads with the streaming policy (always from the L2).
struct NoNeed422 { static constexpr bool nt = false; DEV void operator()(int) const {} };
struct Need422 {
    static constexpr bool nt = true;
    pipe::lds_flag f;
    int W;
    int *seen;
    DEV void operator()(int bytes) const { pipe::wait_ge(f, bytes < W ? bytes : W, *seen); }
};
template <int NWB, class F, class NEED = NoNeed422>
DEV void sweep_blocks(const Plane422 &pl, int N, F body, NEED need = NEED())
{
    constexpr int B = 4 * NWB;
    const int nwords = (N + 3) >> 2;
    uint32_t c[NWB], n[NWB];
    auto ldw = [&](int q) -> uint32_t {
        if constexpr (NEED::nt) return __builtin_nontemporal_load((__attribute__((address_space(1))) const uint32_t *)(pl.p + (size_t)q * pl.S));
        else return pl.word(q);
    };
    need(4 * NWB);
#pragma unroll
    for (int i = 0; i < NWB; i++) c[i] = i < nwords ? ldw(i) : 0u;
    int x0 = 0;
    for (; x0 + B <= N; x0 += B) {
        const int q = (x0 >> 2) + NWB;
        need(4 * (q + NWB));
#pragma unroll
        for (int i = 0; i < NWB; i++) n[i] = q + i < nwords ? ldw(q + i) : 0u;
#pragma unroll
        for (int j = 0; j < B; j++) body(x0 + j, j, (int)((c[j >> 2] >> (8 * (j & 3))) & 0xFFu));
#pragma unroll
        for (int i = 0; i < NWB; i++) c[i] = n[i];
    }
    if (x0 < N) {
#pragma unroll
        for (int j = 0; j < B; j++)
            if (x0 + j < N) body(x0 + j, j, (int)((c[j >> 2] >> (8 * (j & 3))) & 0xFFu));
    }
}

// 16 luma / 8 chroma bytes of a frame row as words, zero past the row; one vector load when the
// row is aligned and the block lies inside it
DEV void load_block16(const uint8_t *row, int x0, int n, bool al16, uint32_t (&w)[4])
{
#ifdef F422_AB_NOLOAD     // timing-only A/B build (WRONG frames): no frame loads
    w[0] = w[1] = w[2] = w[3] = 0x40404040u + (uint32_t)x0; return;
#endif
    if (al16 && x0 + 16 <= n) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 v = *(__attribute__((address_space(1))) const v4 *)(row + x0);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {                                       // row ends and unaligned rows: byte by byte (kept as a loop: small code)
        uint32_t a[4] = {0, 0, 0, 0};
#pragma unroll 1
        for (int b = 0; b < 16; b++) {
            const int x = x0 + b;
            const uint32_t v = x < n ? (uint32_t)((g_cu8)row)[x] : 0u;
            const uint32_t sh = v << (8 * (b & 3));
            a[0] |= (b >> 2) == 0 ? sh : 0u; a[1] |= (b >> 2) == 1 ? sh : 0u;
            a[2] |= (b >> 2) == 2 ? sh : 0u; a[3] |= (b >> 2) == 3 ? sh : 0u;
        }
        w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3];
    }
}
DEV void load_block8(const uint8_t *row, int x0, int n, bool al8, uint32_t (&w)[2])
{
#ifdef F422_AB_NOLOAD
    w[0] = w[1] = 0x80808080u + (uint32_t)x0; return;
#endif
    if (al8 && x0 + 8 <= n) {
        typedef uint32_t v2 __attribute__((ext_vector_type(2)));
        const v2 v = *(__attribute__((address_space(1))) const v2 *)(row + x0);
        w[0] = v.x; w[1] = v.y;
    } else {
        uint32_t a0 = 0, a1 = 0;
#pragma unroll 1
        for (int b = 0; b < 8; b++) {
            const int x = x0 + b;
            const uint32_t v = x < n ? (uint32_t)((g_cu8)row)[x] : 0u;
            const uint32_t sh = v << (8 * (b & 3));
            a0 |= b < 4 ? sh : 0u; a1 |= b < 4 ? 0u : sh;
        }
        w[0] = a0; w[1] = a1;
    }
}
// The same for frame rows whose start and linesize are multiples of 16 / 8 bytes (the streamed forms): ONE
// unconditional vector load -- a block that starts inside the linesize lies inside it entirely (the bytes
// behind the row's width are padding nobody looks at), a block past it re-reads the row's last one (nobody
// looks at that either).  No branch, no byte loop: the loads of a whole group stay in flight across the
// blocks that are being worked on (with the guarded form hipcc drains vmcnt before the block loop).
DEV void load_block16_al(const uint8_t *row, int x0, int ls, uint32_t (&w)[4])
{
#ifdef F422_AB_NOLOAD
    w[0] = w[1] = w[2] = w[3] = 0x40404040u + (uint32_t)x0; return;
#endif
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const int xa = x0 + 16 <= ls ? x0 : ls - 16;
    const v4 v = *(__attribute__((address_space(1))) const v4 *)(row + xa);
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
DEV void load_block8_al(const uint8_t *row, int x0, int ls, uint32_t (&w)[2])
{
#ifdef F422_AB_NOLOAD
    w[0] = w[1] = 0x80808080u + (uint32_t)x0; return;
#endif
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    const int xa = x0 + 8 <= ls ? x0 : ls - 8;
    const v2 v = *(__attribute__((address_space(1))) const v2 *)(row + xa);
    w[0] = v.x; w[1] = v.y;
}
// A group's 64 + 32 + 32 bytes of 64 rows, loaded COOPERATIVELY (like fastenc::CoopLoader): four consecutive lanes
// fetch the four 16- / 8-byte blocks of ONE row, 16 rows per load instruction -- contiguous 64- / 32-byte pieces
// instead of 64 scattered 16- / 8-byte ones -- and the blocks go through an LDS tile from which every lane reads its
// own row back.  The tile is the frame-row stage of the streamed pass, which is idle during sweep A.
// Tile (words): luma [64][20] | U [64][12] | V [64][12].
struct CoopRows {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    const uint8_t *py[4], *pu[4], *pv[4];     // this lane's block of row group i (16 rows each), at x = 0
    uint32_t *tile;
    int lsy, lsu, lsv, blk, wy, wc, ry, rc;
    DEV static const uint8_t *row_of(const uint8_t *mine, int from)
    {
        const unsigned lo = (unsigned)(uintptr_t)mine, hi = (unsigned)((uintptr_t)mine >> 32);
        const unsigned l2 = (unsigned)__shfl((int)lo, from), h2 = (unsigned)__shfl((int)hi, from);
        return (const uint8_t *)(((uintptr_t)h2 << 32) | l2);
    }
    DEV void begin(const uint8_t *fy, const uint8_t *fu, const uint8_t *fv, int ly, int lu, int lv, uint32_t *t, int lane)
    {
        tile = t; lsy = ly; lsu = lu; lsv = lv; blk = lane & 3;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int from = 16 * i + (lane >> 2);
            py[i] = row_of(fy, from); pu[i] = row_of(fu, from); pv[i] = row_of(fv, from);
        }
        wy = (lane >> 2) * 20 + blk * 4; wc = (lane >> 2) * 12 + blk * 2;
        ry = lane * 20; rc = lane * 12;
    }
    // luma bytes from x0 (64 per row), chroma bytes from c0 (32 per row); a block past the linesize re-reads the
    // row's last one, like load_block16_al / load_block8_al
    DEV void request(int x0, int c0, v4 (&qy)[4], v2 (&qu)[4], v2 (&qv)[4]) const
    {
        const int xb = x0 + 16 * blk, cb = c0 + 8 * blk;
        const int xa = xb + 16 <= lsy ? xb : lsy - 16;
        const int ua = cb + 8 <= lsu ? cb : lsu - 8, va = cb + 8 <= lsv ? cb : lsv - 8;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            qy[i] = *(__attribute__((address_space(1))) const v4 *)(py[i] + xa);
            qu[i] = *(__attribute__((address_space(1))) const v2 *)(pu[i] + ua);
            qv[i] = *(__attribute__((address_space(1))) const v2 *)(pv[i] + va);
        }
    }
    DEV void deliver(const v4 (&qy)[4], const v2 (&qu)[4], const v2 (&qv)[4], uint32_t (&gy)[4][4], uint32_t (&gu)[4][2],
                     uint32_t (&gv)[4][2]) const
    {
        uint32_t *tu = tile + 64 * 20, *tv = tile + 64 * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            *reinterpret_cast<uint4 *>(&tile[16 * 20 * i + wy]) = make_uint4(qy[i].x, qy[i].y, qy[i].z, qy[i].w);
            *reinterpret_cast<uint2 *>(&tu[16 * 12 * i + wc]) = make_uint2(qu[i].x, qu[i].y);
            *reinterpret_cast<uint2 *>(&tv[16 * 12 * i + wc]) = make_uint2(qv[i].x, qv[i].y);
        }
        // (wave-private tile, lock-step wave: the reads below follow the writes above in LDS order)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&tile[ry + 4 * b]);
            gy[b][0] = v.x; gy[b][1] = v.y; gy[b][2] = v.z; gy[b][3] = v.w;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 a = *reinterpret_cast<const uint4 *>(&tu[rc + 4 * h]);
            const uint4 c = *reinterpret_cast<const uint4 *>(&tv[rc + 4 * h]);
            gu[2 * h][0] = a.x; gu[2 * h][1] = a.y; gu[2 * h + 1][0] = a.z; gu[2 * h + 1][1] = a.w;
            gv[2 * h][0] = c.x; gv[2 * h][1] = c.y; gv[2 * h + 1][0] = c.z; gv[2 * h + 1][1] = c.w;
        }
    }
};
DEV int byte_of(uint32_t w, int b) { return (int)((w >> (8 * b)) & 0xFFu); }

// composite_video_chroma_lowpass :353-393 as a stream: push(raw) returns the filtered value that
// lands `delay` samples back
struct ChromaLpFull {
    OnePole hp;
    Casc3<double> lp;
    double a_lp, a_hp;
    DEV void begin(double alp, double ahp) { a_lp = alp; a_hp = ahp; hp.p = 128; lp.reset(128, alp); }
    DEV int push(int raw)
    {
        double s = raw;
        s += hp.hp(s, a_hp);
        return clampu8((int)lp.push(s, a_lp));
    }
    // the three poles alone: composite_video_chroma_lowpass_lite :395-431
    DEV int push_plain(int raw) { return clampu8((int)lp.push((double)raw, a_lp)); }
};

// ---------------------------------------------------------------------------------- sweep A
// Stream index m = the chroma sample being modulated; the low-passes run D = 4 (NTSC: the V delay)
// or 2 (PAL) samples ahead of it.
// FAST (the preset kernel's streamed form; launcher preconditions: NTSC, no pre-emphasis, luma noise on,
// even scanline phase, subcarrier amplitude 50, aligned rows): blocks of 8 chroma samples that lie strictly
// inside the row run a body without bounds tests or row-tail selects, with (chroma * 50) / 50 == chroma
// (:470) and the modulation sign (:463-466: bit 1 of xi + x) as a per-lane mask chosen by the unrolled
// position -- the same results as the general body, which still runs the first and the last block(s).
// ALROWS (the streamed forms): rows aligned to 16 / 8 bytes, linesizes lsy / lsc known: branch-free loads.
struct NoPub422 { DEV void operator()(int) const {} };
// PUB: called at the end of every group of 32 chroma inputs with the group's first index g0 (k422_pipe: the sweep as a
// ROLE that publishes how far the composite bytes have got)
template <bool NTSC, bool ALIGNED, bool FAST = false, bool ALROWS = false, class PUB = NoPub422>
DEV void sweep_a(const DevParams &P, const Row422 &R, const uint8_t *fy, const uint8_t *fu, const uint8_t *fv,
                 int W, unsigned xi, LumaPost422 &post, double a_hp_i, double a_hp_q, int lsy = 0, int lsu = 0, int lsv = 0,
                 uint32_t *tile = nullptr, PUB pub = PUB())
{
    constexpr int D = NTSC ? 4 : 2, DU = 2, DV = NTSC ? 4 : 2;
    const int W2 = W / 2;
    const bool al16 = ALIGNED || P.src_al16 != 0, al8 = ALIGNED || P.dst_al16 != 0;
    ChromaLpFull lU, lV;
    lU.begin(P.a_in_i, a_hp_i);
    lV.begin(NTSC ? P.a_in_q : P.a_in_i, NTSC ? a_hp_q : a_hp_i);
    Packer422 oy; oy.begin(R.Y);
    // history (newest last): raw U, V of the last D+1 inputs, filtered U of the last D-DU+1 pushes
    int rawU[D + 1], rawV[D + 1], filU[D - DU + 1];
#pragma unroll
    for (int i = 0; i <= D; i++) { rawU[i] = 128; rawV[i] = 128; }
#pragma unroll
    for (int i = 0; i <= D - DU; i++) filU[i] = 128;
    int filV = 128;
    const bool hi = (xi & 2u) != 0;
    const int sm0 = fastdec::opaque_v(hi ? -1 : 0), sm1 = fastdec::opaque_v(hi ? 0 : -1);
    // one block of 8 chroma samples strictly inside the row (FAST): D <= c < W2, 0 <= m < W2 - D
    auto fast_block = [&](int c0, const uint32_t (&ly)[4], const uint32_t (&cu)[2], const uint32_t (&cv)[2],
                          const uint32_t (&ly_prev)[4]) {
            const int q0 = (c0 - D) >> 1;                   // scratch word of luma 2 (c0 - D) (c0 is a multiple of 8)
#pragma unroll
            for (int j = 0; j < 8; j++) {
#pragma unroll
                for (int i = 0; i < D; i++) { rawU[i] = rawU[i + 1]; rawV[i] = rawV[i + 1]; }
#pragma unroll
                for (int i = 0; i < D - DU; i++) filU[i] = filU[i + 1];
                const int u = byte_of(cu[j >> 2], j & 3), v = byte_of(cv[j >> 2], j & 3);
                rawU[D] = u; rawV[D] = v;
                filU[D - DU] = lU.push(u);
                filV = lV.push(v);
                const int U1 = filU[0], V1 = filV;
                const int sm = (j & 1) ? sm1 : sm0;           // x & 2 = 2 ((c0 + j - D) & 1) = 2 (j & 1)
#pragma unroll
                for (int sx = 0; sx < 2; sx++) {
                    const int lb = 2 * (j - D) + sx;
                    const int yin = lb >= 0 ? byte_of(ly[lb >> 2], lb & 3) : byte_of(ly_prev[(lb + 16) >> 2], (lb + 16) & 3);
                    const int cval = (sx ? V1 : U1) - 128;
                    int yv = clampu8(yin + ((cval ^ sm) - sm));
                    yv = clampu8(yv + post.noise);
                    post.noise = sdiv2(post.noise + (int)umod31(post.rng.next(post.ring, post.lane), P.m_noise) - P.noise_k);
                    // byte (2 j + sx) & 3 of scratch word q0 + ((2 j + sx) >> 2)
                    const int kb = (2 * j + sx) & 3;
                    oy.acc = kb ? (oy.acc | ((uint32_t)yv << (8 * kb))) : (uint32_t)yv;
                    if (kb == 3) oy.pl.set_word(q0 + ((2 * j + sx) >> 2), oy.acc);
                }
            }
    };
    // Groups of 32 chroma inputs = 64 luma bytes = 4 blocks of 8 chroma inputs.  A group's 64 + 32 + 32
    // frame bytes are requested together one group ahead (the pieces of one cache line back to back,
    // so a line is fetched twice / four times per row instead of 8 / 16 times) and rotate through
    // the "current block" registers.  Block c0 .. c0+7 modulates the 16 luma bytes from 2*(c0 - D):
    // the upper part of the previous luma block and the lower part of the current one.
    uint32_t gy[4][4], gu[4][2], gv[4][2], ny[4][4], nu[4][2], nv[4][2], ly_prev[4] = {0, 0, 0, 0};
    CoopRows CR;
    CoopRows::v4 qy[4];
    CoopRows::v2 qu[4], qv[4];
    if (ALROWS) {
        CR.begin(fy, fu, fv, lsy, lsu, lsv, tile, post.lane);
        CR.request(0, 0, qy, qu, qv);
        CR.deliver(qy, qu, qv, gy, gu, gv);
    }
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (ALROWS) {
        } else {
            load_block16(fy, 16 * b, W, al16, gy[b]);
            load_block8(fu, 8 * b, W2, al8, gu[b]); load_block8(fv, 8 * b, W2, al8, gv[b]);
        }
    }
    for (int g0 = 0; g0 < W2 + D; g0 += 32) {
      if (ALROWS) CR.request(2 * g0 + 64, g0 + 32, qy, qu, qv);
#pragma unroll
      for (int b = 0; b < 4; b++) {
          if (ALROWS) {
          } else {
              load_block16(fy, 2 * g0 + 64 + 16 * b, W, al16, ny[b]);
              load_block8(fu, g0 + 32 + 8 * b, W2, al8, nu[b]); load_block8(fv, g0 + 32 + 8 * b, W2, al8, nv[b]);
          }
      }
      if (FAST && g0 >= 32 && g0 + 32 <= W2) {
          // a group whose four blocks are all inside the row: straight-line code, no block loop -- the loads of the
          // next group (above) stay in flight while these 32 samples are worked on (hipcc drains vmcnt in front of a
          // loop that contains stores)
          fast_block(g0, gy[0], gu[0], gv[0], ly_prev);
          fast_block(g0 + 8, gy[1], gu[1], gv[1], gy[0]);
          fast_block(g0 + 16, gy[2], gu[2], gv[2], gy[1]);
          fast_block(g0 + 24, gy[3], gu[3], gv[3], gy[2]);
#pragma unroll
          for (int q = 0; q < 4; q++) ly_prev[q] = gy[3][q];
      } else
#pragma unroll 1
      for (int c0 = g0; c0 < g0 + 32 && c0 < W2 + D; c0 += 8) {
        uint32_t (&ly)[4] = gy[0], (&cu)[2] = gu[0], (&cv)[2] = gv[0];
        if (FAST && c0 >= 8 && c0 + 8 <= W2) {
            fast_block(c0, ly, cu, cv, ly_prev);
        } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = c0 + j;
            // ---- push input sample c (if any): filtered U lands at c - DU, filtered V at c - D
#pragma unroll
            for (int i = 0; i < D; i++) { rawU[i] = rawU[i + 1]; rawV[i] = rawV[i + 1]; }
#pragma unroll
            for (int i = 0; i < D - DU; i++) filU[i] = filU[i + 1];
            if (c < W2) {
                const int u = byte_of(cu[j >> 2], j & 3), v = byte_of(cv[j >> 2], j & 3);
                rawU[D] = u; rawV[D] = v;
                filU[D - DU] = lU.push(u);
                filV = lV.push(v);
            }
            // ---- modulate chroma sample m = c - D onto luma 2m, 2m+1 (:434-477)
            const int m = c - D;
            if (m < 0 || m >= W2) continue;
            // the last `delay` samples of a row keep their input (:386)
            const int U1 = m < W2 - DU ? filU[0] : rawU[0];
            const int V1 = m < W2 - DV ? filV : rawV[0];
#pragma unroll
            for (int sx = 0; sx < 2; sx++) {
                const int x = 2 * m + sx;
                // byte 2*(j - D) + sx of the current luma block, or of the previous one if negative
                const int lb = 2 * (j - D) + sx;
                const int yin = lb >= 0 ? byte_of(ly[lb >> 2], lb & 3) : byte_of(ly_prev[(lb + 16) >> 2], (lb + 16) & 3);
                const unsigned s = (xi + (unsigned)x) & 3u;
                int chroma = ((s & 1u) ? V1 - 128 : U1 - 128) * P.amp;
                if (s & 2u) chroma = -chroma;
                int yv = clampu8(yin + chroma / 50);
                if (post.pre_on) {
                    double sd = yv;
                    sd += post.pre.hp(sd, P.a_pre) * P.pre_gain;
                    yv = clampu8((int)sd);
                }
                if (post.noise_on) {
                    yv = clampu8(yv + post.noise);
                    post.noise = sdiv2(post.noise + (int)umod31(post.rng.next(post.ring, post.lane), P.m_noise) - P.noise_k);
                }
                oy.put(x, yv);
            }
        }
        }
        // next block of the group becomes the current one
#pragma unroll
        for (int q = 0; q < 4; q++) ly_prev[q] = gy[0][q];
#pragma unroll
        for (int b = 0; b < 3; b++) {
#pragma unroll
            for (int q = 0; q < 4; q++) gy[b][q] = gy[b + 1][q];
            gu[b][0] = gu[b + 1][0]; gu[b][1] = gu[b + 1][1];
            gv[b][0] = gv[b + 1][0]; gv[b][1] = gv[b + 1][1];
        }
      }
      if (ALROWS) CR.deliver(qy, qu, qv, gy, gu, gv);
      else {
#pragma unroll
        for (int b = 0; b < 4; b++) {
#pragma unroll
          for (int q = 0; q < 4; q++) gy[b][q] = ny[b][q];
          gu[b][0] = nu[b][0]; gu[b][1] = nu[b][1]; gv[b][0] = nv[b][0]; gv[b][1] = nv[b][1];
        }
      }
      pub(g0);
    }
    oy.finish(W);
}

// ---------------------------------------------------------------------------------- luma chain (B1)
// VHS luma low-pass + emphasis :812-831, then sharpen :887-901, each clamped like its own sweep
struct LumaVhs {
    Casc3<double> vl, sh;
    fastdec::PoleHp<double> pre;
    double a_vl, a_sh, sharpen;
    DEV void begin(double avl, double ash, double sharp)
    {
        a_vl = avl; a_sh = ash; sharpen = sharp;
        vl.reset(16, avl); pre.reset(16, avl); sh.reset(16, ash);
    }
    DEV int run(int yb)
    {
        double m2;
        double s = vl.push((double)yb, a_vl, m2);
        s += pre.hp(s, m2, a_vl) * 1.6;
        const double y1 = clampu8((int)s);
        const double ts = sh.push(y1, a_sh);
        return clampu8((int)(y1 + ((y1 - ts) * sharpen)));
    }
};

// ---------------------------------------------------------------------------------- frame sink (B3)
// chroma dropout :932-942 -> output chroma low-pass :948-951 -> the frame row
template <class WY, class WC>
struct FrameSinkT {
    WY wy;
    WC wu, wv;
    bool drop;
    int mode, dU, dV, W2;             // mode: 0 none, 1 lite (:395-431), 2 full (:353-393)
    ChromaLpFull fU, fV;              // full: high-pass boost + three poles; lite: its three poles alone, at a_tv
    int u1, u2, u3, u4, v1, v2, v3, v4;   // the last 4 inputs, 1 = newest (row tails); named, not an
                                          // array: a dynamically indexed array would live in scratch
    DEV void begin(const DevParams &P, bool aligned, int out_lp, uint8_t *fy, uint8_t *fu, uint8_t *fv, bool is_out, bool dropped,
                   double a_hp_i, double a_hp_q, int W)
    {
        wy.begin(fy, aligned || P.src_al16 != 0, is_out); wu.begin(fu, aligned || P.dst_al16 != 0, is_out);
        wv.begin(fv, aligned || P.dst_al16 != 0, is_out);
        drop = dropped; mode = out_lp; W2 = W / 2;
        dU = mode == 2 ? 2 : (mode == 1 ? 1 : 0);
        dV = mode == 2 ? (P.ntsc ? 4 : 2) : (mode == 1 ? 1 : 0);
        if (mode == 1) { fU.begin(P.a_tv, 0.0); fV.begin(P.a_tv, 0.0); }      // (both filters start from 128)
        else {
            fU.begin(P.a_in_i, a_hp_i);
            fV.begin(P.ntsc ? P.a_in_q : P.a_in_i, P.ntsc ? a_hp_q : a_hp_i);
        }
        u1 = u2 = u3 = u4 = v1 = v2 = v3 = v4 = 128;
    }
    DEV void luma(int x, int y) { wy.put(x, y); }
    DEV void chroma(int c, int u, int v)
    {
        if (drop) { u = 128; v = 128; }
        chroma_nodrop(c, u, v);
    }
    DEV void chroma_nodrop(int c, int u, int v)      // (the caller has applied the dropout)
    {
        u4 = u3; u3 = u2; u2 = u1; u1 = u;
        v4 = v3; v3 = v2; v2 = v1; v1 = v;
        chroma_inner(c, u, v);
    }
    // a sample at least 4 before the row end: the row-tail history (u1 .. v4) is rebuilt from the samples
    // behind it, all of which come through chroma() / chroma_nodrop()
    DEV void chroma_inner(int c, int u, int v)
    {
        if (mode == 0) { wu.put(c, u); wv.put(c, v); return; }
        int ou, ov;
        if (mode == 2) { ou = fU.push(u); ov = fV.push(v); }
        else { ou = fU.push_plain(u); ov = fV.push_plain(v); }
        if (c >= dU) wu.put(c - dU, ou);
        if (c >= dV) wv.put(c - dV, ov);
    }
    DEV void finish(int W)
    {
        // the last `delay` (1, 2 or 4) samples keep their input
        if (dU == 4) { wu.put(W2 - 4, u4); wu.put(W2 - 3, u3); }
        if (dU >= 2) wu.put(W2 - 2, u2);
        if (dU >= 1) wu.put(W2 - 1, u1);
        if (dV == 4) { wv.put(W2 - 4, v4); wv.put(W2 - 3, v3); }
        if (dV >= 2) wv.put(W2 - 2, v2);
        if (dV >= 1) wv.put(W2 - 1, v1);
        wy.finish(W); wu.finish(W2); wv.finish(W2);
    }
};
typedef FrameSinkT<RowWriter<4>, RowWriter<2>> FrameSink;
typedef FrameSinkT<BurstWriter<4, 16>, BurstWriter<2, 16>> FrameSinkBurst;

// composite_ntsc_to_yuv :480-553 in one sweep over scratch plane R.Y (see demodulate422).  LUMA:
// the separated luma goes through the VHS luma chain and back to R.Y, chroma to R.U / R.V.  SINK:
// everything goes to the frame through `sink`.  POST: the chroma pair passes the chroma / phase noise on its way
// (the first separation of the path; SINK + POST = the whole decode side of a switch set WITHOUT the VCR, k422_direct).
template <bool SINK, bool POST = !SINK, class SINKT = FrameSink, class NEED = NoNeed422>
DEV void demod(const DevParams &P, const Row422 &R, int W, unsigned xi, const Magic31 &mA, int oob0, int oob1,
               ChromaPost422 &cpost_in, LumaVhs &lv_in, SINKT &sink_in, NEED need = NEED())
{
    // private copies: the states live in registers for the whole sweep
    ChromaPost422 cpost = cpost_in;
    LumaVhs lv = lv_in;
    SINKT sink = sink_in;
    const int W2 = W / 2;
    unsigned d0 = 16, d1 = 16, d2 = 0, d3 = 0, sum = 0;
    int ch_even = 0;
    Packer422 oy, ou, ov;
    if (!SINK) { oy.begin(R.Y); ou.begin(R.U); ov.begin(R.V); }
    sweep_blocks<4>(R.Y, W + 2, [&](int x, int, int in) {
        const int c_in = x < W ? in : (x == W ? oob0 : oob1);   // Y[r]; r >= W: the caller's bytes (:496)
        if (x == 0) { d2 = (unsigned)c_in; sum = 32 + d2; }
        else if (x == 1) { d3 = (unsigned)c_in; sum += d3; }
        else {
            const int xo = x - 2;
            const unsigned c = (unsigned)c_in;
            sum -= d0;
            d0 = d1; d1 = d2; d2 = d3; d3 = c;
            sum += c;
            const unsigned yb = (sum / 4u) & 0xFFu;
            int ch = clampu8((int)c + 128 - (int)yb);
            if (SINK) sink.luma(xo, (int)yb);
            else oy.put(xo, lv.run((int)yb));
            const unsigned g = (unsigned)(xo - 2 + (int)xi) & 3u;
            if ((g == 0u && xo >= 2) || (g == 1u && xo >= 3)) ch = 255 - ch;
            ch = clampu8(sdivm((ch - 128) * 50, mA) + 128);
            if (!(xo & 1)) ch_even = ch;
            else {
                const int a = ch_even, b = ch;
                int u = (xi & 1u) ? 255 - b : 255 - a, v = (xi & 1u) ? 255 - a : 255 - b;
                if (POST) chroma_post422(P, cpost, u, v);      // chroma noise :738-754, phase noise :755-781
                if (SINK) sink.chroma(xo >> 1, u, v);
                else { ou.put(xo >> 1, u); ov.put(xo >> 1, v); }
            }
        }
    }, need);
    if (!SINK) { oy.finish(W); ou.finish(W2); ov.finish(W2); }
    else sink.finish(W);
}

// ---------------------------------------------------------------------------------- sweep B2
// VHS chroma low-pass :834-855 (output lands d samples back, the last d keep their input) ->
// vertical blend :862-882 -> chroma sharpen :904-924 -> modulate onto the luma in R.Y :926-928.
// DS: the chroma delay as a constant (4 = SP tape speed), 0 = read it from P
template <int DS>
DEV void sweep_b2(const DevParams &P, const Row422 &R, int W, unsigned xi, int k, double a_sh_c, double sharpen_c)
{
    const int W2 = W / 2;
    const int d = DS ? DS : P.cdelay;
    const bool blend = P.vblend && P.ntsc;
    Casc3<double> lU, lV, sU, sV;
    lU.reset(128, P.a_vc); lV.reset(128, P.a_vc); sU.reset(128, a_sh_c); sV.reset(128, a_sh_c);
    int wu[7] = {0, 0, 0, 0, 0, 0, 0}, wv[7] = {0, 0, 0, 0, 0, 0, 0};   // last 7 inputs, [6] newest
    Reader422 rv; rv.begin(R.V, W2);
    // the luma plane is read and rewritten in place, 2 samples per chroma output
    Reader422 ry; ry.begin(R.Y, W);
    Packer422 oy; oy.begin(R.Y);
    SWEEP_BEGIN(R.U, W2 + d)
        if (j_ == 0) rv.prefetch(x0_);
        const int inV = rv.get(j_);
        int fU = 0, fV = 0;
        if (x < W2) {
            fU = clampu8((int)lU.push((double)in, P.a_vc));
            fV = clampu8((int)lV.push((double)inV, P.a_vc));
        }
#pragma unroll
        for (int q = 0; q < 6; q++) { wu[q] = wu[q + 1]; wv[q] = wv[q + 1]; }
        wu[6] = in; wv[6] = inV;
        const int xo = x - d;
        if (xo >= 0) {
            const int rawU = d == 4 ? wu[2] : (d == 5 ? wu[1] : wu[0]);
            const int rawV = d == 4 ? wv[2] : (d == 5 ? wv[1] : wv[0]);
            int u = xo < W2 - d ? fU : rawU, v = xo < W2 - d ? fV : rawV;
            const int upU = __shfl_up(u, 1), upV = __shfl_up(v, 1);
            if (blend && k >= 1) {
                u = ((k >= 2 ? upU : 128) + u + 1) >> 1;
                v = ((k >= 2 ? upV : 128) + v + 1) >> 1;
            }
            double s = u;
            double ts = sU.push(s, a_sh_c);
            u = clampu8((int)(s + ((s - ts) * sharpen_c)));
            s = v;
            ts = sV.push(s, a_sh_c);
            v = clampu8((int)(s + ((s - ts) * sharpen_c)));
            // composite_video_yuv_to_ntsc :434-477 for luma 2*xo, 2*xo+1
            const int lj = (2 * xo) & 7;                 // position inside the luma reader's block
            if (lj == 0) { if (xo > 0) ry.advance(); ry.prefetch(2 * xo); }
#pragma unroll
            for (int sx = 0; sx < 2; sx++) {
                const int lx = 2 * xo + sx;
                const unsigned ph = (xi + (unsigned)lx) & 3u;
                int chroma = ((ph & 1u) ? v - 128 : u - 128) * P.amp;
                if (ph & 2u) chroma = -chroma;
                oy.put(lx, clampu8(ry.get(lj + sx) + chroma / 50));
            }
        }
        if (j_ == BK - 1 || x == W2 + d - 1) rv.advance();
    SWEEP_END
    oy.finish(W);
}


// ---------------------------------------------------------------------------------- streamed B pass
// B1 + B2 + B3 of the '-vhs' preset's switch set as ONE pass over the composite bytes of sweep A, every
// stage's state in registers (the BGRA decoder's model, ntsc_decode_fast.hip): nothing but the composite
// plane goes through HBM scratch between the frame row that comes in and the frame row that goes out.
// One iteration i = one chroma sample period = two luma samples:
//   main stream   composite bytes 2i, 2i+1 -> Y/C separation :480-553 at positions 2i-2, 2i-1: box luma
//                 for the luma chain, chroma pair m1 = i-1 -> chroma noise :738-754, phase noise :755-781
//   VCR chroma    U1/V1[i-1] -> VHS chroma low-pass :834-855, whose output lands d = 4 samples back, at
//                 m2 = i-5 (the last d outputs of a row keep their raw input) -> vertical blend :862-882
//                 (row above = lane-1, same iteration) -> chroma sharpen :904-924
//   VCR luma      box luma -> VHS low-pass + emphasis :812-831 -> sharpen :887-901, held in an 8-deep
//                 register delay line until the chroma of its position arrives (2d luma samples later)
//   re-modulate   :926-928 luma 2m2, 2m2+1 + chroma m2 -> composite bytes, straight into
//   separation 2  :929 at positions 2m2-2, 2m2-1 -> luma to the frame; chroma pair m3 = m2-1 -> dropout
//                 :932-942 -> full output chroma low-pass :948-951 (U lands 2 back, V 4) -> the frame
// The reference's read of two bytes past the row (:496) feeds both separations (oob0 / oob1).
// iter<true> is the guarded form for any i (row start, row end, drain); iter<false> assumes every stage
// strictly inside the row (D + 2 <= i <= W/2 - 1) and is what the steady loop unrolls four times -- the
// delay lines are shift registers in the source and plain register renaming in the unrolled loop.
template <int D>                                 // chroma delay of the tape speed :793-808: 4 (SP), 5 (LP), 6 (EP)
struct StreamB {
    int W, W2, oob0, oob1, k;
    unsigned xi;
    bool blend;
    double a_vc, a_sh_c, sharpen_c;
    unsigned a0, a1, a2, a3, asum;               // separation 1: Y[x-1 .. x+2] window :486-500
    int ev1;
    ChromaPost422 cp;
    LumaVhs lv;
    int yq[2 * D];                               // VCR luma of the last 2 D positions, [2 D - 1] newest
    Casc3<double> lU, lV, sU, sV;
    int ru[D + 1], rv[D + 1];                    // chroma into the VHS low-pass, last D + 1 samples, [D] newest
    unsigned b0, b1, b2, b3, bsum;               // separation 2
    int ev3;
    FrameSinkBurst sink;

    DEV void begin(const DevParams &P, int W_, unsigned xi_, int k_, int o0, int o1, double ashc, double shc)
    {
        W = W_; W2 = W_ / 2; xi = xi_; k = k_; oob0 = o0; oob1 = o1;
        blend = P.vblend && P.ntsc;
        a_vc = P.a_vc; a_sh_c = ashc; sharpen_c = shc;
        a0 = a1 = 16; a2 = a3 = 0; asum = 0; ev1 = 0;
        b0 = b1 = 16; b2 = b3 = 0; bsum = 0; ev3 = 0;
#pragma unroll
        for (int q = 0; q < 2 * D; q++) yq[q] = 0;
#pragma unroll
        for (int q = 0; q <= D; q++) { ru[q] = 0; rv[q] = 0; }
        lv.begin(P.a_vl, P.a_sh, P.sharpen);
        lU.reset(128, P.a_vc); lV.reset(128, P.a_vc); sU.reset(128, ashc); sV.reset(128, ashc);
    }
    // the flip of the half-cycle positions :524-527 and the rescale :529-531 of one separated chroma sample
    DEV int unflip(int ch, int xo, const Magic31 &mA) const
    {
        const unsigned g = (unsigned)(xo - 2 + (int)xi) & 3u;
        if ((g == 0u && xo >= 2) || (g == 1u && xo >= 3)) ch = 255 - ch;
        return clampu8(sdivm((ch - 128) * 50, mA) + 128);
    }
    // separation 2 consumes composite sample x of the VCR's output (:929) and feeds the frame sink
    template <bool EDGE>
    DEV void sep2(const DevParams &P, int x, int c_in)
    {
        const unsigned c = (unsigned)c_in;
        if (EDGE && x == 0) { b2 = c; bsum = 32 + b2; return; }
        if (EDGE && x == 1) { b3 = c; bsum += b3; return; }
        const int xo = x - 2;
        bsum -= b0;
        b0 = b1; b1 = b2; b2 = b3; b3 = c;
        bsum += c;
        const unsigned yb = (bsum / 4u) & 0xFFu;
        int ch = clampu8((int)c + 128 - (int)yb);
        sink.luma(xo, (int)yb);
        ch = unflip(ch, xo, P.m_amp);
        if (!(xo & 1)) ev3 = ch;
        else {
            const int a = ev3, b = ch;
            int u = (xi & 1u) ? 255 - b : 255 - a, v = (xi & 1u) ? 255 - a : 255 - b;
            if (EDGE) sink.chroma(xo >> 1, u, v);
            else {
                if (sink.drop) { u = 128; v = 128; }
                sink.chroma_inner(xo >> 1, u, v);
            }
        }
    }
    // c0, c1: composite bytes 2i, 2i+1 after head switching (past the row: the caller's two bytes, then unused)
    // SV (S-Video out of the VCR, -vhs-svideo 1, :926: no re-modulation and no second separation): the VCR's luma and
    // chroma of sample m2 go to the frame sink as they are -- luma 2 m2, 2 m2 + 1 out of the delay line, chroma m2.
    template <bool EDGE, bool SV = false>
    DEV void iter(const DevParams &P, int i, int c0, int c1) { uint32_t none = 0; iter_part<EDGE, SV, 0>(P, i, c0, c1, none, 0u, 0u); }
    // PART (k422_pipe: the pass as two ROLES): 0 the whole iteration; 1 its FRONT -- separation 1, chroma / phase noise, the
    // VCR luma chain -- whose products leave as one packed word (luma 2i-2, luma 2i-1, U1, V1: four bytes, zero where the
    // guarded form has nothing); 2 its BACK -- VCR chroma low-pass, blend, sharpen, re-modulation, separation 2, the sink --
    // fed with this iteration's word (chroma) and the word of iteration i - D (the luma the delay line would hand over).
    template <bool EDGE, bool SV, int PART>
    DEV void iter_part(const DevParams &P, int i, int c0, int c1, uint32_t &xf_out, uint32_t xf_now, uint32_t xf_del)
    {
        // ---- main stream: separation 1 at x = 2i, 2i+1
        int yb_[2] = {0, 0};
        bool yok[2] = {false, false};
        int U1 = 0, V1 = 0;
        bool pair = false;
        if constexpr (PART == 2) {
            U1 = (int)((xf_now >> 16) & 0xFFu); V1 = (int)(xf_now >> 24);
            pair = !EDGE || (i >= 1 && 2 * i + 1 <= W + 1);      // (what the loop below would have found)
        }
        const int yd0 = PART == 2 ? (int)(xf_del & 0xFFu) : yq[0], yd1 = PART == 2 ? (int)((xf_del >> 8) & 0xFFu) : yq[1];
        if constexpr (PART != 2) {
#pragma unroll
        for (int sx = 0; sx < 2; sx++) {
            const int x = 2 * i + sx;
            if (EDGE && x > W + 1) continue;
            const unsigned c = (unsigned)(sx ? c1 : c0);
            if (EDGE && x == 0) { a2 = c; asum = 32 + a2; continue; }
            if (EDGE && x == 1) { a3 = c; asum += a3; continue; }
            const int xo = x - 2;
            asum -= a0;
            a0 = a1; a1 = a2; a2 = a3; a3 = c;
            asum += c;
            const unsigned yb = (asum / 4u) & 0xFFu;
            int ch = clampu8((int)c + 128 - (int)yb);
            yb_[sx] = (int)yb; yok[sx] = true;
            ch = unflip(ch, xo, P.m_amp_back);
            if (!(xo & 1)) ev1 = ch;
            else {
                const int a = ev1, b = ch;
                U1 = (xi & 1u) ? 255 - b : 255 - a; V1 = (xi & 1u) ? 255 - a : 255 - b;
                chroma_post422(P, cp, U1, V1);
                pair = true;
            }
        }
        }
        if constexpr (PART != 1) {
        // ---- VCR chroma: input m1 = i - 1, output m2 = i - 1 - D
        int fU = 0, fV = 0;
        if (!EDGE || pair) {
            fU = clampu8((int)lU.push((double)U1, a_vc));
            fV = clampu8((int)lV.push((double)V1, a_vc));
        }
        if (EDGE) {       // (the steady loop ends D + 1 inputs before the row does: the guarded steps refill this window)
#pragma unroll
            for (int q = 0; q < D; q++) { ru[q] = ru[q + 1]; rv[q] = rv[q + 1]; }
            ru[D] = U1; rv[D] = V1;
        }
        const int m2 = i - 1 - D;
        if (!EDGE || (m2 >= 0 && m2 < W2)) {
            int u = fU, v = fV;
            if (EDGE && m2 >= W2 - D) { u = ru[0]; v = rv[0]; }          // row tail keeps its input :848-853
            const int upU = fastdec::wave_up(u), upV = fastdec::wave_up(v);
            if (blend && k >= 1) {
                u = ((k >= 2 ? upU : 128) + u + 1) >> 1;
                v = ((k >= 2 ? upV : 128) + v + 1) >> 1;
            }
            double s = u;
            double ts = sU.push(s, a_sh_c);
            u = clampu8((int)(s + ((s - ts) * sharpen_c)));
            s = v;
            ts = sV.push(s, a_sh_c);
            v = clampu8((int)(s + ((s - ts) * sharpen_c)));
            if constexpr (SV) {
                sink.luma(2 * m2, yd0);
                sink.luma(2 * m2 + 1, yd1);
                if (EDGE) sink.chroma(m2, u, v);
                else {
                    if (sink.drop) { u = 128; v = 128; }
                    sink.chroma_inner(m2, u, v);
                }
            } else {
            // re-modulate :434-477 onto the VCR luma of 2 m2, 2 m2 + 1, separate again
#pragma unroll
            for (int sx = 0; sx < 2; sx++) {
                const int lx = 2 * m2 + sx;
                const unsigned ph = (xi + (unsigned)lx) & 3u;
                int chroma = ((ph & 1u) ? v - 128 : u - 128) * P.amp;
                if (ph & 2u) chroma = -chroma;
                sep2<EDGE>(P, lx, clampu8((sx ? yd1 : yd0) + chroma / 50));
            }
            }
        } else if (EDGE && !SV) {
            // lanes must stay in step for the wave shift of the blend: nothing to shift here (m2 is wave-uniform)
            if (m2 == W2) { sep2<true>(P, W, oob0); sep2<true>(P, W + 1, oob1); }   // Y[x+2] past the row :496
        }
        }
        // ---- VCR luma of positions 2i-2, 2i-1 enters the delay line (a dummy while outside the row)
        if constexpr (PART != 2) {
            int y1_[2];
#pragma unroll
        for (int sx = 0; sx < 2; sx++) {
            const int y1 = (!EDGE || yok[sx]) ? lv.run(yb_[sx]) : 0;
            y1_[sx] = y1;
            if constexpr (PART == 0) {
#pragma unroll
            for (int q = 0; q < 2 * D - 1; q++) yq[q] = yq[q + 1];
            yq[2 * D - 1] = y1;
            }
        }
            if constexpr (PART == 1)
                xf_out = (uint32_t)y1_[0] | ((uint32_t)y1_[1] << 8) | ((uint32_t)U1 << 16) | ((uint32_t)V1 << 24);
        }
    }

    // ---- the steady-state form of iter() for the preset's arithmetic identities.  Preconditions (launcher):
    // even scanline phase (xi is 0 or 2) and subcarrier amplitude 50 both ways, so that
    //   * (x * 50) / 50 == x: the rescale :529-531 and the modulation's (chroma * amp) / 50 :470 vanish;
    //   * the U/V pick :535-550 is u = 255 - a, v = 255 - b;
    //   * the half-cycle flip :524-527 at position xo >= 3 is "iff bit 1 of xo + xi is set": the XOR of a
    //     per-lane mask (xi == 2) and a property of the unrolled loop position (J: i = 4n + J);
    //   * the sign of the modulated chroma at luma lx :463-466 is bit 1 of xi + lx, likewise.
    // fm0 / fm1: 255 where a flip happens for xo & 2 == 0 / != 0; sm0 / sm1: -1 where the chroma is negated
    // for lx & 2 == 0 / != 0; bA / b128 / bC: the vertical blend as mask arithmetic; dmask: dropout.
    int fm0, fm1, sm0, sm1, bA, b128, bC, dmask;
    DEV void begin_fast(bool dropped)
    {
        const bool hi = (xi & 2u) != 0;
        fm0 = fastdec::opaque_v(hi ? 255 : 0); fm1 = fastdec::opaque_v(hi ? 0 : 255);
        sm0 = fastdec::opaque_v(hi ? -1 : 0); sm1 = fastdec::opaque_v(hi ? 0 : -1);
        const bool on = blend && k >= 1;
        bA = fastdec::opaque_v(on && k >= 2 ? -1 : 0);
        b128 = fastdec::opaque_v(on && k < 2 ? 128 : 0);
        bC = fastdec::opaque_v(on ? 1 : 0);
        dmask = fastdec::opaque_v(dropped ? 0 : -1);
    }
    template <int XO2>       // XO2 = xo & 2 of the position being separated
    DEV void sep2_fast(int xo, bool odd, int c_in)
    {
        const unsigned c = (unsigned)c_in;
        bsum -= b0;
        b0 = b1; b1 = b2; b2 = b3; b3 = c;
        bsum += c;
        const unsigned yb = (bsum >> 2) & 0xFFu;
        int ch = clampu8((int)c + 128 - (int)yb);
        sink.luma(xo, (int)yb);
        ch ^= XO2 ? fm1 : fm0;
        if (!odd) ev3 = ch;
        else {
            // dropout :932-942 as a mask: 128 + ((x - 128) & dmask)
            const int u = 128 + ((127 - ev3) & dmask), v = 128 + ((127 - ch) & dmask);
            sink.chroma_inner(xo >> 1, u, v);
        }
    }
    template <int J>
    DEV void iter_fast(const DevParams &P, int ib, int c0, int c1) { uint32_t none = 0; iter_fast_part<J, 0>(P, ib, c0, c1, none, 0u, 0u); }
    template <int J, int PART>                   // PART: as iter_part
    DEV void iter_fast_part(const DevParams &P, int ib, int c0, int c1, uint32_t &xf_out, uint32_t xf_now, uint32_t xf_del)
    {
        static_assert(D == 4, "the preset form: SP tape speed");
        const int i = ib + J;                     // ib is a multiple of 4
        // ---- main stream: separation 1 at x = 2i, 2i+1 (xo = 2i-2 even, 2i-1 odd; xo & 2 = (2J - 2) & 2)
        constexpr int XO2A = (2 * J + 2) & 2;
        int yb_[2] = {0, 0};
        int U1 = 0, V1 = 0;
        if constexpr (PART == 2) { U1 = (int)((xf_now >> 16) & 0xFFu); V1 = (int)(xf_now >> 24); }
        const int yd0 = PART == 2 ? (int)(xf_del & 0xFFu) : yq[0], yd1 = PART == 2 ? (int)((xf_del >> 8) & 0xFFu) : yq[1];
        if constexpr (PART != 2) {
        {
            const unsigned c = (unsigned)c0;
            asum -= a0; a0 = a1; a1 = a2; a2 = a3; a3 = c; asum += c;
            const unsigned yb = (asum >> 2) & 0xFFu;
            yb_[0] = (int)yb;
            ev1 = clampu8((int)c + 128 - (int)yb) ^ (XO2A ? fm1 : fm0);
        }
        {
            const unsigned c = (unsigned)c1;
            asum -= a0; a0 = a1; a1 = a2; a2 = a3; a3 = c; asum += c;
            const unsigned yb = (asum >> 2) & 0xFFu;
            yb_[1] = (int)yb;
            const int ch = clampu8((int)c + 128 - (int)yb) ^ (XO2A ? fm1 : fm0);
            U1 = 255 - ev1; V1 = 255 - ch;
            chroma_post422(P, cp, U1, V1);
        }
        }
        if constexpr (PART != 1) {
        // ---- VCR chroma: input m1 = i - 1, output m2 = i - 5 (strictly inside the row: filtered value)
        int u = clampu8((int)lU.push((double)U1, a_vc));
        int v = clampu8((int)lV.push((double)V1, a_vc));
        u = (((fastdec::wave_up(u) & bA) + b128) + u + bC) >> bC;
        v = (((fastdec::wave_up(v) & bA) + b128) + v + bC) >> bC;
        double s = u;
        double ts = sU.push(s, a_sh_c);
        u = clampu8((int)(s + ((s - ts) * sharpen_c)));
        s = v;
        ts = sV.push(s, a_sh_c);
        v = clampu8((int)(s + ((s - ts) * sharpen_c)));
        // ---- re-modulate onto the VCR luma of lx = 2i-10, 2i-9 (even: U, odd: V; lx & 2 = (2J - 10) & 2), and
        // separate again at xo = lx - 2 (xo & 2 = (2J) & 2)
        constexpr int LX2 = (2 * J + 2) & 2, XO2B = (2 * J) & 2;
        const int sm = LX2 ? sm1 : sm0;
        const int lx = 2 * i - 10;
        sep2_fast<XO2B>(lx - 2, false, clampu8(yd0 + (((u - 128) ^ sm) - sm)));
        sep2_fast<XO2B>(lx - 1, true, clampu8(yd1 + (((v - 128) ^ sm) - sm)));
        }
        // ---- VCR luma of positions 2i-2, 2i-1 enters the delay line
        if constexpr (PART != 2) {
            int y1_[2];
#pragma unroll
        for (int sx = 0; sx < 2; sx++) {
            const int y1 = lv.run(yb_[sx]);
            y1_[sx] = y1;
            if constexpr (PART == 0) {
#pragma unroll
            for (int q = 0; q < 7; q++) yq[q] = yq[q + 1];
            yq[7] = y1;
            }
        }
            if constexpr (PART == 1)
                xf_out = (uint32_t)y1_[0] | ((uint32_t)y1_[1] << 8) | ((uint32_t)U1 << 16) | ((uint32_t)V1 << 24);
        }
    }
};

} // namespace fused422

// SPEC: the switch set `ffmpeg_to_composite -vhs` runs (NTSC, SP tape speed, no pre-emphasis, luma / chroma /
// phase noise on, FULL output chroma low-pass -- ffmpeg_to_composite.cpp:278 default, selection :948-951; frame
// rows 16- / 8-byte aligned) as compile-time constants -- the sweeps then carry no state of branches
// the preset never takes.  Same arithmetic; every other switch set runs the SPEC = false kernel.
#ifndef F422_WAVES
#define F422_WAVES 1
#endif
// STREAM: sweeps B1-B3 as the one streamed pass of StreamB above (aligned frame rows; DD = the chroma delay
// of the tape speed).  With SPEC its steady loop is the preset's iter_fast, otherwise the guard-free form of
// the general iteration with the switches read at run time.
// SVID (with STREAM, not SPEC): the VCR with S-Video out -- the streamed pass without its re-modulation / second
// separation (StreamB::iter<EDGE, true>): "k422_fused_sv<D>".
template <bool SPEC, bool STREAM = false, int DD = 4, bool SVID = false>
__global__ __launch_bounds__(64, STREAM ? 2 : F422_WAVES) void k422_fused(DevParams P, GeomDev G,
                                                 const Field422Dev *__restrict__ fields,
                                                 Scratch422 Sc,
                                                 const uint32_t *__restrict__ rs_luma,
                                                 const int *__restrict__ n0_luma,
                                                 const uint32_t *__restrict__ rs_chroma,
                                                 const int *__restrict__ n0_u,
                                                 const int *__restrict__ n0_v,
                                                 const int *__restrict__ hs_shift,
                                                 const int *__restrict__ pn_noise,
                                                 const int *__restrict__ dropout,
                                                 double a_hp_i, double a_hp_q, double a_sh_c,
                                                 double sharpen_c)
{
    using namespace fused422;
    __shared__ uint32_t ring[31 * 64];
    // frame-row staging of the streamed form: 16 words (one 64-byte burst) per lane and plane, unpadded --
    // with the rand() ring 20,224 bytes per wave, so that eight waves still fit a CU's 160 KiB
    __shared__ __attribute__((aligned(16))) uint32_t fstage[STREAM ? 64 * 16 * 3 : 4];
    const int lane = threadIdx.x;
    const int gidx = blockIdx.x * 63 + lane - 1;          // lane 0 = halo (row above)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const Field422Dev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok && !(fd.flags & F422_NOCOMP);
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const unsigned xi = scan_phase422(P, y, fd.fieldno);
    const int W = P.W;
    const size_t slot = (size_t)blockIdx.x * 64 + lane;
    Row422 R;
    R.Y.p = Sc.Y + slot; R.T.p = Sc.T + slot; R.U.p = Sc.U + slot; R.V.p = Sc.V + slot;
    R.Y.S = R.T.S = R.U.S = R.V.S = Sc.S;
    uint8_t *fy = fd.dst[0] + (size_t)fd.dst_ls[0] * y;
    uint8_t *fu = fd.dst[1] + (size_t)fd.dst_ls[1] * y;
    uint8_t *fv = fd.dst[2] + (size_t)fd.dst_ls[2] * y;
    halo_redirect(Sc, lane, fy, fu, fv);
    // the two bytes the reference's Y/C separator reads past the row (:496), see ntsc422_kernels.hip
    int oob0 = 16, oob1 = 16;
    {
        const size_t off = (size_t)fd.dst_ls[0] * y + (size_t)W, end = (size_t)fd.dst_ls[0] * (size_t)P.H;
        if (off < end) oob0 = fy[W];
        if (off + 1 < end) oob1 = fy[W + 1];
    }

#ifdef F422_AB_TIMES
    unsigned long long t_prev__ = __builtin_readcyclecounter();
#endif
    // ---- A: frame row -> composite bytes
    {
        LumaPost422 lp_;
        lp_.pre_on = SPEC ? false : P.pre_on != 0; lp_.noise_on = SPEC ? true : P.noise_k != 0;
        lp_.pre.p = 16; lp_.noise = 0; lp_.ring = ring; lp_.lane = lane;
        if (lp_.noise_on) { lp_.rng.init(ring, rs_luma + rc, P.Rpad, lane); lp_.noise = n0_luma[rc]; }
        if (SPEC || P.ntsc) sweep_a<true, SPEC, SPEC && STREAM, STREAM>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q,
                                                                        fd.dst_ls[0], fd.dst_ls[1], fd.dst_ls[2], fstage);
        else sweep_a<false, SPEC, false, STREAM>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q, fd.dst_ls[0], fd.dst_ls[1],
                                                 fd.dst_ls[2], fstage);
    }
    F422_STAMP(0);
#if defined(F422_AB_STOP) && F422_AB_STOP <= 1      // timing-only A/B builds (WRONG frames): stop after a sweep
    return;
#endif
    // ---- head switching :669-732 (displaced copy, fill value 16)
    if (P.hs) {
        const int hs = hs_shift[rc];
        if (__any(hs != 0)) {
            // gathered from plane Y into plane T, 16 samples per step (16 independent loads in flight), then
            // the two planes trade places for the rest of the kernel -- the wave's columns of the scratch
            // planes are its own, and every lane of the wave takes this path (shift 0 = plain copy)
            const int tw = W + W / 10;
            Packer422 o; o.begin(R.T);
            constexpr int HB = 16;
            for (int x0 = 0; x0 < W; x0 += HB) {
                int v[HB], ix[HB];
#pragma unroll
                for (int j = 0; j < HB; j++) {
                    int idx = x0 + j + hs;
                    idx += (idx >> 31) & tw;
                    idx -= (idx >= tw) ? tw : 0;
                    ix[j] = idx;
                    v[j] = R.Y.byte_at(idx < W ? idx : W - 1);
                }
#pragma unroll
                for (int j = 0; j < HB; j++)
                    if (x0 + j < W) o.put(x0 + j, ix[j] < W ? v[j] : 16);
            }
            o.finish(W);
            { const Plane422 t = R.Y; R.Y = R.T; R.T = t; }
        }
    }
    F422_STAMP(1);
#if defined(F422_AB_STOP) && F422_AB_STOP <= 2
    return;
#endif
    if constexpr (STREAM) {
        // ---- B1 + B2 + B3 in one streamed pass (StreamB)
        StreamB<DD> B;
        B.begin(P, W, xi, k, oob0, oob1, a_sh_c, sharpen_c);
        B.cp.noise_on = SPEC ? true : P.cnoise_k != 0; B.cp.phase_on = SPEC ? true : P.pnoise_k != 0;
        B.cp.nU = B.cp.nV = 0; B.cp.cosv = 1; B.cp.sinv = 0; B.cp.ring = ring; B.cp.lane = lane;
        if (B.cp.noise_on) { B.cp.rng.init(ring, rs_chroma + rc, P.Rpad, lane); B.cp.nU = n0_u[rc]; B.cp.nV = n0_v[rc]; }
        if (B.cp.phase_on) {
            int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
            n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
            B.cp.cosv = G.ptab[2 * n]; B.cp.sinv = G.ptab[2 * n + 1];
        }
        B.sink.begin(P, true, SPEC ? 2 : P.out_lp, fy, fu, fv, is_out, P.loss && dropout[rc] != 0, a_hp_i, a_hp_q, W);
        B.sink.wy.st = fstage + lane * 16; B.sink.wu.st = fstage + (64 + lane) * 16; B.sink.wv.st = fstage + (128 + lane) * 16;
        if constexpr (SPEC) B.begin_fast(P.loss && dropout[rc] != 0);
        const int W2 = W / 2, NIT = W2 + DD + 2;
        auto in_byte = [&](int x) -> int { return x < W ? R.Y.byte_at(x) : (x == W ? oob0 : (x == W + 1 ? oob1 : 0)); };
        int i = 0;
        for (; i < 8 && i < NIT; i++) B.template iter<true, SVID>(P, i, in_byte(2 * i), in_byte(2 * i + 1));
        // steady iterations: every stage inside the row, and none of the last DD + 1 chroma inputs of the row
        // (those refill the row-tail windows of the low-passes, which only the guarded iteration maintains)
        if (i == 8 && i + 3 <= W2 - 2 - DD) {
            uint32_t w0 = R.Y.word(i >> 1), w1 = R.Y.word((i >> 1) + 1);
            for (; i + 3 <= W2 - 2 - DD; i += 4) {
                // the next two words are requested before this pair is worked on (clamped at the row end)
                const int qn = (i >> 1) + 2, qmax = (W - 1) >> 2;
                const uint32_t n0 = R.Y.word(qn <= qmax ? qn : qmax), n1 = R.Y.word(qn + 1 <= qmax ? qn + 1 : qmax);
                if constexpr (SPEC) {
                    B.template iter_fast<0>(P, i, byte_of(w0, 0), byte_of(w0, 1));
                    B.template iter_fast<1>(P, i, byte_of(w0, 2), byte_of(w0, 3));
                    B.template iter_fast<2>(P, i, byte_of(w1, 0), byte_of(w1, 1));
                    B.template iter_fast<3>(P, i, byte_of(w1, 2), byte_of(w1, 3));
                } else {
                    B.template iter<false, SVID>(P, i, byte_of(w0, 0), byte_of(w0, 1));
                    B.template iter<false, SVID>(P, i + 1, byte_of(w0, 2), byte_of(w0, 3));
                    B.template iter<false, SVID>(P, i + 2, byte_of(w1, 0), byte_of(w1, 1));
                    B.template iter<false, SVID>(P, i + 3, byte_of(w1, 2), byte_of(w1, 3));
                }
                w0 = n0; w1 = n1;
            }
        }
        for (; i < NIT; i++) B.template iter<true, SVID>(P, i, in_byte(2 * i), in_byte(2 * i + 1));
        B.sink.finish(W);
        F422_STAMP(4);
        return;
    }
    // ---- B1: Y/C separation + chroma noise + phase noise | VHS luma low-pass + sharpen
    {
        ChromaPost422 cp_;
        cp_.noise_on = SPEC ? true : P.cnoise_k != 0; cp_.phase_on = SPEC ? true : P.pnoise_k != 0;
        cp_.nU = cp_.nV = 0; cp_.cosv = 1; cp_.sinv = 0; cp_.ring = ring; cp_.lane = lane;
        if (cp_.noise_on) { cp_.rng.init(ring, rs_chroma + rc, P.Rpad, lane); cp_.nU = n0_u[rc]; cp_.nV = n0_v[rc]; }
        if (cp_.phase_on) {
            int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
            n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
            cp_.cosv = G.ptab[2 * n]; cp_.sinv = G.ptab[2 * n + 1];
        }
        LumaVhs lv;
        lv.begin(P.a_vl, P.a_sh, P.sharpen);
        FrameSink none;
        demod<false>(P, R, W, xi, P.m_amp_back, oob0, oob1, cp_, lv, none);
    }
    F422_STAMP(2);
#if defined(F422_AB_STOP) && F422_AB_STOP <= 3
    return;
#endif
    // ---- B2: VHS chroma low-pass + blend + sharpen | re-modulate
    sweep_b2<SPEC ? 4 : 0>(P, R, W, xi, k, a_sh_c, sharpen_c);
    F422_STAMP(3);
#if defined(F422_AB_STOP) && F422_AB_STOP <= 4
    return;
#endif
    // ---- B3: Y/C separation | dropout | output chroma low-pass -> frame
    {
        FrameSink sink;
        sink.begin(P, SPEC, SPEC ? 2 : P.out_lp, fy, fu, fv, is_out, P.loss && dropout[rc] != 0, a_hp_i, a_hp_q, W);
        ChromaPost422 nocp;
        LumaVhs nolv;
        demod<true>(P, R, W, xi, P.m_amp, oob0, oob1, nocp, nolv, sink);
    }
    F422_STAMP(4);
}

// ---------------------------------------------------------------------------------- the latency form (round 6)
// k422_fused's streamed form as four ROLES of one workgroup (the BGRA tool's k_field_pipe, ntsc_pipe.hip: a lone wavefront
// gets one instruction through per ~5 cycles, so a launch of a few fields is bound by the length of one row's instruction
// stream): wavefront 0 runs sweep A; wavefront 1 the head-switch gather (only in workgroups that hold a switched row);
// wavefront 2 the FRONT of the streamed B pass -- separation 1, chroma and phase noise, the VCR luma chain --; wavefront 3
// its BACK -- VCR chroma low-pass, blend, sharpen, re-modulation, second separation, dropout, output low-pass, the frame.
// The SAME code as the one-wave form (sweep_a; StreamB's iterations cut where only bytes cross: iter_part / iter_fast_part),
// so the same bytes; what is new is that the stages of a row run side by side.  Hand-offs: the composite bytes travel
// through the scratch planes as before (plane Y: A -> G, F; plane T: G -> F), behind byte counts in LDS: A publishes a group
// of 64 bytes once the group BEHIND it has been stored (its loads of the next group, issued before those stores, have
// returned: vmcnt is one in-order counter), G one step of 16 bytes behind its stores; the readers use streaming loads
// (always from the L2) and wait for the highest byte they are about to request.  F -> K: one packed word per lane and
// iteration (luma 2i-2, luma 2i-1, U1, V1) in a ring of 32 iterations in LDS; K reads word i for the chroma and word i - D
// for the luma (the delay line of the one-wave form).  A leads K by construction, and K's frame bursts only cover bytes
// A has consumed long before (A reads the frame a group ahead of the bytes it emits), so running in place stays safe.
// Launcher: the streamed forms' preconditions, launches of the host-frame engine.
template <bool SPEC, int DD = 4, bool SVID = false>
__global__ __launch_bounds__(256) void k422_pipe(DevParams P, GeomDev G, const Field422Dev *__restrict__ fields, Scratch422 Sc,
                                                 const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                                                 const uint32_t *__restrict__ rs_chroma, const int *__restrict__ n0_u,
                                                 const int *__restrict__ n0_v, const int *__restrict__ hs_shift,
                                                 const int *__restrict__ pn_noise, const int *__restrict__ dropout,
                                                 double a_hp_i, double a_hp_q, double a_sh_c, double sharpen_c,
                                                 unsigned *__restrict__ fault)
{
    using namespace fused422;
    using pipe::lds_flag;
    __shared__ uint32_t ring_a[31 * 64];                                   // A: luma noise
    __shared__ uint32_t ring_b[31 * 64];                                   // B: chroma noise
    __shared__ __attribute__((aligned(16))) uint32_t tile_a[64 * 44];      // A: cooperative frame-row loads (CoopRows)
    __shared__ __attribute__((aligned(16))) uint32_t fstage[64 * 16 * 3];  // B: frame bursts
    constexpr int XR = 32;                                                 // iterations the front may run ahead of the back
    __shared__ uint32_t xring[XR * 64];                                    // F -> K: one packed word per lane and iteration
    __shared__ uint32_t flags[4];      // [0] bytes of plane Y, [1] bytes of plane T, [2] iterations F has produced, [3] K has finished
    typedef __attribute__((address_space(3))) uint32_t *lds_w;
    const lds_w xr = (lds_w)xring;
    const int role = (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 63 + lane - 1;          // lane 0 = halo (row above)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const Field422Dev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok && !(fd.flags & F422_NOCOMP);
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const unsigned xi = scan_phase422(P, y, fd.fieldno);
    const int W = P.W;
    const size_t slot = (size_t)blockIdx.x * 64 + lane;
    Row422 R;
    R.Y.p = Sc.Y + slot; R.T.p = Sc.T + slot; R.U.p = Sc.U + slot; R.V.p = Sc.V + slot;
    R.Y.S = R.T.S = R.U.S = R.V.S = Sc.S;
    uint8_t *fy = fd.dst[0] + (size_t)fd.dst_ls[0] * y;
    uint8_t *fu = fd.dst[1] + (size_t)fd.dst_ls[1] * y;
    uint8_t *fv = fd.dst[2] + (size_t)fd.dst_ls[2] * y;
    halo_redirect(Sc, lane, fy, fu, fv);
    int oob0 = 16, oob1 = 16;
    {
        const size_t off = (size_t)fd.dst_ls[0] * y + (size_t)W, end = (size_t)fd.dst_ls[0] * (size_t)P.H;
        if (off < end) oob0 = fy[W];
        if (off + 1 < end) oob1 = fy[W + 1];
    }
    const int hs = P.hs ? hs_shift[rc] : 0;
    const bool gather = P.hs && __any(hs != 0);          // (the same 64 rows in every role: the same answer)
    if (threadIdx.x < 4) flags[threadIdx.x] = 0u;
    if (threadIdx.x < 8) pipe::g_waited[threadIdx.x] = 0u;
    if (threadIdx.x == 0) pipe::g_fault = 0u;
    __syncthreads();
    const lds_flag fl = (lds_flag)flags;
    typedef __attribute__((address_space(1))) const uint32_t *g_cw;

    if (role == 0) {
        // ---- A: frame row -> composite bytes (plane Y)
        constexpr int D = 4;                               // (NTSC; PAL: 2 -- the count below only has to be a lower bound)
        LumaPost422 lp_;
        lp_.pre_on = SPEC ? false : P.pre_on != 0; lp_.noise_on = SPEC ? true : P.noise_k != 0;
        lp_.pre.p = 16; lp_.noise = 0; lp_.ring = ring_a; lp_.lane = lane;
        if (lp_.noise_on) { lp_.rng.init(ring_a, rs_luma + rc, P.Rpad, lane); lp_.noise = n0_luma[rc]; }
        // group g0 emits the bytes of chroma samples g0 - D .. g0 + 31 - D; when its hook runs, the group before it is in
        // memory (at most this group's 16 word stores are still in flight)
        auto pub = [&](int g0) {
            const int done = 2 * (g0 - D);
            if (done > 0) { NTSC_PIPE_VMCNT(16); pipe::publish(fl, done < W ? done : W); }
        };
        if (SPEC || P.ntsc) sweep_a<true, SPEC, SPEC, true>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q,
                                                            fd.dst_ls[0], fd.dst_ls[1], fd.dst_ls[2], tile_a, pub);
        else sweep_a<false, SPEC, false, true>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q, fd.dst_ls[0], fd.dst_ls[1],
                                               fd.dst_ls[2], tile_a, pub);
        NTSC_PIPE_VMCNT(0);
        pipe::publish(fl, W);
    } else if (role == 1) {
        // ---- G: head switching :669-732 (displaced copy Y -> T, fill value 16), 16 samples per step
        if (gather) {
            const int tw = W + W / 10;
            // how far ahead of x a lane may read: its forward displacement -- and the whole row when a backward displacement
            // beyond W/10 makes the row's first samples come from its end (ntsc_pipe.hip: wg_reach)
            int reach;
            {
                int mx = hs, mn = hs;
#pragma unroll
                for (int o_ = 32; o_ >= 1; o_ >>= 1) {
                    const int a_ = __shfl_xor(mx, o_), b_ = __shfl_xor(mn, o_);
                    mx = a_ > mx ? a_ : mx; mn = b_ < mn ? b_ : mn;
                }
                mx = __builtin_amdgcn_readfirstlane(mx); mn = __builtin_amdgcn_readfirstlane(mn);
                reach = mn < -(W / 10) ? W : (mx > 0 ? mx : 0) + 2;
            }
            int seen = 0;
            Packer422 o; o.begin(R.T);
            constexpr int HB = 16;
            for (int x0 = 0; x0 < W; x0 += HB) {
                { const int c = x0 + HB + reach; pipe::wait_ge(fl, c < W ? c : W, seen); }
                int v[HB], ix[HB];
#pragma unroll
                for (int j = 0; j < HB; j++) {
                    int idx = x0 + j + hs;
                    idx += (idx >> 31) & tw;
                    idx -= (idx >= tw) ? tw : 0;
                    ix[j] = idx;
                    const int xr = idx < W ? idx : W - 1;
                    const uint32_t wv = __builtin_nontemporal_load((g_cw)(R.Y.p + (size_t)(xr >> 2) * R.Y.S));
                    v[j] = (int)((wv >> (8 * (xr & 3))) & 0xFFu);
                }
#pragma unroll
                for (int j = 0; j < HB; j++)
                    if (x0 + j < W) o.put(x0 + j, ix[j] < W ? v[j] : 16);
                // (this step's 4 word stores may still be in flight; everything before them has landed)
                NTSC_PIPE_VMCNT(4);
                pipe::publish(fl + 1, x0);
            }
            o.finish(W);
            NTSC_PIPE_VMCNT(0);
            pipe::publish(fl + 1, W);
        }
    } else if (role == 2) {
        // ---- F: the FRONT of the streamed B pass (StreamB::iter_part<.., 1>): separation 1, chroma / phase noise, VCR luma
        // chain over plane Y (T where the workgroup gathered) -> one packed word per iteration in the LDS ring
        const Plane422 IN = gather ? R.T : R.Y;
        const lds_flag fin = gather ? fl + 1 : fl;
        int seen = 0, kseen = 0;
        StreamB<DD> B;
        B.begin(P, W, xi, k, oob0, oob1, a_sh_c, sharpen_c);
        B.cp.noise_on = SPEC ? true : P.cnoise_k != 0; B.cp.phase_on = SPEC ? true : P.pnoise_k != 0;
        B.cp.nU = B.cp.nV = 0; B.cp.cosv = 1; B.cp.sinv = 0; B.cp.ring = ring_b; B.cp.lane = lane;
        if (B.cp.noise_on) { B.cp.rng.init(ring_b, rs_chroma + rc, P.Rpad, lane); B.cp.nU = n0_u[rc]; B.cp.nV = n0_v[rc]; }
        if (B.cp.phase_on) {
            int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
            n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
            B.cp.cosv = G.ptab[2 * n]; B.cp.sinv = G.ptab[2 * n + 1];
        }
        if constexpr (SPEC) B.begin_fast(false);
        const int W2 = W / 2, NIT = W2 + DD + 2;
        auto need = [&](int bytes) { pipe::wait_ge(fin, bytes < W ? bytes : W, seen); };
        auto ldw = [&](int q) -> uint32_t { return __builtin_nontemporal_load((g_cw)(IN.p + (size_t)q * IN.S)); };
        auto in_byte = [&](int x) -> int {
            if (x >= W) return x == W ? oob0 : (x == W + 1 ? oob1 : 0);
            need(x + 1);
            return (int)((ldw(x >> 2) >> (8 * (x & 3))) & 0xFFu);
        };
        // iteration w may be written once the back has finished iteration w - XR + DD (it reads words i and i - DD)
        auto room = [&](int w) { pipe::wait_ge(fl + 3, w - XR + 1 + DD, kseen); };
        auto edge = [&](int i) {
            uint32_t xf = 0;
            const int c0 = in_byte(2 * i), c1 = in_byte(2 * i + 1);
            B.template iter_part<true, SVID, 1>(P, i, c0, c1, xf, 0u, 0u);
            room(i);
            xr[(i & (XR - 1)) * 64 + lane] = xf;
            pipe::publish(fl + 2, i + 1);
        };
        int i = 0;
        for (; i < 8 && i < NIT; i++) edge(i);
        if (i == 8 && i + 3 <= W2 - 2 - DD) {
            need(4 * ((i >> 1) + 2));
            uint32_t w0 = ldw(i >> 1), w1 = ldw((i >> 1) + 1);
            for (; i + 3 <= W2 - 2 - DD; i += 4) {
                // the next two words are requested before this pair is worked on (clamped at the row end)
                const int qn = (i >> 1) + 2, qmax = (W - 1) >> 2;
                need(4 * (qn + 2));
                const uint32_t n0 = ldw(qn <= qmax ? qn : qmax), n1 = ldw(qn + 1 <= qmax ? qn + 1 : qmax);
                uint32_t xf[4];
                if constexpr (SPEC) {
                    B.template iter_fast_part<0, 1>(P, i, byte_of(w0, 0), byte_of(w0, 1), xf[0], 0u, 0u);
                    B.template iter_fast_part<1, 1>(P, i, byte_of(w0, 2), byte_of(w0, 3), xf[1], 0u, 0u);
                    B.template iter_fast_part<2, 1>(P, i, byte_of(w1, 0), byte_of(w1, 1), xf[2], 0u, 0u);
                    B.template iter_fast_part<3, 1>(P, i, byte_of(w1, 2), byte_of(w1, 3), xf[3], 0u, 0u);
                } else {
                    B.template iter_part<false, SVID, 1>(P, i, byte_of(w0, 0), byte_of(w0, 1), xf[0], 0u, 0u);
                    B.template iter_part<false, SVID, 1>(P, i + 1, byte_of(w0, 2), byte_of(w0, 3), xf[1], 0u, 0u);
                    B.template iter_part<false, SVID, 1>(P, i + 2, byte_of(w1, 0), byte_of(w1, 1), xf[2], 0u, 0u);
                    B.template iter_part<false, SVID, 1>(P, i + 3, byte_of(w1, 2), byte_of(w1, 3), xf[3], 0u, 0u);
                }
                room(i + 3);
#pragma unroll
                for (int j = 0; j < 4; j++) xr[((i + j) & (XR - 1)) * 64 + lane] = xf[j];
                pipe::publish(fl + 2, i + 4);
                w0 = n0; w1 = n1;
            }
        }
        for (; i < NIT; i++) edge(i);
    } else {
        // ---- K: the BACK of the pass (iter_part<.., 2>): VCR chroma low-pass, blend, sharpen, re-modulation, second
        // separation, dropout, output low-pass -> frame bursts
        int fseen = 0;
        StreamB<DD> B;
        B.begin(P, W, xi, k, oob0, oob1, a_sh_c, sharpen_c);
        B.cp.noise_on = false; B.cp.phase_on = false;
        B.sink.begin(P, true, SPEC ? 2 : P.out_lp, fy, fu, fv, is_out, P.loss && dropout[rc] != 0, a_hp_i, a_hp_q, W);
        B.sink.wy.st = fstage + lane * 16; B.sink.wu.st = fstage + (64 + lane) * 16; B.sink.wv.st = fstage + (128 + lane) * 16;
        if constexpr (SPEC) B.begin_fast(P.loss && dropout[rc] != 0);
        const int W2 = W / 2, NIT = W2 + DD + 2;
        auto word = [&](int it) -> uint32_t { return it >= 0 ? xr[(it & (XR - 1)) * 64 + lane] : 0u; };
        auto edge = [&](int i) {
            pipe::wait_ge(fl + 2, i + 1, fseen);
            uint32_t none = 0;
            const uint32_t now = word(i), del = word(i - DD);
            B.template iter_part<true, SVID, 2>(P, i, 0, 0, none, now, del);
            pipe::publish(fl + 3, i + 1);
        };
        int i = 0;
        for (; i < 8 && i < NIT; i++) edge(i);
        if (i == 8 && i + 3 <= W2 - 2 - DD) {
            for (; i + 3 <= W2 - 2 - DD; i += 4) {
                pipe::wait_ge(fl + 2, i + 4, fseen);
                uint32_t now[4], del[4], none = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { now[j] = word(i + j); del[j] = word(i + j - DD); }
                if constexpr (SPEC) {
                    B.template iter_fast_part<0, 2>(P, i, 0, 0, none, now[0], del[0]);
                    B.template iter_fast_part<1, 2>(P, i, 0, 0, none, now[1], del[1]);
                    B.template iter_fast_part<2, 2>(P, i, 0, 0, none, now[2], del[2]);
                    B.template iter_fast_part<3, 2>(P, i, 0, 0, none, now[3], del[3]);
                } else {
                    B.template iter_part<false, SVID, 2>(P, i, 0, 0, none, now[0], del[0]);
                    B.template iter_part<false, SVID, 2>(P, i + 1, 0, 0, none, now[1], del[1]);
                    B.template iter_part<false, SVID, 2>(P, i + 2, 0, 0, none, now[2], del[2]);
                    B.template iter_part<false, SVID, 2>(P, i + 3, 0, 0, none, now[3], del[3]);
                }
                pipe::publish(fl + 3, i + 4);
            }
        }
        for (; i < NIT; i++) edge(i);
        B.sink.finish(W);
    }
    if (lane == 0 && *(lds_flag)&pipe::g_fault) *fault = 1u + blockIdx.x;
}

// ---------------------------------------------------------------------------------- the short form
// A switch set WITHOUT the VCR (the tool's default preset, ffmpeg_to_composite.cpp:267-333) in two sweeps instead of the
// twelve of k422_process (same sweeps, same arithmetic, same scratch plane as k422_fused<false>): A, then ONE decode
// sweep -- Y/C separation :480-553 -> chroma noise :738-754 -> phase noise :755-781 -> dropout :932-942 -> output
// chroma low-pass :948-951 -> the frame row ("k422_direct").
// (The VCR with S-Video out, -vhs-svideo 1, is the streamed pass without its re-modulation: k422_fused<false,true,D,true>.
//  A three-sweep form of it -- A, B1, B2 to the frame -- was built in round 5 and measured SLOWER than the twelve sweeps of
//  k422_process, 608k against 631k frames/s: its sweeps use 166-220 registers and wait on the scratch planes; dropped.)
// Preconditions as k422_fused (launcher): colour subcarrier on, input chroma low-pass on, no
// -nocolor-subcarrier-after-yc-sep, no extra -yc-recomb passes.
// FASTA (launcher: NTSC, no pre-emphasis, luma noise on, even scanline phase, subcarrier amplitude 50, frame rows aligned
// to 16 / 8 bytes -- the tool's default preset qualifies): sweep A in the streamed preset's form (cooperative 64-byte row
// loads through an LDS tile, guard-free blocks inside the row), and the decode sweep writes the frame in 64-byte bursts
// staged in the same tile ("k422_direct_fast").
template <bool FASTA = false>
__global__ __launch_bounds__(64, F422_WAVES) void k422_short(DevParams P, GeomDev G,
                                                 const Field422Dev *__restrict__ fields,
                                                 Scratch422 Sc,
                                                 const uint32_t *__restrict__ rs_luma,
                                                 const int *__restrict__ n0_luma,
                                                 const uint32_t *__restrict__ rs_chroma,
                                                 const int *__restrict__ n0_u,
                                                 const int *__restrict__ n0_v,
                                                 const int *__restrict__ hs_shift,
                                                 const int *__restrict__ pn_noise,
                                                 const int *__restrict__ dropout,
                                                 double a_hp_i, double a_hp_q, double a_sh_c,
                                                 double sharpen_c)
{
    using namespace fused422;
    __shared__ uint32_t ring[31 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t fstage[FASTA ? 64 * 16 * 3 : 4];
    const int lane = threadIdx.x;
    const int gidx = blockIdx.x * 63 + lane - 1;          // (the VCR forms' row mapping, halo lane included: one launcher)
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const Field422Dev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok && !(fd.flags & F422_NOCOMP);
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const unsigned xi = scan_phase422(P, y, fd.fieldno);
    const int W = P.W;
    const size_t slot = (size_t)blockIdx.x * 64 + lane;
    Row422 R;
    R.Y.p = Sc.Y + slot; R.T.p = Sc.T + slot; R.U.p = Sc.U + slot; R.V.p = Sc.V + slot;
    R.Y.S = R.T.S = R.U.S = R.V.S = Sc.S;
    uint8_t *fy = fd.dst[0] + (size_t)fd.dst_ls[0] * y;
    uint8_t *fu = fd.dst[1] + (size_t)fd.dst_ls[1] * y;
    uint8_t *fv = fd.dst[2] + (size_t)fd.dst_ls[2] * y;
    halo_redirect(Sc, lane, fy, fu, fv);
    int oob0 = 16, oob1 = 16;                             // the separator's two bytes past the row (:496)
    {
        const size_t off = (size_t)fd.dst_ls[0] * y + (size_t)W, end = (size_t)fd.dst_ls[0] * (size_t)P.H;
        if (off < end) oob0 = fy[W];
        if (off + 1 < end) oob1 = fy[W + 1];
    }
    // ---- A: frame row -> composite bytes
    {
        LumaPost422 lp_;
        lp_.pre_on = P.pre_on != 0; lp_.noise_on = P.noise_k != 0;
        lp_.pre.p = 16; lp_.noise = 0; lp_.ring = ring; lp_.lane = lane;
        if (lp_.noise_on) { lp_.rng.init(ring, rs_luma + rc, P.Rpad, lane); lp_.noise = n0_luma[rc]; }
        if constexpr (FASTA) sweep_a<true, true, true, true>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q, fd.dst_ls[0], fd.dst_ls[1],
                                                              fd.dst_ls[2], fstage);
        else if (P.ntsc) sweep_a<true, false>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q);
        else sweep_a<false, false>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q);
    }
    // ---- head switching :669-732 (as in k422_fused)
    if (P.hs) {
        const int hs = hs_shift[rc];
        if (__any(hs != 0)) {
            const int tw = W + W / 10;
            Packer422 o; o.begin(R.T);
            constexpr int HB = 16;
            for (int x0 = 0; x0 < W; x0 += HB) {
                int v[HB], ix[HB];
#pragma unroll
                for (int j = 0; j < HB; j++) {
                    int idx = x0 + j + hs;
                    idx += (idx >> 31) & tw;
                    idx -= (idx >= tw) ? tw : 0;
                    ix[j] = idx;
                    v[j] = R.Y.byte_at(idx < W ? idx : W - 1);
                }
#pragma unroll
                for (int j = 0; j < HB; j++)
                    if (x0 + j < W) o.put(x0 + j, ix[j] < W ? v[j] : 16);
            }
            o.finish(W);
            { const Plane422 t = R.Y; R.Y = R.T; R.T = t; }
        }
    }
    ChromaPost422 cp_;
    cp_.noise_on = P.cnoise_k != 0; cp_.phase_on = P.pnoise_k != 0;
    cp_.nU = cp_.nV = 0; cp_.cosv = 1; cp_.sinv = 0; cp_.ring = ring; cp_.lane = lane;
    if (cp_.noise_on) { cp_.rng.init(ring, rs_chroma + rc, P.Rpad, lane); cp_.nU = n0_u[rc]; cp_.nV = n0_v[rc]; }
    if (cp_.phase_on) {
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        cp_.cosv = G.ptab[2 * n]; cp_.sinv = G.ptab[2 * n + 1];
    }
    if constexpr (FASTA) {
        // ---- the whole decode side in one sweep, the frame row leaving in 64-byte bursts (aligned rows)
        FrameSinkBurst bs;
        bs.begin(P, true, P.out_lp, fy, fu, fv, is_out, P.loss && dropout[rc] != 0, a_hp_i, a_hp_q, W);
        bs.wy.st = fstage + lane * 16; bs.wu.st = fstage + (64 + lane) * 16; bs.wv.st = fstage + (128 + lane) * 16;
        LumaVhs nolv;
        demod<true, true, FrameSinkBurst>(P, R, W, xi, P.m_amp_back, oob0, oob1, cp_, nolv, bs);
    } else {
        // ---- the whole decode side in one sweep
        FrameSink sink;
        sink.begin(P, false, P.out_lp, fy, fu, fv, is_out, P.loss && dropout[rc] != 0, a_hp_i, a_hp_q, W);
        LumaVhs nolv;
        demod<true, true>(P, R, W, xi, P.m_amp_back, oob0, oob1, cp_, nolv, sink);
    }
    (void)a_sh_c; (void)sharpen_c;
}

// The no-VCR form's latency form: sweep A | head-switch gather | the decode sweep as three wavefronts of one workgroup
// (k422_pipe's arrangement and hand-off protocol; `demod` reads the composite bytes behind A's byte count through the NEED
// hook of sweep_blocks).  FASTA preconditions (the tool's default preset qualifies).
__global__ __launch_bounds__(192) void k422_short_pipe(DevParams P, GeomDev G, const Field422Dev *__restrict__ fields, Scratch422 Sc,
                                                       const uint32_t *__restrict__ rs_luma, const int *__restrict__ n0_luma,
                                                       const uint32_t *__restrict__ rs_chroma, const int *__restrict__ n0_u,
                                                       const int *__restrict__ n0_v, const int *__restrict__ hs_shift,
                                                       const int *__restrict__ pn_noise, const int *__restrict__ dropout,
                                                       double a_hp_i, double a_hp_q, unsigned *__restrict__ fault)
{
    using namespace fused422;
    using pipe::lds_flag;
    __shared__ uint32_t ring_a[31 * 64];
    __shared__ uint32_t ring_b[31 * 64];
    __shared__ __attribute__((aligned(16))) uint32_t tile_a[64 * 44];
    __shared__ __attribute__((aligned(16))) uint32_t fstage[64 * 16 * 3];
    __shared__ uint32_t flags[4];
    const int role = (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int gidx = blockIdx.x * 63 + lane - 1;
    const int rc = gidx < 0 ? 0 : (gidx < P.R ? gidx : P.R - 1);
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const Field422Dev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const bool is_out = lane >= 1 && gidx < P.R && rowok && !(fd.flags & F422_NOCOMP);
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;
    const unsigned xi = scan_phase422(P, y, fd.fieldno);
    const int W = P.W;
    const size_t slot = (size_t)blockIdx.x * 64 + lane;
    Row422 R;
    R.Y.p = Sc.Y + slot; R.T.p = Sc.T + slot; R.U.p = Sc.U + slot; R.V.p = Sc.V + slot;
    R.Y.S = R.T.S = R.U.S = R.V.S = Sc.S;
    uint8_t *fy = fd.dst[0] + (size_t)fd.dst_ls[0] * y;
    uint8_t *fu = fd.dst[1] + (size_t)fd.dst_ls[1] * y;
    uint8_t *fv = fd.dst[2] + (size_t)fd.dst_ls[2] * y;
    halo_redirect(Sc, lane, fy, fu, fv);
    int oob0 = 16, oob1 = 16;
    {
        const size_t off = (size_t)fd.dst_ls[0] * y + (size_t)W, end = (size_t)fd.dst_ls[0] * (size_t)P.H;
        if (off < end) oob0 = fy[W];
        if (off + 1 < end) oob1 = fy[W + 1];
    }
    const int hs = P.hs ? hs_shift[rc] : 0;
    const bool gather = P.hs && __any(hs != 0);
    if (threadIdx.x < 4) flags[threadIdx.x] = 0u;
    if (threadIdx.x < 8) pipe::g_waited[threadIdx.x] = 0u;
    if (threadIdx.x == 0) pipe::g_fault = 0u;
    __syncthreads();
    const lds_flag fl = (lds_flag)flags;
    typedef __attribute__((address_space(1))) const uint32_t *g_cw;
    if (role == 0) {
        LumaPost422 lp_;
        lp_.pre_on = P.pre_on != 0; lp_.noise_on = P.noise_k != 0;
        lp_.pre.p = 16; lp_.noise = 0; lp_.ring = ring_a; lp_.lane = lane;
        if (lp_.noise_on) { lp_.rng.init(ring_a, rs_luma + rc, P.Rpad, lane); lp_.noise = n0_luma[rc]; }
        auto pub = [&](int g0) {
            const int done = 2 * (g0 - 4);
            if (done > 0) { NTSC_PIPE_VMCNT(16); pipe::publish(fl, done < W ? done : W); }
        };
        sweep_a<true, true, true, true>(P, R, fy, fu, fv, W, xi, lp_, a_hp_i, a_hp_q, fd.dst_ls[0], fd.dst_ls[1], fd.dst_ls[2], tile_a, pub);
        NTSC_PIPE_VMCNT(0);
        pipe::publish(fl, W);
    } else if (role == 1) {
        if (gather) {
            const int tw = W + W / 10;
            // how far ahead of x a lane may read: its forward displacement -- and the whole row when a backward displacement
            // beyond W/10 makes the row's first samples come from its end (ntsc_pipe.hip: wg_reach)
            int reach;
            {
                int mx = hs, mn = hs;
#pragma unroll
                for (int o_ = 32; o_ >= 1; o_ >>= 1) {
                    const int a_ = __shfl_xor(mx, o_), b_ = __shfl_xor(mn, o_);
                    mx = a_ > mx ? a_ : mx; mn = b_ < mn ? b_ : mn;
                }
                mx = __builtin_amdgcn_readfirstlane(mx); mn = __builtin_amdgcn_readfirstlane(mn);
                reach = mn < -(W / 10) ? W : (mx > 0 ? mx : 0) + 2;
            }
            int seen = 0;
            Packer422 o; o.begin(R.T);
            constexpr int HB = 16;
            for (int x0 = 0; x0 < W; x0 += HB) {
                { const int c = x0 + HB + reach; pipe::wait_ge(fl, c < W ? c : W, seen); }
                int v[HB], ix[HB];
#pragma unroll
                for (int j = 0; j < HB; j++) {
                    int idx = x0 + j + hs;
                    idx += (idx >> 31) & tw;
                    idx -= (idx >= tw) ? tw : 0;
                    ix[j] = idx;
                    const int xr = idx < W ? idx : W - 1;
                    const uint32_t wv = __builtin_nontemporal_load((g_cw)(R.Y.p + (size_t)(xr >> 2) * R.Y.S));
                    v[j] = (int)((wv >> (8 * (xr & 3))) & 0xFFu);
                }
#pragma unroll
                for (int j = 0; j < HB; j++)
                    if (x0 + j < W) o.put(x0 + j, ix[j] < W ? v[j] : 16);
                NTSC_PIPE_VMCNT(4);
                pipe::publish(fl + 1, x0);
            }
            o.finish(W);
            NTSC_PIPE_VMCNT(0);
            pipe::publish(fl + 1, W);
        }
    } else {
        if (gather) R.Y = R.T;
        int seen = 0;
        Need422 need{gather ? fl + 1 : fl, W, &seen};
        ChromaPost422 cp_;
        cp_.noise_on = P.cnoise_k != 0; cp_.phase_on = P.pnoise_k != 0;
        cp_.nU = cp_.nV = 0; cp_.cosv = 1; cp_.sinv = 0; cp_.ring = ring_b; cp_.lane = lane;
        if (cp_.noise_on) { cp_.rng.init(ring_b, rs_chroma + rc, P.Rpad, lane); cp_.nU = n0_u[rc]; cp_.nV = n0_v[rc]; }
        if (cp_.phase_on) {
            int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
            n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
            cp_.cosv = G.ptab[2 * n]; cp_.sinv = G.ptab[2 * n + 1];
        }
        FrameSinkBurst bs;
        bs.begin(P, true, P.out_lp, fy, fu, fv, is_out, P.loss && dropout[rc] != 0, a_hp_i, a_hp_q, W);
        bs.wy.st = fstage + lane * 16; bs.wu.st = fstage + (64 + lane) * 16; bs.wv.st = fstage + (128 + lane) * 16;
        LumaVhs nolv;
        demod<true, true, FrameSinkBurst, Need422>(P, R, W, xi, P.m_amp_back, oob0, oob1, cp_, nolv, bs, need);
    }
    if (lane == 0 && *(lds_flag)&pipe::g_fault) *fault = 1u + blockIdx.x;
}

} // namespace ntscsim
