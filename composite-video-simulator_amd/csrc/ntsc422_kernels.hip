// ntsc422_kernels.hip -- the 8-bit YUV422P sibling of the field path:
// ffmpeg_to_composite.cpp composite_video_process() :629-952 (+ helpers :353-553),
// render_field() :1001-1129, black_key_feedback() :954-999.
//
// First-cut mapping: same execution model as ntsc_kernels.hip -- ONE LANE =
// ONE SCANLINE, 63 rows + 1 halo row per wavefront -- but the stage chain is NOT yet fused: every
// loop of the reference is one sweep of the lane over its row, on byte planes kept TRANSPOSED and
// PACKED in HBM scratch (four samples per 32-bit word, plane[x>>2][slot], slot = wave*64 + lane),
// so that each wave access moves two full 128-byte lines.  Every stage clamps to uint8 like the
// reference (clampu8 :335), so the sweeps communicate through bytes exactly as the reference's
// in-place frame does.
//
// The reference's Y/C separator reads Y[x+2] two bytes past the row (:496).  For every row whose two
// following bytes lie inside the caller's luma plane (the next row's first pixels, or linesize
// padding) that read is deterministic, and the kernel reproduces it: the two bytes are fetched once
// per row before anything is modified (they belong to a row of the OTHER field or to padding, which
// this call never writes) and fed to every Y/C separation of the row (oracle: TOCOMP_OOB_MEMORY).
// Only where the read would leave the plane (last row with linesize < W + 2) it returns 16, the box
// filter's own pre-charge value.  The reference's three writes past its chroma scratch array
// (:529-532) are dropped, as in the oracle.
//
// Included by ntscsim_hip.hip after ntsc_kernels.hip (shares OnePole/Lp3/LaneRand/helpers).
#pragma clang fp contract(off)

namespace ntscsim {

DEV int clampu8(int x) { return x > 255 ? 255 : (x < 0 ? 0 : x); }

struct Field422Dev {
    uint8_t *dst[3];
    const uint8_t *src[3];
    uint8_t *flt[3];
    int32_t dst_ls[3], src_ls[3], flt_ls[3];
    int32_t src_height;
    uint32_t field, flags;
    uint64_t fieldno;
};
enum : uint32_t {
    F422_INTERLACED = 1u, F422_TFF = 2u, F422_SRC420 = 4u, F422_SECOND = 8u, F422_NOCOMP = 16u
};

struct Scratch422 {
    uint32_t *Y, *T, *U, *V;       // words of 4 samples: [ceil(W/4)][S] x2, [ceil(W/8)][S] x2
    size_t S;                      // slots = waves * 64
    uint8_t *halo;                 // [workgroup][3 planes][halo_pitch]: the INPUT of the row above every workgroup (k422_halo)
    uint32_t halo_pitch;
};

// The halo lane (lane 0 of every workgroup but the first) re-computes the row ABOVE the workgroup's rows for the vertical
// blend -- a row that the neighbouring workgroup rewrites IN PLACE.  A workgroup dispatched after its neighbour has got
// that far (a launch of more workgroups than the chip holds at once, or one that shares the chip with other streams) would
// read output where it needs input: tools/halo_race_probe.py (1,600 fields in one launch: the first row of workgroup 2,048
// differed from the same fields in launches of 8).  So the rows in question are copied aside by k422_halo BEFORE the
// process kernel starts, and the halo lane reads the copy (64-byte aligned, as long as the frame's linesize + the two
// bytes the separator reads past the row :496).
DEV void halo_redirect(const Scratch422 &Sc, int lane, uint8_t *&fy, uint8_t *&fu, uint8_t *&fv)
{
    if (lane == 0 && blockIdx.x > 0 && Sc.halo) {
        uint8_t *b = Sc.halo + (size_t)blockIdx.x * 3u * Sc.halo_pitch;
        fy = b; fu = b + Sc.halo_pitch; fv = b + 2u * (size_t)Sc.halo_pitch;
    }
}

// scanline phase of the 8-bit tool (:449-460 / :508-522): phase 0 ignores the offset, PAL differs
DEV unsigned scan_phase422(const DevParams &P, unsigned y, uint64_t fieldno)
{
    if (P.ntsc) {
        const unsigned off = (unsigned)P.phase_off;
        if (P.phase_mode == 90)  return (unsigned)((fieldno + off + (y >> 1)) & 3);
        if (P.phase_mode == 180) return (unsigned)((((fieldno + y) & 2) + off) & 3);
        if (P.phase_mode == 270) return (unsigned)((fieldno + off - (y >> 1)) & 3);
        return 0;
    }
    return (unsigned)((fieldno + y) & 3);
}

// ------------------------------------------------------------------------------ halo rows (see halo_redirect)
// block b copies the input of the row above workgroup b + 1 (row rc = 63 (b + 1) - 1 of the batch's row space): linesize
// bytes of each plane, + 2 of the luma plane (the separator's bytes past the row), never past the end of a plane
__global__ __launch_bounds__(256) void k422_halo(DevParams P, const Field422Dev *__restrict__ fields, Scratch422 Sc)
{
    const int wg = (int)blockIdx.x + 1;
    const int rc = wg * 63 - 1;
    if (rc >= P.R) return;
    const int f = rc / P.Lslot, k = rc - f * P.Lslot;
    const Field422Dev &fd = fields[f];
    const unsigned field = fd.field & 1u;
    const bool rowok = (int)(field + 2u * k) < P.H;
    const unsigned y = rowok ? field + 2u * (unsigned)k : field;       // (the process kernels' row of that lane)
    uint8_t *out = Sc.halo + (size_t)wg * 3u * Sc.halo_pitch;
    for (int pl = 0; pl < 3; pl++) {
        const size_t ls = (size_t)fd.dst_ls[pl];
        const uint8_t *src = fd.dst[pl] + ls * y;
        size_t n = ls + (pl == 0 ? 2u : 0u);
        const size_t left = ls * (size_t)(P.H - (int)y);               // bytes from the row's start to the end of the plane
        if (n > left) n = left;
        if (n > Sc.halo_pitch) n = Sc.halo_pitch;
        uint8_t *dst = out + (size_t)pl * Sc.halo_pitch;
        for (size_t t = threadIdx.x; t < n; t += blockDim.x) dst[t] = src[t];
    }
}

// ------------------------------------------------------------------------------ render_field
// One thread per output byte of the field's rows (all three planes); pure integer lerp between
// two source rows (:1076-1128).
__global__ void k422_render(DevParams P, const Field422Dev *__restrict__ fields)
{
    const int f = blockIdx.z;
    const Field422Dev &fd = fields[f];
    if (!fd.src[0]) return;
    const unsigned field = fd.field & 1u;
    const int k = blockIdx.y;
    const unsigned y = field + 2u * (unsigned)k;
    if ((int)y >= P.H) return;
    const bool is420 = (fd.flags & F422_SRC420) != 0;
    const unsigned sh = (unsigned)fd.src_height;
    const unsigned chroma_height = is420 ? sh >> 1 : sh;
    unsigned sy = (y * 0x100u * sh) / (unsigned)P.H;
    unsigned syf = sy & 0xFF;
    sy >>= 8;
    unsigned csy = sy, csyf = syf, sy2, csy2;
    if (is420) { if (!(csy & 1)) csyf = 0; csy >>= 1; }
    if (fd.flags & F422_INTERLACED) {
        unsigned which = (fd.flags & F422_TFF) ? 0u : 1u;
        if (fd.flags & F422_SECOND) which ^= 1u;
        if (which == 0) { sy++; if (!(sy & 1u)) syf = 0; else sy--; }
        else if (!(sy & 1u)) { syf = 0; sy++; }
        if (which == 0) { csy++; if (!(csy & 1u)) csyf = 0; else csy--; }
        else if (!(csy & 1u)) { csyf = 0; csy++; }
        if (sy >= sh - 2) { sy = sh - 2; syf = 0; }
        sy2 = sy + 2;
        if (csy >= chroma_height - 2) { csy = chroma_height - 2; csyf = 0; }
        csy2 = csy + 1;
    } else {
        if (sy >= sh - 1) { sy = sh - 1; syf = 0; }
        sy2 = sy + 1;
        if (csy >= chroma_height - 1) { csy = chroma_height - 1; csyf = 0; }
        csy2 = csy + 1;
    }
    const int W2 = P.W / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.W + 2 * W2; i += gridDim.x * blockDim.x) {
        const int pl = i < P.W ? 0 : (i < P.W + W2 ? 1 : 2);
        const int x = pl == 0 ? i : (pl == 1 ? i - P.W : i - P.W - W2);
        const bool c420 = is420 && pl > 0;
        const unsigned r1 = c420 ? csy : sy, r2 = c420 ? csy2 : sy2, fr = c420 ? csyf : syf;
        const uint8_t *s1 = fd.src[pl] + (size_t)fd.src_ls[pl] * r1;
        uint8_t *o = fd.dst[pl] + (size_t)fd.dst_ls[pl] * y;
        if (fr == 0) o[x] = s1[x];
        else {
            const uint8_t *s2 = fd.src[pl] + (size_t)fd.src_ls[pl] * r2;
            o[x] = (uint8_t)(s1[x] + ((uint8_t)((((int)s2[x] - (int)s1[x]) * (int)fr) >> 8)));
        }
    }
}

// ------------------------------------------------------------------------------ black key
// black_key_feedback :954-999: one thread per chroma sample (pixel pair).
__global__ void k422_bkey(DevParams P, const Field422Dev *__restrict__ fields, int level)
{
    const int f = blockIdx.z;
    const Field422Dev &fd = fields[f];
    if (!fd.flt[0]) return;
    const unsigned y = (fd.field & 1u) + 2u * blockIdx.y;
    if ((int)y >= P.H) return;
    const int W2 = (P.W + 1) / 2;       // the reference steps x += 2 while x < width
    uint8_t *dY = fd.dst[0] + (size_t)fd.dst_ls[0] * y, *dU = fd.dst[1] + (size_t)fd.dst_ls[1] * y,
            *dV = fd.dst[2] + (size_t)fd.dst_ls[2] * y;
    uint8_t *fY = fd.flt[0] + (size_t)fd.flt_ls[0] * y, *fU = fd.flt[1] + (size_t)fd.flt_ls[1] * y,
            *fV = fd.flt[2] + (size_t)fd.flt_ls[2] * y;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < W2; c += gridDim.x * blockDim.x) {
        int u = dU[c], v = dV[c];
        // first pixel of the pair (with chroma)
        {
            int yy = dY[2 * c];
            const int dl = yy - (16 + level);
            int dc = u + v - 256; dc = (dc < 0 ? -dc : dc) - level;
            if (dl + dc <= 0) { yy = fY[2 * c]; u = fU[c]; v = fV[c]; }
            dY[2 * c] = (uint8_t)yy; fY[2 * c] = (uint8_t)yy;
            dU[c] = (uint8_t)u; dV[c] = (uint8_t)v; fU[c] = (uint8_t)u; fV[c] = (uint8_t)v;
        }
        // second pixel: keyed against the (possibly replaced) chroma, luma only
        {
            int yy = dY[2 * c + 1];
            const int dl = yy - (16 + level);
            int dc = u + v - 256; dc = (dc < 0 ? -dc : dc) - level;
            if (dl + dc <= 0) yy = fY[2 * c + 1];
            dY[2 * c + 1] = (uint8_t)yy; fY[2 * c + 1] = (uint8_t)yy;
        }
    }
}

// ------------------------------------------------------------------------------ output frame
// The pixel work of output_frame() ffmpeg_to_composite.cpp:1177-1236: the rows of `field` of a
// processed YUV422P frame are line-doubled ("bob") into the encoder's frame, which is YUV422P
// (:1177-1196) or YUV420P (:1197-1236; there the chroma rows are DECIMATED by the copy itself, no
// swscale involved).  One block per destination luma row.
struct Out422Dev {
    const uint8_t *frame[3];
    uint8_t *bob[3];
    int32_t frame_ls[3], bob_ls[3];
    uint32_t field, mode;
    uint32_t crows, _pad;       // chroma rows of the bob frame that may be written (0: no limit), see OUT422_INTERLACED420
};
enum : uint32_t { OUT422_BOB422 = 0u, OUT422_BOB420 = 1u, OUT422_INTERLACED420 = 2u, OUT422_FRAME = 3u };

DEV void copy_row422(uint8_t *__restrict__ d, const uint8_t *__restrict__ s, int nbytes, bool al4)
{
    if (al4) {
        const int nw = nbytes >> 2;
        for (int i = threadIdx.x; i < nw; i += blockDim.x)
            ((uint32_t *)d)[i] = ((const uint32_t *)s)[i];
        for (int i = (nw << 2) + threadIdx.x; i < nbytes; i += blockDim.x) d[i] = s[i];
    } else {
        for (int i = threadIdx.x; i < nbytes; i += blockDim.x) d[i] = s[i];
    }
}

// one row of output_frame's copy (a workgroup per row)
DEV void output422_row(const DevParams &P, const Out422Dev &o, const unsigned y, int al4)
{
    if (o.mode > OUT422_FRAME) return;                             // (a record of the host engine's ring with no output_frame)
    const unsigned H = (unsigned)P.H;
    unsigned sy;
    if (o.mode == OUT422_INTERLACED420 || o.mode == OUT422_FRAME) sy = y;     // :1202-1203; :1158 the frame as it is
    else if (o.field) sy = y | 1u;                                 // 1, 1, 3, 3, ...  :1181-1184
    else sy = (y + 1u) & ~1u;                                      // 0, 2, 2, 4, 4, ...
    if (sy >= H) sy -= 2u;                                         // :1186-1187
    copy_row422(o.bob[0] + (size_t)o.bob_ls[0] * y, o.frame[0] + (size_t)o.frame_ls[0] * sy, P.W, al4);
    bool chroma = true;
    unsigned cy = y;
    if (o.mode == OUT422_BOB420) { chroma = (y & 1u) == 0u; cy = y >> 1; }                   // :1225-1226
    else if (o.mode == OUT422_INTERLACED420) { chroma = (y & 2u) == 0u; cy = (y & 1u) + ((y & ~3u) >> 1); }  // :1215-1216
    if (o.crows && cy >= o.crows) chroma = false;    // (the repack's row past a 4:2:0 plane, :1215-1223, when the frame has no room for it)
    if (chroma)
        for (int p = 1; p <= 2; p++)
            copy_row422(o.bob[p] + (size_t)o.bob_ls[p] * cy, o.frame[p] + (size_t)o.frame_ls[p] * sy,
                        P.W / 2, al4);
}

__global__ void k422_output(DevParams P, const Out422Dev *__restrict__ outs, int al4)
{
    output422_row(P, outs[blockIdx.y], blockIdx.x, al4);
}

// ------------------------------------------------------------------------------ the field
// Row planes in HBM scratch: FOUR consecutive samples per 32-bit word, words transposed:
// plane[x >> 2][slot].  One wave access = 64 lanes x 4 B = two full 128-byte lines.
struct Plane422 {
    uint32_t *p;      // this lane's column
    size_t S;
    DEV uint32_t word(int q) const { return p[(size_t)q * S]; }
    DEV void set_word(int q, uint32_t w) const { p[(size_t)q * S] = w; }
    DEV int byte_at(int x) const { return (int)((word(x >> 2) >> (8 * (x & 3))) & 0xFFu); }   // random access
};

// Sequential byte writer: samples arrive in increasing x, a word is stored when its 4th byte does.
struct Packer422 {
    Plane422 pl;
    uint32_t acc;
    DEV void begin(const Plane422 &q) { pl = q; acc = 0; }
    DEV void put(int x, int v)
    {
        const uint32_t sh = 8u * (unsigned)(x & 3);
        acc = (x & 3) ? (acc | ((uint32_t)v << sh)) : (uint32_t)v;
        if ((x & 3) == 3) pl.set_word(x >> 2, acc);
    }
    // n samples were put; positions >= n of the last word keep what the plane holds
    DEV void finish(int n)
    {
        if (n & 3) {
            const uint32_t mask = (1u << (8 * (n & 3))) - 1u;
            const uint32_t old = pl.word(n >> 2);
            pl.set_word(n >> 2, (old & ~mask) | (acc & mask));
        }
    }
};

// Blocks of 8 samples (two words); the next block is requested before the current one is
// processed, so the memory latency overlaps the filter arithmetic.
constexpr int BK = 8;
struct Reader422 {
    Plane422 pl;
    int nwords;
    uint32_t c0, c1, n0, n1;
    DEV void begin(const Plane422 &q, int n)
    {
        pl = q; nwords = (n + 3) >> 2;
        c0 = nwords > 0 ? pl.word(0) : 0u;
        c1 = nwords > 1 ? pl.word(1) : 0u;
        n0 = n1 = 0;
    }
    DEV void prefetch(int x0)           // block starting at x0 + 8
    {
        const int q = (x0 >> 2) + 2;
        n0 = q < nwords ? pl.word(q) : 0u;
        n1 = q + 1 < nwords ? pl.word(q + 1) : 0u;
    }
    DEV int get(int j) const { return (int)(((j < 4 ? c0 : c1) >> (8 * (j & 3))) & 0xFFu); }
    DEV void advance() { c0 = n0; c1 = n1; }
};

#define SWEEP_BEGIN(PLANE, N)                                                    \
    {                                                                            \
        Reader422 rd_; rd_.begin((PLANE), (N));                                  \
        for (int x0_ = 0; x0_ < (N); x0_ += BK) {                                \
            rd_.prefetch(x0_);                                                   \
            _Pragma("unroll") for (int j_ = 0; j_ < BK; j_++) {                  \
                const int x = x0_ + j_;                                          \
                if (x >= (N)) break;                                             \
                const int in = rd_.get(j_);
#define SWEEP_END                                                                \
            }                                                                    \
            rd_.advance();                                                       \
        }                                                                        \
    }

struct Row422 {
    Plane422 Y, T, U, V;
};

// composite_video_chroma_lowpass :353-393 (full) on one plane of one row: output for input x
// lands at x - delay; the last `delay` samples keep their input
DEV void chroma_lp_full422(const Plane422 &P0, int W2, double a_lp, double a_hp, int delay)
{
    Lp3 lp; lp.reset(128);
    OnePole hp; hp.p = 128;
    Packer422 out; out.begin(P0);
    SWEEP_BEGIN(P0, W2)
        double s = in;
        s += hp.hp(s, a_hp);
        s = lp.push(s, a_lp);
        if (x >= delay) out.put(x - delay, clampu8((int)s));
    SWEEP_END
    out.finish(W2 > delay ? W2 - delay : 0);
}
// composite_video_chroma_lowpass_lite :395-431 and the VHS chroma low-pass :834-855
DEV void chroma_lp_plain422(const Plane422 &P0, int W2, double a, int delay)
{
    Lp3 lp; lp.reset(128);
    Packer422 out; out.begin(P0);
    SWEEP_BEGIN(P0, W2)
        const double s = lp.push((double)in, a);
        if (x >= delay) out.put(x - delay, clampu8((int)s));
    SWEEP_END
    out.finish(W2 > delay ? W2 - delay : 0);
}
// Post-stages that are pointwise/causal on the freshly modulated luma and can therefore ride in
// the modulation sweep: pre-emphasis :636-651 and luma noise :654-666 (each with its own clamp).
struct LumaPost422 {
    bool pre_on, noise_on;
    OnePole pre;
    LaneRand rng;
    int noise;
    uint32_t *ring;
    int lane;
};

// composite_video_yuv_to_ntsc :434-477 (one chroma sample modulates two luma samples)
DEV void modulate422(const DevParams &P, const Row422 &R, int W, unsigned xi, int amp, bool nocolor,
                     LumaPost422 *post = nullptr)
{
    const int W2 = W / 2;
    Reader422 ru, rv;
    ru.begin(R.U, W2); rv.begin(R.V, W2);
    Packer422 oy; oy.begin(R.Y);
    int cu = 0, cv = 0;
    // 8 luma samples use 4 chroma samples: the chroma readers advance every other luma block
    SWEEP_BEGIN(R.Y, W)
        if (!(x & 1)) {
            const int cj = (x >> 1) & 7;
            if (cj == 0 && x > 0) { ru.advance(); rv.advance(); }
            if (cj == 0) { ru.prefetch(x >> 1); rv.prefetch(x >> 1); }
            cu = ((x >> 1) < W2 ? ru.get(cj) : 128) - 128;
            cv = ((x >> 1) < W2 ? rv.get(cj) : 128) - 128;
        }
        const unsigned s = (xi + (unsigned)x) & 3u;
        int chroma = ((s & 1u) ? cv : cu) * amp;
        if (s & 2u) chroma = -chroma;
        int yv = clampu8(in + chroma / 50);
        if (post) {
            if (post->pre_on) {
                double sd = yv;
                sd += post->pre.hp(sd, P.a_pre) * P.pre_gain;
                yv = clampu8((int)sd);
            }
            if (post->noise_on) {
                yv = clampu8(yv + post->noise);
                post->noise = sdiv2(post->noise + (int)umod31(post->rng.next(post->ring, post->lane), P.m_noise) - P.noise_k);
            }
        }
        oy.put(x, yv);
    SWEEP_END
    oy.finish(W);
    if (nocolor) {
        const int nw = (W2 + 3) >> 2;
        for (int q = 0; q < nw; q++) { R.U.set_word(q, 0x80808080u); R.V.set_word(q, 0x80808080u); }
    }
}
// composite_ntsc_to_yuv :480-553 in ONE sweep (out-of-row read = 16, out-of-array writes dropped).
// The reference's flip loop (:524-527) negates positions x+2, x+3 for x = (4-xi)&3 + 4m, x < W:
// position p is flipped iff g = (p-2+xi)&3 is 0 (and p >= 2) or 1 (and p >= 3); flip, rescale
// (:529-531) and the U/V pick (:535-550) are applied as soon as a pixel pair is complete.
// Chroma post-stages riding in the Y/C separation sweep: chroma noise :738-754 and the phase
// noise :755-781 (pointwise on the samples the separator has just produced, each with its clamp).
struct ChromaPost422 {
    bool noise_on, phase_on;
    LaneRand rng;
    int nU, nV;
    double cosv, sinv;
    uint32_t *ring;
    int lane;
};
DEV void chroma_post422(const DevParams &P, ChromaPost422 &cp, int &u, int &v)
{
    if (cp.noise_on) {
        u = clampu8(u + cp.nU);
        v = clampu8(v + cp.nV);
        cp.nU = sdiv2(cp.nU + (int)umod31(cp.rng.next(cp.ring, cp.lane), P.m_cnoise) - P.cnoise_k);
        cp.nV = sdiv2(cp.nV + (int)umod31(cp.rng.next(cp.ring, cp.lane), P.m_cnoise) - P.cnoise_k);
    }
    if (cp.phase_on) {
        const double du = u - 128, dv = v - 128;
        const double u_ = (du * cp.cosv) - (du * cp.sinv);
        const double v_ = (dv * cp.cosv) + (dv * cp.sinv);
        u = clampu8((int)(u_ + 128));
        v = clampu8((int)(v_ + 128));
    }
}

// POST: the chroma post-stages ride in this sweep; their state is copied into registers for the sweep
// (a struct reached through a pointer would live in scratch memory)
template <bool POST>
DEV void demodulate422(const DevParams &P, const Row422 &R, int W, unsigned xi, const Magic31 &mA,
                       bool after_yc_sep, int oob0, int oob1, const ChromaPost422 &cpost_in)
{
    ChromaPost422 cpost = cpost_in;
    const int W2 = W / 2;
    // the reader runs over the row; sample x+2 is needed at step x: keep a 2-sample look-ahead
    unsigned d0 = 16, d1 = 16, d2 = 0, d3 = 0, sum = 0;
    int la0 = 16, la1 = 16;            // Y[x], Y[x+1] relative to the reader position
    int ch_even = 0;
    Packer422 oy, ou, ov;
    oy.begin(R.Y); ou.begin(R.U); ov.begin(R.V);
    // Step the reader position r = x + 2.  Outputs at x = r - 2 go to words the reader has passed.
    SWEEP_BEGIN(R.Y, W + 2)
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
            if (after_yc_sep) {                                  // :503-507
                oy.put(xo, ch);
                if (!(xo & 1)) { int u = 128, v = 128; if (POST) chroma_post422(P, cpost, u, v); ou.put(xo >> 1, u); ov.put(xo >> 1, v); }
            } else {
                oy.put(xo, (int)yb);
                const unsigned g = (unsigned)(xo - 2 + (int)xi) & 3u;
                if ((g == 0u && xo >= 2) || (g == 1u && xo >= 3)) ch = 255 - ch;
                ch = clampu8(sdivm((ch - 128) * 50, mA) + 128);
                if (!(xo & 1)) ch_even = ch;
                else {
                    const int a = ch_even, b = ch;
                    int u = (xi & 1u) ? 255 - b : 255 - a, v = (xi & 1u) ? 255 - a : 255 - b;
                    if (POST) chroma_post422(P, cpost, u, v);
                    ou.put(xo >> 1, u);
                    ov.put(xo >> 1, v);
                }
            }
        }
        (void)la0; (void)la1;
    SWEEP_END
    oy.finish(W); ou.finish(W2); ov.finish(W2);
}

__global__ __launch_bounds__(64) void k422_process(DevParams P, GeomDev G,
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
                                                   double sharpen_c, int yc_recombine,
                                                   int after_yc_sep)
{
    __shared__ uint32_t ring[31 * 64];
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
    const int W = P.W, W2 = P.W / 2;
    const size_t slot = (size_t)blockIdx.x * 64 + lane;
    Row422 R;
    R.Y.p = Sc.Y + slot; R.T.p = Sc.T + slot; R.U.p = Sc.U + slot; R.V.p = Sc.V + slot;
    R.Y.S = R.T.S = R.U.S = R.V.S = Sc.S;
    uint8_t *fy = fd.dst[0] + (size_t)fd.dst_ls[0] * y;
    uint8_t *fu = fd.dst[1] + (size_t)fd.dst_ls[1] * y;
    uint8_t *fv = fd.dst[2] + (size_t)fd.dst_ls[2] * y;
    halo_redirect(Sc, lane, fy, fu, fv);
    // the two bytes the reference's Y/C separator reads past the row (:496): inside the luma plane
    // they are the caller's own bytes, outside it (last row, linesize < W + 2) the defined value 16
    int oob0 = 16, oob1 = 16;
    {
        const size_t off = (size_t)fd.dst_ls[0] * y + (size_t)W, end = (size_t)fd.dst_ls[0] * (size_t)P.H;
        if (off < end) oob0 = fy[W];
        if (off + 1 < end) oob1 = fy[W + 1];
    }

    // ---- frame row -> packed transposed scratch.  A scratch word holds 4 consecutive samples in
    // memory order, so aligned rows are moved 16 (luma) / 8 (chroma) bytes per lane per load.
    {
        int x0 = 0;
        if (P.src_al16)
            for (; x0 + 16 <= W; x0 += 16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(fy + x0);
                const int q = x0 >> 2;
                R.Y.set_word(q, v.x); R.Y.set_word(q + 1, v.y); R.Y.set_word(q + 2, v.z); R.Y.set_word(q + 3, v.w);
            }
        Packer422 o; o.begin(R.Y);
        for (; x0 < W; x0++) o.put(x0, fy[x0]);
        o.finish(W);
        int c0 = 0;
        if (P.dst_al16)
            for (; c0 + 8 <= W2; c0 += 8) {
                const uint2 a = *reinterpret_cast<const uint2 *>(fu + c0);
                const uint2 b = *reinterpret_cast<const uint2 *>(fv + c0);
                const int q = c0 >> 2;
                R.U.set_word(q, a.x); R.U.set_word(q + 1, a.y);
                R.V.set_word(q, b.x); R.V.set_word(q + 1, b.y);
            }
        Packer422 ou, ov; ou.begin(R.U); ov.begin(R.V);
        for (; c0 < W2; c0++) { ou.put(c0, fu[c0]); ov.put(c0, fv[c0]); }
        ou.finish(W2); ov.finish(W2);
    }

    // ---- input chroma low-pass :632
    if (P.in_lp) {
        chroma_lp_full422(R.U, W2, P.a_in_i, a_hp_i, 2);
        chroma_lp_full422(R.V, W2, P.ntsc ? P.a_in_q : P.a_in_i, P.ntsc ? a_hp_q : a_hp_i, P.ntsc ? 4 : 2);
    }
    // ---- modulate :633 + pre-emphasis :636-651 + luma noise :654-666 in one sweep
    {
        LumaPost422 lp_;
        lp_.pre_on = P.pre_on != 0; lp_.noise_on = P.noise_k != 0;
        lp_.pre.p = 16; lp_.noise = 0; lp_.ring = ring; lp_.lane = lane;
        if (lp_.noise_on) { lp_.rng.init(ring, rs_luma + rc, P.Rpad, lane); lp_.noise = n0_luma[rc]; }
        modulate422(P, R, W, xi, P.amp, P.nocolor != 0, &lp_);
    }
    // ---- head switching :669-732 (displaced copy, fill value 16)
    if (P.hs) {
        const int hs = hs_shift[rc];
        if (__any(hs != 0)) {
            const int tw = W + W / 10;
            const int nw = (W + 3) >> 2;
            for (int q = 0; q < nw; q++) R.T.set_word(q, R.Y.word(q));
            Packer422 o; o.begin(R.Y);
            for (int x0 = 0; x0 < W; x0 += BK) {
                int v[BK], ix[BK];
#pragma unroll
                for (int j = 0; j < BK; j++) {
                    int idx = x0 + j + hs;
                    idx += (idx >> 31) & tw;
                    idx -= (idx >= tw) ? tw : 0;
                    ix[j] = idx;
                    v[j] = R.T.byte_at(idx < W ? idx : W - 1);
                }
#pragma unroll
                for (int j = 0; j < BK; j++)
                    if (x0 + j < W) o.put(x0 + j, ix[j] < W ? v[j] : 16);
            }
            o.finish(W);
        }
    }
    // ---- Y/C separation :734 + chroma noise :738-754 + phase noise :755-781
    {
        ChromaPost422 cp_;
        cp_.noise_on = P.cnoise_k != 0; cp_.phase_on = P.pnoise_k != 0;
        cp_.nU = cp_.nV = 0; cp_.cosv = 1; cp_.sinv = 0; cp_.ring = ring; cp_.lane = lane;
        if (cp_.noise_on) { cp_.rng.init(ring, rs_chroma + rc, P.Rpad, lane); cp_.nU = n0_u[rc]; cp_.nV = n0_v[rc]; }
        if (cp_.phase_on) {
            int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
            n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
            cp_.cosv = G.ptab[2 * n]; cp_.sinv = G.ptab[2 * n + 1];
        }
        if (!P.nocolor) demodulate422<true>(P, R, W, xi, P.m_amp_back, after_yc_sep != 0, oob0, oob1, cp_);
        else if (cp_.noise_on || cp_.phase_on) {
            // no separation (-nocolor-subcarrier): the noise stages still run on the stored chroma
            Reader422 rv; rv.begin(R.V, W2);
            Packer422 ou, ov; ou.begin(R.U); ov.begin(R.V);
            SWEEP_BEGIN(R.U, W2)
                if (j_ == 0) rv.prefetch(x0_);
                int u = in, v = rv.get(j_);
                chroma_post422(P, cp_, u, v);
                ou.put(x, u); ov.put(x, v);
                if (j_ == BK - 1 || x == W2 - 1) rv.advance();
            SWEEP_END
            ou.finish(W2); ov.finish(W2);
        }
    }
    // ---- VHS block :786-930
    if (P.vhs) {
        {   // luma low-pass + emphasis :812-831, then sharpen :887-901 (both causal: one sweep)
            Lp3 lp; lp.reset(16);
            OnePole pre; pre.p = 16;
            Lp3 sh; sh.reset(16);
            Packer422 o; o.begin(R.Y);
            SWEEP_BEGIN(R.Y, W)
                double s = in;
                s = lp.push(s, P.a_vl);
                s += pre.hp(s, P.a_vl) * 1.6;
                const double y1 = clampu8((int)s);
                const double ts = sh.push(y1, P.a_sh);
                o.put(x, clampu8((int)(y1 + ((y1 - ts) * P.sharpen))));
            SWEEP_END
            o.finish(W);
        }
        {   // chroma low-pass :834-855 (output lands d samples back, the last d keep their input)
            // -> vertical blend :862-882 -> chroma sharpen :904-924, all in output order
            const int d = P.cdelay;
            const bool blend = P.vblend && P.ntsc;
            Lp3 lU, lV, sU, sV;
            lU.reset(128); lV.reset(128); sU.reset(128); sV.reset(128);
            int wu[7] = {0, 0, 0, 0, 0, 0, 0}, wv[7] = {0, 0, 0, 0, 0, 0, 0};   // last 7 inputs, [6] newest
            Reader422 rv; rv.begin(R.V, W2);
            Packer422 ou, ov; ou.begin(R.U); ov.begin(R.V);
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
                    // raw input at xo = d samples back (d is 4, 5 or 6)
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
                    ou.put(xo, clampu8((int)(s + ((s - ts) * sharpen_c))));
                    s = v;
                    ts = sV.push(s, a_sh_c);
                    ov.put(xo, clampu8((int)(s + ((s - ts) * sharpen_c))));
                }
                if (j_ == BK - 1 || x == W2 + d - 1) rv.advance();
            SWEEP_END
            ou.finish(W2); ov.finish(W2);
        }
        if (!P.svideo) {                                     // :926-929
            modulate422(P, R, W, xi, P.amp, P.nocolor != 0);
            demodulate422<false>(P, R, W, xi, P.m_amp, after_yc_sep != 0, oob0, oob1, ChromaPost422());
        }
    }
    // ---- chroma dropout :932-942
    if (P.loss && dropout[rc]) {
        const int nw = (W2 + 3) >> 2;
        for (int q = 0; q < nw; q++) { R.U.set_word(q, 0x80808080u); R.V.set_word(q, 0x80808080u); }
    }
    // ---- extra Y/C recombine passes :943-946
    for (int i = 0; i < yc_recombine; i++) {
        modulate422(P, R, W, xi, P.amp, P.nocolor != 0);
        demodulate422<false>(P, R, W, xi, P.m_amp, after_yc_sep != 0, oob0, oob1, ChromaPost422());
    }
    // ---- output chroma low-pass :948-951 (full if "out", else lite if "lite")
    if (P.out_lp == 2) {
        chroma_lp_full422(R.U, W2, P.a_in_i, a_hp_i, 2);
        chroma_lp_full422(R.V, W2, P.ntsc ? P.a_in_q : P.a_in_i, P.ntsc ? a_hp_q : a_hp_i, P.ntsc ? 4 : 2);
    } else if (P.out_lp == 1) {
        chroma_lp_plain422(R.U, W2, P.a_tv, 1);
        chroma_lp_plain422(R.V, W2, P.a_tv, 1);
    }
    // ---- packed transposed scratch -> frame row (64-byte bursts per lane on aligned rows)
    if (is_out) {
        int x0 = 0;
        if (P.src_al16) {
            for (; x0 + 64 <= W; x0 += 64) {
                uint32_t w[16];
                const int q = x0 >> 2;
#pragma unroll
                for (int j = 0; j < 16; j++) w[j] = R.Y.word(q + j);
                uint4 *o = reinterpret_cast<uint4 *>(fy + x0);
                o[0] = make_uint4(w[0], w[1], w[2], w[3]);   o[1] = make_uint4(w[4], w[5], w[6], w[7]);
                o[2] = make_uint4(w[8], w[9], w[10], w[11]); o[3] = make_uint4(w[12], w[13], w[14], w[15]);
            }
            for (; x0 + 16 <= W; x0 += 16) {
                const int q = x0 >> 2;
                *reinterpret_cast<uint4 *>(fy + x0) =
                    make_uint4(R.Y.word(q), R.Y.word(q + 1), R.Y.word(q + 2), R.Y.word(q + 3));
            }
        }
        for (; x0 < W; x0++) fy[x0] = (uint8_t)R.Y.byte_at(x0);
        int c0 = 0;
        if (P.dst_al16) {
            for (; c0 + 32 <= W2; c0 += 32) {
                uint32_t a[8], b[8];
                const int q = c0 >> 2;
#pragma unroll
                for (int j = 0; j < 8; j++) { a[j] = R.U.word(q + j); b[j] = R.V.word(q + j); }
                uint2 *ou2 = reinterpret_cast<uint2 *>(fu + c0), *ov2 = reinterpret_cast<uint2 *>(fv + c0);
#pragma unroll
                for (int j = 0; j < 4; j++) { ou2[j] = make_uint2(a[2 * j], a[2 * j + 1]); ov2[j] = make_uint2(b[2 * j], b[2 * j + 1]); }
            }
            for (; c0 + 8 <= W2; c0 += 8) {
                const int q = c0 >> 2;
                *reinterpret_cast<uint2 *>(fu + c0) = make_uint2(R.U.word(q), R.U.word(q + 1));
                *reinterpret_cast<uint2 *>(fv + c0) = make_uint2(R.V.word(q), R.V.word(q + 1));
            }
        }
        for (; c0 < W2; c0++) { fu[c0] = (uint8_t)R.U.byte_at(c0); fv[c0] = (uint8_t)R.V.byte_at(c0); }
    }
}

} // namespace ntscsim
