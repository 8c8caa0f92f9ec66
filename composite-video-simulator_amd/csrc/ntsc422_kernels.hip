// ntsc422_kernels.hip -- the 8-bit YUV422P sibling of the field path:
// ffmpeg_to_composite.cpp composite_video_process() :629-952 (+ helpers :353-553),
// render_field() :1001-1129, black_key_feedback() :954-999.
//
// First-cut mapping: same execution model as ntsc_kernels.hip -- ONE LANE =
// ONE SCANLINE, 63 rows + 1 halo row per wavefront -- but the stage chain is NOT yet fused: every
// loop of the reference is one sweep of the lane over its row, on byte planes kept TRANSPOSED in
// HBM scratch (plane[x][slot], slot = wave*64 + lane) so that each wave access is one coalesced
// 64-byte line.  Every stage clamps to uint8 like the reference (clampu8 :335), so the sweeps
// communicate through bytes exactly as the reference's in-place frame does.
//
// The reference's Y/C separator reads Y[x+2] two bytes past the row (:496, undefined behaviour);
// here that read returns 16, the box filter's own pre-charge value (oracle: TOCOMP_OOB_DEFINED).
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
    uint8_t *Y, *T, *Cc, *U, *V;   // [W][S], [W][S], [W][S], [W/2][S], [W/2][S]
    size_t S;                      // slots = waves * 64
};

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

// ------------------------------------------------------------------------------ the field
struct Row422 {
    uint8_t *Y, *T, *Cc, *U, *V;
    size_t S;
    DEV uint8_t &y(int x) const { return Y[(size_t)x * S]; }
    DEV uint8_t &t(int x) const { return T[(size_t)x * S]; }
    DEV uint8_t &c(int x) const { return Cc[(size_t)x * S]; }
    DEV uint8_t &u(int x) const { return U[(size_t)x * S]; }
    DEV uint8_t &v(int x) const { return V[(size_t)x * S]; }
};

// composite_video_chroma_lowpass :353-393 (full) on one plane of one row
DEV void chroma_lp_full422(uint8_t *P0, size_t S, int W2, double a_lp, double a_hp, int delay)
{
    Lp3 lp; lp.reset(128);
    OnePole hp; hp.p = 128;
    for (int x = 0; x < W2; x++) {
        double s = P0[(size_t)x * S];
        s += hp.hp(s, a_hp);
        s = lp.push(s, a_lp);
        if (x >= delay) P0[(size_t)(x - delay) * S] = (uint8_t)clampu8((int)s);
    }
}
// composite_video_chroma_lowpass_lite :395-431 and the VHS chroma low-pass :834-855
DEV void chroma_lp_plain422(uint8_t *P0, size_t S, int W2, double a, int delay)
{
    Lp3 lp; lp.reset(128);
    for (int x = 0; x < W2; x++) {
        double s = P0[(size_t)x * S];
        s = lp.push(s, a);
        if (x >= delay) P0[(size_t)(x - delay) * S] = (uint8_t)clampu8((int)s);
    }
}
// composite_video_yuv_to_ntsc :434-477
DEV void modulate422(const DevParams &P, const Row422 &R, int W, unsigned xi, int amp, bool nocolor)
{
    for (int x = 0; x < W; x += 2) {
        const int cu = (int)R.u(x >> 1) - 128, cv = (int)R.v(x >> 1) - 128;
        for (int sx = 0; sx < 2 && x + sx < W; sx++) {
            const unsigned s = (xi + (unsigned)x + (unsigned)sx) & 3u;
            int chroma = ((s & 1u) ? cv : cu) * amp;
            if (s & 2u) chroma = -chroma;
            R.y(x + sx) = (uint8_t)clampu8((int)R.y(x + sx) + chroma / 50);
        }
        if (nocolor) { R.u(x >> 1) = 128; R.v(x >> 1) = 128; }
    }
}
// composite_ntsc_to_yuv :480-553 (out-of-row read = 16, out-of-array writes dropped)
DEV void demodulate422(const DevParams &P, const Row422 &R, int W, unsigned xi, const Magic31 &mA,
                       bool after_yc_sep)
{
    unsigned d0 = 16, d1 = 16, d2 = R.y(0), d3 = R.y(1);
    unsigned sum = 16 * 2 + d2 + d3;
    for (int x = 0; x < W; x++) {
        const unsigned c = (x + 2 < W) ? (unsigned)R.y(x + 2) : 16u;
        sum -= d0;
        d0 = d1; d1 = d2; d2 = d3; d3 = c;
        sum += c;
        const unsigned yb = (sum / 4u) & 0xFFu;
        R.y(x) = (uint8_t)yb;
        const int ch = clampu8((int)c + 128 - (int)yb);
        R.c(x) = (uint8_t)ch;
        if (after_yc_sep) { R.y(x) = (uint8_t)ch; R.u(x >> 1) = 128; R.v(x >> 1) = 128; }
    }
    if (after_yc_sep) return;
    for (int x = (int)((4u - xi) & 3u); x < W; x += 4) {
        if (x + 2 < W) R.c(x + 2) = (uint8_t)(255 - R.c(x + 2));
        if (x + 3 < W) R.c(x + 3) = (uint8_t)(255 - R.c(x + 3));
    }
    const int W2 = W / 2;
    for (int x = 0; x < W2; x++) {
        const int a = clampu8(sdivm(((int)R.c(2 * x) - 128) * 50, mA) + 128);
        const int b = clampu8(sdivm(((int)R.c(2 * x + 1) - 128) * 50, mA) + 128);
        if (xi & 1u) { R.u(x) = (uint8_t)(255 - b); R.v(x) = (uint8_t)(255 - a); }
        else         { R.u(x) = (uint8_t)(255 - a); R.v(x) = (uint8_t)(255 - b); }
    }
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
    R.S = Sc.S;
    R.Y = Sc.Y + slot; R.T = Sc.T + slot; R.Cc = Sc.Cc + slot; R.U = Sc.U + slot; R.V = Sc.V + slot;
    uint8_t *fy = fd.dst[0] + (size_t)fd.dst_ls[0] * y;
    uint8_t *fu = fd.dst[1] + (size_t)fd.dst_ls[1] * y;
    uint8_t *fv = fd.dst[2] + (size_t)fd.dst_ls[2] * y;

    // ---- frame row -> transposed scratch
    for (int x = 0; x < W; x++) R.y(x) = fy[x];
    for (int x = 0; x < W2; x++) { R.u(x) = fu[x]; R.v(x) = fv[x]; }

    // ---- input chroma low-pass :632
    if (P.in_lp) {
        chroma_lp_full422(R.U, R.S, W2, P.a_in_i, a_hp_i, P.ntsc ? 2 : 2);
        chroma_lp_full422(R.V, R.S, W2, P.ntsc ? P.a_in_q : P.a_in_i, P.ntsc ? a_hp_q : a_hp_i, P.ntsc ? 4 : 2);
    }
    // ---- modulate :633
    modulate422(P, R, W, xi, P.amp, P.nocolor != 0);
    // ---- pre-emphasis :636-651
    if (P.pre_on) {
        OnePole pre; pre.p = 16;
        for (int x = 0; x < W; x++) {
            double s = R.y(x);
            s += pre.hp(s, P.a_pre) * P.pre_gain;
            R.y(x) = (uint8_t)clampu8((int)s);
        }
    }
    // ---- luma noise :654-666
    if (P.noise_k) {
        LaneRand rng;
        rng.init(ring, rs_luma + rc, P.Rpad, lane);
        int noise = n0_luma[rc];
        for (int x = 0; x < W; x++) {
            R.y(x) = (uint8_t)clampu8((int)R.y(x) + noise);
            noise = sdiv2(noise + (int)umod31(rng.next(ring, lane), P.m_noise) - P.noise_k);
        }
    }
    // ---- head switching :669-732 (displaced copy, fill value 16)
    if (P.hs) {
        const int hs = hs_shift[rc];
        if (__any(hs != 0)) {
            const int tw = W + W / 10;
            for (int x = 0; x < W; x++) R.t(x) = R.y(x);
            for (int x = 0; x < W; x++) {
                int idx = x + hs;
                idx += (idx >> 31) & tw;
                idx -= (idx >= tw) ? tw : 0;
                const uint8_t v = R.t(idx < W ? idx : W - 1);
                if (hs != 0) R.y(x) = idx < W ? v : (uint8_t)16;
            }
        }
    }
    // ---- Y/C separation :734
    if (!P.nocolor) demodulate422(P, R, W, xi, P.m_amp_back, after_yc_sep != 0);
    // ---- chroma noise :738-754
    if (P.cnoise_k) {
        LaneRand rng;
        rng.init(ring, rs_chroma + rc, P.Rpad, lane);
        int nU = n0_u[rc], nV = n0_v[rc];
        for (int x = 0; x < W2; x++) {
            R.u(x) = (uint8_t)clampu8((int)R.u(x) + nU);
            R.v(x) = (uint8_t)clampu8((int)R.v(x) + nV);
            nU = sdiv2(nU + (int)umod31(rng.next(ring, lane), P.m_cnoise) - P.cnoise_k);
            nV = sdiv2(nV + (int)umod31(rng.next(ring, lane), P.m_cnoise) - P.cnoise_k);
        }
    }
    // ---- chroma phase noise :755-781 (u*cos - u*sin, v*cos + v*sin: not a rotation)
    if (P.pnoise_k) {
        int n = (rowok ? pn_noise[rc] : 0) + P.pnoise_k;
        n = n < 0 ? 0 : (n > 2 * P.pnoise_k ? 2 * P.pnoise_k : n);
        const double cosv = G.ptab[2 * n], sinv = G.ptab[2 * n + 1];
        for (int x = 0; x < W2; x++) {
            const double u = (int)R.u(x) - 128, v = (int)R.v(x) - 128;
            const double u_ = (u * cosv) - (u * sinv);
            const double v_ = (v * cosv) + (v * sinv);
            R.u(x) = (uint8_t)clampu8((int)(u_ + 128));
            R.v(x) = (uint8_t)clampu8((int)(v_ + 128));
        }
    }
    // ---- VHS block :786-930
    if (P.vhs) {
        {   // luma low-pass + emphasis :812-831
            Lp3 lp; lp.reset(16);
            OnePole pre; pre.p = 16;
            for (int x = 0; x < W; x++) {
                double s = R.y(x);
                s = lp.push(s, P.a_vl);
                s += pre.hp(s, P.a_vl) * 1.6;
                R.y(x) = (uint8_t)clampu8((int)s);
            }
        }
        chroma_lp_plain422(R.U, R.S, W2, P.a_vc, P.cdelay);     // :834-855
        chroma_lp_plain422(R.V, R.S, W2, P.a_vc, P.cdelay);
        if (P.vblend && P.ntsc) {                                // :862-882, delay line starts at 128
            for (int x = 0; x < W2; x++) {
                const int cU = R.u(x), cV = R.v(x);
                const int upU = __shfl_up(cU, 1), upV = __shfl_up(cV, 1);
                if (k >= 1) {
                    R.u(x) = (uint8_t)(((k >= 2 ? upU : 128) + cU + 1) >> 1);
                    R.v(x) = (uint8_t)(((k >= 2 ? upV : 128) + cV + 1) >> 1);
                }
            }
        }
        {   // luma sharpen :887-901
            Lp3 lp; lp.reset(16);
            for (int x = 0; x < W; x++) {
                const double s = R.y(x);
                const double ts = lp.push(s, P.a_sh);
                R.y(x) = (uint8_t)clampu8((int)(s + ((s - ts) * P.sharpen)));
            }
        }
        {   // chroma sharpen :904-924
            Lp3 lU, lV; lU.reset(128); lV.reset(128);
            for (int x = 0; x < W2; x++) {
                double s = R.u(x);
                double ts = lU.push(s, a_sh_c);
                R.u(x) = (uint8_t)clampu8((int)(s + ((s - ts) * sharpen_c)));
                s = R.v(x);
                ts = lV.push(s, a_sh_c);
                R.v(x) = (uint8_t)clampu8((int)(s + ((s - ts) * sharpen_c)));
            }
        }
        if (!P.svideo) {                                         // :926-929
            modulate422(P, R, W, xi, P.amp, P.nocolor != 0);
            demodulate422(P, R, W, xi, P.m_amp, after_yc_sep != 0);
        }
    }
    // ---- chroma dropout :932-942
    if (P.loss && dropout[rc]) {
        for (int x = 0; x < W2; x++) { R.u(x) = 128; R.v(x) = 128; }
    }
    // ---- extra Y/C recombine passes :943-946
    for (int i = 0; i < yc_recombine; i++) {
        modulate422(P, R, W, xi, P.amp, P.nocolor != 0);
        demodulate422(P, R, W, xi, P.m_amp, after_yc_sep != 0);
    }
    // ---- output chroma low-pass :948-951 (full if "out", else lite if "lite")
    if (P.out_lp == 2) {
        chroma_lp_full422(R.U, R.S, W2, P.a_in_i, a_hp_i, 2);
        chroma_lp_full422(R.V, R.S, W2, P.ntsc ? P.a_in_q : P.a_in_i, P.ntsc ? a_hp_q : a_hp_i, P.ntsc ? 4 : 2);
    } else if (P.out_lp == 1) {
        chroma_lp_plain422(R.U, R.S, W2, P.a_tv, 1);
        chroma_lp_plain422(R.V, R.S, W2, P.a_tv, 1);
    }
    // ---- transposed scratch -> frame row
    if (is_out) {
        for (int x = 0; x < W; x++) fy[x] = R.y(x);
        for (int x = 0; x < W2; x++) { fu[x] = R.u(x); fv[x] = R.v(x); }
    }
}

} // namespace ntscsim
