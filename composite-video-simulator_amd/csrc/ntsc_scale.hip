// ntsc_scale.hip -- SURVEY 8(f) row f2, the INPUT side: what the tool does with libswscale before
// the field loop (sws_getContext(any format, any size -> BGRA W x H, SWS_BILINEAR), ffmpeg_ntsc.cpp
// :573-583, sws_scale :603; ffmpeg_to_composite.cpp:1773).  libswscale is a third-party library that
// is not part of the reference tree, so this is NOT a bit-clone of it (parity unpinned): it is the
// documented definition below, implemented identically by the test oracle (tests/_libs.py).
//
//   geometry   pixel centres aligned: source position of destination column x is
//              ((2x + 1) * sw) / (2W) - 1/2, clamped to [0, sw - 1], in 16.16 fixed point
//              pos = (((2x + 1) * sw) << 15) / W - 32768; x0 = pos >> 16, 8-bit weight (pos >> 8) & 255;
//              rows alike; chroma planes are resampled with the same formula on their own size
//   filter     bilinear, v = ((a*(256-fx) + b*fx) * (256-fy) + (c*(256-fx) + d*fx) * fy + 32768) >> 16
//   colour     BT.601 limited range, C = Y-16, D = U-128, E = V-128:
//              R = (298C + 409E + 128) >> 8, G = (298C - 100D - 208E + 128) >> 8, B = (298C + 516D + 128) >> 8,
//              clamped to 0..255; alpha = 255 (BGRA sources: alpha is resampled like a colour channel)
namespace ntscsim {

struct ScaleDev {
    const uint8_t *src[3];
    uint8_t *dst;
    int32_t src_ls[3];
    int32_t dst_ls;
    int32_t sw, sh, fmt;       // fmt: 0 BGRA, 1 YUV420P, 2 YUV422P
    int32_t _pad;
};

DEV void scale_pos(int x, int sn, int dn, int &i0, int &i1, int &f)
{
    long long pos = ((((long long)(2 * x + 1) * sn) << 15) / dn) - 32768;
    const long long hi = (long long)(sn - 1) << 16;
    pos = pos < 0 ? 0 : (pos > hi ? hi : pos);
    i0 = (int)(pos >> 16);
    f = (int)((pos >> 8) & 255);
    i1 = i0 + 1 < sn ? i0 + 1 : sn - 1;
}
DEV int bilerp(int a, int b, int c, int d, int fx, int fy)
{
    return ((a * (256 - fx) + b * fx) * (256 - fy) + (c * (256 - fx) + d * fx) * fy + 32768) >> 16;
}
DEV int plane_sample(const uint8_t *p, int ls, int pw, int ph, int x, int y, int W, int H)
{
    int x0, x1, fx, y0, y1, fy;
    scale_pos(x, pw, W, x0, x1, fx);
    scale_pos(y, ph, H, y0, y1, fy);
    const uint8_t *r0 = p + (size_t)ls * y0, *r1 = p + (size_t)ls * y1;
    return bilerp(r0[x0], r0[x1], r1[x0], r1[x1], fx, fy);
}
DEV int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one thread per destination pixel
__global__ void k_scale_to_bgra(const ScaleDev *__restrict__ descs, int W, int H)
{
    const ScaleDev &d = descs[blockIdx.z];
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    uint32_t px;
    if (d.fmt == 0) {
        int x0, x1, fx, y0, y1, fy;
        scale_pos(x, d.sw, W, x0, x1, fx);
        scale_pos(y, d.sh, H, y0, y1, fy);
        const uint8_t *r0 = d.src[0] + (size_t)d.src_ls[0] * y0, *r1 = d.src[0] + (size_t)d.src_ls[0] * y1;
        px = 0;
        for (int c = 0; c < 4; c++)
            px |= (uint32_t)bilerp(r0[4 * x0 + c], r0[4 * x1 + c], r1[4 * x0 + c], r1[4 * x1 + c], fx, fy) << (8 * c);
    } else {
        const int cw = (d.sw + 1) / 2, ch = d.fmt == 1 ? (d.sh + 1) / 2 : d.sh;
        const int Y = plane_sample(d.src[0], d.src_ls[0], d.sw, d.sh, x, y, W, H);
        const int U = plane_sample(d.src[1], d.src_ls[1], cw, ch, x, y, W, H);
        const int V = plane_sample(d.src[2], d.src_ls[2], cw, ch, x, y, W, H);
        const int C = Y - 16, D = U - 128, E = V - 128;
        // The shifted sums are laundered through an empty asm: ROCm 7.2's hipcc otherwise fuses
        // "(x >> 8) clamped to 0..255, two of them packed" into gfx950's v_ashr_pk_u8_i32, assumes
        // that instruction clears the upper half of its destination, and ORs the red byte on top --
        // but the hardware keeps the destination's old upper 16 bits, so red comes out as 255 (old
        // value negative) or off by one (only this kernel had the pattern).
        int tr = (298 * C + 409 * E + 128) >> 8, tg = (298 * C - 100 * D - 208 * E + 128) >> 8,
            tb = (298 * C + 516 * D + 128) >> 8;
        asm volatile("" : "+v"(tr), "+v"(tg), "+v"(tb));
        const int r = clamp255(tr), g = clamp255(tg), b = clamp255(tb);
        px = (uint32_t)b | ((uint32_t)g << 8) | ((uint32_t)r << 16) | 0xFF000000u;

    }
    *reinterpret_cast<uint32_t *>(d.dst + (size_t)d.dst_ls * y + 4 * (size_t)x) = px;
}

} // namespace ntscsim
