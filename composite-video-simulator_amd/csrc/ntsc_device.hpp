// ntsc_device.hpp -- POD types shared by the host launcher and the HIP kernels.
#pragma once
#include <cstdint>
#include "glibc_rand.hpp"

namespace ntscsim {

// Uniform (kernel-argument) snapshot of ntscsim_params plus everything derived from it on the
// host: filter alphas (computed with the same expression as LowpassFilter::setFilter,
// ffmpeg_ntsc.cpp:78-86), division magics, geometry.
struct DevParams {
    int W, H;
    int Lslot;        // rows reserved per field slot = ceil(H/2)
    int nfields;
    int R;            // nfields * Lslot  (global scanline index space)
    int Rpad;         // row stride of the transposed planes, multiple of 64
    int ntsc;
    int phase_mode, phase_off;       // video_scanline_phase_shift(_offset)
    int in_lp;                       // composite_in_chroma_lowpass
    int out_lp;                      // 0 none, 1 lite (composite_lowpass_tv), 2 full
    int amp, amp_back;
    Magic31 m_amp, m_amp_back;
    int noise_k;  Magic31 m_noise;
    int cnoise_k; Magic31 m_cnoise;
    int pnoise_k; Magic31 m_pnoise;
    int loss;
    int hs, hs_noise_on;
    double hs_point, hs_phase, hs_pn;
    int nocolor;
    int vhs, vblend, svideo, cdelay;
    int pre_on;
    double pre_gain, a_pre;
    double a_in_i, a_in_q, a_tv, a_vl, a_vc, a_sh;
    double sharpen;                  // vhs_out_sharpen
    int src_al16, dst_al16;          // all src/dst rows 16-byte aligned
    int warm_luma, warm_chroma;      // warm-up draws used by k_row_states
    int variant;                     // 0 ffmpeg_ntsc (BGRA), 1 ffmpeg_to_composite (YUV422P)
    int ghost_taps, ghost_delay[4], ghost_gain[4];   // extension (not in the reference)
};

struct FieldDev {
    const uint8_t *src;
    uint8_t *dst;
    int32_t src_ls, dst_ls;
    uint32_t field, flags;
    uint64_t fieldno;
    uint32_t rng[61];   // rand() window at the field's first draw, extended by 30 words
    uint32_t _pad;
};
static_assert(sizeof(FieldDev) == 288, "FieldDev layout");

// per-geometry jump tables (device memory), see ntscsim_hip.hip: Geometry
struct GeomDev {
    const uint32_t *lskip;    // [2 par][31]  x^(draws before the head-switch draws)
    const uint32_t *pskip;    // [2 par][31]  x^(draws before the phase-noise / dropout draws)
    const uint32_t *jrow;     // [2 stream][2 par][Lslot][31]  x^(row start - warm-up)
    const int32_t  *jwarm;    // [2 stream][2 par][Lslot]      warm-up draws (== start -> exact)
    const uint32_t *sstart;   // [2 stream][2 par][31]         x^(stream start within the field)
    const double   *ptab;     // [2*pnoise_k+1][2]             cos,sin of noise*pi/100
};

} // namespace ntscsim
