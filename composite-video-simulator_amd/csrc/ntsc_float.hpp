// ntsc_float.hpp -- host entry points of the all-float pipeline (NTSCSIM_MODE_FLOAT), a translation unit of its own
// (ntsc_float.hip: different floating-point flags than the exact kernels).
#pragma once
#include <hip/hip_runtime.h>
#include "ntsc_device.hpp"

namespace ntscsim {

// k_encode_fp: BGRA rows -> float composite plane comp[x][row] (bit patterns in the int plane of the exact mode)
void launch_encode_fp(hipStream_t st, const DevParams &D, const FieldDev *fields, const uint32_t *rs_luma, const int *n0_luma,
                      int *comp);
// k_decode_fp<vhs>: composite plane -> BGRA rows.  `variant`: developer switch (NTSCSIM_FP_VARIANT)
void launch_decode_fp(hipStream_t st, const DevParams &D, const GeomDev &G, const FieldDev *fields, const int *comp,
                      const uint32_t *rs_chroma, const int *n0_u, const int *n0_v, const int *hs_shift, const int *pn_noise,
                      const int *dropout, int *tails, int variant);

} // namespace ntscsim
