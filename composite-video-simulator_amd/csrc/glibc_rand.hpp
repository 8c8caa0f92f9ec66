// glibc_rand.hpp -- exact re-creation of glibc's default rand() stream with O(log n) jump-ahead.
//
// The reference draws all of its noise from libc rand() (ffmpeg_ntsc.cpp:1640,1655,1729,1731,
// 1744,1896), never seeded, i.e. glibc TYPE_3 (x^31 + x^3 + 1 additive feedback, stdlib/random_r.c)
// with seed 1.  The word sequence obeys  s[i] = s[i-31] + s[i-3]  (mod 2^32)  and the k-th rand()
// returns s[344+k] >> 1.  Because the recurrence is linear over Z/2^32, "advance by n" is
// multiplication by x^n in Z/2^32[x] / (x^31 - x^28 - 1); that makes every field -- and every
// scanline inside a field -- independently addressable, which is what lets the GPU run
// scanlines in parallel while reproducing the reference's serial stream bit for bit.
#pragma once
#include <cstdint>
#include <cstddef>

namespace ntscsim {

struct RandPoly { uint32_t c[31]; };   // sum c[k] x^k
struct RandState { uint32_t w[31]; };  // w[j] = s[313+pos+j]; next draw = (w[0]+w[28]) >> 1

RandPoly  rand_poly_one();
RandPoly  rand_poly_mul(const RandPoly &a, const RandPoly &b);
RandPoly  rand_poly_pow(uint64_t n);                 // x^n
RandState rand_state_origin();                       // position 0 (seed 1, after glibc's warm-up)
RandState rand_state_apply(const RandPoly &p, const RandState &s); // s advanced by the poly's n
RandState rand_state_at(uint64_t pos);

// sequential generator over a RandState (host side, used for small serial pieces and tests)
struct RandSeq {
    uint32_t r[31];
    int i;
    explicit RandSeq(const RandState &s) : i(0) { for (int j = 0; j < 31; j++) r[j] = s.w[j]; }
    uint32_t next()
    {
        int j = i + 28; if (j >= 31) j -= 31;
        uint32_t v = r[i] + r[j];
        r[i] = v;
        i = (i + 1 == 31) ? 0 : i + 1;
        return v >> 1;
    }
};

// unsigned division of a 31-bit value by a small constant via multiply-high:
// q = mulhi(n, M) >> sh  is exact for all n < 2^31 (Granlund-Montgomery, N = 31).
struct Magic31 { uint32_t mul; uint32_t shift; uint32_t div; };
Magic31 magic31(uint32_t d);

} // namespace ntscsim
