// glibc_rand.cpp -- see glibc_rand.hpp.  Algorithm source: glibc 2.35 stdlib/random_r.c
// (__srandom_r / __random_r, TYPE_3); nothing of it is copied, the recurrence is restated.
#include "glibc_rand.hpp"
#include "ntscsim.h"

#include <cstring>
#include <mutex>

namespace ntscsim {

RandPoly rand_poly_one()
{
    RandPoly p;
    std::memset(&p, 0, sizeof(p));
    p.c[0] = 1;
    return p;
}

// (a*b) mod (x^31 - x^28 - 1), coefficients mod 2^32
RandPoly rand_poly_mul(const RandPoly &a, const RandPoly &b)
{
    uint32_t t[61];
    std::memset(t, 0, sizeof(t));
    for (int i = 0; i < 31; i++) {
        const uint32_t ai = a.c[i];
        if (!ai) continue;
        for (int j = 0; j < 31; j++) t[i + j] += ai * b.c[j];
    }
    // x^d = x^(d-3) + x^(d-31) for d >= 31; fold from the top so every touched term is lower
    for (int d = 60; d >= 31; d--) {
        const uint32_t v = t[d];
        t[d - 3] += v;
        t[d - 31] += v;
    }
    RandPoly r;
    std::memcpy(r.c, t, sizeof(r.c));
    return r;
}

namespace {
struct Pow2Table {
    RandPoly p[64];   // p[k] = x^(2^k)
    Pow2Table()
    {
        std::memset(&p[0], 0, sizeof(RandPoly));
        p[0].c[1] = 1;
        for (int k = 1; k < 64; k++) p[k] = rand_poly_mul(p[k - 1], p[k - 1]);
    }
};
const Pow2Table &pow2_table()
{
    static const Pow2Table t;   // thread-safe static init
    return t;
}
} // namespace

RandPoly rand_poly_pow(uint64_t n)
{
    const Pow2Table &t = pow2_table();
    RandPoly r = rand_poly_one();
    for (int k = 0; n; k++, n >>= 1)
        if (n & 1) r = rand_poly_mul(r, t.p[k]);
    return r;
}

RandState rand_state_origin()
{
    // __srandom_r(seed = 1): 31 words from the minimal-standard LCG (Schrage form), then the
    // generator is clocked 310 times; the first rand() is word 344 of the linear sequence.
    uint32_t s[344];
    int32_t word = 1;
    s[0] = 1;
    for (int i = 1; i < 31; i++) {
        const int32_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        s[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; i++) s[i] = s[i - 31];
    for (int i = 34; i < 344; i++) s[i] = s[i - 31] + s[i - 3];
    RandState st;
    for (int j = 0; j < 31; j++) st.w[j] = s[313 + j];
    return st;
}

RandState rand_state_apply(const RandPoly &p, const RandState &s)
{
    // extend the window to 61 words, then out[j] = sum_k c[k] * w[j+k]
    uint32_t w[61];
    std::memcpy(w, s.w, sizeof(s.w));
    for (int i = 31; i < 61; i++) w[i] = w[i - 31] + w[i - 3];
    RandState o;
    for (int j = 0; j < 31; j++) {
        uint32_t acc = 0;
        for (int k = 0; k < 31; k++) acc += p.c[k] * w[j + k];
        o.w[j] = acc;
    }
    return o;
}

RandState rand_state_at(uint64_t pos)
{
    static const RandState origin = rand_state_origin();
    if (pos == 0) return origin;
    return rand_state_apply(rand_poly_pow(pos), origin);
}

Magic31 magic31(uint32_t d)
{
    Magic31 m;
    m.div = d;
    if (d <= 1) { m.mul = 0; m.shift = 0; return m; }   // mul == 0: sdivm() returns n (d == 1)
    uint32_t l = 0;
    while ((1ull << l) < d) l++;                          // l = ceil(log2 d)
    const unsigned __int128 num = (unsigned __int128)1 << (31 + l);
    const uint64_t M = (uint64_t)((num + d - 1) / d);     // < 2^32 because 2^l < 2d
    m.mul = (uint32_t)M;
    m.shift = l - 1;                                      // total shift 31+l = 32 (mulhi) + l-1
    return m;
}

} // namespace ntscsim

extern "C" void ntscsim_rng_draw(uint64_t pos, size_t n, uint32_t *out)
{
    if (!out) return;
    ntscsim::RandSeq g(ntscsim::rand_state_at(pos));
    for (size_t i = 0; i < n; i++) out[i] = g.next();
}
