// Developer probe: cost of DEPENDENT fp64 chains on gfx950, the shape the NTSC filters have.
// One "pole step" is the reference's one-pole low-pass in its exact operation order
//     m = p*a;  s1 = s*a;  s2 = p - m;  p = s1 + s2
// (4 VALU instructions, dependency depth 3, and p feeds the next step).  NCH independent poles are
// interleaved instruction by instruction, so the distance between an instruction and its consumer
// is NCH issue slots.  Reported: shader cycles per VALU instruction per SIMD for 1..4 resident waves
// per SIMD -- i.e. how much instruction-level parallelism a wave needs before the fp64 pipe, not
// the dependency latency, sets the pace.
//   hipcc -O2 --offload-arch=gfx950 tools/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define STEPS 16      // pole steps per chain per unrolled body
#define ITERS 256

template <int NCH>
__global__ void k_chain(double *out, uint64_t *cyc, double a, double s)
{
    double p[NCH], m[NCH], s1[NCH], s2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) p[c] = 16.0 + c + threadIdx.x;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < STEPS; k++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(m[c]) : "v"(p[c]), "v"(a));
#pragma unroll
            for (int c = 0; c < NCH; c++) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(s1[c]) : "v"(s), "v"(a));
#pragma unroll
            for (int c = 0; c < NCH; c++) asm volatile("v_add_f64 %0, %1, -%2" : "=v"(s2[c]) : "v"(p[c]), "v"(m[c]));
#pragma unroll
            for (int c = 0; c < NCH; c++) asm volatile("v_add_f64 %0, %1, %2" : "=v"(p[c]) : "v"(s1[c]), "v"(s2[c]));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    double acc = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) acc += p[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// the same with 2-cycle-class integer instructions (v_add_u32 chains), for the int stages
template <int NCH>
__global__ void k_chain_i32(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t s)
{
    uint32_t p[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) p[c] = c + threadIdx.x;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < STEPS * 4; k++) {
#pragma unroll
            for (int c = 0; c < NCH; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(p[c]) : "v"(a));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t acc = s;
#pragma unroll
    for (int c = 0; c < NCH; c++) acc += p[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Exactly `wps` waves on every SIMD: one workgroup of 256 x wps threads per CU, forced by a
// 160 KiB dynamic-LDS request (see tools/valu_rate_probe.hip).
template <class T, class F>
static void run(const char *name, int nch, F kern, T a, T s, int wps)
{
    const int blocks = 256, threads = 256 * wps;
    const size_t lds = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    T *out; uint64_t *cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(T));
    (void)hipMalloc(&cyc, (size_t)blocks * (threads / 64) * sizeof(uint64_t));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, cyc, a, s);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, cyc, a, s);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h((size_t)blocks * (threads / 64));
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)STEPS * 4 * ITERS * nch;
    const double med = (double)h[h.size() / 2];
    printf("%-10s chains %d waves/SIMD %d: %6.2f cyc/instr/SIMD median wave, %6.2f slowest, %6.2f from the kernel time (%.3f ms); one wave: %6.2f cyc between its own instructions\n",
           name, nch, wps, med / n / wps, (double)h.back() / n / wps, ms * 1e-3 * 2.4e9 / n / wps, ms, med / n);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 3, 4}) {
        run<double>("pole_f64", 1, k_chain<1>, 0.1, 100.0, w);
        run<double>("pole_f64", 2, k_chain<2>, 0.1, 100.0, w);
        run<double>("pole_f64", 3, k_chain<3>, 0.1, 100.0, w);
        run<double>("pole_f64", 4, k_chain<4>, 0.1, 100.0, w);
        run<double>("pole_f64", 6, k_chain<6>, 0.1, 100.0, w);
        run<double>("pole_f64", 8, k_chain<8>, 0.1, 100.0, w);
        run<uint32_t>("add_u32", 1, k_chain_i32<1>, 3u, 5u, w);
        run<uint32_t>("add_u32", 2, k_chain_i32<2>, 3u, 5u, w);
        run<uint32_t>("add_u32", 4, k_chain_i32<4>, 3u, 5u, w);
    }
    return 0;
}
