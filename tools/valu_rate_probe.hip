// Developer probe: throughput (shader cycles per wave64 instruction per SIMD) of the VALU
// instructions the NTSC kernels are made of, on gfx950.  Each kernel runs REPS x 8 independent
// copies of one instruction per wave, WAVES waves per SIMD; s_memtime brackets the loop.
//   hipcc -O2 --offload-arch=gfx950 tools/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REPS 64
#define ITERS 256

#define BODY8(ASM, C) \
    asm volatile(ASM : "+" C(r0) : C(a), C(b)); asm volatile(ASM : "+" C(r1) : C(a), C(b)); \
    asm volatile(ASM : "+" C(r2) : C(a), C(b)); asm volatile(ASM : "+" C(r3) : C(a), C(b)); \
    asm volatile(ASM : "+" C(r4) : C(a), C(b)); asm volatile(ASM : "+" C(r5) : C(a), C(b)); \
    asm volatile(ASM : "+" C(r6) : C(a), C(b)); asm volatile(ASM : "+" C(r7) : C(a), C(b));

#define KERNEL(NAME, T, ASM)                                                                    \
    __global__ void NAME(T *out, uint64_t *cyc, T a, T b)                                       \
    {                                                                                           \
        T r0 = a, r1 = b, r2 = a, r3 = b, r4 = a, r5 = b, r6 = a, r7 = b;                       \
        r0 += (T)threadIdx.x; r3 += (T)threadIdx.x;                                             \
        const uint64_t t0 = __builtin_readcyclecounter();                                      \
        for (int it = 0; it < ITERS; it++) {                                                    \
            _Pragma("unroll") for (int k = 0; k < REPS / 8; k++) { BODY8(ASM, "v") }           \
        }                                                                                       \
        const uint64_t t1 = __builtin_readcyclecounter();                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;     \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0; \
    }

KERNEL(k_add_f64, double, "v_add_f64 %0, %0, %1")
KERNEL(k_mul_f64, double, "v_mul_f64 %0, %0, %1")
KERNEL(k_fma_f64, double, "v_fma_f64 %0, %0, %1, %2")
KERNEL(k_trunc_f64, double, "v_trunc_f64 %0, %0")
KERNEL(k_add_f32, float, "v_add_f32 %0, %0, %1")
KERNEL(k_fma_f32, float, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_trunc_f32, float, "v_trunc_f32 %0, %0")
KERNEL(k_pk_fma_f32, double, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL(k_add_u32, uint32_t, "v_add_u32 %0, %0, %1")
KERNEL(k_add3_u32, uint32_t, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_and_b32, uint32_t, "v_and_b32 %0, %0, %1")
KERNEL(k_lshl_b32, uint32_t, "v_lshlrev_b32 %0, 3, %0")
KERNEL(k_ashr_i32, uint32_t, "v_ashrrev_i32 %0, 1, %0")
KERNEL(k_lshl_add, uint32_t, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(k_bfe_i32, uint32_t, "v_bfe_i32 %0, %0, 1, 30")
KERNEL(k_mov_b32, uint32_t, "v_mov_b32 %0, %1")
KERNEL(k_max_i32, uint32_t, "v_max_i32 %0, %0, %1")
KERNEL(k_med3_i32, uint32_t, "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_mul_lo_u32, uint32_t, "v_mul_lo_u32 %0, %0, %1")
KERNEL(k_mul_hi_u32, uint32_t, "v_mul_hi_u32 %0, %0, %1")
KERNEL(k_mul_u24, uint32_t, "v_mul_u32_u24 %0, %0, %1")
KERNEL(k_mad_u24, uint32_t, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_mad_i24, uint32_t, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_cvt_f32_i32, uint32_t, "v_cvt_f32_i32 %0, %0")
KERNEL(k_cvt_i32_f32, uint32_t, "v_cvt_i32_f32 %0, %0")
KERNEL(k_cndmask, uint32_t, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(k_cmp_cnd, uint32_t, "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(k_sub_co, uint32_t, "v_sub_co_u32 %0, vcc, %0, %1")
KERNEL(k_perm, uint32_t, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_ldexp_f64, double, "v_ldexp_f64 %0, %0, -8")
KERNEL(k_xor_b32, uint32_t, "v_xor_b32 %0, %0, %1")
KERNEL(k_sub_u32, uint32_t, "v_sub_u32 %0, %0, %1")
KERNEL(k_lshr_b32, uint32_t, "v_lshrrev_b32 %0, 1, %0")
KERNEL(k_lshl_or, uint32_t, "v_lshl_or_b32 %0, %0, 8, %1")
KERNEL(k_xad_u32, uint32_t, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_min_i32, uint32_t, "v_min_i32 %0, %0, %1")
KERNEL(k_dpp_wave_shr, uint32_t, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_cndmask_sgpr, uint32_t, "v_cndmask_b32 %0, %0, %1, s[10:11]")

// 64-bit <-> 32-bit conversions need mixed register widths: hand-written bodies
__global__ void k_cvt_f64_i32(double *out, uint64_t *cyc, int a, int b)
{
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    int s = a + threadIdx.x, t = b;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < REPS / 8; k++) {
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r0) : "v"(s)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r1) : "v"(t));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r2) : "v"(s)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r3) : "v"(t));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r4) : "v"(s)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r5) : "v"(t));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r6) : "v"(s)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r7) : "v"(t));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void k_cvt_i32_f64(int *out, uint64_t *cyc, double a, double b)
{
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    double s = a + threadIdx.x, t = b;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < REPS / 8; k++) {
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r0) : "v"(s)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r1) : "v"(t));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r2) : "v"(s)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r3) : "v"(t));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r4) : "v"(s)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r5) : "v"(t));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r6) : "v"(s)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r7) : "v"(t));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void k_mad_u64_u32(uint64_t *out, uint64_t *cyc, uint32_t a, uint32_t b)
{
    uint64_t r0 = 1, r1 = 2, r2 = 3, r3 = 4, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    uint32_t s = a + threadIdx.x, t = b;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < REPS / 8; k++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r0) : "v"(s), "v"(t) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r1) : "v"(s), "v"(t) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r2) : "v"(s), "v"(t) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r3) : "v"(s), "v"(t) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r4) : "v"(s), "v"(t) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r5) : "v"(s), "v"(t) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r6) : "v"(s), "v"(t) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r7) : "v"(s), "v"(t) : "vcc");
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
__global__ void k_ds_read_b32(uint32_t *out, uint64_t *cyc, uint32_t a, uint32_t b)
{
    __shared__ uint32_t lds[64 * 32 * 4];
    for (int i = threadIdx.x; i < 64 * 32 * 4; i += blockDim.x) lds[i] = i * a;
    __syncthreads();
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
    const uint32_t base = (threadIdx.x * 4u) & 8191u;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < REPS / 8; k++) {
            asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(r0) : "v"(base)); asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(r1) : "v"(base));
            asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(r2) : "v"(base)); asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(r3) : "v"(base));
            asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(r4) : "v"(base)); asm volatile("ds_read_b32 %0, %1 offset:1280" : "=v"(r5) : "v"(base));
            asm volatile("ds_read_b32 %0, %1 offset:1536" : "=v"(r6) : "v"(base)); asm volatile("ds_read_b32 %0, %1 offset:1792" : "=v"(r7) : "v"(base));
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + b;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// Exactly `waves_per_simd` waves on every SIMD: ONE workgroup of 256 x waves_per_simd threads per
// CU (wave i of a workgroup lands on SIMD i % 4), forced by a 160 KiB dynamic-LDS request that no
// second workgroup can fit beside.  Both the median wave (s_memtime) and the kernel wall time are
// reported; they agree when the placement is what it should be.
template <class T, class A, class F>
static void run(const char *name, F kern, A a, A b, int waves_per_simd, size_t lds = 160 * 1024)
{
    const int blocks = 256, threads = 256 * waves_per_simd;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    T *out; uint64_t *cyc;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(T));
    (void)hipMalloc(&cyc, (size_t)blocks * (threads / 64) * sizeof(uint64_t));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, cyc, a, b);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, cyc, a, b);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h((size_t)blocks * (threads / 64));
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)REPS * ITERS;
    const double med = (double)h[h.size() / 2], mx = (double)h.back();
    printf("%-16s waves/SIMD %d: %6.2f cyc/instr/SIMD median wave, %6.2f slowest wave, %6.2f from the kernel time (%.3f ms, 2.4 GHz)\n",
           name, waves_per_simd, med / n / waves_per_simd, mx / n / waves_per_simd,
           ms * 1e-3 * 2.4e9 / n / waves_per_simd, ms);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 3, 4}) {
#define R(K, T, A, x, y) run<T, A>(#K, K, (A)x, (A)y, w)
#define RL(K, T, A, x, y, L) run<T, A>(#K, K, (A)x, (A)y, w, L)
        R(k_add_f64, double, double, 1.000001, 0.999999);
        R(k_mul_f64, double, double, 1.000001, 0.999999);
        R(k_fma_f64, double, double, 1.000001, 0.999999);
        R(k_trunc_f64, double, double, 1.5, 2.5);
        R(k_cvt_f64_i32, double, int, 3, 5);
        R(k_cvt_i32_f64, int, double, 3.5, 5.5);
        R(k_add_f32, float, float, 1.000001f, 0.999999f);
        R(k_fma_f32, float, float, 1.000001f, 0.999999f);
        R(k_trunc_f32, float, float, 1.5f, 2.5f);
        R(k_pk_fma_f32, double, double, 1.000001, 0.999999);
        R(k_cvt_f32_i32, uint32_t, uint32_t, 3, 5);
        R(k_cvt_i32_f32, uint32_t, uint32_t, 3, 5);
        R(k_add_u32, uint32_t, uint32_t, 3, 5);
        R(k_add3_u32, uint32_t, uint32_t, 3, 5);
        R(k_and_b32, uint32_t, uint32_t, 3, 5);
        R(k_lshl_b32, uint32_t, uint32_t, 3, 5);
        R(k_ashr_i32, uint32_t, uint32_t, 3, 5);
        R(k_lshl_add, uint32_t, uint32_t, 3, 5);
        R(k_bfe_i32, uint32_t, uint32_t, 3, 5);
        R(k_mov_b32, uint32_t, uint32_t, 3, 5);
        R(k_max_i32, uint32_t, uint32_t, 3, 5);
        R(k_med3_i32, uint32_t, uint32_t, 3, 5);
        R(k_cndmask, uint32_t, uint32_t, 3, 5);
        R(k_cmp_cnd, uint32_t, uint32_t, 3, 5);
        R(k_sub_co, uint32_t, uint32_t, 3, 5);
        R(k_perm, uint32_t, uint32_t, 3, 5);
        R(k_ldexp_f64, double, double, 1.000001, 0.999999);
        R(k_xor_b32, uint32_t, uint32_t, 3, 5);
        R(k_sub_u32, uint32_t, uint32_t, 3, 5);
        R(k_lshr_b32, uint32_t, uint32_t, 3, 5);
        R(k_lshl_or, uint32_t, uint32_t, 3, 5);
        R(k_xad_u32, uint32_t, uint32_t, 3, 5);
        R(k_min_i32, uint32_t, uint32_t, 3, 5);
        R(k_dpp_wave_shr, uint32_t, uint32_t, 3, 5);
        R(k_cndmask_sgpr, uint32_t, uint32_t, 3, 5);
        R(k_mul_lo_u32, uint32_t, uint32_t, 3, 5);
        R(k_mul_hi_u32, uint32_t, uint32_t, 3, 5);
        R(k_mul_u24, uint32_t, uint32_t, 3, 5);
        R(k_mad_u24, uint32_t, uint32_t, 3, 5);
        R(k_mad_i24, uint32_t, uint32_t, 3, 5);
        R(k_mad_u64_u32, uint64_t, uint32_t, 3, 5);
        RL(k_ds_read_b32, uint32_t, uint32_t, 3, 5, 128 * 1024);
    }
    return 0;
}
