// Developer probe: what one follower step of raw28_decode.hip costs a LONE wavefront on gfx950 (one
// wavefront per CU, nothing else resident), from registers -- no memory, no LDS, no barrier.
//   A  slow block as the compiler emits it: level*om, lv*a, add            (3 instructions, 2 dependent)
//   B  products made ahead:                 level*om, add                  (2 instructions on the chain + 1 ahead)
//   C  the general step: compare, 2 v_cndmask, 1 - a, 2 multiplies, add    (7 instructions, 5 dependent)
//   hipcc -O2 --offload-arch=gfx950 tools/follow_probe.hip -o tools/bin/follow_probe && tools/bin/follow_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang fp contract(off)
#define ITERS 4096

template <int MODE>
__global__ void k_follow(double *out, double a_slow, double a_fast, double seed)
{
    double lv[16], level = seed + threadIdx.x * 1e-3;
#pragma unroll
    for (int j = 0; j < 16; j++) lv[j] = 100.0 + j + threadIdx.x;
    const double om = 1.0 - a_slow;
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 1) {
            double pr[16];
#pragma unroll
            for (int j = 0; j < 16; j++) pr[j] = lv[j] * a_slow;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; j++) level = (level * om) + pr[j];
        } else if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) level = (level * om) + (lv[j] * a_slow);
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const double a = level > lv[j] ? a_fast : a_slow;
                level = (level * (1.0 - a)) + (lv[j] * a);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("" : "+v"(lv[j]));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = level;
}

template <class F>
static void run(const char *name, F kern, int threads)
{
    const int blocks = 256;
    const size_t lds = 160 * 1024;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    double *out;
    (void)hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, 3.5e-6, 0.0105, 50.0);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, out, 3.5e-6, 0.0105, 50.0);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %d wave(s) per CU: %.3f ms = %.1f cycles per step at 2.4 GHz\n", name, threads / 64, ms, ms * 1e-3 * 2.4e9 / (16.0 * ITERS));
    (void)hipFree(out);
}

// where the wavefronts of a 320-thread workgroup (one per CU) go: HW_ID bits 5:4 = SIMD, 3:0 = wave slot
__global__ void k_where(unsigned *out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}

int main()
{
    {
        unsigned *o; (void)hipMalloc(&o, 256 * 5 * 4);
        (void)hipFuncSetAttribute((const void *)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 139264);
        hipLaunchKernelGGL(k_where, dim3(256), dim3(320), 139264, 0, o);
        unsigned h[20]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; b++) {
            printf("workgroup %d: SIMD of its five wavefronts:", b);
            for (int w = 0; w < 5; w++) printf(" %u", (h[b * 5 + w] >> 4) & 3);
            printf("  (CU %u)\n", (h[b * 5] >> 8) & 15);
        }
    }
    for (int t : {64, 256}) {
        run("A slow (mul, mul, add)", k_follow<0>, t);
        run("B slow, products ahead", k_follow<1>, t);
        run("C general (7 instructions)", k_follow<2>, t);
    }
    return 0;
}
