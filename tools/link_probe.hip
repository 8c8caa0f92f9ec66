// link_probe.hip -- host link rates for the transfer shapes of the submit engine (csrc/ntscsim_submit.hip):
// 1.4 MB frames, host memory either hipHostMalloc'ed or posix_memalign'ed + hipHostRegister'ed in place.
//   H2D: hipMemcpyAsync per frame          D2H: hipMemcpyAsync per frame | a kernel storing into the mapped host frame
//   both directions at once
// build: hipcc -O3 --offload-arch=gfx950 tools/link_probe.hip -o tools/bin/link_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__global__ void k_store(v4u *__restrict__ dst, const v4u *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main()
{
    const size_t FB = 720 * 486 * 4, NF = 64;
    unsigned char *dev = nullptr;
    CHECK(hipMalloc((void **)&dev, FB * NF * 2));
    hipStream_t up, dn;
    CHECK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&dn, hipStreamNonBlocking));
    for (int kind = 0; kind < 2; kind++) {
        std::vector<unsigned char *> h((size_t)NF * 2), hd((size_t)NF * 2);
        for (size_t i = 0; i < NF * 2; i++) {
            if (kind == 0) { CHECK(hipHostMalloc((void **)&h[i], FB, hipHostMallocDefault)); hd[i] = h[i]; }
            else {
                void *p = nullptr;
                if (posix_memalign(&p, 4096, (FB + 4095) & ~(size_t)4095)) return 1;
                h[i] = (unsigned char *)p;
                for (size_t q = 0; q < FB; q += 4096) h[i][q] = 1;
                CHECK(hipHostRegister(p, (FB + 4095) & ~(size_t)4095, hipHostRegisterDefault));
                void *d = nullptr; CHECK(hipHostGetDevicePointer(&d, p, 0)); hd[i] = (unsigned char *)d;
            }
        }
        auto run = [&](int mode) -> double {     // 0 H2D, 1 D2H memcpy, 2 D2H kernel, 3 H2D + D2H memcpy, 4 H2D + D2H kernel
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 4; rep++)
                for (size_t i = 0; i < NF; i++) {
                    if (mode == 0 || mode >= 3) hipMemcpyAsync(dev + FB * i, h[i], FB, hipMemcpyHostToDevice, up);
                    if (mode == 1 || mode == 3) hipMemcpyAsync(h[NF + i], dev + FB * (NF + i), FB, hipMemcpyDeviceToHost, dn);
                    if (mode == 2 || mode == 4) hipLaunchKernelGGL(k_store, dim3(64), dim3(256), 0, dn, (v4u *)hd[NF + i], (const v4u *)(dev + FB * (NF + i)), FB / 16);
                }
            hipDeviceSynchronize();
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        };
        run(0); run(1);
        const char *names[5] = {"H2D memcpy", "D2H memcpy", "D2H kernel stores", "H2D + D2H memcpy", "H2D memcpy + D2H kernel stores"};
        for (int mode = 0; mode < 5; mode++) {
            const double t = run(mode);
            const double gb = 4.0 * NF * FB / 1e9;
            std::printf("%-22s %-32s %.2f GB/s per direction (%.0f frames/s)\n", kind ? "registered in place" : "hipHostMalloc", names[mode], gb / t, 4.0 * NF / t);
        }
    }
    return 0;
}
