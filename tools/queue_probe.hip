// queue_probe.hip -- do kernels of different HIP streams run side by side on this GPU / runtime?
// A kernel of `wgs` single-wave workgroups spins for ~`us` microseconds.  N streams get one launch each; if the N
// launches take the time of one, the streams' queues are served concurrently; if they take N times as long, they are not.
// Variants: plain launches; each launch preceded by a hipStreamWaitEvent on an event of another stream (what the submit
// engine does: uploads on one stream, kernels on a lane's stream); each launch preceded by a hipMemsetAsync and followed
// by a hipEventRecord.
// build: hipcc -O3 --offload-arch=gfx950 tools/queue_probe.hip -o tools/bin/queue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_spin(long long cycles, int *sink)
{
    const long long t0 = clock64();
    int v = 0;
    while (clock64() - t0 < cycles) v++;
    if (v == -1) sink[0] = v;
}

int main()
{
    int *sink = nullptr, *ms = nullptr;
    CHECK(hipMalloc((void **)&sink, 1 << 20));
    CHECK(hipMalloc((void **)&ms, 1 << 20));
    const long long cyc = 1000000;          // ~1 ms
    hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, 0, cyc, sink);
    CHECK(hipDeviceSynchronize());
    auto time_it = [&](int n, int mode, unsigned flags) -> double {
        std::vector<hipStream_t> st((size_t)n);
        for (auto &s : st) hipStreamCreateWithFlags(&s, flags);
        hipStream_t up; hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
        std::vector<hipEvent_t> ev((size_t)n), done((size_t)n);
        for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        for (auto &e : done) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            if (mode >= 1) { hipEventRecord(ev[(size_t)i], up); hipStreamWaitEvent(st[(size_t)i], ev[(size_t)i], 0); }
            if (mode >= 2) hipMemsetAsync(ms + 1024 * i, 0, 4096, st[(size_t)i]);
            hipLaunchKernelGGL(k_spin, dim3(128), dim3(64), 0, st[(size_t)i], cyc, sink);
            if (mode >= 2) hipEventRecord(done[(size_t)i], st[(size_t)i]);
        }
        hipDeviceSynchronize();
        const double ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (auto &s : st) hipStreamDestroy(s);
        hipStreamDestroy(up);
        for (auto &e : ev) hipEventDestroy(e);
        for (auto &e : done) hipEventDestroy(e);
        return ms_;
    };
    const double one = time_it(1, 0, hipStreamNonBlocking);
    std::printf("one launch of 128 wavefronts spinning: %.3f ms\n", one);
    const char *names[3] = {"plain launches", "+ hipStreamWaitEvent on another stream's event", "+ hipMemsetAsync before, hipEventRecord after"};
    for (int mode = 0; mode < 3; mode++)
        for (int n : {2, 3, 4, 8}) {
            const double t = time_it(n, mode, hipStreamNonBlocking);
            std::printf("%-55s %d streams: %.3f ms = %.2f x one launch\n", names[mode], n, t, t / one);
        }
    return 0;
}
