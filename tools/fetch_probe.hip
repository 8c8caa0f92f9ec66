// fetch_probe.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of this library
// (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Every kernel moves a KNOWN number of bytes out of / into a buffer far larger than the 256 MiB Infinity Cache:
//   k_read16_stream   16 B per lane, consecutive lanes consecutive addresses (1 KiB per wave access): the guide's case
//   k_read16_rows     16 B per lane, FOUR lanes per frame row (64 contiguous bytes), 16 rows per wave access, rows one
//                     frame line pair apart: k_encode_fast's CoopLoader
//   k_read4_stream    4 B per lane, 256 B per wave access: k_decode_fast's composite plane loads (comp[x][row])
//   k_write16_rows    16 B per lane, four lanes per row: k_decode_fast's cooperative pixel bursts
//   k_write4_stream   4 B per lane, 256 B per wave access: k_encode_fast's composite plane stores
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_probe.hip -o tools/bin/fetch_probe
// run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/bin/fetch_probe      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ void k_read16_stream(const v4u *__restrict__ in, unsigned *__restrict__ out, size_t n16)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = tid; i < n16; i += nt) { const v4u v = in[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[tid] = acc;
}
// frames of H rows x rowbytes; a wave takes 64 rows of one field (every second row) and walks along them in
// 64-byte pieces: lane = 4 * row_in_group + piece, 16 rows per access, 4 accesses per 64 rows
__global__ void k_read16_rows(const unsigned char *__restrict__ in, unsigned *__restrict__ out, int rowbytes, int rows_total)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    unsigned acc = 0;
    for (int x = 0; x + 64 <= rowbytes; x += 64)
        for (int g = 0; g < 4; g++) {
            const int row = wave * 64 + g * 16 + (lane >> 2);
            if (row < rows_total) {
                const v4u v = *reinterpret_cast<const v4u *>(in + (size_t)(2 * row) * rowbytes + x + (lane & 3) * 16);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    if (acc == 0x12345678u) out[wave * 64 + lane] = acc;
}
__global__ void k_read4_stream(const unsigned *__restrict__ in, unsigned *__restrict__ out, size_t n4)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (size_t i = tid; i < n4; i += nt) acc += in[i];
    if (acc == 0x12345678u) out[tid] = acc;
}
__global__ void k_write16_rows(unsigned char *__restrict__ outp, int rowbytes, int rows_total)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    for (int x = 0; x + 64 <= rowbytes; x += 64)
        for (int g = 0; g < 4; g++) {
            const int row = wave * 64 + g * 16 + (lane >> 2);
            if (row < rows_total)
                *reinterpret_cast<v4u *>(outp + (size_t)(2 * row) * rowbytes + x + (lane & 3) * 16) = v4u{(unsigned)row, (unsigned)x, 3u, 4u};
        }
}
__global__ void k_write4_stream(unsigned *__restrict__ outp, size_t n4)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n4; i += nt) outp[i] = (unsigned)i;
}

int main()
{
    const int rowbytes = 2880, H = 486;                 // 720 x 486 BGRA
    const int frames = 900;                             // 1.26 GB >> 256 MiB
    const size_t bytes = (size_t)frames * H * rowbytes;
    const int rows_total = frames * H / 2;              // one field of every frame
    unsigned char *buf = nullptr; unsigned *out = nullptr;
    CHECK(hipMalloc((void **)&buf, bytes + 4096));
    CHECK(hipMalloc((void **)&out, 64u << 20));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const size_t n16 = bytes / 16, n4 = bytes / 4;
    const size_t row_bytes_read = (size_t)rows_total * (rowbytes / 64 * 64);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_read16_stream, dim3(8192), dim3(256), 0, 0, (const v4u *)buf, out, n16);
        hipLaunchKernelGGL(k_read16_rows, dim3((rows_total + 63) / 64), dim3(64), 0, 0, buf, out, rowbytes, rows_total);
        hipLaunchKernelGGL(k_read4_stream, dim3(8192), dim3(256), 0, 0, (const unsigned *)buf, out, n4);
        hipLaunchKernelGGL(k_write16_rows, dim3((rows_total + 63) / 64), dim3(64), 0, 0, buf, rowbytes, rows_total);
        hipLaunchKernelGGL(k_write4_stream, dim3(8192), dim3(256), 0, 0, (unsigned *)buf, n4);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("known_bytes k_read16_stream %zu\nknown_bytes k_read16_rows %zu\nknown_bytes k_read4_stream %zu\n"
                "known_bytes k_write16_rows %zu\nknown_bytes k_write4_stream %zu\n", bytes, row_bytes_read, bytes, row_bytes_read, bytes);
    return 0;
}
