// development probe (round 6): what a READ-ONLY stream reaches on MI355X -- the GroupNorm statistics / backward-reduce passes sit at 3.9-4.4 TB/s while
// the passes that also write reach 5.2-5.3 TB/s (read + write bytes).  One or two input streams of 16-byte loads, U loads in flight per thread, grid-stride
// over a buffer far larger than the Infinity Cache; plain and nontemporal loads; and a copy for comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int U, int NS, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, size_t n, unsigned* out) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4_t v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
            if (NS == 2) w[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3]; if (NS == 2) acc += w[u][0] ^ w[u][3]; }
    }
    for (; i < n; i += stride) { acc += a[i][0]; if (NS == 2) acc += b[i][0]; }
    if (acc == 0x12345678u) out[0] = acc;
}
// contiguous chunk per block (what the GroupNorm stream kernels do: a block owns a pixel range)
template <int U, int NS>
__global__ __launch_bounds__(256) void read_chunk_kernel(const u32x4_t* __restrict__ a, const u32x4_t* __restrict__ b, size_t n, unsigned* out) {
    unsigned acc = 0;
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t s = (size_t)blockIdx.x * per, e = s + per < n ? s + per : n;
    size_t i = s + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        u32x4_t v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = a[i + u * 256]; if (NS == 2) w[u] = b[i + u * 256]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3]; if (NS == 2) acc += w[u][0] ^ w[u][3]; }
    }
    for (; i < e; i += 256) { acc += a[i][0]; if (NS == 2) acc += b[i][0]; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const u32x4_t* __restrict__ a, u32x4_t* __restrict__ c, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) c[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) c[i] = a[i];
}
#define TIME(NAME, BYTES, LAUNCH)                                                                        \
    {                                                                                                    \
        for (int r = 0; r < 3; ++r) { LAUNCH; }                                                          \
        (void)hipEventRecord(e0);                                                                              \
        for (int r = 0; r < 10; ++r) { LAUNCH; }                                                         \
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                                     \
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);                                                      \
        printf("%-56s %8.1f us  %6.2f TB/s\n", NAME, ms * 100.f, (double)(BYTES) / (ms * 1e-4) / 1e12);   \
    }
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;           // 1 GiB per stream
    u32x4_t *a, *b, *c; unsigned* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    if (hipMemset(a, 1, bytes) != hipSuccess || hipMemset(b, 2, bytes) != hipSuccess || hipMemset(c, 0, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    char nm[128];
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        snprintf(nm, 128, "read 1 stream  U=4  grid %d", g);  TIME(nm, bytes, hipLaunchKernelGGL((read_kernel<4, 1, false>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 1 stream  U=8  grid %d", g);  TIME(nm, bytes, hipLaunchKernelGGL((read_kernel<8, 1, false>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 1 stream  U=16 grid %d", g);  TIME(nm, bytes, hipLaunchKernelGGL((read_kernel<16, 1, false>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 1 stream  U=8  nt grid %d", g);  TIME(nm, bytes, hipLaunchKernelGGL((read_kernel<8, 1, true>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 2 streams U=4  grid %d", g);  TIME(nm, 2 * bytes, hipLaunchKernelGGL((read_kernel<4, 2, false>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 2 streams U=8  grid %d", g);  TIME(nm, 2 * bytes, hipLaunchKernelGGL((read_kernel<8, 2, false>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 2 streams U=4  chunked grid %d", g);  TIME(nm, 2 * bytes, hipLaunchKernelGGL((read_chunk_kernel<4, 2>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "read 2 streams U=8  chunked grid %d", g);  TIME(nm, 2 * bytes, hipLaunchKernelGGL((read_chunk_kernel<8, 2>), dim3(g), dim3(256), 0, 0, a, b, n, out))
        snprintf(nm, 128, "copy (read + write) U=4 grid %d", g);  TIME(nm, 2 * bytes, hipLaunchKernelGGL((copy_kernel<4>), dim3(g), dim3(256), 0, 0, a, c, n))
        snprintf(nm, 128, "copy (read + write) U=8 grid %d", g);  TIME(nm, 2 * bytes, hipLaunchKernelGGL((copy_kernel<8>), dim3(g), dim3(256), 0, 0, a, c, n))
    }
    return 0;
}
