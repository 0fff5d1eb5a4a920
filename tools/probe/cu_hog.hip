// development probe (round 5, tools/overlap_probe.py): a kernel that only OCCUPIES compute units the way the MFMA weight-gradient kernels do -- 512 threads,
// a large LDS allocation, ~230 VGPRs per wave (two waves per SIMD take the whole register file), matrix-core instructions back to back, no memory traffic.
// What a memory-bound kernel on another queue loses beside it is lost to occupancy alone; beside the real weight-gradient kernel the difference is traffic.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <bool SLEEP>
__global__ __launch_bounds__(512) void hog_kernel(int iters, float* out) {
    extern __shared__ char smem[];
    f32x4_t acc[56];
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)(float)(threadIdx.x & 7); }
    for (int it = 0; it < iters; ++it) {
        if constexpr (SLEEP) {                       // the same footprint, the matrix pipe idle: 56 x 16 cycles asleep
#pragma unroll
            for (int i = 0; i < 7; ++i) __builtin_amdgcn_s_sleep(2);
#pragma unroll
            for (int i = 0; i < 56; ++i) asm volatile("" : "+v"(acc[i]));      // (the registers stay allocated)
        } else {
#pragma unroll
            for (int i = 0; i < 56; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 56; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) { smem[threadIdx.x] = 1; out[blockIdx.x] = s + smem[(threadIdx.x + 1) & 511]; }
}

extern "C" int hog_launch(int wgs, int lds_bytes, int iters, float* out, hipStream_t st, int sleep) {
    static int cfg = 0;
    if (!cfg) {
        if (hipFuncSetAttribute((const void*)hog_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)hog_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        cfg = 1;
    }
    if (sleep) hipLaunchKernelGGL(hog_kernel<true>, dim3(wgs), dim3(512), lds_bytes, st, iters, out);
    else hipLaunchKernelGGL(hog_kernel<false>, dim3(wgs), dim3(512), lds_bytes, st, iters, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
