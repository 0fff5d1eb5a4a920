// Depth-edge annotation post-processing on device for gfx950.
//
// What infer_edge_estimation.py:191-206 / :240-256 does on the host to every predicted edge-probability map
// (4 scales per image over the whole KITTI training set, ~1 s of python loops per map) as four kernels:
//   prob = pred * scale ('/2')                                         infer_edge_estimation.py:192
//   5x5 Sobel (cv2.Sobel CV_64F ksize=5, BORDER_REFLECT_101) -> normals = uint8(((atan2(-sy,sx)*180/pi+180)/360)*255)   :194-200
//   non_max_suppression: 4-direction NMS on the quantised gradient angle, zero frame      packnet_sfm/utils/tools.py:9-46
//   hysteresis 0.3/0.7 + repeated raster 'DFS' to a fixed point                           packnet_sfm/utils/tools.py:49-92
// The Sobel follows OpenCV's published algorithm (getSobelKernels + sepFilter2D: rows first, taps accumulated left to
// right in double; symmetric column filter centre first then pairs, antisymmetric pairs outward); every product is exact
// in double and -ffp-contract=off keeps the summation order, so signs of zero and angle bins agree with the host.
// The reference's sweeps converge to "weak pixels 8-connected to a strong pixel through weak pixels"; here that is a
// tile-local fixed point in LDS iterated over global sweeps with a device-side change flag (a sweep that follows a
// sweep without changes returns immediately), so the host only has to look at one flag per batch of sweeps.
// HBM-bound byte work: 4 B read + 5 B written per pixel for Sobel/NMS, ~1 B per pixel and sweep for the propagation.
#include "common.hpp"
#include <algorithm>

namespace {

constexpr int TX = 64, TY = 4;            // Sobel/NMS output tile, one pixel per thread
constexpr int PW = TX + 4, PH = TY + 4;

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

__global__ __launch_bounds__(256) void sobel_nms_kernel(const float* __restrict__ pred, float scale, unsigned char* __restrict__ normals,
                                                        float* __restrict__ nms, int H, int W) {
    __shared__ float sp[PH * PW];
    const int b = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const float* img = pred + (long)b * H * W;
    for (int i = threadIdx.x; i < PH * PW; i += 256) {
        const int ly = i / PW, lx = i % PW;
        sp[i] = img[(long)reflect101(y0 + ly - 2, H) * W + reflect101(x0 + lx - 2, W)] * scale;
    }
    __syncthreads();
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= W || y >= H) return;
    const double D[5] = {-1.0, -2.0, 0.0, 2.0, 1.0}, S[5] = {1.0, 4.0, 6.0, 4.0, 1.0};
    double rd[5], rs[5];                  // row-filtered values of the five rows around y (derivative / smoothing taps)
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const float* row = &sp[(ly + r) * PW + lx];
        double ad = D[0] * (double)row[0], as = S[0] * (double)row[0];
#pragma unroll
        for (int j = 1; j < 5; ++j) { ad = ad + D[j] * (double)row[j]; as = as + S[j] * (double)row[j]; }
        rd[r] = ad; rs[r] = as;
    }
    double sx = S[2] * rd[2];
    sx = sx + S[3] * (rd[3] + rd[1]);
    sx = sx + S[4] * (rd[4] + rd[0]);
    double sy = D[3] * (rs[3] - rs[1]);
    sy = sy + D[4] * (rs[4] - rs[0]);
    const long o = ((long)b * H + y) * W + x;
    if (normals) {
        const double ang = atan2(-sy, sx);
        normals[o] = (unsigned char)(int)(((ang * (180.0 / 3.141592653589793) + 180.0) / 360.0) * 255.0);
    }
    if (nms) {
        float keep = 0.f;
        if (x >= 1 && y >= 1 && x < W - 1 && y < H - 1) {
            double a = atan2(sy, sx) * (180.0 / 3.141592653589793);
            if (a < 0.0) a += 180.0;
            const float* c = &sp[(ly + 2) * PW + lx + 2];
            float q = 1.f, r = 1.f;
            if ((0.0 <= a && a < 22.5) || (157.5 <= a && a <= 180.0)) { q = c[1]; r = c[-1]; }
            else if (22.5 <= a && a < 67.5) { q = c[-PW - 1]; r = c[PW + 1]; }
            else if (67.5 <= a && a < 112.5) { q = c[PW]; r = c[-PW]; }
            else if (112.5 <= a && a < 157.5) { q = c[PW - 1]; r = c[-PW + 1]; }
            if (c[0] >= q && c[0] >= r) keep = c[0];
        }
        nms[o] = keep;
    }
}

// ---- hysteresis -------------------------------------------------------------------------------------------------
// state: 0 none, 1 weak, 2 strong (interior); 3 frame pixel (keeps its value), 4 frame pixel whose value is exactly 2.0
// info[b]: [0] any strong, [1] order key of the frame maximum (0 = no frame pixel seen), [2] a NaN in the frame

__device__ __forceinline__ unsigned order_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ __launch_bounds__(256) void hyst_classify_kernel(const float* __restrict__ img, unsigned char* __restrict__ state,
                                                            unsigned* __restrict__ info, int H, int W, double t_low, double t_high) {
    const int b = blockIdx.y;
    const int n = H * W;                                       // < 2^31, checked by the caller
    bool strong = false, nan = false;
    unsigned fkey = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = i / W, x = i - y * W;
        const float v = img[(long)b * n + i];
        unsigned char s;
        if (x >= 1 && y >= 1 && x < W - 1 && y < H - 1) {
            s = (double)v > t_high ? 2 : (double)v < t_low ? 0 : 1;
            strong |= s == 2;
        } else {
            s = v == 2.0f ? 4 : 3;
            if (v != v) nan = true; else fkey = max(fkey, order_key(v));
        }
        state[(long)b * n + i] = s;
    }
    // one atomic per block and word: device-scope atomics on one address serialise at ~8 ns each across the 8 XCDs
    __shared__ unsigned skey[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fkey = max(fkey, (unsigned)__shfl_xor((int)fkey, o, 64));
    if ((threadIdx.x & 63) == 0) skey[threadIdx.x >> 6] = fkey;
    const int any_strong = __syncthreads_or(strong), any_nan = __syncthreads_or(nan);
    if (threadIdx.x == 0) {
        if (any_strong) atomicOr(&info[b * 4 + 0], 1u);
        if (any_nan) atomicOr(&info[b * 4 + 2], 1u);
        const unsigned k = max(max(skey[0], skey[1]), max(skey[2], skey[3]));
        if (k) atomicMax(&info[b * 4 + 1], k);
    }
}

__device__ __forceinline__ bool is_strong(unsigned char v) { return v == 2 || v == 4; }

constexpr int HX = 64, HY = 16;           // propagation tile (4 pixels per thread) + 1 halo
constexpr int HW_ = HX + 2, HH_ = HY + 2;

__global__ __launch_bounds__(256) void hyst_propagate_kernel(unsigned char* __restrict__ state, int* __restrict__ flags, int sweep, int H, int W) {
    if (flags[sweep - 1] == 0) return;                         // the previous sweep changed nothing: fixed point reached
    __shared__ unsigned char st[HH_ * HW_];
    const int b = blockIdx.z, x0 = blockIdx.x * HX, y0 = blockIdx.y * HY;
    unsigned char* s = state + (long)b * H * W;
    for (int i = threadIdx.x; i < HH_ * HW_; i += 256) {
        const int ly = i / HW_, lx = i % HW_;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        st[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? s[(long)gy * W + gx] : 0;
    }
    __syncthreads();
    const int lx = threadIdx.x % HX, ly0 = (threadIdx.x / HX) * 4;
    bool any_change = false;
    for (;;) {
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned char* c = &st[(ly0 + k + 1) * HW_ + lx + 1];
            if (*c == 1) {
                const bool near = is_strong(c[-HW_ - 1]) || is_strong(c[-HW_]) || is_strong(c[-HW_ + 1]) || is_strong(c[-1]) ||
                                  is_strong(c[1]) || is_strong(c[HW_ - 1]) || is_strong(c[HW_]) || is_strong(c[HW_ + 1]);
                if (near) { *c = 2; changed = true; }
            }
        }
        any_change |= changed;
        if (!__syncthreads_or(changed)) break;
    }
    if (any_change) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gy = y0 + ly0 + k, gx = x0 + lx;
            if (gy < H && gx < W) {
                const unsigned char v = st[(ly0 + k + 1) * HW_ + lx + 1];
                if (v == 2) s[(long)gy * W + gx] = 2;
            }
        }
    }
    if (__syncthreads_or(any_change) && threadIdx.x == 0) atomicOr(&flags[sweep], 1);      // one atomic per block (see classify)
}

__global__ __launch_bounds__(256) void hyst_finish_kernel(const float* __restrict__ img, const unsigned char* __restrict__ state,
                                                          const unsigned* __restrict__ info, float* __restrict__ out, int H, int W) {
    const int b = blockIdx.y;
    const int n = H * W;
    const bool interior = H > 2 && W > 2;
    double maxv = -__builtin_huge_val();
    if (info[b * 4 + 1]) maxv = (double)key_to_float(info[b * 4 + 1]);
    if (interior) maxv = fmax(maxv, info[b * 4 + 0] ? 2.0 : 0.0);
    if (info[b * 4 + 2]) maxv = __builtin_nan("");
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = img[(long)b * n + i];
        const unsigned char s = state[(long)b * n + i];
        const double t = s >= 3 ? (double)v : s == 2 ? 2.0 : 0.0;
        out[(long)b * n + i] = (float)((double)v * (t / maxv));
    }
}

}  // namespace

extern "C" int mte_dee_sobel_nms(const float* pred, float scale, unsigned char* normals_u8, float* nms, int B, int H, int W, hipStream_t stream) {
    if (!pred || (!normals_u8 && !nms) || B <= 0 || H <= 0 || W <= 0) return MTE_ERR_ARG;
    hipLaunchKernelGGL(sobel_nms_kernel, dim3(cdiv(W, TX), cdiv(H, TY), B), dim3(256), 0, stream, pred, scale, normals_u8, nms, H, W);
    return mte_check_launch();
}

extern "C" int mte_hysteresis_begin(const float* img, unsigned char* state, int* info, int B, int H, int W, double t_low, double t_high,
                                    hipStream_t stream) {
    if (!img || !state || !info || B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 30)) return MTE_ERR_ARG;
    if (mte_memset_async(info, 0, (size_t)B * 4 * sizeof(int), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    const int bx = std::min(cdiv((long)H * W, 256 * 8), 512);
    hipLaunchKernelGGL(hyst_classify_kernel, dim3(bx, B), dim3(256), 0, stream, img, state, (unsigned*)info, H, W, t_low, t_high);
    return mte_check_launch();
}

extern "C" int mte_hysteresis_propagate(unsigned char* state, int* flags, int sweeps, int B, int H, int W, hipStream_t stream) {
    if (!state || !flags || sweeps <= 0 || B <= 0 || H <= 0 || W <= 0) return MTE_ERR_ARG;
    if (mte_memset_async(flags, 0, (size_t)(sweeps + 1) * sizeof(int), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    if (mte_memset_async(flags, 1, 1, stream) != hipSuccess) return MTE_ERR_LAUNCH;             // flags[0] != 0: the first sweep always runs
    for (int k = 1; k <= sweeps; ++k)
        hipLaunchKernelGGL(hyst_propagate_kernel, dim3(cdiv(W, HX), cdiv(H, HY), B), dim3(256), 0, stream, state, flags, k, H, W);
    return mte_check_launch();
}

extern "C" int mte_hysteresis_finish(const float* img, const unsigned char* state, const int* info, float* out, int B, int H, int W,
                                     hipStream_t stream) {
    if (!img || !state || !info || !out || B <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1L << 30)) return MTE_ERR_ARG;
    const int bx = std::min(cdiv((long)H * W, 256 * 4), 2048);
    hipLaunchKernelGGL(hyst_finish_kernel, dim3(bx, B), dim3(256), 0, stream, img, state, (const unsigned*)info, out, H, W);
    return mte_check_launch();
}
