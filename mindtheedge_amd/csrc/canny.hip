// Canny step of the validation edge metrics on device for gfx950.  PARITY UNPINNED: OpenCV is not
// available where this was written; the kernels implement OpenCV's published algorithm (imgproc/src/canny.cpp,
// apertureSize 3, L1 gradient) as restated in oracle/canny_oracle.py and are tested against that restatement only.
//
// ModelWrapper.compute_edge_metrics (packnet_sfm/models/model_wrapper.py:376-400):
//   vis = uint8(depth * (255 / max(depth)));  cv2.Canny(vis, 10, 20), (20, 40), (30, 60)
// One pass computes the image maximum, one LDS-tiled pass does uint8 conversion + 3x3 Sobel (replicated border) + L1
// magnitude + sector non-maximum suppression ONCE and classifies the pixel for up to 4 threshold pairs (the pairs differ
// only in the thresholds), the hysteresis is the propagation kernel of dee_post.hip (weak -> strong to a fixed point),
// and a last pass writes 255 / 0.  Integer work on one byte per pixel and map; HBM-bound.
#include "common.hpp"

extern "C" int mte_hysteresis_propagate(unsigned char* state, int* flags, int sweeps, int B, int H, int W, hipStream_t stream);

namespace {

constexpr int TX = 64, TY = 4;
constexpr int UW = TX + 4, UH = TY + 4;       // uint8 tile with a 2-pixel halo
constexpr int MW = TX + 2, MH = TY + 2;       // magnitude tile with a 1-pixel halo
constexpr int TG22 = 13573;                   // round(tan(22.5 deg) * 2^15)

struct Pairs { int n; int low[4]; int high[4]; };

__global__ __launch_bounds__(256) void image_max_kernel(const float* __restrict__ depth, unsigned* __restrict__ maxbits, int n) {
    const int b = blockIdx.y;
    float m = 0.f;                                                             // depths are positive
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, depth[(long)b * n + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&maxbits[b], __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}

// state[pair][b][H][W]: 0 not an edge, 1 candidate, 2 edge
__global__ __launch_bounds__(256) void canny_nms_kernel(const float* __restrict__ depth, const unsigned* __restrict__ maxbits,
                                                        unsigned char* __restrict__ vis, unsigned char* __restrict__ state,
                                                        int B, int H, int W, Pairs pr) {
    __shared__ int su[UH * UW];
    __shared__ int sdx[MH * MW], sdy[MH * MW], smag[MH * MW];
    const int b = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const float factor = 255.0f / __uint_as_float(maxbits[b]);
    const float* img = depth + (long)b * H * W;
    for (int i = threadIdx.x; i < UH * UW; i += 256) {
        const int ly = i / UW, lx = i % UW;
        const int gy = min(max(y0 + ly - 2, 0), H - 1), gx = min(max(x0 + lx - 2, 0), W - 1);      // BORDER_REPLICATE
        su[i] = (int)(unsigned char)(int)(img[(long)gy * W + gx] * factor);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MH * MW; i += 256) {
        const int ly = i / MW, lx = i % MW;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        int dx = 0, dy = 0, mag = 0;
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            const int* c = &su[(ly + 1) * UW + lx + 1];
            dx = (c[-UW + 1] + 2 * c[1] + c[UW + 1]) - (c[-UW - 1] + 2 * c[-1] + c[UW - 1]);
            dy = (c[UW - 1] + 2 * c[UW] + c[UW + 1]) - (c[-UW - 1] + 2 * c[-UW] + c[-UW + 1]);
            mag = abs(dx) + abs(dy);
        }
        sdx[i] = dx; sdy[i] = dy; smag[i] = mag;                                                   // magnitude outside the image = 0
    }
    __syncthreads();
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= W || y >= H) return;
    const int ci = (ly + 1) * MW + lx + 1;
    const int dx = sdx[ci], dy = sdy[ci], m = smag[ci];
    const long ax = abs(dx), ay = (long)abs(dy) << 15;
    const long tg22x = ax * TG22, tg67x = tg22x + (ax << 16);
    bool is_max;
    if (ay < tg22x) is_max = m > smag[ci - 1] && m >= smag[ci + 1];
    else if (ay > tg67x) is_max = m > smag[ci - MW] && m >= smag[ci + MW];
    else {
        const int s = ((dx ^ dy) < 0) ? -1 : 1;
        is_max = m > smag[ci - MW - s] && m > smag[ci + MW + s];
    }
    const long o = (long)y * W + x;
    if (vis) vis[(long)b * H * W + o] = (unsigned char)su[(ly + 2) * UW + lx + 2];
    for (int p = 0; p < pr.n; ++p)
        state[((long)p * B + b) * H * W + o] = (is_max && m > pr.low[p]) ? (m > pr.high[p] ? 2 : 1) : 0;
}

// cv2.resize(src, (W, H), interpolation=INTER_LINEAR) on float32 (model_wrapper.py:386-387), restated from OpenCV's resize.cpp:
// half-pixel centres, source index clamped with the fraction zeroed at the borders, rows interpolated first, then columns
__global__ __launch_bounds__(256) void resize_linear_kernel(const float* __restrict__ src, float* __restrict__ dst, int h, int w, int H, int W) {
    const int b = blockIdx.y;
    const long n = (long)H * W;
    const float sx_scale = (float)((double)w / W), sy_scale = (float)((double)h / H);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int dy = (int)(i / W), dx = (int)(i - (long)dy * W);
        float fx = (float)((dx + 0.5) * (double)sx_scale - 0.5), fy = (float)((dy + 0.5) * (double)sy_scale - 0.5);
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        fx -= (float)sx; fy -= (float)sy;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= w - 1) { sx = w - 1; fx = 0.f; }
        if (sy < 0) { sy = 0; fy = 0.f; }
        if (sy >= h - 1) { sy = h - 1; fy = 0.f; }
        const int sx1 = min(sx + 1, w - 1), sy1 = min(sy + 1, h - 1);
        const float* p = src + (long)b * h * w;
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float r0 = p[(long)sy * w + sx] * a0 + p[(long)sy * w + sx1] * a1;
        const float r1 = p[(long)sy1 * w + sx] * a0 + p[(long)sy1 * w + sx1] * a1;
        dst[(long)b * n + i] = r0 * b0 + r1 * b1;
    }
}

__global__ __launch_bounds__(256) void canny_finish_kernel(const unsigned char* __restrict__ state, float* __restrict__ edges, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) edges[i] = state[i] == 2 ? 255.f : 0.f;
}

}  // namespace

extern "C" int mte_canny_begin(const float* depth, int B, int H, int W, int n_pairs, const int* thresholds, unsigned* max_ws,
                               unsigned char* vis_u8, unsigned char* state, hipStream_t stream) {
    if (!depth || !thresholds || !max_ws || !state || B <= 0 || H <= 0 || W <= 0 || n_pairs < 1 || n_pairs > 4 || (long)H * W >= (1L << 30))
        return MTE_ERR_ARG;
    Pairs pr{};
    pr.n = n_pairs;
    for (int p = 0; p < n_pairs; ++p) {
        pr.low[p] = thresholds[2 * p] < thresholds[2 * p + 1] ? thresholds[2 * p] : thresholds[2 * p + 1];
        pr.high[p] = thresholds[2 * p] < thresholds[2 * p + 1] ? thresholds[2 * p + 1] : thresholds[2 * p];
    }
    if (mte_memset_async(max_ws, 0, sizeof(unsigned) * B, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    int bx = cdiv((long)H * W, 256 * 8); if (bx > 256) bx = 256;
    hipLaunchKernelGGL(image_max_kernel, dim3(bx, B), dim3(256), 0, stream, depth, max_ws, H * W);
    hipLaunchKernelGGL(canny_nms_kernel, dim3(cdiv(W, TX), cdiv(H, TY), B), dim3(256), 0, stream, depth, max_ws, vis_u8, state, B, H, W, pr);
    return mte_check_launch();
}

extern "C" int mte_resize_linear(const float* src, int B, int h, int w, float* dst, int H, int W, hipStream_t stream) {
    if (!src || !dst || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return MTE_ERR_ARG;
    long g = ((long)H * W + 256 * 4 - 1) / (256 * 4); if (g > 1024) g = 1024;
    hipLaunchKernelGGL(resize_linear_kernel, dim3((unsigned)g, B), dim3(256), 0, stream, src, dst, h, w, H, W);
    return mte_check_launch();
}

extern "C" int mte_canny_propagate(unsigned char* state, int* flags, int sweeps, int maps, int H, int W, hipStream_t stream) {
    return mte_hysteresis_propagate(state, flags, sweeps, maps, H, W, stream);
}

extern "C" int mte_canny_finish(const unsigned char* state, float* edges, int maps, int H, int W, hipStream_t stream) {
    if (!state || !edges || maps <= 0 || H <= 0 || W <= 0) return MTE_ERR_ARG;
    const long n = (long)maps * H * W;
    long g = (n + 256 * 8 - 1) / (256 * 8); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(canny_finish_kernel, dim3((unsigned)g), dim3(256), 0, stream, state, edges, n);
    return mte_check_launch();
}
