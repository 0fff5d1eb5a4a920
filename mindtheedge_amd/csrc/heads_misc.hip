// Small HBM-bound kernels around the conv stack (gfx950, NHWC activations):
//   * InvDepth head: sigmoid(conv3x3(x, C->1) + b) / 0.5 and its backward   (layers01.py:99-123)
//   * NCHW fp32 image -> NHWC compute-dtype tensor with zero channel padding (+ optional horizontal flip,
//     SfmModel.py:58-96 / model_utils.py:98-117)
//   * nearest x2 up-sampling of an inv-depth map into one channel block of a decoder concat buffer
//     (PackNetSAN01.py:92-94,118-143) and its backward
//   * channel-slice copy (skip connections into concat buffers)
//   * fused Adam step over the flat fp32 master-parameter buffer (model_wrapper.py:142-180: Adam, wd 0)
#include "common.hpp"

namespace {

template <typename T> __device__ __forceinline__ void ld8(const T* p, float* v);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float* v) { unpack16<bf16_t>(*(const u32x4_t*)p, v); }
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* v) {
    unpack16<float>(*(const u32x4_t*)p, v); unpack16<float>(*(const u32x4_t*)(p + 4), v + 4);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float* v);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float* v) { *(u32x4_t*)p = pack16<bf16_t>(v); }
template <> __device__ __forceinline__ void st8<float>(float* p, const float* v) {
    *(u32x4_t*)p = pack16<float>(v); *(u32x4_t*)(p + 4) = pack16<float>(v + 4);
}

struct HeadArgs {
    const void* x; long ldx;
    const float* w; const float* bias;      // w: [C][3][3] (OIHW with O = 1), bias [1]
    float* out;                             // [B,H,W] inv-depth
    const float* dlogit;                    // [B,H,W]
    void* dx; long lddx;
    float* dw;                              // [C*9 + 1] (dw then db), atomically accumulated
    int B, H, W, C;
    float inv_min_depth;                    // 1 / min_depth = 2
    long npix;
};

// thread = (pixel, 8-channel block); the C/8 lanes of a pixel are adjacent lanes of one wave
template <typename T>
__global__ __launch_bounds__(256) void invdepth_fwd_kernel(HeadArgs a) {
    const int cb = a.C >> 3;
    const int j = threadIdx.x % cb;
    const int c0 = j * 8;
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[t][i] = a.w[(c0 + i) * 9 + t];
    const float bias = a.bias[0];
    const long ppb = 256 / cb;                                  // pixels per block iteration
    const long iters = (a.npix + (long)gridDim.x * ppb - 1) / ((long)gridDim.x * ppb);
    for (long it = 0; it < iters; ++it) {
        const long pixl = (it * gridDim.x + blockIdx.x) * ppb + threadIdx.x / cb;
        const bool live = pixl < a.npix;
        const long pix = live ? pixl : a.npix - 1;
        const int x = (int)(pix % a.W); const long t2 = pix / a.W; const int y = (int)(t2 % a.H);
        // all nine 16-byte loads are issued before the first multiply (clamped addresses, out-of-image taps masked afterwards)
        u32x4_t raw[9];
        bool ok[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
            ok[t] = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const long q = ok[t] ? pix + (long)(t / 3 - 1) * a.W + (t % 3 - 1) : pix;
            if constexpr (sizeof(T) == 2) raw[t] = *(const u32x4_t*)((const T*)a.x + q * a.ldx + c0);
        }
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v[8];
            if constexpr (sizeof(T) == 2) unpack16<T>(raw[t], v);
            else ld8<T>((const T*)a.x + (ok[t] ? pix + (long)(t / 3 - 1) * a.W + (t % 3 - 1) : pix) * a.ldx + c0, v);
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) part = fmaf(v[i], wr[t][i], part);
            acc += ok[t] ? part : 0.f;
        }
        for (int o = cb >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (live && j == 0) a.out[pix] = a.inv_min_depth / (1.f + __expf(-(acc + bias)));
    }
}

// dlogit = dout * d(inv)/d(logit), inv = s/(1+e^-z) with s = 1/min_depth: d inv / dz = inv * (1 - inv/s)
__global__ void invdepth_dlogit_kernel(const float* __restrict__ dout, const float* __restrict__ inv, float* __restrict__ dl, long n, float s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = inv[i];
        dl[i] = dout[i] * v * (1.f - v / s);
    }
}

// Backward of the head, as two independent streams over the pixels (the logit gradient dlogit is one fp32 per pixel):
//   data   : dx[q][c]  = sum_t dlogit[q - t] * w[c][t]        -- needs no activation at all: a pure 16-B store stream
//   weight : dw[c][t] += sum_q dlogit[q - t] * x[q][c], db += sum_q dlogit[q]   -- a pure load stream with 72 register
//            accumulators per thread, one LDS + one global atomic pass per block
// Split because together they need 72 weights + 72 accumulators per thread (256 VGPRs, 8 waves/CU, latency-bound);
// apart each keeps < 128 VGPRs, and the weight half can run on the weight-gradient side stream.
template <typename T>
__global__ __launch_bounds__(256) void invdepth_bwd_data_kernel(HeadArgs a) {
    const int cb = a.C >> 3;
    const int j = threadIdx.x % cb;
    const int c0 = j * 8;
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[t][i] = a.w[(c0 + i) * 9 + t];
    const long ppb = 256 / cb;
    const long stride = (long)gridDim.x * ppb;
    constexpr int U = 2;
#pragma unroll 1
    for (long base = blockIdx.x * ppb + threadIdx.x / cb; base < a.npix; base += stride * U) {
        float dl[U][9];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = base + u * stride;
            if (pix < a.npix) {
                const int x = (int)(pix % a.W); const int y = (int)((pix / a.W) % a.H);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int oy = t / 3 - 1, ox = t % 3 - 1;
                    const bool ok = (unsigned)(y - oy) < (unsigned)a.H && (unsigned)(x - ox) < (unsigned)a.W;
                    dl[u][t] = ok ? a.dlogit[pix - (long)oy * a.W - ox] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = base + u * stride;
            if (pix >= a.npix) break;
            float dxv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) dxv[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) dxv[i] = fmaf(dl[u][t], wr[t][i], dxv[i]);
            st8<T>((T*)a.dx + pix * a.lddx + c0, dxv);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void invdepth_bwd_weight_kernel(HeadArgs a) {
    extern __shared__ float sdw[];                               // [C*9 + 1] block-level gradient accumulators
    for (int i = threadIdx.x; i < a.C * 9 + 1; i += 256) sdw[i] = 0.f;
    __syncthreads();
    const int cb = a.C >> 3;
    const int j = threadIdx.x % cb;
    const int c0 = j * 8;
    float gw[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) gw[t][i] = 0.f;
    float gb = 0.f;
    const long ppb = 256 / cb;
    const long stride = (long)gridDim.x * ppb;
    constexpr int U = 4;                                         // 16-byte activation loads in flight per thread
#pragma unroll 1
    for (long base = blockIdx.x * ppb + threadIdx.x / cb; base < a.npix; base += stride * U) {
        u32x4_t xr[U];
        float dl[U][9];
#pragma unroll
        for (int u = 0; u < U; ++u) {                              // every load of the batch first: activations and logit gradients
            const long pix = base + u * stride;
            if (pix < a.npix) {
                if constexpr (sizeof(T) == 2) xr[u] = *(const u32x4_t*)((const T*)a.x + pix * a.ldx + c0);
                const int x = (int)(pix % a.W); const int y = (int)((pix / a.W) % a.H);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int oy = t / 3 - 1, ox = t % 3 - 1;
                    const bool ok = (unsigned)(y - oy) < (unsigned)a.H && (unsigned)(x - ox) < (unsigned)a.W;
                    dl[u][t] = ok ? a.dlogit[pix - (long)oy * a.W - ox] : 0.f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long pix = base + u * stride;
            if (pix >= a.npix) break;
            float v[8];
            if constexpr (sizeof(T) == 2) unpack16<T>(xr[u], v);
            else ld8<T>((const T*)a.x + pix * a.ldx + c0, v);            // fp32 validation mode: 8 channels = two chunks
            if (j == 0) gb += dl[u][4];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) gw[t][i] = fmaf(dl[u][t], v[i], gw[t][i]);
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&sdw[(c0 + i) * 9 + t], gw[t][i]);
    if (j == 0) atomicAdd(&sdw[a.C * 9], gb);
    __syncthreads();
    for (int i = threadIdx.x; i < a.C * 9 + 1; i += 256) atomicAdd(&a.dw[i], sdw[i]);
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W, int Cp, long ldd, int flip) {
    const long npix = (long)B * H * W;
    for (long pix = blockIdx.x * (long)blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const int x = (int)(pix % W); const long t = pix / W; const int y = (int)(t % H); const int b = (int)(t / H);
        const int sx = flip ? W - 1 - x : x;
        for (int c0 = 0; c0 < Cp; c0 += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (c0 + i < C) ? src[(((long)b * C + c0 + i) * H + y) * W + sx] : 0.f;
            st8<T>(dst + pix * ldd + c0, v);
        }
    }
}

// dst[b, 2h+dy, 2w+dx, 0..7] = (inv[b,h,w], 0, ..., 0)
template <typename T>
__global__ void upsample_inv_kernel(const float* __restrict__ inv, T* __restrict__ dst, long ldd, int B, int h, int w) {
    const long n = (long)B * 4 * h * w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % (2 * w)); const long t = i / (2 * w); const int Y = (int)(t % (2 * h)); const int b = (int)(t / (2 * h));
        float v[8] = {inv[((long)b * h + (Y >> 1)) * w + (X >> 1)], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        st8<T>(dst + i * ldd, v);
    }
}
template <typename T>
__global__ void upsample_inv_bwd_kernel(const T* __restrict__ dsrc, long lds_, float* __restrict__ dinv, int B, int h, int w, int accumulate) {
    const long n = (long)B * h * w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w); const long t = i / w; const int y = (int)(t % h); const int b = (int)(t / h);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            s += Elem<T>::ld(dsrc + (((long)b * 2 * h + 2 * y + (d >> 1)) * (2 * w) + 2 * x + (d & 1)) * lds_);
        dinv[i] = accumulate ? dinv[i] + s : s;
    }
}

template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long n = npix * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cpr; const int cc = (int)(i % cpr);
        *(u32x4_t*)(dst + pix * ldd + cc * P) = *(const u32x4_t*)(src + pix * lds_ + cc * P);
    }
}

// out = a + b on channel-slice views (fp32 sum, rounded once): the gradient of a tensor with two consumers
template <typename T>
__global__ void add_channels_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, T* __restrict__ out, long ldo,
                                    long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long n = npix * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / cpr; const int cc = (int)(i % cpr);
        float va[P], vb[P];
        unpack16<T>(*(const u32x4_t*)(a + pix * lda + cc * P), va);
        unpack16<T>(*(const u32x4_t*)(b + pix * ldb + cc * P), vb);
#pragma unroll
        for (int k = 0; k < P; ++k) va[k] += vb[k];
        *(u32x4_t*)(out + pix * ldo + cc * P) = pack16<T>(va);
    }
}

// torch.optim.Adam semantics (no weight decay, no amsgrad): bc1 = 1 - b1^t, bc2s = sqrt(1 - b2^t) from the host
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float bc1, float bc2s, float gscale) {
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4_t pp = ((f32x4_t*)p)[i], gg = ((const f32x4_t*)g)[i], mm = ((f32x4_t*)m)[i], vv = ((f32x4_t*)v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale;
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vv[k]) / bc2s + eps;
            pp[k] -= (lr / bc1) * (mm[k] / denom);
        }
        ((f32x4_t*)p)[i] = pp; ((f32x4_t*)m)[i] = mm; ((f32x4_t*)v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * gscale;
        const float mk = b1 * m[i] + (1.f - b1) * gk, vk = b2 * v[i] + (1.f - b2) * gk * gk;
        m[i] = mk; v[i] = vk;
        p[i] -= (lr / bc1) * (mk / (sqrtf(vk) / bc2s + eps));
    }
}

// the same update with lr / bias corrections read from device memory (hyper = {lr, 1 - b1^t, sqrt(1 - b2^t)}): the launch carries no
// per-step host scalar, so it can be replayed from a HIP graph while the host refreshes `hyper` before each replay
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long n, const float* __restrict__ hyper, float b1, float b2, float eps, float gscale) {
    const float lr = hyper[0], bc1 = hyper[1], bc2s = hyper[2];
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4_t pp = ((f32x4_t*)p)[i], gg = ((const f32x4_t*)g)[i], mm = ((f32x4_t*)m)[i], vv = ((f32x4_t*)v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale;
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            const float denom = sqrtf(vv[k]) / bc2s + eps;
            pp[k] -= (lr / bc1) * (mm[k] / denom);
        }
        ((f32x4_t*)p)[i] = pp; ((f32x4_t*)m)[i] = mm; ((f32x4_t*)v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * gscale;
        const float mk = b1 * m[i] + (1.f - b1) * gk, vk = b2 * v[i] + (1.f - b2) * gk * gk;
        m[i] = mk; v[i] = vk;
        p[i] -= (lr / bc1) * (mk / (sqrtf(vk) / bc2s + eps));
    }
}

// F.interpolate(x, size, mode='bilinear', align_corners=False) on fp32 [B,h,w] maps and its adjoint (GradLoss.forward's resize of
// the prediction to the label size, grad_loss.py:127).  Source index as ATen's area_pixel_compute_source_index: scale * (dst + 0.5)
// - 0.5, clamped at 0; weights in float.
__device__ __forceinline__ void bilin_src(int dst, int in, float scale, int& i0, int& i1, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 < in - 1 ? i0 + 1 : i0;
    l1 = src - (float)i0;
}
__global__ void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ dx, const float* __restrict__ dy,
                                       int B, int h, int w, int H, int W) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long n = (long)B * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        int y0, y1, x0, x1; float ly, lx;
        bilin_src(Y, h, sh, y0, y1, ly); bilin_src(X, w, sw, x0, x1, lx);
        const long base = (long)b * h * w;
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
        if (y) {
            y[i] = w00 * x[base + (long)y0 * w + x0] + w01 * x[base + (long)y0 * w + x1] + w10 * x[base + (long)y1 * w + x0] + w11 * x[base + (long)y1 * w + x1];
        } else {
            const float g = dy[i];
            atomicAdd(dx + base + (long)y0 * w + x0, w00 * g); atomicAdd(dx + base + (long)y0 * w + x1, w01 * g);
            atomicAdd(dx + base + (long)y1 * w + x0, w10 * g); atomicAdd(dx + base + (long)y1 * w + x1, w11 * g);
        }
    }
}

inline int stream_grid(long n, int per = 256) { long g = (n + per - 1) / per; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }
inline bool head_ok(int C) { const int cb = C >> 3; return C % 8 == 0 && cb >= 1 && cb <= 64 && (cb & (cb - 1)) == 0; }

}  // namespace

extern "C" {

int mte_invdepth_fwd(const void* x, long ldx, const float* w, const float* bias, float* out,
                     int B, int H, int W, int C, float min_depth, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !w || !bias || !out || !head_ok(C)) return MTE_ERR_ARG;
    HeadArgs a{}; a.x = x; a.ldx = ldx; a.w = w; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W; a.C = C;
    a.inv_min_depth = 1.f / min_depth; a.npix = (long)B * H * W;
    const int ppb = 256 / (C / 8);
    const int grid = stream_grid(a.npix, ppb * 4);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(invdepth_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(invdepth_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// dlogit_scratch [B*H*W] fp32; dwb [C*9 + 1] fp32 (zeroed here): dw (OIHW order, O = 1) followed by db
int mte_invdepth_bwd_data(const float* w, const float* inv_out, const float* dout, float* dlogit,
                          void* dx, long lddx, int B, int H, int W, int C, float min_depth, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!w || !inv_out || !dout || !dlogit || !dx || !head_ok(C)) return MTE_ERR_ARG;
    const long npix = (long)B * H * W;
    hipLaunchKernelGGL(invdepth_dlogit_kernel, dim3(stream_grid(npix)), dim3(256), 0, stream, dout, inv_out, dlogit, npix, 1.f / min_depth);
    HeadArgs a{}; a.w = w; a.dlogit = dlogit; a.dx = dx; a.lddx = lddx; a.B = B; a.H = H; a.W = W; a.C = C; a.npix = npix;
    const int ppb = 256 / (C / 8);
    const int grid = stream_grid(npix, ppb * 4);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(invdepth_bwd_data_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(invdepth_bwd_data_kernel<float>, dim3(grid), dim3(256), 0, stream, a);
    return mte_check_launch();
}

int mte_invdepth_bwd_weight(const void* x, long ldx, const float* dlogit, float* dwb,
                            int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dlogit || !dwb || !head_ok(C)) return MTE_ERR_ARG;
    const long npix = (long)B * H * W;
    if (mte_memset_async(dwb, 0, sizeof(float) * (C * 9 + 1), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    HeadArgs a{}; a.x = x; a.ldx = ldx; a.dlogit = dlogit; a.dw = dwb; a.B = B; a.H = H; a.W = W; a.C = C; a.npix = npix;
    const int ppb = 256 / (C / 8);
    // few, long-running blocks: every block ends with C*9+1 same-address global atomics (contended adds are ~14x slower);
    // each block should stream >= ~4x more 16-byte chunks than it issues atomics: npix * C/8 / g >= 4 * 9C
    long g = npix / 288; if (g > 1024) g = 1024; if (g < 64) g = 64;
    if (g > (npix + ppb - 1) / ppb) g = (npix + ppb - 1) / ppb;
    const size_t lds = sizeof(float) * (C * 9 + 4);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(invdepth_bwd_weight_kernel<bf16_t>, dim3((unsigned)g), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(invdepth_bwd_weight_kernel<float>, dim3((unsigned)g), dim3(256), lds, stream, a);
    return mte_check_launch();
}

int mte_nchw_to_nhwc(const float* src, void* dst, long ldd, int B, int C, int H, int W, int Cp, int flip_w, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!src || !dst || Cp % 8 != 0 || Cp < C) return MTE_ERR_ARG;
    const int grid = stream_grid((long)B * H * W);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, src, (bf16_t*)dst, B, C, H, W, Cp, ldd, flip_w);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, stream, src, (float*)dst, B, C, H, W, Cp, ldd, flip_w);
    return mte_check_launch();
}

int mte_upsample_inv_fwd(const float* inv, void* dst, long ldd, int B, int h, int w, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!inv || !dst) return MTE_ERR_ARG;
    const int grid = stream_grid((long)B * 4 * h * w);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(upsample_inv_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, inv, (bf16_t*)dst, ldd, B, h, w);
    else hipLaunchKernelGGL(upsample_inv_kernel<float>, dim3(grid), dim3(256), 0, stream, inv, (float*)dst, ldd, B, h, w);
    return mte_check_launch();
}
int mte_upsample_inv_bwd(const void* dsrc, long lds_, float* dinv, int B, int h, int w, int accumulate, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dsrc || !dinv) return MTE_ERR_ARG;
    const int grid = stream_grid((long)B * h * w);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(upsample_inv_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dsrc, lds_, dinv, B, h, w, accumulate);
    else hipLaunchKernelGGL(upsample_inv_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)dsrc, lds_, dinv, B, h, w, accumulate);
    return mte_check_launch();
}

int mte_copy_channels(const void* src, long lds_, void* dst, long ldd, long npix, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!src || !dst || C % 8 != 0) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    const int grid = stream_grid(npix * (C / per16));
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(copy_channels_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, npix, C);
    else hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)src, lds_, (float*)dst, ldd, npix, C);
    return mte_check_launch();
}

int mte_add_channels(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long npix, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!a || !b || !out || C % per16 != 0) return MTE_ERR_ARG;
    const int grid = stream_grid(npix * (C / per16));
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(add_channels_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, npix, C);
    else hipLaunchKernelGGL(add_channels_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)a, lda, (const float*)b, ldb, (float*)out, ldo, npix, C);
    return mte_check_launch();
}

// In-place Adam over flat fp32 buffers (16-byte aligned).  step >= 1.  gscale multiplies the gradient first
// (1/world_size when the all-reduce summed instead of averaged; 1 otherwise).
int mte_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  int step, float gscale, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!p || !g || !m || !v || n <= 0 || step < 1) return MTE_ERR_ARG;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       (float)bc1, (float)sqrt(bc2), gscale);
    return mte_check_launch();
}

// y[B,H,W] = bilinear resize (align_corners = False, as F.interpolate) of x[B,h,w]
int mte_resize_bilinear_fwd(const float* x, float* y, int B, int h, int w, int H, int W, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !y || B < 1 || h < 1 || w < 1 || H < 1 || W < 1) return MTE_ERR_ARG;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(stream_grid((long)B * H * W)), dim3(256), 0, stream, x, y, (float*)nullptr, (const float*)nullptr, B, h, w, H, W);
    return mte_check_launch();
}
// dx[B,h,w] (overwritten) = adjoint of the resize applied to dy[B,H,W]
int mte_resize_bilinear_bwd(const float* dy, float* dx, int B, int h, int w, int H, int W, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dy || !dx || B < 1 || h < 1 || w < 1 || H < 1 || W < 1) return MTE_ERR_ARG;
    if (mte_memset_async(dx, 0, sizeof(float) * (size_t)B * h * w, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(stream_grid((long)B * H * W)), dim3(256), 0, stream, (const float*)nullptr, (float*)nullptr, dx, dy, B, h, w, H, W);
    return mte_check_launch();
}

// The same step with {lr, 1 - beta1^t, sqrt(1 - beta2^t)} in DEVICE memory (`hyper`, 3 floats written by the caller before the
// launch): nothing in the launch changes from step to step, so a captured HIP graph of the training step stays valid.
int mte_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2, float eps,
                      float gscale, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!p || !g || !m || !v || !hyper || n <= 0) return MTE_ERR_ARG;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(stream_grid(n >> 2)), dim3(256), 0, stream, p, g, m, v, n, hyper, beta1, beta2, eps, gscale);
    return mte_check_launch();
}

}  // extern "C"
