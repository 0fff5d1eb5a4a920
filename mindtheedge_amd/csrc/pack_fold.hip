// Folding PackLayerConv3d's Conv3d(1->4, 3x3x3) into its k x k Conv2d -- gfx950 helpers.
//
// Reference op chain (packnet_sfm/networks/layers/packnet/layers01.py:241-247):
//     P = packing(x)                       [B, D = 4C, H/2, W/2]
//     T = conv3d(P.unsqueeze(1)).view(B, 4D, H/2, W/2)          channel f*D + d,  zero pad 1 in (d, h, w)
//     y = conv2d(zero_pad_{k/2}(T), W) + b                      W: [Co, 4D, k, k]
// conv3d and conv2d are adjacent LINEAR maps, so away from the image border
//     y = conv2d(zero_pad_{k/2+1}(P), W') + b'     with   W'[co][c'][u] = sum_{f,kd,dh,dw} W[co][f*D + c'-kd+1][u-(dh,dw)] * K3[f][kd][dh][dw]
//                                                        b'[co]       = b[co] + sum_f b3[f] * sum_{d,t} W[co][f*D+d][t]
// which needs (k+2)^2 * 4C instead of k^2 * 16C multiply-adds per output (0.49x for the 5x5 pack1, 0.69x for the 3x3
// packs) and never materialises the 16C-channel tensor T.  Within k/2 pixels of the border the two forms differ (the
// reference zeroes T outside the image, the folded form sees conv3d's one-pixel spill-over and the bias only where T
// exists); those thin bands are recomputed with the unfolded kernels by the host logic (kernels.PackFoldedConvFn).
//
// This file: the weight fold (forward), its transpose (backward: dW, dK3, db3 from dW', db'), pixel (un)shuffle between
// x and P layouts, and a rectangle copy/add/zero used to cut and paste the border bands.
#include "common.hpp"

namespace {

// W' [Co][D][K2][K2] (K2 = k+2) <- W [Co][4D][k][k], K3 [4][3][3][3];   b' [Co]
__global__ void fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ K3, const float* __restrict__ b,
                                    const float* __restrict__ b3, float* __restrict__ Wf, float* __restrict__ bf,
                                    int Co, int D, int k) {
    __shared__ float s3[108];
    if (threadIdx.x < 108) s3[threadIdx.x] = K3[threadIdx.x];
    __syncthreads();
    const int K2 = k + 2, kk = k * k;
    const long total = (long)Co * D * K2 * K2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ux = (int)(i % K2); long t = i / K2;
        const int uy = (int)(t % K2); t /= K2;
        const int c = (int)(t % D); const int co = (int)(t / D);
        float acc = 0.f;
        for (int f = 0; f < 4; ++f)
            for (int kd = 0; kd < 3; ++kd) {
                const int d = c - kd + 1;
                if (d < 0 || d >= D) continue;
                const float* wrow = W + ((long)co * 4 * D + f * D + d) * kk;
#pragma unroll
                for (int dh = 0; dh < 3; ++dh) {
                    const int ty = uy - dh;
                    if (ty < 0 || ty >= k) continue;
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const int tx = ux - dw;
                        if (tx < 0 || tx >= k) continue;
                        acc = fmaf(wrow[ty * k + tx], s3[((f * 3 + kd) * 3 + dh) * 3 + dw], acc);
                    }
                }
            }
        Wf[i] = acc;
    }
    // b'[co] = b[co] + sum_f b3[f] * sum_{d,t} W[co][f][d][t]: one block per co (a wave per co walked the 4 * D * k * k weights in 64-wide
    // steps, one dependent load after the other: 80 us of a 90 us launch at D = 512); fixed summation order
    __shared__ float s_w[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int co = blockIdx.x; co < Co; co += gridDim.x) {
        float sf[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float* wr = W + ((long)co * 4 * D + (long)f * D) * kk;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const long n = (long)D * kk;
            long j = threadIdx.x;
            for (; j + 768 < n; j += 1024) { a0 += wr[j]; a1 += wr[j + 256]; a2 += wr[j + 512]; a3 += wr[j + 768]; }
            for (; j < n; j += 256) a0 += wr[j];
            sf[f] = wave_sum((a0 + a1) + (a2 + a3));
        }
        __syncthreads();                                      // s_w free again
        if (lane == 0)
#pragma unroll
            for (int f = 0; f < 4; ++f) s_w[f][wave] = sf[f];
        __syncthreads();
        if (threadIdx.x == 0) {
            float sum = 0.f;
            for (int f = 0; f < 4; ++f) sum = fmaf((s_w[f][0] + s_w[f][1]) + (s_w[f][2] + s_w[f][3]), b3[f], sum);
            bf[co] = b[co] + sum;
        }
    }
}

// dW [Co][4D][k][k] = sum_{kd,dh,dw} dWf[co][d+kd-1][t+(dh,dw)] * K3[f][kd][dh][dw]  +  dbf[co] * b3[f]
__global__ void unfold_dw_kernel(const float* __restrict__ dWf, const float* __restrict__ dbf, const float* __restrict__ K3,
                                 const float* __restrict__ b3, float* __restrict__ dW, int Co, int D, int k, int accumulate) {
    __shared__ float s3[108];
    if (threadIdx.x < 108) s3[threadIdx.x] = K3[threadIdx.x];
    __syncthreads();
    const int K2 = k + 2, kk = k * k;
    const long total = (long)Co * 4 * D * kk;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tx = (int)(i % k); long t = i / k;
        const int ty = (int)(t % k); t /= k;
        const int fd = (int)(t % (4 * D)); const int co = (int)(t / (4 * D));
        const int f = fd / D, d = fd - f * D;
        float acc = dbf[co] * b3[f];
        for (int kd = 0; kd < 3; ++kd) {
            const int c = d + kd - 1;
            if (c < 0 || c >= D) continue;
            const float* g = dWf + (((long)co * D + c) * K2) * K2;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw)
                    acc = fmaf(g[(ty + dh) * K2 + tx + dw], s3[((f * 3 + kd) * 3 + dh) * 3 + dw], acc);
        }
        dW[i] = accumulate ? dW[i] + acc : acc;
    }
}

// dK3[f][kd][dh][dw] += sum_{co,d,t} dWf[co][d+kd-1][t+(dh,dw)] * W[co][f*D+d][t];  db3[f] += sum_co dbf[co] * sum_{d,t} W[co][fD+d][t]
// grid.x strides over (co, d, tap); out[112] = dK3[108] then db3[4], atomically accumulated
__global__ __launch_bounds__(256) void unfold_dk3_kernel(const float* __restrict__ dWf, const float* __restrict__ dbf,
                                                         const float* __restrict__ W, float* __restrict__ out, int Co, int D, int k) {
    __shared__ float sred[4 * 112];
    const int K2 = k + 2, kk = k * k;
    float acc[112];
#pragma unroll
    for (int i = 0; i < 112; ++i) acc[i] = 0.f;
    // a thread takes (co, d, tap) items (with whole (co, d) rows per thread the 32 x 128 rows of pack1 were 16 workgroups walking 25 taps x 31
    // dependent loads each: 100 us for 0.4 MB)
    const long items = (long)Co * D * kk;
    for (long ix = blockIdx.x * (long)blockDim.x + threadIdx.x; ix < items; ix += (long)gridDim.x * blockDim.x) {
        const int t = (int)(ix % kk); const long rix = ix / kk;
        const int d = (int)(rix % D); const int co = (int)(rix / D);
        const float dbv = dbf[co];
        {
            const int ty = t / k, tx = t - ty * k;
            float wv[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) { wv[f] = W[((long)co * 4 * D + f * D + d) * kk + t]; acc[108 + f] = fmaf(dbv, wv[f], acc[108 + f]); }
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const int c = d + kd - 1;
                if (c < 0 || c >= D) continue;
                const float* g = dWf + (((long)co * D + c) * K2) * K2;
#pragma unroll
                for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const float gv = g[(ty + dh) * K2 + tx + dw];
#pragma unroll
                        for (int f = 0; f < 4; ++f) acc[((f * 3 + kd) * 3 + dh) * 3 + dw] = fmaf(gv, wv[f], acc[((f * 3 + kd) * 3 + dh) * 3 + dw]);
                    }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 112; ++i) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) sred[wave * 112 + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 112) atomicAdd(out + threadIdx.x, sred[threadIdx.x] + sred[112 + threadIdx.x] + sred[224 + threadIdx.x] + sred[336 + threadIdx.x]);
}

// P[b,h,w,4c+s] = x[b,2h+(s>>1),2w+(s&1),c]  (dir 0)  /  x <- P (dir 1)
template <typename T>
__global__ void pixel_shuffle_kernel(const T* __restrict__ src, long lds_, T* __restrict__ dst, long ldd, int B, int H, int W, int C, int dir) {
    const int H2 = H >> 1, W2 = W >> 1, cb = C >> 3;
    const long total = (long)B * H2 * W2 * cb;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % cb); long t = i / cb;
        const int w = (int)(t % W2); t /= W2;
        const int h = (int)(t % H2); const int b = (int)(t / H2);
        float v[4][8];
        if (dir == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const T* p = src + (((long)b * H + 2 * h + (s >> 1)) * W + 2 * w + (s & 1)) * lds_ + j * 8;
                if constexpr (sizeof(T) == 2) unpack16<bf16_t>(*(const u32x4_t*)p, v[s]);
                else { unpack16<float>(*(const u32x4_t*)p, v[s]); unpack16<float>(*(const u32x4_t*)(p + 4), v[s] + 4); }
            }
            T* q = dst + (((long)b * H2 + h) * W2 + w) * ldd + 32 * j;
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {              // depths 32j + 8*k8 .. +8  =  channels 2*k8, 2*k8+1 x 4 positions
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = v[e & 3][2 * k8 + (e >> 2)];
                if constexpr (sizeof(T) == 2) *(u32x4_t*)(q + 8 * k8) = pack16<bf16_t>(o);
                else { *(u32x4_t*)(q + 8 * k8) = pack16<float>(o); *(u32x4_t*)(q + 8 * k8 + 4) = pack16<float>(o + 4); }
            }
        } else {
            const T* q = src + (((long)b * H2 + h) * W2 + w) * lds_ + 32 * j;
            float o[4][8];
#pragma unroll
            for (int k8 = 0; k8 < 4; ++k8) {
                if constexpr (sizeof(T) == 2) unpack16<bf16_t>(*(const u32x4_t*)(q + 8 * k8), o[k8]);
                else { unpack16<float>(*(const u32x4_t*)(q + 8 * k8), o[k8]); unpack16<float>(*(const u32x4_t*)(q + 8 * k8 + 4), o[k8] + 4); }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float xv[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) xv[c] = o[c >> 1][(c & 1) * 4 + s];
                T* p = dst + (((long)b * H + 2 * h + (s >> 1)) * W + 2 * w + (s & 1)) * ldd + j * 8;
                if constexpr (sizeof(T) == 2) *(u32x4_t*)p = pack16<bf16_t>(xv);
                else { *(u32x4_t*)p = pack16<float>(xv); *(u32x4_t*)(p + 4) = pack16<float>(xv + 4); }
            }
        }
    }
}

// rectangle [h x w] of a [B,Hs,Ws,(lds)] tensor at (sy,sx)  ->  rectangle of a [B,Hd,Wd,(ldd)] tensor at (dy,dx)
// mode 0: copy, 1: add into dst, 2: zero the dst rectangle (src unused)
template <typename T>
__global__ void copy_rect_kernel(const T* __restrict__ src, long lds_, int Hs, int Ws, int sy, int sx,
                                 T* __restrict__ dst, long ldd, int Hd, int Wd, int dy, int dx, int B, int h, int w, int C, int mode) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = (long)B * h * w * cpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cpr); long t = i / cpr;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h); const int b = (int)(t / h);
        T* q = dst + (((long)b * Hd + dy + y) * Wd + dx + x) * ldd + cc * P;
        if (mode == 2) { *(u32x4_t*)q = u32x4_t{0u, 0u, 0u, 0u}; continue; }
        const u32x4_t sv = *(const u32x4_t*)(src + (((long)b * Hs + sy + y) * Ws + sx + x) * lds_ + cc * P);
        if (mode == 0) *(u32x4_t*)q = sv;
        else {
            float a[P], c[P];
            unpack16<T>(sv, a); unpack16<T>(*(const u32x4_t*)q, c);
#pragma unroll
            for (int e = 0; e < P; ++e) c[e] += a[e];
            *(u32x4_t*)q = pack16<T>(c);
        }
    }
}

// up to 8 rectangle operations in one launch (blockIdx.y = operation): the folded pack layers move four thin border bands
// in, out, and back per pass -- 60 launches of ~5 us per training step when issued one by one
struct RectOp { const void* src; long lds_; int Hs, Ws, sy, sx; void* dst; long ldd; int Hd, Wd, dy, dx, h, w, mode; };
struct RectOps { RectOp op[8]; };

template <typename T>
__global__ void copy_rects_kernel(RectOps ops, int B, int C) {
    const RectOp& o = ops.op[blockIdx.y];
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = (long)B * o.h * o.w * cpr;
    const T* src = (const T*)o.src;
    T* dst = (T*)o.dst;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % cpr); long t = i / cpr;
        const int x = (int)(t % o.w); t /= o.w;
        const int y = (int)(t % o.h); const int b = (int)(t / o.h);
        T* q = dst + (((long)b * o.Hd + o.dy + y) * o.Wd + o.dx + x) * o.ldd + cc * P;
        if (o.mode == 2) { *(u32x4_t*)q = u32x4_t{0u, 0u, 0u, 0u}; continue; }
        const u32x4_t sv = *(const u32x4_t*)(src + (((long)b * o.Hs + o.sy + y) * o.Ws + o.sx + x) * o.lds_ + cc * P);
        if (o.mode == 0) *(u32x4_t*)q = sv;
        else {
            float a[P], c[P];
            unpack16<T>(sv, a); unpack16<T>(*(const u32x4_t*)q, c);
#pragma unroll
            for (int e = 0; e < P; ++e) c[e] += a[e];
            *(u32x4_t*)q = pack16<T>(c);
        }
    }
}

inline int sgrid(long n) { long g = (n + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" {

// Wf [Co][D][k+2][k+2], bf [Co] (fp32)  <-  W [Co][4D][k][k], K3 [4][1][3][3][3], b [Co], b3 [4]
int mte_fold_pack_weights(const float* W, const float* K3, const float* b, const float* b3, float* Wf, float* bf,
                          int Co, int D, int k, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!W || !K3 || !b || !b3 || !Wf || !bf || (k & 1) == 0) return MTE_ERR_ARG;
    hipLaunchKernelGGL(fold_weights_kernel, dim3(sgrid((long)Co * D * (k + 2) * (k + 2))), dim3(256), 0, stream, W, K3, b, b3, Wf, bf, Co, D, k);
    return mte_check_launch();
}

// transpose of the fold: dW [Co][4D][k][k] (overwritten, or += if accumulate), dk3b [112] += (dK3[108], db3[4])
int mte_unfold_pack_wgrad(const float* dWf, const float* dbf, const float* W, const float* K3, const float* b3,
                          float* dW, float* dk3b, int Co, int D, int k, int accumulate, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dWf || !dbf || !W || !K3 || !b3 || !dW || !dk3b) return MTE_ERR_ARG;
    hipLaunchKernelGGL(unfold_dw_kernel, dim3(sgrid((long)Co * 4 * D * k * k)), dim3(256), 0, stream, dWf, dbf, K3, b3, dW, Co, D, k, accumulate);
    long items = (long)Co * D * k * k;
    long g = (items + 255) / 256; if (g > 512) g = 512; if (g < 1) g = 1;
    hipLaunchKernelGGL(unfold_dk3_kernel, dim3((unsigned)g), dim3(256), 0, stream, dWf, dbf, W, dk3b, Co, D, k);
    return mte_check_launch();
}

// dir 0: P[B,H/2,W/2,4C] <- x[B,H,W,C] (space-to-depth, d = 4c + 2dy + dx);  dir 1: x <- P.   H, W, C describe x.
int mte_pixel_shuffle(const void* src, long lds_, void* dst, long ldd, int B, int H, int W, int C, int dir, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!src || !dst || C % 8 != 0 || (H & 1) || (W & 1)) return MTE_ERR_ARG;
    const int grid = sgrid((long)B * (H / 2) * (W / 2) * (C / 8));
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(pixel_shuffle_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, lds_, (bf16_t*)dst, ldd, B, H, W, C, dir);
    else hipLaunchKernelGGL(pixel_shuffle_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)src, lds_, (float*)dst, ldd, B, H, W, C, dir);
    return mte_check_launch();
}

// rectangle copy (mode 0) / add (1) / zero-fill of the destination rectangle (2) between NHWC tensors
int mte_copy_rect(const void* src, long lds_, int Hs, int Ws, int sy, int sx, void* dst, long ldd, int Hd, int Wd, int dy, int dx,
                  int B, int h, int w, int C, int mode, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dst || (mode != 2 && !src) || C % 8 != 0 || h <= 0 || w <= 0) return MTE_ERR_ARG;
    if (sy < 0 || sx < 0 || dy < 0 || dx < 0 || dy + h > Hd || dx + w > Wd || (mode != 2 && (sy + h > Hs || sx + w > Ws))) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    const int grid = sgrid((long)B * h * w * (C / per16));
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(copy_rect_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, lds_, Hs, Ws, sy, sx, (bf16_t*)dst, ldd, Hd, Wd, dy, dx, B, h, w, C, mode);
    else hipLaunchKernelGGL(copy_rect_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)src, lds_, Hs, Ws, sy, sx, (float*)dst, ldd, Hd, Wd, dy, dx, B, h, w, C, mode);
    return mte_check_launch();
}

// n <= 8 rectangle operations (HOST array of mte_rect_op, see the header) on tensors of equal batch and channel count
int mte_copy_rects(const void* ops_host, int n, int B, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!ops_host || n < 1 || n > 8 || C % 8 != 0) return MTE_ERR_ARG;
    RectOps ops{};
    long most = 0;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    for (int i = 0; i < n; ++i) {
        const RectOp& o = ((const RectOp*)ops_host)[i];
        if (!o.dst || (o.mode != 2 && !o.src) || o.h <= 0 || o.w <= 0) return MTE_ERR_ARG;
        if (o.sy < 0 || o.sx < 0 || o.dy < 0 || o.dx < 0 || o.dy + o.h > o.Hd || o.dx + o.w > o.Wd ||
            (o.mode != 2 && (o.sy + o.h > o.Hs || o.sx + o.w > o.Ws))) return MTE_ERR_ARG;
        ops.op[i] = o;
        const long t = (long)B * o.h * o.w * (C / per16);
        if (t > most) most = t;
    }
    const dim3 grid(sgrid(most), n);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(copy_rects_kernel<bf16_t>, grid, dim3(256), 0, stream, ops, B, C);
    else hipLaunchKernelGGL(copy_rects_kernel<float>, grid, dim3(256), 0, stream, ops, B, C);
    return mte_check_launch();
}

}  // extern "C"
