// Sparse auxiliary (SAN) branch of PackNet-SAN for gfx950, inference and training.  PARITY UNPINNED: the
// reference runs this branch on MinkowskiEngine (third-party CUDA, not available here); what is implemented is the
// dense-equivalent of its published semantics (see oracle/san_oracle.py for the statement being followed):
//   sparsify_depth   active set = pixels with depth > 0, one feature = the depth       (networks/layers/minkowski.py:33-57)
//   MinkowskiMaxPooling(3, stride 2): a coarse cell exists iff one of its 2x2 fine cells does; its value is the maximum
//                    over the ACTIVE fine cells of the centred 3x3 window                (minkowski_encoder.py:55,83-84)
//   MinkowskiConvolution(k, stride 1, no bias): dense convolution of the zero-filled map, evaluated on the active set only
//   MinkowskiBatchNorm (eval) + MinkowskiReLU on the active set                          (minkowski_encoder.py:27-54)
//   densify_features = the zero-filled dense map                                         (minkowski.py:60-79)
//   fusion  skip * w + sparse + b                                                        (networks/depth/PackNetSAN01.py:248-258)
// The convolutions themselves run on the dense MFMA kernels of this library (mte_conv2d_igemm / _patch_fwd) on the
// zero-filled NHWC maps; the kernels below are the HBM-bound glue around them.  At 5 % LiDAR density the dense form
// does ~20x the arithmetic of a gather-scatter sparse convolution but needs no coordinate maps and stays on the MFMA
// path; the whole branch is < 3 % of a forward pass.
#include "common.hpp"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void sparsify_depth_kernel(const float* __restrict__ depth, T* __restrict__ feat, long ldf,
                                                             unsigned char* __restrict__ mask, long npix) {
    constexpr int P = Elem<T>::PER16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        const float d = depth[i];
        const bool on = d > 0.f;
        mask[i] = on ? 1 : 0;
        float v[8] = {on ? d : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < 8; c += P) *(u32x4_t*)(feat + i * ldf + c) = pack16<T>(v + c);
    }
}

// out = mask ? relu(bn(a + b + c)) : 0 ; b, c nullable ; bn(x) = (x - mean) / sqrt(var + eps) * gamma + beta
template <typename T>
__global__ __launch_bounds__(256) void sparse_bn_relu_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb,
                                                             const T* __restrict__ c, long ldc, const unsigned char* __restrict__ mask,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                             T* __restrict__ out, long ldo, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = npix * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        float v[P];
        if (mask[p]) {
            unpack16<T>(*(const u32x4_t*)(a + p * lda + ch0), v);
            if (b) { float t[P]; unpack16<T>(*(const u32x4_t*)(b + p * ldb + ch0), t); for (int k = 0; k < P; ++k) v[k] += t[k]; }
            if (c) { float t[P]; unpack16<T>(*(const u32x4_t*)(c + p * ldc + ch0), t); for (int k = 0; k < P; ++k) v[k] += t[k]; }
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float s = gamma[ch0 + k] / sqrtf(var[ch0 + k] + eps);
                v[k] = fmaxf((v[k] - mean[ch0 + k]) * s + beta[ch0 + k], 0.f);
            }
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k) v[k] = 0.f;
        }
        *(u32x4_t*)(out + p * ldo + ch0) = pack16<T>(v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sparse_maxpool_kernel(const T* __restrict__ in, long ldi, const unsigned char* __restrict__ mask_in,
                                                             T* __restrict__ out, long ldo, unsigned char* __restrict__ mask_out,
                                                             int B, int H, int W, int C) {
    constexpr int P = Elem<T>::PER16;
    const int Ho = H / 2, Wo = W / 2, cpr = C / P;
    const long total = (long)B * Ho * Wo * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((long)Wo * Ho));
        const unsigned char* m = mask_in + (long)b * H * W;
        const int y0 = 2 * yo, x0 = 2 * xo;
        const bool on = m[(long)y0 * W + x0] | m[(long)y0 * W + x0 + 1] | m[(long)(y0 + 1) * W + x0] | m[(long)(y0 + 1) * W + x0 + 1];
        float v[P];
#pragma unroll
        for (int k = 0; k < P; ++k) v[k] = on ? -__builtin_huge_valf() : 0.f;
        if (on) {
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int y = y0 + dy, x = x0 + dx;
                    if ((unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W || !m[(long)y * W + x]) continue;
                    float t[P];
                    unpack16<T>(*(const u32x4_t*)(in + (((long)b * H + y) * W + x) * ldi + ch0), t);
#pragma unroll
                    for (int k = 0; k < P; ++k) v[k] = fmaxf(v[k], t[k]);
                }
        }
        *(u32x4_t*)(out + p * ldo + ch0) = pack16<T>(v);
        if (ch0 == 0) mask_out[p] = on ? 1 : 0;
    }
}

// out = skip * w + sparse + b   (w, b: one device float each)
template <typename T>
__global__ __launch_bounds__(256) void san_fuse_kernel(const T* __restrict__ skip, long lds_, const T* __restrict__ sparse, long ldp,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       T* __restrict__ out, long ldo, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = npix * cpr;
    const float ww = *w, bb = *bias;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        float s[P], q[P];
        unpack16<T>(*(const u32x4_t*)(skip + p * lds_ + ch0), s);
        unpack16<T>(*(const u32x4_t*)(sparse + p * ldp + ch0), q);
#pragma unroll
        for (int k = 0; k < P; ++k) s[k] = s[k] * ww + q[k] + bb;
        *(u32x4_t*)(out + p * ldo + ch0) = pack16<T>(s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Training path of the branch (round 2).  MinkowskiBatchNorm in training mode = BatchNorm1d over the ACTIVE points of the
// whole batch (biased variance for the normalisation); the kernels below are the masked statistics / backward passes around
// the dense MFMA convolutions, plus the backward of the pooling and of the fusion.
// Work item = (pixel, 16-byte channel chunk); a thread keeps ONE chunk for its whole pixel loop (256 % chunks-per-pixel == 0),
// so per-channel sums live in registers and are folded per workgroup in LDS, then added to the fp64 global sums.
// ---------------------------------------------------------------------------------------------------------------------
template <int P> __device__ __forceinline__ void fold_channel_sums(float (&s0)[P], float (&s1)[P], int chunk, int cpr, double* __restrict__ gsum, int C) {
    extern __shared__ float s_red[];                       // [2][cpr * P]
    for (int i = threadIdx.x; i < 2 * cpr * P; i += 256) s_red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P; ++k) { atomicAdd(&s_red[chunk * P + k], s0[k]); atomicAdd(&s_red[cpr * P + chunk * P + k], s1[k]); }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * cpr * P; i += 256) {
        const float v = s_red[i];
        if (v != 0.f) atomicAdd(gsum + (i < cpr * P ? i : C + (i - cpr * P)), (double)v);
    }
}

template <typename T> __device__ __forceinline__ void load_sum3(const T* a, long lda, const T* b, long ldb, const T* c, long ldc, long p, int ch0, float* v) {
    constexpr int P = Elem<T>::PER16;
    unpack16<T>(*(const u32x4_t*)(a + p * lda + ch0), v);
    if (b) { float t[P]; unpack16<T>(*(const u32x4_t*)(b + p * ldb + ch0), t); for (int k = 0; k < P; ++k) v[k] += t[k]; }
    if (c) { float t[P]; unpack16<T>(*(const u32x4_t*)(c + p * ldc + ch0), t); for (int k = 0; k < P; ++k) v[k] += t[k]; }
}

// sums[0..C) += sum x, sums[C..2C) += sum x^2 over the active pixels (x = a + b + c), sums[2C] += number of active pixels
template <typename T>
__global__ __launch_bounds__(256) void sparse_bn_stats_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, const T* __restrict__ c, long ldc,
                                                              const unsigned char* __restrict__ mask, double* __restrict__ sums, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P, chunk = threadIdx.x % cpr, ppb = 256 / cpr;
    float s0[P], s1[P]; float cnt = 0.f;
#pragma unroll
    for (int k = 0; k < P; ++k) s0[k] = s1[k] = 0.f;
    for (long p = (long)blockIdx.x * ppb + threadIdx.x / cpr; p < npix; p += (long)gridDim.x * ppb) {
        if (!mask[p]) continue;
        float v[P];
        load_sum3<T>(a, lda, b, ldb, c, ldc, p, chunk * P, v);
#pragma unroll
        for (int k = 0; k < P; ++k) { s0[k] += v[k]; s1[k] = fmaf(v[k], v[k], s1[k]); }
        if (chunk == 0) cnt += 1.f;
    }
    fold_channel_sums<P>(s0, s1, chunk, cpr, sums, C);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt != 0.f) atomicAdd(sums + 2 * C, (double)cnt);
}

// dy = dout * [out > 0] (out is zero off the active set): sums[0..C) += sum dy, sums[C..2C) += sum dy * xhat, xhat = (x - mean) * invstd
template <typename T>
__global__ __launch_bounds__(256) void sparse_bn_bwd_reduce_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, const T* __restrict__ c, long ldc,
                                                                   const T* __restrict__ out, long ldo, const T* __restrict__ dout, long ldd,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   double* __restrict__ sums, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P, chunk = threadIdx.x % cpr, ppb = 256 / cpr, ch0 = chunk * P;
    float s0[P], s1[P], mu[P], is[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { s0[k] = s1[k] = 0.f; mu[k] = mean[ch0 + k]; is[k] = invstd[ch0 + k]; }
    for (long p = (long)blockIdx.x * ppb + threadIdx.x / cpr; p < npix; p += (long)gridDim.x * ppb) {
        float o[P], g[P], v[P];
        unpack16<T>(*(const u32x4_t*)(out + p * ldo + ch0), o);
        bool any = false;
#pragma unroll
        for (int k = 0; k < P; ++k) any = any || o[k] > 0.f;
        if (!any) continue;
        unpack16<T>(*(const u32x4_t*)(dout + p * ldd + ch0), g);
        load_sum3<T>(a, lda, b, ldb, c, ldc, p, ch0, v);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float dy = o[k] > 0.f ? g[k] : 0.f;
            s0[k] += dy; s1[k] = fmaf(dy, (v[k] - mu[k]) * is[k], s1[k]);
        }
    }
    fold_channel_sums<P>(s0, s1, chunk, cpr, sums, C);
}

// dx = mask ? gamma * invstd * (dy - m1 - xhat * m2) : 0, m1 = sum dy / N, m2 = sum dy xhat / N (N = active pixels)
template <typename T>
__global__ __launch_bounds__(256) void sparse_bn_bwd_apply_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, const T* __restrict__ c, long ldc,
                                                                  const T* __restrict__ out, long ldo, const T* __restrict__ dout, long ldd,
                                                                  const unsigned char* __restrict__ mask, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  const double* __restrict__ sums, const double* __restrict__ n_active, T* __restrict__ dx, long ldx, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = npix * cpr;
    const double inv_n = n_active[0] > 0.0 ? 1.0 / n_active[0] : 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        float r[P];
        if (mask[p]) {
            float o[P], g[P], v[P];
            unpack16<T>(*(const u32x4_t*)(out + p * ldo + ch0), o);
            unpack16<T>(*(const u32x4_t*)(dout + p * ldd + ch0), g);
            load_sum3<T>(a, lda, b, ldb, c, ldc, p, ch0, v);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float is = invstd[ch0 + k], xh = (v[k] - mean[ch0 + k]) * is;
                const float dy = o[k] > 0.f ? g[k] : 0.f;
                r[k] = gamma[ch0 + k] * is * (dy - (float)(sums[ch0 + k] * inv_n) - xh * (float)(sums[C + ch0 + k] * inv_n));
            }
        } else {
#pragma unroll
            for (int k = 0; k < P; ++k) r[k] = 0.f;
        }
        *(u32x4_t*)(dx + p * ldx + ch0) = pack16<T>(r);
    }
}

// backward of sparse_maxpool_kernel: a fine cell receives the gradient of every coarse cell whose window maximum it is
// (first maximum in row-major window order among the active cells); gathered per fine cell, no atomics
template <typename T>
__global__ __launch_bounds__(256) void sparse_maxpool_bwd_kernel(const T* __restrict__ in, long ldi, const unsigned char* __restrict__ mask_in,
                                                                 const T* __restrict__ dout, long ldd, T* __restrict__ din, long ldn, int B, int H, int W, int C) {
    constexpr int P = Elem<T>::PER16;
    const int Ho = H / 2, Wo = W / 2, cpr = C / P;
    const long total = (long)B * H * W * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        const unsigned char* m = mask_in + (long)b * H * W;
        float r[P];
#pragma unroll
        for (int k = 0; k < P; ++k) r[k] = 0.f;
        if (m[(long)y * W + x]) {
            float mine[P];
            unpack16<T>(*(const u32x4_t*)(in + p * ldi + ch0), mine);
            // coarse cells (yo, xo) whose window rows 2yo-1..2yo+1 / columns 2xo-1..2xo+1 contain (y, x)
            for (int yo = y >> 1; yo <= (y + 1) >> 1; ++yo)
                for (int xo = x >> 1; xo <= (x + 1) >> 1; ++xo) {
                    if (yo >= Ho || xo >= Wo) continue;
                    const int y0 = 2 * yo, x0 = 2 * xo;
                    if (!(m[(long)y0 * W + x0] | m[(long)y0 * W + x0 + 1] | m[(long)(y0 + 1) * W + x0] | m[(long)(y0 + 1) * W + x0 + 1])) continue;
                    bool win[P];
#pragma unroll
                    for (int k = 0; k < P; ++k) win[k] = true;
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int yy = y0 + dy, xx = x0 + dx;
                            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W || !m[(long)yy * W + xx] || (yy == y && xx == x)) continue;
                            float t[P];
                            unpack16<T>(*(const u32x4_t*)(in + (((long)b * H + yy) * W + xx) * ldi + ch0), t);
                            const bool before = yy < y || (yy == y && xx < x);      // earlier in window order: wins ties
#pragma unroll
                            for (int k = 0; k < P; ++k) win[k] = win[k] && (before ? t[k] < mine[k] : t[k] <= mine[k]);
                        }
                    float g[P];
                    unpack16<T>(*(const u32x4_t*)(dout + (((long)b * Ho + yo) * Wo + xo) * ldd + ch0), g);
#pragma unroll
                    for (int k = 0; k < P; ++k) r[k] += win[k] ? g[k] : 0.f;
                }
        }
        *(u32x4_t*)(din + p * ldn + ch0) = pack16<T>(r);
    }
}

// backward of san_fuse: dskip = dout * w; sums[0] += sum dout * skip (dw), sums[1] += sum dout (db)
template <typename T>
__global__ __launch_bounds__(256) void san_fuse_bwd_kernel(const T* __restrict__ skip, long lds_, const T* __restrict__ dout, long ldd, const float* __restrict__ w,
                                                           T* __restrict__ dskip, long ldk, double* __restrict__ sums, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = npix * cpr;
    const float ww = *w;
    float s0 = 0.f, s1 = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        float s[P], g[P];
        unpack16<T>(*(const u32x4_t*)(skip + p * lds_ + ch0), s);
        unpack16<T>(*(const u32x4_t*)(dout + p * ldd + ch0), g);
#pragma unroll
        for (int k = 0; k < P; ++k) { s0 = fmaf(g[k], s[k], s0); s1 += g[k]; s[k] = g[k] * ww; }
        *(u32x4_t*)(dskip + p * ldk + ch0) = pack16<T>(s);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { atomicAdd(sums, (double)s0); atomicAdd(sums + 1, (double)s1); }
}

// feature-matching loss of the two passes (PackNetSAN01.py:340-342): sum (a - b)^2; backward: db = -2 (a - b) * scale
template <typename T>
__global__ __launch_bounds__(256) void feat_l2_kernel(const T* __restrict__ a, long lda, const T* __restrict__ b, long ldb, double* __restrict__ sum,
                                                      T* __restrict__ db, long ldg, const float* __restrict__ gscale, float inv_n, long npix, int C) {
    constexpr int P = Elem<T>::PER16;
    const int cpr = C / P;
    const long total = npix * cpr;
    const float gs = db ? -2.f * inv_n * (gscale ? *gscale : 1.f) : 0.f;
    float s = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long p = i / cpr;
        const int ch0 = (int)(i - p * cpr) * P;
        float x[P], y[P];
        unpack16<T>(*(const u32x4_t*)(a + p * lda + ch0), x);
        unpack16<T>(*(const u32x4_t*)(b + p * ldb + ch0), y);
#pragma unroll
        for (int k = 0; k < P; ++k) { const float d = x[k] - y[k]; s = fmaf(d, d, s); x[k] = d * gs; }
        if (db) *(u32x4_t*)(db + p * ldg + ch0) = pack16<T>(x);
    }
    if (sum) { s = wave_sum(s); if ((threadIdx.x & 63) == 0) atomicAdd(sum, (double)s); }
}

inline unsigned grid_for(long total) { long g = (total + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1; return (unsigned)g; }


// ---- site list of the active set (round 3: the sparse convolutions run over the ACTIVE pixels only, mte_conv2d_igemm_sparse) ----------
// sites[0 .. count) = linear pixel indices (b*H*W + y*W + x) of the pixels with mask != 0, in raster order.  Three small kernels, no
// atomics: per-block counts (a block = 1024 pixels), an exclusive scan of the counts by one block, the ordered write.
__device__ __forceinline__ int block_excl_scan_256(int v, int* s_w, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + inc - v;
}
__device__ __forceinline__ unsigned load_mask4(const unsigned char* mask, long p, long npix) {
    if (p + 4 <= npix && ((p & 3) == 0)) return *(const unsigned*)(mask + p);
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) if (p + k < npix) r |= (unsigned)mask[p + k] << (8 * k);
    return r;
}
__global__ __launch_bounds__(256) void site_count_kernel(const unsigned char* __restrict__ mask, long npix, int* __restrict__ counts) {
    __shared__ int s_w[4];
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const unsigned m = p < npix ? load_mask4(mask, p, npix) : 0u;
    const int c = ((m & 0xffu) != 0) + ((m & 0xff00u) != 0) + ((m & 0xff0000u) != 0) + ((m & 0xff000000u) != 0);
    int total;
    (void)block_excl_scan_256(c, s_w, total);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void site_scan_kernel(int* __restrict__ counts, int nb, int* __restrict__ count) {
    __shared__ int s_w[4];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const int v = i < nb ? counts[i] : 0;
        int total;
        const int ex = block_excl_scan_256(v, s_w, total);
        const int carry = s_carry;
        if (i < nb) counts[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) count[0] = s_carry;
}
__global__ __launch_bounds__(256) void site_write_kernel(const unsigned char* __restrict__ mask, long npix, const int* __restrict__ offsets, int* __restrict__ sites) {
    __shared__ int s_w[4];
    const long p = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const unsigned m = p < npix ? load_mask4(mask, p, npix) : 0u;
    const int c = ((m & 0xffu) != 0) + ((m & 0xff00u) != 0) + ((m & 0xff0000u) != 0) + ((m & 0xff000000u) != 0);
    int total;
    int o = offsets[blockIdx.x] + block_excl_scan_256(c, s_w, total);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((m >> (8 * k)) & 0xffu) sites[o++] = (int)(p + k);
}

}  // namespace

extern "C" {

int mte_sparsify_depth(const float* depth, void* feat, long ldf, unsigned char* mask, int B, int H, int W, int dtype, hipStream_t stream) {
    if (!depth || !feat || !mask || B <= 0 || H <= 0 || W <= 0 || ldf < 8) return MTE_ERR_ARG;
    const long npix = (long)B * H * W;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(sparsify_depth_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, stream, depth, (bf16_t*)feat, ldf, mask, npix);
    else hipLaunchKernelGGL(sparsify_depth_kernel<float>, dim3(grid_for(npix)), dim3(256), 0, stream, depth, (float*)feat, ldf, mask, npix);
    return mte_check_launch();
}

// sites[0 .. *count) = raster-ordered pixel indices of the active set (mask != 0) of npix = B*H*W pixels; `count` (1 int) and `sites`
// (npix ints) in device memory; ws: mte_sparse_site_list_workspace_elems(npix) ints of scratch
long mte_sparse_site_list_workspace_elems(long npix) { return npix > 0 ? (npix + 1023) / 1024 : 0; }
int mte_sparse_site_list(const unsigned char* mask, long npix, int* sites, int* count, int* ws, hipStream_t stream) {
    (void)hipGetLastError();
    if (!mask || !sites || !count || !ws || npix <= 0 || npix > 0x7fffffffL) return MTE_ERR_ARG;
    const int nb = (int)((npix + 1023) / 1024);
    hipLaunchKernelGGL(site_count_kernel, dim3(nb), dim3(256), 0, stream, mask, npix, ws);
    hipLaunchKernelGGL(site_scan_kernel, dim3(1), dim3(256), 0, stream, ws, nb, count);
    hipLaunchKernelGGL(site_write_kernel, dim3(nb), dim3(256), 0, stream, mask, npix, (const int*)ws, sites);
    return mte_check_launch();
}

int mte_sparse_bn_relu(const void* a, long lda, const void* b, long ldb, const void* c, long ldc, const unsigned char* mask,
                       const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                       void* out, long ldo, long npix, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!a || !mask || !gamma || !beta || !mean || !var || !out || npix <= 0 || C <= 0 || C % per16 != 0) return MTE_ERR_ARG;
    const unsigned g = grid_for(npix * (C / per16));
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(sparse_bn_relu_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (const bf16_t*)c, ldc,
                           mask, gamma, beta, mean, var, eps, (bf16_t*)out, ldo, npix, C);
    else
        hipLaunchKernelGGL(sparse_bn_relu_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)a, lda, (const float*)b, ldb, (const float*)c, ldc,
                           mask, gamma, beta, mean, var, eps, (float*)out, ldo, npix, C);
    return mte_check_launch();
}

int mte_sparse_maxpool3s2(const void* in, long ldi, const unsigned char* mask_in, void* out, long ldo, unsigned char* mask_out,
                          int B, int H, int W, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!in || !mask_in || !out || !mask_out || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C % per16 != 0) return MTE_ERR_ARG;
    const unsigned g = grid_for((long)B * (H / 2) * (W / 2) * (C / per16));
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(sparse_maxpool_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, (const bf16_t*)in, ldi, mask_in, (bf16_t*)out, ldo, mask_out, B, H, W, C);
    else
        hipLaunchKernelGGL(sparse_maxpool_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)in, ldi, mask_in, (float*)out, ldo, mask_out, B, H, W, C);
    return mte_check_launch();
}

int mte_san_fuse(const void* skip, long ld_skip, const void* sparse, long ld_sparse, const float* w, const float* bias,
                 void* out, long ldo, long npix, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!skip || !sparse || !w || !bias || !out || npix <= 0 || C % per16 != 0) return MTE_ERR_ARG;
    const unsigned g = grid_for(npix * (C / per16));
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(san_fuse_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, (const bf16_t*)skip, ld_skip, (const bf16_t*)sparse, ld_sparse, w, bias, (bf16_t*)out, ldo, npix, C);
    else
        hipLaunchKernelGGL(san_fuse_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)skip, ld_skip, (const float*)sparse, ld_sparse, w, bias, (float*)out, ldo, npix, C);
    return mte_check_launch();
}

// ---- training path (see the kernel comments above)
static inline bool chunks_ok(int C, int per16) { const int cpr = C / per16; return C % per16 == 0 && cpr >= 1 && cpr <= 256 && 256 % cpr == 0; }

int mte_sparse_bn_stats(const void* a, long lda, const void* b, long ldb, const void* c, long ldc, const unsigned char* mask,
                        double* sums, long npix, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!a || !mask || !sums || npix <= 0 || !chunks_ok(C, per16)) return MTE_ERR_ARG;
    if (mte_memset_async(sums, 0, sizeof(double) * (2 * C + 1), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    const int ppb = 256 / (C / per16);
    long g = (npix + ppb - 1) / ppb; if (g > 1024) g = 1024;
    const size_t lds = sizeof(float) * 2 * C;
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(sparse_bn_stats_kernel<bf16_t>, dim3((unsigned)g), dim3(256), lds, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (const bf16_t*)c, ldc, mask, sums, npix, C);
    else
        hipLaunchKernelGGL(sparse_bn_stats_kernel<float>, dim3((unsigned)g), dim3(256), lds, stream, (const float*)a, lda, (const float*)b, ldb, (const float*)c, ldc, mask, sums, npix, C);
    return mte_check_launch();
}

int mte_sparse_bn_relu_bwd(const void* a, long lda, const void* b, long ldb, const void* c, long ldc, const void* out, long ldo,
                           const void* dout, long ldd, const unsigned char* mask, const float* gamma, const float* mean, const float* invstd,
                           const double* n_active, double* sums, void* dx, long ldx, long npix, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!a || !out || !dout || !mask || !gamma || !mean || !invstd || !n_active || !sums || !dx || npix <= 0 || !chunks_ok(C, per16)) return MTE_ERR_ARG;
    if (mte_memset_async(sums, 0, sizeof(double) * 2 * C, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    const int ppb = 256 / (C / per16);
    long g = (npix + ppb - 1) / ppb; if (g > 1024) g = 1024;
    const size_t lds = sizeof(float) * 2 * C;
    const unsigned ga = grid_for(npix * (C / per16));
    if (dtype == MTE_DT_BF16) {
        hipLaunchKernelGGL(sparse_bn_bwd_reduce_kernel<bf16_t>, dim3((unsigned)g), dim3(256), lds, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (const bf16_t*)c, ldc,
                           (const bf16_t*)out, ldo, (const bf16_t*)dout, ldd, mean, invstd, sums, npix, C);
        hipLaunchKernelGGL(sparse_bn_bwd_apply_kernel<bf16_t>, dim3(ga), dim3(256), 0, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (const bf16_t*)c, ldc,
                           (const bf16_t*)out, ldo, (const bf16_t*)dout, ldd, mask, gamma, mean, invstd, sums, n_active, (bf16_t*)dx, ldx, npix, C);
    } else {
        hipLaunchKernelGGL(sparse_bn_bwd_reduce_kernel<float>, dim3((unsigned)g), dim3(256), lds, stream, (const float*)a, lda, (const float*)b, ldb, (const float*)c, ldc,
                           (const float*)out, ldo, (const float*)dout, ldd, mean, invstd, sums, npix, C);
        hipLaunchKernelGGL(sparse_bn_bwd_apply_kernel<float>, dim3(ga), dim3(256), 0, stream, (const float*)a, lda, (const float*)b, ldb, (const float*)c, ldc,
                           (const float*)out, ldo, (const float*)dout, ldd, mask, gamma, mean, invstd, sums, n_active, (float*)dx, ldx, npix, C);
    }
    return mte_check_launch();
}

int mte_sparse_maxpool3s2_bwd(const void* in, long ldi, const unsigned char* mask_in, const void* dout, long ldd, void* din, long ldn,
                              int B, int H, int W, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!in || !mask_in || !dout || !din || B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || C % per16 != 0) return MTE_ERR_ARG;
    const unsigned g = grid_for((long)B * H * W * (C / per16));
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(sparse_maxpool_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, stream, (const bf16_t*)in, ldi, mask_in, (const bf16_t*)dout, ldd, (bf16_t*)din, ldn, B, H, W, C);
    else
        hipLaunchKernelGGL(sparse_maxpool_bwd_kernel<float>, dim3(g), dim3(256), 0, stream, (const float*)in, ldi, mask_in, (const float*)dout, ldd, (float*)din, ldn, B, H, W, C);
    return mte_check_launch();
}

int mte_san_fuse_bwd(const void* skip, long ld_skip, const void* dout, long ldd, const float* w, void* dskip, long ldk, double* sums,
                     long npix, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!skip || !dout || !w || !dskip || !sums || npix <= 0 || C % per16 != 0) return MTE_ERR_ARG;
    if (mte_memset_async(sums, 0, sizeof(double) * 2, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    long g = (npix * (C / per16) + 255) / 256; if (g > 1024) g = 1024;
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(san_fuse_bwd_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)skip, ld_skip, (const bf16_t*)dout, ldd, w, (bf16_t*)dskip, ldk, sums, npix, C);
    else
        hipLaunchKernelGGL(san_fuse_bwd_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)skip, ld_skip, (const float*)dout, ldd, w, (float*)dskip, ldk, sums, npix, C);
    return mte_check_launch();
}

int mte_feat_l2(const void* a, long lda, const void* b, long ldb, double* sum, void* db, long ldg, const float* gscale, float inv_n,
                long npix, int C, int dtype, hipStream_t stream) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (!a || !b || (!sum && !db) || npix <= 0 || C % per16 != 0) return MTE_ERR_ARG;
    if (sum && mte_memset_async(sum, 0, sizeof(double), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    long g = (npix * (C / per16) + 255) / 256; if (g > 1024) g = 1024;
    if (dtype == MTE_DT_BF16)
        hipLaunchKernelGGL(feat_l2_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, sum, (bf16_t*)db, ldg, gscale, inv_n, npix, C);
    else
        hipLaunchKernelGGL(feat_l2_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, (const float*)a, lda, (const float*)b, ldb, sum, (float*)db, ldg, gscale, inv_n, npix, C);
    return mte_check_launch();
}

}  // extern "C"
