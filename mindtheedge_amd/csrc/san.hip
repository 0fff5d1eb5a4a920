// Sparse auxiliary (SAN) branch of PackNet-SAN for gfx950, inference only.  PARITY UNPINNED: the
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

inline unsigned grid_for(long total) { long g = (total + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1; return (unsigned)g; }

}  // namespace

extern "C" {

int mte_sparsify_depth(const float* depth, void* feat, long ldf, unsigned char* mask, int B, int H, int W, int dtype, hipStream_t stream) {
    if (!depth || !feat || !mask || B <= 0 || H <= 0 || W <= 0 || ldf < 8) return MTE_ERR_ARG;
    const long npix = (long)B * H * W;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(sparsify_depth_kernel<bf16_t>, dim3(grid_for(npix)), dim3(256), 0, stream, depth, (bf16_t*)feat, ldf, mask, npix);
    else hipLaunchKernelGGL(sparsify_depth_kernel<float>, dim3(grid_for(npix)), dim3(256), 0, stream, depth, (float*)feat, ldf, mask, npix);
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

}  // extern "C"
