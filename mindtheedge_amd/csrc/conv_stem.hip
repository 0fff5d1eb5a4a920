// Stem convolution: 8 input channels (the rgb image, zero-padded 3 -> 8), k x k taps, C_out <= 32 -- gfx950, bf16.
// Reference op: encoder.pre_calc = Conv2D(3, 32, 5, 1) of PackNetSlimEnc01 (networks/depth/PackNetSAN01.py:27,
// networks/layers/packnet/layers01.py:29-31), forward and weight gradient (the image needs no data gradient).
//
// Why a third conv path: the LDS-patch kernels (conv_patch.hip) stage 32-channel slices, so an 8-channel input fills a quarter of
// every MFMA's K range -- the stem ran at 250 TFLOP/s forward and 204 TFLOP/s in the weight gradient, 4x the MFMA work its
// 1.18 GMAC per image need and far above its HBM time (63 MB in, 252 MB out per 8-frame batch).  With 8 channels a pixel IS one
// 16-byte chunk, so the reduction index can run over (tap, channel) with the tap as the chunk index: one v_mfma_f32_32x32x16_bf16
// covers two taps, 13 of them the whole 5 x 5 x 8 reduction (200 of 208 K positions used).
//   forward : A = the (16+k-1) x (32+k-1)-pixel patch of the tile in LDS, one ds_read_b128 per lane per MFMA at the tap's pixel
//             shift; B = the layer's whole weight tensor as 13 fragment registers per lane, read once per workgroup straight from
//             the generic [N][taps][8] pack (a lane's fragment is 16 contiguous bytes of it).  Write-bound: 64 B out per pixel.
//   wgrad   : dW[n][(tap, c)] = sum_px dy[px][n] * x[px + tap][c]: M = 32 output channels, 7 column blocks of 4 taps x 8 channels,
//             K = pixels; both operands are pixel-major, so fragments come from ds_read_b64_tr_b16 (as conv_patch_wgrad_kernel);
//             the blocks are dealt to the four waves, one partial slab per workgroup.
#include "common.hpp"

namespace {

constexpr int TW = 32;

struct StemArgs {
    const bf16_t* x; long ldx;                     // [B,H,W] pixels, 8 channels (ldx >= 8 elements per pixel)
    const bf16_t* wf;                              // generic forward pack [N][taps][8]
    const float* bias;
    bf16_t* y; long ldy;
    const bf16_t* dy; long lddy;                   // wgrad
    float* dw; long part_stride; int groups;       // wgrad: stage [N][taps][8] fp32, one slab per workgroup (part_stride) or atomics (0)
    int B, H, W, N;
};

template <int K>
__global__ __launch_bounds__(256, 2) void conv_stem_fwd_kernel(StemArgs a) {
    constexpr int TH = 16, PAD = K / 2, PH = TH + K - 1, PW = TW + K - 1, TAPS = K * K;
    constexpr int NKK = (TAPS + 1) / 2;                            // MFMAs (16 reduction positions = 2 taps x 8 channels) per output tile
    constexpr int PPX = PH * PW, NCH = (PPX + 255) / 256;
    constexpr int OBYTES = TH * TW * 64, PBYTES = PPX * 16;
    __shared__ __attribute__((aligned(16))) char smem[OBYTES > PBYTES ? OBYTES : PBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = a.W / TW, tiles_y = (a.H + TH - 1) / TH;
    int id = xcd_remap(blockIdx.x, tiles_x * tiles_y * a.B);
    const int tx_ = id % tiles_x; id /= tiles_x;
    const int ty_ = id % tiles_y; const int b = id / tiles_y;
    const int x0 = tx_ * TW, y0 = ty_ * TH;

    // the patch: one 16-byte pixel per thread and pass, zero outside the image
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int p = tid + i * 256;
        if (PPX % 256 == 0 || p < PPX) {
            const int py = p / PW, px = p - py * PW;
            const int iy = y0 + py - PAD, ix = x0 + px - PAD;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = *(const u32x4_t*)(a.x + (((long)b * a.H + iy) * a.W + ix) * a.ldx);
            *(u32x4_t*)(smem + p * 16) = v;
        }
    }
    // every weight of the layer as fragments: lane (n = r, h) of MFMA kk holds w[n][tap = 2 kk + h][0..7]
    u32x4_t bq[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int t = 2 * kk + h;
        bq[kk] = (t < TAPS && r < a.N) ? *(const u32x4_t*)(a.wf + ((long)r * TAPS + t) * 8) : u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    constexpr int MM = TH / 4;                                     // pixel rows per wave
    f32x16_t acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int t = min(2 * kk + h, TAPS - 1);                   // (the odd tap past the end multiplies zero weights: any address will do)
        const int dy = t / K, dx = t - dy * K;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const u32x4_t fa = *(const u32x4_t*)(smem + ((wave * MM + m + dy) * PW + r + dx) * 16);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, bq[kk]), acc[m], 0, 0, 0);
        }
    }
    __syncthreads();                                               // every wave is done with the patch: reuse it as the output staging
    {
        const float bv = (a.bias && r < a.N) ? a.bias[r] : 0.f;
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int px = (e & 3) + 8 * (e >> 2) + 4 * h;     // D[row = pixel][col = channel r]
                *(bf16_t*)(smem + ((wave * MM + m) * TW + px) * 64 + r * 2) = f2bf(acc[m][e] + bv);
            }
    }
    __syncthreads();
    const int cpp = a.N >> 3;                                      // valid 16-byte chunks per pixel
#pragma unroll
    for (int i = 0; i < TH * TW * 4 / 256; ++i) {
        const int idc = tid + i * 256;
        const int pix = idc >> 2, c = idc & 3;
        const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
        if (yy < a.H && c < cpp) *(u32x4_t*)(a.y + (((long)b * a.H + yy) * a.W + xx) * a.ldy + c * 8) = *(const u32x4_t*)(smem + pix * 64 + c * 16);
    }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(StemArgs a) {
    constexpr int TH = 8, PAD = K / 2, PH = TH + K - 1, PW = TW + K - 1, TAPS = K * K;
    constexpr int NBLK = (TAPS + 3) / 4;                           // column blocks of 4 taps x 8 channels
    constexpr int BPW = (NBLK + 3) / 4;                            // blocks per wave
    constexpr int PPX = PH * PW, NXC = (PPX + 255) / 256;
    constexpr int YRS = 64, NYC = TH * TW * 4 / 256;
    __shared__ __attribute__((aligned(16))) char X[PPX * 16];
    __shared__ __attribute__((aligned(16))) char Y[TH * TW * YRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = a.W / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.B;
    const int per = (ntiles + a.groups - 1) / a.groups;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    const int npp = a.N >> 3;

    u32x4_t sx[NXC], sy[NYC];
    auto load_tile = [&](int tile) {
        int id = tile;
        const int tx_ = id % tiles_x; id /= tiles_x;
        const int ty_ = id % tiles_y; const int b = id / tiles_y;
        const int x0 = tx_ * TW, y0 = ty_ * TH;
#pragma unroll
        for (int i = 0; i < NXC; ++i) {
            const int p = tid + i * 256;
            const int py = p / PW, px = p - py * PW;
            const int iy = y0 + py - PAD, ix = x0 + px - PAD;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if ((PPX % 256 == 0 || p < PPX) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *(const u32x4_t*)(a.x + (((long)b * a.H + iy) * a.W + ix) * a.ldx);
            sx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NYC; ++i) {
            const int idc = tid + i * 256;
            const int pix = idc >> 2, c = idc & 3;
            const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (yy < a.H && c < npp) v = *(const u32x4_t*)(a.dy + (((long)b * a.H + yy) * a.W + xx) * a.lddy + c * 8);
            sy[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NXC; ++i) {
            const int p = tid + i * 256;
            if (PPX % 256 == 0 || p < PPX) *(u32x4_t*)(X + p * 16) = sx[i];
        }
#pragma unroll
        for (int i = 0; i < NYC; ++i) {
            const int idc = tid + i * 256;
            *(u32x4_t*)(Y + (idc >> 2) * YRS + (idc & 3) * 16) = sy[i];
        }
    };

    f32x16_t acc[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // transposing-read lane roles (conv_patch_wgrad_kernel): 16-lane group g: columns 16*(g&1) + 4*pp .., pixels 8*(g>>1) + q (+4 for the 2nd read)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int chb = 16 * (g & 1) + 4 * pp, pxb = 8 * (g >> 1) + q;
    // a column of the x operand is (tap, channel): columns chb .. chb+3 = channels 4*(pp&1) .. +3 of tap 2*(g&1) + (pp>>1) of the block
    const int tl = 2 * (g & 1) + (pp >> 1), cb8 = (pp & 1) * 8;
    int xoff[BPW];                                                 // byte offset of this lane's tap inside the patch, per block of this wave
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int t = min((wave + 4 * i) * 4 + tl, TAPS - 1);      // (columns past the last tap are never stored)
        xoff[i] = ((t / K) * PW + (t % K)) * 16 + cb8;
    }

    if (t_begin < t_end) {
        load_tile(t_begin);
        for (int tile = t_begin; tile < t_end; ++tile) {
            __syncthreads();                                       // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (tile + 1 < t_end) load_tile(tile + 1);             // in flight (registers) while this tile is multiplied
#pragma unroll 1
            for (int ks = 0; ks < TH * 2; ++ks) {                  // 16 pixels of one tile row per k-step
                const int row = ks >> 1, col0 = (ks & 1) * 16;
                const char* yb = Y + (row * TW + col0 + pxb) * YRS + chb * 2;
                s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(yb));
                s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(yb + 4 * YRS));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                const u32x4_t fy = u32x4_t{l2.x, l2.y, h2.x, h2.y};
#pragma unroll
                for (int i = 0; i < BPW; ++i) {
                    if (wave + 4 * i < NBLK) {
                        const char* xb = X + (row * PW + col0 + pxb) * 16 + xoff[i];
                        s16x4_t xl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(xb));
                        s16x4_t xh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(xb + 4 * 16));
                        uint2 a2 = __builtin_bit_cast(uint2, xl), b2 = __builtin_bit_cast(uint2, xh);
                        const u32x4_t fx = u32x4_t{a2.x, a2.y, b2.x, b2.y};
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fy), __builtin_bit_cast(bf16x8_t, fx), acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[row = cout][col = (tap, c)]: col = lane & 31
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
        const int blk = wave + 4 * i;
        const int tap = blk * 4 + (r >> 3), c = r & 7;
        if (blk >= NBLK || tap >= TAPS) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * h;
            if (co < a.N) {
                float* dst = a.dw + (long)blockIdx.x * a.part_stride + ((long)co * TAPS + tap) * 8 + c;
                if (a.part_stride) *dst = acc[i][e]; else atomicAdd(dst, acc[i][e]);
            }
        }
    }
}

inline bool stem_ok(int W, int Cin_p, int N, int KH, int KW) {
    return W % TW == 0 && Cin_p == 8 && N % 8 == 0 && N <= 32 && KH == KW && (KH == 3 || KH == 5 || KH == 7);
}

}  // namespace

extern "C" {

// 1 if the stem kernels cover this conv shape (bf16, exactly 8 input channels, C_out <= 32, W % 32 == 0, k in {3, 5, 7})
int mte_conv2d_stem_supported(int W, int Cin_p, int N, int KH, int KW, int dtype) {
    return (dtype == MTE_DT_BF16 && stem_ok(W, Cin_p, N, KH, KW)) ? 1 : 0;
}

// y = conv(x, w) + bias; wf = the generic forward pack [N][KH*KW][8] of mte_pack_conv_weights (bf16)
int mte_conv2d_stem_fwd(const void* x, long ldx, const void* wf, const float* bias, void* y, long ldy,
                        int B, int H, int W, int N, int KH, int KW, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wf || !y || !stem_ok(W, 8, N, KH, KW) || ldx % 8 != 0) return MTE_ERR_ARG;
    StemArgs a{}; a.x = (const bf16_t*)x; a.ldx = ldx; a.wf = (const bf16_t*)wf; a.bias = bias; a.y = (bf16_t*)y; a.ldy = ldy;
    a.B = B; a.H = H; a.W = W; a.N = N;
    const long tiles = (long)(W / TW) * ((H + 15) / 16) * B;
    switch (KH) {
        case 3: hipLaunchKernelGGL(conv_stem_fwd_kernel<3>, dim3((unsigned)tiles), dim3(256), 0, stream, a); break;
        case 5: hipLaunchKernelGGL(conv_stem_fwd_kernel<5>, dim3((unsigned)tiles), dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(conv_stem_fwd_kernel<7>, dim3((unsigned)tiles), dim3(256), 0, stream, a); break;
    }
    return mte_check_launch();
}

// dw_stage = stage_parts x [N][KH*KW][8] fp32: one partial gradient per workgroup (*parts_out of them; mte_unpack_conv_wgrad adds them
// up), or -- stage_parts too small -- fp32 atomics into part 0 (*parts_out = 1)
int mte_conv2d_stem_wgrad(const void* x, long ldx, const void* dy, long lddy, float* dw_stage, int stage_parts, int* parts_out,
                          int B, int H, int W, int N, int KH, int KW, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dy || !dw_stage || !stem_ok(W, 8, N, KH, KW) || ldx % 8 != 0) return MTE_ERR_ARG;
    StemArgs a{}; a.x = (const bf16_t*)x; a.ldx = ldx; a.dy = (const bf16_t*)dy; a.lddy = lddy; a.dw = dw_stage;
    a.B = B; a.H = H; a.W = W; a.N = N;
    const long ntiles = (long)(W / TW) * ((H + 7) / 8) * B;
    long groups = stage_parts > 1 ? (stage_parts < 2048 ? stage_parts : 2048) : 1024;      // one slab per workgroup where the stage has room
    if (groups > ntiles) groups = ntiles;
    a.groups = (int)groups;
    const long per = (long)N * KH * KW * 8;
    if (groups > 1 && groups <= stage_parts) { a.part_stride = per; if (parts_out) *parts_out = (int)groups; }
    else {
        a.part_stride = 0;
        if (parts_out) *parts_out = 1;
        if (mte_memset_async(dw_stage, 0, sizeof(float) * (size_t)per, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    }
    switch (KH) {
        case 3: hipLaunchKernelGGL(conv_stem_wgrad_kernel<3>, dim3((unsigned)groups), dim3(256), 0, stream, a); break;
        case 5: hipLaunchKernelGGL(conv_stem_wgrad_kernel<5>, dim3((unsigned)groups), dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(conv_stem_wgrad_kernel<7>, dim3((unsigned)groups), dim3(256), 0, stream, a); break;
    }
    return mte_check_launch();
}

}  // extern "C"
