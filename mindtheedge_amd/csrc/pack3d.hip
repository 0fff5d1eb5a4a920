// 3-D packing / unpacking stencils for gfx950 (NHWC activations).
//
// PackLayerConv3d  (layers01.py:214-248): x[B,C,H,W] -> pixel-unshuffle(2) -> Conv3d(1->4, 3x3x3, pad 1) over
//   (packed channel d = 4c + 2*dy + dx, H/2, W/2) -> view [B, 16C, H/2, W/2] with channel = f*4C + d.
// UnpackLayerConv3d (layers01.py:251-287): x[B,C,H,W] -> Conv3d(1->4) over (c, H, W) -> channel q = f*C + c
//   -> PixelShuffle(2): out[b, q>>2, 2h + ((q&3)>>1), 2w + (q&1)].
// The space-to-depth / depth-to-space permutes are folded into the stencil's addressing, so neither the
// packed tensor nor the un-shuffled 4C tensor ever exists in HBM.
//
// Work split: one thread = one (volume pixel, block of 8 NHWC channels).  Lanes of one pixel are adjacent
// (C/8 is a power of two <= 64), so the +-1 neighbours along the conv3d "depth" axis come from the
// adjacent lane by a wave shuffle instead of a second, unaligned load.  27-tap fp32 FMA stencil; the 108
// weights + 4 biases are wave-uniform scalar loads.  These kernels move 2..10 bytes per 27 FMAs: they sit
// between the HBM and the fp32-VALU roofs, not on MFMA (N = 4 features cannot fill a matrix tile).
#include "common.hpp"
#ifndef MTE_P3W_WGS
#define MTE_P3W_WGS 512      // workgroups of the conv3d weight-gradient launches (side queue; end of round 5, after the kernel's instruction diet, same box: 768 -> 22.00 ms per step, 512 -> 21.94, 384 -> 21.97, 256 -> 22.09)
#endif


namespace {

template <typename T> __device__ __forceinline__ void load8(const T* p, float* v);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float* v) { unpack16<bf16_t>(*(const u32x4_t*)p, v); }
template <> __device__ __forceinline__ void load8<float>(const float* p, float* v) {
    unpack16<float>(*(const u32x4_t*)p, v); unpack16<float>(*(const u32x4_t*)(p + 4), v + 4);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float* v) { *(u32x4_t*)p = pack16<bf16_t>(v); }
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
    *(u32x4_t*)p = pack16<float>(v); *(u32x4_t*)(p + 4) = pack16<float>(v + 4);
}

struct P3Args {
    const void* x; long ldx;       // un-packed side  [B,H,W,C]      (pack: input;  unpack: input of conv3d)
    const void* o; long ldo;       // feature side    pack: [B,H/2,W/2,16C]; unpack: [B,2H,2W,C] (after shuffle)
    void* dst; long lddst;         // output of this launch
    const float* w3; const float* b3;   // [4][3][3][3], [4]
    float* dw3; float* db3;        // [108], [4] (atomic accumulate)
    int B, H, W, C;                // dims of the UN-PACKED side tensor x
    long total;                    // threads with work
};

// ------------------------------------------------------------------------------------------------
// PACK.  Volume pixel (h,w) in [H/2, W/2]; a thread owns CPT x-channels c0.. -> DB = 4*CPT packed depths d = 4c+s.
// pv[1 + 4*ci + s] = x[b, 2h'+(s>>1), 2w'+(s&1), c0+ci];  pv[0] = depth 4*c0-1, pv[DB+1] = depth 4*c0+DB.
// CPT = 4 (8-byte loads, 16 depths, all four 3-D features in one pass) is used when C/4 <= 64 lanes so the depth
// halo stays a wave shuffle; CPT = 8 otherwise (C = 512).
// ------------------------------------------------------------------------------------------------
template <typename T, int CPT> __device__ __forceinline__ void loadc(const T* p, float* v);
template <> __device__ __forceinline__ void loadc<bf16_t, 8>(const bf16_t* p, float* v) { load8<bf16_t>(p, v); }
template <> __device__ __forceinline__ void loadc<float, 8>(const float* p, float* v) { load8<float>(p, v); }
template <> __device__ __forceinline__ void loadc<bf16_t, 4>(const bf16_t* p, float* v) {
    const uint2 c = *(const uint2*)p;
    v[0] = __uint_as_float(c.x << 16); v[1] = __uint_as_float(c.x & 0xffff0000u);
    v[2] = __uint_as_float(c.y << 16); v[3] = __uint_as_float(c.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void loadc<float, 4>(const float* p, float* v) { unpack16<float>(*(const u32x4_t*)p, v); }
template <typename T, int CPT> __device__ __forceinline__ void storec(T* p, const float* v);
template <> __device__ __forceinline__ void storec<bf16_t, 8>(bf16_t* p, const float* v) { store8<bf16_t>(p, v); }
template <> __device__ __forceinline__ void storec<float, 8>(float* p, const float* v) { store8<float>(p, v); }
template <> __device__ __forceinline__ void storec<bf16_t, 4>(bf16_t* p, const float* v) { *(uint2*)p = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3])); }
template <> __device__ __forceinline__ void storec<float, 4>(float* p, const float* v) { *(u32x4_t*)p = pack16<float>(v); }

template <typename T, int CPT>
__device__ __forceinline__ void pack_load_window(const P3Args& a, int b, int hh, int ww, int c0, int j, int cb, bool live, float* pv) {
    constexpr int DB = 4 * CPT;
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const bool in = live && (unsigned)hh < (unsigned)H2 && (unsigned)ww < (unsigned)W2;
    float last3 = 0.f, first0 = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[CPT];
        if (in) loadc<T, CPT>((const T*)a.x + (((long)b * a.H + 2 * hh + (s >> 1)) * a.W + 2 * ww + (s & 1)) * a.ldx + c0, v);
        else {
#pragma unroll
            for (int i = 0; i < CPT; ++i) v[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) pv[1 + 4 * i + s] = v[i];
        if (s == 3) last3 = v[CPT - 1];
        if (s == 0) first0 = v[0];
    }
    const float lo = __shfl_up(last3, 1, 64), hi = __shfl_down(first0, 1, 64);
    pv[0] = j > 0 ? lo : 0.f;
    pv[DB + 1] = j < cb - 1 ? hi : 0.f;
}

#define P3_THREAD_MAP_N(HV, WV, CPT_)                                          \
    const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;              \
    const int cb = a.C / (CPT_);                                               \
    const bool live = gid < a.total;                                           \
    const long g = live ? gid : a.total - 1;                                   \
    const int j = (int)(g % cb);                                               \
    long pix = g / cb;                                                         \
    const int w = (int)(pix % (WV)); pix /= (WV);                              \
    const int h = (int)(pix % (HV));                                           \
    const int b = (int)(pix / (HV));                                           \
    const int c0 = j * (CPT_);
#define P3_THREAD_MAP(HV, WV) P3_THREAD_MAP_N(HV, WV, 8)

template <typename T, int CPT>
__global__ __launch_bounds__(256, 3) void pack3d_fwd_kernel(P3Args a) {
    constexpr int DB = 4 * CPT, FP = CPT == 8 ? 2 : 4;           // features per pass (register budget)
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    P3_THREAD_MAP_N(H2, W2, CPT);
    const int D = a.C * 4;
    T* op = (T*)a.dst + (((long)b * H2 + h) * W2 + w) * a.lddst;
#pragma unroll 1
    for (int f0 = 0; f0 < 4; f0 += FP) {
        float acc[FP][DB];
#pragma unroll
        for (int f = 0; f < FP; ++f) {
            const float bv = a.b3[f0 + f];
#pragma unroll
            for (int i = 0; i < DB; ++i) acc[f][i] = bv;
        }
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - 3 * kh;
            float pv[DB + 2];
            pack_load_window<T, CPT>(a, b, h + kh - 1, w + kw - 1, c0, j, cb, live, pv);
#pragma unroll
            for (int f = 0; f < FP; ++f)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = a.w3[(((f0 + f) * 3 + kd) * 3 + kh) * 3 + kw];
#pragma unroll
                    for (int i = 0; i < DB; ++i) acc[f][i] = fmaf(wv, pv[i + kd], acc[f][i]);
                }
        }
        if (live) {
#pragma unroll
            for (int f = 0; f < FP; ++f)
#pragma unroll
                for (int k = 0; k < DB / 8; ++k) store8<T>(op + (f0 + f) * D + 4 * c0 + 8 * k, &acc[f][8 * k]);
        }
    }
}

// dP(d,h,w) = sum_f sum_taps w3[f][kd][kh][kw] * dO[f][d-kd+1][h-kh+1][w-kw+1];  scatter back to x layout
template <typename T, int CPT>
__global__ __launch_bounds__(256) void pack3d_bwd_data_kernel(P3Args a) {
    constexpr int DB = 4 * CPT;
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    P3_THREAD_MAP_N(H2, W2, CPT);
    const int D = a.C * 4;
    float acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) acc[i] = 0.f;
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t - 3 * kh;
        const int hh = h - kh + 1, ww = w - kw + 1;
        const bool in = live && (unsigned)hh < (unsigned)H2 && (unsigned)ww < (unsigned)W2;
#pragma unroll 2
        for (int f = 0; f < 4; ++f) {
            float pv[DB + 2];
            const T* src = (const T*)a.o + (((long)b * H2 + hh) * W2 + ww) * a.ldo + f * D + 4 * c0;
#pragma unroll
            for (int k = 0; k < DB / 8; ++k) {
                if (in) load8<T>(src + 8 * k, &pv[1 + 8 * k]);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv[1 + 8 * k + i] = 0.f;
                }
            }
            const float lo = __shfl_up(pv[DB], 1, 64), hi = __shfl_down(pv[1], 1, 64);
            pv[0] = j > 0 ? lo : 0.f;
            pv[DB + 1] = j < cb - 1 ? hi : 0.f;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float wv = a.w3[((f * 3 + kd) * 3 + kh) * 3 + kw];
                // dP[d] += w[kd] * dO[d - kd + 1]  -> window index (i + 1) - kd + 1 = i + 2 - kd
#pragma unroll
                for (int i = 0; i < DB; ++i) acc[i] = fmaf(wv, pv[i + 2 - kd], acc[i]);
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) v[i] = acc[4 * i + s];
        storec<T, CPT>((T*)a.dst + (((long)b * a.H + 2 * h + (s >> 1)) * a.W + 2 * w + (s & 1)) * a.lddst + c0, v);
    }
}

// vals[FS*10]: per feature [kd*3+kw] for fixed kh, [9] = bias sum -> wave sums -> LDS -> one atomic per value per block
template <int FS>
__device__ __forceinline__ void block_reduce_atomic_fs(float (&vals)[FS * 10], int f0, int kh, float* dst, float* sred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < FS * 10; ++i) {
        const float s = wave_sum(vals[i]);
        if (lane == 0) sred[wave * FS * 10 + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < FS * 10) {
        const int f = f0 + threadIdx.x / 10, i = threadIdx.x % 10;
        float s = 0.f;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) s += sred[wv * FS * 10 + threadIdx.x];
        if (i < 9) atomicAdd(dst + ((f * 3 + i / 3) * 3 + kh) * 3 + (i % 3), s);
        else if (kh == 1) atomicAdd(dst + 108 + f, s);
    }
    __syncthreads();
}

// dw3[f][kd][kh][kw] = sum dO[f][d][h][w] * P(d+kd-1, h+kh-1, w+kw-1);  db3[f] = sum dO[f].  grid.y = kh.
template <typename T, int CPT>
__global__ __launch_bounds__(256, 3) void pack3d_bwd_weight_kernel(P3Args a) {
    constexpr int DB = 4 * CPT, FS = CPT == 8 ? 1 : 2;             // features per sweep over the pixels (register budget)
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    __shared__ float sred[4 * FS * 10];
    const int kh = blockIdx.y;
    const int cb = a.C / CPT;
    const int D = a.C * 4;
    const long nthreads = (long)gridDim.x * blockDim.x;
    const long iters = (a.total + nthreads - 1) / nthreads;
#pragma unroll 1
    for (int f0 = 0; f0 < 4; f0 += FS) {
        float acc[FS * 10];
#pragma unroll
        for (int i = 0; i < FS * 10; ++i) acc[i] = 0.f;
#pragma unroll 1
        for (long it = 0; it < iters; ++it) {
            const long gid = it * nthreads + blockIdx.x * (long)blockDim.x + threadIdx.x;
            const bool live = gid < a.total;
            const long g = live ? gid : a.total - 1;
            const int j = (int)(g % cb);
            long pix = g / cb;
            const int w = (int)(pix % W2); pix /= W2;
            const int h = (int)(pix % H2);
            const int b = (int)(pix / H2);
            const int c0 = j * CPT;
            const T* src = (const T*)a.o + (((long)b * H2 + h) * W2 + w) * a.ldo + 4 * c0;
            float go[FS][DB];
#pragma unroll
            for (int f = 0; f < FS; ++f) {
#pragma unroll
                for (int k = 0; k < DB / 8; ++k) {
                    if (live) load8<T>(src + (f0 + f) * D + 8 * k, &go[f][8 * k]);
                    else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) go[f][8 * k + i] = 0.f;
                    }
                }
                float sb = 0.f;
#pragma unroll
                for (int i = 0; i < DB; ++i) sb += go[f][i];
                acc[f * 10 + 9] += sb;
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                float pv[DB + 2];
                pack_load_window<T, CPT>(a, b, h + kh - 1, w + kw - 1, c0, j, cb, live, pv);
#pragma unroll
                for (int f = 0; f < FS; ++f)
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        float sacc = 0.f;
#pragma unroll
                        for (int i = 0; i < DB; ++i) sacc = fmaf(go[f][i], pv[i + kd], sacc);
                        acc[f * 10 + kd * 3 + kw] += sacc;
                    }
            }
        }
        block_reduce_atomic_fs<FS>(acc, f0, kh, a.dw3, sred);
    }
}

// vals[40] per thread: [0..35] = dw3[f][kd][kh fixed][kw] at i = (f*3+kd)*3+kw, [36..39] = db3[f] (kh == 1 blocks only)
__device__ __forceinline__ void block_reduce_atomic(float (&vals)[40], int kh, float* dst, float* sred) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 40; ++i) {
        const float s = wave_sum(vals[i]);
        if (lane == 0) sred[wave * 40 + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 40) {
        const int i = threadIdx.x;
        float s = 0.f;
        for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) s += sred[wv * 40 + i];
        if (i < 36) atomicAdd(dst + (i / 3) * 9 + kh * 3 + (i % 3), s);
        else if (kh == 1) atomicAdd(dst + 108 + (i - 36), s);
    }
}

// ------------------------------------------------------------------------------------------------
// UNPACK.  Volume = x itself: depth = channel c (C of them), pixel (h,w) in [H,W]; thread block of 8 depths.
// feature side tensor o = [B,2H,2W,C] after the pixel shuffle: o3[f][c] lives at channel q>>2 of output
// pixel (2h + ((q&3)>>1), 2w + (q&1)), q = f*C + c.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void unpack_load_window(const P3Args& a, int b, int hh, int ww, int c0, int j, int cb, bool live, float* pv) {
    const bool in = live && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
    if (in) load8<T>((const T*)a.x + (((long)b * a.H + hh) * a.W + ww) * a.ldx + c0, &pv[1]);
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[1 + i] = 0.f;
    }
    const float lo = __shfl_up(pv[8], 1, 64), hi = __shfl_down(pv[1], 1, 64);
    pv[0] = j > 0 ? lo : 0.f;
    pv[9] = j < cb - 1 ? hi : 0.f;
}

// feature-side gather/scatter of the 8 values o3[f][c0..c0+7] at volume pixel (h,w)
template <typename T>
__device__ __forceinline__ void shuffled_ptrs(const P3Args& a, const void* base, long ld, int b, int h, int w, int f, int c0, const T** p) {
    const int q0 = f * a.C + c0;            // multiple of 8 -> channels q0>>2 and (q0>>2)+1, 4 positions each
#pragma unroll
    for (int s = 0; s < 4; ++s)
        p[s] = (const T*)base + (((long)b * 2 * a.H + 2 * h + (s >> 1)) * (2 * a.W) + 2 * w + (s & 1)) * ld + (q0 >> 2);
}

template <typename T>
__global__ __launch_bounds__(256, 3) void unpack3d_fwd_kernel(P3Args a) {
    P3_THREAD_MAP(a.H, a.W);
    float acc[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const float bv = a.b3[f];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[f][i] = bv;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            float pv[10];
            unpack_load_window<T>(a, b, h + kh - 1, w + kw - 1, c0, j, cb, live, pv);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = a.w3[((f * 3 + kd) * 3 + kh) * 3 + kw];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[f][i] = fmaf(wv, pv[i + kd], acc[f][i]);
                }
        }
    if (!live) return;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const T* p[4];
        shuffled_ptrs<T>(a, a.dst, a.lddst, b, h, w, f, c0, p);
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // q = q0 + i: position s = i&3, channel (q0>>2) + (i>>2)
            T* d = (T*)p[s];
            if constexpr (sizeof(T) == 2) *(unsigned*)d = pack2bf(acc[f][s], acc[f][4 + s]);
            else *(float2*)d = make_float2(acc[f][s], acc[f][4 + s]);
        }
    }
}

template <typename T>
__device__ __forceinline__ void unpack_load_feat(const P3Args& a, int b, int hh, int ww, int f, int c0, int j, int cb, bool live, float* pv) {
    const bool in = live && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
    if (in) {
        const T* p[4];
        shuffled_ptrs<T>(a, a.o, a.ldo, b, hh, ww, f, c0, p);
#pragma unroll
        for (int s = 0; s < 4; ++s) {                 // two adjacent channels per shuffled position: one 4-byte (bf16) load
            if constexpr (sizeof(T) == 2) {
                const unsigned u = *(const unsigned*)p[s];
                pv[1 + s] = __uint_as_float(u << 16); pv[1 + 4 + s] = __uint_as_float(u & 0xffff0000u);
            } else {
                const float2 u = *(const float2*)p[s];
                pv[1 + s] = u.x; pv[1 + 4 + s] = u.y;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[1 + i] = 0.f;
    }
    const float lo = __shfl_up(pv[8], 1, 64), hi = __shfl_down(pv[1], 1, 64);
    pv[0] = j > 0 ? lo : 0.f;
    pv[9] = j < cb - 1 ? hi : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void unpack3d_bwd_data_kernel(P3Args a) {
    P3_THREAD_MAP(a.H, a.W);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll 1
    for (int t = 0; t < 9; ++t)
#pragma unroll 1
            for (int f = 0; f < 4; ++f) {
                const int kh = t / 3, kw = t - 3 * kh;
                float pv[10];
                unpack_load_feat<T>(a, b, h - kh + 1, w - kw + 1, f, c0, j, cb, live, pv);
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = a.w3[((f * 3 + kd) * 3 + kh) * 3 + kw];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(wv, pv[i + 2 - kd], acc[i]);
                }
            }
    if (!live) return;
    store8<T>((T*)a.dst + (((long)b * a.H + h) * a.W + w) * a.lddst + c0, acc);
}

template <typename T>
__global__ __launch_bounds__(256, 3) void unpack3d_bwd_weight_kernel(P3Args a) {
    __shared__ float sred[4 * 40];
    float acc[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = 0.f;
    const int kh = blockIdx.y;
    const int cb = a.C >> 3;
    const long nthreads = (long)gridDim.x * blockDim.x;   // grid.x only; grid.y = kh
    const long iters = (a.total + nthreads - 1) / nthreads;
    for (long it = 0; it < iters; ++it) {
        const long gid = it * nthreads + blockIdx.x * (long)blockDim.x + threadIdx.x;
        const bool live = gid < a.total;
        const long g = live ? gid : a.total - 1;
        const int j = (int)(g % cb);
        long pix = g / cb;
        const int w = (int)(pix % a.W); pix /= a.W;
        const int h = (int)(pix % a.H);
        const int b = (int)(pix / a.H);
        const int c0 = j * 8;
        float go[4][8];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float tmp[10];
            unpack_load_feat<T>(a, b, h, w, f, c0, j, cb, live, tmp);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { go[f][i] = tmp[1 + i]; s += tmp[1 + i]; }
            acc[36 + f] += s;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            float pv[10];
            unpack_load_window<T>(a, b, h + kh - 1, w + kw - 1, c0, j, cb, live, pv);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s = fmaf(go[f][i], pv[i + kd], s);
                    acc[(f * 3 + kd) * 3 + kw] += s;
                }
        }
    }
    block_reduce_atomic(acc, kh, a.dw3, sred);
}

template <typename KB, typename KF>
int launch_p3(int dtype, KB kb, KF kf, const P3Args& a, long threads, hipStream_t st, int gy = 1) {
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(kb, dim3(grid, gy), dim3(256), 0, st, a);
    else if (dtype == MTE_DT_F32) hipLaunchKernelGGL(kf, dim3(grid, gy), dim3(256), 0, st, a);
    else return MTE_ERR_UNSUPPORTED;
    return mte_check_launch();
}

bool p3_ok(int C) { const int cb = C >> 3; return C % 8 == 0 && cb >= 1 && cb <= 64 && (cb & (cb - 1)) == 0; }

}  // namespace

// =====================================================================================================
// LDS-tiled PACK stencils (bf16).  The gather versions above re-fetch every input 9x (fwd) / 36x (bwd) through
// L1/L2 and run at ~10 % of the fp32-VALU roof; here a workgroup stages the (TH+2)x(TW+2) halo tile of the
// PACKED volume (all D = 4C depths, depth-contiguous) in LDS once, so each tap is one ds_read_b128 + two 2-byte
// halo reads, and HBM sees every tensor element about once.  Tile area is chosen so that TH*TW*C = 4096.
// =====================================================================================================
namespace {

struct P3Tile { int TH, TW; };
int g_p3_small_tiles = 1;                            // development knob (mte_debug_set(12, v))
inline P3Tile p3_tile(int C) {
    if (g_p3_small_tiles) {
        if (C <= 32) return {4, 16};
        if (C <= 64) return {4, 8};
        if (C <= 128) return {2, 8};
        if (C <= 256) return {2, 4};
        return {2, 2};
    }
    if (C <= 32) return {8, 16};
    if (C <= 64) return {4, 16};
    if (C <= 128) return {4, 8};
    if (C <= 256) return {2, 8};
    return {2, 4};
}
// LDS tiles keep 8 pad elements (16 B) after the D depths of every pixel: with a power-of-two pixel stride the 16-byte window
// reads of a wave (64..1024 B apart) fall on a few banks only -- SQ_LDS_BANK_CONFLICT was 47-84 % of the LDS-active cycles.
#define LDP(D) ((D) + 8)
inline size_t p3_lds_bytes(int C) { P3Tile t = p3_tile(C); return (size_t)(t.TH + 2) * (t.TW + 2) * LDP(4 * C) * 2; }

struct P3LArgs {
    const bf16_t* x; long ldx;         // un-packed side [B,H,W,C]
    const bf16_t* o; long ldo;         // feature side [B,H/2,W/2,16C]
    bf16_t* dst; long lddst;
    const float* w3; const float* b3;
    float* dwb;                        // [112]
    int B, H, W, C, TH, TW;
    int tiles_h, tiles_w, ntiles;
    int dshift, tshift;                // matrix-core weight gradient: log2(depth pairs per pixel) or -1 when not a power of two; log2(TW)
};

// Tile fill, FU chunks per thread in flight (round 6).  The plain `for (idx ...) { v = in-image ? load : 0; store to LDS; }` loops of these kernels compiled to
// load -> s_waitcnt vmcnt(0) -> ds_write per iteration (tools/loopaudit.py): a workgroup walked ~11 DEPENDENT memory round trips per staged plane.  Here a batch
// issues FU unguarded loads from clamped (always valid) addresses, selects zeros for the out-of-image chunks and then stores: one round trip per FU chunks.
//   chunk idx -> (pixel p = idx / cpr of the (rows x PW) tile with origin (h0 - 1, w0 - 1), 16-byte chunk dc = idx % cpr of the pixel's `cpr` chunks)
//   src: element offset of chunk 0 of image pixel (0, 0) of sample b (+ plane offset); pixel stride ld; image Hi x Wi; LDS pixel stride lds_stride elements
template <int FU>
__device__ __forceinline__ void fill_tile_chunks(bf16_t* tile, const bf16_t* src, long ld, int Hi, int Wi, int h0, int w0, int PW, int npix, int cpr, int lds_stride) {
    const int total = npix * cpr;
    for (int idx0 = threadIdx.x; idx0 < total; idx0 += (int)blockDim.x * FU) {
        u32x4_t v[FU];
        int off[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int idx = idx0 + u * (int)blockDim.x;
            const int ii = idx < total ? idx : total - 1;
            const int dc = ii % cpr, p = ii / cpr;
            const int hh = h0 - 1 + p / PW, ww = w0 - 1 + p % PW;
            const bool in = (unsigned)hh < (unsigned)Hi && (unsigned)ww < (unsigned)Wi;
            const int hc = min(max(hh, 0), Hi - 1), wc = min(max(ww, 0), Wi - 1);
            const u32x4_t t = *(const u32x4_t*)(src + ((long)hc * Wi + wc) * ld + dc * 8);
            v[u] = in ? t : u32x4_t{0u, 0u, 0u, 0u};
            off[u] = p * lds_stride + dc * 8;
        }
#pragma unroll
        for (int u = 0; u < FU; ++u)
            if (idx0 + u * (int)blockDim.x < total) *(u32x4_t*)(tile + off[u]) = v[u];
    }
}

// stage packed P tile: tile[(ph)][(pw)][d], d = 4c + s, origin (h0-1, w0-1), zero outside the image.
// One work item = (packed pixel, 8 channels): the four sub-pixel chunks are loaded as 16 bytes each, interleaved in registers
// (depth d = 4c + s: word k of the 32 consecutive depths pairs sub-pixels 2(k&1), 2(k&1)+1 of channel k>>1) and stored as four
// ds_write_b128 -- the element-wise version issued 32 two-byte LDS stores per item.
__device__ __forceinline__ void stage_packed_tile(const P3LArgs& a, bf16_t* tile, int b, int h0, int w0) {
    const int PH = a.TH + 2, PW = a.TW + 2, cpp = a.C >> 3, D = 4 * a.C;
    const int H2 = a.H >> 1, W2 = a.W >> 1;
    const int total = PH * PW * cpp;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int cc = idx % cpp; const int t = idx / cpp;
        const int px = t % PW, py = t / PW;
        const int hh = h0 - 1 + py, ww = w0 - 1 + px;
        u32x4_t v[4];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) v[sp] = u32x4_t{0u, 0u, 0u, 0u};
        if ((unsigned)hh < (unsigned)H2 && (unsigned)ww < (unsigned)W2) {
            const bf16_t* src = a.x + (((long)b * a.H + 2 * hh) * a.W + 2 * ww) * a.ldx + cc * 8;
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) v[sp] = *(const u32x4_t*)(src + ((long)(sp >> 1) * a.W + (sp & 1)) * a.ldx);
        }
        bf16_t* dstp = tile + (py * PW + px) * LDP(D) + 32 * cc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // output chunk q: depths 8q .. 8q+7 = channels 2q, 2q+1
            u32x4_t o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                   // word k of the chunk: channel 2q + (k >> 1), sub-pixels 2(k&1), 2(k&1)+1
                const unsigned lo = v[2 * (k & 1)][q], hi = v[2 * (k & 1) + 1][q];
                o[k] = (k >> 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            }
            *(u32x4_t*)(dstp + 8 * q) = o;
        }
    }
}

// window of NB*8 depths + 1 halo each side at tile pixel `pix`, depth d0 -> pv[0 .. NB*8+1]
template <int NB>
__device__ __forceinline__ void lds_window(const bf16_t* tile, int pix, int D, int d0, float* pv) {
    const bf16_t* p = tile + pix * LDP(D) + d0;
#pragma unroll
    for (int k = 0; k < NB; ++k) unpack16<bf16_t>(*(const u32x4_t*)(p + 8 * k), &pv[1 + 8 * k]);
    pv[0] = d0 > 0 ? bf2f(p[-1]) : 0.f;
    pv[NB * 8 + 1] = d0 + NB * 8 < D ? bf2f(p[NB * 8]) : 0.f;
}

__device__ __forceinline__ void tile_coords(const P3LArgs& a, int tile, int& b, int& h0, int& w0) {
    const int tw = tile % a.tiles_w; int t = tile / a.tiles_w;
    const int th = t % a.tiles_h; b = t / a.tiles_h;
    h0 = th * a.TH; w0 = tw * a.TW;
}

__global__ __launch_bounds__(256, 4) void pack3d_fwd_lds_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ float sw[112];
    if (threadIdx.x < 108) sw[threadIdx.x] = a.w3[threadIdx.x];
    if (threadIdx.x < 4) sw[108 + threadIdx.x] = a.b3[threadIdx.x];
    int b, h0, w0;
    tile_coords(a, blockIdx.x, b, h0, w0);
    stage_packed_tile(a, tile, b, h0, w0);
    __syncthreads();
    const int D = 4 * a.C, dbs = D >> 3, H2 = a.H >> 1, W2 = a.W >> 1;
    const int items = a.TH * a.TW * dbs;
#pragma unroll 1
    for (int it = threadIdx.x; it < items; it += 256) {
        const int db = it % dbs; const int p = it / dbs;
        const int pw = p % a.TW, ph = p / a.TW;
        const int d0 = db * 8;
        float acc[4][8];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[f][i] = sw[108 + f];
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {                      // runtime tap loop: the 12 weights of a tap are LDS broadcasts
            const int kh = t / 3, kw = t - 3 * kh;
            float pv[10];
            lds_window<1>(tile, (ph + kh) * (a.TW + 2) + pw + kw, D, d0, pv);
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = sw[(f * 3 + kd) * 9 + t];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[f][i] = fmaf(wv, pv[i + kd], acc[f][i]);
                }
        }
        const int h = h0 + ph, w = w0 + pw;
        if (h < H2 && w < W2) {
            bf16_t* op = a.dst + (((long)b * H2 + h) * W2 + w) * a.lddst + d0;
#pragma unroll
            for (int f = 0; f < 4; ++f) *(u32x4_t*)(op + f * D) = pack16<bf16_t>(acc[f]);
        }
    }
}

// dx (un-packed) = conv3d^T(dO); one feature plane of dO in LDS at a time, 2 items (pixel, 32 depths) per thread
__global__ __launch_bounds__(256) void pack3d_bwd_data_lds_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ float sw[108];
    if (threadIdx.x < 108) sw[threadIdx.x] = a.w3[threadIdx.x];
    int b, h0, w0;
    tile_coords(a, blockIdx.x, b, h0, w0);
    const int D = 4 * a.C, H2 = a.H >> 1, W2 = a.W >> 1, cbs = a.C >> 3;
    const int PW = a.TW + 2, npix = (a.TH + 2) * PW, cpp = D >> 3;
    float acc[2][32];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[k][i] = 0.f;
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
        __syncthreads();                                   // previous plane fully consumed
        fill_tile_chunks<6>(tile, a.o + (long)b * H2 * W2 * a.ldo + f * D, a.ldo, H2, W2, h0, w0, PW, npix, cpp, LDP(D));
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int it = threadIdx.x + k * 256;           // TH*TW*C/8 == 512 items (fewer for C < 32)
            const int cb = it % cbs; const int p = it / cbs;
            const int pw = p % a.TW, ph = p / a.TW;
            if (ph >= a.TH) continue;
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t - 3 * kh;
                float pv[34];
                lds_window<4>(tile, (ph + 2 - kh) * PW + pw + 2 - kw, D, 32 * cb, pv);
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = sw[((f * 3 + kd) * 3 + kh) * 3 + kw];
#pragma unroll
                    for (int i = 0; i < 32; ++i) acc[k][i] = fmaf(wv, pv[i + 2 - kd], acc[k][i]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int it = threadIdx.x + k * 256;
        const int cb = it % cbs; const int p = it / cbs;
        const int pw = p % a.TW, ph = p / a.TW;
        const int h = h0 + ph, w = w0 + pw;
        if (ph < a.TH && h < H2 && w < W2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[k][4 * i + s];
                *(u32x4_t*)(a.dst + (((long)b * a.H + 2 * h + (s >> 1)) * a.W + 2 * w + (s & 1)) * a.lddst + cb * 8) = pack16<bf16_t>(v);
            }
        }
    }
}

// dw3/db3: persistent over tiles; all 27 taps x 4 features accumulate in registers
__global__ __launch_bounds__(256) void pack3d_bwd_weight_lds_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ float sred[4 * 112];
    const int D = 4 * a.C, dbs = D >> 3, H2 = a.H >> 1, W2 = a.W >> 1;
    const int items = a.TH * a.TW * dbs;
    float acc[112];
#pragma unroll
    for (int i = 0; i < 112; ++i) acc[i] = 0.f;
    for (int tl = blockIdx.x; tl < a.ntiles; tl += gridDim.x) {
        int b, h0, w0;
        tile_coords(a, tl, b, h0, w0);
        __syncthreads();
        stage_packed_tile(a, tile, b, h0, w0);
        __syncthreads();
#pragma unroll 1
        for (int it = threadIdx.x; it < items; it += 256) {
            const int db = it % dbs; const int p = it / dbs;
            const int pw = p % a.TW, ph = p / a.TW;
            const int h = h0 + ph, w = w0 + pw;
            if (h >= H2 || w >= W2) continue;
            const int d0 = db * 8;
            const bf16_t* src = a.o + (((long)b * H2 + h) * W2 + w) * a.ldo + d0;
#pragma unroll
            for (int f = 0; f < 4; ++f) {                  // one feature's 8 gradient values live at a time
                float go[8];
                unpack16<bf16_t>(*(const u32x4_t*)(src + f * D), go);
                float sb = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sb += go[i];
                acc[108 + f] += sb;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        float pv[10];
                        lds_window<1>(tile, (ph + kh) * (a.TW + 2) + pw + kw, D, d0, pv);
#pragma unroll
                        for (int kd = 0; kd < 3; ++kd) {
                            float sacc = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) sacc = fmaf(go[i], pv[i + kd], sacc);
                            acc[((f * 3 + kd) * 3 + kh) * 3 + kw] += sacc;
                        }
                        asm volatile("" ::: "memory");   // keep window reads from being hoisted together (VGPR budget)
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
    if (threadIdx.x < 112)
        atomicAdd(a.dwb + threadIdx.x, sred[threadIdx.x] + sred[112 + threadIdx.x] + sred[224 + threadIdx.x] + sred[336 + threadIdx.x]);
}

// ---- LDS-tiled UNPACK backward.  Volume = x [B,H,W,C] (depth = channel).  The feature side is the pixel-shuffled
// gradient dout [B,2H,2W,C]: feature plane f of the volume is the space-to-depth of dout's channels [f*C/4, (f+1)*C/4),
// so staging it is stage_packed_tile with (x := dout + f*C/4, C := C/4, H := 2H, W := 2W).
inline P3Tile up_tile(int C) {                       // TH*TW*C = 16384 (2 items of 32 depths per thread)
    if (C <= 32) return {16, 32};
    if (C <= 64) return {8, 32};
    if (C <= 128) return {8, 16};
    if (C <= 256) return {4, 16};
    return {4, 8};
}
inline size_t up_lds_bytes(int C) { P3Tile t = up_tile(C); return (size_t)(t.TH + 2) * (t.TW + 2) * LDP(C) * 2; }

__device__ __forceinline__ void up_tile_coords(const P3LArgs& a, int tile, int& b, int& h0, int& w0) {
    const int tw = tile % a.tiles_w; int t = tile / a.tiles_w;
    const int th = t % a.tiles_h; b = t / a.tiles_h;
    h0 = th * a.TH; w0 = tw * a.TW;
}

__global__ __launch_bounds__(256) void unpack3d_bwd_data_lds_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ float sw[108];
    if (threadIdx.x < 108) sw[threadIdx.x] = a.w3[threadIdx.x];
    int b, h0, w0;
    up_tile_coords(a, blockIdx.x, b, h0, w0);
    const int D = a.C, cbs = a.C >> 5, PW = a.TW + 2;
    P3LArgs t = a;                                      // plane staging view of dout
    t.ldx = a.ldo; t.C = a.C >> 2; t.H = 2 * a.H; t.W = 2 * a.W;
    float acc[2][32];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[k][i] = 0.f;
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
        __syncthreads();
        t.x = a.o + f * (a.C >> 2);
        stage_packed_tile(t, tile, b, h0, w0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int it = threadIdx.x + k * 256;
            const int cb = it % cbs; const int p = it / cbs;
            const int pw = p % a.TW, ph = p / a.TW;
            if (ph >= a.TH) continue;
#pragma unroll 1
            for (int tp = 0; tp < 9; ++tp) {
                const int kh = tp / 3, kw = tp - 3 * kh;
                float pv[34];
                lds_window<4>(tile, (ph + 2 - kh) * PW + pw + 2 - kw, D, 32 * cb, pv);
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = sw[((f * 3 + kd) * 3 + kh) * 3 + kw];
#pragma unroll
                    for (int i = 0; i < 32; ++i) acc[k][i] = fmaf(wv, pv[i + 2 - kd], acc[k][i]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int it = threadIdx.x + k * 256;
        const int cb = it % cbs; const int p = it / cbs;
        const int pw = p % a.TW, ph = p / a.TW;
        const int h = h0 + ph, w = w0 + pw;
        if (ph < a.TH && h < a.H && w < a.W) {
            bf16_t* dp = a.dst + (((long)b * a.H + h) * a.W + w) * a.lddst + 32 * cb;
#pragma unroll
            for (int j = 0; j < 4; ++j) *(u32x4_t*)(dp + 8 * j) = pack16<bf16_t>(&acc[k][8 * j]);
        }
    }
}

// Same operator, all four feature planes of the shuffled gradient staged at once (C <= 128, where the tensors are large):
// the plane-by-plane kernel above touches every 64..256-byte pixel record of dout in four separate sweeps, and by the time a
// block comes back for the next plane the record has left L2 -- measured 3.4 GB fetched per step for 0.49 GB of dout.
// The x-tile is four times smaller instead (TH*TW*C = 4096, one 16-depth item per thread) so the four planes fit the
// same ~46-61 KB of LDS.
inline P3Tile up4_tile(int C) {
    if (C <= 32) return {8, 16};
    if (C <= 64) return {4, 16};
    return {4, 8};
}
inline size_t up4_lds_bytes(int C) { P3Tile t = up4_tile(C); return (size_t)4 * (t.TH + 2) * (t.TW + 2) * LDP(C) * 2; }

__global__ __launch_bounds__(256) void unpack3d_bwd_data_lds4_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ float sw[108];
    if (threadIdx.x < 108) sw[threadIdx.x] = a.w3[threadIdx.x];
    int b, h0, w0;
    up_tile_coords(a, blockIdx.x, b, h0, w0);
    const int D = a.C, cbs = a.C >> 4, PW = a.TW + 2;
    const int plane = (a.TH + 2) * PW * LDP(D);         // elements of one staged feature plane
    P3LArgs t = a;                                      // plane staging view of dout
    t.ldx = a.ldo; t.C = a.C >> 2; t.H = 2 * a.H; t.W = 2 * a.W;
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {                       // four sweeps over the SAME records back to back: the re-reads hit L2
        t.x = a.o + f * (a.C >> 2);
        stage_packed_tile(t, tile + f * plane, b, h0, w0);
    }
    __syncthreads();
    const int it = threadIdx.x;
    const int cb = it % cbs; const int p = it / cbs;
    const int pw = p % a.TW, ph = p / a.TW;
    if (ph >= a.TH) return;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
#pragma unroll 1
        for (int tp = 0; tp < 9; ++tp) {
            const int kh = tp / 3, kw = tp - 3 * kh;
            float pv[18];
            lds_window<2>(tile + f * plane, (ph + 2 - kh) * PW + pw + 2 - kw, D, 16 * cb, pv);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float wv = sw[((f * 3 + kd) * 3 + kh) * 3 + kw];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = fmaf(wv, pv[i + 2 - kd], acc[i]);
            }
        }
    }
    const int h = h0 + ph, w = w0 + pw;
    if (h < a.H && w < a.W) {
        bf16_t* dp = a.dst + (((long)b * a.H + h) * a.W + w) * a.lddst + 16 * cb;
#pragma unroll
        for (int j = 0; j < 2; ++j) *(u32x4_t*)(dp + 8 * j) = pack16<bf16_t>(&acc[8 * j]);
    }
}

// ---- conv3d data paths on the matrix cores (round 4).  The fp32-VALU stencils above run at ~23 cycles per output and SIMD on the large
// layers (108 FMAs + the bf16 unpacking of every window, two workgroups per CU, staging and arithmetic one after the other) -- a seventh of
// the vector peak and a fifth of the HBM rate.  The depth taps make the stencil a GEMM against a BANDED matrix: for one (feature, kh, kw)
//     out[pixel][d0 + n] += sum_kk  X[pixel'][d0 - 8 + kk] * T[kk][n],     T[kk][n] = w[kd = n + 9 - kk] for 0 <= kd <= 2, else 0
// i.e. one v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 depths with K = a 32-deep window of the LDS tile that starts 8 depths (one
// 16-byte chunk: aligned ds_read_b128) below the block.  Only 3 of 32 K entries per column carry weight, and that is still 5x faster: 36
// (feature, kh, kw) x 2 MFMAs (weights split into bf16 hi + lo parts, so the fp32 conv3d weights keep ~16 bits) = 72 MFMAs x 16 cycles per
// 256 outputs = 4.5 cycles per output, with one LDS read per 2 MFMAs.  The banded operand is never stored: a lane's 8 entries are byte
// permutes (v_perm_b32, selectors fixed per lane) of the two dwords {w0 w1} {w2 0} read from a 36-entry LDS table.
// Operands are SWAPPED (weights first): an accumulator register then holds 4 consecutive DEPTHS of one pixel -> 8-byte stores.
//
// unpack backward data, C = 32 / 64: tile = TH x 16 pixels, the four feature planes of the pixel-shuffled gradient staged together
// ([plane][pixel][8 zero | C depths | 8 zero] bf16, 69 KB: two workgroups per CU); a wave owns 4 (tile row, depth block) units.
int g_p3_tr_passes = 4;                              // development knob (mte_debug_set(1, 3000 + v), v = 1 / 2 / 4): output passes of unpack3d_fwd_tr_kernel
int g_p3_persist_wgs = 1024;                         // development knob (mte_debug_set(1, 2000 + v)): workgroups of the persistent matrix-core conv3d kernels
int g_p3_mfma_data = 239;                             // development knob (mte_debug_set(1, 300 + v)): bit 0 = unpack backward data on the matrix cores, bit 1 = its LDS-DMA form for C = 32, bit 2 = 4 waves per workgroup there (0: 2 waves x 4 rows, measured slower), bit 3 = unpack forward on the matrix cores, bit 4 = conv3d weights as ONE bf16 value (no lo part: half the MFMAs), bit 5 = unpack forward with the spatial taps in K (third form), bit 6 = pack forward in that form, bit 7 = pack backward data on the matrix cores (round 6)

__device__ __forceinline__ bf16x8_t banded_fragment(unsigned t0, unsigned t1, const unsigned sel[4]) {
    u32x4_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_perm(t1, t0, sel[j]);
    return __builtin_bit_cast(bf16x8_t, r);
}

template <int C, bool HILO>
__global__ __launch_bounds__(256, 2) void unpack3d_bwd_data_mfma_kernel(P3LArgs a) {
    constexpr int TH = C == 32 ? 8 : 4, TW = 16, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, LDR = C + 16, DB = C / 16, PLANE = NPIX * LDR;
    constexpr int CPP = C / 32;                           // 16-byte chunks (8 channels) of one feature plane in a sub-pixel's record
    constexpr int ITEMS = NPIX * (C / 8), NIT = (ITEMS + 255) / 256;
    static_assert(TH * DB == 16, "four units per wave");
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ __attribute__((aligned(16))) unsigned wtab[36 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b, h0, w0;
    up_tile_coords(a, xcd_remap(blockIdx.x, gridDim.x), b, h0, w0);
    // ---- the gradient records of the tile + halo: item = (tile pixel, chunk q of the C-channel record), q fastest: the C / 8 lanes of a
    //      pixel read one contiguous record per sub-pixel.  Addresses clamped into the image, values selected to zero.
    u32x4_t v[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        const int q = idx % (C / 8), t = idx / (C / 8);
        const int hh = h0 - 1 + t / PW, ww = w0 - 1 + t % PW;
        const bool ok = idx < ITEMS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        const int ch = min(max(hh, 0), a.H - 1), cw = min(max(ww, 0), a.W - 1);
        const bf16_t* src = a.o + (((long)b * 2 * a.H + 2 * ch) * (2 * a.W) + 2 * cw) * a.ldo + q * 8;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const u32x4_t r = *(const u32x4_t*)(src + ((long)(sp >> 1) * (2 * a.W) + (sp & 1)) * a.ldo);
            v[it][sp] = u32x4_t{ok ? r[0] : 0u, ok ? r[1] : 0u, ok ? r[2] : 0u, ok ? r[3] : 0u};
#if defined(MTE_P3_ABLATE) && (MTE_P3_ABLATE & 1)
            v[it][sp] = u32x4_t{(unsigned)idx, 0u, 0u, 0u};          // diagnostic: no global loads
#endif
        }
    }
    // ---- weights: {w0 w1} {w2 0} as bf16 hi parts and lo parts (w - hi) per (feature, kh, kw); zero pads of every pixel row
    if (tid < 36) {
        const int f = tid / 9, k9 = tid - 9 * f;
        unsigned hi[3], lo[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const float w = a.w3[(f * 3 + kd) * 9 + k9];
            hi[kd] = f2bf(w); lo[kd] = f2bf(w - bf2f((bf16_t)hi[kd]));
        }
        *(u32x4_t*)(wtab + 4 * tid) = u32x4_t{hi[0] | (hi[1] << 16), hi[2], lo[0] | (lo[1] << 16), lo[2]};
    }
    for (int i = tid; i < 4 * NPIX * 2; i += 256) *(u32x4_t*)(tile + (i >> 1) * LDR + ((i & 1) ? C + 8 : 0)) = u32x4_t{0u, 0u, 0u, 0u};
    // ---- records -> planes: depth d = 4 c + s (c: channel of the plane, s: sub-pixel); see stage_packed_tile for the interleave
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        if (idx < ITEMS) {
            const int q = idx % (C / 8), t = idx / (C / 8);
            const int f = q / CPP, cc = q % CPP;
            bf16_t* dstp = tile + f * PLANE + t * LDR + 8 + 32 * cc;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                u32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned lo = v[it][2 * (k & 1)][qq], hi = v[it][2 * (k & 1) + 1][qq];
                    o[k] = (k >> 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
                }
                *(u32x4_t*)(dstp + 8 * qq) = o;
            }
        }
    }
    // ---- this lane's part of the banded operand: row n = lane % 16 (output depth), K entries 8 g .. 8 g + 7; entry kk carries w[kd = n + 9 - kk]
    const int n = lane & 15, g = lane >> 4;
    unsigned sel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned sv = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int kd = n + 9 - (8 * g + 2 * j + e);
            // bytes of the pool {t1 (4..7), t0 (0..3)}: w0 = 0,1  w1 = 2,3  w2 = 4,5; 0x0c = constant zero
            const unsigned two = kd == 0 ? 0x0100u : (kd == 1 ? 0x0302u : (kd == 2 ? 0x0504u : 0x0c0cu));
            sv |= two << (16 * e);
        }
        sel[j] = sv;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // unit u of this wave: tile row and depth block
    auto unit_row = [&](int u) { return C == 32 ? 2 * wave + (u >> 1) : wave; };
    auto unit_db = [&](int u) { return C == 32 ? (u & 1) : u; };
#pragma unroll 1
#if defined(MTE_P3_ABLATE) && (MTE_P3_ABLATE & 2)
    for (int f = 0; f < 1; ++f) {                                  // diagnostic: a quarter of the MFMA loop
#else
    for (int f = 0; f < 4; ++f) {
#endif
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) {
            const int kh = k9 / 3, kw = k9 - 3 * kh;
            const u32x4_t wq = *(const u32x4_t*)(wtab + 4 * (f * 9 + k9));
            const bf16x8_t bhi = banded_fragment(wq[0], wq[1], sel), blo = banded_fragment(wq[2], wq[3], sel);
            bf16x8_t x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pix = (unit_row(u) + 2 - kh) * PW + n + 2 - kw;        // source pixel p + 1 - k of output pixel (row, n), tile origin (-1, -1)
                x[u] = *(const bf16x8_t*)(tile + f * PLANE + pix * LDR + 16 * unit_db(u) + 8 * g);
            }
            // the four units between the two MFMAs of one accumulator (a dependent MFMA waits out the first one's passes)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi, x[u], acc[u], 0, 0, 0);
            if constexpr (HILO) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo, x[u], acc[u], 0, 0, 0);
            }
        }
    }
    // ---- accumulator register j of unit u: depth 16 db + 4 g + j of pixel (row, n)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int h = h0 + unit_row(u), w = w0 + n;
        if (h < a.H && w < a.W) {
            unsigned* dp = (unsigned*)(a.dst + (((long)b * a.H + h) * a.W + w) * a.lddst + 16 * unit_db(u) + 4 * g);
            dp[0] = pack2bf(acc[u][0], acc[u][1]);
            dp[1] = pack2bf(acc[u][2], acc[u][3]);
        }
    }
}

// unpack backward data, C = 32, second form: the gradient records go into LDS AS THEY ARE ([tile pixel][16 chunks = 4 sub-pixels x 4 feature
// planes] x 16 bytes, 46 KB: three workgroups per CU) by buffer_load ... lds -- no registers, no interleave, no LDS stores.  The order of the
// K entries of an MFMA is free as long as both operands agree, so the data operand of lane (pixel, g) is simply sub-pixel g's 8 channels of
// plane f (depths 4 t + g, t = 0..7) and the banded operand's selectors are built for that order; likewise the ROWS of the banded operand
// are assigned to output depths so that a lane ends up with 8 consecutive depths (row 4 g' + j of block blk = depth 8 g' + 4 blk + j): one
// 16-byte store.  With C = 32 one 32-deep window holds the whole depth range, so ONE LDS read feeds the four MFMAs of a (feature, kh, kw)
// (two depth blocks x hi / lo) and depth padding needs no storage.  The 16 chunks of a pixel are XOR-swizzled with the pixel's column so that
// the 16 lanes of a read phase (16 consecutive pixels, same chunk) hit 16 different bank groups.
template <int WAVES, bool HILO>                       // waves per workgroup; a wave owns RW = 8 / WAVES tile rows: the 16 v_perm of a (feature, kh, kw) are shared by 4 RW MFMAs
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 3) void unpack3d_bwd_data_dma32_kernel(P3LArgs a) {
    constexpr int TH = 8, PW = 18, NPIX = 180, RW = TH / WAVES;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    __shared__ __attribute__((aligned(16))) unsigned wtab[36 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, h0, w0;
    up_tile_coords(a, xcd_remap(blockIdx.x, gridDim.x), b, h0, w0);
    const long total = ((long)a.B * 4 * a.H * a.W - 1) * a.ldo + 32;                          // elements of the gradient tensor (launcher: < 2^30)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.o, 0, (int)(total * 2), 0x00020000);
#pragma unroll 1
    for (int k = wave; k < NPIX / 4; k += WAVES) {                 // one instruction = 4 tile pixels x 16 chunks = 1 KB of LDS
        const int t = 4 * k + (lane >> 4);
        const int py = t / PW, px = t - PW * py;
        const int chunk = (lane & 15) ^ (px & 15), sp = chunk >> 2, f = chunk & 3;
        const int hh = h0 - 1 + py, ww = w0 - 1 + px;
        const bool ok = (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        const long el = (((long)b * 2 * a.H + 2 * hh + (sp >> 1)) * (2 * a.W) + 2 * ww + (sp & 1)) * a.ldo + 8 * f;
        const unsigned off = ok ? (unsigned)(el * 2) : 0x7ffffff0u;                            // out of range: the load returns zeros
#if defined(MTE_P3_ABLATE) && (MTE_P3_ABLATE & 1)
        if (off == 0x12345u)                                       // diagnostic: no staging
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem_ + k * 1024), 16, off, 0, 0, 0);
    }
    if (tid < 36) {
        const int f = tid / 9, k9 = tid - 9 * f;
        unsigned hi[3], lo[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const float w = a.w3[(f * 3 + kd) * 9 + k9];
            hi[kd] = f2bf(w); lo[kd] = f2bf(w - bf2f((bf16_t)hi[kd]));
        }
        *(u32x4_t*)(wtab + 4 * tid) = u32x4_t{hi[0] | (hi[1] << 16), hi[2], lo[0] | (lo[1] << 16), lo[2]};
    }
    // banded operand of this lane: row m = lane % 16 <-> output depth 8 (m / 4) + 4 blk + m % 4; K entry t of group g <-> input depth 4 t + g
    const int n = lane & 15, g = lane >> 4;
    unsigned sel[2][4];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned sv = 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kd = (8 * (n >> 2) + 4 * blk + (n & 3)) + 1 - (4 * (2 * j + e) + g);
                const unsigned two = kd == 0 ? 0x0100u : (kd == 1 ? 0x0302u : (kd == 2 ? 0x0504u : 0x0c0cu));
                sv |= two << (16 * e);
            }
            sel[blk][j] = sv;
        }
    f32x4_t acc[RW][2];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) acc[r][blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
#if defined(MTE_P3_ABLATE) && (MTE_P3_ABLATE & 2)
    for (int f = 0; f < 1; ++f) {                                  // diagnostic: a quarter of the MFMA loop
#else
    for (int f = 0; f < 4; ++f) {
#endif
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) {
            const int kh = k9 / 3, kw = k9 - 3 * kh;
            const u32x4_t wq = *(const u32x4_t*)(wtab + 4 * (f * 9 + k9));
            bf16x8_t x[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) {                         // rows RW wave + r of the tile; source pixel p + 1 - k, tile origin (-1, -1)
                const int px = n + 2 - kw, t = (RW * wave + r + 2 - kh) * PW + px;
                x[r] = *(const bf16x8_t*)(smem_ + t * 256 + (((4 * g + f) ^ (px & 15)) << 4));
            }
            const bf16x8_t h0f = banded_fragment(wq[0], wq[1], sel[0]), h1f = banded_fragment(wq[0], wq[1], sel[1]);
            const bf16x8_t l0f = banded_fragment(wq[2], wq[3], sel[0]), l1f = banded_fragment(wq[2], wq[3], sel[1]);
            // 2 RW independent accumulators between the two MFMAs of one accumulator (a dependent MFMA waits out the first one's passes)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0f, x[r], acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1f, x[r], acc[r][1], 0, 0, 0);
            }
            if constexpr (HILO) {
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0f, x[r], acc[r][0], 0, 0, 0);
                    acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1f, x[r], acc[r][1], 0, 0, 0);
                }
            }
        }
    }
    // lane (pixel n, g): register j of block blk = depth 8 g + 4 blk + j
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int h = h0 + RW * wave + r, w = w0 + n;
#if defined(MTE_P3_ABLATE) && (MTE_P3_ABLATE & 4)
        if (h < a.H && w < a.W && acc[r][0][0] == 123.f) {         // diagnostic: no stores
#else
        if (h < a.H && w < a.W) {
#endif
            u32x4_t o = {pack2bf(acc[r][0][0], acc[r][0][1]), pack2bf(acc[r][0][2], acc[r][0][3]), pack2bf(acc[r][1][0], acc[r][1][1]), pack2bf(acc[r][1][2], acc[r][1][3])};
            *(u32x4_t*)(a.dst + (((long)b * a.H + h) * a.W + w) * a.lddst + 8 * g) = o;
        }
    }
}

// pack backward data (round 6): dx = un-shuffle(conv3d^T(dO)), C % 32 == 0, the same banded-operand GEMM.  The fp32-VALU stencil above (pack3d_bwd_data_lds_kernel)
// was the largest non-convolution, non-GroupNorm kernel of the main queue: 0.48 ms per step alone, 0.70 ms beside the weight-gradient queue, at 1.8 TB/s
// (profiles/r06_v6_pmc_traffic_T8.txt: 800 MB read, 85 MB written) -- arithmetic-bound at 48 v_pk_fma_f32 + ~50 unpacking instructions per 32 outputs, tap and plane.
//   * work item = (tile of 4 x 16 packed pixels, segment of 128 depths of the D = 4C); a wave owns ONE unit of 32 depths and all four tile rows: the six halo rows
//     it reads per (plane, kw) feed 4 rows x 3 kh x 2 blocks x (hi, lo) = 48 MFMAs -- one ds_read_b128 per 4 MFMAs.  (First version: a wave = one row, four
//     units -- one read per 2 MFMAs; the same time to the microsecond: 9.6 cycles per output and SIMD against the 4.5 of the MFMAs, the rest is load latency + the
//     four meetings per workgroup.)
//   * the feature side dO[b][h][w][f D + d] is plain [pixel][depth] per plane f: the 16-byte chunks -1 .. 16 of the segment (the two outer ones are the depth
//     halo: the neighbouring segment's, or zeros at depths -1 / D) of the 6 x 18 halo pixels go to LDS by buffer_load ... lds, chunk-major inside groups of 16
//     pixels ([group][slot][pixel] x 16 B: the 16 lanes of a read phase -- 16 consecutive pixels, one slot -- cover 256 contiguous bytes); out-of-image pixels and
//     out-of-range depths load zeros through the buffer bounds check.
//     One plane at a time, 36 KB: four workgroups per CU.  (Built and measured slower: persistent workgroups, two per CU, with the next plane's records in flight
//     into a second buffer -- 160 vs 123 us at C = 32 @192x640, 83 vs 61 at C = 64: two waves per SIMD do not cover the LDS latency between the taps.)
//   * 16-depth block at d0: K entry t of lane group g <-> input depth d0 - 8 + 8 g + t (slot 4 u + 2 blk + g: one aligned ds_read_b128), row m = 4 g' + j of the
//     banded operand <-> OUTPUT depth d0 + 4 j + g' = channel d0 / 4 + j, sub-pixel g'.  The accumulator registers of lane (pixel n, g) are then 4 consecutive
//     channels of sub-pixel g, and the two blocks of the unit 8 channels: ONE 16-byte store into dx[b][2 h + g / 2][2 w + g % 2][c0 .. c0 + 7].
//   * the banded fragments (8 v_perm_b32 for hi + lo) are the same for every block and row: built once per (plane, kh, kw), used by 16 MFMAs.
// Segments of 256 depths (a wave = two units) were built too and lost on every layer with D >= 256 (74 vs 61 us at C = 64, 56 vs 40 at C = 512: profiles/r06_pack_bwd_mfma.txt).
template <bool HILO>
__global__ __launch_bounds__(256, 4) void pack3d_bwd_data_mfma_kernel(P3LArgs a) {
    constexpr int DS = 128, TH = 4, PW = 18, NPIX = (TH + 2) * PW, NG = (NPIX + 15) / 16, NCH = DS / 8, NSL = (NCH + 2 + 3) & ~3, GS = NSL * 256;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    __shared__ __attribute__((aligned(16))) unsigned wtab[36 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = 4 * a.C, H2 = a.H >> 1, W2 = a.W >> 1, nseg = D / DS;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int seg0 = (id % nseg) * DS;
    int b, h0, w0;
    tile_coords(a, id / nseg, b, h0, w0);
    const long total = ((long)a.B * H2 * W2 - 1) * a.ldo + 4 * D;                              // elements of the gradient tensor (launcher: < 2^30)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.o, 0, (int)(total * 2), 0x00020000);
    auto stage = [&](int f) __attribute__((always_inline)) {
#pragma unroll 1
        for (int k = wave; k < NG * (NSL / 4); k += 4) {               // one instruction = 16 tile pixels x 4 slots = 1 KB of LDS
            const int grp = k / (NSL / 4), q = k - grp * (NSL / 4);
            const int t = 16 * grp + (lane & 15), slot = 4 * q + (lane >> 4);
            const int py = t / PW, px = t - PW * py;
            const int hh = h0 - 1 + py, ww = w0 - 1 + px, dd = seg0 + 8 * (slot - 1);
            const bool ok = t < NPIX && slot < NCH + 2 && (unsigned)hh < (unsigned)H2 && (unsigned)ww < (unsigned)W2 && (unsigned)dd < (unsigned)D;
            const long el = (((long)b * H2 + hh) * W2 + ww) * a.ldo + f * D + dd;
            const unsigned off = ok ? (unsigned)(el * 2) : 0x7ffffff0u;                        // out of range: the load returns zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem_ + grp * GS + q * 1024), 16, off, 0, 0, 0);
        }
    };
    stage(0);
    if (tid < 36) {
        const int f = tid / 9, k9 = tid - 9 * f;
        unsigned hi[3], lo[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const float w = a.w3[(f * 3 + kd) * 9 + k9];
            hi[kd] = f2bf(w); lo[kd] = f2bf(w - bf2f((bf16_t)hi[kd]));
        }
        *(u32x4_t*)(wtab + 4 * tid) = u32x4_t{hi[0] | (hi[1] << 16), hi[2], lo[0] | (lo[1] << 16), lo[2]};
    }
    // banded operand of this lane: row m = lane % 16 <-> output depth d0 + 4 (m % 4) + m / 4; K entry t of group g <-> input depth d0 - 8 + 8 g + t;
    // dP[d_out] += w[kd] dO[d_out - kd + 1]: the entry carries w[kd = d_out - d_in + 1]
    const int n = lane & 15, g = lane >> 4;
    unsigned sel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned sv = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int kd = (4 * (n & 3) + (n >> 2)) - (8 * g - 8 + 2 * j + e) + 1;
            const unsigned two = kd == 0 ? 0x0100u : (kd == 1 ? 0x0302u : (kd == 2 ? 0x0504u : 0x0c0cu));
            sv |= two << (16 * e);
        }
        sel[j] = sv;
    }
    f32x4_t acc[TH][2];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) acc[r][blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
        if (f) {
            __syncthreads();                                           // every wave is past its reads of the previous plane
            stage(f);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            bf16x8_t x[TH + 2][2];                                     // halo rows 0 .. 5, column n + 2 - kw (source pixel p + 1 - k, tile origin (-1, -1)), this wave's unit
#pragma unroll
            for (int py = 0; py < TH + 2; ++py) {
                const int t = py * PW + n + 2 - kw;
                const char* base = smem_ + (t >> 4) * GS + (t & 15) * 16 + (4 * wave + g) * 256;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) x[py][blk] = *(const bf16x8_t*)(base + 2 * blk * 256);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const u32x4_t wq = *(const u32x4_t*)(wtab + 4 * (f * 9 + 3 * kh + kw));
                const bf16x8_t bhi = banded_fragment(wq[0], wq[1], sel), blo = banded_fragment(wq[2], wq[3], sel);
                // 8 independent accumulators between the two MFMAs of one accumulator (a dependent MFMA waits out the first one's passes)
#pragma unroll
                for (int r = 0; r < TH; ++r)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) acc[r][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhi, x[r + 2 - kh][blk], acc[r][blk], 0, 0, 0);
                if constexpr (HILO) {
#pragma unroll
                    for (int r = 0; r < TH; ++r)
#pragma unroll
                        for (int blk = 0; blk < 2; ++blk) acc[r][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blo, x[r + 2 - kh][blk], acc[r][blk], 0, 0, 0);
                }
            }
        }
    }
    // lane (pixel n, sub-pixel g): register j of block blk = channel seg0 / 4 + 8 wave + 4 blk + j of dx pixel (2 h + g / 2, 2 w + g % 2)
    const int w = w0 + n;
#pragma unroll
    for (int r = 0; r < TH; ++r) {
        const int h = h0 + r;
        if (h < H2 && w < W2)
            *(u32x4_t*)(a.dst + (((long)b * a.H + 2 * h + (g >> 1)) * a.W + 2 * w + (g & 1)) * a.lddst + (seg0 >> 2) + 8 * wave) =
                u32x4_t{pack2bf(acc[r][0][0], acc[r][0][1]), pack2bf(acc[r][0][2], acc[r][0][3]), pack2bf(acc[r][1][0], acc[r][1][1]), pack2bf(acc[r][1][2], acc[r][1][3])};
    }
}

// unpack forward (conv3d 1 -> 4 + pixel shuffle), C = 32 / 64, on the matrix cores: the same banded-operand GEMM with the four feature planes
// on the OUTPUT side -- one LDS read of a 32-deep window feeds 8 MFMAs (4 features x hi / lo).  The input tile is plain [pixel][depth]; its
// 16-byte chunks go to LDS by buffer_load ... lds in chunk-major order inside groups of 16 pixels ([group][chunk -1 .. C/8][pixel]: the 16
// lanes of a read phase -- 16 consecutive pixels, one chunk -- fall on 16 different bank groups; the two extra chunk slots stay zero and stand
// for depths < 0 and >= C).  Depth block blk reads chunks 2 blk - 1 .. 2 blk + 2.  Rows of the banded operand are assigned to output depths
// so that lane (pixel, s) ends up with sub-pixel s's channels: row 4 s + j of block blk = depth 4 (4 blk + j) + s = channel 4 blk + j of the
// plane -> 16-byte stores that together fill the sub-pixel's whole record.
template <int C, bool HILO>
__global__ __launch_bounds__(256, 4) void unpack3d_fwd_mfma_kernel(P3LArgs a) {
    constexpr int TH = C == 32 ? 8 : 4, RW = TH / 4, PW = 18, NPIX = (TH + 2) * PW, NG = (NPIX + 15) / 16, NCH = C / 8, NB = C / 16;
    constexpr int GS = (NCH + 2) * 256, BUF = NG * GS;             // bytes of a 16-pixel group / of one tile buffer
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) char smem_[];   // two tile buffers: the next tile's records land while this one is multiplied
    __shared__ __attribute__((aligned(16))) unsigned wtab[36 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long total = ((long)a.B * a.H * a.W - 1) * a.ldx + C;                               // elements of x (launcher: < 2^30)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(total * 2), 0x00020000);
    auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
        int b, h0, w0;
        up_tile_coords(a, xcd_remap(tile, a.ntiles), b, h0, w0);
#pragma unroll 1
        for (int k = wave; k < NG * (NCH / 4); k += 4) {           // one instruction = 16 tile pixels x 4 chunks = 1 KB of LDS
            const int grp = k / (NCH / 4), half = k - grp * (NCH / 4);
            const int t = 16 * grp + (lane & 15), ch = 4 * half + (lane >> 4);
            const int py = t / PW, px = t - PW * py;
            const int hh = h0 - 1 + py, ww = w0 - 1 + px;
            const bool ok = t < NPIX && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
            const long el = (((long)b * a.H + hh) * a.W + ww) * a.ldx + 8 * ch;
            const unsigned off = ok ? (unsigned)(el * 2) : 0x7ffffff0u;                        // out of range: the load returns zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem_ + buf * BUF + grp * GS + (1 + 4 * half) * 256), 16, off, 0, 0, 0);
        }
    };
    int tile = blockIdx.x;
    stage(tile, 0);
    for (int i = tid; i < 2 * NG * 2 * 16; i += 256) {             // chunk slots -1 and C / 8 of every group of both buffers
        const int buf = i / (NG * 32), r = i - buf * (NG * 32);
        *(u32x4_t*)(smem_ + buf * BUF + (r >> 5) * GS + ((r >> 4) & 1) * (NCH + 1) * 256 + (r & 15) * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    if (tid < 36) {
        const int f = tid / 9, k9 = tid - 9 * f;
        unsigned hi[3], lo[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const float w = a.w3[(f * 3 + kd) * 9 + k9];
            hi[kd] = f2bf(w); lo[kd] = f2bf(w - bf2f((bf16_t)hi[kd]));
        }
        *(u32x4_t*)(wtab + 4 * tid) = u32x4_t{hi[0] | (hi[1] << 16), hi[2], lo[0] | (lo[1] << 16), lo[2]};
    }
    // banded operand of this lane: row m = lane % 16 <-> output depth 16 blk + 4 (m % 4) + m / 4; K entry t of group g <-> input depth
    // 16 blk - 8 + 8 g + t; entry carries w[kd = d_in - d_out + 1]
    const int n = lane & 15, g = lane >> 4;
    unsigned sel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned sv = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int kd = (8 * g - 8 + 2 * j + e) - (4 * (n & 3) + (n >> 2)) + 1;
            const unsigned two = kd == 0 ? 0x0100u : (kd == 1 ? 0x0302u : (kd == 2 ? 0x0504u : 0x0c0cu));
            sv |= two << (16 * e);
        }
        sel[j] = sv;
    }
    float bias[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) bias[f] = a.b3[f];
    // ---- tiles blockIdx.x, + gridDim.x, ...: wait for this tile's records (and the previous tile's stores), meet, start the next tile's
    //      records into the other buffer (every wave is past its reads of it), multiply, store
#pragma unroll 1
    for (int it = 0; tile < a.ntiles; ++it, tile += gridDim.x) {
        const int buf = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) stage(tile + gridDim.x, buf ^ 1);
        int b, h0, w0;
        up_tile_coords(a, xcd_remap(tile, a.ntiles), b, h0, w0);
        f32x4_t acc[RW][NB][4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) acc[r][blk][f] = f32x4_t{bias[f], bias[f], bias[f], bias[f]};
#pragma unroll 1
        for (int k9 = 0; k9 < 9; ++k9) {
            const int kh = k9 / 3, kw = k9 - 3 * kh;
            bf16x8_t fh[4], fl[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const u32x4_t wq = *(const u32x4_t*)(wtab + 4 * (f * 9 + k9));
                fh[f] = banded_fragment(wq[0], wq[1], sel); fl[f] = banded_fragment(wq[2], wq[3], sel);
            }
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int t = (RW * wave + r + kh) * PW + n + kw;   // source pixel p + k - 1, tile origin (-1, -1)
                const char* base = smem_ + buf * BUF + (t >> 4) * GS + (t & 15) * 16 + g * 256;
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    const bf16x8_t x = *(const bf16x8_t*)(base + 2 * blk * 256);               // chunk slot (2 blk - 1 + g) + 1
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[r][blk][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[f], x, acc[r][blk][f], 0, 0, 0);
                    if constexpr (HILO) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) acc[r][blk][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl[f], x, acc[r][blk][f], 0, 0, 0);
                    }
                }
            }
        }
        // lane (pixel n, sub-pixel g): register j of (blk, f) = channel f C/4 + 4 blk + j of the shuffled output's pixel (2 h + g / 2, 2 w + g % 2)
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int h = h0 + RW * wave + r, w = w0 + n;
            if (h < a.H && w < a.W) {
                bf16_t* op = a.dst + (((long)b * 2 * a.H + 2 * h + (g >> 1)) * (2 * a.W) + 2 * w + (g & 1)) * a.lddst;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int bp = 0; bp < NB / 2; ++bp) {
                        const f32x4_t lo4 = acc[r][2 * bp][f], hi4 = acc[r][2 * bp + 1][f];
                        *(u32x4_t*)(op + f * (C / 4) + 8 * bp) = u32x4_t{pack2bf(lo4[0], lo4[1]), pack2bf(lo4[2], lo4[3]), pack2bf(hi4[0], hi4[1]), pack2bf(hi4[2], hi4[3])};
                    }
            }
        }
    }
}

// unpack forward, third form (round 4, any C = 32 .. 256): the SPATIAL taps in K and the depth taps in the output rows.  Per pixel and block of
// 16 consecutive depths
//     Y[(f, kd)][d] = sum_{tap = (kh, kw)}  w[f][kd][tap] * x[pixel + tap][d]          one v_mfma_f32_16x16x16_bf16: rows n = 4 f + kd (12 of 16), K = 9 of 16 taps
//     out_f[d]      = Y[(f, 0)][d - 1] + Y[(f, 1)][d] + Y[(f, 2)][d + 1] + b3[f]        two DPP row shifts (the 16 lanes of a row = the block's 16 depths)
// 42 % of the MFMA carries weight (the banded form: 9 %) and the weight operand is CONSTANT: four registers (bf16 hi + lo parts) for the whole
// kernel, no v_perm per tap -- 2 MFMAs + ~10 VALU per 64 outputs, so the kernel is as fast as its memory traffic.  The data operand is an im2col
// column block [16 taps][16 depths] that is never stored: each K row is 32 contiguous bytes of the LDS tile (16 depths of one shifted pixel) and
// ds_read_b64_tr_b16 hands lane (depth, k-group) its 4 taps -- one LDS instruction per MFMA pair.  Block edges (d - 1 of lane 0, d + 1 of lane 15)
// come from the neighbouring blocks' accumulators by row rotates.  Results go through an LDS image laid out as the pixel-shuffled output wants
// it ([pixel][sub-pixel d % 4][channel f C/4 + d / 4]) and leave as 16-byte chunks of whole records.
__device__ __forceinline__ float own_vgpr(float v) { asm volatile("" : "+v"(v)); return v; }

template <int C, int TH, int NH>                       // NH: the output image is built and stored in NH passes (a smaller image: more workgroups per CU)
__global__ __launch_bounds__(256, 2) void unpack3d_fwd_tr_kernel(P3LArgs a) {
    constexpr int TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW, NB = C / 16, NCH = C / 8;
    constexpr int RS = (C * 2) % 128 == 64 ? C * 2 : C * 2 + 64;   // bytes of a tile pixel: an odd multiple of 64 (the 4 rows of a transposing read fall on different bank groups)
    constexpr int IN_BYTES = NPIX * RS, OPX = 8 * C;               // output image: 4 sub-pixel records of C channels per pixel
    constexpr int PPW = TH * TW / 4 / NH;                          // pixels per wave and pass
    static_assert(TH * TW % (4 * NH) == 0, "passes");
    typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    char* tin = smem_;
    char* tout = smem_ + IN_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b, h0, w0;
    up_tile_coords(a, xcd_remap(blockIdx.x, gridDim.x), b, h0, w0);
    // ---- tile + halo of x, plain [pixel][depth]; addresses clamped, values selected to zero
    constexpr int ITEMS = NPIX * NCH, NIT = (ITEMS + 255) / 256;
    u32x4_t v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        const int ch = idx % NCH, t = idx / NCH;
        const int hh = h0 - 1 + t / PW, ww = w0 - 1 + t % PW;
        const bool ok = idx < ITEMS && (unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W;
        const u32x4_t r = *(const u32x4_t*)(a.x + (((long)b * a.H + min(max(hh, 0), a.H - 1)) * a.W + min(max(ww, 0), a.W - 1)) * a.ldx + 8 * ch);
        v[it] = u32x4_t{ok ? r[0] : 0u, ok ? r[1] : 0u, ok ? r[2] : 0u, ok ? r[3] : 0u};
    }
    // ---- the constant weight operand: row n = lane % 16 = 4 f + kd, K entries 4 kg .. 4 kg + 3 = taps; bf16 hi and lo parts
    const int n = lane & 15, kg = lane >> 4;
    s16x4_t whi, wlo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int tap = 4 * kg + e, f = n >> 2, kd = n & 3;
        const float w = (kd < 3 && tap < 9) ? a.w3[(f * 3 + kd) * 9 + tap] : 0.f;
        const bf16_t hi = f2bf(w);
        whi[e] = (short)hi; wlo[e] = (short)f2bf(w - bf2f(hi));
    }
    const float bias = a.b3[kg];                                   // accumulator rows of this lane: feature kg
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        if (idx < ITEMS) *(u32x4_t*)(tin + (idx / NCH) * RS + (idx % NCH) * 16) = v[it];
    }
    __syncthreads();
    // ---- this lane's row of the transposing read: tap 4 kg + q (taps >= 9 carry zero weight: any valid row), depths 4 p .. 4 p + 3 of the block
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int tap = 4 * kg + q < 9 ? 4 * kg + q : 0;
    const int tap_off = ((tap / 3) * PW + tap % 3) * RS + 8 * p;
#pragma unroll 1
    for (int pass = 0; pass < NH; ++pass) {
    if (pass) __syncthreads();                                     // the previous pass's image has been read out
#pragma unroll 1
    for (int pi = 0; pi < PPW; ++pi) {
        const int lp = wave * PPW + pi, pix = pass * (TH * TW / NH) + lp, py = pix / TW, px = pix % TW;   // output pixel of the tile (lp: its slot in this pass's image); its 3 x 3 window starts at tile pixel (py, px)
        const char* row = tin + (py * PW + px) * RS + tap_off;
        f32x4_t acc[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(row + 32 * blk));
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi, x, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo, x, acc[blk], 0, 0, 0);
        }
        // lane (m = lane % 16, f = kg): acc[blk][kd] = Y[(f, kd)][16 blk + m]
        char* orow = tout + lp * OPX + (n & 3) * (2 * C) + (kg * (C / 4) + (n >> 2)) * 2;      // sub-pixel m % 4, channel f C/4 + 4 blk + m / 4
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            // Y[(f, 0)] of depth d - 1 and Y[(f, 2)] of depth d + 1: row shifts with zero fill; the block's end lanes take the neighbouring block's
            // value from a row rotate.  own_vgpr: hipcc (ROCm 7.2) drops the sub-register index when a DPP move reads an element of an MFMA
            // result directly -- element 2 was read as element 0 (tools/probe/dpp_probe.hip has the lane semantics, the kd = 0 / kd = 2 delta weights of
            // tests/test_gpu_pack3d_variants.py catch the mix-up)
            const float y0 = own_vgpr(acc[blk][0]), y2 = own_vgpr(acc[blk][2]);
            float t0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y0), 0x111, 0xF, 0xF, true));             // row_shr:1: lane m <- lane m - 1
            float t2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y2), 0x101, 0xF, 0xF, true));             // row_shl:1: lane m <- lane m + 1
            if (blk > 0) {
                const float p0 = own_vgpr(acc[blk - 1][0]);
                const float e = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p0), 0x121, 0xF, 0xF, true));  // row_ror:1: lane 0 <- lane 15
                t0 = n == 0 ? e : t0;
            }
            if (blk + 1 < NB) {
                const float n2 = own_vgpr(acc[blk + 1][2]);
                const float e = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, n2), 0x12F, 0xF, 0xF, true));  // row_ror:15: lane 15 <- lane 0
                t2 = n == 15 ? e : t2;
            }
            const float o = (t0 + acc[blk][1]) + (t2 + bias);
            *(bf16_t*)(orow + blk * 8) = f2bf(o);
        }
    }
    __syncthreads();
    // ---- output image -> the four sub-pixel records of every pixel, 16-byte chunks
    constexpr int OCH = TH * TW / NH * 4 * NCH;
    for (int i = tid; i < OCH; i += 256) {
        const int ch = i % NCH, sp = (i / NCH) & 3, lp = i / (4 * NCH), pix = pass * (TH * TW / NH) + lp;
        const int h = h0 + pix / TW, w = w0 + pix % TW;
        if (h < a.H && w < a.W)
            *(u32x4_t*)(a.dst + (((long)b * 2 * a.H + 2 * h + (sp >> 1)) * (2 * a.W) + 2 * w + (sp & 1)) * a.lddst + 8 * ch) = *(const u32x4_t*)(tout + lp * OPX + sp * (2 * C) + ch * 16);
    }
    }   // passes
}

// pack forward (space-to-depth + conv3d 1 -> 4), the same taps-in-K form for the un-folded pack layers and the border bands (round 4).  The depth
// axis is long here (D = 4 C = 256 .. 2048), so a workgroup takes a tile of TH x 16 packed pixels AND a slab of DS = 128 depths; the tile is staged
// with 32 more depths either side (one 8-channel chunk of each sub-pixel: aligned loads; zeros outside [0, D)), interleaved in registers to
// d = 4 c + s as in stage_packed_tile, and one extra 16-depth block is multiplied on each side of the slab for the edge terms Y[(f,0)][d - 1] /
// Y[(f,2)][d + 1].  Output: plain feature planes [f D + d] through an LDS image, 16-byte chunks.
template <int TH, int NH>
__global__ __launch_bounds__(256, 2) void pack3d_fwd_tr_kernel(P3LArgs a) {
    constexpr int DS = 128, TW = 16, PW = TW + 2, NPIX = (TH + 2) * PW, NBX = DS / 16 + 2, QPP = (DS + 64) / 32;
    constexpr int RS = 448;                                        // (DS + 64) * 2 = 384 bytes of depths per tile pixel -> an odd multiple of 64
    constexpr int IN_BYTES = NPIX * RS, OPX = 4 * DS * 2;          // output image: 4 feature planes x DS depths per pixel
    constexpr int PPW = TH * TW / 4 / NH;
    static_assert(TH * TW % (4 * NH) == 0, "passes");
    typedef __attribute__((address_space(3))) s16x4_t* lds4_t;
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    char* tin = smem_;
    char* tout = smem_ + IN_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = 4 * a.C, H2 = a.H >> 1, W2 = a.W >> 1, nslabs = D / DS;
    const int slab = blockIdx.x % nslabs, S0 = slab * DS;
    int b, h0, w0;
    tile_coords(a, blockIdx.x / nslabs, b, h0, w0);
    // ---- item = (tile pixel, 8 channels c0 .. c0 + 7 of the four sub-pixels) = 32 depths 4 c0 .. 4 c0 + 31
    constexpr int ITEMS = NPIX * QPP, NIT = (ITEMS + 255) / 256;
    u32x4_t v[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        const int q = idx % QPP, t = idx / QPP;
        const int hh = h0 - 1 + t / PW, ww = w0 - 1 + t % PW, c0 = S0 / 4 - 8 + 8 * q;
        const bool ok = idx < ITEMS && (unsigned)hh < (unsigned)H2 && (unsigned)ww < (unsigned)W2 && (unsigned)c0 < (unsigned)a.C;
        const bf16_t* src = a.x + (((long)b * a.H + 2 * min(max(hh, 0), H2 - 1)) * a.W + 2 * min(max(ww, 0), W2 - 1)) * a.ldx + min(max(c0, 0), a.C - 8);
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            const u32x4_t r = *(const u32x4_t*)(src + ((long)(sp >> 1) * a.W + (sp & 1)) * a.ldx);
            v[it][sp] = u32x4_t{ok ? r[0] : 0u, ok ? r[1] : 0u, ok ? r[2] : 0u, ok ? r[3] : 0u};
        }
    }
    const int n = lane & 15, kg = lane >> 4;
    s16x4_t whi, wlo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int tap = 4 * kg + e, f = n >> 2, kd = n & 3;
        const float w = (kd < 3 && tap < 9) ? a.w3[(f * 3 + kd) * 9 + tap] : 0.f;
        const bf16_t hi = f2bf(w);
        whi[e] = (short)hi; wlo[e] = (short)f2bf(w - bf2f(hi));
    }
    const float bias = a.b3[kg];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        if (idx < ITEMS) {
            char* dstp = tin + (idx / QPP) * RS + (idx % QPP) * 64;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {                      // depths 8 qq .. 8 qq + 7 of the item = channels 2 qq, 2 qq + 1 x 4 sub-pixels (see stage_packed_tile)
                u32x4_t o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned lo = v[it][2 * (k & 1)][qq], hi = v[it][2 * (k & 1) + 1][qq];
                    o[k] = (k >> 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
                }
                *(u32x4_t*)(dstp + 16 * qq) = o;
            }
        }
    }
    __syncthreads();
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int tap = 4 * kg + q < 9 ? 4 * kg + q : 0;
    const int tap_off = ((tap / 3) * PW + tap % 3) * RS + 8 * p + 32;          // + 32 bytes: block 0 starts 16 depths below the slab (tile depth S0 - 32 at byte 0)
#pragma unroll 1
    for (int pass = 0; pass < NH; ++pass) {
    if (pass) __syncthreads();
#pragma unroll 1
    for (int pi = 0; pi < PPW; ++pi) {
        const int lp = wave * PPW + pi, pix = pass * (TH * TW / NH) + lp, py = pix / TW, px = pix % TW;
        const char* row = tin + (py * PW + px) * RS + tap_off;
        f32x4_t acc[NBX];
#pragma unroll
        for (int blk = 0; blk < NBX; ++blk) {
            const s16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(row + 32 * blk));
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi, x, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo, x, acc[blk], 0, 0, 0);
        }
        char* orow = tout + lp * OPX + kg * (DS * 2) + n * 2;      // plane f = kg, depth 16 (blk - 1) + m of the slab
#pragma unroll
        for (int blk = 1; blk + 1 < NBX; ++blk) {
            const float y0 = own_vgpr(acc[blk][0]), y2 = own_vgpr(acc[blk][2]), p0 = own_vgpr(acc[blk - 1][0]), n2 = own_vgpr(acc[blk + 1][2]);
            float t0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y0), 0x111, 0xF, 0xF, true));
            float t2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y2), 0x101, 0xF, 0xF, true));
            const float e0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p0), 0x121, 0xF, 0xF, true));
            const float e2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, n2), 0x12F, 0xF, 0xF, true));
            t0 = n == 0 ? e0 : t0;
            t2 = n == 15 ? e2 : t2;
            const float o = (t0 + acc[blk][1]) + (t2 + bias);
            *(bf16_t*)(orow + (blk - 1) * 32) = f2bf(o);
        }
    }
    __syncthreads();
    constexpr int OCH = TH * TW / NH * 4 * (DS / 8);
    for (int i = tid; i < OCH; i += 256) {
        const int ch = i % (DS / 8), f = (i / (DS / 8)) & 3, lp = i / (4 * (DS / 8)), pix = pass * (TH * TW / NH) + lp;
        const int h = h0 + pix / TW, w = w0 + pix % TW;
        if (h < H2 && w < W2)
            *(u32x4_t*)(a.dst + (((long)b * H2 + h) * W2 + w) * a.lddst + (long)f * D + S0 + 8 * ch) = *(const u32x4_t*)(tout + lp * OPX + f * (DS * 2) + ch * 16);
    }
    }   // passes
}

// dw3/db3 for UNPACK: x tile staged directly; the 8 feature gradients of an item are gathered from the shuffled dout
__global__ __launch_bounds__(256) void unpack3d_bwd_weight_lds_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)smem_;
    __shared__ float sred[4 * 112];
    const int D = a.C, dbs = D >> 3, PW = a.TW + 2, npix = (a.TH + 2) * PW;
    const int items = a.TH * a.TW * dbs;
    float acc[112];
#pragma unroll
    for (int i = 0; i < 112; ++i) acc[i] = 0.f;
    for (int tl = blockIdx.x; tl < a.ntiles; tl += gridDim.x) {
        int b, h0, w0;
        up_tile_coords(a, tl, b, h0, w0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < npix * dbs; idx += 256) {
            const int dc = idx % dbs; const int p = idx / dbs;
            const int hh = h0 - 1 + p / PW, ww = w0 - 1 + p % PW;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if ((unsigned)hh < (unsigned)a.H && (unsigned)ww < (unsigned)a.W)
                v = *(const u32x4_t*)(a.x + (((long)b * a.H + hh) * a.W + ww) * a.ldx + dc * 8);
            *(u32x4_t*)(tile + p * LDP(D) + dc * 8) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int it = threadIdx.x; it < items; it += 256) {
            const int db = it % dbs; const int p = it / dbs;
            const int pw = p % a.TW, ph = p / a.TW;
            const int h = h0 + ph, w = w0 + pw;
            if (h >= a.H || w >= a.W) continue;
            const int d0 = db * 8;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                float go[8];
                const int q0 = f * a.C + d0;                           // multiple of 8: channels q0>>2, +1 at the 4 sub-pixel positions
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const unsigned u = *(const unsigned*)(a.o + (((long)b * 2 * a.H + 2 * h + (s4 >> 1)) * (2 * a.W) + 2 * w + (s4 & 1)) * a.ldo + (q0 >> 2));
                    go[s4] = __uint_as_float(u << 16); go[4 + s4] = __uint_as_float(u & 0xffff0000u);
                }
                float sb = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sb += go[i];
                acc[108 + f] += sb;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        float pv[10];
                        lds_window<1>(tile, (ph + kh) * PW + pw + kw, D, d0, pv);
#pragma unroll
                        for (int kd = 0; kd < 3; ++kd) {
                            float sacc = 0.f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) sacc = fmaf(go[i], pv[i + kd], sacc);
                            acc[((f * 3 + kd) * 3 + kh) * 3 + kw] += sacc;
                        }
                        asm volatile("" ::: "memory");
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
    if (threadIdx.x < 112)
        atomicAdd(a.dwb + threadIdx.x, sred[threadIdx.x] + sred[112 + threadIdx.x] + sred[224 + threadIdx.x] + sred[336 + threadIdx.x]);
}

// ---- dw3 / db3 on the matrix cores (pack and unpack).  dW3[f][tap] = sum over positions of G[f][pos] * V[pos + off(tap)] is a GEMM
// with M = 4 features, N = 27 taps (+ one all-ones column that yields db3) and K = every (pixel, depth) position of the volume:
// one v_mfma_f32_32x32x16_bf16 takes 16 positions (2 x 8 consecutive depths of one pixel) for all 4 x 28 outputs at once.  Only 4
// of the 32 rows carry data -- a tenth of the MFMA rate is still 4x what the fp32-VALU version above reached (it spent 864 FMAs +
// 36 window reads per 8 depths and ran 6x over its HBM time).
//   A operand: lane (row f = lane & 31 < 4, half = lane >> 5) holds G[f][pixel][d0 + 8 half .. +8) -- a 16-byte global load (pack)
//              or four 4-byte loads from the pixel-shuffled gradient + a 16-bit interleave (unpack); rows >= 4 are zero.
//   B operand: lane (column = tap = (kd*3 + kh)*3 + kw) holds V[pixel + (kh-1, kw-1)][d0 + 8 half + kd - 1 .. +8) from the LDS tile:
//              the aligned 16-byte chunk plus the 4-byte words either side of it, shifted by one element with v_alignbyte
//              according to the lane's kd.  The 8 pad elements after every pixel's depths (and 16 bytes in front of the tile)
//              are kept zero, so depths -1 and D read as zero without a branch.
//   Accumulators: lane (column = tap) of half 0 holds rows 0..3 = the four features in registers 0..3.
template <bool UNPACK>
__global__ __launch_bounds__(512) void conv3d_bwd_weight_mfma_kernel(P3LArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_[];
    bf16_t* tile = (bf16_t*)(smem_ + 16);
    __shared__ float sred[8 * 112];
    __shared__ __attribute__((aligned(16))) char scratch[8 * 2 * 1024];        // per wave: two 1 KB gradient blocks (see the round loop)
    const int D = UNPACK ? a.C : 4 * a.C;
    const int H2 = UNPACK ? a.H : a.H >> 1, W2 = UNPACK ? a.W : a.W >> 1;     // volume extent in pixels
    const int PW = a.TW + 2, npix = (a.TH + 2) * PW, dbs = D >> 3, dpairs = D >> 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the item index arithmetic stays on the SALU)
    const int r = lane & 31, half = lane >> 5, nwaves = blockDim.x >> 6;
    const int tap = r < 27 ? r : 0;
    const int kd = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
    const int tapoff = (kh * PW + kw) * LDP(D) + 8 * half;                      // elements, relative to the item's pixel / depth pair
    const bool up1 = kd >= 1;
    const unsigned sh = kd == 1 ? 0u : 2u;
    // constant operand lanes read constants: column 27 (the all-ones column that yields db3) from a run of bf16 ones, columns 28..31 and the A rows 4..31 from a
    // run of zeros -- one address select per fragment instead of eight value selects (round 5, end: the loop is instruction-bound)
    // (the runs are long enough for the whole round: the fragment of depth pair u sits u * 32 bytes further, as an immediate offset)
    __shared__ __attribute__((aligned(16))) unsigned s_const[208];               // [0, 80): 0x3f803f80, [80, 208): 0
    if (threadIdx.x < 208) s_const[threadIdx.x] = threadIdx.x < 80 ? 0x3f803f80u : 0u;
    const bf16_t* const cpb = (const bf16_t*)(s_const + (r == 27 ? 4 : 84));    // (a 16-byte chunk with a word either side inside its run, + 7 x 32 bytes)
    const char* const czero = (const char*)(s_const + 80);
    const bool clane = r >= 27;
    if (threadIdx.x == 0) *(u32x4_t*)smem_ = u32x4_t{0u, 0u, 0u, 0u};
    for (int p = threadIdx.x; p < npix; p += blockDim.x) *(u32x4_t*)(tile + p * LDP(D) + D) = u32x4_t{0u, 0u, 0u, 0u};
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int tl = blockIdx.x; tl < a.ntiles; tl += gridDim.x) {
        int b, h0, w0;
        if constexpr (UNPACK) up_tile_coords(a, tl, b, h0, w0); else tile_coords(a, tl, b, h0, w0);
        __syncthreads();
        if constexpr (UNPACK) {
            fill_tile_chunks<4>(tile, a.x + (long)b * a.H * a.W * a.ldx, a.ldx, a.H, a.W, h0, w0, PW, npix, dbs, LDP(D));
        } else {
            stage_packed_tile(a, tile, b, h0, w0);
        }
        __syncthreads();
        // A wave owns the tile pixels p = wave, wave + nwaves, ...; one ROUND = one pixel x U depth pairs (128 depths).  The round's
        // gradient block (4 features x 128 depths = 1 KB) is fetched by ONE 16-byte load per lane, a round ahead, parked in the wave's
        // LDS scratch, and the 8 lanes of an item's A fragment pick their chunks out of it: fetching fragments straight from global
        // memory used 8 of 64 lanes per load instruction (unpack: 4 instructions per item) and the kernel ran at the texture
        // addresser's instruction rate, 180-370 cycles per MFMA.
        constexpr int U = 8;
        const int rpp = (dpairs + U - 1) / U;                    // rounds per pixel
        const int npx = (a.TH * a.TW - wave + nwaves - 1) / nwaves;
        const int nrounds = npx * rpp;
        int fj = 0, fdp = 0;                                      // the fetcher's own position (it runs one round ahead): advanced by every call
        auto fetch = [&](int) -> u32x4_t {                       // this lane's 16 bytes of the next round (zeros outside the image / depth range)
            const int j = fj, dp0 = fdp;
            fdp += U;
            if (fdp >= rpp * U) { fdp = 0; ++fj; }
            const int p = wave + j * nwaves;
            const int pw = p & (a.TW - 1), ph = p >> a.tshift;   // (TW is a power of two in every tile table)
            const int h = h0 + ph, w = w0 + pw;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (h < H2 && w < W2) {
                const int f = lane >> 4;
                if constexpr (UNPACK) {
                    const int sp = (lane >> 2) & 3, q = lane & 3;                       // sub-pixel, 8-channel part
                    if (dp0 * 4 + 8 * q < (a.C >> 2))
                        v = *(const u32x4_t*)(a.o + (((long)b * 2 * a.H + 2 * h + (sp >> 1)) * (2 * a.W) + 2 * w + (sp & 1)) * a.ldo + f * (a.C >> 2) + dp0 * 4 + 8 * q);
                } else {
                    const int c = lane & 15;
                    if (dp0 * 16 + c * 8 < D) v = *(const u32x4_t*)(a.o + (((long)b * H2 + h) * W2 + w) * a.ldo + f * D + dp0 * 16 + c * 8);
                }
            }
            return v;
        };
        u32x4_t cur = nrounds > 0 ? fetch(0) : u32x4_t{0u, 0u, 0u, 0u};
        // (round 5, end: the round's pixel and depth-pair offset are carried along instead of being divided out of k twice per round -- with fetch(k + 1) that was
        //  ~140 scalar instructions per round of 8 MFMAs; the loop is instruction-bound: 45-58 instructions per MFMA before, see profiles/README.md round 5)
        int rj = 0, rdp = 0;                                      // k = rj * rpp + rdp / U
#pragma unroll 1
        for (int k = 0; k < nrounds; ++k) {
            u32x4_t nxt = {0u, 0u, 0u, 0u};
            if (k + 1 < nrounds) nxt = fetch(k + 1);
            const int j = rj, dp0 = rdp;
            rdp += U;
            if (rdp >= rpp * U) { rdp = 0; ++rj; }
            const int p = wave + j * nwaves;
            const int pw = p & (a.TW - 1), ph = p >> a.tshift;
            if (h0 + ph < H2 && w0 + pw < W2) {                  // wave-uniform
                char* sc = scratch + (wave * 2 + (k & 1)) * 1024;
                *(u32x4_t*)(sc + lane * 16) = cur;
                // (kd = 0 lanes start one word early: every lane then reads the five consecutive words w0 .. w4 its shifted fragment is cut from)
                const unsigned* pw0 = (const unsigned*)(clane ? cpb : tile + (ph * PW + pw) * LDP(D) + tapoff + dp0 * 16) - (up1 ? 0 : 1);
                const char* ga = r < 4 ? (UNPACK ? sc + r * 256 + half * 4 : sc + (r * 16 + half) * 16) : czero;     // this lane's part of the gradient block (rows >= 4: zeros)
                const int nd = min(U, dpairs - dp0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u >= nd) break;                          // wave-uniform
                    u32x4_t fa;
                    if constexpr (UNPACK) {
                        // depths d0 .. d0+7 of feature r: channels (d0 >> 2, + 1) at the four sub-pixel positions = word 2(u&1) + half of part u >> 1
                        const char* g = ga + (u >> 1) * 16 + 2 * (u & 1) * 4;
                        const unsigned u0 = *(const unsigned*)g, u1 = *(const unsigned*)(g + 64), u2 = *(const unsigned*)(g + 128), u3 = *(const unsigned*)(g + 192);
                        fa = u32x4_t{(u0 & 0xffffu) | (u1 << 16), (u2 & 0xffffu) | (u3 << 16), (u0 >> 16) | (u1 & 0xffff0000u), (u2 >> 16) | (u3 & 0xffff0000u)};
                    } else {
                        fa = *(const u32x4_t*)(ga + u * 32);
                    }
                    const unsigned* wq = pw0 + u * 8;
                    const unsigned y0 = wq[0], y1 = wq[1], y2 = wq[2], y3 = wq[3], y4 = wq[4];      // (word-aligned only: ds_read2_b32 pairs)
                    u32x4_t fb = {__builtin_amdgcn_alignbyte(y1, y0, sh), __builtin_amdgcn_alignbyte(y2, y1, sh),
                                  __builtin_amdgcn_alignbyte(y3, y2, sh), __builtin_amdgcn_alignbyte(y4, y3, sh)};
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa), __builtin_bit_cast(bf16x8_t, fb), acc, 0, 0, 0);
                }
            }
            cur = nxt;
        }
    }
    // D[row][col]: lane holds column lane & 31, rows (e & 3) + 8 (e >> 2) + 4 half: features 0..3 = registers 0..3 of half 0
    if (half == 0 && r < 28) {
#pragma unroll
        for (int f = 0; f < 4; ++f) sred[wave * 112 + (r < 27 ? f * 27 + r : 108 + f)] = acc[f];
    }
    __syncthreads();
    if (threadIdx.x < 112) {
        float t = 0.f;
        for (int wv = 0; wv < nwaves; ++wv) t += sred[wv * 112 + threadIdx.x];
        atomicAdd(a.dwb + threadIdx.x, t);
    }
}

inline P3LArgs upl_args(int B, int H, int W, int C, bool half_tile = false) {
    P3LArgs a{}; a.B = B; a.H = H; a.W = W; a.C = C;
    P3Tile t = up_tile(C);
    if (half_tile && t.TH >= 4) t.TH /= 2;               // kernels with generic item loops: half the LDS, twice the resident blocks
    a.TH = t.TH; a.TW = t.TW;
    a.tiles_h = (H + t.TH - 1) / t.TH; a.tiles_w = (W + t.TW - 1) / t.TW; a.ntiles = a.tiles_h * a.tiles_w * B;
    return a;
}

template <typename KF> int launch_p3l(KF kf, P3LArgs a, int grid, hipStream_t st, size_t lds_override = 0, int threads = 256) {
    const size_t lds = lds_override ? lds_override : p3_lds_bytes(a.C);
    static const void* done[32] = {};
    bool seen = false;
    for (int i = 0; i < 32; ++i) seen = seen || done[i] == (const void*)kf;
    if (!seen) {
        if (hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) != hipSuccess) return MTE_ERR_LAUNCH;
        for (int i = 0; i < 32; ++i) if (!done[i]) { done[i] = (const void*)kf; break; }
    }
    hipLaunchKernelGGL(kf, dim3(grid), dim3(threads), lds, st, a);
    return mte_check_launch();
}

inline P3LArgs p3l_args(int B, int H, int W, int C) {
    P3LArgs a{}; a.B = B; a.H = H; a.W = W; a.C = C;
    const P3Tile t = p3_tile(C); a.TH = t.TH; a.TW = t.TW;
    a.tiles_h = (H / 2 + t.TH - 1) / t.TH; a.tiles_w = (W / 2 + t.TW - 1) / t.TW; a.ntiles = a.tiles_h * a.tiles_w * B;
    return a;
}

int g_p3_lds = 2;                                   // development knob (mte_debug_set(1, v))
int g_p3_mfma_threads = 512;
int g_p3_mfma = 1;                                  // development knob (mte_debug_set(1, 200 + v)): conv3d weight gradients on the matrix cores

}  // namespace

#ifdef MTE_DEV
extern "C" int mtei_set_pack3d_lds(int value) { if (value >= 3000) { g_p3_tr_passes = value - 3000 == 1 ? 1 : (value - 3000 == 2 ? 2 : 4); return MTE_OK; } if (value >= 2000) { g_p3_persist_wgs = value - 2000; return MTE_OK; } if (value >= 1000) { g_p3_mfma_threads = value - 1000; return MTE_OK; } if (value >= 300) { g_p3_mfma_data = value - 300; return MTE_OK; } if (value >= 200) { g_p3_mfma = value - 200; return MTE_OK; } if (value >= 100) { g_p3_small_tiles = value - 100; return MTE_OK; } g_p3_lds = value; return MTE_OK; }
#endif


extern "C" {

// out[B,H/2,W/2,16C] = conv3d(pixel_unshuffle(x[B,H,W,C]))            (PackLayerConv3d up to its Conv2D)
int mte_pack3d_fwd(const void* x, long ldx, const float* w3, const float* b3, void* out, long ldo,
                   int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !w3 || !b3 || !out || !p3_ok(C) || (H & 1) || (W & 1)) return MTE_ERR_ARG;
    P3Args a{}; a.x = x; a.ldx = ldx; a.dst = out; a.lddst = ldo; a.w3 = w3; a.b3 = b3; a.B = B; a.H = H; a.W = W; a.C = C;
    a.total = (long)B * (H / 2) * (W / 2) * (C / 8);
    if (dtype == MTE_DT_BF16 && g_p3_lds >= 2 && (g_p3_mfma_data & 64) && C % 32 == 0) {
        P3LArgs l{}; l.B = B; l.H = H; l.W = W; l.C = C; l.TH = 2; l.TW = 16;
        l.tiles_h = (H / 2 + l.TH - 1) / l.TH; l.tiles_w = (W / 2 + l.TW - 1) / l.TW; l.ntiles = l.tiles_h * l.tiles_w * B;
        l.x = (const bf16_t*)x; l.ldx = ldx; l.dst = (bf16_t*)out; l.lddst = ldo; l.w3 = w3; l.b3 = b3;
        const long grid = (long)l.ntiles * (4 * C / 128);
        if (grid < (1L << 30)) return launch_p3l(pack3d_fwd_tr_kernel<2, 4>, l, (int)grid, stream, (size_t)4 * 18 * 448 + (size_t)8 * 1024);
    }
    if (dtype == MTE_DT_BF16 && g_p3_lds && C % 8 == 0 && C <= 512) {
        P3LArgs l = p3l_args(B, H, W, C); l.x = (const bf16_t*)x; l.ldx = ldx; l.dst = (bf16_t*)out; l.lddst = ldo; l.w3 = w3; l.b3 = b3;
        return launch_p3l(pack3d_fwd_lds_kernel, l, l.ntiles, stream);
    }
    if (C <= 256) { a.total *= 2; return launch_p3(dtype, pack3d_fwd_kernel<bf16_t, 4>, pack3d_fwd_kernel<float, 4>, a, a.total, stream); }
    return launch_p3(dtype, pack3d_fwd_kernel<bf16_t, 8>, pack3d_fwd_kernel<float, 8>, a, a.total, stream);
}
int mte_pack3d_bwd_data(const void* dout, long ldo, const float* w3, void* dx, long lddx,
                        int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dout || !w3 || !dx || !p3_ok(C)) return MTE_ERR_ARG;
    P3Args a{}; a.o = dout; a.ldo = ldo; a.dst = dx; a.lddst = lddx; a.w3 = w3; a.B = B; a.H = H; a.W = W; a.C = C;
    a.total = (long)B * (H / 2) * (W / 2) * (C / 8);
    if (dtype == MTE_DT_BF16 && g_p3_lds >= 2 && (g_p3_mfma_data & 128) && C % 32 == 0 && ((long)B * (H / 2) * (W / 2) - 1) * ldo + 16L * C < (1L << 30)) {
        P3LArgs l{}; l.B = B; l.H = H; l.W = W; l.C = C; l.TH = 4; l.TW = 16;
        l.tiles_h = (H / 2 + 3) / 4; l.tiles_w = (W / 2 + 15) / 16; l.ntiles = l.tiles_h * l.tiles_w * B;
        l.o = (const bf16_t*)dout; l.ldo = ldo; l.dst = (bf16_t*)dx; l.lddst = lddx; l.w3 = w3;
        const long grid = (long)l.ntiles * (4 * C / 128);
        if (grid < (1L << 30))
            return (g_p3_mfma_data & 16) ? launch_p3l(pack3d_bwd_data_mfma_kernel<false>, l, (int)grid, stream, (size_t)7 * 20 * 256)
                                         : launch_p3l(pack3d_bwd_data_mfma_kernel<true>, l, (int)grid, stream, (size_t)7 * 20 * 256);
    }
    if (dtype == MTE_DT_BF16 && g_p3_lds && C % 8 == 0 && C <= 512) {
        P3LArgs l = p3l_args(B, H, W, C); l.o = (const bf16_t*)dout; l.ldo = ldo; l.dst = (bf16_t*)dx; l.lddst = lddx; l.w3 = w3;
        return launch_p3l(pack3d_bwd_data_lds_kernel, l, l.ntiles, stream);
    }
    if (C <= 256) { a.total *= 2; return launch_p3(dtype, pack3d_bwd_data_kernel<bf16_t, 4>, pack3d_bwd_data_kernel<float, 4>, a, a.total, stream); }
    return launch_p3(dtype, pack3d_bwd_data_kernel<bf16_t, 8>, pack3d_bwd_data_kernel<float, 8>, a, a.total, stream);
}
// dwb[112] (fp32, zeroed here): [0..107] = dw3, [108..111] = db3
int mte_pack3d_bwd_weight(const void* x, long ldx, const void* dout, long ldo, float* dwb,
                          int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dout || !dwb || !p3_ok(C)) return MTE_ERR_ARG;
    if (mte_memset_async(dwb, 0, 112 * sizeof(float), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    P3Args a{}; a.x = x; a.ldx = ldx; a.o = dout; a.ldo = ldo; a.dw3 = dwb; a.B = B; a.H = H; a.W = W; a.C = C;
    a.total = (long)B * (H / 2) * (W / 2) * (C / 8);
    if (dtype == MTE_DT_BF16 && g_p3_lds && C % 8 == 0 && C <= 512) {
        P3LArgs l = p3l_args(B, H, W, C); l.x = (const bf16_t*)x; l.ldx = ldx; l.o = (const bf16_t*)dout; l.ldo = ldo; l.dwb = dwb;
        { const int dpairs = 4 * C / 16; l.dshift = (dpairs & (dpairs - 1)) == 0 ? __builtin_ctz(dpairs) : -1; l.tshift = __builtin_ctz(l.TW); }
        if (g_p3_mfma) return launch_p3l(conv3d_bwd_weight_mfma_kernel<false>, l, l.ntiles < MTE_P3W_WGS ? l.ntiles : MTE_P3W_WGS, stream, p3_lds_bytes(C) + 16, g_p3_mfma_threads);
        return launch_p3l(pack3d_bwd_weight_lds_kernel, l, l.ntiles < 512 ? l.ntiles : 512, stream);
    }
    if (C <= 256) a.total *= 2;
    long threads = a.total < 256L * 1024 ? a.total : 256L * 1024;
    if (C <= 256) return launch_p3(dtype, pack3d_bwd_weight_kernel<bf16_t, 4>, pack3d_bwd_weight_kernel<float, 4>, a, threads, stream, 3);
    return launch_p3(dtype, pack3d_bwd_weight_kernel<bf16_t, 8>, pack3d_bwd_weight_kernel<float, 8>, a, threads, stream, 3);
}

// out[B,2H,2W,C] = pixel_shuffle(conv3d(x[B,H,W,C]))                   (UnpackLayerConv3d after its Conv2D)
int mte_unpack3d_fwd(const void* x, long ldx, const float* w3, const float* b3, void* out, long ldo,
                     int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !w3 || !b3 || !out || !p3_ok(C)) return MTE_ERR_ARG;
    P3Args a{}; a.x = x; a.ldx = ldx; a.dst = out; a.lddst = ldo; a.w3 = w3; a.b3 = b3; a.B = B; a.H = H; a.W = W; a.C = C;
    if (dtype == MTE_DT_BF16 && g_p3_lds >= 2 && (g_p3_mfma_data & 32) && (C == 32 || C == 64 || C == 128 || C == 256)) {
        P3LArgs l{}; l.B = B; l.H = H; l.W = W; l.C = C;
        l.TH = 256 / C; l.TW = 16;
        l.tiles_h = (H + l.TH - 1) / l.TH; l.tiles_w = (W + l.TW - 1) / l.TW; l.ntiles = l.tiles_h * l.tiles_w * B;
        l.x = (const bf16_t*)x; l.ldx = ldx; l.dst = (bf16_t*)out; l.lddst = ldo; l.w3 = w3; l.b3 = b3;
        const int rs = (C * 2) % 128 == 64 ? C * 2 : C * 2 + 64;
        const int nh = g_p3_tr_passes;
        const size_t lds = (size_t)(l.TH + 2) * 18 * rs + (size_t)l.TH * 16 * 8 * C / nh;
        if (nh == 1) {
            if (C == 32) return launch_p3l(unpack3d_fwd_tr_kernel<32, 8, 1>, l, l.ntiles, stream, lds);
            if (C == 64) return launch_p3l(unpack3d_fwd_tr_kernel<64, 4, 1>, l, l.ntiles, stream, lds);
            if (C == 128) return launch_p3l(unpack3d_fwd_tr_kernel<128, 2, 1>, l, l.ntiles, stream, lds);
            return launch_p3l(unpack3d_fwd_tr_kernel<256, 1, 1>, l, l.ntiles, stream, lds);
        }
        if (nh == 2) {
            if (C == 32) return launch_p3l(unpack3d_fwd_tr_kernel<32, 8, 2>, l, l.ntiles, stream, lds);
            if (C == 64) return launch_p3l(unpack3d_fwd_tr_kernel<64, 4, 2>, l, l.ntiles, stream, lds);
            if (C == 128) return launch_p3l(unpack3d_fwd_tr_kernel<128, 2, 2>, l, l.ntiles, stream, lds);
            return launch_p3l(unpack3d_fwd_tr_kernel<256, 1, 2>, l, l.ntiles, stream, lds);
        }
        if (C == 32) return launch_p3l(unpack3d_fwd_tr_kernel<32, 8, 4>, l, l.ntiles, stream, lds);
        if (C == 64) return launch_p3l(unpack3d_fwd_tr_kernel<64, 4, 4>, l, l.ntiles, stream, lds);
        if (C == 128) return launch_p3l(unpack3d_fwd_tr_kernel<128, 2, 4>, l, l.ntiles, stream, lds);
        return launch_p3l(unpack3d_fwd_tr_kernel<256, 1, 4>, l, l.ntiles, stream, lds);
    }
    if (dtype == MTE_DT_BF16 && g_p3_lds >= 2 && (g_p3_mfma_data & 8) && (C == 32 || C == 64) && ((long)B * H * W - 1) * ldx + C < (1L << 30)) {
        P3LArgs l{}; l.B = B; l.H = H; l.W = W; l.C = C;
        l.TH = C == 32 ? 8 : 4; l.TW = 16;
        l.tiles_h = (H + l.TH - 1) / l.TH; l.tiles_w = (W + l.TW - 1) / l.TW; l.ntiles = l.tiles_h * l.tiles_w * B;
        l.x = (const bf16_t*)x; l.ldx = ldx; l.dst = (bf16_t*)out; l.lddst = ldo; l.w3 = w3; l.b3 = b3;
        const size_t lds = (size_t)2 * (((l.TH + 2) * 18 + 15) / 16) * (C / 8 + 2) * 256;      // two tile buffers
        const int grid = l.ntiles < g_p3_persist_wgs ? l.ntiles : g_p3_persist_wgs;            // persistent workgroups (a multiple of 8: tile % 8 = XCD)
        if (g_p3_mfma_data & 16) {
            if (C == 32) return launch_p3l(unpack3d_fwd_mfma_kernel<32, false>, l, grid, stream, lds);
            return launch_p3l(unpack3d_fwd_mfma_kernel<64, false>, l, grid, stream, lds);
        }
        if (C == 32) return launch_p3l(unpack3d_fwd_mfma_kernel<32, true>, l, grid, stream, lds);
        return launch_p3l(unpack3d_fwd_mfma_kernel<64, true>, l, grid, stream, lds);
    }
    a.total = (long)B * H * W * (C / 8);
    return launch_p3(dtype, unpack3d_fwd_kernel<bf16_t>, unpack3d_fwd_kernel<float>, a, a.total, stream);
}
int mte_unpack3d_bwd_data(const void* dout, long ldo, const float* w3, void* dx, long lddx,
                          int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dout || !w3 || !dx || !p3_ok(C)) return MTE_ERR_ARG;
    P3Args a{}; a.o = dout; a.ldo = ldo; a.dst = dx; a.lddst = lddx; a.w3 = w3; a.B = B; a.H = H; a.W = W; a.C = C;
    if (dtype == MTE_DT_BF16 && g_p3_lds >= 2 && (g_p3_mfma_data & 1) && (C == 32 || C == 64)) {
        P3LArgs l{}; l.B = B; l.H = H; l.W = W; l.C = C;
        l.TH = C == 32 ? 8 : 4; l.TW = 16;
        l.tiles_h = (H + l.TH - 1) / l.TH; l.tiles_w = (W + l.TW - 1) / l.TW; l.ntiles = l.tiles_h * l.tiles_w * B;
        l.o = (const bf16_t*)dout; l.ldo = ldo; l.dst = (bf16_t*)dx; l.lddst = lddx; l.w3 = w3;
        const size_t lds = (size_t)4 * (l.TH + 2) * (l.TW + 2) * (C + 16) * 2;
        if (C == 32 && (g_p3_mfma_data & 2) && ((long)B * 4 * H * W - 1) * ldo + 32 < (1L << 30)) {
            if (g_p3_mfma_data & 16) return launch_p3l(unpack3d_bwd_data_dma32_kernel<4, false>, l, l.ntiles, stream, (size_t)180 * 256, 256);
            if (g_p3_mfma_data & 4) return launch_p3l(unpack3d_bwd_data_dma32_kernel<4, true>, l, l.ntiles, stream, (size_t)180 * 256, 256);
            return launch_p3l(unpack3d_bwd_data_dma32_kernel<2, true>, l, l.ntiles, stream, (size_t)180 * 256, 128);
        }
        if (g_p3_mfma_data & 16) {
            if (C == 32) return launch_p3l(unpack3d_bwd_data_mfma_kernel<32, false>, l, l.ntiles, stream, lds);
            return launch_p3l(unpack3d_bwd_data_mfma_kernel<64, false>, l, l.ntiles, stream, lds);
        }
        if (C == 32) return launch_p3l(unpack3d_bwd_data_mfma_kernel<32, true>, l, l.ntiles, stream, lds);
        return launch_p3l(unpack3d_bwd_data_mfma_kernel<64, true>, l, l.ntiles, stream, lds);
    }
    if (dtype == MTE_DT_BF16 && g_p3_lds >= 2 && C % 32 == 0 && C <= 128) {
        P3LArgs l{}; l.B = B; l.H = H; l.W = W; l.C = C;
        const P3Tile t = up4_tile(C); l.TH = t.TH; l.TW = t.TW;
        l.tiles_h = (H + t.TH - 1) / t.TH; l.tiles_w = (W + t.TW - 1) / t.TW; l.ntiles = l.tiles_h * l.tiles_w * B;
        l.o = (const bf16_t*)dout; l.ldo = ldo; l.dst = (bf16_t*)dx; l.lddst = lddx; l.w3 = w3;
        return launch_p3l(unpack3d_bwd_data_lds4_kernel, l, l.ntiles, stream, up4_lds_bytes(C));
    }
    if (dtype == MTE_DT_BF16 && g_p3_lds && C % 32 == 0 && C <= 512) {
        P3LArgs l = upl_args(B, H, W, C); l.o = (const bf16_t*)dout; l.ldo = ldo; l.dst = (bf16_t*)dx; l.lddst = lddx; l.w3 = w3;
        return launch_p3l(unpack3d_bwd_data_lds_kernel, l, l.ntiles, stream, up_lds_bytes(C));
    }
    a.total = (long)B * H * W * (C / 8);
    return launch_p3(dtype, unpack3d_bwd_data_kernel<bf16_t>, unpack3d_bwd_data_kernel<float>, a, a.total, stream);
}
int mte_unpack3d_bwd_weight(const void* x, long ldx, const void* dout, long ldo, float* dwb,
                            int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dout || !dwb || !p3_ok(C)) return MTE_ERR_ARG;
    if (mte_memset_async(dwb, 0, 112 * sizeof(float), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    P3Args a{}; a.x = x; a.ldx = ldx; a.o = dout; a.ldo = ldo; a.dw3 = dwb; a.B = B; a.H = H; a.W = W; a.C = C;
    if (dtype == MTE_DT_BF16 && g_p3_lds && C % 32 == 0 && C <= 512) {
        P3LArgs l = upl_args(B, H, W, C, g_p3_small_tiles != 0); l.x = (const bf16_t*)x; l.ldx = ldx; l.o = (const bf16_t*)dout; l.ldo = ldo; l.dwb = dwb;
        const size_t lds = (size_t)(l.TH + 2) * (l.TW + 2) * LDP(C) * 2;
        const int cap = g_p3_small_tiles ? 1024 : 512;
        { const int dpairs = C / 16; l.dshift = (dpairs & (dpairs - 1)) == 0 ? __builtin_ctz(dpairs) : -1; l.tshift = __builtin_ctz(l.TW); }
        if (g_p3_mfma) return launch_p3l(conv3d_bwd_weight_mfma_kernel<true>, l, l.ntiles < MTE_P3W_WGS ? l.ntiles : MTE_P3W_WGS, stream, lds + 16, g_p3_mfma_threads);
        return launch_p3l(unpack3d_bwd_weight_lds_kernel, l, l.ntiles < cap ? l.ntiles : cap, stream, lds);
    }
    a.total = (long)B * H * W * (C / 8);
    long threads = a.total < 256L * 2048 ? a.total : 256L * 2048;
    return launch_p3(dtype, unpack3d_bwd_weight_kernel<bf16_t>, unpack3d_bwd_weight_kernel<float>, a, threads, stream, 3);
}

}  // extern "C"
