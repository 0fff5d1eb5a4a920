// Small HBM-bound kernels around the conv stack (gfx950, NHWC activations):
//   * InvDepth head: sigmoid(conv3x3(x, C->1) + b) / 0.5 and its backward   (layers01.py:99-123)
//   * NCHW fp32 image -> NHWC compute-dtype tensor with zero channel padding (+ optional horizontal flip,
//     SfmModel.py:58-96 / model_utils.py:98-117)
//   * nearest x2 up-sampling of an inv-depth map into one channel block of a decoder concat buffer
//     (PackNetSAN01.py:92-94,118-143) and its backward
//   * channel-slice copy (skip connections into concat buffers)
//   * fused Adam step over the flat fp32 master-parameter buffer (model_wrapper.py:142-180: Adam, wd 0)
#include "common.hpp"
#include <atomic>

int tap_wgrad_mfma_launch(const bf16_t* x, long ldx, const float* m, float* rec, long rec_stride, int B, int H, int W, int C, int sgn, int up2, int has_db,
                          int max_wgs, int* groups, hipStream_t st);                                // tap_wgrad.hip
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
    float* dw;                              // weight gradient: per-workgroup records [grid][rec_stride] of (dw [C*9], db)
    long rec_stride;
    int B, H, W, C;
    float inv_min_depth;                    // 1 / min_depth = 2
    long npix;
};

// ---- InvDepth head kernels, row-marching form.  thread = (image column x, 8-channel block j); the C/8 lanes of a pixel are
// adjacent lanes of one wave, a wave covers 64 / (C/8) consecutive columns (1 KiB of contiguous NHWC bytes per load instruction).
// A thread walks DOWN a strip of rows and keeps the 3 x 3 window of its column in registers: every new row costs three chunk
// loads (columns x-1, x, x+1 of the incoming row, one row ahead of the arithmetic) instead of nine -- the nine-loads-per-pixel
// form moved 9x the activation bytes through the CU's 64 B/clk vector-cache path and ran 3-4x off the HBM roofline
// (profiles/r02_v4: forward 168 us, weight gradient 203 us for the 252 MB full-resolution feature map).
template <typename T> struct Ch8;                                  // 8 channels of one pixel as loaded
template <> struct Ch8<bf16_t> {
    u32x4_t a;
    __device__ __forceinline__ void load(const bf16_t* p) { a = *(const u32x4_t*)p; }
    __device__ __forceinline__ void zero() { a = u32x4_t{0u, 0u, 0u, 0u}; }
    __device__ __forceinline__ void get(float* v) const { unpack16<bf16_t>(a, v); }
    __device__ __forceinline__ void keep() { asm volatile("" : "+v"(a)); }
};
template <> struct Ch8<float> {
    u32x4_t a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const u32x4_t*)p; b = *(const u32x4_t*)(p + 4); }
    __device__ __forceinline__ void zero() { a = u32x4_t{0u, 0u, 0u, 0u}; b = a; }
    __device__ __forceinline__ void get(float* v) const { unpack16<float>(a, v); unpack16<float>(b, v + 4); }
    __device__ __forceinline__ void keep() { asm volatile("" : "+v"(a), "+v"(b)); }
};

constexpr int HEAD_ROWS = 20;                                       // rows per strip (two halo rows re-read per strip); multiple of the ring periods 4, 5, 20

// A launch runs as many workgroups as the chip holds at once (occupancy x 256 CUs, found once per kernel) and gives every one a
// contiguous, balanced range of strips: with one workgroup per strip the last partial round of workgroups ran at a fraction of the
// occupancy for a whole strip's duration (1,920 strips on 768 resident workgroups = 3 rounds of time for 2.5 rounds of work).
// Every load is UNCONDITIONAL (clamped address, value zeroed afterwards where the tap is outside the image): a load inside a branch
// makes the compiler wait with vmcnt(0) before the first use, which serialised the whole register ring on the row just issued
// (first version of this form: 112 us against 157 us for the nine-loads form; the loop was paying one memory latency per row).
#define HEAD_MAP()                                                                   \
    const int cb = a.C >> 3, j = threadIdx.x % cb, c0 = j * 8, pxw = 256 / cb;       \
    const int strips_x = (a.W + pxw - 1) / pxw, strips_y = (a.H + HEAD_ROWS - 1) / HEAD_ROWS; \
    const long nstrips = (long)strips_x * strips_y * a.B;                            \
    const int s_first = (int)(nstrips * blockIdx.x / gridDim.x), s_last = (int)(nstrips * (blockIdx.x + 1) / gridDim.x);
#define HEAD_STRIP()                                                                 \
    int sid = strip;                                                                 \
    const int sx = sid % strips_x; sid /= strips_x;                                  \
    const int sy = sid % strips_y; const int b = sid / strips_y;                     \
    const int x = sx * pxw + threadIdx.x / cb;                                       \
    const int y0 = sy * HEAD_ROWS, y1 = min(a.H, y0 + HEAD_ROWS);                    \
    const bool xin = x < a.W;                                                        \
    const int xc[3] = {min(max(x - 1, 0), a.W - 1), min(x, a.W - 1), min(x + 1, a.W - 1)};   \
    const bool xok[3] = {xin && x >= 1, xin, xin && x + 1 < a.W};

template <typename T>
__global__ __launch_bounds__(256) void invdepth_fwd_kernel(HeadArgs a) {
    HEAD_MAP();
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[t][i] = a.w[(c0 + i) * 9 + t];
    const float bias = a.bias[0];
#pragma unroll 1
    for (int strip = s_first; strip < s_last; ++strip) {
        HEAD_STRIP();
        const T* img = (const T*)a.x + (long)b * a.H * a.W * a.ldx + c0;
        float* out = a.out + (long)b * a.H * a.W;
        Ch8<T> r[5][3];                                             // rows y-1, y, y+1 and two rows in flight (roles rotate; indices are compile-time)
        bool rok[5];                                                // the row is inside the image
        auto load_row = [&](int y, Ch8<T> (&d)[3], bool& ok) {
            ok = (unsigned)y < (unsigned)a.H;
            const long row = (long)min(max(y, 0), a.H - 1) * a.W;
#pragma unroll
            for (int cx = 0; cx < 3; ++cx) d[cx].load(img + (row + xc[cx]) * a.ldx);
        };
        auto dot_row = [&](const Ch8<T> (&d)[3], bool ok, int trow, float acc) {
#pragma unroll
            for (int cx = 0; cx < 3; ++cx) {
                float v[8];
                Ch8<T> t = d[cx];
                t.keep();                                           // unpack HERE (the compiler otherwise keeps the ring as unpacked floats: 2x the registers)
                t.get(v);
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) part = fmaf(v[i], wr[trow * 3 + cx][i], part);
                acc += (ok && xok[cx]) ? part : 0.f;
            }
            return acc;
        };
        load_row(y0 - 1, r[0], rok[0]); load_row(y0, r[1], rok[1]); load_row(y0 + 1, r[2], rok[2]); load_row(y0 + 2, r[3], rok[3]);
#define HEAD_FWD_STEP(K, R0, R1, R2, R4)                                                          \
    {                                                                                             \
        const int yy = y + K;                                                                     \
        load_row(yy + 3, r[R4], rok[R4]);                                                         \
        float acc = dot_row(r[R0], rok[R0], 0, 0.f);                                              \
        acc = dot_row(r[R1], rok[R1], 1, acc);                                                    \
        acc = dot_row(r[R2], rok[R2], 2, acc);                                                    \
        for (int o = cb >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);                      \
        if (xin && j == 0 && yy < y1) out[(long)yy * a.W + x] = a.inv_min_depth / (1.f + __expf(-(acc + bias))); \
    }
        for (int y = y0; y < y1; y += 5) {
            HEAD_FWD_STEP(0, 0, 1, 2, 4) HEAD_FWD_STEP(1, 1, 2, 3, 0) HEAD_FWD_STEP(2, 2, 3, 4, 1) HEAD_FWD_STEP(3, 3, 4, 0, 2) HEAD_FWD_STEP(4, 4, 0, 1, 3)
        }
#undef HEAD_FWD_STEP
    }
}


// ---- InvDepth head forward on the matrix cores (round 4, bf16 activations, C % 32 == 0) -----------------------------------------------------------
// The row-marching kernel above spends C * 9 fp32 multiply-adds per pixel on the VALU (2.3 TB/s on the full-resolution head).  As a GEMM the head is 9 rows
// (taps) x C x pixels: too few rows for a tile, but per INPUT row r and tap column kx
//     D[ky][x] += sum_c w[c][ky][kx] * in[r][x + kx - 1][c]          (one v_mfma_f32_16x16x32_bf16 per 32 channels: 3 of its 16 rows used)
// is the contribution of input row r to output row r + 1 - ky at column x: three accumulating MFMAs (kx = 0, 1, 2) whose pixel operand is the SAME 16-byte
// chunk of global memory at a shifted pixel -- a 1x1-like operand: lane (pixel, 8-channel block) loads it straight from global memory, no LDS.  A wave owns a
// 16-pixel column group and marches down its rows: lanes 0..15 hold D[0..2] of their pixel, out[y] = D_{y-1}[2] + D_y[1] + D_{y+1}[0] slides through three
// registers.  Weights as bf16 hi + lo pairs (two MFMAs per step): the head's weights stay fp32-accurate as in the VALU kernel.
template <int CS, bool PF>                                          // CS = C / 32; PF: an input row's chunks are loaded one row ahead of their MFMAs
__global__ __launch_bounds__(256) void invdepth_fwd_mfma_kernel(HeadArgs a, int rows_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane & 15, kb = lane >> 4;                      // operand column / row index, 8-channel block of the 32-channel K-step
    const int gx = (a.W + 63) >> 6, gy = (a.H + rows_per_wg - 1) / rows_per_wg;
    int id = blockIdx.x;
    const int tx = id % gx; id /= gx;
    const int ty = id % gy; const int b = id / gy;
    const int x = tx * 64 + wave * 16 + px;                        // this lane's pixel column (as the pixel operand's column and as the result's column)
    const int y0 = ty * rows_per_wg, y1 = min(a.H, y0 + rows_per_wg);
    // weight operand: lane (row i = px -> ky = i (i < 3), K block kb): w[c][ky][kx] for c = 32 s + 8 kb + j, as bf16 hi / lo.  The workgroup converts the
    // 9 C weights once into fragment order in LDS (as 24 CS strided loads + conversions per LANE the prologue cost more than the rows of a short workgroup)
    __shared__ __attribute__((aligned(16))) bf16_t s_w[3 * CS * 2 * 64 * 8];
    for (int e = threadIdx.x; e < 9 * 32 * CS; e += 256) {
        const int c = e / 9, tap = e - c * 9, ky = tap / 3, kx = tap - ky * 3;
        const float f = a.w[e];
        const bf16_t hi = f2bf(f);
        const bf16_t lo = f2bf(f - bf2f(hi));
        const int sl = c >> 5, kbl = (c >> 3) & 3, j = c & 7;
        const int at = (((kx * CS + sl) * 2) * 64 + kbl * 16 + ky) * 8 + j;
        s_w[at] = hi; s_w[at + 64 * 8] = lo;
    }
    __syncthreads();
    if (tx * 64 + wave * 16 >= a.W) return;
    u32x4_t wh[3][CS], wl[3][CS];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int s = 0; s < CS; ++s) {
            const u32x4_t vh = *(const u32x4_t*)(s_w + (((kx * CS + s) * 2) * 64 + lane) * 8);
            const u32x4_t vl = *(const u32x4_t*)(s_w + (((kx * CS + s) * 2 + 1) * 64 + lane) * 8);
            wh[kx][s] = px < 3 ? vh : u32x4_t{0u, 0u, 0u, 0u};
            wl[kx][s] = px < 3 ? vl : u32x4_t{0u, 0u, 0u, 0u};
        }
    const float bias = a.bias[0];
    const bf16_t* img = (const bf16_t*)a.x + (long)b * a.H * a.W * a.ldx + 8 * kb;
    float* out = a.out + (long)b * a.H * a.W;
    float o_prev = 0.f, o_cur = 0.f;                               // partial sums of output rows r - 1 and r while input row r is being added
    // the three shifted chunks of an input row, loaded one row AHEAD of their MFMAs (clamped addresses, values outside the image selected to zero)
    const int xs[3] = {x - 1, x, x + 1};
    const bool xok[3] = {(unsigned)(x - 1) < (unsigned)a.W, (unsigned)x < (unsigned)a.W, (unsigned)(x + 1) < (unsigned)a.W};
    u32x4_t nxt[3][CS];
    auto load_row = [&](int r) {
        const bool rok = (unsigned)r < (unsigned)a.H;
        const long row = (long)(rok ? r : 0) * a.W;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const bf16_t* q = img + (row + (xok[kx] ? xs[kx] : 0)) * a.ldx;
#pragma unroll
            for (int s = 0; s < CS; ++s) {
                const u32x4_t v = *(const u32x4_t*)(q + 32 * s);
                nxt[kx][s] = (rok && xok[kx]) ? v : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    };
    if constexpr (PF) load_row(y0 - 1);
    for (int r = y0 - 1; r <= y1; ++r) {
        if constexpr (!PF) load_row(r);
        u32x4_t in[3][CS];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int s = 0; s < CS; ++s) in[kx][s] = nxt[kx][s];
        if constexpr (PF) { if (r < y1) load_row(r + 1); }
        f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int s = 0; s < CS; ++s) {
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wh[kx][s]), __builtin_bit_cast(bf16x8_t, in[kx][s]), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wl[kx][s]), __builtin_bit_cast(bf16x8_t, in[kx][s]), d, 0, 0, 0);
            }
        // lanes 0..15 (kb = 0): d[ky] = contribution of input row r to output row r + 1 - ky at their pixel
        const float done = o_prev + d[2];                          // output row r - 1 is complete
        if (kb == 0 && r - 1 >= y0 && r - 1 < y1 && x < a.W) out[(long)(r - 1) * a.W + x] = a.inv_min_depth / (1.f + __expf(-(done + bias)));
        o_prev = o_cur + d[1];
        o_cur = d[0];
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
//            accumulators per thread, one record of sums per workgroup
// Split because together they need 72 weights + 72 accumulators per thread (256 VGPRs, 8 waves/CU, latency-bound);
// apart each keeps < 160 VGPRs, and the weight half can run on the weight-gradient side stream.
// Both march down rows with the 3 x 3 window of dlogit in registers: tap t = (ty, tx) of pixel (y, x) meets dlogit(y + 1 - ty, x + 1 - tx).
struct DlRow { float v[3]; bool ok; };                             // raw loads + "row inside the image"; masked where it is USED (HEAD_DL)
#define HEAD_LOAD_DL(Y, D)                                                                        \
    {                                                                                             \
        (D).ok = (unsigned)(Y) < (unsigned)a.H;                                                   \
        const long row_ = (long)min(max((Y), 0), a.H - 1) * a.W;                                  \
        _Pragma("unroll") for (int cx = 0; cx < 3; ++cx) (D).v[cx] = dl[row_ + xc[cx]];           \
    }
#define HEAD_DL(D, CX) (((D).ok && xok[CX]) ? (D).v[CX] : 0.f)

template <typename T>
__global__ __launch_bounds__(256) void invdepth_bwd_data_kernel(HeadArgs a) {
    HEAD_MAP();
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[t][i] = a.w[(c0 + i) * 9 + t];
#pragma unroll 1
    for (int strip = s_first; strip < s_last; ++strip) {
        HEAD_STRIP();
        const float* dl = a.dlogit + (long)b * a.H * a.W;
        T* dx = (T*)a.dx + (long)b * a.H * a.W * a.lddx + c0;
        DlRow d[4];
        HEAD_LOAD_DL(y0 - 1, d[0]) HEAD_LOAD_DL(y0, d[1]) HEAD_LOAD_DL(y0 + 1, d[2])
#define HEAD_BD_STEP(K, R0, R1, R2, R3)                                                           \
    {                                                                                             \
        const int yy = y + K;                                                                     \
        HEAD_LOAD_DL(yy + 2, d[R3])                                                               \
        float dxv[8];                                                                             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) dxv[i] = 0.f;                               \
        _Pragma("unroll") for (int ty = 0; ty < 3; ++ty)                                          \
            _Pragma("unroll") for (int tx = 0; tx < 3; ++tx) {                                    \
                const float g = ty == 0 ? HEAD_DL(d[R2], 2 - tx) : (ty == 1 ? HEAD_DL(d[R1], 2 - tx) : HEAD_DL(d[R0], 2 - tx)); \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) dxv[i] = fmaf(g, wr[ty * 3 + tx][i], dxv[i]); \
            }                                                                                     \
        if (xin && yy < y1) st8<T>(dx + ((long)yy * a.W + x) * a.lddx, dxv);                      \
    }
        for (int y = y0; y < y1; y += 4) {
            HEAD_BD_STEP(0, 0, 1, 2, 3) HEAD_BD_STEP(1, 1, 2, 3, 0) HEAD_BD_STEP(2, 2, 3, 0, 1) HEAD_BD_STEP(3, 3, 0, 1, 2)
        }
#undef HEAD_BD_STEP
    }
}

template <typename T>
__global__ __launch_bounds__(256) void invdepth_bwd_weight_kernel(HeadArgs a) {
    extern __shared__ float sdw[];                               // [C*9 + 1] block-level gradient accumulators
    for (int i = threadIdx.x; i < a.C * 9 + 1; i += 256) sdw[i] = 0.f;
    __syncthreads();
    HEAD_MAP();
    float gw[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) gw[t][i] = 0.f;
    float gb = 0.f;
#pragma unroll 1
    for (int strip = s_first; strip < s_last; ++strip) {
        HEAD_STRIP();
        const float* dl = a.dlogit + (long)b * a.H * a.W;
        const T* img = (const T*)a.x + (long)b * a.H * a.W * a.ldx + c0;
        DlRow d[5];
        Ch8<T> xr[4];                                                // this row's activations and three rows in flight: one 16-byte load per
        bool xrok[4];                                                // thread and row is all this stream reads, so its depth is the memory parallelism
#define HEAD_LOAD_X(Y, S) { xrok[S] = xin && (Y) < y1; xr[S].load(img + ((long)min((Y), a.H - 1) * a.W + xc[1]) * a.ldx); }
        HEAD_LOAD_DL(y0 - 1, d[0]) HEAD_LOAD_DL(y0, d[1]) HEAD_LOAD_DL(y0 + 1, d[2]) HEAD_LOAD_DL(y0 + 2, d[3])
        HEAD_LOAD_X(y0, 0) HEAD_LOAD_X(y0 + 1, 1) HEAD_LOAD_X(y0 + 2, 2)
#define HEAD_BW_STEP(K, R0, R1, R2, R4, X0, X3)                                                   \
    {                                                                                             \
        const int yy = y + K;                                                                     \
        HEAD_LOAD_DL(yy + 3, d[R4])                                                               \
        HEAD_LOAD_X(yy + 3, X3)                                                                   \
        float v[8];                                                                               \
        { Ch8<T> t = xr[X0]; t.keep(); t.get(v); }                                                \
        const bool live = xrok[X0];                                                               \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) v[i] = live ? v[i] : 0.f;                   \
        if (j == 0 && live) gb += HEAD_DL(d[R1], 1);                                              \
        _Pragma("unroll") for (int ty = 0; ty < 3; ++ty)                                          \
            _Pragma("unroll") for (int tx = 0; tx < 3; ++tx) {                                    \
                const float g = ty == 0 ? HEAD_DL(d[R2], 2 - tx) : (ty == 1 ? HEAD_DL(d[R1], 2 - tx) : HEAD_DL(d[R0], 2 - tx)); \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) gw[ty * 3 + tx][i] = fmaf(g, v[i], gw[ty * 3 + tx][i]); \
            }                                                                                     \
    }
        // ring roles after k steps: dlogit rows (k, k+1, k+2 | in flight k+3 | free k+4) mod 5, activations (k | .. | free k+3) mod 4: period 20
        for (int y = y0; y < y1; y += 20) {
            HEAD_BW_STEP(0, 0, 1, 2, 4, 0, 3) HEAD_BW_STEP(1, 1, 2, 3, 0, 1, 0) HEAD_BW_STEP(2, 2, 3, 4, 1, 2, 1) HEAD_BW_STEP(3, 3, 4, 0, 2, 3, 2)
            HEAD_BW_STEP(4, 4, 0, 1, 3, 0, 3) HEAD_BW_STEP(5, 0, 1, 2, 4, 1, 0) HEAD_BW_STEP(6, 1, 2, 3, 0, 2, 1) HEAD_BW_STEP(7, 2, 3, 4, 1, 3, 2)
            HEAD_BW_STEP(8, 3, 4, 0, 2, 0, 3) HEAD_BW_STEP(9, 4, 0, 1, 3, 1, 0) HEAD_BW_STEP(10, 0, 1, 2, 4, 2, 1) HEAD_BW_STEP(11, 1, 2, 3, 0, 3, 2)
            HEAD_BW_STEP(12, 2, 3, 4, 1, 0, 3) HEAD_BW_STEP(13, 3, 4, 0, 2, 1, 0) HEAD_BW_STEP(14, 4, 0, 1, 3, 2, 1) HEAD_BW_STEP(15, 0, 1, 2, 4, 3, 2)
            HEAD_BW_STEP(16, 1, 2, 3, 0, 0, 3) HEAD_BW_STEP(17, 2, 3, 4, 1, 1, 0) HEAD_BW_STEP(18, 3, 4, 0, 2, 2, 1) HEAD_BW_STEP(19, 4, 0, 1, 3, 3, 2)
        }
#undef HEAD_BW_STEP
#undef HEAD_LOAD_X
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&sdw[(c0 + i) * 9 + t], gw[t][i]);
    if (j == 0) atomicAdd(&sdw[a.C * 9], gb);
    __syncthreads();
    // one record per workgroup (plain stores; invdepth_reduce_kernel adds the records up): C*9+1 same-address global atomics per
    // workgroup cost 13-33 us per launch at ~0.09 TB/s of contended adds -- as much as the stream itself on the 256-channel head
    float* rec = a.dw + (long)blockIdx.x * a.rec_stride;
    for (int i = threadIdx.x; i < a.C * 9 + 1; i += 256) rec[i] = sdw[i];
}

// out[i] = sum_p rec[p][i]: block = 64 outputs x 16 record lanes (eight loads in flight per thread), records added in a fixed order
__global__ __launch_bounds__(1024) void invdepth_reduce_kernel(const float* __restrict__ rec, long stride, int parts, int n, float* __restrict__ out) {
    __shared__ float s[16][64];
    const int lo = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + lo;
    float acc = 0.f;
    if (o < n) {
        int p = q;
        for (; p + 7 * 16 < parts; p += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = rec[(long)(p + k * 16) * stride + o];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; p < parts; p += 16) acc += rec[(long)p * stride + o];
    }
    s[q][lo] = acc;
    __syncthreads();
    if (q == 0 && o < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[k][lo];
        out[o] = t;
    }
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

// ---- the inverse-depth channel of the decoder's iconv inputs as a rank-1 term (round 4) -------------------------------------------------------
// Reference: iconv3 / iconv2 / iconv1 see torch.cat((unpack, skip, nearest_up2(inv_depth)), 1) (PackNetSAN01.py:118-143): 193 / 97 / 65 input
// channels.  The single extra channel costs the GEMM kernels a whole 32-channel slice (65 -> 72 -> three slices instead of two) and keeps the layers off
// the kernels that want Cin % 32 = 0.  Convolution is linear in its input channels, so  conv(cat(x, u)) = conv_{C-1}(x) + conv_1(u):  the second term
// is a 3x3 one-channel stencil of the LOW-resolution map (u = nearest_up2(inv)), written first; the GEMM then accumulates onto it.
//   forward      r[p][n] = sum_tap w[n][tap] u[p + tap - 1]                              (zero outside the full-resolution image)
//   data grad    dinv[q'] = sum over the 2x2 pixels q of q':  sum_n sum_tap w[n][tap] dy[q - tap + 1][n]  = sum over the 4x4 window of dy around
//                q' with the 16 COMBINED weight vectors Wc[window position][n] (36 -> 16 multiply-adds per channel)
//   weight grad  dw[n][tap] = sum_p dy[p][n] u[p + tap - 1]: mte_invdepth_bwd_weight(x = dy, dlogit = u) with the taps flipped by the caller.
// w points at channel C-1 of an OIHW tensor: element (n, tap) at w[n * w_stride + tap].
template <typename T>
__global__ __launch_bounds__(256) void rank1_conv_fwd_kernel(const float* __restrict__ inv, const float* __restrict__ w, long wstride, T* __restrict__ y, long ldy,
                                                             int B, int h, int wl, int N) {
    // thread = (image row Y of one sample, 16-byte chunk of channels), marching along X: its 9 x P weights stay in registers, the 3 x 3 window of the
    // up-sampled map slides (one new column = three loads of the small low-resolution map per pixel).  Branch-free: rows / columns outside the image are
    // clamped addresses with the value selected to zero.
    constexpr int P = Elem<T>::PER16, SEG = 64;                    // a thread marches over SEG pixels of its row
    const int cpr = N / P, H = 2 * h, W = 2 * wl, nseg = (W + SEG - 1) / SEG;
    const long items = (long)B * H * nseg * cpr;
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(it % cpr); long t = it / cpr;
        const int X0 = (int)(t % nseg) * SEG; t /= nseg;
        const int Y = (int)(t % H); const int b = (int)(t / H);
        float wr[9][P];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int e = 0; e < P; ++e) wr[tap][e] = w[(long)(cc * P + e) * wstride + tap];
        const float* r0 = inv + ((long)b * h + (max(Y - 1, 0) >> 1)) * wl;
        const float* r1 = inv + ((long)b * h + (Y >> 1)) * wl;
        const float* r2 = inv + ((long)b * h + (min(Y + 1, H - 1) >> 1)) * wl;
        const bool ok0 = Y > 0, ok2 = Y + 1 < H;
        float c0[3], c1[3];                                        // window columns X - 1, X (X + 1 is loaded per pixel)
        {
            const int xm = max(X0 - 1, 0) >> 1, xc = X0 >> 1;
            const bool okm = X0 > 0;
            c0[0] = (ok0 && okm) ? r0[xm] : 0.f; c0[1] = okm ? r1[xm] : 0.f; c0[2] = (ok2 && okm) ? r2[xm] : 0.f;
            c1[0] = ok0 ? r0[xc] : 0.f; c1[1] = r1[xc]; c1[2] = ok2 ? r2[xc] : 0.f;
        }
        T* yr = y + (((long)b * H + Y) * W) * ldy + cc * P;
        const int X1 = min(X0 + SEG, W);
        for (int X = X0; X < X1; ++X) {
            const int xn = min(X + 1, W - 1) >> 1;
            const bool okn = X + 1 < W;
            float c2[3] = {(ok0 && okn) ? r0[xn] : 0.f, okn ? r1[xn] : 0.f, (ok2 && okn) ? r2[xn] : 0.f};
            float v[P];
#pragma unroll
            for (int e = 0; e < P; ++e) {
                float a = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) a = fmaf(c0[ky], wr[ky * 3][e], fmaf(c1[ky], wr[ky * 3 + 1][e], fmaf(c2[ky], wr[ky * 3 + 2][e], a)));
                v[e] = a;
            }
            *(u32x4_t*)(yr + (long)X * ldy) = pack16<T>(v);
#pragma unroll
            for (int k = 0; k < 3; ++k) { c0[k] = c1[k]; c1[k] = c2[k]; }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rank1_conv_bwd_data_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ w, long wstride, float* __restrict__ dinv,
                                                                  int B, int h, int wl, int N, int accumulate) {
    // thread = (low-resolution row, segment of 16 low-resolution pixels, channel chunk), marching along x: the 4 x 4 window of dy records around the item's
    // 2 x 2 block slides in registers (packed; two new columns = 8 loads per item instead of 16); clamped addresses, values outside the image selected to zero
    extern __shared__ __attribute__((aligned(16))) float sw[];     // [9][N] taps, then [16][N] combined window weights
    float* wc = sw + 9 * N;
    for (int i = threadIdx.x; i < 9 * N; i += blockDim.x) sw[i] = w[(long)(i % N) * wstride + i / N];
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * N; i += blockDim.x) {
        const int c = i % N, r = i / N, ry = r >> 2, rx = r & 3;
        float s = 0.f;
        for (int qy = 0; qy < 2; ++qy)
            for (int qx = 0; qx < 2; ++qx) {
                const int ky = qy + 2 - ry, kx = qx + 2 - rx;       // window pixel (ry, rx) reaches block pixel (qy, qx) through tap (ky, kx)
                if ((unsigned)ky < 3u && (unsigned)kx < 3u) s += sw[(ky * 3 + kx) * N + c];
            }
        wc[i] = s;
    }
    __syncthreads();
    constexpr int P = Elem<T>::PER16, SEG = 16;
    const int cpr = N / P, H = 2 * h, W = 2 * wl, nseg = (wl + SEG - 1) / SEG;   // cpr: a power of two <= 32 (launcher), a pixel's chunks sit in consecutive lanes
    const long items = (long)B * h * nseg * cpr;
    const int cc = threadIdx.x % cpr;
    const long step = (long)gridDim.x * blockDim.x;
    for (long i0 = blockIdx.x * (long)blockDim.x; i0 < items; i0 += step) {
        const long it = min(i0 + (long)threadIdx.x, items - 1);   // (tail lanes redo the last item and do not store: the chunk lanes of a pixel reduce together)
        const bool live = i0 + threadIdx.x < items;
        long t = it / cpr;
        const int x0 = (int)(t % nseg) * SEG; t /= nseg;
        const int yl = (int)(t % h); const int b = (int)(t / h);
        const T* base = dy + (long)b * H * W * lddy + cc * P;
        long rowoff[4]; bool Yok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int Y = 2 * yl - 1 + r; Yok[r] = (unsigned)Y < (unsigned)H; rowoff[r] = (long)min(max(Y, 0), H - 1) * W; }
        u32x4_t win[4][4];
        auto load_col = [&](int X, int k) __attribute__((always_inline)) {
            const int Xc = min(max(X, 0), W - 1);
            const bool ok = (unsigned)X < (unsigned)W;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const u32x4_t v = *(const u32x4_t*)(base + (rowoff[r] + Xc) * lddy);
                const bool kk = ok && Yok[r];
                win[r][k] = u32x4_t{kk ? v[0] : 0u, kk ? v[1] : 0u, kk ? v[2] : 0u, kk ? v[3] : 0u};
            }
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) load_col(2 * x0 - 1 + k, k);
        const int x1 = min(x0 + SEG, wl);
        for (int x = x0; x < x1; ++x) {
            float acc = 0.f;
            const int woff = cc * P;                                // (the compiler hoists these 128 combined weights into registers: 253 VGPRs, one wave per SIMD -- and faster than re-reading them: 106 vs 129 us)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float g[P];
                    unpack16<T>(win[r][k], g);
                    const float* wr = wc + (r * 4 + k) * N + woff;
#pragma unroll
                    for (int e = 0; e < P; ++e) acc = fmaf(g[e], wr[e], acc);
                }
            for (int off = 1; off < cpr; off <<= 1) acc += __shfl_xor(acc, off, 64);
            const long pq = ((long)b * h + yl) * wl + x;
            if (live && cc == 0) dinv[pq] = accumulate ? dinv[pq] + acc : acc;
            if (x + 1 < x1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { win[r][0] = win[r][2]; win[r][1] = win[r][3]; }
                load_col(2 * x + 3, 2);
                load_col(2 * x + 4, 3);
            }
        }
    }
}

// weight gradient of the rank-1 term: dw[n][tap] = sum_p dy[p][n] u[p + tap - 1], u = nearest_up2(inv) (0 outside the image).  Thread = (low-resolution row,
// segment of 16 low-resolution pixels, channel chunk): per item the four dy records of its 2 x 2 block (every record of dy is read exactly once) against the
// 3 x 3 window of the low-resolution map sliding in registers; 9 x P register accumulators.  A workgroup leaves one record of N x 9 partial sums (fixed-order
// shuffles + LDS tree); rank1_wgrad_reduce_kernel adds the records in a fixed order: bit-reproducible, no floating-point atomics.
template <typename T>
__global__ __launch_bounds__(256) void rank1_conv_bwd_weight_kernel(const T* __restrict__ dy, long lddy, const float* __restrict__ inv, float* __restrict__ records,
                                                                    int B, int h, int wl, int N) {
    extern __shared__ __attribute__((aligned(16))) float sred[];   // [4][N * 9] per-wave partial sums
    constexpr int P = Elem<T>::PER16, SEG = 16;
    const int cpr = N / P, W = 2 * wl, nseg = (wl + SEG - 1) / SEG;
    const long items = (long)B * h * nseg * cpr;
    float dacc[9][P];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < P; ++e) dacc[tp][e] = 0.f;
    const int cc = threadIdx.x % cpr;                              // (256 and the grid stride are multiples of cpr: a thread keeps its chunk)
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        long t = it / cpr;
        const int x0 = (int)(t % nseg) * SEG; t /= nseg;
        const int yl = (int)(t % h); const int b = (int)(t / h);
        const T* base = dy + (((long)b * 2 * h + 2 * yl) * W) * lddy + cc * P;
        const float* ir[3]; bool iok[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int yy = yl - 1 + r; iok[r] = (unsigned)yy < (unsigned)h; ir[r] = inv + ((long)b * h + min(max(yy, 0), h - 1)) * wl; }
        float uw[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int xx = x0 - 1 + k; const bool ok = (unsigned)xx < (unsigned)wl; const int xc = min(max(xx, 0), wl - 1);
#pragma unroll
            for (int r = 0; r < 3; ++r) uw[r][k] = (ok && iok[r]) ? ir[r][xc] : 0.f;
        }
        const int x1 = min(x0 + SEG, wl);
        for (int x = x0; x < x1; ++x) {
            u32x4_t raw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) raw[q] = *(const u32x4_t*)(base + ((long)(q >> 1) * W + 2 * x + (q & 1)) * lddy);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float g[P];
                unpack16<T>(raw[q], g);
                const int qy = q >> 1, qx = q & 1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        // u at full-resolution (2 yl + qy + ky - 1, 2 x + qx + kx - 1) = map (yl + ((qy + ky - 1) >> 1), x + ((qx + kx - 1) >> 1))
                        const float uu = uw[(qy + ky + 1) >> 1][(qx + kx + 1) >> 1];
#pragma unroll
                        for (int e = 0; e < P; ++e) dacc[ky * 3 + kx][e] = fmaf(g[e], uu, dacc[ky * 3 + kx][e]);
                    }
            }
            if (x + 1 < x1) {
                const int xx = x + 2; const bool ok = xx < wl; const int xc = min(xx, wl - 1);
#pragma unroll
                for (int r = 0; r < 3; ++r) { uw[r][0] = uw[r][1]; uw[r][1] = uw[r][2]; uw[r][2] = (ok && iok[r]) ? ir[r][xc] : 0.f; }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < P; ++e) {
            float v = dacc[tp][e];
            for (int off = cpr; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if (lane < cpr) sred[wave * (N * 9) + (lane * P + e) * 9 + tp] = v;       // lane = chunk (lane % cpr = cc)
        }
    __syncthreads();
    for (int i = threadIdx.x; i < N * 9; i += blockDim.x)
        records[(long)blockIdx.x * (N * 9) + i] = (sred[i] + sred[N * 9 + i]) + (sred[2 * N * 9 + i] + sred[3 * N * 9 + i]);
}
// dw[n * dst_stride + tap] = sum of the workgroup records: one wave per output, lane l adds records l, l + 64, ... in order, then a fixed shuffle tree
__global__ __launch_bounds__(256) void rank1_wgrad_reduce_kernel(const float* __restrict__ records, int nrec, int N, float* __restrict__ dst, long dst_stride) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N * 9) return;
    float s = 0.f;
    for (int r = lane; r < nrec; r += 64) s += records[(long)r * (N * 9) + i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) dst[(long)(i / 9) * dst_stride + i % 9] = s;
}

__global__ void upsample2_f32_kernel(const float* __restrict__ inv, float* __restrict__ out, int B, int h, int w) {
    const long n = (long)B * 4 * h * w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int X = (int)(i % (2 * w)); const long t = i / (2 * w); const int Y = (int)(t % (2 * h)); const int b = (int)(t / (2 * h));
        out[i] = inv[((long)b * h + (Y >> 1)) * w + (X >> 1)];
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

// a [dW; db] record delivered to its two places in the flat gradient buffer by ONE launch (two hipMemcpyAsync = two blit kernels of ~4 us each and two runtime calls)
__global__ void split_record_kernel(const float* __restrict__ src, float* __restrict__ d0, int n0, float* __restrict__ d1, int n1) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += gridDim.x * blockDim.x) {
        const float v = src[i];
        if (i < n0) d0[i] = v; else d1[i - n0] = v;
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
int g_head_mfma = 1;                                 // development knob (mte_debug_set(30, v)): 0 = the VALU row-marching head forward
inline bool head_ok(int C) { const int cb = C >> 3; return C % 8 == 0 && cb >= 1 && cb <= 64 && (cb & (cb - 1)) == 0; }
inline long head_strips(int B, int H, int W, int C) {               // strips of the row-marching head kernels (see HEAD_MAP)
    const int pxw = 256 / (C >> 3);
    return (long)((W + pxw - 1) / pxw) * ((H + HEAD_ROWS - 1) / HEAD_ROWS) * B;
}
// workgroups the chip holds at once for kernel `f` (256 threads, `lds` dynamic bytes); queried once per kernel
// (the dynamic LDS size follows the head's channel count, so the answer is cached per (kernel, C); relaxed atomics: two threads that race
// compute the same value)
template <typename F> int head_resident(F f, size_t lds, std::atomic<int>* cache) {
    int v = cache->load(std::memory_order_relaxed);
    if (v == 0) {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, 256, lds) != hipSuccess || per_cu < 1) per_cu = 2;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        v = per_cu * cus;
        cache->store(v, std::memory_order_relaxed);
    }
    return v;
}
inline long head_record_stride(int C) { return ((long)C * 9 + 1 + 3) / 4 * 4; }
constexpr int HEAD_WGRAD_MAX_BLOCKS = 1024;

}  // namespace

#ifdef MTE_DEV
extern "C" int mtei_set_head_mfma(int v) { g_head_mfma = v; return MTE_OK; }
#endif

extern "C" {

int mte_invdepth_fwd(const void* x, long ldx, const float* w, const float* bias, float* out,
                     int B, int H, int W, int C, float min_depth, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !w || !bias || !out || !head_ok(C)) return MTE_ERR_ARG;
    HeadArgs a{}; a.x = x; a.ldx = ldx; a.w = w; a.bias = bias; a.out = out; a.B = B; a.H = H; a.W = W; a.C = C;
    a.inv_min_depth = 1.f / min_depth; a.npix = (long)B * H * W;
    // one workgroup per strip: workgroups are dispatched in order, so the ones running together walk down horizontally ADJACENT strips and
    // DRAM sees whole image rows requested at once (balanced strip ranges over the resident workgroups -- kept for the weight gradient, whose
    // per-workgroup record must stay small in number -- measured 6 % slower here: 118 vs 111 us on the full-resolution head)
    // matrix-core form for 32 / 64 / 128 channels (same-box: 102 -> 54 us, 57 -> 48 us, 40 -> 28 us; with 256 channels the weight fragments alone are 384 VGPRs and the
    // VALU kernel wins 25 vs 36 us).  Workgroup = 64 pixel columns x `rows` rows (4 waves of 16 columns); rows per workgroup and the row prefetch by measurement
    // (tools/head_bench.py): 32 rows, no prefetch for one K-step per tap column; prefetch from two K-steps on; 16 rows at 128 channels (240 workgroups at 96 x 320)
    if (dtype == MTE_DT_BF16 && g_head_mfma && C % 32 == 0 && C <= ((g_head_mfma & 2) ? 256 : 128) && ldx % 8 == 0) {
        int rows = C >= 128 ? 16 : 32;
        bool pf = C >= 64;
        if (g_head_mfma != 1) {                                     // (development: bit 2 = prefetch, bits 8.. = rows per workgroup)
            pf = (g_head_mfma & 4) != 0;
            if (g_head_mfma >> 8) rows = g_head_mfma >> 8;
        }
        const long g2 = (long)((W + 63) / 64) * ((H + rows - 1) / rows) * B;
        if (g2 <= 0x7fffffffL) {
#define MTE_HEAD_LAUNCH(CS_)                                                                                                              \
            if (pf) hipLaunchKernelGGL((invdepth_fwd_mfma_kernel<CS_, true>), dim3((unsigned)g2), dim3(256), 0, stream, a, rows);       \
            else hipLaunchKernelGGL((invdepth_fwd_mfma_kernel<CS_, false>), dim3((unsigned)g2), dim3(256), 0, stream, a, rows);         \
            return mte_check_launch();
            switch (C / 32) {
                case 1: { MTE_HEAD_LAUNCH(1) }
                case 2: { MTE_HEAD_LAUNCH(2) }
                case 4: { MTE_HEAD_LAUNCH(4) }
                case 8: { MTE_HEAD_LAUNCH(8) }
            }
#undef MTE_HEAD_LAUNCH
        }
    }
    const long grid = head_strips(B, H, W, C);
    if (grid > 0x7fffffffL) return MTE_ERR_UNSUPPORTED;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(invdepth_fwd_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(invdepth_fwd_kernel<float>, dim3((unsigned)grid), dim3(256), 0, stream, a);
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
    const long grid = head_strips(B, H, W, C);
    if (grid > 0x7fffffffL) return MTE_ERR_UNSUPPORTED;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(invdepth_bwd_data_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(invdepth_bwd_data_kernel<float>, dim3((unsigned)grid), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// floats of `records` scratch mte_invdepth_bwd_weight needs (one record of C*9+1 sums per workgroup)
long mte_invdepth_bwd_weight_workspace_elems(int C) { return head_ok(C) ? HEAD_WGRAD_MAX_BLOCKS * head_record_stride(C) : 0; }

int mte_invdepth_bwd_weight(const void* x, long ldx, const float* dlogit, float* dwb, float* records,
                            int B, int H, int W, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dlogit || !dwb || !records || !head_ok(C)) return MTE_ERR_ARG;
    const long npix = (long)B * H * W;
    HeadArgs a{}; a.x = x; a.ldx = ldx; a.dlogit = dlogit; a.dw = records; a.rec_stride = head_record_stride(C);
    a.B = B; a.H = H; a.W = W; a.C = C; a.npix = npix;
    const int n = C * 9 + 1;
    if (dtype == MTE_DT_BF16) {                                  // round 5: the sums as a GEMM over pixels on the matrix cores (tap_wgrad.hip)
        int groups = 0;
        const int rc = tap_wgrad_mfma_launch((const bf16_t*)x, ldx, dlogit, records, a.rec_stride, B, H, W, C, -1, 0, 1, HEAD_WGRAD_MAX_BLOCKS, &groups, stream);
        if (rc == MTE_OK) {
            hipLaunchKernelGGL(invdepth_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, stream, records, a.rec_stride, groups, n, dwb);
            return mte_check_launch();
        }
        if (rc != MTE_ERR_UNSUPPORTED) return rc;
    }
    const size_t lds = sizeof(float) * (C * 9 + 4);
    long g = head_strips(B, H, W, C);
    static std::atomic<int> res_b[64], res_f[64];               // by C / 8 (head_ok: C is a multiple of 8, at most 504)
    const int ci = (C >> 3) & 63;
    const long res = dtype == MTE_DT_BF16 ? head_resident(invdepth_bwd_weight_kernel<bf16_t>, lds, &res_b[ci]) : head_resident(invdepth_bwd_weight_kernel<float>, lds, &res_f[ci]);
    if (g > res) g = res;
    if (g > HEAD_WGRAD_MAX_BLOCKS) g = HEAD_WGRAD_MAX_BLOCKS;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(invdepth_bwd_weight_kernel<bf16_t>, dim3((unsigned)g), dim3(256), lds, stream, a);
    else hipLaunchKernelGGL(invdepth_bwd_weight_kernel<float>, dim3((unsigned)g), dim3(256), lds, stream, a);
    hipLaunchKernelGGL(invdepth_reduce_kernel, dim3((n + 63) / 64), dim3(1024), 0, stream, records, a.rec_stride, (int)g, n, dwb);
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

// conv3x3 of the nearest-up-sampled one-channel map `inv` [B,h,w] with channel C-1 of an OIHW weight (element (n, tap) at w[n * w_stride + tap]) ->
// y [B,2h,2w,N] (overwritten; the caller's GEMM then accumulates the other channels' convolution onto it).  N: a multiple of 32, at most 256.
int mte_rank1_conv_fwd(const float* inv, const float* w, long w_stride, void* y, long ldy, int B, int h, int wl, int N, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!inv || !w || !y || N % 32 != 0 || N > 256 || B < 1 || h < 1 || wl < 1) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    const int grid = stream_grid((long)B * 2 * h * ((2 * wl + 63) / 64) * (N / per16));      // one thread per (image row, 64-pixel segment, channel chunk)
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(rank1_conv_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, inv, w, w_stride, (bf16_t*)y, ldy, B, h, wl, N);
    else hipLaunchKernelGGL(rank1_conv_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, inv, w, w_stride, (float*)y, ldy, B, h, wl, N);
    return mte_check_launch();
}
// its gradient with respect to `inv`: dinv [B,h,w] (+)= ... of dy [B,2h,2w,N].  N: 32, 64 or 128 (a pixel's 16-byte chunks must fit one wave)
int mte_rank1_conv_bwd_data(const void* dy, long lddy, const float* w, long w_stride, float* dinv, int B, int h, int wl, int N, int accumulate, int dtype,
                            hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dy || !w || !dinv || (N != 32 && N != 64 && N != 128) || B < 1 || h < 1 || wl < 1) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    const int grid = stream_grid((long)B * h * ((wl + 15) / 16) * (N / per16));     // one thread per (low-resolution row, 16-pixel segment, channel chunk)
    const size_t lds = sizeof(float) * 25 * N;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(rank1_conv_bwd_data_kernel<bf16_t>, dim3(grid), dim3(256), lds, stream, (const bf16_t*)dy, lddy, w, w_stride, dinv, B, h, wl, N, accumulate);
    else hipLaunchKernelGGL(rank1_conv_bwd_data_kernel<float>, dim3(grid), dim3(256), lds, stream, (const float*)dy, lddy, w, w_stride, dinv, B, h, wl, N, accumulate);
    return mte_check_launch();
}
// ... and with respect to the weight column: dw (element (n, tap) at dw[n * dw_stride + tap], overwritten) = sum_p dy[p][n] up2(inv)[p + tap - 1].
// records: mte_rank1_conv_bwd_records_elems(N) floats of scratch (one N x 9 record per workgroup, summed in a fixed order: bit-reproducible).
constexpr int RANK1_BWD_WGS = 1024;
long mte_rank1_conv_bwd_records_elems(int N) { return (long)RANK1_BWD_WGS * N * 9; }
int mte_rank1_conv_bwd_weight(const void* dy, long lddy, const float* inv, float* dw, long dw_stride, float* records,
                              int B, int h, int wl, int N, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dy || !inv || !dw || !records || (N != 32 && N != 64 && N != 128) || B < 1 || h < 1 || wl < 1) return MTE_ERR_ARG;
    if (dtype == MTE_DT_BF16) {                                  // round 5: tap_wgrad.hip (x = dy at full resolution, the map up-sampled on the way into LDS)
        int groups = 0;
        const int rc = tap_wgrad_mfma_launch((const bf16_t*)dy, lddy, inv, records, (long)N * 9, B, 2 * h, 2 * wl, N, +1, 1, 0, RANK1_BWD_WGS, &groups, stream);
        if (rc == MTE_OK) {
            hipLaunchKernelGGL(rank1_wgrad_reduce_kernel, dim3((N * 9 + 3) / 4), dim3(256), 0, stream, records, groups, N, dw, dw_stride);
            return mte_check_launch();
        }
        if (rc != MTE_ERR_UNSUPPORTED) return rc;
    }
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    const long items = (long)B * h * ((wl + 15) / 16) * (N / per16);
    long g = (items + 255) / 256; if (g > RANK1_BWD_WGS) g = RANK1_BWD_WGS; if (g < 1) g = 1;
    const size_t lds = sizeof(float) * 4 * N * 9;
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(rank1_conv_bwd_weight_kernel<bf16_t>, dim3((unsigned)g), dim3(256), lds, stream, (const bf16_t*)dy, lddy, inv, records, B, h, wl, N);
    else hipLaunchKernelGGL(rank1_conv_bwd_weight_kernel<float>, dim3((unsigned)g), dim3(256), lds, stream, (const float*)dy, lddy, inv, records, B, h, wl, N);
    hipLaunchKernelGGL(rank1_wgrad_reduce_kernel, dim3((N * 9 + 3) / 4), dim3(256), 0, stream, records, (int)g, N, dw, dw_stride);
    return mte_check_launch();
}
// out [B,2h,2w] = nearest_up2(inv [B,h,w]), fp32 (the dense map the rank-1 weight gradient multiplies dy with)
int mte_upsample2_f32(const float* inv, float* out, int B, int h, int wl, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!inv || !out || B < 1 || h < 1 || wl < 1) return MTE_ERR_ARG;
    hipLaunchKernelGGL(upsample2_f32_kernel, dim3(stream_grid((long)B * 4 * h * wl)), dim3(256), 0, stream, inv, out, B, h, wl);
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

// dst0[0..n0) = src[0..n0), dst1[0..n1) = src[n0..n0+n1): a layer's (weight gradient, bias gradient) record into the two views of a flat gradient buffer
int mte_split_record(const float* src, float* dst0, int n0, float* dst1, int n1, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!src || !dst0 || !dst1 || n0 < 0 || n1 < 0) return MTE_ERR_ARG;
    if (n0 + n1 == 0) return MTE_OK;
    const int grid = (n0 + n1 + 255) / 256 < 64 ? (n0 + n1 + 255) / 256 : 64;
    hipLaunchKernelGGL(split_record_kernel, dim3(grid), dim3(256), 0, stream, src, dst0, n0, dst1, n1);
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
