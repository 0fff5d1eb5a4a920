// Weight gradient of the 3x3 convolutions with >= 64 input and >= 128 output channels (round 5) -- ALL NINE TAPS from one staged patch.
//
//   dW[n][tap][c] = sum over pixels p of dy[p][n] * x[p + tap][c]          (nn.Conv2d + ConstantPad2d, layers01.py:29-31,61: their autograd
//                                                                            weight gradient; 3x3, stride 1, zero pad 1)
//
// Behind mte_conv2d_wgrad for the 24x80 / 48x160 / 96x320 layers (the 128..512-channel residual blocks, iconv4 / iconv5, unpack4, pack4): the
// generic kernel (conv_wgrad_dma_kernel, conv_igemm.hip) gives every TAP its own workgroups, so dy and x are re-streamed nine times from L2
// (measured 5.4 GB of traffic against 2.05 GB algorithmic, 620-730 TFLOP/s), and its 64 x 64 wave tiles read two LDS fragments per MFMA.
//
// This kernel: a workgroup owns  128 output channels x 64 input channels x 9 taps  (M = 128, N = 576) and walks a range of K-STEPS of 32
// pixels.  Per K-step it stages, by LDS-DMA (buffer_load ... lds, 1 KiB pieces),
//     dy : the 32 pixels x 128 channels                                      8 KiB
//     x  : the (RK + 2) x (CK + 8)-pixel patch around them x 64 channels   12-15 KiB     (RK x CK = 32 pixels: 1 x 32 where W % 32 == 0,
//                                                                                          2 x 16 for W % 16 == 0 -- 80-pixel rows)
// and all nine taps read their x fragments from that ONE patch at shifted pixel rows: a tap is an address, not a fetch.  Out-of-image
// pixels (the zero padding, the patch's alignment columns) are out-of-range buffer offsets -> the DMA writes zeros.
//   * 8 waves = 2 (64 output channels) x 4 (16 input channels); a wave holds 4 x 9 accumulators of 16 x 16 (144 registers) and per K-step
//     reads 4 dy + 9 x fragments (13 KiB) for 36 MFMAs (v_mfma_f32_16x16x32_bf16): 0.36 fragments per MFMA (the generic kernel: 2), 90 B/clk
//     of LDS reads per CU -- the 8-phase forward kernel's ratio.
//   * operands transposed on the way out of LDS (ds_read_b64_tr_b16: the reduction runs over PIXELS, memory is pixel-major): inline asm with
//     hand-counted lgkmcnt, as in conv_wgrad_dma_kernel (the compiler would fence the intrinsic against the in-flight DMA).
//   * 3-slot ring, one s_barrier per K-step, counted vmcnt (the next K-step's three pieces per wave stay in flight across the barrier).
//   * LDS swizzles (on the DMA source chunk, the same XOR on the read): dy rows of 256 B: chunk ^= 2 * ((k & 3) | ((k >> 3) & 1) << 2);
//     x pixels of 128 B: chunk ^= (((px >> 1) & 1) << 1) | (((px >> 3) & 1) << 2) -- the eight 32-byte blocks a 32-lane half of a transposing
//     read touches (pixels s..s+3 and s+8..s+11, any shift s) fall on 64 distinct banks.
//   * operands swapped (D = X^T dY): a lane ends with 4 consecutive INPUT channels of one output channel = one 16-byte store into the
//     [N][9][Cin_p] staging slab of its pixel split (plain stores; mte_unpack_conv_wgrad adds the slabs in order: no atomics, bit-reproducible).
#include "common.hpp"
#include <type_traits>

struct Wgrad9Args {
    const bf16_t* x; long ldx;
    const bf16_t* dy; long ldy;
    float* dw; long part_stride;        // slab of pixel split s at dw + s * part_stride, layout [N][9][Cin_p]
    int B, H, W, Cin_p, N;
    int tiles_c;                        // input-channel tiles of 64 (output-channel tiles of 128: gridDim / tiles_c / splits)
    int base;                           // tiles_n * tiles_c
    int units, units_per_split;         // K-steps of 32 pixels: all, per pixel split
};

namespace {

constexpr unsigned OOB9 = 0xfffffff0u;

__device__ __forceinline__ unsigned long long tr16(unsigned addr, int off_unused = 0) {
    (void)off_unused;
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ unsigned long long tr16o(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8_t frag(unsigned long long lo, unsigned long long hi) {
    const u32x4_t c{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    return __builtin_bit_cast(bf16x8_t, c);
}
__device__ __forceinline__ int swz_y(int k) { return 2 * ((k & 3) | (((k >> 3) & 1) << 2)); }          // 16-byte chunk XOR of dy row k (0..31)
__device__ __forceinline__ int swz_x(int px) { return (((px >> 1) & 1) << 1) | (((px >> 3) & 1) << 2); } // ... of patch pixel px

// RK = 1: a K-step is 32 consecutive pixels of one image row; RK = 2: 16 columns of two consecutive rows (k < 16: the upper row)
template <int RK>
__global__ __launch_bounds__(512, 2) void conv_wgrad9_kernel(Wgrad9Args a) {
    constexpr int CK = 32 / RK;                       // columns of a K-step
    constexpr int PW = CK + 8, PR = RK + 2;           // patch: PR rows x PW pixels (columns x0 - 4 .. x0 + CK + 3: whole 8-pixel pieces)
    constexpr int PB = PW / 8, NX = PR * PB;          // 1-KiB pieces per patch row, per patch
    constexpr int SLOT = 24 * 1024;                   // 8 dy pieces + NX x pieces + (16 - NX) pieces of padding: 3 pieces per wave and K-step
    constexpr int XOFF = 8 * 1024;
    constexpr int RING = 3;
    static_assert(NX <= 16 && NX >= 8, "three pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv >> 2, wn = wv & 3;              // 64 output channels x 16 input channels of the 128 x 64 tile
    // block -> (pixel split, tile): the tiles of one split are neighbours (one XCD streams one pixel range for all of them)
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int split = id / a.base, tile = id - split * a.base;
    const int tile_n = tile / a.tiles_c, tile_c = tile - tile_n * a.tiles_c;
    const int n0 = tile_n * 128, c0 = tile_c * 64;
    const int u0 = split * a.units_per_split;
    const int u1 = min(a.units, u0 + a.units_per_split);
    const int nst = u1 - u0;
    const int CB = a.W / CK, RB = a.H / RK;           // K-steps per image row (pair), row (pairs) per image

    // ---- DMA lane constants.  piece 0: dy pixels 4 wv .. 4 wv + 3 (16 chunks each); pieces 1, 2: patch pieces wv and wv + 8 (8 pixels x 8 chunks)
    unsigned voffY;
    {
        const int k = 4 * wv + (lane >> 4), c = (lane & 15) ^ swz_y(k);
        const int dr = RK == 1 ? 0 : (k >> 4), dc = RK == 1 ? k : (k & 15);
        voffY = (unsigned)((((long)dr * a.W + dc) * a.ldy + n0 + c * 8) * 2);
    }
    unsigned voffX[2]; int colX[2], rowX[2]; bool realX[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wv + 8 * j;                    // patch piece
        realX[j] = pc < NX;
        const int pr = pc / PB, pb = pc - pr * PB;
        const int px = pb * 8 + (lane >> 3), c = (lane & 7) ^ swz_x(px);
        rowX[j] = pr - 1;                             // image row relative to the K-step's first row
        colX[j] = px - 4;                             // image column relative to x0
        // (the descriptor's base sits (W + 4) pixels in front of the tensor: every lane offset is non-negative)
        voffX[j] = (unsigned)((((long)pr * a.W + px) * a.ldx + c0 + c * 8) * 2);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((((long)a.B * a.H * a.W - 1) * a.ldy + a.N) * 2), 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - (long)(a.W + 4) * a.ldx), 0,
                                                       (int)((((long)a.B * a.H * a.W + 2 * a.W + 16) * a.ldx) * 2), 0x00020000);
#endif
    // the loader's position: K-step f_u = (image f_b, row block f_yb, column block f_xb)
    int f_u = u0, f_b, f_yb, f_xb;
    { const int per_img = RB * CB; f_b = u0 / per_img; const int r = u0 - f_b * per_img; f_yb = r / CB; f_xb = r - f_yb * CB; }
    auto stage = [&](int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
        char* sb = smem + slot * SLOT;
        const bool live = f_u < u1;
        const int y = f_yb * RK, x0 = f_xb * CK;
        const long pix = ((long)f_b * a.H + y) * a.W + x0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lptr_t)(sb + wv * 1024), 16, live ? voffY : OOB9, (int)(pix * a.ldy * 2), 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = y + rowX[j];
            const bool ok = live && realX[j] && (unsigned)iy < (unsigned)a.H && (unsigned)(x0 + colX[j]) < (unsigned)a.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(sb + XOFF + (wv + 8 * j) * 1024), 16, ok ? voffX[j] : OOB9, (int)(pix * a.ldx * 2), 0, 0);
        }
        ++f_u;
        if (++f_xb == CB) { f_xb = 0; if (++f_yb == RB) { f_yb = 0; ++f_b; } }
#else
        (void)slot;
#endif
    };

    // ---- fragment read constants.  16-lane group g holds k = 8 g .. 8 g + 7 (two transposing reads of 4 pixel rows), lane li = 4 q + pp of
    // the group supplies row q, 8-byte quarter pp of the 32-byte channel block
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    unsigned offY[4];                                 // + 1024: the second read (k + 4: same swizzle)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 8 * g + q, cb = wm * 4 + i;
        offY[i] = (unsigned)(k * 256 + (((2 * cb + (pp >> 1)) ^ swz_y(k)) * 16) + (pp & 1) * 8);
    }
    unsigned offX[3][2];                              // [kx][read]; + ky * PW * 128
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int k = 8 * g + q + 4 * rd;
            const int r = RK == 1 ? 0 : (k >> 4);
            const int px = (RK == 1 ? k : (k & 15)) + kx + 3;
            offX[kx][rd] = (unsigned)(XOFF + (r * PW + px) * 128 + (((2 * wn + (pp >> 1)) ^ swz_x(px)) * 16) + (pp & 1) * 8);
        }

    f32x4_t acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#define W9_WAIT_LGKM(N) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
    // taps of one kernel row: x fragments two reads ahead of their MFMAs
    auto tap_row = [&](unsigned sb, const bf16x8_t (&fy)[4], auto KY) {
        constexpr int ky = decltype(KY)::value;
        unsigned long long xa[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            xa[kx][0] = tr16o<ky * PW * 128>(sb + offX[kx][0]);
            xa[kx][1] = tr16o<ky * PW * 128>(sb + offX[kx][1]);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            if (kx == 0) W9_WAIT_LGKM(4) else if (kx == 1) W9_WAIT_LGKM(2) else W9_WAIT_LGKM(0)
            const bf16x8_t fx = frag(xa[kx][0], xa[kx][1]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, fy[i], acc[i][ky * 3 + kx], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    stage(0); stage(1);
    for (int it = 0; it < nst; ++it) {
        // this wave's three pieces of K-step `it` have landed (the three of K-step it + 1 may still be in flight); after the barrier everybody's have
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int slot = it % RING;
        stage((it + 2) % RING);                       // over the slot K-step it - 1 was read from: every wave finished those reads before this barrier
        const unsigned sb = lds0 + slot * SLOT;
        unsigned long long ya[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) { ya[i][0] = tr16(sb + offY[i]); ya[i][1] = tr16o<1024>(sb + offY[i]); }
        W9_WAIT_LGKM(0)
        bf16x8_t fy[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fy[i] = frag(ya[i][0], ya[i][1]);
        tap_row(sb, fy, std::integral_constant<int, 0>{});
        tap_row(sb, fy, std::integral_constant<int, 1>{});
        tap_row(sb, fy, std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zero-filling pieces past the end)

    // ---- epilogue: acc[i][t][e] = dW[n0 + wm * 64 + i * 16 + (lane & 15)][t][c0 + wn * 16 + 4 * (lane >> 4) + e]
    const long Kp = 9L * a.Cin_p;
    float* dst0 = a.dw + (long)split * a.part_stride + (long)(n0 + wm * 64 + (lane & 15)) * Kp + c0 + wn * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) *(f32x4_t*)(dst0 + (long)i * 16 * Kp + (long)t * a.Cin_p) = acc[i][t];
}

int g_cus9 = 0;

}  // namespace

int g_wgrad9 = 1;                                    // development knob (mte_debug_set(26, v)): 0 = the generic per-tap kernel everywhere

// -> MTE_OK and *parts_out slabs written, or MTE_ERR_UNSUPPORTED (the caller takes the generic kernel)
__attribute__((visibility("hidden"))) int wgrad9_launch(const void* x, long ldx, const void* dy, long ldy, float* dw_stage, int parts_cap, int* parts_out,
                                                        int B, int H, int W, int Cin_p, int N, hipStream_t st) {
    if (!g_wgrad9 || N % 128 != 0 || Cin_p % 64 != 0 || parts_cap < 1) return MTE_ERR_UNSUPPORTED;
    const int rk = W % 32 == 0 ? 1 : ((W % 16 == 0 && H % 2 == 0) ? 2 : 0);
    if (!rk) return MTE_ERR_UNSUPPORTED;
    const long M = (long)B * H * W;
    if (((M + 2 * W + 16) * ldx) * 2 >= 0x7ff00000L || ((M - 1) * ldy + N) * 2 >= 0x7ff00000L) return MTE_ERR_UNSUPPORTED;
    if (!g_cus9) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return MTE_ERR_LAUNCH;
        g_cus9 = n;
    }
    Wgrad9Args a{};
    a.x = (const bf16_t*)x; a.ldx = ldx; a.dy = (const bf16_t*)dy; a.ldy = ldy; a.dw = dw_stage;
    a.B = B; a.H = H; a.W = W; a.Cin_p = Cin_p; a.N = N;
    a.tiles_c = Cin_p / 64;
    a.base = (N / 128) * a.tiles_c;
    a.units = (int)(M / 32);
    // one workgroup per CU (72 KB of LDS, 512 threads): pixel splits so that tiles x splits ~ the CU count, at least 12 K-steps each
    long splits = (g_cus9 + a.base / 2) / a.base;
    if (splits < 1) splits = 1;
    if (splits > parts_cap) splits = parts_cap;
    if (splits > a.units / 12) splits = a.units / 12 > 0 ? a.units / 12 : 1;
    a.units_per_split = (int)((a.units + splits - 1) / splits);
    splits = (a.units + a.units_per_split - 1) / a.units_per_split;
    a.part_stride = (long)N * 9 * Cin_p;
    if (parts_out) *parts_out = (int)splits;
    constexpr int LDS = 3 * 24 * 1024;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_wgrad9_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv_wgrad9_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return MTE_ERR_LAUNCH;
        attr = true;
    }
    const dim3 grid((unsigned)(a.base * splits));
    if (rk == 1) hipLaunchKernelGGL(conv_wgrad9_kernel<1>, grid, dim3(512), LDS, st, a);
    else hipLaunchKernelGGL(conv_wgrad9_kernel<2>, grid, dim3(512), LDS, st, a);
    return mte_check_launch();
}
