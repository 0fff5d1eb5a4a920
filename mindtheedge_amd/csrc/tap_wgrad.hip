// Weight gradient of a 3 x 3 convolution with ONE channel on one side, on the matrix cores (round 5):
//     rec[c][tap] = sum over pixels p of  x[p][c] * m[p + sgn * (tap - 1)]        (tap = (ty, tx), tap - 1 = (ty - 1, tx - 1); m = 0 outside the image)
// x: a bf16 NHWC tensor (C = 32 / 64 / 128 / 256 channels of rows of ldx), m: a one-channel fp32 map (optionally the nearest-neighbour up-sampling of a
// half-resolution map).  Two callers (heads_misc.hip):
//   * the InvDepth head (reference packnet/layers01.py:99-123: Conv2d(C, 1, 3) + sigmoid): dw[c][tap] = sum x[q][c] dlogit[q - (tap - 1)], sgn = -1, + db = sum dlogit;
//   * the inverse-depth input channel of the decoder's iconv3/2/1 (PackNetSAN01.py:118-143), kept as a rank-1 term: dw[n][tap] = sum dy[p][n] up2(inv)[p + tap - 1],
//     sgn = +1, x = dy.
// The VALU kernels these replace (invdepth_bwd_weight_kernel, rank1_conv_bwd_weight_kernel: 72 FMAs per 16 bytes loaded) ran at 0.6-1.1 TB/s: 0.39 + 0.17 ms
// per training step for 0.24 + 0.2 GB -- the fp32 vector rate bounds them at about the time HBM needs.  Here the sum is a GEMM with K = pixels:
//   A (M = 16 channels x K = 32 pixels)  = x^T : transposing LDS reads (ds_read_b64_tr_b16) of a [32 pixels][C] tile brought in by LDS-DMA,
//   B (K = 32 pixels x N = 16 columns)   = the nine shifted copies of m, built in registers from three staged rows as bf16 HI and LO halves (m = hi + lo to
//                                          2^-17: two MFMAs per channel block -- the matrix pipe has 30x the time it needs), columns 9..15 zero,
//   D[c][tap] accumulated in fp32 over the wave's pixel range.
// A unit is 32 consecutive pixels of one image row (W % 32 == 0: units tile the tensor linearly).  Waves are independent: each owns a contiguous range of
// units and a private ring of LDS slots (x tile + the three map rows), filled RING - 1 units ahead with counted vmcnt -- no barrier in the loop.  At the end the
// four waves of a workgroup add their sums through LDS in a fixed order and leave ONE record; the callers' reduce kernels add the records in order
// (bit-reproducible, no floating-point atomics).  HBM-bound by design: x is read exactly once, m three times from L2.
#include "common.hpp"
#include <type_traits>

struct TapWgradArgs {
    const bf16_t* x; long ldx;
    const float* m;                     // [B][H][W] fp32, or [B][H/2][W/2] when up2
    float* rec; long rec_stride;        // record of workgroup g at rec + g * rec_stride: [C][9] (+ [C * 9] = sum of m over the image when has_db)
    int B, H, W;                        // of x (full resolution)
    int sgn, up2, has_db;
    int units, units_per_wave;          // units of 32 pixels: all, per wave
};

namespace {

constexpr unsigned OOB_T = 0xfffffff0u;

__device__ __forceinline__ unsigned long long tap_tr16(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF> __device__ __forceinline__ unsigned long long tap_tr16o(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// XOR on the 32-byte channel-block index of pixel row px of the x tile (rows of 2 C bytes): the eight 32-byte blocks a 32-lane half of a transposing read
// touches (pixels 8 g .. 8 g + 3 of two groups g, one channel block) must fall on 8 different 32-byte bank groups.  Rows of 64 B place pixels 0..3 already
// (stride 2 groups): only the second group needs moving; rows of 128 B place the pixel parity; rows >= 256 B place nothing.
template <int CB> __device__ __forceinline__ int tap_swz(int px) {
    if constexpr (CB == 2) return (px >> 3) & 1;
    else if constexpr (CB == 4) return ((px >> 1) & 1) | (((px >> 3) & 1) << 1);
    else return (px & 3) | (((px >> 3) & 1) << 2);
}

template <int CB, int RING>
__global__ __launch_bounds__(256) void tap_wgrad_mfma_kernel(TapWgradArgs a) {
    constexpr int C = 16 * CB, ROWB = 2 * C;          // channels; bytes of one pixel row of the tile
    constexpr int XBYTES = 32 * ROWB, NX = XBYTES / 1024;   // x tile of a unit; its 1-KiB DMA pieces (2 .. 16)
    constexpr int MROW = 34, MBYTES = 512;            // map rows y - 1 .. y + 1, columns x0 - 1 .. x0 + 32: 3 x 34 floats (two DMA pieces of 4 B per lane)
    constexpr int SLOT = XBYTES + MBYTES, NI = NX + 2;
    constexpr int CPR = 2 * CB;                       // 16-byte chunks per pixel row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(wv * RING * SLOT);
    const int gw = blockIdx.x * 4 + wv;
    const long u0l = (long)gw * a.units_per_wave;
    const int u0 = (int)(u0l < a.units ? u0l : a.units);
    const int n = min(a.units_per_wave, a.units - u0);      // this wave's units (0: past the end -- it still takes part in the final sum)

    // ---- DMA lane constants: piece i covers 16-byte chunks 64 i + lane of the [32][C] tile; the SOURCE chunk carries the swizzle
    unsigned voX[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int L = i * 64 + lane, px = L / CPR, c = (L % CPR) ^ (tap_swz<CB>(px) << 1);
        voX[i] = (unsigned)(((long)px * a.ldx + c * 8) * 2);
    }
    int mr[2], mc[2];                                 // map pieces: float i = 64 j + lane of [3][34]
    bool mlive[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int i = 64 * j + lane; mr[j] = i / MROW; mc[j] = i - mr[j] * MROW; mlive[j] = i < 3 * MROW; }
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const long npix = (long)a.B * a.H * a.W;
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(((npix - 1) * a.ldx + C) * 2), 0x00020000);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc((void*)a.m, 0, (int)((a.up2 ? npix / 4 : npix) * 4), 0x00020000);
#endif
    // ---- the loader's position
    int f_x0, f_y, f_b;
    { const long p0 = (long)u0 * 32; f_x0 = (int)(p0 % a.W); const long t = p0 / a.W; f_y = (int)(t % a.H); f_b = (int)(t / a.H); }
    int f_soff = (int)((long)u0 * 32 * a.ldx * 2);
    const int f_step = (int)(32 * a.ldx * 2);
    const int hl = a.H >> 1, wl = a.W >> 1;
    auto stage = [&](int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
        char* dst = smem + wv * RING * SLOT + slot * SLOT;
#pragma unroll
        for (int i = 0; i < NX; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(dst + i * 1024), 16, voX[i], f_soff, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yy = f_y - 1 + mr[j], xx = f_x0 - 1 + mc[j];
            const bool ok = mlive[j] & ((unsigned)yy < (unsigned)a.H) & ((unsigned)xx < (unsigned)a.W);
            const int idx = a.up2 ? ((f_b * hl + (yy >> 1)) * wl + (xx >> 1)) : ((f_b * a.H + yy) * a.W + xx);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (lptr_t)(dst + XBYTES + j * 256), 4, ok ? (unsigned)idx * 4u : OOB_T, 0, 0, 0);
        }
        f_soff += f_step;
        f_x0 += 32;
        if (f_x0 == a.W) { f_x0 = 0; if (++f_y == a.H) { f_y = 0; ++f_b; } }
#else
        (void)slot;
#endif
    };

    // ---- fragment addresses.  A: 16-lane group g holds pixels 8 g .. 8 g + 7 (two transposing reads of 4 pixel rows), lane li = 4 q + pp of the group
    // supplies row q, 8-byte quarter pp of the 32-byte channel block.  B: lane (column li = tap, group g) holds m at pixels 8 g .. 8 g + 7 shifted by its tap.
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    unsigned adA[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const int px = 8 * g + q;
        adA[cb] = lds0 + (unsigned)(px * ROWB + (((2 * cb + (pp >> 1)) ^ (tap_swz<CB>(px) << 1)) * 16) + (pp & 1) * 8);
    }
    const int tap = li < 9 ? li : 4, ty = tap / 3, tx = tap - ty * 3;
    const unsigned adB = lds0 + (unsigned)(XBYTES + ((1 + a.sgn * (ty - 1)) * MROW + 1 + a.sgn * (tx - 1) + 8 * g) * 4);
    const bool colB = li < 9, centre = li == 4;

    f32x4_t acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.f;

    // ---- main loop: unit k of the wave in slot k % RING; DMA runs RING - 1 units ahead
#pragma unroll
    for (int k = 0; k < RING - 1; ++k) if (k < n) stage(k);
    int slot = 0, fslot = RING - 1;
    for (int k = 0; k < n; ++k) {
        const int ahead = n - 1 - k;                  // units after this one
        if (ahead >= RING - 1) { stage(fslot); fslot = fslot + 1 == RING ? 0 : fslot + 1; }
        // own pieces of unit k landed: at most min(ahead, RING - 1) later units may still be in flight
        if (ahead >= RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * NI) : "memory");
        else if (RING > 2 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned so = (unsigned)(slot * SLOT);
        unsigned long long alo[CB], ahi[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) { alo[cb] = tap_tr16(adA[cb] + so); ahi[cb] = tap_tr16o<4 * ROWB>(adA[cb] + so); }
        float mv[8];
        {
            const unsigned ab = adB + so;
            asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\t"
                         "ds_read_b32 %4, %8 offset:16\n\tds_read_b32 %5, %8 offset:20\n\tds_read_b32 %6, %8 offset:24\n\tds_read_b32 %7, %8 offset:28"
                         : "=&v"(mv[0]), "=&v"(mv[1]), "=&v"(mv[2]), "=&v"(mv[3]), "=&v"(mv[4]), "=&v"(mv[5]), "=&v"(mv[6]), "=&v"(mv[7])
                         : "v"(ab) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        u32x4_t bh, bl;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v0 = colB ? mv[2 * j] : 0.f, v1 = colB ? mv[2 * j + 1] : 0.f;
            const bf16_t h0 = f2bf(v0), h1 = f2bf(v1);
            bh[j] = (unsigned)h0 | ((unsigned)h1 << 16);
            bl[j] = pack2bf(v0 - bf2f(h0), v1 - bf2f(h1));
            s += v0 + v1;
        }
        dbacc += centre ? s : 0.f;
        const bf16x8_t fbh = __builtin_bit_cast(bf16x8_t, bh), fbl = __builtin_bit_cast(bf16x8_t, bl);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const u32x4_t av{(unsigned)alo[cb], (unsigned)(alo[cb] >> 32), (unsigned)ahi[cb], (unsigned)(ahi[cb] >> 32)};
            const bf16x8_t fa = __builtin_bit_cast(bf16x8_t, av);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fbh, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fbl, acc[cb], 0, 0, 0);
        }
        slot = slot + 1 == RING ? 0 : slot + 1;
    }

    // ---- the workgroup's record: wave sums through LDS (the rings are dead: every staged unit was consumed), added in a fixed order
    constexpr int NREC = C * 9 + 1;
    float* sred = (float*)smem;
    __syncthreads();
    // D[channel 16 cb + 4 g + r][tap li]
    if (li < 9) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sred[wv * NREC + (16 * cb + 4 * g + r) * 9 + li] = acc[cb][r];
    }
    {
        // centre-tap lanes (li == 4) of the four groups hold the sums of m over their pixels
        float d = centre ? dbacc : 0.f;
        d = wave_sum(d);
        if (lane == 0) sred[wv * NREC + C * 9] = d;
    }
    __syncthreads();
    float* rec = a.rec + (long)blockIdx.x * a.rec_stride;
    const int nout = a.has_db ? NREC : NREC - 1;
    for (int i = tid; i < nout; i += 256) rec[i] = (sred[i] + sred[NREC + i]) + (sred[2 * NREC + i] + sred[3 * NREC + i]);
}

int g_tap_wgrad = 1;                                  // development knob (mte_debug_set(31, v)): 0 = the VALU kernels
int g_tap_cus = 0;

template <int CB, int RING>
int tap_launch(TapWgradArgs a, int max_wgs, int* groups, hipStream_t st) {
    constexpr int C = 16 * CB;
    constexpr size_t ring = (size_t)4 * RING * (64 * C + 512), red = (size_t)4 * (C * 9 + 1) * sizeof(float);
    constexpr size_t lds = ring > red ? ring : red;
    static int configured = 0;
    if (!configured) {
        if (hipFuncSetAttribute((const void*)tap_wgrad_mfma_kernel<CB, RING>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return MTE_ERR_LAUNCH;
        configured = 1;
    }
    if (!g_tap_cus) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return MTE_ERR_LAUNCH;
        g_tap_cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    int per_cu = (int)((160 * 1024) / lds); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    long g = (long)g_tap_cus * per_cu;
    if (g > max_wgs) g = max_wgs;
    const long by_work = (a.units + 4 * 4 - 1) / (4 * 4);                // >= 4 units per wave
    if (g > by_work) g = by_work;
    if (g < 1) g = 1;
    a.units_per_wave = (int)((a.units + 4 * g - 1) / (4 * g));
    g = (a.units + 4L * a.units_per_wave - 1) / (4L * a.units_per_wave);  // (no workgroup without a unit)
    *groups = (int)g;
    hipLaunchKernelGGL((tap_wgrad_mfma_kernel<CB, RING>), dim3((unsigned)g), dim3(256), lds, st, a);
    return mte_check_launch();
}

}  // namespace

// MTE_ERR_UNSUPPORTED: the caller's VALU kernel takes the launch.  *groups = records written.
int tap_wgrad_mfma_launch(const bf16_t* x, long ldx, const float* m, float* rec, long rec_stride, int B, int H, int W, int C, int sgn, int up2, int has_db,
                          int max_wgs, int* groups, hipStream_t st) {
    if (!g_tap_wgrad) return MTE_ERR_UNSUPPORTED;
    const long npix = (long)B * H * W;
    if (W % 32 != 0 || (C != 32 && C != 64 && C != 128 && C != 256) || ldx % 8 != 0 || ((uintptr_t)x & 15) || npix * ldx * 2 >= 0x7ff00000L ||
        npix * 4 >= 0x7ff00000L || (up2 && ((H | W) & 1)))
        return MTE_ERR_UNSUPPORTED;
    TapWgradArgs a{};
    a.x = x; a.ldx = ldx; a.m = m; a.rec = rec; a.rec_stride = rec_stride; a.B = B; a.H = H; a.W = W; a.sgn = sgn; a.up2 = up2; a.has_db = has_db;
    a.units = (int)(npix / 32);
    switch (C) {
        case 32: return tap_launch<2, 3>(a, max_wgs, groups, st);
        case 64: return tap_launch<4, 3>(a, max_wgs, groups, st);
        case 128: return tap_launch<8, 3>(a, max_wgs, groups, st);
        default: return tap_launch<16, 2>(a, max_wgs, groups, st);
    }
}
#ifdef MTE_DEV
extern "C" int mtei_set_tap_wgrad(int v) { g_tap_wgrad = v; return MTE_OK; }
#endif
