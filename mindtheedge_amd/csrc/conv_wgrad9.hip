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
//   * 4-slot ring, one s_barrier per K-step, counted vmcnt (three pieces per wave stay in flight across the barrier); the fragment reads run two
//     taps ahead of their MFMAs and cross the K-step boundary (the next K-step's slot is complete one barrier early).
//   * LDS swizzles (on the DMA source chunk, the same XOR on the read): dy rows of 256 B: chunk ^= 2 * ((k & 3) | ((k >> 3) & 1) << 2);
//     x pixels of 128 B: chunk ^= (((px >> 1) & 1) << 1) | (((px >> 3) & 1) << 2) -- the eight 32-byte blocks a 32-lane half of a transposing
//     read touches (pixels s..s+3 and s+8..s+11, any shift s) fall on 64 distinct banks.
//   * operands swapped (D = X^T dY): a lane ends with 4 consecutive INPUT channels of one output channel = one 16-byte store into the
//     [N][9][Cin_p] staging slab of its pixel split (plain stores; mte_unpack_conv_wgrad adds the slabs in order: no atomics, bit-reproducible).
#include "common.hpp"
#include <type_traits>

// Private diagnostic builds (tools/w9_ablate.py; the shipped library defines none of this): leave out pieces of the main loop to see what a K-step
// costs.  bit 0: no LDS-DMA in the loop, 1: no MFMAs, 2: no fragment reads, 3: no barrier, 4: no epilogue stores, 5: no vmcnt wait in the loop,
// 6: only the dy piece of every K-step, 7: every DMA out of range (issued, nothing fetched).  Results are wrong by design.
#ifndef MTE_W9_ABL
#define MTE_W9_ABL 0
#endif

struct Wgrad9Args {
    const bf16_t* x; long ldx;
    const bf16_t* dy; long ldy;
    float* dw; long part_stride;        // slab of pixel split s at dw + s * part_stride, layout [N][9][Cin_p]
    int B, H, W, Cin_p, N;
    int tiles_c;                        // input-channel tiles of 64 (output-channel tiles of 128: gridDim / tiles_c / splits)
    int base;                           // tiles_n * tiles_c
    int units, units_per_split;         // K-steps of 32 pixels: all, per pixel split
};

MTE_CLOCK_DEFINE(wgrad9)

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

// RK = 1: a K-step is 32 consecutive pixels of one image row; RK = 2: 16 columns of two consecutive rows (k < 16: the upper row).
// K-steps are numbered DOWN the image first: u = (image * CB + column block) * RB + row block -- along a column walk only the first and the last
// K-step touch the zero padding, so the loader's per-lane offsets (valid / out of range) change a few times per column, not every K-step.
//
// What bounds the loop (profiles/r05_wgrad9_steps.txt: ablation of the first two versions): a SIMD issues ~one instruction per 4 cycles for BOTH of
// its waves together -- MFMA time, fragment reads, DMA bookkeeping and waits ADDED UP (2,200 cycles per K-step against 1,152 of MFMA pipe time).
// So this version spends instructions only where it must: 36 MFMAs + 26 reads + 3 DMA per wave and K-step, LDS addresses as immediates (the loop
// is unrolled over the four ring slots; two base registers per lane offset), six counted waits, and a loader that adds two scalar strides.
template <int RK>
__global__ __launch_bounds__(512, 2) void conv_wgrad9_kernel(Wgrad9Args a) {
    constexpr int CK = 32 / RK;                       // columns of a K-step
    constexpr int PW = CK + 8, PR = RK + 2;           // patch: PR rows x PW pixels (columns x0 - 4 .. x0 + CK + 3: whole 8-pixel pieces)
    constexpr int PB = PW / 8, NX = PR * PB;          // 1-KiB pieces per patch row, per patch
    constexpr int RING = 4;
    constexpr int YSLOT = 8 * 1024, XSLOT = 16 * 1024; // per ring slot: 8 dy pieces; NX x pieces (+ 16 - NX pieces of padding): 3 pieces per wave and K-step
    constexpr int XBASE = RING * YSLOT;               // LDS: [RING][dy 8 KiB] then [RING][x 16 KiB] -- every slot, tap row and second read is a 16-bit
                                                      // immediate off ONE base register per lane offset (x: 3 * 16 KiB + 2 * PW * 128 < 64 KiB)
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
    const int CB = a.W / CK, RB = a.H / RK;           // column blocks per image row, row blocks per image

    // ---- DMA lane constants.  piece 0: dy pixels 4 wv .. 4 wv + 3 (16 chunks each); pieces 1, 2: patch pieces wv and wv + 8 (8 pixels x 8 chunks)
    unsigned voffY;
    {
        const int k = 4 * wv + (lane >> 4), c = (lane & 15) ^ swz_y(k);
        const int dr = RK == 1 ? 0 : (k >> 4), dc = RK == 1 ? k : (k & 15);
        voffY = (unsigned)((((long)dr * a.W + dc) * a.ldy + n0 + c * 8) * 2);
    }
    unsigned voffX[2];
    // wave-uniform flags of the two patch pieces, 5 bits each, in ONE scalar register (round 6: as ten bools the compiler kept per-lane copies of their products
    // with the lane halves in VGPRs -- the kernel sits at exactly 256 -- and SPILLED them: three scratch reloads, each behind an s_waitcnt vmcnt(0) that drained
    // the DMA ring, on every K-step that re-classifies its lanes: 3 of a column's 12-24):
    //   bit 0 real piece | 1 / 2 the patch row above / below the K-step's own rows | 3 / 4 the piece holds the patch's left / right alignment columns (its lanes
    //   (lane >> 3) < 4 resp. >= 4 lie left / right of the K-step's own columns + 1)
    unsigned fl = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wv + 8 * j;                    // patch piece
        const int pr = pc / PB, pb = pc - pr * PB;
        const int px = pb * 8 + (lane >> 3), c = (lane & 7) ^ swz_x(px);
        fl |= ((unsigned)(pc < NX) | ((unsigned)(pr == 0) << 1) | ((unsigned)(pr == PR - 1) << 2) | ((unsigned)(pb == 0) << 3) | ((unsigned)(pb == PB - 1) << 4)) << (8 * j);
        // (the descriptor's base sits (W + 4) pixels in front of the tensor: every lane offset is non-negative)
        voffX[j] = (unsigned)((((long)pr * a.W + px) * a.ldx + c0 + c * 8) * 2);
    }
    fl = (unsigned)__builtin_amdgcn_readfirstlane((int)fl);
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((((long)a.B * a.H * a.W - 1) * a.ldy + a.N) * 2), 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - (long)(a.W + 4) * a.ldx), 0,
                                                       (int)((((long)a.B * a.H * a.W + 2 * a.W + 16) * a.ldx) * 2), 0x00020000);
#endif
    // ---- the loader's position: K-step f_u = (image f_b, column block f_xb, row block f_yb); byte offsets of its first pixel; the lane offsets in use
    int f_u = u0, f_b, f_xb, f_yb;
    { const int per_img = RB * CB; f_b = u0 / per_img; const int r = u0 - f_b * per_img; f_xb = r / RB; f_yb = r - f_xb * RB; }
    int soffY, soffX;
    unsigned voY, voX[2];
    const int stepY = (int)((long)RK * a.W * a.ldy * 2), stepX = (int)((long)RK * a.W * a.ldx * 2);
    const unsigned inner = RB > 3 ? (unsigned)(RB - 3) : 0u;          // row blocks 2 .. RB - 2 need no look at the lane offsets (none when RB < 4)
    auto locate = [&]() {                             // offsets of the column's first K-step
        const long pix = ((long)f_b * a.H + (long)f_yb * RK) * a.W + f_xb * CK;
        soffY = (int)(pix * a.ldy * 2); soffX = (int)(pix * a.ldx * 2);
    };
    auto classify = [&]() {                           // which lanes of this K-step's pieces lie outside the image (or past the split's end)
        const bool live = f_u < u1;
        const bool top = f_yb == 0, bot = f_yb == RB - 1, left = f_xb == 0, right = f_xb == CB - 1;
        voY = live ? voffY : OOB9;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // scalar: the whole piece, its lower / upper lane half; then ONE 64-bit lane mask and a shift per lane (no per-lane state beside `lane`)
            const unsigned f = fl >> (8 * j);
            const bool all_bad = (!live) | (!(f & 1u)) | (top & (bool)((f >> 1) & 1u)) | (bot & (bool)((f >> 2) & 1u));
            const bool lo_bad = left & (bool)((f >> 3) & 1u), hi_bad = right & (bool)((f >> 4) & 1u);
            const unsigned long long m = all_bad ? ~0ull : ((lo_bad ? 0x00000000ffffffffull : 0ull) | (hi_bad ? 0xffffffff00000000ull : 0ull));
            voX[j] = ((m >> lane) & 1ull) ? OOB9 : voffX[j];
        }
    };
    locate(); classify();
    auto stage = [&](auto S) {
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int slot = decltype(S)::value;
        if (MTE_W9_ABL & 128) { voY = OOB9; voX[0] = OOB9; voX[1] = OOB9; }
        if (!(MTE_W9_ABL & 1)) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lptr_t)(smem + slot * YSLOT + wv * 1024), 16, voY, soffY, 0, 0);
            if (!(MTE_W9_ABL & 64)) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(smem + XBASE + slot * XSLOT + wv * 1024), 16, voX[0], soffX, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(smem + XBASE + slot * XSLOT + 8192 + wv * 1024), 16, voX[1], soffX, 0, 0);
            }
        }
        // on to the next K-step: one row block down; the lane offsets change only at the ends of a column
        ++f_u; ++f_yb;
        soffY += stepY; soffX += stepX;
        if ((unsigned)(f_yb - 2) >= inner || f_u >= u1) {                       // f_yb in {1, RB - 1, RB (wraps below)} or the end of the split: rare
            if (f_yb >= RB) {
                f_yb = 0;
                if (++f_xb == CB) { f_xb = 0; ++f_b; }
                locate();
            }
            classify();
        }
#else
        (void)S;
#endif
    };

    // ---- fragment read addresses.  16-lane group g holds k = 8 g .. 8 g + 7 (two transposing reads of 4 pixel rows), lane li = 4 q + pp of the
    // group supplies row q, 8-byte quarter pp of the 32-byte channel block.  One register per lane offset; ring slot, tap row and second read are immediates.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    unsigned adY[4];                                  // [block]; + slot * YSLOT, + 1024: the second read (k + 4: same swizzle)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = 8 * g + q, cb = wm * 4 + i;
        adY[i] = lds0 + (unsigned)(k * 256 + (((2 * cb + (pp >> 1)) ^ swz_y(k)) * 16) + (pp & 1) * 8);
    }
    unsigned adX[3][2];                               // [kx][read]; + slot * XSLOT + ky * PW * 128
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int k = 8 * g + q + 4 * rd;
            const int r = RK == 1 ? 0 : (k >> 4);
            const int px = (RK == 1 ? k : (k & 15)) + kx + 3;
            adX[kx][rd] = lds0 + XBASE + (unsigned)((r * PW + px) * 128 + (((2 * wn + (pp >> 1)) ^ swz_x(px)) * 16) + (pp & 1) * 8);
        }

    f32x4_t acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define W9_WAIT_LGKM(N) { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
#define W9_C(N) std::integral_constant<int, N>{}
    // ---- main loop: two PHASES per K-step, the two wave groups (waves w and w + 4 share a SIMD) one barrier apart -- the scheme of the 8-phase
    // forward kernel (conv_igemm8.hip).  Measured on the first versions of this kernel (profiles/r05_wgrad9_steps.txt): with both waves of a SIMD
    // running the same instruction mix, MFMA time, fragment reads and LDS-DMA issue ADD UP (a DMA instruction holds its wave ~100 cycles); apart by a
    // barrier, one wave's load phase sits under the other's MFMAs.
    //   L(i): DMA of K-step i + 3 (3 pieces) | all 26 fragment reads of K-step i into registers | vmcnt: own pieces of K-step i + 1 landed | lgkmcnt(0) | barrier
    //   M(i): 36 MFMAs from registers | barrier
    // Hazards by barrier count (group 0: L(i) between barriers 2i and 2i + 1, M(i) up to 2i + 2; group 1 one later):
    //   RAW  a wave reads slot i + 1 in L(i + 1); every wave waited for its pieces of K-step i + 1 before the barrier that ends its L(i), which is not
    //        later than barrier 2i + 2, where the earlier group's L(i + 1) begins.
    //   WAR  stage i + 3 goes over slot i - 1 (RING = 4); its last reads are retired (lgkmcnt(0)) before the barrier that ends group 1's L(i - 1) =
    //        barrier 2i, where group 0's L(i) begins.
    unsigned long long ya[4][2], xt[9][2];
    auto rd_y = [&](auto S) {
        constexpr int s = decltype(S)::value, o = s * YSLOT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (MTE_W9_ABL & 4) { asm volatile("" : "+v"(ya[i][0]), "+v"(ya[i][1])); continue; }
            ya[i][0] = tr16o<o>(adY[i]); ya[i][1] = tr16o<o + 1024>(adY[i]);
        }
    };
    auto rd_x = [&](auto S, auto T) {
        constexpr int s = decltype(S)::value, t = decltype(T)::value, ky = t / 3, kx = t % 3, o = s * XSLOT + ky * PW * 128;
        if (MTE_W9_ABL & 4) { asm volatile("" : "+v"(xt[t][0]), "+v"(xt[t][1])); return; }
        xt[t][0] = tr16o<o>(adX[kx][0]); xt[t][1] = tr16o<o>(adX[kx][1]);
    };
    auto mma = [&](auto T) {
        constexpr int t = decltype(T)::value;
        if (MTE_W9_ABL & 2) { asm volatile("" ::"v"(xt[t][0]), "v"(xt[t][1]), "v"(ya[0][0]), "v"(ya[1][1]), "v"(ya[2][0]), "v"(ya[3][1])); return; }
        const bf16x8_t fx = frag(xt[t][0], xt[t][1]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, frag(ya[i][0], ya[i][1]), acc[i][t], 0, 0, 0);
    };
#define W9_BARRIER() { __builtin_amdgcn_sched_barrier(0); if (!(MTE_W9_ABL & 8)) __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    auto kstep = [&](auto S) {
        constexpr int s = decltype(S)::value;
        stage(W9_C((s + 3) & 3));
        rd_y(S);
        rd_x(S, W9_C(0)); rd_x(S, W9_C(1)); rd_x(S, W9_C(2)); rd_x(S, W9_C(3)); rd_x(S, W9_C(4));
        rd_x(S, W9_C(5)); rd_x(S, W9_C(6)); rd_x(S, W9_C(7)); rd_x(S, W9_C(8));
        if (!(MTE_W9_ABL & 32)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        W9_WAIT_LGKM(0)
        W9_BARRIER()
        __builtin_amdgcn_s_setprio(1);
        mma(W9_C(0)); mma(W9_C(1)); mma(W9_C(2)); mma(W9_C(3)); mma(W9_C(4)); mma(W9_C(5)); mma(W9_C(6)); mma(W9_C(7)); mma(W9_C(8));
        __builtin_amdgcn_s_setprio(0);
        W9_BARRIER()
    };

    stage(W9_C(0)); stage(W9_C(1)); stage(W9_C(2));
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K-step 0 has landed
    W9_BARRIER()
    if (wm == 1) W9_BARRIER()                          // group 1 runs one barrier behind from here on
    // (no early exit from the unrolled body: with exits between the four K-steps the compiler no longer keeps the 144 accumulator registers in place
    //  -- 441 spilled registers; a split whose K-step count is not a multiple of four runs up to three K-steps on zero-filled slots instead, and the
    //  launcher deals multiples of four)
    MTE_CLOCK_BEGIN()
    for (int it = 0; it < nst; it += 4) { kstep(W9_C(0)); kstep(W9_C(1)); kstep(W9_C(2)); kstep(W9_C(3)); }
    MTE_CLOCK_END(wgrad9)
    if (wm == 0) W9_BARRIER()
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the zero-filling pieces past the end)

    // ---- epilogue: acc[i][t][e] = dW[n0 + wm * 64 + i * 16 + (lane & 15)][t][c0 + wn * 16 + 4 * (lane >> 4) + e]
    const long Kp = 9L * a.Cin_p;
    float* dst0 = a.dw + (long)split * a.part_stride + (long)(n0 + wm * 64 + (lane & 15)) * Kp + c0 + wn * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if ((MTE_W9_ABL & 16) && acc[i][t][0] != 12345.f) continue;
            *(f32x4_t*)(dst0 + (long)i * 16 * Kp + (long)t * a.Cin_p) = acc[i][t];
        }
}

int g_cus9 = 0;

}  // namespace

int g_wgrad9 = 1;                                    // development knob (mte_debug_set(26, v)): 0 = the generic per-tap kernel everywhere
// Workgroups aimed for per launch when the caller runs the weight gradients BESIDE the data-gradient chain (MTE_OPT_WGRAD_SHARES_CHIP): HALF the chip.
// With one workgroup per CU this kernel's 96 KB of LDS, 512 threads and whole register file take every CU away from the main queue's kernels for its whole length
// and it writes twice the slabs (what the main queue loses is less the CUs than the clock: profiles/r05_overlap_probe.txt).  Same-box step times (profiles/r05_side_queue_width.txt): 256 -> 23.59 ms, 192 -> 23.50, 128 -> 23.26, 96 -> 23.35, 64 -> 23.79.
// Alone on the chip (option off: serial profiling runs, a binding without a second stream) it takes one workgroup per CU.
#ifndef MTE_W9_WGS
#define MTE_W9_WGS 128
#endif
int g_wgrad9_wgs = MTE_W9_WGS;                       // development knob (mte_debug_set(27, v)): workgroups aimed for per shared-chip launch (0 = one per CU)

static int wgrad9_form(int H, int W, int Cin_p, int N) {     // 0: not this kernel's; 1: 1 x 32 K-steps; 2: 2 x 16
    if (!g_wgrad9 || N % 128 != 0 || Cin_p % 64 != 0) return 0;
    // 1 x 32 K-steps where the rows allow it: measured 2-4 % faster on the 48x160 layers than 2 x 16 ones although they stage 15 patch pieces against 12
    // (development knob 26 = 2: 2 x 16 first)
    const bool ok2 = W % 16 == 0 && H % 2 == 0, ok1 = W % 32 == 0;
    return g_wgrad9 == 2 ? (ok2 ? 2 : (ok1 ? 1 : 0)) : (ok1 ? 1 : (ok2 ? 2 : 0));
}
extern "C" int mte_conv2d_wgrad_nine_tap(int H, int W, int Cin_p, int N, int KH, int KW, int dtype) {
    return dtype == MTE_DT_BF16 && KH == 3 && KW == 3 && H > 0 && W > 0 && wgrad9_form(H, W, Cin_p, N) ? 1 : 0;
}

// -> MTE_OK and *parts_out slabs written, or MTE_ERR_UNSUPPORTED (the caller takes the generic kernel)
__attribute__((visibility("hidden"))) int wgrad9_launch(const void* x, long ldx, const void* dy, long ldy, float* dw_stage, int parts_cap, int* parts_out,
                                                        int B, int H, int W, int Cin_p, int N, hipStream_t st) {
    if (parts_cap < 1) return MTE_ERR_UNSUPPORTED;
    const int rk = wgrad9_form(H, W, Cin_p, N);
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
    // one workgroup per CU (96 KB of LDS, 512 threads): pixel splits so that tiles x splits ~ the CU count, at least 12 K-steps each
    const int target = (g_mte_wgrad_shared && g_wgrad9_wgs > 0) ? g_wgrad9_wgs : g_cus9;
    long splits = (target + a.base / 2) / a.base;
    if (splits < 1) splits = 1;
    if (splits > parts_cap) splits = parts_cap;
    if (splits > a.units / 12) splits = a.units / 12 > 0 ? a.units / 12 : 1;
    a.units_per_split = (int)((a.units + splits - 1) / splits);
    a.units_per_split = (a.units_per_split + 3) & ~3;  // the main loop is unrolled over its four ring slots
    splits = (a.units + a.units_per_split - 1) / a.units_per_split;
    a.part_stride = (long)N * 9 * Cin_p;
    if (parts_out) *parts_out = (int)splits;
    constexpr int LDS = 4 * (8 + 16) * 1024;
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
