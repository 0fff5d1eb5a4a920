// Implicit-GEMM 2-D convolution (stride 1, zero pad k/2) for gfx950, NHWC activations.
//
//   forward / dgrad :  y[m][n] = bias[n] + sum_{tap,c} x[pix(m)+tap][c] * w[n][tap][c]
//        GEMM view   :  M = B*H*W pixels, N = C_out, K = KH*KW*Cin_p   (dgrad = same kernel with the
//                       180-degree-rotated, in/out-swapped weight pack produced by mte_pack_conv_weights)
//   wgrad           :  dw[n][tap][c] = sum_m dy[m][n] * x[pix(m)+tap][c]   (reduction over pixels)
//
// Replaces the reference's nn.Conv2d + ConstantPad2d pairs (packnet_sfm/networks/layers/packnet/
// layers01.py:29-31,61,116-117) and their autograd backward.  The pad is folded into the tile loader
// (out-of-image taps load zeros), no padded copy is ever materialised.
//
// Data movement is expressed in 16-byte chunks so one template serves both element types:
//   bf16 : chunk = 8 elements, v_mfma_f32_32x32x16_bf16  (one MFMA per 32 bytes of K per row)
//   f32  : chunk = 4 elements, v_mfma_f32_32x32x2_f32 x4 (exact-fp32 validation mode, 1/16 rate)
// Lane (r = lane&31, h = lane>>5) reads the 16-byte chunk (2*kk + h) of tile row r: for bf16 that is
// k = 16*kk + 8*h + j (the native operand map); for f32 the four MFMAs take component i of both
// operands, i.e. k = 8*kk + 4*h + i -- any k assignment is valid as long as A and B agree.
//
// Tile: 256 threads = 4 waves arranged WM x WN, each wave owns (TM*32) x (TN*32) outputs in fp32
// accumulators; K step = 64 bytes per row; LDS double-buffered, register-staged (global loads of step
// s+1 are in flight while step s computes).  LDS rows are 64 B; 16-B chunk c of row r is stored at chunk
// slot c ^ ((r>>2)&3), which makes every ds_read_b128 lane group hit 16 distinct 16-B bank slots.
#include "common.hpp"
#include "conv_args.hpp"

// conv_wgrad9.hip: the nine-tap 3x3 weight gradient (bf16).  MTE_ERR_UNSUPPORTED when the shape is outside what the kernel covers.
__attribute__((visibility("hidden"))) int wgrad9_launch(const void* x, long ldx, const void* dy, long ldy, float* dw_stage, int parts_cap, int* parts_out,
                                                        int B, int H, int W, int Cin_p, int N, hipStream_t st);

// In-loop s_memtime sums of the DMA main loop (diagnostic build only: -DMTE_STAMPS, tools/igemm_stamps.py).  Round-3 reading, cycles per K-step
// and wave: 256 x 128 tile (512 -> 512 @24x80) stage wait 57 | barrier 314 | DMA issue 267 | fragment reads + MFMA issue 468 | total 1192
// (MFMA pipe busy 512); 256 x 256 tile (256 -> 256 @48x160) 68 | 861 | 207 | 484 | 1707 (pipe busy 1024).  Issuing the DMA behind the MFMAs
// instead of in front of the fragment reads was tried on that evidence and lost 1.7 % (same-box A/B), s_setprio around the MFMAs 0.7 %.
// The ping-pong loop built on it (256 -> 256 @48x160): load phase 499 | first barrier 101 | MFMA issue 568 | wait + second barrier 179 per K-step
// and wave -- the two groups' MFMA phases fill ~76 % of a SIMD's K-step (60 % before).
#ifdef MTE_STAMPS
__device__ unsigned long long g_igemm_stamps[16384 * 8];
extern "C" int mtei_igemm_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_igemm_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

namespace {


template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x16_t& c) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[i]), __uint_as_float(b[i]), c, 0, 0, 0);
    }
};

// MFMA shape of the forward / data-gradient kernel.  1 (round 3): v_mfma_f32_16x16x32_bf16 -- the same FLOPs per issue cycle as
// 32x32x16, but MFMA-dense loops on random data hold a higher clock with it (MI355X_MICROARCH.md, DVFS item 7: 1.12-1.15x the FLOP/s
// at equal cycles per FLOP).  One K-step (64 B per row) is exactly one 16x16x32 reduction: lane (r16 = lane & 15, q = lane >> 4) reads
// the 16-byte chunk q of tile row r16 -- the same LDS bytes per MFMA FLOP as before.  0 = the 32x32x16 form (same-box A/B builds).
#ifndef MTE_IGEMM_MFMA16
#define MTE_IGEMM_MFMA16 1
#endif
#if MTE_IGEMM_MFMA16
// chunk c of row r sits at slot c ^ ((r >> 1) & 3): conflict-free for the 16-row fragment reads (every ds_read_b128 lane group
// {0-3, 12-15, 20-27}, ... meets 16 distinct 16-byte slots; the (r >> 2) form of the 32-row reads is 2-way conflicted here)
__device__ __forceinline__ int lds_swz(int row) { return (row >> 1) & 3; }
#else
__device__ __forceinline__ int lds_swz(int row) { return (row >> 2) & 3; }
#endif
__device__ __forceinline__ int lds_chunk_off(int row, int kc) { return row * 64 + ((kc ^ lds_swz(row)) << 4); }

template <typename T> struct Mma16;
template <> struct Mma16<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i]), __uint_as_float(b[i]), c, 0, 0, 0);
    }
};

__device__ u32x4_t g_zero16 = {0u, 0u, 0u, 0u};      // source of zero chunks for the LDS-DMA loader (image border / K tail)

// DMA = true: tiles are staged global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write); the
// LDS image is lane-linear per wave instruction, so the XOR swizzle is applied to the per-lane SOURCE chunk instead.
// LD selects the tile loader: 0 = register staging (2 LDS buffers), 1 = LDS-DMA with per-lane global pointers (any shape),
// 2 = LDS-DMA through buffer descriptors (Cin_p % 32 == 0, tensors < 2 GiB): the per-lane byte offset is constant per
// filter tap and the K advance is a wave-uniform SGPR offset, so a K-step costs ~4 VALU instead of ~130 -- the loader's
// address arithmetic, not HBM or LDS, was what held the MFMA pipe at ~25 %.  Out-of-image taps use an out-of-range offset
// (the buffer bounds check returns zeros).
// RING = LDS ring slots of the DMA loaders (RING - 1 K-steps in flight); MINW = waves per SIMD the register allocation must allow.
// Measured on the 8-wave 256 x 128 tile (round-2 A/B, profiles/README.md): six slots instead of four change nothing (7.56 vs
// 7.52 ms / step over all launches) -- the loop is not short of bytes in flight; three slots with two workgroups per CU (74 KB
// each, MINW = 4) are 15-20 % faster on the short-reduction layers because one workgroup's prologue / epilogue overlaps the
// other's main loop, but the pair claims 148 of the 160 KB of LDS and slows the step down when the weight-gradient stream wants
// the same CUs, so it is used for MTE_CONV_SOLO launches only (forward pass, inference).  Two wave groups half a K-step apart
// and fragment reads one K-step ahead of the MFMAs were also tried: no gain, removed.  tools/igemm_ablate.py (ABL below) shows
// why: LDS-DMA alone takes as long as the MFMAs alone (~ 50 of the CU's 64 B / clk L1 -> LDS path), so the tile's bytes per
// flop, not the schedule, bound the loop.
// ABL (development builds only, -DMTE_DEV, tools/igemm_ablate.py): bit set of main-loop pieces to LEAVE OUT (1 MFMAs, 2 in-loop
// LDS-DMA, 4 fragment ds_reads) -- the time that remains when a piece is removed says which pipe the loop is waiting on.  Results
// are garbage when ABL != 0; the product library never instantiates ABL != 0.
template <typename T, int WM, int WN, int TM, int TN, int LD, int RING = 4, int MINW = 1, int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64, MINW) void conv_igemm_kernel(ConvArgs a) {
    constexpr bool DMA = LD != 0;
    constexpr int NTHR = WM * WN * 64;                 // one wave per (TM*32) x (TN*32) sub-tile
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int A_CH = BM * 4 / NTHR;                 // 16-B chunks of the A tile per thread
    static_assert(BM * 4 % NTHR == 0, "the A tile must split evenly over the threads");
    constexpr int B_CH = (BN * 4 + NTHR - 1) / NTHR;         // (BN = 32: only threads < 128 load)
    constexpr int ST = DMA ? RING : 2;                 // LDS ring depth
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                                   // [ST][BM*64]
    char* sB = smem + ST * BM * 64;                    // [ST][BN*64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (int)((a.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int split = blockIdx.x / ntiles;
    const int id = xcd_remap(blockIdx.x - split * ntiles, ntiles);
    const int tile_n = id % tiles_n, tile_m = id / tiles_n;
    const long m0 = (long)tile_m * BM;
    const long Mrows = a.rows ? (long)*a.nrows : a.M;          // GEMM rows that exist
    if (m0 >= Mrows) return;                                    // (wave-uniform, before any barrier)
    const int n0 = tile_n * BN;

    const int taps = a.KH * a.KW, pad_h = a.KH >> 1, pad_w = a.KW >> 1;
    const int cpt = a.Cin_p / PER16;                   // chunks per tap
    const int total_chunks = taps * cpt;
    const int ksteps_all = (total_chunks + 3) >> 2;
    const int per_split = (ksteps_all + a.splits - 1) / a.splits;
    const int s_begin = split * per_split;
    const int s_end = min(ksteps_all, s_begin + per_split);
    const long Kp = (long)taps * a.Cin_p;
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ wp = (const T*)a.w;

    // ---- per-thread loader state: all chunks of a thread share the K-chunk index kc.  Register staging: kc = tid&3 and
    // the swizzle is applied when writing LDS.  DMA: LDS slot (tid&3) of row r must hold source chunk (tid&3) ^ swz(r),
    // and swz(r) = (r>>2)&3 = (tid>>4)&3 for every row this thread touches.
    // (row = (tid >> 2) + i * NTHR / 4 and NTHR / 16 is a multiple of 4, so lds_swz(row) is the same for every row this thread fills)
    const int kc = DMA ? ((tid & 3) ^ lds_swz(tid >> 2)) : (tid & 3);
    int c, ty, tx;                                     // chunk-in-tap, tap row/col of global chunk q = 4*s + kc
    {
        const int q0 = 4 * s_begin + kc, tap0 = q0 / cpt;
        c = q0 - tap0 * cpt; ty = tap0 / a.KW; tx = tap0 - ty * a.KW;
    }
    long a_pix[A_CH]; int a_oy[A_CH], a_ox[A_CH]; bool a_ok[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int row = (tid + i * NTHR) >> 2;
        const long m = m0 + row;
        a_ok[i] = m < Mrows;
        const long mm = a_ok[i] ? (a.rows ? (long)a.rows[m] : m) : 0;
        const int hw = a.H * a.W;
        const int b = (int)(mm / hw), rem = (int)(mm - (long)b * hw);
        a_oy[i] = rem / a.W; a_ox[i] = rem - a_oy[i] * a.W;
        a_pix[i] = mm;
    }
    u32x4_t ra[A_CH], rb[B_CH];

    auto load_step = [&](int s) {
        const bool tap_ok = ty < a.KH;
        const int dy = ty - pad_h, dx = tx - pad_w;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
            const bool ok = a_ok[i] && tap_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const u32x4_t*)(xp + (a_pix[i] + (long)dy * a.W + dx) * a.ldx + c * PER16);
            ra[i] = v;
        }
        const int q = 4 * s + kc;
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int idx = tid + i * NTHR;
            const int row = idx >> 2, n = n0 + row;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (idx < BN * 4 && n < a.N && q < total_chunks) v = *(const u32x4_t*)(wp + (long)n * Kp + (long)q * PER16);
            rb[i] = v;
        }
        // advance to global chunk q + 4
        c += 4;
        while (c >= cpt) { c -= cpt; if (++tx == a.KW) { tx = 0; ++ty; } }
    };
    auto dma_step = [&](int s, int buf) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const bool tap_ok = ty < a.KH;
        const int dy = ty - pad_h, dx = tx - pad_w;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
            const bool ok = a_ok[i] && tap_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const void* src = ok ? (const void*)(xp + (a_pix[i] + (long)dy * a.W + dx) * a.ldx + c * PER16) : (const void*)&g_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + buf * BM * 64 + (wv * 16 + i * (NTHR / 4)) * 64), 16, 0, 0);
        }
        const int q = 4 * s + kc;
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            if (wv * 64 + i * NTHR < BN * 4) {                       // wave-uniform (BN*4 is a multiple of 64)
                const int row = (tid + i * NTHR) >> 2, n = n0 + row;
                const void* src = (n < a.N && q < total_chunks) ? (const void*)(wp + (long)n * Kp + (long)q * PER16) : (const void*)&g_zero16;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sB + buf * BN * 64 + (wv * 16 + i * (NTHR / 4)) * 64), 16, 0, 0);
            }
        }
        c += 4;
        while (c >= cpt) { c -= cpt; if (++tx == a.KW) { tx = 0; ++ty; } }
    };
    // ---- LD == 2: buffer-descriptor DMA loader state
    constexpr unsigned OOB = 0xfffffff0u;
    constexpr int ES = (int)sizeof(T);
    unsigned voffA[A_CH], voffT[A_CH], voffB[B_CH];
    int f_ty = 0, f_tx = 0, f_cb = 0;                              // wave-uniform: tap row/col, chunk base inside the tap
    auto set_tap = [&]() {
        const int dy = f_ty - pad_h, dx = f_tx - pad_w;
        const int delta = (dy * a.W + dx) * (int)a.ldx * ES;       // byte shift of this tap (may be negative)
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int iy = a_oy[i] + dy, ix = a_ox[i] + dx;
            const bool ok = a_ok[i] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            voffT[i] = ok ? voffA[i] + (unsigned)delta : OOB;
        }
    };
    if constexpr (LD == 2) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) voffA[i] = (unsigned)((a_pix[i] * a.ldx + kc * PER16) * ES);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int n = n0 + ((tid + i * NTHR) >> 2);
            voffB[i] = n < a.N ? (unsigned)(((long)n * Kp + kc * PER16) * ES) : OOB;
        }
        const int q0 = 4 * s_begin, tap0 = q0 / cpt;
        f_cb = q0 - tap0 * cpt; f_ty = tap0 / a.KW; f_tx = tap0 - f_ty * a.KW;
        set_tap();
    }
    auto dma_fast = [&](int s, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist only in the device pass (the host pass just needs the stub)
        typedef __attribute__((address_space(3))) void* lptr_t;
        const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (int)(((a.M - 1) * a.ldx + a.Cin_p) * ES), 0x00020000);
        const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, (int)((long)a.N * Kp * ES), 0x00020000);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        if (f_cb == cpt) {                                         // next tap (wave-uniform)
            f_cb = 0;
            if (++f_tx == a.KW) { f_tx = 0; ++f_ty; }
            set_tap();
        }
        const int soffA = f_cb * PER16 * ES, soffB = s * 64;
#pragma unroll
        for (int i = 0; i < A_CH; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(sA + slot * BM * 64 + (wv * 16 + i * (NTHR / 4)) * 64), 16, voffT[i], soffA, 0, 0);
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(sB + slot * BN * 64 + (wv * 16 + i * (NTHR / 4)) * 64), 16, voffB[i], soffB, 0, 0);
        f_cb += 4;
#else
        (void)s; (void)slot;
#endif
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int row = (tid + i * NTHR) >> 2;
            *(u32x4_t*)(sA + buf * BM * 64 + lds_chunk_off(row, kc)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int idx = tid + i * NTHR;
            if (idx < BN * 4) *(u32x4_t*)(sB + buf * BN * 64 + lds_chunk_off(idx >> 2, kc)) = rb[i];
        }
    };

    const int wm = wave / WN, wn = wave % WN;
#if MTE_IGEMM_MFMA16
    static_assert(ABL == 0, "the ablation arms exist for the 32x32x16 form only");
    // a wave's (TM*32) x (TN*32) sub-tile as 2TM x 2TN blocks of 16 x 16.  Round 6: the operands are swapped (D = W X^T, as in conv_igemm8.hip and the LDS-patch
    // kernels): a lane holds 4 CONSECUTIVE output channels 4 q16 + e of pixel r16, so the tile is staged with one 8-byte LDS write per block instead of four 2-byte
    // ones that fell on the same banks, and the split-K / fp32 epilogues store 16 contiguous bytes.  Same products, same K order: bit-identical results.
    constexpr int AM = TM * 2, AN = TN * 2;
    f32x4_t acc[AM][AN];
#pragma unroll
    for (int i = 0; i < AM; ++i)
#pragma unroll
        for (int j = 0; j < AN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, q16 = lane >> 4;
    u32x4_t fa[AM], fb[AN];
    auto read_frags = [&](int slot) {
        const char* pA = sA + slot * BM * 64;
        const char* pB = sB + slot * BN * 64;
#pragma unroll
        for (int i = 0; i < AM; ++i) fa[i] = *(const u32x4_t*)(pA + lds_chunk_off((wm * AM + i) * 16 + r16, q16));
#pragma unroll
        for (int j = 0; j < AN; ++j) fb[j] = *(const u32x4_t*)(pB + lds_chunk_off((wn * AN + j) * 16 + r16, q16));
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < AM; ++i)
#pragma unroll
            for (int j = 0; j < AN; ++j) Mma16<T>::run(fb[j], fa[i], acc[i][j]);
    };
    auto compute = [&](int slot) { read_frags(slot); mfmas(); };
    // element e of block (i, j): tile row (pixel) / column (channel)
    auto acc_row = [&](int i, int e) { (void)e; return (wm * AM + i) * 16 + r16; };
    auto acc_col = [&](int j, int e) { return (wn * AN + j) * 16 + 4 * q16 + e; };
    constexpr int AE = 4;
#else
    constexpr int AM = TM, AN = TN;
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int r = lane & 31, h = lane >> 5;

    auto compute = [&](int slot) {
        const char* pA = sA + slot * BM * 64;
        const char* pB = sB + slot * BN * 64;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4_t fa[TM], fb[TN];
            if constexpr (ABL & 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { fa[i] = u32x4_t{(unsigned)lane, (unsigned)i, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(fa[i])); }
#pragma unroll
                for (int j = 0; j < TN; ++j) { fb[j] = u32x4_t{(unsigned)lane, (unsigned)j, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(fb[j])); }
            } else {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const u32x4_t*)(pA + lds_chunk_off((wm * TM + i) * 32 + r, 2 * kk + h));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const u32x4_t*)(pB + lds_chunk_off((wn * TN + j) * 32 + r, 2 * kk + h));
            }
            if constexpr (ABL & 1) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
            } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
            }
        }
    };
    auto acc_row = [&](int i, int e) { return (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h; };
    auto acc_col = [&](int j, int e) { (void)e; return (wn * TN + j) * 32 + r; };
    constexpr int AE = 16;
#endif
    if constexpr (DMA) {
        // ST-slot LDS ring filled by LDS-DMA ST-1 K-steps ahead.  vmcnt is COUNTED (the younger stages stay in
        // flight across the barrier): the only wait per K-step is for the stage about to be consumed, so HBM/L2
        // latency (1-2k cycles under load) is covered by ST-1 K-steps of MFMA work instead of one.
        static_assert((BN * 4) % NTHR == 0, "every wave must issue the same number of DMA instructions per stage");
        static_assert(ST == 3 || ST == 4 || ST == 6, "wait ladder below is written for 3, 4 or 6 slots");
        constexpr int LPS = A_CH + B_CH;
        constexpr int AHEAD = ST - 1;
        const int nst = s_end - s_begin;
#pragma unroll
        for (int p = 0; p < AHEAD; ++p)
            if (p < nst) { if constexpr (LD == 2) dma_fast(s_begin + p, p); else dma_step(s_begin + p, p); }
#if MTE_IGEMM_MFMA16
        // PING-PONG form (round 3; the 8-wave 256 x 256 tile: two wave groups = the two M halves, waves g and g + 4 share a SIMD).  In-loop
        // stamps of the one-barrier loop showed every wave of a workgroup issuing its DMA (207-267 cycles) and its fragment reads together
        // with the MFMA pipe idle, then all queueing on it (pipe busy 43-60 % of a K-step).  Here a K-step is two half-steps separated by a
        // barrier: group 0 reads the fragments of K-step k and issues its part of stage k + 3 while group 1 runs the MFMAs of K-step
        // k - 1, then they swap -- one group's loads always sit under the other group's MFMAs.  Hazards: a wave finishes its fragment
        // reads (lgkmcnt(0)) BEFORE the barrier that ends its load phase, so the slot of K-step k - 1 is free for the DMA of stage k + 3
        // from the next half-step on; every wave waits for its parts of stage k + 1 before the barrier that ends the half-step in which
        // it handled K-step k, which is at least one barrier before anyone reads that stage.
        // (the same loop on 16 waves of 64 x 64, two waves of each group per SIMD, measured no better: 66.9 vs 66.5 us on the 256 -> 256 layer)
        constexpr bool PP = LD == 2 && WM == 2 && WN == 4 && TM == 4 && TN == 2 && ST == 4;
        if constexpr (PP) {
            const int grp = __builtin_amdgcn_readfirstlane(wm);
            {   // stage 0 has landed
                if (nst >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
                else if (nst == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // both groups run the same body -- load phase, barrier, MFMA phase, barrier -- group 1 one barrier behind group 0.
            // (Issuing half of the stage's DMA behind the MFMAs instead -- the load phase is the longer one -- was tried: 66 -> 73 us on the
            //  256 -> 256 layer; a DMA instruction behind the wave's own MFMAs is slower still.)
            if (grp == 1) __builtin_amdgcn_s_barrier();
#ifdef MTE_STAMPS
            unsigned long long t_wait = 0, t_bar = 0, t_dma = 0, t_cmp = 0, t_a, t_b;
            const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#define PP_STAMP(ACC) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); ACC += t_b - t_a; t_a = t_b; __builtin_amdgcn_sched_barrier(0); }
#else
#define PP_STAMP(ACC)
#endif
            for (int k = 0; k < nst; ++k) {
#ifdef MTE_STAMPS
                t_a = __builtin_amdgcn_s_memtime();
#endif
                const int rem = nst - 2 - k;                   // stages this wave has issued behind stage k + 1: min(rem, 2)
                read_frags(k % ST);
                if (k + AHEAD < nst) dma_fast(s_begin + k + AHEAD, (k + AHEAD) % ST);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // (group 1 ends its load phase right before group 0 reads stage k + 1: its parts of that stage must be in)
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PP_STAMP(t_dma)                                // load phase: fragment reads + DMA issue + their waits
                __builtin_amdgcn_s_barrier();
                PP_STAMP(t_bar)
                mfmas();
                PP_STAMP(t_cmp)                                // MFMA issue
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                PP_STAMP(t_wait)                               // wait + second barrier
            }
#ifdef MTE_STAMPS
            if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048 && (threadIdx.x >> 6) < 4) {
                unsigned long long* o = g_igemm_stamps + ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
                o[0] = t_wait; o[1] = t_bar; o[2] = t_dma; o[3] = t_cmp; o[4] = __builtin_amdgcn_s_memtime() - t_start; o[5] = nst; o[6] = 1;
            }
#endif
#undef PP_STAMP
            if (grp == 0) __builtin_amdgcn_s_barrier();
        } else {
#endif
#ifdef MTE_STAMPS
        unsigned long long t_wait = 0, t_bar = 0, t_dma = 0, t_cmp = 0, t_a, t_b;
        const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
        int i = 0;
#if !defined(MTE_STAMPS) && !defined(MTE_IGEMM_ROLLED)
        // Round 6: steady state in groups of ST K-steps with COMPILE-TIME ring slots.  tools/loopaudit.py on the rolled loop: ~48 scalar + 5 vector instructions per
        // 16 MFMAs and K-step -- two divisions by ST for the slots (s_mul_hi), slot address arithmetic, the rem ladder -- with four waves per SIMD that is more issue
        // slots than the MFMAs leave (profiles/r05_wgrad9_steps.txt: one instruction per ~4 cycles per SIMD for all of its waves).  While a whole group and the AHEAD
        // stages it issues exist, no `rem` test is needed: the wait is the constant (AHEAD - 1) * LPS, the slots are u and (u + AHEAD) % ST.  The rolled loop
        // below finishes the last < ST + AHEAD K-steps.  Same K-steps in the same order: bit-identical.  Not for the four-wave 128 x 128 / 128 x 64 forms: unrolled,
        // their accumulators + fragments leave three waves per SIMD for two and they run 14-28 % slower (serial traces r06_v1 vs r06_v2: 126.7 -> 144.5 us / 4 calls);
        // the others gain 2-11 % (1346.7 -> 1290.8 us / 17 calls of <4,2,2,2>, 226.0 -> 201.3 us / 5 calls of <4,2,2,2, RING 4>, 173.4 -> 154.5 us of <2,3,3,1>).
        if constexpr (LD == 2 && ABL == 0 && WM * WN >= 6) {
            for (; i + ST + AHEAD <= nst; i += ST) {
#pragma unroll
                for (int u = 0; u < ST; ++u) {
                    // lgkmcnt(0): unrolled, the compiler leaves the last fragment reads of compute(u - 1) in flight across this barrier (their MFMAs follow it), and
                    // the slot they read is the one re-staged right behind it.  A ds_read that has not RETIRED before the barrier is not ordered against another
                    // wave's LDS-DMA into the same bytes: without this wait the eval forward of the benchmark network differed from call to call in 12-23 of 23
                    // calls (one sample of eight, |d depth| 0.2-1.4; profiles/r06_igemm_unroll_race.txt) while every single-launch parity test passed.
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * LPS) : "memory");
                    __builtin_amdgcn_s_barrier();
                    dma_fast(s_begin + i + u + AHEAD, (u + AHEAD) % ST);
                    compute(u);
                }
            }
        }
#endif
        for (; i < nst; ++i) {
#ifdef MTE_STAMPS
            t_a = __builtin_amdgcn_s_memtime();
#endif
            const int rem = nst - 1 - i;                   // stages issued after stage i and possibly still in flight: min(rem, AHEAD - 1)
            if constexpr (ST == 6) {
                if (rem >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * LPS) : "memory");
                else if (rem == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
                else if (rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if constexpr (ST == 4) {
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#ifdef MTE_STAMPS
            t_b = __builtin_amdgcn_s_memtime(); t_wait += t_b - t_a; t_a = t_b;
#endif
            __builtin_amdgcn_s_barrier();
#ifdef MTE_STAMPS
            t_b = __builtin_amdgcn_s_memtime(); t_bar += t_b - t_a; t_a = t_b;
#endif
            if constexpr (!(ABL & 2))
            if (i + AHEAD < nst) { if constexpr (LD == 2) dma_fast(s_begin + i + AHEAD, (i + AHEAD) % ST); else dma_step(s_begin + i + AHEAD, (i + AHEAD) % ST); }
#ifdef MTE_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            t_b = __builtin_amdgcn_s_memtime(); t_dma += t_b - t_a; t_a = t_b;
            __builtin_amdgcn_sched_barrier(0);
#endif
            compute(i % ST);
#ifdef MTE_STAMPS
            __builtin_amdgcn_sched_barrier(0);
            t_b = __builtin_amdgcn_s_memtime(); t_cmp += t_b - t_a;
#endif
        }
#ifdef MTE_STAMPS
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {
            unsigned long long* o = g_igemm_stamps + ((long)blockIdx.x * 4 + (threadIdx.x >> 6) % 4) * 8;   // first four waves of a workgroup
            if ((threadIdx.x >> 6) < 4) { o[0] = t_wait; o[1] = t_bar; o[2] = t_dma; o[3] = t_cmp; o[4] = __builtin_amdgcn_s_memtime() - t_start; o[5] = nst; }
        }
#endif
#if MTE_IGEMM_MFMA16
        }   // !PP
#endif
    } else {
        if (s_begin < s_end) {
            load_step(s_begin); store_step(0);
            __syncthreads();
        }
        for (int s = s_begin; s < s_end; ++s) {
            const int buf = (s - s_begin) & 1;
            if (s + 1 < s_end) load_step(s + 1);
            compute(buf);
            if (s + 1 < s_end) store_step(buf ^ 1);
            __syncthreads();
        }
    }

    if constexpr (DMA) {
        if (a.splits == 1 && !a.out_f32) {
            // ---- epilogue through LDS: [BM][BN] tile in T, then 16-byte row-contiguous stores
            constexpr int ES = (int)sizeof(T);
            static_assert(BM * BN * ES <= ST * (BM + BN) * 64, "output tile must fit the ring");
            __syncthreads();                              // every wave is done reading the ring
            // staged image: 16-byte chunk c of tile row r at slot (c + r) % CPR_ -- the 16 lanes of a block write 16 different rows at the same chunk, the rotation
            // spreads them over the banks (the read side below undoes it)
            constexpr int CPR_ = BN / PER16;
#if MTE_IGEMM_MFMA16
#pragma unroll
            for (int j = 0; j < AN; ++j) {
                const int col0 = acc_col(j, 0);
                float bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (a.bias && n0 + col0 + e < a.N) ? a.bias[n0 + col0 + e] : 0.f;
#pragma unroll
                for (int i = 0; i < AM; ++i) {
                    const int row = acc_row(i, 0);
                    char* dst = smem + (row * BN) * ES + (((col0 / PER16) + row) % CPR_) * 16 + (col0 % PER16) * ES;
                    if constexpr (sizeof(T) == 2) {
                        uint2 pk;
                        pk.x = pack2bf(acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]);
                        pk.y = pack2bf(acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]);
                        *(uint2*)dst = pk;
                    } else {
                        *(f32x4_t*)dst = f32x4_t{acc[i][j][0] + bv[0], acc[i][j][1] + bv[1], acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]};
                    }
                }
            }
#else
#pragma unroll
            for (int j = 0; j < AN; ++j) {
                const int col = acc_col(j, 0);
                const int n = n0 + col;
                const float bv = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < AM; ++i)
#pragma unroll
                    for (int e = 0; e < AE; ++e) {
                        const int row = acc_row(i, e);
                        Elem<T>::st((T*)(smem + (row * BN) * ES + (((col / PER16) + row) % CPR_) * 16 + (col % PER16) * ES), acc[i][j][e] + bv);
                    }
            }
#endif
            __syncthreads();
            constexpr int CPR = BN / PER16;                // 16-B chunks per tile row
            const int cvalid = (a.N - n0) / PER16;         // chunks of this tile inside N (N % 8 == 0)
            const int cc = tid % CPR;
            if constexpr (sizeof(T) == 2 && BN % 32 == 0) {
                if (a.unshuffle_c) {
                    // work item = (tile row, group c8 of 8 channels = 32 packed depths = the four staged chunks 4 c8 .. 4 c8 + 3): chunk j holds channels 2j, 2j + 1 as
                    // [c][s] -- element 4 e + s = channel 2j + e, sub-pixel s.  Four ds_read_b128, then for every sub-pixel s the 8 channels are 4 byte permutes
                    // (v_perm_b32: the two halves e = 0 / 1 of chunk j from dwords s / 2 and 2 + s / 2) and leave as ONE 16-byte store into the un-shuffled tensor.
                    // (First version: a thread per (row, s, c8) with eight 2-byte LDS reads per store -- +40 us per launch: gpurun_out/r06_unshuffle.txt.)
                    constexpr int C8 = BN / 32;
                    const int hw = a.H * a.W;
                    for (int idx = tid; idx < BM * C8; idx += NTHR) {
                        const int row = idx / C8, c8l = idx - row * C8;
                        const int c0 = (n0 >> 2) + c8l * 8;
                        const long m = m0 + row;
                        if (m < Mrows && c0 + 8 <= a.unshuffle_c) {
                            const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
                            const int h = rem / a.W, w = rem - h * a.W;
                            const char* rowp = smem + (row * BN) * ES;
                            u32x4_t q[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) q[j] = *(const u32x4_t*)(rowp + ((4 * c8l + j + row) % CPR) * 16);
                            T* dst0 = (T*)a.y + ((((long)b * 2 * a.H + 2 * h) * (2 * a.W) + 2 * w) * a.ldy + c0);
#pragma unroll
                            for (int s_ = 0; s_ < 4; ++s_) {
                                u32x4_t c;
#pragma unroll
                                for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_perm(q[j][2 + (s_ >> 1)], q[j][s_ >> 1], (s_ & 1) ? 0x07060302u : 0x05040100u);
                                T* dst = dst0 + ((long)(s_ >> 1) * (2 * a.W) + (s_ & 1)) * a.ldy;
                                if (a.accum) {
                                    float vn[PER16], vo[PER16];
                                    unpack16<T>(c, vn);
                                    unpack16<T>(*(const u32x4_t*)dst, vo);
#pragma unroll
                                    for (int k = 0; k < PER16; ++k) vn[k] += vo[k];
                                    c = pack16<T>(vn);
                                }
                                *(u32x4_t*)dst = c;
                            }
                        }
                    }
                    return;
                }
            }
#pragma unroll
            for (int it = 0; it < BM * CPR / NTHR; ++it) {
                const int row = tid / CPR + it * (NTHR / CPR);
                const long m = m0 + row;
                if (m < Mrows && cc < cvalid) {
                    const long pm = a.rows ? (long)a.rows[m] : m;      // output pixel of this tile row
                    u32x4_t c = *(const u32x4_t*)(smem + (row * BN) * ES + ((cc + row) % CPR) * 16);
                    if (a.accum) {
                        float vn[PER16], vo[PER16];
                        unpack16<T>(c, vn);
                        unpack16<T>(*(const u32x4_t*)((const T*)a.y + pm * a.ldy + n0 + cc * PER16), vo);
#pragma unroll
                        for (int k = 0; k < PER16; ++k) vn[k] += vo[k];
                        c = pack16<T>(vn);
                    }
                    *(u32x4_t*)((T*)a.y + pm * a.ldy + n0 + cc * PER16) = c;
                }
            }
            return;
        }
    }

    // ---- epilogue: D[row][col]: col = lane&31 (channel n), row = (e&3) + 8*(e>>2) + 4*h (pixel m)
#pragma unroll
    for (int j = 0; j < AN; ++j) {
#pragma unroll
        for (int i = 0; i < AM; ++i) {
#pragma unroll
            for (int e = 0; e < AE; ++e) {
                const int n = n0 + acc_col(j, e);
                if (n >= a.N) continue;
                const float bv = a.bias ? a.bias[n] : 0.f;
                const long m = m0 + acc_row(i, e);
                if (m < Mrows) {
                    const float v = acc[i][j][e] + bv;
                    const long pm = a.rows ? (long)a.rows[m] : m;
                    if (a.splits > 1) a.ws[((long)split * a.M + m) * a.N + n] = acc[i][j][e];
                    else if (a.out_f32) ((float*)a.y)[pm * a.ldy + n] = a.accum ? v + ((float*)a.y)[pm * a.ldy + n] : v;
                    else Elem<T>::st((T*)a.y + pm * a.ldy + n, a.accum ? v + Elem<T>::ld((const T*)a.y + pm * a.ldy + n) : v);
                }
            }
        }
    }
}

int g_igemm_dma = 1;                                 // development knob (mte_debug_set(0, v))

// y = T(sum_s ws[s] + bias) after a split-K launch: the slabs are added in split order (fixed), rounded once
template <typename T>
__global__ void splitk_finish_kernel(const float* __restrict__ ws, int splits, const float* __restrict__ bias, T* y, long ldy, long M, int N, int accum) {
    const long n4 = M * N / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long m = (i * 4) / N; const int n = (int)((i * 4) % N);
        f32x4_t v = ((const f32x4_t*)ws)[i];
        for (int s = 1; s < splits; ++s) v += ((const f32x4_t*)ws)[(long)s * n4 + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) Elem<T>::st(y + m * ldy + n + k, v[k] + (bias ? bias[n + k] : 0.f) + (accum ? Elem<T>::ld(y + m * ldy + n + k) : 0.f));
    }
}

// split-K factor for small-M / huge-K layers (pack5.conv: 120 tiles for 256 CUs); 1 = no split
inline int choose_splits(long tiles, int ksteps, long M, int N, long ws_elems, int nthr = 256) {
    const long want = 768L * 256 / nthr;               // ~3 four-wave workgroups per CU, or their equivalent in larger ones
    if (tiles >= want / 2 || ws_elems < M * N || N % 4 != 0) return 1;
    long s = (want + tiles - 1) / tiles;
    const long max_s = ksteps / 16;                    // keep >= 16 K-steps (1 KiB of K per row) per split
    if (s > max_s) s = max_s;
    if (s > ws_elems / (M * N)) s = ws_elems / (M * N);   // one [M][N] slab per split
    return (int)(s < 1 ? 1 : s);
}

int g_igemm_pair_ksteps = 72;                        // development knob (mte_debug_set(19, v))
#ifdef MTE_DEV
int g_igemm_ablate = 0;                              // development knob (mte_debug_set(17, v)): main-loop ablation, see ABL
#endif
int g_igemm_ring6 = 0;                               // development knob (mte_debug_set(15, v)) for the 8-wave 256 x 128 tile: 1 = 6-slot ring, 3 = 3-slot ring with two workgroups per CU

template <typename T, int WM, int WN, int TM, int TN>
int launch_igemm(ConvArgs a, long ws_elems, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NTHR = WM * WN * 64;
    const long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int ksteps = (a.KH * a.KW * (a.Cin_p / Elem<T>::PER16) + 3) / 4;
    a.splits = (a.ws && !a.out_f32 && !a.rows) ? choose_splits(tiles, ksteps, a.M, a.N, ws_elems, NTHR) : 1;
    if (a.unshuffle_c && (a.splits != 1 || a.out_f32 || a.rows || sizeof(T) != 2 || BN % 32 != 0 || (BN * 4) % NTHR != 0 || !g_igemm_dma)) return MTE_ERR_UNSUPPORTED;
    if constexpr ((BN * 4) % NTHR == 0) {
        if (g_igemm_dma) {
            const size_t lds4 = 4 * (BM + BN) * 64;
            const long es = (long)sizeof(T);
            const bool fast = g_igemm_dma == 1 && a.Cin_p % (4 * Elem<T>::PER16) == 0 &&
                              ((a.M - 1) * a.ldx + a.Cin_p) * es < 0x7ff00000L && (long)a.N * a.KH * a.KW * a.Cin_p * es < 0x7ff00000L;
#if defined(MTE_DEV) && !MTE_IGEMM_MFMA16
            if constexpr (sizeof(T) == 2 && ((BM == 256 && BN == 128) || (BM == 128 && BN == 128) || (BM == 256 && BN == 256))) {
                if (fast && g_igemm_ablate) {
                    const dim3 g((unsigned)(tiles * a.splits)), b(NTHR);
                    switch (g_igemm_ablate) {
                    case 1: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 1>), g, b, lds4, st, a); break;
                    case 2: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 2>), g, b, lds4, st, a); break;
                    case 3: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 3>), g, b, lds4, st, a); break;
                    case 4: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 4>), g, b, lds4, st, a); break;
                    case 5: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 5>), g, b, lds4, st, a); break;
                    case 6: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 6>), g, b, lds4, st, a); break;
                    default: hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 4, 1, 7>), g, b, lds4, st, a); break;
                    }
                    goto launched;
                }
            }
#endif
            if constexpr (sizeof(T) == 2 && BM == 256 && BN == 128) {
                // two 74 KB workgroups per CU (3-slot ring): always for solo launches; beside the weight-gradient stream only for short
                // reductions over several rounds of tiles, where the prologue / epilogue share is largest (same-box step 30.46 -> 30.30 ms;
                // for every launch it costs the step 0.2 ms)
                const bool pair = a.solo || (ksteps <= g_igemm_pair_ksteps && tiles >= 512);
                if (fast && (g_igemm_ring6 == 3 || (g_igemm_ring6 == 0 && pair))) {
                    constexpr size_t lds3 = 3 * (BM + BN) * 64;
                    static bool attr3 = false;
                    if (!attr3) {
                        if (hipFuncSetAttribute((const void*)conv_igemm_kernel<T, WM, WN, TM, TN, 2, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess) return MTE_ERR_LAUNCH;
                        attr3 = true;
                    }
                    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 3, 4>), dim3((unsigned)(tiles * a.splits)), dim3(NTHR), lds3, st, a);
                    goto launched;
                }
                if (fast && g_igemm_ring6 == 1) {
                    constexpr size_t lds6 = 6 * (BM + BN) * 64;
                    static bool attr = false;
                    if (!attr) {
                        if (hipFuncSetAttribute((const void*)conv_igemm_kernel<T, WM, WN, TM, TN, 2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds6) != hipSuccess) return MTE_ERR_LAUNCH;
                        attr = true;
                    }
                    hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2, 6>), dim3((unsigned)(tiles * a.splits)), dim3(NTHR), lds6, st, a);
                    goto launched;
                }
            }
            if (fast) hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 2>), dim3((unsigned)(tiles * a.splits)), dim3(NTHR), lds4, st, a);
            else hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 1>), dim3((unsigned)(tiles * a.splits)), dim3(NTHR), lds4, st, a);
            goto launched;
        }
    }
    if constexpr (NTHR == 256)
        hipLaunchKernelGGL((conv_igemm_kernel<T, WM, WN, TM, TN, 0>), dim3((unsigned)(tiles * a.splits)), dim3(256), 2 * (BM + BN) * 64, st, a);
    else return MTE_ERR_UNSUPPORTED;
launched:
    if (a.splits > 1) {
        long g = (a.M * a.N / 4 + 255) / 256; if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3((unsigned)g), dim3(256), 0, st, a.ws, a.splits, a.bias, (T*)a.y, a.ldy, a.M, a.N, a.accum);
    }
    return mte_check_launch();
}

int g_igemm_big_min_tiles = 224;
// Default 51 = 19 | 32: TAP-major.  The slice-major order of round 5 (ConvArgs.kslice) cuts the implicit GEMM's fetches beyond L2 (the nine tap sweeps
// of a 64-channel slice re-use the lines of the first), but those fetches were Infinity-Cache hits, not HBM reads: same box, the training step takes
// 24.80 ms with it and 24.77 ms without, and in isolation the kernels are 5 % SLOWER (a tap change, i.e. new lane offsets, every K-tile): 6.74 against
// 6.43 ms over the step's launches (profiles/r05_igemm8_korder.txt).  It stays selectable: knob 23 without bit 5, or -DMTE_IGEMM8_DEFAULT=19.
#ifndef MTE_IGEMM8_DEFAULT
#define MTE_IGEMM8_DEFAULT 51
#endif
int g_igemm8 = MTE_IGEMM8_DEFAULT;                                   // development knob (mte_debug_set(23, v)): bit 0 = 8-phase 256 x 256 kernel, bit 1 = its 256 x 128 form, bit 2 every eligible launch, bit 5 (32) tap-major K order (the default; without it: slice-major where Cin_p % 64 == 0); bits 3 / 4 belonged to the tile-walking form, removed in round 5
int g_igemm_n32_dma = 1;                              // development knob (mte_debug_set(32, v)): N <= 32 on the two-wave LDS-DMA form
int g_igemm8_split_bn128 = 64;                        // development knob (mte_debug_set(29, v)): see the split-K rule of the 8-phase kernels
int g_igemm8_min_tiles = 200;                        // development knob (mte_debug_set(24, v))
int g_igemm_pp = 1;                                  // development knob (mte_debug_set(21, v)): 0 = the 16-wave one-barrier loop on the 256 x 256 tile
int g_igemm_big = 3;                                 // development knob (mte_debug_set(6, v)): 0 128x128 only, 1 + 256x128, 2 + 256x256, 3 + 192x96

template <typename T> int dispatch_igemm(const ConvArgs& a, long ws_elems, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        // 256 x 128 tile, 8 waves: 24 KB of operands per K-step feed twice the MFMA work of a 128 x 128 tile (16 KB).  The
        // 4-wave kernel runs at ~14 TB/s of L2->LDS traffic with three stages in flight -- the latency-bandwidth product,
        // not the MFMA pipe, bounds it -- so fewer bytes per flop is the lever.  Needs the buffer-DMA loader and enough
        // tiles to cover the CUs.
        const long tiles_big = ((a.M + 255) / 256) * ((a.N + 127) / 128);
        const bool dma_ok = g_igemm_dma == 1 && a.Cin_p % 32 == 0 && ((a.M - 1) * a.ldx + a.Cin_p) * 2 < 0x7ff00000L &&
                            (long)a.N * a.KH * a.KW * a.Cin_p * 2 < 0x7ff00000L;
        // (odd widths -- 72 / 104 / 200 input channels of the decoder concats as data-gradient N -- take the tile that covers
        //  them in ONE column block: the padded columns cost the same MFMA work as two narrower blocks, A is read once)
        const long n256 = (a.N + 255) / 256;
        const long t256 = ((a.M + 255) / 256) * n256;
        const int ksteps = a.KH * a.KW * (a.Cin_p / 32);
        // few tiles but a huge reduction (pack4/pack5.conv: K = 9 x 4096 / 8192): the big tiles keep their bytes-per-flop
        // advantage when the K range is split over workgroups (fp32 atomics into the workspace, then the finish kernel)
        const bool can_split = a.ws && ws_elems >= 2 * a.M * a.N && a.N % 4 == 0;
        long smax = can_split ? ws_elems / (a.M * a.N) : 1;                    // one [M][N] slab per split
        if (smax > 8) smax = 8;
        const long reach256 = t256 * (can_split ? (ksteps / 16 < smax ? (ksteps / 16 > 0 ? ksteps / 16 : 1) : smax) : 1);
        // ---- round 4: the 8-phase kernels (conv_igemm8.hip) take every launch the 256-row tiles took
        if (g_igemm8 && dma_ok && !a.out_f32 && !a.rows && a.N > 64 && !a.unshuffle_c) {
            const bool wide = a.N > 128 && (a.N % 256 == 0 || a.N % 256 > 128);
            const int nkt = (ksteps + 1) / 2;                                                      // K-tiles of 64
            int bn = 0, splits = 1;
            // Same-box A/B against the older tile forms over the training step's shapes (tools/igemm8_check.py bench; profiles/r04_igemm8_ab.txt):
            // one workgroup per CU, so a launch of SEVERAL rounds of tiles pays prologue + epilogue (~ 6 K-tiles' worth) per round, where the
            // 256 x 128 kernel it replaces runs two workgroups per CU: short reductions over many tiles stay with the older forms.
            if (g_igemm8 & 4) {                                                                    // (development: every eligible launch)
                if (wide && t256 >= g_igemm8_min_tiles) bn = 256;
                else if (tiles_big >= g_igemm8_min_tiles) bn = 128;
            } else if (wide && (g_igemm8 & 1) && t256 >= g_igemm8_min_tiles) {
                if (nkt >= (t256 <= 256 ? 18 : 36)) bn = 256;
            } else if ((g_igemm8 & 2) && tiles_big >= g_igemm8_min_tiles && tiles_big <= 256 && nkt >= 36) bn = 128;
            if (!bn && wide && (g_igemm8 & 1) && can_split && t256 < 128 && ksteps >= 32) {        // few tiles, long reduction: split K
                bn = 256;
                long tsp = t256;
                // Round 6: very few tiles AND a short reduction (the 512-channel 12x40 / 24x80 layers: K = 4608) -> 256 x 128 tiles with half the K splits.  Such a
                // launch is dominated by its fp32 slabs (8 x 7.9 MB written and read back for a 3.9 MB result at 12x40); half the slabs: 41.7 -> 35.2 us forward,
                // 39.7 -> 33.4 data gradient at 512 -> 512 @12x40, 55.9 -> 52.4 at 512 -> 256 @24x80.  With a long reduction (pack4 / pack5.conv: K = 36,864 /
                // 73,728) the slabs do not matter and the narrower tile loses 15 % (265 -> 305 us): profiles/r06_lowres_split.txt.  Knob 29 = tile bound (0: off).
                if (t256 <= g_igemm8_split_bn128 && ksteps <= 288) { bn = 128; tsp = tiles_big; }
                long sp = 256 / tsp;                                                               // one round of workgroups
                if (sp > smax) sp = smax;
                if (sp > ksteps / 16) sp = ksteps / 16;                                            // >= 8 K-tiles per split
                const int per = (int)((ksteps + sp - 1) / (sp < 1 ? 1 : sp));
                splits = (ksteps + per - 1) / per;                                                 // (no empty split)
            }
            if (bn) {
                ConvArgs b = a;
                b.splits = splits;
                // round 5: 64-channel slices outer, taps inner where asked for and the channels allow it (ConvArgs.kslice; knob 23 bit 5 = tap-major, the order of
                // every other tile form and the default -- see g_igemm8).  The tile-walking (persistent) form of round 4 is gone: it was 15-20 % slower per
                // launch than one workgroup per tile and no launch used it.
                b.kslice = (!(g_igemm8 & 32) && a.Cin_p % 64 == 0) ? 1 : 0;
                const int rc = igemm8_launch(b, bn, st);
                if (rc == MTE_OK) {
                    if (splits > 1) {
                        long g = (a.M * a.N / 4 + 255) / 256; if (g > 4096) g = 4096;
                        hipLaunchKernelGGL(splitk_finish_kernel<T>, dim3((unsigned)g), dim3(256), 0, st, a.ws, splits, a.bias, (T*)a.y, a.ldy, a.M, a.N, a.accum);
                    }
                    return mte_check_launch();
                }
                if (rc != MTE_ERR_UNSUPPORTED) return rc;
            }
        }
        if (g_igemm_big >= 2 && dma_ok && !a.out_f32 && a.N > 128 && (a.N % 256 == 0 || a.N % 256 > 128) &&
            (t256 >= g_igemm_big_min_tiles || (can_split && t256 < 96 && reach256 >= 160))) {   // (96: below it choose_splits does split)
            // enough tiles without a K split: 8 waves of 128 x 64 in the ping-pong loop (same-box A/B per layer: 256 -> 256 3x3 @48x160
            // 70.2 -> 66 us, 384 -> 256 106 -> 95-101, 5x5 64 -> 256 @96x320 226 -> 212, 128 -> 512 @48x160 203 -> 189); the split-K
            // launches (few tiles, short per-split reductions) lose with it and keep the 16-wave one-barrier loop
            if (g_igemm_pp && t256 >= g_igemm_big_min_tiles) return launch_igemm<T, 2, 4, 4, 2>(a, 0, st);
            return launch_igemm<T, 4, 4, 2, 2>(a, t256 >= g_igemm_big_min_tiles ? 0 : ws_elems, st);   // 256 x 256, 16 waves
        }
        // 65..96 columns (the 72-channel decoder concat as data-gradient N): a 192 x 96 tile of 6 waves wastes a quarter of
        // the MFMA work instead of the 44 % a 128-wide tile does
        if (g_igemm_big >= 3 && dma_ok && !a.out_f32 && a.N > 64 && a.N <= 96 && ((a.M + 191) / 192) >= g_igemm_big_min_tiles)
            return launch_igemm<T, 2, 3, 3, 1>(a, 0, st);
        if (g_igemm_big && dma_ok && !a.out_f32 && a.N > 64 && tiles_big >= g_igemm_big_min_tiles)
            return launch_igemm<T, 4, 2, 2, 2>(a, 0, st);
    }
    if (a.N <= 32) {
        // round 6: two waves of 64 x 32 where the LDS-DMA loader applies (its B stage needs (BN * 4) % threads == 0, which the four-wave 128 x 32 form misses: that
        // one stages through registers) -- the 32-output band convolutions of the folded pack layers (K = 25 x 512)
        if constexpr (sizeof(T) == 2) {
            if (g_igemm_n32_dma && g_igemm_dma == 1 && a.Cin_p % 32 == 0 && !a.out_f32 && ((a.M - 1) * a.ldx + a.Cin_p) * 2 < 0x7ff00000L &&
                (long)a.N * a.KH * a.KW * a.Cin_p * 2 < 0x7ff00000L)
                return launch_igemm<T, 2, 1, 2, 1>(a, ws_elems, st);
        }
        return launch_igemm<T, 4, 1, 1, 1>(a, ws_elems, st);       // 128 x 32
    }
    if (a.N <= 64) return launch_igemm<T, 2, 2, 2, 1>(a, ws_elems, st);        // 128 x 64
    return launch_igemm<T, 2, 2, 2, 2>(a, ws_elems, st);                      // 128 x 128
}

// =====================================================================================================
// wgrad: dw_stage[n][tap][c] (+)= sum_m dy[m][n] * x[pix(m)+tap][c]
// One workgroup = (n tile, one tap, c tile, pixel split).  The reduction index (pixel) is the slow memory
// dimension of both NHWC operands, so tiles are staged as [32 pixels][channels] and the MFMA operands
// (8 consecutive k per lane) are fetched with the transposing LDS read ds_read_b64_tr_b16 (bf16) or with
// plain 4-byte reads (f32 mode, one k per lane per MFMA).  Row strides are padded to an odd multiple of
// 64 B so the four pixel rows a transposing read touches fall in different bank groups.
// =====================================================================================================
struct WgradArgs {
    const void* x; long ldx;
    const void* dy; long ldy;
    float* dw;                      // [N][taps][Cin_p] fp32 staging (zeroed by caller when splits > 1)
    int B, H, W, Cin_p, N, KH, KW;
    long M;
    int splits, blocks_per_split;   // pixel blocks of 32
    int tiles_c, tiles_n;
    long part_stride;               // > 0: pixel split s stores its partial gradient at dw + s * part_stride (no atomics)
};

template <int BYTES> __device__ __forceinline__ int padded_row(void) {
    return (BYTES % 128 == 64) ? BYTES : BYTES + 64;
}

// FL = true: pixel blocks are 32-pixel segments of ONE image row, tiles are fetched with buffer loads whose per-lane byte
// offset is a kernel-lifetime constant and whose row base is a wave-uniform SGPR offset (same idea as igemm LD = 2);
// FL = false: generic flattened-pixel loader (any tensor size).
template <typename T, int WNO, int WC, int TNO, int TC, bool FL>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int BNO = WNO * TNO * 32, BC = WC * TC * 32;          // cout x cin tile
    constexpr int PER16 = Elem<T>::PER16, ES = (int)sizeof(T);
    constexpr int RS_Y = (BNO * ES % 128 == 64) ? BNO * ES : BNO * ES + 64;   // LDS row strides (bytes)
    constexpr int RS_X = (BC * ES % 128 == 64) ? BC * ES : BC * ES + 64;
    constexpr int KB = 32;                                             // pixels per step
    constexpr int YCH = KB * BNO / PER16, XCH = KB * BC / PER16;       // chunks per tile
    constexpr int YCT = (YCH + 255) / 256, XCT = (XCH + 255) / 256;
    constexpr int YCPR = BNO / PER16, XCPR = BC / PER16;               // chunks per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sY = smem;                                    // [2][KB*RS_Y]
    char* sX = smem + 2 * KB * RS_Y;                    // [2][KB*RS_X]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int id = blockIdx.x;
    const int taps = a.KH * a.KW;
    const int tile_c = id % a.tiles_c; id /= a.tiles_c;
    const int tap = id % taps; id /= taps;
    const int tile_n = id % a.tiles_n; id /= a.tiles_n;
    const int split = id;
    const int ty = tap / a.KW, tx = tap % a.KW;
    const int dyo = ty - (a.KH >> 1), dxo = tx - (a.KW >> 1);
    const int n0 = tile_n * BNO, c0 = tile_c * BC;
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ yp = (const T*)a.dy;

    const int bpr = (a.W + KB - 1) / KB;                 // FL: blocks per image row
    const long nblk_total = FL ? (long)a.B * a.H * bpr : (a.M + KB - 1) / KB;
    const long blk0 = (long)split * a.blocks_per_split;
    long blk1 = blk0 + a.blocks_per_split; if (blk1 > nblk_total) blk1 = nblk_total;

    // loader state: per chunk slot the (row, chunk-in-row) is fixed; pixel coords advance by KB per step
    int y_row[YCT], y_cc[YCT], x_row[XCT], x_cc[XCT];
    int x_b[XCT], x_oy[XCT], x_ox[XCT];
#pragma unroll
    for (int i = 0; i < YCT; ++i) { const int idx = tid + i * 256; y_row[i] = idx / YCPR; y_cc[i] = idx % YCPR; }
#pragma unroll
    for (int i = 0; i < XCT; ++i) {
        const int idx = tid + i * 256; x_row[i] = idx / XCPR; x_cc[i] = idx % XCPR;
        const long m = blk0 * KB + x_row[i];
        const int hw = a.H * a.W;
        const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
        x_b[i] = b; x_oy[i] = rem / a.W; x_ox[i] = rem - x_oy[i] * a.W;
    }
    u32x4_t ry[YCT], rx[XCT];
    auto load_step = [&](long blk) {
#pragma unroll
        for (int i = 0; i < YCT; ++i) {
            const long m = blk * KB + y_row[i];
            const int n = n0 + y_cc[i] * PER16;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if ((YCH % 256 == 0 || tid + i * 256 < YCH) && m < a.M && n < a.N) v = *(const u32x4_t*)(yp + m * a.ldy + n);
            ry[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XCT; ++i) {
            const int cc = c0 + x_cc[i] * PER16;
            const int iy = x_oy[i] + dyo, ix = x_ox[i] + dxo;
            const bool ok = (XCH % 256 == 0 || tid + i * 256 < XCH) && x_b[i] < a.B && cc < a.Cin_p &&
                            (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const u32x4_t*)(xp + (((long)x_b[i] * a.H + iy) * a.W + ix) * a.ldx + cc);
            rx[i] = v;
            // advance this slot's pixel by KB
            x_ox[i] += KB;
            while (x_ox[i] >= a.W) { x_ox[i] -= a.W; if (++x_oy[i] == a.H) { x_oy[i] = 0; ++x_b[i]; } }
        }
    };
    // ---- FL loader state: block (b, y, xb) is wave-uniform and advances incrementally
    constexpr unsigned OOB = 0xfffffff0u;
    const int padw = a.KW >> 1;
    unsigned voffY[YCT], voffX[XCT];
    int f_xb = 0, f_y = 0, f_b = 0;
    if constexpr (FL) {
        f_xb = (int)(blk0 % bpr); const long t = blk0 / bpr; f_y = (int)(t % a.H); f_b = (int)(t / a.H);
#pragma unroll
        for (int i = 0; i < YCT; ++i) {
            const int n = n0 + y_cc[i] * PER16;
            voffY[i] = ((YCH % 256 == 0 || tid + i * 256 < YCH) && n < a.N) ? (unsigned)((y_row[i] * a.ldy + n) * ES) : OOB;
        }
#pragma unroll
        for (int i = 0; i < XCT; ++i) {
            const int cc = c0 + x_cc[i] * PER16;
            voffX[i] = ((XCH % 256 == 0 || tid + i * 256 < XCH) && cc < a.Cin_p) ? (unsigned)(((x_row[i] + dxo + padw) * a.ldx + cc) * ES) : OOB;
        }
    }
    auto load_fast = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)yp, 0, (int)(((a.M - 1) * a.ldy + a.N) * ES), 0x00020000);
        const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(xp - (long)padw * a.ldx), 0, (int)(((a.M - 1 + padw) * a.ldx + a.Cin_p) * ES), 0x00020000);
        const int x0 = f_xb * KB, rem = a.W - x0;
        const int yy = f_y + dyo;
        const bool row_ok = (unsigned)yy < (unsigned)a.H;
        const int soffY = (int)((((long)f_b * a.H + f_y) * a.W + x0) * a.ldy * ES);
        const int soffX = (int)((((long)f_b * a.H + yy) * a.W + x0) * a.ldx * ES);
#pragma unroll
        for (int i = 0; i < YCT; ++i)
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, y_row[i] < rem ? voffY[i] : OOB, soffY, 0);
#pragma unroll
        for (int i = 0; i < XCT; ++i) {
            const bool ok = row_ok && (unsigned)(x0 + x_row[i] + dxo) < (unsigned)a.W;
            rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? voffX[i] : OOB, row_ok ? soffX : 0, 0);
        }
        if (++f_xb == bpr) { f_xb = 0; if (++f_y == a.H) { f_y = 0; ++f_b; } }
#endif
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < YCT; ++i)
            if (YCH % 256 == 0 || tid + i * 256 < YCH) *(u32x4_t*)(sY + buf * KB * RS_Y + y_row[i] * RS_Y + y_cc[i] * 16) = ry[i];
#pragma unroll
        for (int i = 0; i < XCT; ++i)
            if (XCH % 256 == 0 || tid + i * 256 < XCH) *(u32x4_t*)(sX + buf * KB * RS_X + x_row[i] * RS_X + x_cc[i] * 16) = rx[i];
    };

    f32x16_t acc[TNO][TC];
#pragma unroll
    for (int i = 0; i < TNO; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wno = wave / WC, wc = wave % WC;
    const int r = lane & 31, h = lane >> 5;

    if (blk0 < blk1) {
        if constexpr (FL) load_fast(); else load_step(blk0);
        store_step(0);
        __syncthreads();
        for (long blk = blk0; blk < blk1; ++blk) {
            const int buf = (int)((blk - blk0) & 1);
            if (blk + 1 < blk1) { if constexpr (FL) load_fast(); else load_step(blk + 1); }
            const char* pY = sY + buf * KB * RS_Y;
            const char* pX = sX + buf * KB * RS_X;
            if constexpr (sizeof(T) == 2) {
                // transposing read: 16-lane group g covers channels 16*(g&1).., pixels 8*(g>>1) + {0..3 | 4..7};
                // lane 4q+p of the group supplies &tile[pixel base + q][channel base + 4p]
                const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
                const int chb = 16 * (g & 1) + 4 * p, pxb = 8 * (g >> 1) + q;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {          // two k16 steps per 32-pixel block
                    u32x4_t fy[TNO], fx[TC];
#pragma unroll
                    for (int i = 0; i < TNO; ++i) {
                        const char* base = pY + (kk * 16 + pxb) * RS_Y + ((wno * TNO + i) * 32 + chb) * 2;
                        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * RS_Y));
                        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                        fy[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
                    }
#pragma unroll
                    for (int j = 0; j < TC; ++j) {
                        const char* base = pX + (kk * 16 + pxb) * RS_X + ((wc * TC + j) * 32 + chb) * 2;
                        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * RS_X));
                        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                        fx[j] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
                    }
#pragma unroll
                    for (int i = 0; i < TNO; ++i)
#pragma unroll
                        for (int j = 0; j < TC; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fy[i]),
                                                                                __builtin_bit_cast(bf16x8_t, fx[j]), acc[i][j], 0, 0, 0);
                }
            } else {
                // f32: one k (pixel) per lane-half per MFMA; lane (r,h) reads tile[pixel 2*k2 + h][channel r]
#pragma unroll 4
                for (int k2 = 0; k2 < KB / 2; ++k2) {
                    float fy[TNO], fx[TC];
#pragma unroll
                    for (int i = 0; i < TNO; ++i) fy[i] = *(const float*)(pY + (2 * k2 + h) * RS_Y + ((wno * TNO + i) * 32 + r) * 4);
#pragma unroll
                    for (int j = 0; j < TC; ++j) fx[j] = *(const float*)(pX + (2 * k2 + h) * RS_X + ((wc * TC + j) * 32 + r) * 4);
#pragma unroll
                    for (int i = 0; i < TNO; ++i)
#pragma unroll
                        for (int j = 0; j < TC; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fy[i], fx[j], acc[i][j], 0, 0, 0);
                }
            }
            if (blk + 1 < blk1) store_step(buf ^ 1);
            __syncthreads();
        }
    }
    // ---- epilogue: D[row = cout][col = cin]; col = lane&31 -> contiguous fp32 in the staging buffer
    const long Kp = (long)taps * a.Cin_p;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int cc = c0 + (wc * TC + j) * 32 + r;
        if (cc >= a.Cin_p) continue;
#pragma unroll
        for (int i = 0; i < TNO; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wno * TNO + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < a.N) {
                    float* dst = a.dw + (long)n * Kp + (long)tap * a.Cin_p + cc;
                    if (a.splits > 1) atomicAdd(dst, acc[i][j][e]);
                    else *dst = acc[i][j][e];
                }
            }
        }
    }
}

// ---- bf16 wgrad with an LDS-DMA ring (the generic kernel above stays for fp32 validation mode and > 2 GiB tensors).
// Same work decomposition (n tile x one tap x c tile x pixel split) and the same transposing-read MFMA feed, but the
// [32 pixels][channels] tiles are written straight into a 4-slot LDS ring by buffer_load ... lds (no register staging,
// three stages in flight, one counted vmcnt + one barrier per 32-pixel step -- the igemm LD = 2 pipeline).  LDS-DMA
// writes 64 consecutive 16-byte slots per wave instruction, so rows cannot be padded; instead the 16-byte chunk index
// is XOR-swizzled on the global side (chunk' = chunk ^ swz(row)) such that the 32 lanes of a ds_read_b64_tr_b16 half
// (4 pixel rows x 2 channel groups x 32 B) cover all 64 banks:  256-byte rows: swz = 4*(row&3); 128-byte rows (two rows
// per bank sweep): swz = 4*((row>>1)&1).
// FL = true: 32-pixel blocks are segments of one image row (wave-uniform SGPR offsets, lane offsets constant);
// FL = false: flattened pixels, each lane tracks the image coordinates of its tile rows incrementally.
template <int OFF> __device__ __forceinline__ unsigned long long lds_tr16(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ bf16x8_t tr_operand(unsigned long long lo, unsigned long long hi) {
    const u32x4_t c{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
    return __builtin_bit_cast(bf16x8_t, c);
}
constexpr int WG_RING = 4;        // LDS ring slots of the DMA wgrad kernel (3 was tried so that it co-resides with 96 KB main-stream workgroups: no gain)
template <int CPRW> __device__ __forceinline__ int wg_swz(int row) { return CPRW >= 16 ? 4 * (row & 3) : 4 * ((row >> 1) & 1); }

template <int WNO, int WC, int TNO, int TC, bool FL>
__global__ __launch_bounds__(WNO * WC * 64) void conv_wgrad_dma_kernel(WgradArgs a) {
    typedef bf16_t T;
    constexpr int NW = WNO * WC;                                      // waves, arranged WNO (cout) x WC (cin)
    constexpr int BNO = WNO * TNO * 32, BC = WC * TC * 32;            // cout x cin tile
    constexpr int KB = 32, ES = 2;
    constexpr int CPR_Y = BNO / 8, CPR_X = BC / 8;                    // 16-byte chunks per tile row (8, 16 or 32)
    constexpr int ROWB_Y = BNO * ES, ROWB_X = BC * ES;
    constexpr int YB = KB * ROWB_Y, XB = KB * ROWB_X, STAGE = YB + XB;
    constexpr int YI = YB / 1024 / NW, XI = XB / 1024 / NW;            // DMA instructions per wave and stage
    static_assert(YI * 1024 * NW == YB && XI * 1024 * NW == XB, "tiles must split evenly over the waves' DMA instructions");
    static_assert(YI >= 1 && XI >= 1, "tile too small for one DMA instruction per wave");
    constexpr int LPS = YI + XI;
    constexpr int ST = WG_RING;                                       // LDS ring slots (ST - 1 stages in flight)
    constexpr unsigned OOB = 0xfffffff0u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int id = xcd_remap(blockIdx.x, gridDim.x);
    const int taps = a.KH * a.KW;
    const int tile_c = id % a.tiles_c; id /= a.tiles_c;
    const int tap = id % taps; id /= taps;
    const int tile_n = id % a.tiles_n; id /= a.tiles_n;
    const int split = id;
    const int ty = tap / a.KW, tx = tap % a.KW;
    const int dyo = ty - (a.KH >> 1), dxo = tx - (a.KW >> 1), padw = a.KW >> 1;
    const int n0 = tile_n * BNO, c0 = tile_c * BC;
    const T* xp = (const T*)a.x;
    const T* yp = (const T*)a.dy;

    const int bpr = (a.W + KB - 1) / KB;
    const long nblk_total = FL ? (long)a.B * a.H * bpr : (a.M + KB - 1) / KB;
    const long blk0 = (long)split * a.blocks_per_split;
    long blk1 = blk0 + a.blocks_per_split; if (blk1 > nblk_total) blk1 = nblk_total;
    const int nst = (int)(blk1 - blk0);

    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // lane constants of the DMA side: tile row inside the instruction and the (swizzled) source chunk
    int rowY[YI], rowX[XI];
    unsigned voffY[YI], voffX[XI];
    int x_oy[XI], x_ox[XI];                                           // !FL: image coordinates of this lane's tile rows
#pragma unroll
    for (int i = 0; i < YI; ++i) {
        rowY[i] = (wv * YI + i) * (64 / CPR_Y) + lane / CPR_Y;
        const int n = n0 + ((lane % CPR_Y) ^ wg_swz<CPR_Y>(rowY[i])) * 8;
        voffY[i] = n < a.N ? (unsigned)((rowY[i] * a.ldy + n) * ES) : OOB;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        rowX[i] = (wv * XI + i) * (64 / CPR_X) + lane / CPR_X;
        const int cc = c0 + ((lane % CPR_X) ^ wg_swz<CPR_X>(rowX[i])) * 8;
        if constexpr (FL) voffX[i] = cc < a.Cin_p ? (unsigned)(((rowX[i] + dxo + padw) * a.ldx + cc) * ES) : OOB;
        else {
            voffX[i] = cc < a.Cin_p ? (unsigned)(cc * ES) : OOB;
            const long m = blk0 * KB + rowX[i];
            const long hw = (long)a.H * a.W;
            const int rem = (int)(m % hw);
            x_oy[i] = rem / a.W; x_ox[i] = rem - x_oy[i] * a.W;
        }
    }
    int f_xb = 0, f_y = 0, f_b = 0;
    long f_blk = blk0;
    if constexpr (FL) { f_xb = (int)(blk0 % bpr); const long t = blk0 / bpr; f_y = (int)(t % a.H); f_b = (int)(t / a.H); }

    auto dma_stage = [&](int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void* lptr_t;
        char* sY = smem + slot * STAGE;
        char* sX = sY + YB;
        const auto rsY = __builtin_amdgcn_make_buffer_rsrc((void*)yp, 0, (int)(((a.M - 1) * a.ldy + a.N) * ES), 0x00020000);
        if constexpr (FL) {
            const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(xp - (long)padw * a.ldx), 0,
                                                               (int)(((a.M - 1 + padw) * a.ldx + a.Cin_p) * ES), 0x00020000);
            const int x0 = f_xb * KB, rem = a.W - x0;
            const int yy = f_y + dyo;
            const bool row_ok = (unsigned)yy < (unsigned)a.H;
            const int soffY = (int)((((long)f_b * a.H + f_y) * a.W + x0) * a.ldy * ES);
            const int soffX = row_ok ? (int)((((long)f_b * a.H + yy) * a.W + x0) * a.ldx * ES) : 0;
#pragma unroll
            for (int i = 0; i < YI; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lptr_t)(sY + (wv * YI + i) * 1024), 16,
                                                         rowY[i] < rem ? voffY[i] : OOB, soffY, 0, 0);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const bool ok = row_ok && rowX[i] < rem && (unsigned)(x0 + rowX[i] + dxo) < (unsigned)a.W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(sX + (wv * XI + i) * 1024), 16, ok ? voffX[i] : OOB, soffX, 0, 0);
            }
            if (++f_xb == bpr) { f_xb = 0; if (++f_y == a.H) { f_y = 0; ++f_b; } }
        } else {
            const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (int)(((a.M - 1) * a.ldx + a.Cin_p) * ES), 0x00020000);
            const int soffY = (int)(f_blk * KB * a.ldy * ES);
#pragma unroll
            for (int i = 0; i < YI; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lptr_t)(sY + (wv * YI + i) * 1024), 16,
                                                         f_blk * KB + rowY[i] < a.M ? voffY[i] : OOB, soffY, 0, 0);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const long m = f_blk * KB + rowX[i];
                const int iy = x_oy[i] + dyo, ix = x_ox[i] + dxo;
                const bool ok = m < a.M && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                const unsigned off = voffX[i] == OOB ? OOB : voffX[i] + (unsigned)((m + (long)dyo * a.W + dxo) * a.ldx * ES);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(sX + (wv * XI + i) * 1024), 16, ok ? off : OOB, 0, 0, 0);
                x_ox[i] += KB;
                while (x_ox[i] >= a.W) { x_ox[i] -= a.W; if (++x_oy[i] == a.H) x_oy[i] = 0; }
            }
            ++f_blk;
        }
#else
        (void)slot;
#endif
    };

    f32x16_t acc[TNO][TC];
#pragma unroll
    for (int i = 0; i < TNO; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wno = wave / WC, wc = wave % WC;
    const int r = lane & 31, h = lane >> 5;
    // transposing-read lane constants: 16-lane group g covers channels 16*(g&1).., pixels 8*(g>>1) + q (+4 for the high half)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int prow = 8 * (g >> 1) + q;
    int offY[TNO], offX[TC];
#pragma unroll
    for (int i = 0; i < TNO; ++i)
        offY[i] = prow * ROWB_Y + ((((wno * TNO + i) * 4 + 2 * (g & 1) + (pp >> 1)) ^ wg_swz<CPR_Y>(prow)) * 16) + (pp & 1) * 8;
#pragma unroll
    for (int j = 0; j < TC; ++j)
        offX[j] = prow * ROWB_X + ((((wc * TC + j) * 4 + 2 * (g & 1) + (pp >> 1)) ^ wg_swz<CPR_X>(prow)) * 16) + (pp & 1) * 8;

    // The transposing reads are issued through inline asm: the compiler does not know which LDS bytes the intrinsic form
    // touches and would fence it with s_waitcnt vmcnt(0) against the LDS-DMA writes still in flight for the NEXT stages,
    // serialising the ring.  The asm results are tied to hand-placed s_waitcnt lgkmcnt (LDS returns in order).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto compute = [&](int slot) {
        unsigned long long y0[TNO][2], y1[TNO][2], x0[TC][2], x1[TC][2];      // [operand][lo/hi pixel quad], k16 step 0 / 1
        const unsigned sb = lds0 + slot * STAGE;
#pragma unroll
        for (int i = 0; i < TNO; ++i) {
            const unsigned ad = sb + offY[i];
            y0[i][0] = lds_tr16<0>(ad); y0[i][1] = lds_tr16<4 * ROWB_Y>(ad);
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const unsigned ad = sb + YB + offX[j];
            x0[j][0] = lds_tr16<0>(ad); x0[j][1] = lds_tr16<4 * ROWB_X>(ad);
        }
#pragma unroll
        for (int i = 0; i < TNO; ++i) {
            const unsigned ad = sb + offY[i];
            y1[i][0] = lds_tr16<16 * ROWB_Y>(ad); y1[i][1] = lds_tr16<20 * ROWB_Y>(ad);
        }
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const unsigned ad = sb + YB + offX[j];
            x1[j][0] = lds_tr16<16 * ROWB_X>(ad); x1[j][1] = lds_tr16<20 * ROWB_X>(ad);
        }
        // first k16 step's operands are in once the second step's 2 * (TNO + TC) reads are all that is outstanding (LDS returns in order)
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TNO + TC)) : "memory");
        __builtin_amdgcn_sched_barrier(0);            // (the compiler moves register-only MFMAs across an asm wait: pin both sides)
#pragma unroll
        for (int i = 0; i < TNO; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(y0[i][0], y0[i][1]), tr_operand(x0[j][0], x0[j][1]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);            // keep the first k16 step's MFMAs ahead of the wait for the second step's reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TNO; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(y1[i][0], y1[i][1]), tr_operand(x1[j][0], x1[j][1]), acc[i][j], 0, 0, 0);
    };

#pragma unroll
    for (int p = 0; p < ST - 1; ++p)
        if (p < nst) dma_stage(p);
    int slot = 0, fill = ST - 1;                                      // ring positions of the stage consumed / filled next
    for (int i = 0; i < nst; ++i) {
        const int rem = nst - 1 - i;
        if (ST == 4 && rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + ST - 1 < nst) dma_stage(fill);
        compute(slot);
        slot = slot + 1 == ST ? 0 : slot + 1;
        fill = fill + 1 == ST ? 0 : fill + 1;
    }

    // ---- epilogue: D[row = cout][col = cin]; col = lane&31 -> contiguous fp32 in the staging buffer
    const long Kp = (long)taps * a.Cin_p;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
        const int cc = c0 + (wc * TC + j) * 32 + r;
        if (cc >= a.Cin_p) continue;
#pragma unroll
        for (int i = 0; i < TNO; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wno * TNO + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (n < a.N) {
                    float* dst = a.dw + (long)split * a.part_stride + (long)n * Kp + (long)tap * a.Cin_p + cc;
                    if (a.splits > 1 && a.part_stride == 0) atomicAdd(dst, acc[i][j][e]);
                    else *dst = acc[i][j][e];
                }
            }
        }
    }
}

int g_wgrad_dma = 1;                                 // development knob (mte_debug_set(4, v))

#ifndef MTE_WGRAD_WGS
#define MTE_WGRAD_WGS 512
#endif
int g_wgrad_wgs = MTE_WGRAD_WGS;                     // development knob (mte_debug_set(9, v)): workgroups aimed for (pixel splits)
int g_wgrad_big = 1;                                 // development knob (mte_debug_set(8, v)): 256 x 256 / 256 x 128 / 128 x 256 tiles
                                                     // (4x fewer re-reads of dy / x; pays once the pixel splits are few: g_wgrad_wgs)

template <int WNO, int WC, int TNO, int TC>
int launch_wgrad_dma(WgradArgs a, hipStream_t st, int parts_cap, int* parts_out) {
    constexpr int BNO = WNO * TNO * 32, BC = WC * TC * 32, NTHR = WNO * WC * 64;
    a.tiles_n = (a.N + BNO - 1) / BNO;
    a.tiles_c = (a.Cin_p + BC - 1) / BC;
    const int taps = a.KH * a.KW;
    // row-aligned 32-pixel blocks waste MFMA work when W is not a multiple of 32 (W = 40: 37 %): use them for wide rows only
    const bool fl = a.W % 32 == 0 || a.W >= 160;
    const long nblk = fl ? (long)a.B * a.H * ((a.W + 31) / 32) : (a.M + 31) / 32;
    const long base_wgs = (long)a.tiles_n * a.tiles_c * taps;
    // ~4 workgroups of 256 threads (or 1 of 1024) per CU; three quarters of that beside the data-gradient chain (MTE_OPT_WGRAD_SHARES_CHIP; end of round 5,
    // same box, ms per step: 512 -> 22.85, 384 -> 22.70, 256 -> 22.79, 768 -> 22.77 -- profiles/r05_side_queue_width.txt)
    const long want = (long)(g_mte_wgrad_shared ? g_wgrad_wgs * 3 / 4 : g_wgrad_wgs) * 256 / NTHR;
    long splits = (want + base_wgs - 1) / base_wgs;
    const long max_splits = (nblk + 15) / 16;                   // at least 16 pixel blocks per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (!g_mte_wgrad_shared) {
        // alone on the chip the launch runs in whole rounds of (CUs x workgroups per CU): 50 tiles x 6 splits = 300 one-per-CU workgroups took two rounds, the second
        // 17 % full (5x5 512 -> 128 @48x160: 0.366 ms at 550 TFLOP/s; 5 splits = 250 workgroups: one round).  Among the split counts around the one above take the
        // cheapest in rounds per split; ties go to fewer partial slabs.
        static int cus = 0;
        if (!cus) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
        const size_t lds_wg = (size_t)WG_RING * 32 * (BNO + BC) * 2;
        long per_cu = (long)(160 * 1024 / lds_wg);
        if (per_cu > 2048 / NTHR) per_cu = 2048 / NTHR;
        if (per_cu < 1) per_cu = 1;
        const long slots = per_cu * cus;
        long best = splits; double best_cost = 1e30;
        for (long sc = splits > 2 ? splits - 2 : 1; sc <= splits + 2 && sc <= max_splits && sc <= (parts_cap < 1 ? 1 : parts_cap); ++sc) {
            const double cost = (double)((base_wgs * sc + slots - 1) / slots) / (double)sc;
            if (cost < best_cost * 0.98) { best_cost = cost; best = sc; }
        }
        splits = best;
    }
    // one partial gradient per pixel split (plain stores, summed in part order by the unpack pass): never more splits than the caller's stage
    // has parts -- round 4: the fp32-atomic combine that used to take over beyond stage_parts is gone from this kernel's launch path, the
    // weight gradient is a fixed-order sum (the atomics cost 0.4 ms per step when they were the default, and made the result order-dependent)
    if (splits > parts_cap) splits = parts_cap < 1 ? 1 : parts_cap;
    a.blocks_per_split = (int)((nblk + splits - 1) / splits);
    a.splits = (int)((nblk + a.blocks_per_split - 1) / a.blocks_per_split);
    if (a.splits > 1 && a.splits <= parts_cap) {
        a.part_stride = (long)a.N * taps * a.Cin_p;
        if (parts_out) *parts_out = a.splits;
    } else {
        a.part_stride = 0;
        if (parts_out) *parts_out = 1;
        if (a.splits > 1 && mte_memset_async(a.dw, 0, sizeof(float) * (size_t)a.N * taps * a.Cin_p, st) != hipSuccess) return MTE_ERR_LAUNCH;
    }
    const size_t lds = WG_RING * 32 * (BNO + BC) * 2;
    const dim3 grid((unsigned)(base_wgs * a.splits));
    if (fl) hipLaunchKernelGGL((conv_wgrad_dma_kernel<WNO, WC, TNO, TC, true>), grid, dim3(NTHR), lds, st, a);
    else hipLaunchKernelGGL((conv_wgrad_dma_kernel<WNO, WC, TNO, TC, false>), grid, dim3(NTHR), lds, st, a);
    return mte_check_launch();
}

template <typename T, int WNO, int WC, int TNO, int TC>
int launch_wgrad(WgradArgs a, hipStream_t st) {
    constexpr int BNO = WNO * TNO * 32, BC = WC * TC * 32, ES = (int)sizeof(T);
    constexpr int RS_Y = (BNO * ES % 128 == 64) ? BNO * ES : BNO * ES + 64;
    constexpr int RS_X = (BC * ES % 128 == 64) ? BC * ES : BC * ES + 64;
    a.tiles_n = (a.N + BNO - 1) / BNO;
    a.tiles_c = (a.Cin_p + BC - 1) / BC;
    const int taps = a.KH * a.KW;
    const long es = (long)sizeof(T);
    // row-aligned 32-pixel blocks waste MFMA work when W is not a multiple of 32 (W = 40: 37 %): use them for wide rows only
    const bool fl = (a.W % 32 == 0 || a.W >= 160) &&
                    ((a.M + a.KW) * a.ldx + a.Cin_p) * es < 0x7ff00000L && ((a.M - 1) * a.ldy + a.N) * es < 0x7ff00000L;
    const long nblk = fl ? (long)a.B * a.H * ((a.W + 31) / 32) : (a.M + 31) / 32;
    const long base_wgs = (long)a.tiles_n * a.tiles_c * taps;
    long splits = (1024 + base_wgs - 1) / base_wgs;            // aim for >= ~4 workgroups per CU
    const long max_splits = (nblk + 15) / 16;                   // at least 16 pixel blocks per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    a.blocks_per_split = (int)((nblk + splits - 1) / splits);
    a.splits = (int)((nblk + a.blocks_per_split - 1) / a.blocks_per_split);
    if (a.splits > 1) {
        hipError_t e = mte_memset_async(a.dw, 0, sizeof(float) * (size_t)a.N * taps * a.Cin_p, st);
        if (e != hipSuccess) return MTE_ERR_LAUNCH;
    }
    const size_t lds = 2 * 32 * (RS_Y + RS_X);
    if (fl) hipLaunchKernelGGL((conv_wgrad_kernel<T, WNO, WC, TNO, TC, true>), dim3((unsigned)(base_wgs * a.splits)), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, WNO, WC, TNO, TC, false>), dim3((unsigned)(base_wgs * a.splits)), dim3(256), lds, st, a);
    return mte_check_launch();
}

template <typename T> int dispatch_wgrad(const WgradArgs& a, hipStream_t st, int parts_cap, int* parts_out) {
    if (parts_out) *parts_out = 1;
    if constexpr (sizeof(T) == 2) {
        const bool fits = ((a.M + a.KW) * a.ldx + a.Cin_p) * 2 < 0x7ff00000L && ((a.M - 1) * a.ldy + a.N) * 2 < 0x7ff00000L;
        if (g_wgrad_dma && fits && a.N > 32 && a.Cin_p > 32)
        {
            // (round 4, measured and NOT adopted: 8 waves of 128 x 64 on the same tile -- 1.5 transposing reads per MFMA instead of 2 -- run 20-30 % slower
            //  than the 16-wave form on every 256-multiple layer, 256 -> 256 @48x160 167 vs 130 us with the unpack pass: with one barrier per
            //  32-pixel step the loop needs its four waves per SIMD; development knob 8 = 3 selects it)
            if (g_wgrad_big == 3 && a.N % 256 == 0 && a.Cin_p % 256 == 0) return launch_wgrad_dma<2, 4, 4, 2>(a, st, parts_cap, parts_out);   // 256 x 256, 8 waves
            if (g_wgrad_big && a.N % 256 == 0 && a.Cin_p % 256 == 0) return launch_wgrad_dma<4, 4, 2, 2>(a, st, parts_cap, parts_out);   // 256 x 256, 16 waves
            if (g_wgrad_big && a.N % 256 == 0 && a.Cin_p >= 128) return launch_wgrad_dma<4, 2, 2, 2>(a, st, parts_cap, parts_out);     // 256 x 128, 8 waves
            if (g_wgrad_big && a.N >= 128 && a.Cin_p % 256 == 0) return launch_wgrad_dma<2, 4, 2, 2>(a, st, parts_cap, parts_out);     // 128 x 256, 8 waves
            return a.N <= 64 ? launch_wgrad_dma<2, 2, 1, 2>(a, st, parts_cap, parts_out) : launch_wgrad_dma<2, 2, 2, 2>(a, st, parts_cap, parts_out);
        }
    }
    if (a.N <= 32) {
        if (a.Cin_p <= 32) return launch_wgrad<T, 1, 4, 1, 1>(a, st);      // 32 x 128 would waste: 32 x (4*32)
        return launch_wgrad<T, 1, 4, 1, 1>(a, st);                          // cout 32 x cin 128
    }
    if (a.N <= 64) return launch_wgrad<T, 2, 2, 1, 2>(a, st);              // cout 64 x cin 128
    return launch_wgrad<T, 2, 2, 2, 2>(a, st);                              // cout 128 x cin 128
}

// ---- weight packing: OIHW fp32 master -> [N][taps][Cin_p] (forward) and [Cin_p8][taps flipped][Cout_p] (dgrad).
// Both passes go through LDS so that global reads AND writes are contiguous runs (the naive gather read the fp32
// master with a taps*4-byte stride and cost 1.2 ms per step for the 77 M parameters).
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_fwd_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                                               int Cout, int Cin, int taps, int Cin_p) {
    extern __shared__ float s_w[];                                   // [64 channels][taps]
    const int n = blockIdx.x, c0 = blockIdx.y * 64;
    const int nc = min(64, Cin - c0);                                // real channels in this chunk (may be <= 0)
    const float* src = w + ((long)n * Cin + c0) * taps;
    for (int i = threadIdx.x; i < nc * taps; i += 256) s_w[i] = src[i];
    __syncthreads();
    const int ncp = min(64, Cin_p - c0);                             // channels incl. zero padding
    for (int i = threadIdx.x; i < taps * ncp; i += 256) {
        const int tap = i / ncp, cl = i - tap * ncp;
        Elem<T>::st(wf + ((long)n * taps + tap) * Cin_p + c0 + cl, cl < nc ? s_w[cl * taps + tap] : 0.f);
    }
}

// wb[(c*taps + tb)*Cout_p + n] = wf[n*Kp + (taps-1-tb)*Cin_p + c]   (64 x 64 tiles through LDS)
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_bwd_kernel(const T* __restrict__ wf, T* __restrict__ wb,
                                                               int Cout, int taps, int Cin_p, int Cout_p) {
    __shared__ T tile[64][64 + 2];
    const int Kp = taps * Cin_p;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i >> 6, cidx = i & 63;
        const int n = n0 + r, k = k0 + cidx;
        tile[r][cidx] = (n < Cout && k < Kp) ? wf[(long)n * Kp + k] : (T)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int kk = i >> 6, nn = i & 63;
        const int k = k0 + kk, n = n0 + nn;
        if (k < Kp && n < Cout_p) {
            const int tap = k / Cin_p, c = k - tap * Cin_p;
            wb[((long)c * taps + (taps - 1 - tap)) * Cout_p + n] = n < Cout ? tile[nn][kk] : (T)0;
        }
    }
}

// ---- every weight pack of the network in ONE launch (per <= 48 layers).  After an optimizer step ~100 conv weights need their
// kernel-ready copies rebuilt: forward pack, data-gradient pack and, for the LDS-patch layers, the two fragment-block packs -- up to
// four tiny launches per layer, ~400 per training step, 2-10 us each and launch-bound (0.88 ms of GPU time and ~2 ms of host time
// per step for 0.77 GB of traffic).  Here a job table travels as a kernel argument and every block finds its (job, phase) by a
// scalar scan; all four packs are formed from the fp32 OIHW master directly, so the phases are independent of each other.
struct PackJob {                    // == mte_pack_job of include/mte_kernels.h
    const float* w;                 // OIHW fp32 master [Cout][Cin][taps]
    void* wf;                       // [Cout][taps][Cin_p]                       (always)
    void* wb;                       // [Cin_p][taps rot180][Cout]                (nullable)
    void* pf;                       // fragment blocks of wf (conv_patch.hip)    (nullable, bf16 only)
    void* pb;                       // fragment blocks of wb                     (nullable, bf16 only)
    int Cout, Cin, taps, Cin_p;
    int end_f, end_b, end_pf, end_pb;   // filled by the launcher: cumulative block ends of the job's four phases
};
constexpr int PACK_JOBS_MAX = 48;
struct PackTable { int n; int pad_; PackJob j[PACK_JOBS_MAX]; };
constexpr int PACK_FRAG_PER_BLOCK = 256 * 8;                        // fragment-pack elements per block (one 16-byte chunk per thread)

template <typename T> __device__ __forceinline__ void store8(T* dst, const float* v);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* dst, const float* v) { *(u32x4_t*)dst = pack16<bf16_t>(v); }
template <> __device__ __forceinline__ void store8<float>(float* dst, const float* v) {
    *(u32x4_t*)dst = pack16<float>(v); *(u32x4_t*)(dst + 4) = pack16<float>(v + 4);
}

// output channels per block of the forward-pack phase: what fits the 64 x 65 floats of LDS beside 64 input channels x taps, at most 16
__host__ __device__ inline int pack_fn(int taps) { const int f = (64 * 65) / (64 * taps); return f < 1 ? 1 : (f > 16 ? 16 : f); }

template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackTable tb) {
    __shared__ float s_w[64 * 65];                                   // phase f: [64 channels][taps <= 49]; phase b: [64][65] tile
    int ji = 0;
    while (ji + 1 < tb.n && (int)blockIdx.x >= tb.j[ji].end_pb) ++ji;    // wave-uniform scalar scan over <= 48 jobs
    const PackJob& J = tb.j[ji];
    const int start = ji ? tb.j[ji - 1].end_pb : 0;
    const int bid = blockIdx.x;
    const int Cout = J.Cout, Cin = J.Cin, taps = J.taps, Cin_p = J.Cin_p;
    const float* __restrict__ w = J.w;
    if (bid < J.end_f) {
        // ---- forward pack: block = (FN output channels, 64-channel chunk); OIHW runs of 64*taps floats -> LDS -> one 16-byte store of 8 channels per
        // (n, tap, chunk).  (Round 4: one n per block and 2-byte stores ran the ~1 GB of pack traffic of a step at 1.8 TB/s.)
        const int fn = pack_fn(taps);
        const int cchunks = (Cin_p + 63) >> 6;
        const int lb = bid - start, n0 = (lb / cchunks) * fn, c0 = (lb - (lb / cchunks) * cchunks) * 64;
        const int nc = min(64, Cin - c0), nn = min(fn, Cout - n0);
        const int run = nc > 0 ? nc * taps : 0, rs = 64 * taps;     // floats of one n in global memory (contiguous) / in LDS
        for (int i = threadIdx.x; i < nn * run; i += 256) {
            const int ni = i / run, r = i - ni * run;
            s_w[ni * rs + r] = w[((long)(n0 + ni) * Cin + c0) * taps + r];
        }
        __syncthreads();
        const int ncp = min(64, Cin_p - c0), nch = ncp >> 3;        // channels incl. zero padding (a multiple of 8), 16-byte chunks of them
        T* wf = (T*)J.wf;
        for (int i = threadIdx.x; i < nn * taps * nch; i += 256) {
            const int ch8 = i % nch, t2 = i / nch, tap = t2 % taps, ni = t2 / taps;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cl = ch8 * 8 + j;
                v[j] = cl < nc ? s_w[ni * rs + cl * taps + tap] : 0.f;
            }
            store8<T>(wf + ((long)(n0 + ni) * taps + tap) * Cin_p + c0 + ch8 * 8, v);
        }
        return;
    }
    if (bid < J.end_b) {
        // ---- data-gradient pack: wb[(c*taps + (taps-1-tap))*Cout + n] = w[n][c][tap]; tile = 64 n x 64 consecutive (c, tap)
        const int Kp = taps * Cin_p, Kr = taps * Cin;                // (c, tap) positions incl. / excl. channel padding
        const int ktiles = (Kp + 63) >> 6;
        const int lb = bid - J.end_f, n0 = (lb / ktiles) * 64, k0 = (lb % ktiles) * 64;
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int r = i >> 6, kk = i & 63;
            const int n = n0 + r, k = k0 + kk;
            s_w[r * 65 + kk] = (n < Cout && k < Kr) ? w[(long)n * Kr + k] : 0.f;
        }
        __syncthreads();
        T* wb = (T*)J.wb;
        if (Cout % 8 == 0) {                                         // 16-byte stores of 8 output channels
            for (int i = threadIdx.x; i < 64 * 8; i += 256) {
                const int kk = i >> 3, n8 = i & 7;
                const int k = k0 + kk, n = n0 + n8 * 8;
                if (k < Kp && n < Cout) {
                    const int c = k / taps, tap = k - c * taps;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = s_w[(n8 * 8 + j) * 65 + kk];
                    store8<T>(wb + ((long)c * taps + (taps - 1 - tap)) * Cout + n, v);
                }
            }
            return;
        }
        for (int i = threadIdx.x; i < 64 * 64; i += 256) {
            const int kk = i >> 6, nn = i & 63;
            const int k = k0 + kk, n = n0 + nn;
            if (k < Kp && n < Cout) {
                const int c = k / taps, tap = k - c * taps;
                Elem<T>::st(wb + ((long)c * taps + (taps - 1 - tap)) * Cout + n, s_w[nn * 65 + kk]);
            }
        }
        return;
    }
    // ---- fragment-block packs of the LDS-patch kernels: [slice][tap][kk][nt][lane = h*32 + r][8]; one 16-byte chunk per thread
    const bool fwd = bid < J.end_pf;
    const int N = fwd ? Cout : Cin_p, C = fwd ? Cin_p : Cout;        // GEMM columns / reduction channels of this direction
    const int NT = (N + 31) >> 5;
    const long total8 = (long)((C + 31) >> 5) * taps * 2 * NT * 64;
    const long i8 = (long)(bid - (fwd ? J.end_b : J.end_pf)) * 256 + threadIdx.x;
    if (i8 >= total8) return;
    long t = i8;
    const int lane = (int)(t & 63); t >>= 6;
    const int nt = (int)(t % NT); t /= NT;
    const int kk = (int)(t & 1); t >>= 1;
    const int tap = (int)(t % taps); const int sl = (int)(t / taps);
    const int n = nt * 32 + (lane & 31), c0 = sl * 32 + kk * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        float x = 0.f;
        if (n < N && c < C) {
            if (fwd) { if (c < Cin) x = w[((long)n * Cin + c) * taps + tap]; }                    // wf[n][tap][c]
            else if (n < Cin) x = w[((long)c * Cin + n) * taps + (taps - 1 - tap)];               // wb[n' = cin][tap][c' = cout]
        }
        v[j] = x;
    }
    store8<T>((T*)(fwd ? J.pf : J.pb) + i8 * 8, v);
}

// staging [N][taps][Cin_p] fp32 -> OIHW fp32 gradient (overwrite).  One block = (output channel n, 64 input channels): the
// [taps][64] slab is read along c (coalesced), transposed in LDS and written as one contiguous run of 64*taps floats.
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ st, float* __restrict__ dw, int Cout, int Cin, int taps, int Cin_p, int parts) {
    extern __shared__ float s_t[];                                   // [taps][64 + 1]
    const int n = blockIdx.x, c0 = blockIdx.y * 64;
    const int nc = min(64, Cin - c0);
    if (nc <= 0) return;
    const float* src = st + (long)n * taps * Cin_p + c0;
    const long pstride = (long)Cout * taps * Cin_p;
    // 16-byte loads along c, the parts of an element eight at a time in flight, added IN PART ORDER (round 4: this pass reads 2.6 GB per
    // training step -- with one 4-byte load per part in a dependent chain it ran at the memory latency, not the bandwidth)
    for (int i = threadIdx.x; i < taps * 16; i += 256) {
        const int tap = i >> 4, c4 = (i & 15) * 4;
        if (c0 + c4 >= Cin_p) continue;                              // (Cin_p is a multiple of 8: a float4 at a multiple of 4 below it is inside the row)
        const float* q = src + (long)tap * Cin_p + c4;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        int p = 0;
        for (; p + 8 <= parts; p += 8) {
            f32x4_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *(const f32x4_t*)(q + (long)(p + k) * pstride);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        for (; p < parts; ++p) acc += *(const f32x4_t*)(q + (long)p * pstride);
#pragma unroll
        for (int e = 0; e < 4; ++e) s_t[tap * 65 + c4 + e] = acc[e];
    }
    __syncthreads();
    float* dst = dw + ((long)n * Cin + c0) * taps;
    for (int i = threadIdx.x; i < nc * taps; i += 256) {
        const int cl = i / taps, tap = i - cl * taps;
        dst[i] = s_t[tap * 65 + cl];
    }
}

// First level of the two-level combine of many partial gradients (the LDS-patch weight gradient leaves one slab per workgroup,
// 128..512 of them).  Block (x, y) adds the RP_PC consecutive parts [y * RP_PC, (y + 1) * RP_PC) over 1024 consecutive floats with
// 16-byte loads, all RP_PC loads of a thread in flight together, IN PART ORDER, and stores the sum as slab y of the scratch area BEHIND the
// parts (st + parts * elems); the unpack pass then adds the <= 32 scratch slabs in order.  Round 4: plain stores instead of float atomics
// onto part 0 -- the weight gradient no longer depends on the order in which workgroups arrive (bit-reproducible backward for the conv
// weights), for (parts / 16) / parts = 6 % more bytes.  (History: the one-level fp32-atomic combine pushed every workgroup's whole slab,
// 75-105 MB per launch, through the ~1.3 TB/s atomic path onto the SAME 150-200 KB: 58-80 us of a 190-330 us kernel.)
constexpr int RP_PC = 16;
__global__ __launch_bounds__(256) void reduce_parts_kernel(float* __restrict__ st, long elems, int parts, int pc) {
    const long i4 = blockIdx.x * 256L + threadIdx.x;                 // float4 index inside a slab
    if (i4 * 4 >= elems) return;
    const int p0 = blockIdx.y * pc, p1 = min(parts, p0 + pc);        // this block's parts (pc = a multiple of RP_PC)
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = p0; q < p1; q += RP_PC) {
        f32x4_t v[RP_PC];
#pragma unroll
        for (int k = 0; k < RP_PC; ++k)
            v[k] = q + k < p1 ? *(const f32x4_t*)(st + (long)(q + k) * elems + i4 * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < RP_PC; ++k) acc += v[k];
    }
    *(f32x4_t*)(st + ((long)parts + blockIdx.y) * elems + i4 * 4) = acc;
}

// column sums of a [M][N] (row stride ld) matrix: bias gradient
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ y, long ld, long M, int N, float* __restrict__ out) {
    // block handles a slab of rows; thread t owns column (t % NC) of a 16-B chunk group
    constexpr int PER16 = Elem<T>::PER16;
    const int cpr = N / PER16;                       // chunks per row (N multiple of PER16)
    const int tid = threadIdx.x;
    float acc[PER16];
#pragma unroll
    for (int i = 0; i < PER16; ++i) acc[i] = 0.f;
    const int cc = tid % cpr;                        // requires cpr <= 256 and 256 % cpr == 0 or we guard
    const int rlane = tid / cpr, rstep = 256 / cpr;
    extern __shared__ float s_col[];                  // [N]
    for (int i = tid; i < N; i += 256) s_col[i] = 0.f;
    __syncthreads();
    if (rlane < rstep) {
        for (long m = (long)blockIdx.x * rstep + rlane; m < M; m += (long)gridDim.x * rstep) {
            float v[PER16];
            unpack16<T>(*(const u32x4_t*)(y + m * ld + cc * PER16), v);
#pragma unroll
            for (int i = 0; i < PER16; ++i) acc[i] += v[i];
        }
#pragma unroll
        for (int i = 0; i < PER16; ++i) atomicAdd(&s_col[cc * PER16 + i], acc[i]);      // LDS atomics
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) atomicAdd(out + i, s_col[i]);                     // one global atomic per column per block
}

}  // namespace

extern "C" {

#ifdef MTE_DEV
// development knobs (libmte_hip_dev.so only, -DMTE_DEV): key 0 = igemm tile loader (1 = LDS-DMA ring with buffer descriptors where possible, 2 = pointer DMA only,
// 0 = register staging); key 1 = conv3d pack stencils
// (1 = LDS-tiled, 0 = gather).  Not part of the product contract.
extern "C" int mtei_set_pack3d_lds(int value);
extern "C" int mtei_set_gn(int which, int value);
extern "C" int mtei_set_patch_tall(int v);
extern "C" int mtei_set_tap_wgrad(int v);            // tap_wgrad.hip
extern "C" int mtei_set_head_mfma(int v);
extern int g_wgrad9, g_wgrad9_wgs, g_igemm8_one;
int mte_debug_set(int key, int value) {
    if (key == 26) { g_wgrad9 = value; return MTE_OK; }
    if (key == 27) { g_wgrad9_wgs = value; return MTE_OK; }
    if (key == 28) { g_igemm8_one = value; return MTE_OK; }
    if (key == 30) return mtei_set_head_mfma(value);
    if (key == 31) return mtei_set_tap_wgrad(value);
    if (key == 0) { g_igemm_dma = value; return MTE_OK; }
    if (key == 1) return mtei_set_pack3d_lds(value);
    if (key == 2 || key == 3) return mtei_set_gn(key - 2, value);
    if (key == 13) return mtei_set_gn(2, value);
    if (key == 15) { g_igemm_ring6 = value; return MTE_OK; }
    if (key == 17) { g_igemm_ablate = value; return MTE_OK; }
    if (key == 19) { g_igemm_pair_ksteps = value; return MTE_OK; }
    if (key == 21) { g_igemm_pp = value; return MTE_OK; }
    if (key == 23) { g_igemm8 = value; return MTE_OK; }
    if (key == 24) { g_igemm8_min_tiles = value; return MTE_OK; }
    if (key == 29) { g_igemm8_split_bn128 = value; return MTE_OK; }
    if (key == 32) { g_igemm_n32_dma = value; return MTE_OK; }
    if (key == 14) return mtei_set_gn(3, value);
    if (key == 25) return mtei_set_gn(4, value);
    if (key == 4) { g_wgrad_dma = value; return MTE_OK; }
    if (key == 6) { g_igemm_big = value; return MTE_OK; }
    if (key == 8) { g_wgrad_big = value; return MTE_OK; }
    if (key == 9) { g_wgrad_wgs = value; return MTE_OK; }
    if (key == 11) return mtei_set_patch_tall(value);
    if (key == 7) { g_igemm_big_min_tiles = value; return MTE_OK; }
    return MTE_ERR_ARG;
}
#endif

// y[B,H,W,(ldy)] = conv(x[B,H,W,(ldx)], wpack[N][KH*KW][Cin_p]) + bias; stride 1, zero pad k/2.
// workspace (nullable): fp32 scratch of workspace_elems >= 2*B*H*W*N elements enables split-K for small-M / huge-K shapes
// (one [M][N] slab per split, at most 8; slabs are added in order by a finish kernel: bit-reproducible, nothing to clear).
int mte_conv2d_igemm(const void* x, long ldx, const void* wpack, const float* bias, void* y, long ldy, int out_f32,
                     int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype,
                     float* workspace, long workspace_elems, int accumulate, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpack || !y || B <= 0 || H <= 0 || W <= 0 || N <= 0) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (Cin_p % 8 != 0 || ldx % per16 != 0 || (KH & 1) == 0 || (KW & 1) == 0) return MTE_ERR_ARG;
    ConvArgs a{x, ldx, wpack, bias, y, ldy, out_f32, B, H, W, Cin_p, N, KH, KW, (long)B * H * W, 1, workspace, accumulate & 1, (accumulate >> 1) & 1,
               nullptr, nullptr};
    if (dtype == MTE_DT_BF16) return dispatch_igemm<bf16_t>(a, workspace_elems, stream);
    if (dtype == MTE_DT_F32) return dispatch_igemm<float>(a, workspace_elems, stream);
    return MTE_ERR_UNSUPPORTED;
}

// The data gradient of a folded pack layer written WITHOUT the pixel-shuffle pass behind it (round 6): N = 4 C packed depths d = 4 c + s per pixel of the [B][H][W]
// packed grid, y = the un-shuffled tensor [B][2H][2W][C] (pixel stride ldy); accumulate as in mte_conv2d_igemm (bit 0).  MTE_ERR_UNSUPPORTED where the launch
// would not take a tile form that stages its result in LDS (the caller then runs mte_conv2d_igemm + mte_pixel_shuffle).  bf16 only.
int mte_conv2d_igemm_unshuffle(const void* x, long ldx, const void* wpack, void* y, long ldy,
                               int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype, int accumulate, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpack || !y || B <= 0 || H <= 0 || W <= 0 || N <= 0) return MTE_ERR_ARG;
    if (dtype != MTE_DT_BF16) return MTE_ERR_UNSUPPORTED;
    if (Cin_p % 8 != 0 || ldx % 8 != 0 || ldy % 8 != 0 || (KH & 1) == 0 || (KW & 1) == 0 || N % 32 != 0) return MTE_ERR_ARG;
    ConvArgs a{x, ldx, wpack, nullptr, y, ldy, 0, B, H, W, Cin_p, N, KH, KW, (long)B * H * W, 1, nullptr, accumulate & 1, 0, nullptr, nullptr, 0, N / 4};
    if (!(g_igemm_dma == 1 && Cin_p % 32 == 0)) return MTE_ERR_UNSUPPORTED;
    return dispatch_igemm<bf16_t>(a, 0, stream);
}

// The same convolution over the ACTIVE SITES of a sparse map only (SAN branch): x / y are the dense zero-filled NHWC maps, `sites` the
// pixel indices of the active set (mte_sparse_site_list), `count` their number in DEVICE memory.  y is written at the active sites and
// nowhere else.  The grid covers the dense capacity and tiles past *count return at once, so no count crosses to the host.
int mte_conv2d_igemm_sparse(const void* x, long ldx, const void* wpack, const float* bias, void* y, long ldy,
                            int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype,
                            const int* sites, const int* count, int accumulate, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpack || !y || !sites || !count || B <= 0 || H <= 0 || W <= 0 || N <= 0 || (long)B * H * W > 0x7fffffffL) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (Cin_p % 8 != 0 || ldx % per16 != 0 || (KH & 1) == 0 || (KW & 1) == 0) return MTE_ERR_ARG;
    ConvArgs a{x, ldx, wpack, bias, y, ldy, 0, B, H, W, Cin_p, N, KH, KW, (long)B * H * W, 1, nullptr, accumulate & 1, (accumulate >> 1) & 1,
               sites, count};
    if (dtype == MTE_DT_BF16) return dispatch_igemm<bf16_t>(a, 0, stream);
    if (dtype == MTE_DT_F32) return dispatch_igemm<float>(a, 0, stream);
    return MTE_ERR_UNSUPPORTED;
}

// dw_stage[N][KH*KW][Cin_p] (fp32) = sum over pixels of dy (x) shifted x.
int mte_conv2d_wgrad(const void* x, long ldx, const void* dy, long ldy, float* dw_stage, int stage_parts, int* parts_out,
                     int B, int H, int W, int Cin_p, int N, int KH, int KW, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dy || !dw_stage) return MTE_ERR_ARG;
    if (Cin_p % 8 != 0 || N % 8 != 0) return MTE_ERR_ARG;
    WgradArgs a{x, ldx, dy, ldy, dw_stage, B, H, W, Cin_p, N, KH, KW, (long)B * H * W, 1, 0, 0, 0};
    if (dtype == MTE_DT_BF16 && KH == 3 && KW == 3 && stage_parts >= 1 && parts_out) {
        // round 5: the 3x3 layers with >= 64 / 128 channels take all nine taps from one staged patch (conv_wgrad9.hip)
        const int rc = wgrad9_launch(x, ldx, dy, ldy, dw_stage, stage_parts, parts_out, B, H, W, Cin_p, N, stream);
        if (rc != MTE_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == MTE_DT_BF16) return dispatch_wgrad<bf16_t>(a, stream, stage_parts, parts_out);
    if (dtype == MTE_DT_F32) return dispatch_wgrad<float>(a, stream, stage_parts, parts_out);
    return MTE_ERR_UNSUPPORTED;
}

int mte_pack_conv_weights(const float* w_oihw, void* wfwd, void* wbwd, int Cout, int Cin, int KH, int KW,
                          int Cin_p, int Cout_p, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!w_oihw || !wfwd) return MTE_ERR_ARG;
    const int taps = KH * KW;
    if (Cout_p != Cout && wbwd) return MTE_ERR_ARG;
    const dim3 gf(Cout, (Cin_p + 63) / 64), gb((Cout + 63) / 64, (taps * Cin_p + 63) / 64);
    const size_t lds = sizeof(float) * 64 * taps;
    if (dtype == MTE_DT_BF16) {
        hipLaunchKernelGGL(pack_weights_fwd_kernel<bf16_t>, gf, dim3(256), lds, stream, w_oihw, (bf16_t*)wfwd, Cout, Cin, taps, Cin_p);
        if (wbwd) hipLaunchKernelGGL(pack_weights_bwd_kernel<bf16_t>, gb, dim3(256), 0, stream, (const bf16_t*)wfwd, (bf16_t*)wbwd, Cout, taps, Cin_p, Cout_p);
    } else {
        hipLaunchKernelGGL(pack_weights_fwd_kernel<float>, gf, dim3(256), lds, stream, w_oihw, (float*)wfwd, Cout, Cin, taps, Cin_p);
        if (wbwd) hipLaunchKernelGGL(pack_weights_bwd_kernel<float>, gb, dim3(256), 0, stream, (const float*)wfwd, (float*)wbwd, Cout, taps, Cin_p, Cout_p);
    }
    return mte_check_launch();
}

// Every weight pack of `njobs` conv layers (HOST array of mte_pack_job) in ceil(njobs / 48) launches: see pack_multi_kernel.
int mte_pack_conv_weights_multi(const void* jobs_host, int njobs, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!jobs_host || njobs < 0 || (dtype != MTE_DT_BF16 && dtype != MTE_DT_F32)) return MTE_ERR_ARG;
    const PackJob* src = (const PackJob*)jobs_host;
    for (int j0 = 0; j0 < njobs; j0 += PACK_JOBS_MAX) {
        PackTable tb{};
        tb.n = njobs - j0 < PACK_JOBS_MAX ? njobs - j0 : PACK_JOBS_MAX;
        long blocks = 0;
        for (int i = 0; i < tb.n; ++i) {
            PackJob J = src[j0 + i];
            if (!J.w || !J.wf || J.taps < 1 || J.taps > 49 || J.Cin_p % 8 != 0 || J.Cin > J.Cin_p) return MTE_ERR_ARG;
            if ((J.pf || J.pb) && dtype != MTE_DT_BF16) return MTE_ERR_ARG;
            if (J.pb && !J.wb) return MTE_ERR_ARG;
            blocks += (long)((J.Cout + pack_fn(J.taps) - 1) / pack_fn(J.taps)) * ((J.Cin_p + 63) / 64);
            J.end_f = (int)blocks;
            if (J.wb) blocks += (long)((J.Cout + 63) / 64) * ((J.taps * J.Cin_p + 63) / 64);
            J.end_b = (int)blocks;
            if (J.pf) blocks += ((long)((J.Cin_p + 31) / 32) * J.taps * 2 * ((J.Cout + 31) / 32) * 64 + 255) / 256;
            J.end_pf = (int)blocks;
            if (J.pb) blocks += ((long)((J.Cout + 31) / 32) * J.taps * 2 * ((J.Cin_p + 31) / 32) * 64 + 255) / 256;
            J.end_pb = (int)blocks;
            if (blocks > 0x7fffffffL) return MTE_ERR_UNSUPPORTED;
            tb.j[i] = J;
        }
        if (blocks == 0) continue;
        if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(pack_multi_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, tb);
        else hipLaunchKernelGGL(pack_multi_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, tb);
    }
    return mte_check_launch();
}

// dgrad pack from an existing forward pack (same dtype): wbwd[Cin_p][taps rot180][Cout] <- wfwd[Cout][taps][Cin_p]
int mte_pack_conv_weights_bwd(const void* wfwd, void* wbwd, int Cout, int KH, int KW, int Cin_p, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!wfwd || !wbwd) return MTE_ERR_ARG;
    const int taps = KH * KW;
    const dim3 gb((Cout + 63) / 64, (taps * Cin_p + 63) / 64);
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(pack_weights_bwd_kernel<bf16_t>, gb, dim3(256), 0, stream, (const bf16_t*)wfwd, (bf16_t*)wbwd, Cout, taps, Cin_p, Cout);
    else hipLaunchKernelGGL(pack_weights_bwd_kernel<float>, gb, dim3(256), 0, stream, (const float*)wfwd, (float*)wbwd, Cout, taps, Cin_p, Cout);
    return mte_check_launch();
}

int mte_unpack_conv_wgrad(float* dw_stage, int parts, float* dw_oihw, int Cout, int Cin, int KH, int KW, int Cin_p, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dw_stage || !dw_oihw || parts < 1) return MTE_ERR_ARG;
    const dim3 grid(Cout, (Cin + 63) / 64);
    if (parts > 32) {                                                // many partials: parallel first level into the scratch slabs behind them, then <= 32 to add
        const long elems = (long)Cout * KH * KW * Cin_p;
        const int pc = ((parts + 31) / 32 + RP_PC - 1) / RP_PC * RP_PC;              // parts per first-level block: <= 32 groups
        const int groups = (parts + pc - 1) / pc;
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((elems / 4 + 255) / 256), (unsigned)groups), dim3(256), 0, stream, dw_stage, elems, parts, pc);
        dw_stage += (long)parts * elems;
        parts = groups;
    }
    hipLaunchKernelGGL(unpack_wgrad_kernel, grid, dim3(256), sizeof(float) * KH * KW * 65, stream, dw_stage, dw_oihw, Cout, Cin, KH * KW, Cin_p, parts);
    return mte_check_launch();
}

// out[N] (fp32, zeroed here) = column sums of y[M][N] (bias gradient). N multiple of 8, N/8 must divide 256 or be <= 256.
int mte_colsum(const void* y, long ld, long M, int N, float* out, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!y || !out || N % 8 != 0) return MTE_ERR_ARG;
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (N / per16 > 256) return MTE_ERR_UNSUPPORTED;
    if (mte_memset_async(out, 0, sizeof(float) * N, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    const int rstep = 256 / (N / per16);
    long want = (M + rstep - 1) / rstep;
    want = (want + 31) / 32;                          // >= 32 rows per thread
    const int grid = (int)(want > 1024 ? 1024 : (want < 1 ? 1 : want));
    if (dtype == MTE_DT_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(grid), dim3(256), sizeof(float) * N, stream, (const bf16_t*)y, ld, M, N, out);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3(grid), dim3(256), sizeof(float) * N, stream, (const float*)y, ld, M, N, out);
    return mte_check_launch();
}

}  // extern "C"
