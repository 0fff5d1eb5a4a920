// LDS-patch convolution kernels for the high-resolution, few-channel layers (C_out <= 64) -- gfx950, bf16.
//
// Why a second conv path: with N = C_out = 32/64 the implicit GEMM of conv_igemm.hip re-streams the activation
// tile once per filter tap (K = taps * C_in), i.e. 128 B of global->LDS traffic per MFMA cycle per CU for N = 32 --
// several times what L2 delivers.  These layers (pre_calc, conv1 7x7, pack1.conv 5x5, conv2.*, iconv1/2,
// unpack1/2.conv and their gradients) hold ~45 % of the network's MACs.  Here a workgroup stages the input PATCH
// of its 8 x 32-pixel output tile once per 32-channel slice -- (8+k-1) x (32+k-1) pixels x 64 B -- and all k*k taps
// read their A fragments from that patch at shifted LDS addresses, cutting global->LDS traffic by ~k*k/1.7.
//
//   forward / dgrad : weights come as 1-KiB fragment blocks [slice][tap][kk][nt][lane][8] straight from global/L2
//                     into registers (coalesced, one tap ahead); accumulators 2 x NT tiles of 32x32 per wave.
//   wgrad           : dW[n][tap][c] = sum_px dy[px][n] * x[px+tap][c]; both operands are pixel-major, so the
//                     8-consecutive-k fragments come from ds_read_b64_tr_b16; every tap's 32x32 accumulator stays in
//                     registers (taps dealt round-robin to the 4 waves) while the workgroup sweeps its pixel tiles;
//                     one fp32 atomic pass at the end.
//
// Reference ops replaced: nn.Conv2d + ConstantPad2d of Conv2D / ResidualConv (layers01.py:29-31,61) and their autograd.
#include "common.hpp"
#include <type_traits>

// In-kernel phase stamps of the forward kernels: a separate diagnostic build only (-DMTE_STAMPS, tools/patch_stamps.py); no stamp executes in
// the product or the development library.  Thread 0 of a workgroup stores s_memrealtime (100 MHz) at each phase boundary.
#ifdef MTE_STAMPS
__device__ unsigned long long g_patch_stamps[16384 * 16];
extern "C" int mtei_patch_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_patch_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#define PATCH_STAMP_DECL int stamp_i_ = 0
#define PATCH_STAMP()                                                                                                              \
    {                                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x < 16384 && stamp_i_ < 16) g_patch_stamps[blockIdx.x * 16 + stamp_i_] = __builtin_amdgcn_s_memrealtime(); \
        ++stamp_i_;                                                                                                                 \
    }
#else
#define PATCH_STAMP_DECL
#define PATCH_STAMP()
#endif

MTE_CLOCK_DEFINE(patch)

namespace {

constexpr int TH = 8, TW = 32;                     // output tile (pixels)
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

__device__ __forceinline__ int swz_off(int p, int kc) { return p * 64 + ((kc ^ ((p >> 2) & 3)) << 4); }

struct PatchArgs {
    const bf16_t* x; long ldx;
    const bf16_t* wp;                              // patch-packed weights
    const float* bias;
    bf16_t* y; long ldy;
    int B, H, W, Cin_p, N;
    int accum;                                     // 1: y += conv (fp32 sum, rounded once)
    // rank-1 term (3x3 second form only, mte_conv2d_patch_fwd_rank1): y += conv_1(nearest_up2(r1_inv)) with one more input channel's weights
    const float* r1_inv;                           // [B][H/2][W/2] fp32, or nullptr
    const float* r1_w; long r1_ws;                 // element (n, tap) at r1_w[n * r1_ws + tap]
    // second K source (first form, mte_conv2d_patch_fwd_plus1x1): y += conv_1x1(x2, wp2) -- more K-steps of the same tile at the centre tap only
    const bf16_t* x2; long ldx2; const bf16_t* wp2; int C2;
    // round 5 (mte_conv2d_patch_fwd_gn): GroupNorm(16) statistics of the tile AS STORED, one record of 32 floats (sum, sum of squares per group) per tile at
    // gn_rec + ((b * tiles_y + ty) * tiles_x + tx) * 32 -- the consumer's stand-alone statistics pass over y (252 MB at full resolution) is not needed
    float* gn_rec;
};

// ---- GroupNorm statistics in the store loop (round 5) ----------------------------------------------------------------------------------------------------------
// Both forward forms end with the tile staged in LDS as bf16 and every thread storing the 16-byte chunk column c = tid % (NT * 4) of 4 NT pixels: exactly the
// thread -> (channel chunk, pixel rows) map of the statistics pass (norm_act.hip: gn_stats_kernel).  With a.gn_rec set the thread also adds the values it stores
// (after the bf16 rounding, after the accumulation: what every later pass reads back) into 8 + 8 fp32 sums; the lanes of a wave that share c are added by
// xor butterflies, the four waves and the channels of a group in a fixed order by 32 threads: bit-reproducible, no atomics.  mte_gn_stats_from_records adds
// the tiles of a sample in tile order (fp64).  Cost: ~25 VALU per stored chunk + ~150 instructions per tile, against one pass over the tensor.
__device__ __forceinline__ void patch_gn_add(const u32x4_t& v, bool ok, float* s, float* q) {
    float f[8];
    unpack16<bf16_t>(v, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float t = ok ? f[k] : 0.f; s[k] += t; q[k] = fmaf(t, t, q[k]); }
}
template <int NT>
__device__ __forceinline__ void patch_gn_record(const PatchArgs& a, float* s, float* q, int tid, long tile, char* smem) {
    constexpr int NC = NT * 4;                                     // chunk columns of the staged tile
    const int lane = tid & 63, wave = tid >> 6, c = tid % NC;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int off = NC; off < 64; off <<= 1) { s[k] += __shfl_xor(s[k], off, 64); q[k] += __shfl_xor(q[k], off, 64); }
    __syncthreads();                                               // every thread has read its chunks of the staged tile: the buffer is free
    float* sred = (float*)smem;                                    // [4 waves][NC][16]
    if (lane < NC) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sred[(wave * NC + c) * 16 + k] = s[k]; sred[(wave * NC + c) * 16 + 8 + k] = q[k]; }
    }
    __syncthreads();
    if (tid < 32) {
        const int g = tid >> 1, which = tid & 1, gs = a.N >> 4;    // group, sum / sum of squares, channels per group (launcher: N % 16 == 0)
        float t = 0.f;
        for (int ch = g * gs; ch < (g + 1) * gs; ++ch)
#pragma unroll
            for (int w = 0; w < 4; ++w) t += sred[(w * NC + (ch >> 3)) * 16 + which * 8 + (ch & 7)];
        a.gn_rec[tile * 32 + tid] = t;
    }
}


// ---- second K source at the centre tap (EXTRA; mte_conv2d_patch_fwd_plus1x1) ---------------------------------------------------------------------------------
// y += conv_1x1(x2, wp2): for a 1x1 every input element meets the MFMA exactly once per 32-column tile, so its fragments need no LDS patch -- lane (r, h) of
// pixel row m loads the 16 bytes [channels 32 s + 16 kk + 8 h ..] of pixel (y0 + mrow0 + m, x0 + r) straight from global memory.  Issued in the PROLOGUE, beside the
// first patch of the 3x3 (one exposed memory latency serves both), two slices at a time; consumed before the main loop, whose register peak is unchanged.
// (First try: the second source as further slices of the LDS patch loop -- each 1-tap slice exposed its own patch load: +36..44 us on an 84 us launch.)
template <int MM, int Q> struct ExtraFrags { u32x4_t px[Q][MM][2], w[Q][2]; };   // Q slices of 32 channels in flight
template <int MM, int NT, int Q>
__device__ __forceinline__ void extra_load(ExtraFrags<MM, Q>& f, const PatchArgs& a, int s2, int b, int yrow0, int xcol, int h, int nsel, int lane) {
    const int cp2 = a.C2 >> 3, n2 = (a.C2 + 31) >> 5;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int sl = s2 + q;
        const bool sok = sl < n2;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4_t wv = *((const u32x4_t*)a.wp2 + lane + ((long)((sok ? sl : 0) * 2 + kk) * NT + nsel) * 64);
            f.w[q][kk] = sok ? wv : u32x4_t{0u, 0u, 0u, 0u};
            const int cc = sl * 4 + kk * 2 + h;
            const bool cok = sok && cc < cp2;
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const int yy = yrow0 + m;
                const bool ok = cok && yy < a.H;
                const u32x4_t v = *(const u32x4_t*)(a.x2 + (((long)b * a.H + (yy < a.H ? yy : a.H - 1)) * a.W + xcol) * a.ldx2 + (cok ? cc : 0) * 8);
                f.px[q][m][kk] = ok ? v : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    }
}

// the same for the 16x16x32 forms (round 6): px[q][m][ph] = chunk g16 (channels 32 sl + 8 g16 ..) of pixel x0 + 16 ph + c16, w[q][c] = the lane's 16 bytes of the
// 16-channel half c (gathered from the two fragment blocks of the 32x32x16 pack: see conv_patch_fwd2_kernel, M16)
template <int MM, int NT, int Q>
__device__ __forceinline__ void extra_load16(ExtraFrags<MM, Q>& f, const PatchArgs& a, int s2, int b, int yrow0, int x0, int c16, int g16, unsigned w16off) {
    const int cp2 = a.C2 >> 3, n2 = (a.C2 + 31) >> 5;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int sl = s2 + q;
        const bool sok = sl < n2;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u32x4_t wv = *(const u32x4_t*)((const char*)a.wp2 + (long)(sok ? sl : 0) * (2 * NT * 1024) + w16off + c * 256);
            f.w[q][c] = sok ? wv : u32x4_t{0u, 0u, 0u, 0u};
        }
        const int cc = sl * 4 + g16;
        const bool cok = sok && cc < cp2;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const int yy = yrow0 + m;
            const bool ok = cok && yy < a.H;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const u32x4_t v = *(const u32x4_t*)(a.x2 + (((long)b * a.H + (yy < a.H ? yy : a.H - 1)) * a.W + x0 + 16 * ph + c16) * a.ldx2 + (cok ? cc : 0) * 8);
                f.px[q][m][ph] = ok ? v : u32x4_t{0u, 0u, 0u, 0u};
            }
        }
    }
}

// ---- forward / dgrad --------------------------------------------------------------------------------------
// TALL (NT = 1, one 32-channel input slice): 16 x 32-pixel tile, four pixel rows per wave -- every weight fragment fetched
// from L2 feeds twice the MFMAs (with two rows per wave the 7x7 full-resolution layers pulled 6 GB of weight fragments
// per launch, ~10 TB/s of L2 bandwidth); the single slice needs only one patch buffer, so two workgroups still fit a CU.
// M16 (round 6): v_mfma_f32_16x16x32_bf16 with the operands the other way round (D = W X^T, as in the second form): a lane owns 4 consecutive output channels of
// one pixel, so the tile is staged with 8-byte LDS writes instead of 64 two-byte ones; swizzle, fragment reads and weight gather as conv_patch_fwd2_kernel's M16.
template <int K, int NT, bool TALL, bool ACC = false, bool EXTRA = false, bool M16 = false>
__global__ __launch_bounds__(256, 2) void conv_patch_fwd_kernel(PatchArgs a) {
    static_assert(!TALL || NT == 1, "tall tiles are for the 32-output kernels");
    static_assert(!EXTRA || K == 3, "the 1x1 second source rides the 3x3 kernels");
    constexpr int TH = TALL ? 16 : 8;                               // (shadows the file-level 8-row tile of the wgrad kernels)
    constexpr int PAD = K / 2, PH = TH + K - 1, PW = TW + K - 1, TAPS = K * K;
    constexpr int PCH = PH * PW * 4;                               // 16-B chunks per patch slice
    constexpr int NCH = (PCH + 255) / 256;
    constexpr int PBYTES = PH * PW * 64;
    constexpr int OSTR = M16 ? NT * 64 + 16 : NT * 64;             // bytes per staged pixel (M16: padded, the 8-byte writes of 16 pixels then go 2-way)
    constexpr int OBYTES = TH * TW * OSTR;                         // output staging (bf16 [256 px][32*NT])
    // patch buffers (tall 5x5 / 7x7 tiles: single-slice layers only).  Round 3 tried ONE buffer for the tall 3x3 tiles too (39 KB instead
    // of 78 KB of LDS per workgroup): the kernel holds 196-224 VGPRs, so two workgroups per CU is all it gets either way, and forcing
    // three or four waves per SIMD spills (72 -> 32 at 384x1280: 400 -> 694 us)
    constexpr int NBUF = (TALL && K > 3) ? 1 : 2;
    constexpr int LDS_BYTES = ((NBUF * PBYTES > OBYTES) ? NBUF * PBYTES : OBYTES);
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = a.W / TW, tiles_y = (a.H + TH - 1) / TH;
    int id = xcd_remap(blockIdx.x, tiles_x * tiles_y * a.B);
    const int tx_ = id % tiles_x; id /= tiles_x;
    const int ty_ = id % tiles_y; const int b = id / tiles_y;
    const int x0 = tx_ * TW, y0 = ty_ * TH;
    const int cpt = a.Cin_p >> 3;                                  // 16-B chunks per pixel
    const int nslices = (a.Cin_p + 31) >> 5;

    u32x4_t st[NCH];
    auto load_patch = [&](int s) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idc = tid + i * 256;
            const int p = idc >> 2, kc = idc & 3;
            const int py = p / PW, px = p - py * PW;
            const int iy = y0 + py - PAD, ix = x0 + px - PAD;
            const int cc = s * 4 + kc;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if ((PCH % 256 == 0 || idc < PCH) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && cc < cpt)
                v = *(const u32x4_t*)(a.x + (((long)b * a.H + iy) * a.W + ix) * a.ldx + cc * 8);
            st[i] = v;
        }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int idc = tid + i * 256;
            const int off = M16 ? (idc >> 2) * 64 + (((idc & 3) ^ ((idc >> 3) & 2)) << 4) : swz_off(idc >> 2, idc & 3);      // (M16: slot = chunk ^ 2 ((p >> 2) & 1))
            if (PCH % 256 == 0 || idc < PCH) *(u32x4_t*)(smem + buf * PBYTES + off) = st[i];
        }
    };

    // Wave -> sub-tile map.  NT = 1: wave w owns pixel rows 2w, 2w+1.  NT = 2: waves are split over the two 32-channel output
    // tiles (n = w >> 1) and own four pixel rows each: a wave then fetches half of the slice's weight fragments from L2
    // (every wave reading all of them made weight traffic, 1.1 GB per 64->64 launch, ~9x the activation traffic).
    constexpr int MM = (NT == 2 || TALL) ? 4 : 2;
    const int nsel = NT == 2 ? wave >> 1 : 0;
    const int mrow0 = NT == 2 ? (wave & 1) * 4 : wave * MM;
    f32x16_t acc[M16 ? 1 : MM];
#pragma unroll
    for (int m = 0; m < (M16 ? 1 : MM); ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    const int c16 = lane & 15, g16 = lane >> 4;                    // M16: pixel column / channel row of the lane, its K group (and its four output channels 4 g16 ..)
    const unsigned w16off = (unsigned)((g16 >> 1) * NT * 1024 + nsel * 1024 + ((g16 & 1) * 32 + c16) * 16);
    f32x4_t acc16[M16 ? MM : 1][2][2];                              // [pixel row][16-channel half][16-pixel half]
#pragma unroll
    for (int m = 0; m < (M16 ? MM : 1); ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) acc16[m][c][ph] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    PATCH_STAMP_DECL;
    PATCH_STAMP();
    if constexpr (EXTRA) {                                         // second K source: all of its MFMAs before the main loop (see extra_load)
        // one slice in flight: with two, this form (167 VGPRs, three workgroups per CU) drops to two workgroups per CU
        const int n2 = (a.C2 + 31) >> 5;
        ExtraFrags<MM, 1> ef;
        if constexpr (M16) extra_load16<MM, NT, 1>(ef, a, 0, b, y0 + mrow0, x0, c16, g16, w16off);
        else extra_load<MM, NT, 1>(ef, a, 0, b, y0 + mrow0, x0 + r, h, nsel, lane);
        for (int s2 = 0; s2 < n2; ++s2) {
            if constexpr (M16) {
#pragma unroll
                for (int m = 0; m < MM; ++m)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
                            acc16[m][c][ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ef.w[0][c]), __builtin_bit_cast(bf16x8_t, ef.px[0][m][ph]), acc16[m][c][ph], 0, 0, 0);
            } else {
#pragma unroll
                for (int m = 0; m < MM; ++m)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ef.px[0][m][kk]), __builtin_bit_cast(bf16x8_t, ef.w[0][kk]), acc[m], 0, 0, 0);
            }
            asm volatile("" ::: "memory");                         // (the next slice's loads stay behind these MFMAs: one set of fragment registers)
            if (s2 + 1 < n2) {
                if constexpr (M16) extra_load16<MM, NT, 1>(ef, a, s2 + 1, b, y0 + mrow0, x0, c16, g16, w16off);
                else extra_load<MM, NT, 1>(ef, a, s2 + 1, b, y0 + mrow0, x0 + r, h, nsel, lane);
            }
        }
        asm volatile("" ::: "memory");                             // (the patch staging registers are not live beside the fragment set)
    }
    const u32x4_t* wl = (const u32x4_t*)a.wp + lane;               // fragment block = 64 lanes x 16 B
    // weight fragments come straight from L2 (hundreds of cycles): PD taps in flight in a register ring; fragment kk (M16: 16-channel half) of tap t of slice s
    constexpr int PD = TAPS < 4 ? TAPS : 4;
    u32x4_t bq[PD][2];
    auto wload = [&](int s, int t, int kk) {
        if constexpr (M16) return *(const u32x4_t*)((const char*)a.wp + (long)(s * TAPS + t) * (2 * NT * 1024) + w16off + kk * 256);
        else return wl[(((long)(s * TAPS + t) * 2 + kk) * NT + nsel) * 64];
    };
    // one tap: fragments of the MM pixel rows at the tap's shift, MM x 2 (M16: x 4) MFMAs against ring slot d
    auto tap_mfmas = [&](const char* P, int t, int d) {
        const int dy = t / K, dx = t - dy * K;
        u32x4_t fa[MM][2];
        if constexpr (M16) {
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int p = (mrow0 + m + dy) * PW + dx + 16 * ph + c16;
                    fa[m][ph] = *(const u32x4_t*)(P + p * 64 + ((g16 ^ ((p >> 1) & 2)) << 4));
                }
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph)
                        acc16[m][c][ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq[d][c]),
                                                                                  __builtin_bit_cast(bf16x8_t, fa[m][ph]), acc16[m][c][ph], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const int p = (mrow0 + m + dy) * PW + dx + r;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fa[m][kk] = *(const u32x4_t*)(P + swz_off(p, 2 * kk + h));
            }
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[m][kk]),
                                                                     __builtin_bit_cast(bf16x8_t, bq[d][kk]), acc[m], 0, 0, 0);
        }
    };
    load_patch(0);
    store_patch(0);
    __syncthreads();
    PATCH_STAMP();
    MTE_CLOCK_BEGIN()
    for (int s = 0; s < nslices; ++s) {
        const char* P = smem + (s & 1) * PBYTES;
        if (s + 1 < nslices) load_patch(s + 1);
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bq[d][kk] = wload(s, d, kk);
        if constexpr (TAPS > PD) __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1                                                   // (round 5, end: unrolled over the 3x3 taps this loop needs 254 instead of 167 VGPRs -- two workgroups per CU instead of
                                                                   //  three -- and the 64 -> 64 @192x640 forward went from 92 to 107 us; round 6 tried again with pinned loads: 248)
        for (int t0 = 0; t0 < TAPS; t0 += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int t = t0 + d;
                if (t < TAPS) tap_mfmas(P, t, d);                  // wave-uniform
                // Refill, UNCONDITIONAL and pinned here (round 6, tools/loopaudit.py on the round-5 loop): inside the `t + PD < TAPS` branch the compiler could not
                // count the load and waited for vmcnt(1) / vmcnt(0) at the head of every group of PD taps -- the ring bought no prefetch across groups.  Past the
                // last tap the slot re-reads the last tap's fragment (an L2 hit nobody waits for).
                if constexpr (TAPS > PD) {                         // (a 1x1 has nothing to refill)
                    __builtin_amdgcn_sched_barrier(0);
                    const int tn = t + PD < TAPS ? t + PD : TAPS - 1;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) bq[d][kk] = wload(s, tn, kk);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        PATCH_STAMP();
        if (s + 1 < nslices) store_patch((s + 1) & 1);
        __syncthreads();
        PATCH_STAMP();
    }
    MTE_CLOCK_END(patch)
    // ---- epilogue: stage the tile as bf16 [pixel][N] in LDS, then 16-byte coalesced stores
    constexpr int NB = OSTR;                                       // bytes per staged pixel
    if constexpr (M16) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch0 = nsel * 32 + 16 * c + 4 * g16;          // (N % 8 == 0: the lane's four channels are inside or outside as a whole)
            float bv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (a.bias && ch0 < a.N) ? a.bias[ch0 + e] : 0.f;
#pragma unroll
            for (int m = 0; m < MM; ++m)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    u32x2_t v;
                    v[0] = pack2bf(acc16[m][c][ph][0] + bv[0], acc16[m][c][ph][1] + bv[1]);
                    v[1] = pack2bf(acc16[m][c][ph][2] + bv[2], acc16[m][c][ph][3] + bv[3]);
                    *(u32x2_t*)(smem + ((mrow0 + m) * TW + 16 * ph + c16) * NB + ch0 * 2) = v;
                }
        }
    } else {
        const int ch = nsel * 32 + r;
        const float bv = (a.bias && ch < a.N) ? a.bias[ch] : 0.f;
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int px = (e & 3) + 8 * (e >> 2) + 4 * h;
                *(bf16_t*)(smem + ((mrow0 + m) * TW + px) * NB + ch * 2) = f2bf(acc[m][e] + bv);
            }
    }
    __syncthreads();
    constexpr int OCH = TH * TW * NT * 4;                          // 16-B chunks of the tile
    const int cpp = a.N >> 3;                                      // valid chunks per pixel
    const int c = tid % (NT * 4);
    const bool gnrec = a.gn_rec != nullptr;                        // (wave-uniform)
    float gs_[8], gq_[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gs_[k] = 0.f; gq_[k] = 0.f; }
    if constexpr (ACC) {
        // accumulating: branch-free, every LDS and global read of the thread's pixels issued before the first use (see the second form's epilogue;
        // same-box A/B, 32 -> 64 at 384 x 1280: the guarded read -> wait -> add -> store loop cost 100 us over the plain store)
        constexpr int NI = OCH / 256;
        u32x4_t v16[NI], vold[NI];
        long off[NI];
        bool ok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pix = tid / (NT * 4) + i * (256 / (NT * 4));
            const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
            ok[i] = yy < a.H && c < cpp;
            off[i] = (((long)b * a.H + (yy < a.H ? yy : a.H - 1)) * a.W + xx) * a.ldy + (c < cpp ? c : 0) * 8;
            v16[i] = *(const u32x4_t*)(smem + pix * NB + c * 16);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) vold[i] = *(const u32x4_t*)(a.y + off[i]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float vn[8], vo[8];
            unpack16<bf16_t>(v16[i], vn);
            unpack16<bf16_t>(vold[i], vo);
#pragma unroll
            for (int k = 0; k < 8; ++k) vn[k] += vo[k];
            const u32x4_t pk = pack16<bf16_t>(vn);
            if (ok[i]) *(u32x4_t*)(a.y + off[i]) = pk;
            if (gnrec) patch_gn_add(pk, ok[i], gs_, gq_);
        }
    } else {
#pragma unroll
        for (int i = 0; i < OCH / 256; ++i) {
            const int pix = tid / (NT * 4) + i * (256 / (NT * 4));
            const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
            const u32x4_t pk = *(const u32x4_t*)(smem + pix * NB + c * 16);
            if (yy < a.H && c < cpp) *(u32x4_t*)(a.y + (((long)b * a.H + yy) * a.W + xx) * a.ldy + c * 8) = pk;
            if (gnrec) patch_gn_add(pk, yy < a.H && c < cpp, gs_, gq_);
            asm volatile("" ::: "memory");                         // one read -> store per iteration: with the reads hoisted the stores leave in one burst (2-4 % slower)
        }
    }
    if (gnrec) patch_gn_record<NT>(a, gs_, gq_, tid, ((long)b * tiles_y + ty_) * tiles_x + tx_, smem);
    PATCH_STAMP();
}

// ---- forward / dgrad, second form (round 3) ------------------------------------------------------------------------
// In-kernel stamps of the first form (3 slices of a 3x3 filter, 16 x 32 tile, two workgroups per CU) put a tile at 21.8 us: 2.9 us of
// cold prologue, 2.4 us of epilogue, and tap loops at ~55 % of the MFMA rate that take 6.1 / 4.9 / 4.0 us -- the first two longer because
// vmcnt retires in order and every weight fragment requested after the next slice's patch loads waits out that whole HBM fetch.  Changes:
//   * the patch travels global -> LDS by LDS-DMA (no staging registers, no ds_write): a wave instruction fills 1 KiB of the patch image,
//     the XOR swizzle is applied to the SOURCE chunk a lane asks for, out-of-image chunks read a zero chunk (no branch, so every wait is
//     counted exactly), and the 3x3 kernels request all nine taps' weight fragments BEFORE the DMA of the next slice is issued;
//   * the A fragments slide down the patch: for a fixed tap column the wave keeps MM + 1 pixel rows in registers and each tap row reads
//     ONE new row (2 ds_read_b128 instead of 2 MM), issued a whole tap ahead of its first MFMA;
//   * the MFMA operands are swapped (D = W * X^T): a lane then owns 4 CONSECUTIVE output channels of one pixel per accumulator quad, so
//     the tile is staged with 16 ds_write_b64 per thread instead of 64 two-byte writes that all fell on two LDS banks.
// M16 (round 6): the same tile on v_mfma_f32_16x16x32_bf16 -- per tap and slice 4 MM instructions of 16 cycles instead of 2 MM of 32: equal matrix-pipe cycles,
// but the chip HOLDS a higher clock on that shape (MI355X_MICROARCH.md 'DVFS give-back' item 7; measured here: profiles/r06_inloop_clock.txt).  A = weights
// (16 channels x one whole 32-channel slice: lane (row, g) gathers its 16 bytes from the two fragment blocks of the 32x32x16 pack, K order = channel order),
// B = 16 pixels (lane (column, g) reads chunk g of pixel column: one ds_read_b128; the patch image then takes the swizzle slot = chunk ^ 2 ((p >> 2) & 1),
// conflict-free for that read at every alignment -- exhaustive check in tests/test_patch_swizzle.py), D: lane holds 4 consecutive channels of one pixel.
template <int K, int NT, bool TALL, bool R1 = false, bool ACC = false, bool EXTRA = false, bool M16 = false>
__global__ __launch_bounds__(256, 2) void conv_patch_fwd2_kernel(PatchArgs a) {
    static_assert(!TALL || NT == 1, "tall tiles are for the 32-output kernels");
    static_assert(!R1 || K == 3, "the rank-1 term is a 3x3 stencil");
    constexpr int TH = TALL ? 16 : 8;
    constexpr int PAD = K / 2, PH = TH + K - 1, PW = TW + K - 1, TAPS = K * K;
    constexpr int PCH = PH * PW * 4;                               // 16-B chunks of one patch slice
    constexpr int NDMA = (PCH + 255) / 256;                        // DMA instructions per wave and slice
    constexpr int PBYTES = NDMA * 256 * 16;                        // (whole instructions: the tail lanes deposit zero chunks behind the patch)
    constexpr int NB = NT * 64, OSTR = NB + 16;                    // bytes per staged pixel, padded stride (b64 writes of 16 pixels: 2-way)
    constexpr int OBYTES = TH * TW * OSTR;
    constexpr int NBUF = (TALL && K > 3) ? 1 : 2;                  // tall 5x5 / 7x7 tiles: single-slice layers only
    constexpr int LDS_BYTES = NBUF * PBYTES > OBYTES ? NBUF * PBYTES : OBYTES;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    // (round 5 added a 128-byte __shared__ array beside this buffer: 82,048 bytes no longer fit twice into a CU's 163,840 and every 64 -> 32 / 72 -> 32 3x3
    //  launch ran at ONE workgroup per CU -- profiles/r06_lds_fix.txt)
    static_assert(!(K == 3 && TALL) || LDS_BYTES <= 81920, "the tall 3x3 form must keep two workgroups per CU");
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = a.W / TW, tiles_y = (a.H + TH - 1) / TH;
    int id = xcd_remap(blockIdx.x, tiles_x * tiles_y * a.B);
    const int tx_ = id % tiles_x; id /= tiles_x;
    const int ty_ = id % tiles_y; const int b = id / tiles_y;
    const int x0 = tx_ * TW, y0 = ty_ * TH;
    const int cpt = a.Cin_p >> 3;                                  // 16-B chunks per pixel
    const int nslices = (a.Cin_p + 31) >> 5;

    // ---- DMA plan of this thread: chunk idc = (i * 4 + wave) * 64 + lane of the patch image; LDS slot (idc & 3) of pixel p = idc >> 2 holds
    // source chunk kc = slot ^ swz(p)
    // (buffer-descriptor form of the DMA, as in conv_igemm.hip: the per-lane byte offset is fixed per tile, the slice advance is a wave-uniform
    //  scalar offset, and a chunk outside the image asks for an offset beyond the descriptor's range, which returns zeros.  The compiler also
    //  keeps LDS reads that follow a global_load_lds behind vmcnt(0) -- it cannot tell the two patch buffers apart -- but not these.)
    constexpr unsigned OOB = 0xfffffff0u;
    unsigned doff[NDMA];                                           // byte offset of the chunk for slice 0 (launcher: the tensor is < 2 GiB)
    unsigned dkc = 0;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int idc = (i * 4 + wave) * 64 + lane;
        const int p = idc >> 2, kc = (idc & 3) ^ (M16 ? ((p >> 1) & 2) : ((p >> 2) & 3));
        const int py = p / PW, px = p - py * PW;
        const int iy = y0 + py - PAD, ix = x0 + px - PAD;
        const bool ok = idc < PCH && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        doff[i] = ok ? (unsigned)(((((long)b * a.H + iy) * a.W + ix) * a.ldx + kc * 8) * 2) : OOB;
        dkc |= (unsigned)kc << (2 * i);
    }
    auto dma_patch = [&](int s, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)   // buffer-resource builtins exist only in the device pass
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((((long)a.B * a.H * a.W - 1) * a.ldx + a.Cin_p) * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int cc = s * 4 + (int)((dkc >> (2 * i)) & 3u);
            const unsigned voff = cc < cpt ? doff[i] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + buf * PBYTES + (i * 4 + wave) * 1024), 16, voff, s * 64, 0, 0);
        }
#else
        (void)s; (void)buf;
#endif
    };

    constexpr int MM = (NT == 2 || TALL) ? 4 : 2;                  // pixel rows per wave
    const int nsel = NT == 2 ? wave >> 1 : 0;
    const int mrow0 = NT == 2 ? (wave & 1) * 4 : wave * MM;
    f32x16_t acc[M16 ? 1 : MM];
#pragma unroll
    for (int m = 0; m < (M16 ? 1 : MM); ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    const int c16 = lane & 15, g16 = lane >> 4;                    // M16: pixel column / channel row of the lane, its K group (and its four output channels 4 g16 ..)
    f32x4_t acc16[M16 ? MM : 1][2][2];                              // [pixel row][16-channel half][16-pixel half]
#pragma unroll
    for (int m = 0; m < (M16 ? MM : 1); ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) acc16[m][c][ph] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- rank-1 term (R1): one more input channel given as a LOW-resolution map (the decoder's up-sampled inverse depth), as ONE 16-deep MFMA step that
    // initialises the accumulators.  The 3x3 window of the up-sampled map around a pixel covers only 2 x 2 pixels of the low-resolution map: for an even row
    // the window rows (Y-1, Y, Y+1) are low rows (k-1, k, k), for an odd row (k, k, k+1), and the same along x -- so the term is
    //     sum over a, b in {0, 1} of  Wc[row parity][column parity][a][b][n] * map[base row + a][base column + b]
    // with COMBINED weights Wc = the sum of the 1, 2 or 4 taps that fall on low pixel (a, b) (the zero padding of the full-resolution image is a zero border
    // of the low-resolution one: H and W are even).  As a GEMM step: K index = (parity class) * 4 + (a * 2 + b) -- exactly 16 -- with the map's four values
    // in the slots of the pixel's own class and zeros elsewhere.  Loaded beside the first patch, consumed before the main loop: no register lives across it.
    constexpr int LR = MM / 2 + 2;                                  // low-resolution rows under the wave's MM pixel rows (y0 + mrow0 is even)
    // (M16: the lane's channel depends on the 16-channel half c, its pixel on the 16-pixel half ph: two sets each, index q)
    constexpr int RQ = (R1 && M16) ? 2 : 1;
    float r1t[RQ][R1 ? 9 : 1], r1m[RQ][R1 ? LR : 1][2];
    if constexpr (R1) {
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int n = M16 ? nsel * 32 + 16 * q + c16 : nsel * 32 + r;
            const float* wc = a.r1_w + (n < a.N ? (long)n * a.r1_ws : 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) r1t[q][t] = wc[t];
            const int hl = a.H >> 1, wl_ = a.W >> 1;
            const int pxl = M16 ? 16 * q + c16 : r;
            const int ly0 = ((y0 + mrow0) >> 1) - 1, lx0 = ((x0 + pxl + 1) >> 1) - 1;   // low pixel of window row / column -1 of the wave's first row / the lane's pixel
#pragma unroll
            for (int i = 0; i < LR; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ly = ly0 + i, lx = lx0 + j;
                    const bool ok = (unsigned)ly < (unsigned)hl && (unsigned)lx < (unsigned)wl_;
                    const float v = a.r1_inv[ok ? ((long)b * hl + ly) * wl_ + lx : 0];
                    r1m[q][i][j] = ok ? v : 0.f;
                }
        }
    }
    auto rank1_step = [&]() {
        if constexpr (R1) {
            // weights operand: lane (r, h) holds channel n, K slots 8 h .. 8 h + 7 = row parity h, column parity px = slot >> 2, (a, b) = slot & 3
            // (M16: K = 32 with the 16 slots in K groups 0 and 1 -- the lane's row parity is g16 -- and zeros in groups 2 and 3)
            const int hh = M16 ? (g16 & 1) : h;
            const bool kreal = M16 ? g16 < 2 : true;
            u32x4_t wf[RQ];
#pragma unroll
            for (int q = 0; q < RQ; ++q) {
                float R[2][3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    R[0][kx] = r1t[q][kx] + (hh ? r1t[q][3 + kx] : 0.f);          // low row a = 0: tap row 0 (+ tap row 1 on odd rows)
                    R[1][kx] = r1t[q][6 + kx] + (hh ? 0.f : r1t[q][3 + kx]);      // low row a = 1: tap row 2 (+ tap row 1 on even rows)
                }
                const bool nok = kreal && (M16 ? nsel * 32 + 16 * q + c16 : nsel * 32 + r) < a.N;
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        const float c0 = px ? R[aa][0] + R[aa][1] : R[aa][0], c1 = px ? R[aa][2] : R[aa][1] + R[aa][2];
                        wf[q][px * 2 + aa] = nok ? pack2bf(c0, c1) : 0u;
                    }
            }
#pragma unroll
            for (int m = 0; m < MM; ++m) {
                const int la = (m + 1) >> 1;
                const bool mine = kreal && hh == (m & 1);            // the pixel row's parity class sits in this half of the K slots
                const bool odd = (M16 ? c16 : r) & 1;
                u32x4_t uf[RQ];
#pragma unroll
                for (int q = 0; q < RQ; ++q) {
                    const unsigned w01 = pack2bf(r1m[q][la][0], r1m[q][la][1]), w23 = pack2bf(r1m[q][la + 1][0], r1m[q][la + 1][1]);
                    uf[q][0] = (mine && !odd) ? w01 : 0u; uf[q][1] = (mine && !odd) ? w23 : 0u;
                    uf[q][2] = (mine && odd) ? w01 : 0u;  uf[q][3] = (mine && odd) ? w23 : 0u;
                }
                if constexpr (M16) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
                            acc16[m][c][ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[c]), __builtin_bit_cast(bf16x8_t, uf[ph]), acc16[m][c][ph], 0, 0, 0);
                } else {
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[0]), __builtin_bit_cast(bf16x8_t, uf[0]), acc[m], 0, 0, 0);
                }
            }
        }
    };

    PATCH_STAMP_DECL;
    PATCH_STAMP();
    ExtraFrags<EXTRA ? MM : 1, 2> ef;
    const unsigned w16off = (unsigned)((g16 >> 1) * NT * 1024 + nsel * 1024 + ((g16 & 1) * 32 + c16) * 16);      // (M16: see wfrag)
    if constexpr (EXTRA) {
        if constexpr (M16) extra_load16<MM, NT, 2>(ef, a, 0, b, y0 + mrow0, x0, c16, g16, w16off);
        else extra_load<MM, NT, 2>(ef, a, 0, b, y0 + mrow0, x0 + r, h, nsel, lane);
    }
    dma_patch(0, 0);
    // weight fragments: block (slice, tap, kk, nt) of 64 lanes x 16 B, straight from L2
    const u32x4_t* wl = (const u32x4_t*)a.wp + lane + nsel * 64;
    const unsigned wlane = (unsigned)(lane + nsel * 64) * 16u;      // this lane's byte offset inside a fragment block
    // M16: second index = 16-channel half c; lane (row, g) takes channel nsel * 32 + 16 c + row, K = 8 g .. 8 g + 7 of the slice = lane (g & 1) * 32 + 16 c + row of block kk = g >> 1
    auto wfrag = [&](int s, int t, int kk) {
        if constexpr (M16) return *(const u32x4_t*)((const char*)a.wp + (long)(s * TAPS + t) * (2 * NT * 1024) + w16off + kk * 256);
        else return wl[((long)(s * TAPS + t) * 2 + kk) * (NT * 64)];
    };
    // ring of fragments in VISITING order (tap column dx outer, tap row dy inner): the 3x3 / 1x1 kernels hold a whole slice (<= 72 VGPRs, all
    // requested before the next slice's DMA), the 5x5 / 7x7 kernels one tap column (slot dy, refilled with the next column right after use)
    constexpr bool WHOLE = TAPS <= 9;
    constexpr int RS = WHOLE ? TAPS : K;
    u32x4_t bq[RS][2];
    if constexpr (!WHOLE) {
#pragma unroll
        for (int dy = 0; dy < K; ++dy)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) bq[dy][kk] = wfrag(0, dy * K, kk);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WHOLE ? 0 : 2 * K) : "memory");     // the patch of slice 0 (the ring may still be in flight)
    __syncthreads();
    PATCH_STAMP();
    MTE_CLOCK_BEGIN()
    rank1_step();
    if constexpr (EXTRA) {                                         // second K source (see extra_load; this form's MFMA takes the weights first)
        const int n2 = (a.C2 + 31) >> 5;
        for (int s2 = 0; s2 < n2; s2 += 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    if constexpr (M16) {
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int ph = 0; ph < 2; ++ph)
                                acc16[m][c][ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ef.w[q][c]), __builtin_bit_cast(bf16x8_t, ef.px[q][m][ph]), acc16[m][c][ph], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ef.w[q][kk]), __builtin_bit_cast(bf16x8_t, ef.px[q][m][kk]), acc[m], 0, 0, 0);
                    }
                }
            if (s2 + 2 < n2) {
                if constexpr (M16) extra_load16<MM, NT, 2>(ef, a, s2 + 2, b, y0 + mrow0, x0, c16, g16, w16off);
                else extra_load<MM, NT, 2>(ef, a, s2 + 2, b, y0 + mrow0, x0 + r, h, nsel, lane);
            }
        }
    }

    constexpr int WR = MM + 1;                                     // window rows in registers
    u32x4_t win[WR][2];
    // bias of the 16 channels this lane owns, as four 16-byte loads issued behind the last slice's MFMAs (its wait + barrier cover them; as 16
    // guarded scalar loads at the top of the epilogue they cost the tile 1.3-2 us)
    f32x4_t bv4[4];
    auto load_bias = [&]() {
#pragma unroll
        for (int g = 0; g < (M16 ? 2 : 4); ++g) {
            const int ch0 = M16 ? nsel * 32 + 16 * g + 4 * g16 : nsel * 32 + 8 * g + 4 * h;             // (N % 8 == 0: a group of four is inside or outside as a whole)
            const f32x4_t v = *(const f32x4_t*)((a.bias ? a.bias : (const float*)a.wp) + (ch0 < a.N ? ch0 : 0));
            bv4[g] = (a.bias && ch0 < a.N) ? v : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    // pixel row J of this wave's patch rows (tap column DX) -> window slot
#define PF2_LDROW(J, SLOT, DX)                                                                                         \
    { if constexpr (M16) {                                                                                               \
        _Pragma("unroll") for (int ph = 0; ph < 2; ++ph) {                                                             \
            const int p_ = (mrow0 + (J)) * PW + (DX) + 16 * ph + c16;                                                  \
            win[SLOT][ph] = *(const u32x4_t*)(P + p_ * 64 + ((g16 ^ ((p_ >> 1) & 2)) << 4));                           \
        }                                                                                                              \
    } else {                                                                                                           \
        const int p_ = (mrow0 + (J)) * PW + (DX) + r;                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) win[SLOT][kk] = *(const u32x4_t*)(P + swz_off(p_, 2 * kk + h)); \
    } }
#define PF2_MFMAS(DY, BQ)                                                                                              \
    { if constexpr (M16) {                                                                                               \
        _Pragma("unroll") for (int m = 0; m < MM; ++m)                                                                 \
            _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                              \
                _Pragma("unroll") for (int ph = 0; ph < 2; ++ph)                                                       \
                    acc16[m][c][ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, (BQ)[c]),   \
                                          __builtin_bit_cast(bf16x8_t, win[((DY) + m) % WR][ph]), acc16[m][c][ph], 0, 0, 0); \
    } else {                                                                                                           \
        _Pragma("unroll") for (int m = 0; m < MM; ++m)                                                                 \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                           \
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, (BQ)[kk]),               \
                                                                 __builtin_bit_cast(bf16x8_t, win[((DY) + m) % WR][kk]), acc[m], 0, 0, 0); \
    } }
    if constexpr (WHOLE) {
        // The compiler does not count LDS-DMA in its s_waitcnt bookkeeping, and beside a DMA it drains vmcnt(0) in front of every use of an
        // ordinary load: the fragment loads of these kernels are hidden from it in asm statements and waited for by hand.  Fragment pair Q of
        // the visiting order has landed when at most the operations issued after it are outstanding: the later pairs and, when there is a
        // next slice (MORE), its NDMA patch instructions.  ONE wait statement per pair, in straight-line code between the load and the MFMAs
        // and naming the pair as read-write operands: with the wait in two arms of a branch the compiler copied the (not yet loaded)
        // registers in front of one of them.
#define PF2_TAP(DX, DY, MORE)                                                                                          \
        if constexpr ((DX) < K && (DY) < K) {                                                                          \
            constexpr int Q = ((DX) < K && (DY) < K) ? (DX) * K + (DY) : 0;                                            \
            if ((DY) + 1 < K) PF2_LDROW((DY) + MM, ((DY) + MM) % WR, DX)                                               \
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(bq[Q][0]), "+v"(bq[Q][1]) : "n"(2 * (TAPS - 1 - Q) + ((MORE) ? NDMA : 0)) : "memory"); \
            PF2_MFMAS(DY, bq[Q])                                                                                       \
        }
#define PF2_COLUMN(DX, MORE)                                                                                           \
        if constexpr ((DX) < K) {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < MM; ++j) PF2_LDROW(j, j, DX)                                         \
            PF2_TAP(DX, 0, MORE) PF2_TAP(DX, 1, MORE) PF2_TAP(DX, 2, MORE)                                             \
        }
#define PF2_SLICE(S, MORE)                                                                                             \
        {                                                                                                              \
            const char* P = smem + (((S) & 1) * PBYTES);                                                               \
            _Pragma("unroll") for (int dx = 0; dx < K; ++dx)                                                           \
                _Pragma("unroll") for (int dy = 0; dy < K; ++dy)                                                       \
                    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                 \
                        /* scalar base of the fragment block + one per-lane byte offset for all of them (18 VGPR pointer pairs otherwise) */ \
                        const char* blk = M16 ? (const char*)a.wp + (long)((S) * TAPS + dy * K + dx) * (2 * NT * 1024) + kk * 256 /* half kk of the tap's two blocks */ \
                                              : (const char*)a.wp + (((long)((S) * TAPS + dy * K + dx) * 2 + kk) * NT) * 1024;   \
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(bq[dx * K + dy][kk]) : "v"(M16 ? w16off : wlane), "s"(blk) : "memory"); \
                    }                                                                                                  \
            if (MORE) dma_patch((S) + 1, ((S) + 1) & 1);                                                               \
            PF2_COLUMN(0, MORE) PF2_COLUMN(1, MORE) PF2_COLUMN(2, MORE)                                                \
            if (!(MORE)) load_bias();                                                                                   \
            PATCH_STAMP();                                                                                             \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* the next slice's patch has landed */            \
            __syncthreads();                                                                                           \
            PATCH_STAMP();                                                                                             \
        }
        static_assert(NBUF == 2, "the whole-slice kernels double-buffer the patch");
        for (int s = 0; s + 1 < nslices; ++s) PF2_SLICE(s, true)
        PF2_SLICE(nslices - 1, false)
#undef PF2_SLICE
#undef PF2_COLUMN
#undef PF2_TAP
    } else {
        for (int s = 0; s < nslices; ++s) {
            const char* P = smem + (NBUF == 2 ? (s & 1) * PBYTES : 0);
            const bool more = s + 1 < nslices;                     // (wave-uniform)
            if (more) dma_patch(s + 1, (s + 1) & 1);
            auto taps_of_column = [&](int dx, bool last_column) {
#pragma unroll
                for (int j = 0; j < MM; ++j) PF2_LDROW(j, j, dx)
#pragma unroll
                for (int dy = 0; dy < K; ++dy) {
                    if (dy + 1 < K) PF2_LDROW(dy + MM, (dy + MM) % WR, dx)     // next tap row's new pixel row, a tap ahead of its first MFMA
                    PF2_MFMAS(dy, bq[dy])
                    // refill: the same tap row of the next column, or of the next slice's first column.  The scheduling barriers pin the loads HERE (round 6: the
                    // compiler had sunk all 2 K refills of a column behind its last MFMA -- the ring then held a column's fragments for ~25 instructions instead of
                    // a column's worth of MFMAs, and every column opened with an exposed L2 round trip: tools/loopaudit.py, profiles/r06_ring_refills.txt)
                    __builtin_amdgcn_sched_barrier(0);
                    if (!last_column) {
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) bq[dy][kk] = wfrag(s, dy * K + dx + 1, kk);
                    } else if (more) {
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) bq[dy][kk] = wfrag(s + 1, dy * K, kk);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll 1
            for (int dx = 0; dx < K - 1; ++dx) taps_of_column(dx, false);
            taps_of_column(K - 1, true);
            if (!more) load_bias();
            PATCH_STAMP();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next slice's patch has landed (and nothing else is outstanding)
            __syncthreads();
            PATCH_STAMP();
        }
    }
#undef PF2_MFMAS
#undef PF2_LDROW
    MTE_CLOCK_END(patch)

    // ---- epilogue: lane (r, h) holds, for pixel r of row m, channels nsel*32 + 8g + 4h + (0..3) in acc[m][4g .. 4g+3]
    // (M16: lane (c16, g16) holds, for pixel 16 ph + c16 of row m, channels nsel*32 + 16 c + 4 g16 + (0..3) in acc16[m][c][ph])
    if constexpr (M16) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    u32x2_t v;
                    v[0] = pack2bf(acc16[m][c][ph][0] + bv4[c][0], acc16[m][c][ph][1] + bv4[c][1]);
                    v[1] = pack2bf(acc16[m][c][ph][2] + bv4[c][2], acc16[m][c][ph][3] + bv4[c][3]);
                    *(u32x2_t*)(smem + ((mrow0 + m) * TW + 16 * ph + c16) * OSTR + (nsel * 32 + 16 * c + 4 * g16) * 2) = v;
                }
    } else {
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2_t v;
                v[0] = pack2bf(acc[m][4 * g] + bv4[g][0], acc[m][4 * g + 1] + bv4[g][1]);
                v[1] = pack2bf(acc[m][4 * g + 2] + bv4[g][2], acc[m][4 * g + 3] + bv4[g][3]);
                *(u32x2_t*)(smem + ((mrow0 + m) * TW + r) * OSTR + (nsel * 32 + 8 * g + 4 * h) * 2) = v;
            }
    }
    __syncthreads();
    constexpr int OCH = TH * TW * NT * 4;                          // 16-B chunks of the tile
    const int cpp = a.N >> 3;                                      // valid chunks per pixel
    const int c = tid % (NT * 4);
    const bool gnrec = a.gn_rec != nullptr;                        // (wave-uniform)
    float gs_[8], gq_[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gs_[k] = 0.f; gq_[k] = 0.f; }
    if constexpr (ACC) {
        // accumulating (a consumer's gradient added onto another's, or onto a term written first; an instantiation of its own: as a run-time branch beside
        // the plain loop it cost the tall plain kernels 2 %): branch-free, with the LDS and global reads of all of the
        // thread's NI pixels issued before the first use.  As a loop of guarded read -> wait -> add -> store bodies the epilogue was NI dependent round trips to
        // memory (same-box A/B, 64 -> 32 at 384 x 1280: 283 -> 256 us; 96 -> 64 at 192 x 640: 141 -> 131 us).  The plain store loop below keeps its guarded form:
        // hoisting its LDS reads the same way cost the tall 3 x 3 kernel 8 % (230 -> 250 us) -- the stores then leave in one burst at the end of the tile
        constexpr int NI = OCH / 256;
        u32x4_t v16[NI], vold[NI];
        long off[NI];
        bool ok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pix = tid / (NT * 4) + i * (256 / (NT * 4));
            const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
            ok[i] = yy < a.H && c < cpp;
            off[i] = (((long)b * a.H + (yy < a.H ? yy : a.H - 1)) * a.W + xx) * a.ldy + (c < cpp ? c : 0) * 8;
            v16[i] = *(const u32x4_t*)(smem + pix * OSTR + c * 16);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) vold[i] = *(const u32x4_t*)(a.y + off[i]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float vn[8], vo[8];
            unpack16<bf16_t>(v16[i], vn);
            unpack16<bf16_t>(vold[i], vo);
#pragma unroll
            for (int k = 0; k < 8; ++k) vn[k] += vo[k];
            const u32x4_t pk = pack16<bf16_t>(vn);
            if (ok[i]) *(u32x4_t*)(a.y + off[i]) = pk;
            if (gnrec) patch_gn_add(pk, ok[i], gs_, gq_);
        }
    } else {
#pragma unroll
        for (int i = 0; i < OCH / 256; ++i) {
            const int pix = tid / (NT * 4) + i * (256 / (NT * 4));
            const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
            const u32x4_t pk = *(const u32x4_t*)(smem + pix * OSTR + c * 16);
            if (yy < a.H && c < cpp) *(u32x4_t*)(a.y + (((long)b * a.H + yy) * a.W + xx) * a.ldy + c * 8) = pk;
            if (gnrec) patch_gn_add(pk, yy < a.H && c < cpp, gs_, gq_);
            asm volatile("" ::: "memory");                         // one read -> store per iteration (see the first form)
        }
    }
    if (gnrec) patch_gn_record<NT>(a, gs_, gq_, tid, ((long)b * tiles_y + ty_) * tiles_x + tx_, smem);
    PATCH_STAMP();
}

// generic pack [N][taps][Cin_p] -> fragment blocks [slice][tap][kk][nt][lane = h*32 + r][8]
__global__ void repack_patch_kernel(const bf16_t* __restrict__ wg, bf16_t* __restrict__ wp, int N, int taps, int Cin_p, int NT) {
    const int nslices = (Cin_p + 31) >> 5;
    const long total = (long)nslices * taps * 2 * NT * 64 * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7); long t = i >> 3;
        const int lane = (int)(t & 63); t >>= 6;
        const int nt = (int)(t % NT); t /= NT;
        const int kk = (int)(t & 1); t >>= 1;
        const int tap = (int)(t % taps); const int s = (int)(t / taps);
        const int n = nt * 32 + (lane & 31), c = s * 32 + kk * 16 + (lane >> 5) * 8 + j;
        wp[i] = (n < N && c < Cin_p) ? wg[((long)n * taps + tap) * Cin_p + c] : (bf16_t)0;
    }
}

// ---- wgrad ---------------------------------------------------------------------------------------------------
struct PatchWgradArgs {
    const bf16_t* x; long ldx;
    const bf16_t* dy; long lddy;
    float* dw;                                     // [N][taps][Cin_p] fp32 stage, zeroed by the launcher
    int B, H, W, Cin_p, N;
    int groups;                                    // workgroups per channel slice
    long part_stride;                              // > 0: group g stores its partial gradient at dw + g * part_stride (no atomics)
};

// SL = 32-channel input slices per workgroup (2 halves the dy re-reads and doubles the MFMA work per staged tile:
// the 3x3 high-resolution layers are HBM-bound, ~9 k MAC per 128 B).  LDS is single-buffered: the next tile waits in
// registers while the current one is consumed.
// NW = waves per workgroup: 4, or 8 (two waves per SIMD share the staged tile: the kernel holds one 114 KB workgroup per CU, and
// with a single wave per SIMD every LDS read latency and barrier of the k-loop was exposed).
// NH = 2 (round 3, 65..128 output channels): the dy tile holds NH * NT tiles of 32 channels and the waves are split over the two
// halves (wave & 1), each half dealing the (tap, slice) units to its NW / 2 waves -- the 128-channel 3x3 layers at 96x320 used to run
// on the generic per-tap weight gradient, which re-streams x and dy nine times from L2 (3.0x HBM-side traffic, 510-540 TFLOP/s).
template <int K, int NT, int SL, int NW = 4, int NH = 1, int THW = 8>
__global__ __launch_bounds__(NW * 64) void conv_patch_wgrad_kernel(PatchWgradArgs a) {
    constexpr int TH = THW;                                        // tile rows (shadows the file-level 8; the wide variant stages 4-row tiles:
                                                                   // with 8 rows the next tile parked in registers pushed it past 256 VGPRs)
    constexpr int NTHR = NW * 64;
    constexpr int PAD = K / 2, PH = TH + K - 1, PW = TW + K - 1, TAPS = K * K;
    constexpr int UNITS = TAPS * SL;                               // (tap, slice) accumulator units, dealt round-robin to the waves
    constexpr int SLOTS = NW / NH;                                 // waves that share the units of one output half
    constexpr int TPW = (UNITS + SLOTS - 1) / SLOTS;                     // (dealing whole taps left 3x3 layers at 3:2:2:2 -- 25 % of the MFMA slots idle)
    constexpr int XRS = SL == 1 ? 64 : 192;                        // x / dy pixel row strides: odd multiples of 64 B
    constexpr int XCH = PH * PW * 4 * SL, NXC = (XCH + NTHR - 1) / NTHR;
    constexpr int NTT = NT * NH;                                   // 32-channel tiles of the staged dy tile
    constexpr int YRS = NTT == 1 ? 64 : (NTT == 2 ? 192 : 320);
    constexpr int YCH = TH * TW * NTT * 4, NYC = YCH / NTHR;
    static_assert(YCH % NTHR == 0, "the dy tile must split evenly over the threads");
    constexpr int XBYTES = PH * PW * XRS;
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [XBYTES] then [TH*TW*YRS]
    char* X = smem;
    char* Y = smem + XBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.y;                                  // group of SL 32-channel slices
    const int tiles_x = a.W / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * a.B;
    const int per = (ntiles + a.groups - 1) / a.groups;
    const int t_begin = blockIdx.x * per, t_end = min(ntiles, t_begin + per);
    const int cpt = a.Cin_p >> 3, npp = a.N >> 3;

    u32x4_t sx[NXC], sy[NYC];
    auto load_tile = [&](int tile) {
        int id = tile;
        const int tx_ = id % tiles_x; id /= tiles_x;
        const int ty_ = id % tiles_y; const int b = id / tiles_y;
        const int x0 = tx_ * TW, y0 = ty_ * TH;
#pragma unroll
        for (int i = 0; i < NXC; ++i) {
            const int idc = tid + i * NTHR;
            const int p = idc / (4 * SL), kc = idc - p * (4 * SL);
            const int py = p / PW, px = p - py * PW;
            const int iy = y0 + py - PAD, ix = x0 + px - PAD;
            const int cc = slice * 4 * SL + kc;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if ((XCH % NTHR == 0 || idc < XCH) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && cc < cpt)
                v = *(const u32x4_t*)(a.x + (((long)b * a.H + iy) * a.W + ix) * a.ldx + cc * 8);
            sx[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NYC; ++i) {
            const int idc = tid + i * NTHR;
            const int pix = idc / (NTT * 4), c = idc - pix * (NTT * 4);
            const int yy = y0 + pix / TW, xx = x0 + (pix & (TW - 1));
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (yy < a.H && c < npp) v = *(const u32x4_t*)(a.dy + (((long)b * a.H + yy) * a.W + xx) * a.lddy + c * 8);
            sy[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NXC; ++i) {
            const int idc = tid + i * NTHR;
            const int p = idc / (4 * SL), kc = idc - p * (4 * SL);
            if (XCH % NTHR == 0 || idc < XCH) *(u32x4_t*)(X + p * XRS + kc * 16) = sx[i];
        }
#pragma unroll
        for (int i = 0; i < NYC; ++i) {
            const int idc = tid + i * NTHR;
            const int pix = idc / (NTT * 4), c = idc - pix * (NTT * 4);
            *(u32x4_t*)(Y + pix * YRS + c * 16) = sy[i];
        }
    };

    f32x16_t acc[TPW][NT];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][n][e] = 0.f;

    // transposing-read lane roles: 16-lane group g: channels 16*(g&1) + 4p.., pixels 8*(g>>1) + q (+4 for the 2nd read)
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int chb = 16 * (g & 1) + 4 * pp, pxb = 8 * (g >> 1) + q;
    const int nh = NH == 2 ? (wave & 1) : 0, slot = NH == 2 ? (wave >> 1) : wave;      // output half of this wave, its place among the half's waves

    // Round 5 (end): the fragment addresses as ONE base per lane and unit, computed once, + a K-step offset that is a compile-time constant where the K-step loop
    // is unrolled (3x3 kernels: no spills) and one scalar add where it is not.  The loop is issue-bound (profiles/r05_wgrad9_steps.txt): with the unit -> (tap,
    // slice) division and the whole address recomputed per unit and K-step it spent 51 instructions per K-step on 6 MFMAs.
    const char* yb[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) yb[n] = Y + pxb * YRS + ((nh * NT + n) * 32 + chb) * 2;
    const char* xb[TPW];
    bool uok[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int u = slot + SLOTS * i;
        uok[i] = u < UNITS;                                        // (wave-uniform)
        const int uu = uok[i] ? u : 0, tap = uu / SL, sl = uu - tap * SL, dyy = tap / K, dxx = tap - dyy * K;
        xb[i] = X + (dyy * PW + dxx + pxb) * XRS + sl * 64 + chb * 2;
    }
    constexpr bool UNROLL_KS = K == 3 && NH == 1;
    constexpr int FULLU = UNITS / SLOTS;
    const bool extra = uok[TPW - 1];                               // (only read where FULLU < TPW)
    MTE_CLOCK_BEGIN()
    if (t_begin < t_end) {
        load_tile(t_begin);
        for (int tile = t_begin; tile < t_end; ++tile) {
            __syncthreads();                                       // previous tile fully consumed
            store_tile();
            __syncthreads();
            if (tile + 1 < t_end) load_tile(tile + 1);             // in flight (registers) while this tile is multiplied
#pragma unroll UNROLL_KS ? TH * 2 : 1                              // (5x5 / 7x7: unrolled by two they measured 2-4 % SLOWER -- 1000+ TFLOP/s kernels, not issue-bound)
            for (int ks = 0; ks < TH * 2; ++ks) {                  // 16 pixels of one tile row per k-step
                const int row = ks >> 1, col0 = (ks & 1) * 16;
                const int yo = (row * TW + col0) * YRS, xo = (row * PW + col0) * XRS;
                u32x4_t fy[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const char* base = yb[n] + yo;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * YRS));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fy[n] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
                }
                // every wave has FULLU units, the first UNITS % SLOTS waves of a half one more (wave-uniform): all fragment reads of the K-step first, then its MFMAs
                u32x4_t fx[TPW];
                auto read_x = [&](int i) {
                    const char* base = xb[i] + xo;
                    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base));
                    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(base + 4 * XRS));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fx[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
                };
                auto mma_x = [&](int i) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fy[n]),
                                                                            __builtin_bit_cast(bf16x8_t, fx[i]), acc[i][n], 0, 0, 0);
                };
                if constexpr (NH == 1) {
#pragma unroll
                    for (int i = 0; i < FULLU; ++i) read_x(i);
                    if constexpr (FULLU < TPW) { if (extra) read_x(FULLU); }
#pragma unroll
                    for (int i = 0; i < FULLU; ++i) mma_x(i);
                    if constexpr (FULLU < TPW) { if (extra) mma_x(FULLU); }
                } else {                                           // (the wide variant is at its register limit: one unit's fragment at a time)
#pragma unroll
                    for (int i = 0; i < FULLU; ++i) { read_x(i); mma_x(i); }
                    if constexpr (FULLU < TPW) { if (extra) { read_x(FULLU); mma_x(FULLU); } }
                }
            }
        }
    }
    MTE_CLOCK_END(patch)
    // D[row = cout][col = cin]: col = lane&31 -> contiguous fp32 in the stage
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int u = slot + SLOTS * i;
        if (u >= UNITS) continue;
        const int tap = u / SL, sl = u - tap * SL;
        const int cc = (slice * SL + sl) * 32 + r;
        if (cc >= a.Cin_p) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = (nh * NT + n) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < a.N) {
                    float* dst = a.dw + (long)blockIdx.x * a.part_stride + ((long)co * TAPS + tap) * a.Cin_p + cc;
                    if (a.part_stride) *dst = acc[i][n][e]; else atomicAdd(dst, acc[i][n][e]);
                }
            }
    }
}

// Workgroup groups of the weight-gradient launches WHEN THEY SHARE THE CHIP with the data-gradient chain (MTE_OPT_WGRAD_SHARES_CHIP; round 5, same-box
// step times with the two-stream schedule, profiles/r05_side_queue_width.txt):
// 512 -> 256 groups: 23.25 -> 23.16 ms per step (192: the same, 128: 23.60); the wide (65..128-output) variant 256 -> 128: a further -0.08 ms.  Fewer,
// longer workgroups leave CUs to the data-gradient chain and halve the slabs the unpack pass adds up.
// (end of round 5, after the kernel's instruction diet: 256 -> 22.75 ms per step, 192 -> 22.67, 160 -> 22.65, 128 -> 22.95 on one box; 23.24 / 23.13 / -- on another: 192)
#ifndef MTE_PATCH_WGRAD_WGS
#define MTE_PATCH_WGRAD_WGS 192
#endif
#ifndef MTE_PATCH_WGRAD_WIDE_WGS
#define MTE_PATCH_WGRAD_WIDE_WGS 128
#endif
int g_patch_wgrad_wgs = MTE_PATCH_WGRAD_WGS;         // development knob (mte_debug_set(12, v))
int g_patch_tall = 1;                                // development knob (mte_debug_set(11, v))

#ifdef MTE_PATCH_FWD1
int g_patch_fwd2 = 0;                                // (diagnostic builds: tools/patch_stamps.py v1)
#else
int g_patch_fwd2 = 1;
#endif
//                               // development knob (mte_debug_set(11, 400 + v)): 0 = the first form of the forward kernel

// Round 6: every forward form runs on v_mfma_f32_16x16x32_bf16 (M16); the 32x32x16 forms of rounds 1-5 are instantiated in the development library only
// (knobs below; tests/test_gpu_conv_variants.py compares the two).  Same-box A/B per layer: profiles/r06_m16_ab.txt, profiles/r06_inloop_clock.txt.
int g_patch_m16 = 1;                                 // development knob (mte_debug_set(11, 500 + v)): 0 = the 5x5 / 7x7 second form on v_mfma_f32_32x32x16_bf16
int g_patch_m16_3 = 1;                               // development knob (mte_debug_set(11, 700 + v)): 0 = the 3x3 / 1x1 second form on v_mfma_f32_32x32x16_bf16
int g_patch_m16_f1 = 1;                              // development knob (mte_debug_set(11, 600 + v)): 0 = the first form on v_mfma_f32_32x32x16_bf16

// one forward launch: second (F2) or first form of the kernel; accumulating or not (a.accum; never with R1 / EXTRA); MFMA shape
template <bool F2, int K, int NT, bool TALL, bool R1 = false, bool EXTRA = false>
static void launch_form(const PatchArgs& a, long tiles, hipStream_t st, bool m16) {
    const dim3 g((unsigned)tiles), b(256);
    static_assert(F2 || !R1, "the rank-1 term rides the second form");
#ifdef MTE_DEV
    if (!m16) {
        if constexpr (F2) {
            if constexpr (!R1 && !EXTRA) { if (a.accum) { hipLaunchKernelGGL((conv_patch_fwd2_kernel<K, NT, TALL, false, true>), g, b, 0, st, a); return; } }
            hipLaunchKernelGGL((conv_patch_fwd2_kernel<K, NT, TALL, R1, false, EXTRA>), g, b, 0, st, a);
        } else {
            if constexpr (!EXTRA) { if (a.accum) { hipLaunchKernelGGL((conv_patch_fwd_kernel<K, NT, TALL, true>), g, b, 0, st, a); return; } }
            hipLaunchKernelGGL((conv_patch_fwd_kernel<K, NT, TALL, false, EXTRA>), g, b, 0, st, a);
        }
        return;
    }
#endif
    (void)m16;
    if constexpr (F2) {
        if constexpr (!R1 && !EXTRA) { if (a.accum) { hipLaunchKernelGGL((conv_patch_fwd2_kernel<K, NT, TALL, false, true, false, true>), g, b, 0, st, a); return; } }
        hipLaunchKernelGGL((conv_patch_fwd2_kernel<K, NT, TALL, R1, false, EXTRA, true>), g, b, 0, st, a);
    } else {
        if constexpr (!EXTRA) { if (a.accum) { hipLaunchKernelGGL((conv_patch_fwd_kernel<K, NT, TALL, true, false, true>), g, b, 0, st, a); return; } }
        hipLaunchKernelGGL((conv_patch_fwd_kernel<K, NT, TALL, false, EXTRA, true>), g, b, 0, st, a);
    }
}

template <int K, int NT> int launch_fwd(const PatchArgs& a, hipStream_t st, int* tile_rows = nullptr) {
    // the second form addresses the input through a buffer descriptor (< 2 GiB)
    // Same-box A/B over the network's shapes (tools/conv_shape_bench.py): 7x7 -12..-15 %, 5x5 -8..-12 %, 3x3 with 32 outputs -4..-12 %, 3x3
    // with 64 outputs -8 % from three slices on; with one or two slices the first form wins by 8-15 % (167 VGPRs, three workgroups per CU,
    // against 244), and the 1x1 layers are HBM-bound either way
    const bool v2 = g_patch_fwd2 && (K >= 5 || (K == 3 && (NT == 1 || a.Cin_p > 64))) &&
                    (((long)a.B * a.H * a.W - 1) * a.ldx + a.Cin_p) * 2 < 0x7ff00000L && ((uintptr_t)a.bias & 15) == 0;   // (it reads the bias in 16-byte groups)
    const bool m16 = v2 ? (K >= 5 ? g_patch_m16 : g_patch_m16_3) != 0 : g_patch_m16_f1 != 0;
    if constexpr (NT == 1) {
        if (g_patch_tall && (a.Cin_p <= 32 || K <= 3) && a.H >= 16) {
            if (tile_rows) *tile_rows = 16;
            const long tiles = (long)(a.W / TW) * ((a.H + 15) / 16) * a.B;
            if (v2) launch_form<true, K, NT, true>(a, tiles, st, m16);
            else launch_form<false, K, NT, true>(a, tiles, st, m16);
            return mte_check_launch();
        }
    }
    if (tile_rows) *tile_rows = TH;
    const long tiles = (long)(a.W / TW) * ((a.H + TH - 1) / TH) * a.B;
    if (v2) launch_form<true, K, NT, false>(a, tiles, st, m16);
    else launch_form<false, K, NT, false>(a, tiles, st, m16);
    return mte_check_launch();
}
template <int NT> int dispatch_fwd(const PatchArgs& a, int K, hipStream_t st, int* tile_rows = nullptr) {
    switch (K) {
        case 1: return launch_fwd<1, NT>(a, st, tile_rows);
        case 3: return launch_fwd<3, NT>(a, st, tile_rows);
        case 5: return launch_fwd<5, NT>(a, st, tile_rows);
        case 7: return launch_fwd<7, NT>(a, st, tile_rows);
    }
    return MTE_ERR_UNSUPPORTED;
}

int g_patch_wgrad_8w = 1;                            // development knob (mte_debug_set(11, 200 + v)): 0 = four waves per workgroup everywhere
int g_patch_wgrad_wide = 1;                          // development knob (mte_debug_set(11, 300 + v)): 0 = 65..128 output channels stay on the generic weight gradient

template <int K, int NT, int SL, int NW = 4, int NH = 1, int THW = 8> int launch_wgrad_sl(PatchWgradArgs a, hipStream_t st, int parts_cap, int* parts_out) {
    constexpr int TH = THW;
    constexpr int PH = TH + K - 1, PW = TW + K - 1;
    constexpr int XRS = SL == 1 ? 64 : 192, YRS = NT * NH == 1 ? 64 : (NT * NH == 2 ? 192 : 320);
    const size_t lds = PH * PW * XRS + TH * TW * YRS;
    const int nslices = (a.Cin_p + 32 * SL - 1) / (32 * SL);
    const long ntiles = (long)(a.W / TW) * ((a.H + TH - 1) / TH) * a.B;
    // ~2 workgroups per CU in total; the 147 KB wide variant (NH = 2) holds one per CU: one round of workgroups, half the slabs to add up
    // (alone on the chip -- MTE_OPT_WGRAD_SHARES_CHIP off -- twice the groups: the round-4 geometry)
    const int want = (NH == 2 ? MTE_PATCH_WGRAD_WIDE_WGS : g_patch_wgrad_wgs) * (g_mte_wgrad_shared ? 1 : 2);
    long groups = (want + nslices - 1) / nslices;
    if (groups > ntiles) groups = ntiles;
    if (groups > parts_cap) groups = parts_cap < 1 ? 1 : parts_cap;   // one slab per workgroup group, always (round 4: no fp32-atomic combine on this launch path)
    a.groups = (int)groups;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv_patch_wgrad_kernel<K, NT, SL, NW, NH, THW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return MTE_ERR_LAUNCH;
        attr_set = true;
    }
    if (a.groups > 1 && a.groups <= parts_cap) {                // one partial gradient per group, summed by the unpack pass
        a.part_stride = (long)a.N * K * K * a.Cin_p;
        if (parts_out) *parts_out = a.groups;
    } else {
        a.part_stride = 0;
        if (parts_out) *parts_out = 1;
        if (mte_memset_async(a.dw, 0, sizeof(float) * (size_t)a.N * K * K * a.Cin_p, st) != hipSuccess) return MTE_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((conv_patch_wgrad_kernel<K, NT, SL, NW, NH, THW>), dim3((unsigned)groups, nslices), dim3(NW * 64), lds, st, a);
    return mte_check_launch();
}
template <int K, int NT> int launch_wgrad(const PatchWgradArgs& a, hipStream_t st, int parts_cap, int* parts_out) {
    // two slices per workgroup where the accumulators still fit (3x3 and 1x1; 5x5 with C_out <= 32) and there is more than one slice
    // eight waves (two per SIMD on the one workgroup a CU holds) where a wave still gets enough accumulator units: two output
    // tiles per unit, or >= 32 units.  Same-box A/B per launch: 7x7 32->32 @384x1280 0.510 -> 0.388 ms, 3x3 64->64 @192x640
    // 0.146 -> 0.108, 5x5 256->64 @96x320 0.272 -> 0.205; the one-tile launches with 18 / 25 units lose 14-20 % and stay on four.
    if constexpr (K == 3 && NT == 1) {
        // 65..96 input channels (iconv1: the 72-channel decoder concat): all three 32-channel slices in ONE workgroup -- the 192-byte
        // pixel rows of the two-slice layout are exactly full, dy is read once instead of once per slice pair (0.40 -> 0.29 ms)
        if (a.Cin_p > 64 && a.Cin_p <= 96 && g_patch_wgrad_8w) return launch_wgrad_sl<K, NT, 3, 8>(a, st, parts_cap, parts_out);
    }
    if constexpr (K == 3 && NT == 2) {
        // round 6: the same for 64 outputs (iconv2: the 96-channel concat @192x640) -- 27 (tap, slice) units over 8 waves instead of two slice groups of 18, the
        // second one half empty: a third fewer MFMA steps (0.191 -> 0.147 ms, profiles/r06_lowres_split.txt; 4-row tiles: with 8 rows the staged next tile pushed it past 256 VGPRs)
        if (a.Cin_p > 64 && a.Cin_p <= 96 && g_patch_wgrad_8w) return launch_wgrad_sl<K, NT, 3, 8, 1, 4>(a, st, parts_cap, parts_out);
    }
    if constexpr (K <= 3 || (K == 5 && NT == 1)) {
        if (a.Cin_p > 32) {
            if constexpr (K * K * 2 >= 16 && (NT == 2 || K * K * 2 >= 32)) {
                if (g_patch_wgrad_8w) return launch_wgrad_sl<K, NT, 2, 8>(a, st, parts_cap, parts_out);
            }
            return launch_wgrad_sl<K, NT, 2>(a, st, parts_cap, parts_out);
        }
    }
    if constexpr (K * K >= 16 && (NT == 2 || K * K >= 32)) {
        if (g_patch_wgrad_8w) return launch_wgrad_sl<K, NT, 1, 8>(a, st, parts_cap, parts_out);
    }
    return launch_wgrad_sl<K, NT, 1>(a, st, parts_cap, parts_out);
}
template <int NT> int dispatch_wgrad(const PatchWgradArgs& a, int K, hipStream_t st, int parts_cap, int* parts_out) {
    switch (K) {
        case 1: return launch_wgrad<1, NT>(a, st, parts_cap, parts_out);
        case 3: return launch_wgrad<3, NT>(a, st, parts_cap, parts_out);
        case 5: return launch_wgrad<5, NT>(a, st, parts_cap, parts_out);
        case 7: return launch_wgrad<7, NT>(a, st, parts_cap, parts_out);
    }
    return MTE_ERR_UNSUPPORTED;
}

// weight gradient only: 65..128 output channels, 3x3, at least one 64-channel slice pair (the wide variant of conv_patch_wgrad_kernel)
inline bool patch_wgrad_wide_ok(int W, int Cin_p, int N, int KH, int KW) {
    return g_patch_wgrad_wide && W % TW == 0 && Cin_p % 8 == 0 && Cin_p >= 64 && N % 8 == 0 && N > 64 && N <= 128 && KH == 3 && KW == 3;
}
inline bool patch_shape_ok(int W, int Cin_p, int N, int KH, int KW) {
    if (KH == 7 && N > 32) return false;             // 13 taps x 2 tiles of accumulators per wave would spill in wgrad
    return W % TW == 0 && Cin_p % 8 == 0 && N % 8 == 0 && N <= 64 && KH == KW && (KH == 1 || KH == 3 || KH == 5 || KH == 7);
}

}  // namespace

#ifdef MTE_DEV
extern "C" int mtei_set_patch_tall(int v) { if (v >= 700 && v < 710) { g_patch_m16_3 = v - 700; return MTE_OK; } if (v >= 600 && v < 610) { g_patch_m16_f1 = v - 600; return MTE_OK; } if (v >= 500 && v < 510) { g_patch_m16 = v - 500; return MTE_OK; } if (v >= 400 && v < 410) { g_patch_fwd2 = v - 400; return MTE_OK; } if (v >= 300 && v < 310) { g_patch_wgrad_wide = v - 300; return MTE_OK; } if (v >= 200 && v < 210) { g_patch_wgrad_8w = v - 200; return MTE_OK; } if (v >= 100) { g_patch_wgrad_wgs = v; return MTE_OK; } g_patch_tall = v; return MTE_OK; }
#endif

extern "C" {

// 1 if the LDS-patch kernels cover this conv shape (bf16, C_out <= 64, W % 32 == 0, k in {1,3,5,7}), else 0.
int mte_conv2d_patch_supported(int W, int Cin_p, int N, int KH, int KW, int dtype) {
    return (dtype == MTE_DT_BF16 && patch_shape_ok(W, Cin_p, N, KH, KW)) ? 1 : 0;
}

// 1 if mte_conv2d_patch_wgrad covers this shape: everything mte_conv2d_patch_supported covers, plus 3x3 layers with 65..128 output channels
int mte_conv2d_patch_wgrad_supported(int W, int Cin_p, int N, int KH, int KW, int dtype) {
    return (dtype == MTE_DT_BF16 && (patch_shape_ok(W, Cin_p, N, KH, KW) || patch_wgrad_wide_ok(W, Cin_p, N, KH, KW))) ? 1 : 0;
}

// elements (bf16) of the fragment-block weight pack for mte_conv2d_patch_fwd
long mte_conv2d_patch_pack_elems(int Cin_p, int N, int KH, int KW) {
    return (long)((Cin_p + 31) / 32) * KH * KW * 2 * ((N + 31) / 32) * 64 * 8;
}

// generic pack [N][taps][Cin_p] (bf16, from mte_pack_conv_weights) -> fragment blocks
int mte_conv2d_patch_repack(const void* wgeneric, void* wpatch, int Cin_p, int N, int KH, int KW, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!wgeneric || !wpatch) return MTE_ERR_ARG;
    const long total = mte_conv2d_patch_pack_elems(Cin_p, N, KH, KW);
    long g = (total + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(repack_patch_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const bf16_t*)wgeneric, (bf16_t*)wpatch, N, KH * KW, Cin_p, (N + 31) / 32);
    return mte_check_launch();
}

// y = conv(x, wpatch) + bias for C_out <= 64 (forward, or data-gradient with the backward pack); bf16 only.
int mte_conv2d_patch_fwd(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy,
                         int B, int H, int W, int Cin_p, int N, int KH, int KW, int accumulate, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpatch || !y || !patch_shape_ok(W, Cin_p, N, KH, KW)) return MTE_ERR_ARG;
    PatchArgs a{(const bf16_t*)x, ldx, (const bf16_t*)wpatch, bias, (bf16_t*)y, ldy, B, H, W, Cin_p, N, accumulate ? 1 : 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0};
    return N <= 32 ? dispatch_fwd<1>(a, KH, stream) : dispatch_fwd<2>(a, KH, stream);
}

// mte_conv2d_patch_fwd that also leaves the GroupNorm(16) statistics of y as per-tile records (round 5): rec needs mte_conv2d_patch_fwd_gn_elems(B, H, W) floats;
// *tiles_per_sample_out = records written per sample (the tile height depends on the kernel form), to be handed to mte_gn_stats_from_records.
// With accumulate the records describe the SUMS this launch stores.  N % 16 == 0 (whole groups).
long mte_conv2d_patch_fwd_gn_elems(int B, int H, int W) { return (long)B * (W / TW) * ((H + TH - 1) / TH) * 32; }
int mte_conv2d_patch_fwd_gn(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy, int B, int H, int W, int Cin_p, int N, int KH, int KW,
                            int accumulate, float* rec, long rec_elems, int* tiles_per_sample_out, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpatch || !y || !rec || !tiles_per_sample_out || !patch_shape_ok(W, Cin_p, N, KH, KW)) return MTE_ERR_ARG;
    if (N % 16 != 0) return MTE_ERR_UNSUPPORTED;
    if (rec_elems < mte_conv2d_patch_fwd_gn_elems(B, H, W)) return MTE_ERR_ARG;
    PatchArgs a{(const bf16_t*)x, ldx, (const bf16_t*)wpatch, bias, (bf16_t*)y, ldy, B, H, W, Cin_p, N, accumulate ? 1 : 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, rec};
    int rows = TH;
    const int rc = N <= 32 ? dispatch_fwd<1>(a, KH, stream, &rows) : dispatch_fwd<2>(a, KH, stream, &rows);
    *tiles_per_sample_out = (W / TW) * ((H + rows - 1) / rows);
    return rc;
}

// The 3x3 forward with ONE MORE input channel given as a low-resolution map: y = conv_3(x, wpatch) + bias + conv_1(nearest_up2(inv), w1).  inv [B][H/2][W/2] fp32;
// element (n, tap) of the extra channel's weights at w1[n * w1_stride + tap] (channel C-1 of an OIHW tensor: w1 = w + (C-1)*9, w1_stride = C*9).
// The term is formed in the store loop of the tile from LDS tables (the map under the tile + halo, the 9 x N weights): no pass over y before, no read of y.
// _ok: 1 when this launch form exists for the shape (the caller otherwise writes the term with mte_rank1_conv_fwd and accumulates onto it).
int mte_conv2d_patch_fwd_rank1_ok(const float* bias, long ldx, int B, int H, int W, int Cin_p, int N) {
    if (!patch_shape_ok(W, Cin_p, N, 3, 3) || !g_patch_fwd2 || (H & 1) || (W & 1)) return 0;
    if (N > 32 && Cin_p <= 64) return 0;                            // (those shapes run the first form of the kernel)
    if ((((long)B * H * W - 1) * ldx + Cin_p) * 2 >= 0x7ff00000L || ((uintptr_t)bias & 15) != 0) return 0;
    return 1;
}
int mte_conv2d_patch_fwd_rank1(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy, int B, int H, int W, int Cin_p, int N,
                               const float* inv, const float* w1, long w1_stride, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpatch || !y || !inv || !w1) return MTE_ERR_ARG;
    if (!mte_conv2d_patch_fwd_rank1_ok(bias, ldx, B, H, W, Cin_p, N)) return MTE_ERR_UNSUPPORTED;
    PatchArgs a{(const bf16_t*)x, ldx, (const bf16_t*)wpatch, bias, (bf16_t*)y, ldy, B, H, W, Cin_p, N, 0, inv, w1, w1_stride, nullptr, 0, nullptr, 0};
    if (N <= 32) {
        if (g_patch_tall && H >= 16) launch_form<true, 3, 1, true, true>(a, (long)(W / TW) * ((H + 15) / 16) * B, stream, g_patch_m16_3 != 0);
        else launch_form<true, 3, 1, false, true>(a, (long)(W / TW) * ((H + TH - 1) / TH) * B, stream, g_patch_m16_3 != 0);
    } else {
        launch_form<true, 3, 2, false, true>(a, (long)(W / TW) * ((H + TH - 1) / TH) * B, stream, g_patch_m16_3 != 0);
    }
    return mte_check_launch();
}

// y = conv_3x3(x, wpatch) + conv_1x1(x2, wpatch2) + bias in ONE launch: the second term's C2 channels are further K-steps of every tile at the centre tap
// (wpatch2: the fragment-block pack of the 1x1 weights for the same N).  Written for the data gradient of a residual block's input (reference layers01.py:55-73:
// conv1 (3x3) and the 1x1 shortcut conv3 read the same x), dx = conv3x3^T(dy1) + conv1x1^T(dy3): instead of a 1x1 launch and an ACCUMULATING 3x3 launch.
static int launch_plus1x1(const PatchArgs& a, hipStream_t stream, int* rows) {
    const int B = a.B, H = a.H, W = a.W, N = a.N;
    const bool v2 = g_patch_fwd2 && (N <= 32 || a.Cin_p > 64) && (((long)B * H * W - 1) * a.ldx + a.Cin_p) * 2 < 0x7ff00000L && ((uintptr_t)a.bias & 15) == 0;   // (as launch_fwd)
    *rows = TH;
    const bool m16 = v2 ? g_patch_m16_3 != 0 : g_patch_m16_f1 != 0;
    if (N <= 32) {
        if (g_patch_tall && H >= 16) {
            *rows = 16;
            const long tiles = (long)(W / TW) * ((H + 15) / 16) * B;
            if (v2) launch_form<true, 3, 1, true, false, true>(a, tiles, stream, m16);
            else launch_form<false, 3, 1, true, false, true>(a, tiles, stream, m16);
        } else {
            const long tiles = (long)(W / TW) * ((H + TH - 1) / TH) * B;
            if (v2) launch_form<true, 3, 1, false, false, true>(a, tiles, stream, m16);
            else launch_form<false, 3, 1, false, false, true>(a, tiles, stream, m16);
        }
    } else {
        const long tiles = (long)(W / TW) * ((H + TH - 1) / TH) * B;
        if (v2) launch_form<true, 3, 2, false, false, true>(a, tiles, stream, m16);
        else launch_form<false, 3, 2, false, false, true>(a, tiles, stream, m16);
    }
    return mte_check_launch();
}
int mte_conv2d_patch_fwd_plus1x1(const void* x, long ldx, const void* wpatch, const float* bias, void* y, long ldy, int B, int H, int W, int Cin_p, int N,
                                 const void* x2, long ldx2, const void* wpatch2, int C2, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !wpatch || !y || !x2 || !wpatch2 || C2 < 8 || C2 % 8 != 0 || !patch_shape_ok(W, Cin_p, N, 3, 3)) return MTE_ERR_ARG;
    PatchArgs a{(const bf16_t*)x, ldx, (const bf16_t*)wpatch, bias, (bf16_t*)y, ldy, B, H, W, Cin_p, N, 0, nullptr, nullptr, 0, (const bf16_t*)x2, ldx2, (const bf16_t*)wpatch2, C2};
    int rows;
    return launch_plus1x1(a, stream, &rows);
}
// dw_stage[N][KH*KW][Cin_p] fp32 (overwritten) for C_out <= 64; bf16 only.
int mte_conv2d_patch_wgrad(const void* x, long ldx, const void* dy, long lddy, float* dw_stage, int stage_parts, int* parts_out,
                           int B, int H, int W, int Cin_p, int N, int KH, int KW, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!x || !dy || !dw_stage) return MTE_ERR_ARG;
    PatchWgradArgs a{(const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, dw_stage, B, H, W, Cin_p, N, 1};
    if (parts_out) *parts_out = 1;
    if (!patch_shape_ok(W, Cin_p, N, KH, KW)) {
        if (!patch_wgrad_wide_ok(W, Cin_p, N, KH, KW)) return MTE_ERR_ARG;
        return launch_wgrad_sl<3, 2, 2, 8, 2, 4>(a, stream, stage_parts, parts_out);
    }
    return N <= 32 ? dispatch_wgrad<1>(a, KH, stream, stage_parts, parts_out) : dispatch_wgrad<2>(a, KH, stream, stage_parts, parts_out);
}

}  // extern "C"
