// Implicit-GEMM 2-D convolution, 8-PHASE main loop (round 4) -- forward and data gradient of the layers with >= 128 output columns.
//
//   y[m][n] = bias[n] + sum_{tap,c} x[pix(m)+tap][c] * w[n][tap][c]      M = B*H*W pixels, N = C_out, K = KH*KW*Cin_p
//
// Replaces, for those layers, the one-barrier / ping-pong loops of conv_igemm.hip behind the same entry point (mte_conv2d_igemm:
// nn.Conv2d + ConstantPad2d of packnet_sfm/networks/layers/packnet/layers01.py:29-31,61 and their autograd data gradient).  Same K
// order and the same fp32 accumulation chain per output as every other tile form (one 16x16x32 MFMA per 32-element K-step, K-steps in
// order), so the results are bit-identical to them (tests/test_gpu_conv_variants.py, tools/igemm_race_stress.py).
//
// Structure (cdna_hip_programming.md, "The 256^2 8-phase template"; in-loop stamps of round 3 had the old loops at 45-60 % MFMA
// occupancy: every wave of a workgroup issued its LDS-DMA and its fragment reads together, pipe idle, then all queued on the pipe):
//   * tile 256 pixels x BN columns (BN = 256: 8 waves as 2 x 4, 128 x 64 outputs per wave; BN = 128: 4 x 2, 64 x 64 per wave),
//     K-tile = 64 elements = two K-steps ("K-halves") of 32, each with its own filter tap (Cin_p % 32 == 0 is all that is asked).
//   * the tile is cut into HALF-TILES: A0 / A1 = pixel rows 0..127 / 128..255, B0 / B1 = the two column halves; a wave owns a
//     quarter of each quadrant (A_h x B_h'), so one half-tile feeds one quadrant row / column of EVERY wave.
//   * a K-tile is four PHASES, one quadrant (x K = 64) each, in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0):
//         fragment reads of the phase (B0 + A0 | B1 | A1 | none)  ||  LDS-DMA of ONE half-tile, 7 half-tiles ahead in the stream
//         s_barrier ; lgkmcnt(0) ; MFMAs of the quadrant ; s_barrier
//     and the two wave groups (pixel halves of the wave grid; waves g and g + 4 share a SIMD) run ONE barrier apart, so the loads of one
//     group sit under the MFMAs of the other in every phase.
//   * LDS: 2 buffers x (A0, A1, B0, B1) x 2 K-halves x [rows][64 B]; the half-tile stream is B0 A0 B1 A1 per K-tile, phase p of
//     K-tile T stages stream element 4T + p + 6: A1(T+1), B0(T+2), A0(T+2), B1(T+2).  vmcnt is waited for ONCE per K-tile (phase 4,
//     counted: the three youngest half-tiles stay in flight across it), never 0 inside the loop.
//   * hazards, by barrier count (group 1 one barrier behind group 0):
//       RAW  every wave waits for its part of K-tile T+1 BEFORE its first barrier of phase 4; nobody reads that K-tile before its own
//            second barrier of phase 4, which is not earlier than any wave's first.
//       WAR  A1(T+1) over A1(T-1): read in phase 3 of T-1, staged in phase 1 of T;  A0(T+2) over A0(T): read in phase 1 (retired by
//            lgkmcnt(0) before the reader's second barrier), staged in phase 3;  B1(T+2) over B1(T): read in phase 2, staged in phase 4;
//            B0(T+2) over B0(T): staged ONE phase after the read -- the B0 reads are issued first and retired by lgkmcnt(8) BEFORE the
//            reader's first barrier of phase 1.
//   * the MFMA operands are swapped (D = W * X^T): a lane ends up with 4 consecutive channels of one pixel, the epilogue stages the
//     tile with 8-byte LDS writes (2-byte ones before) and stores 16-byte row-contiguous chunks.
// Out-of-image taps, rows past M, columns past N and K-steps past the end of the (split's) reduction are out-of-range buffer offsets:
// the LDS-DMA writes zeros for them.
//
// What the in-loop stamps say (tools/igemm8_stamps.py, profiles/r04_igemm8_stamps_v1.txt; 256 -> 256 3x3 @48x160): a K-tile takes ~3500 cycles for
// 2 x 4 x 16 MFMAs = 2050 cycles of pipe time per SIMD (58 %, 1385 TFLOP/s in the main loop = the template's own rate on random data); the MFMA
// sections run at the pipe rate (312 per 16), load sections with 8-12 fragment reads + 2 DMA take 270, and phase 2 -- whose load section also
// carries the K-half tap bookkeeping (`advance`) -- 420-900.  A wave issues ONE instruction per ~4-5 cycles whatever its kind (the CU visits a
// SIMD every fourth cycle), so a load section has room for ~60 instructions beside the other group's MFMAs; the round's attempts to shrink that
// bookkeeping are recorded in profiles/r04_igemm8_loader_ablation.txt and tools/probe/ (tap shift in the scalar buffer offset, validity bit
// masks, branch-free stepping -- each form correct, each SLOWER here: 36 branch-free scalar instructions per K-half cost more than the branchy
// form's 5 on the common path, and vector work scheduled between the MFMAs starves both the matrix pipe and the other group's loads).
#include "common.hpp"
#include "conv_args.hpp"

#ifdef MTE_STAMPS
__device__ unsigned long long g_igemm8_stamps[4096 * 24];
extern "C" int mtei_igemm8_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_igemm8_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

MTE_CLOCK_DEFINE(igemm8)

namespace {

constexpr unsigned OOB8 = 0xfffffff0u;

__device__ __forceinline__ f32x4_t mma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// SM: slice-major K order (ConvArgs.kslice) -- a template parameter, not a run-time flag: as a flag its bookkeeping sat in the tap-major loop as well (373 instead
// of 267 instructions per K-tile of the 256 x 256 form, 325 instead of 213 of the 256 x 128 form, ~75 of them scalar; found at the end of round 5 by counting the
// loop's instructions -- the loop is issue-bound, profiles/r05_wgrad9_steps.txt).  Only the development library instantiates SM = true.
// ONE (tap-major only): Cin_p % 64 == 0 and the split starts on an even K-step, so the two K-halves of every K-tile sit in the same tap, 32 channels apart: one tap
// state instead of two (the general form keeps them apart for Cin_p % 64 == 32 and for splits that start on an odd K-step).
template <int BN, bool SM = false, bool ONE = false>
__global__ __launch_bounds__(512, 2) void conv_igemm8_kernel(ConvArgs a) {
    static_assert(!(SM && ONE), "the slice-major order has its own single tap state");
    typedef bf16_t T;
    constexpr int BM = 256, HM = 128, HN = BN / 2;                  // tile, half-tile rows of A / B
    constexpr int WN = BN == 256 ? 4 : 2;                            // wave grid WM x WN (WM = 8 / WN)
    constexpr int QA = BN == 256 ? 4 : 2, QB = 2;                    // 16-row blocks of a wave's quadrant: QA pixel blocks x QB column blocks
    constexpr int IA = 2, IB = BN == 256 ? 2 : 1;                    // LDS-DMA instructions per wave and half-tile
    constexpr int A_BYTES = 4 * HM * 64, B_BYTES = 4 * HN * 64;      // [half][K-half][rows][64 B]
    constexpr int BUF = A_BYTES + B_BYTES;                           // one K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / WN, wn = wv % WN;
    const int grp = wv >> 2;                                         // waves g and g + 4 share a SIMD: the two groups alternate on it
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tiles_m = (int)((a.M + BM - 1) / BM);
    const int ntiles = tiles_m * tiles_n;
    const int split = blockIdx.x / ntiles;
    const int id = xcd_remap(blockIdx.x - split * ntiles, ntiles);
    const int tile_n = id % tiles_n, tile_m = id / tiles_n;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;

    const int pad_h = a.KH >> 1, pad_w = a.KW >> 1;
    const int taps = a.KH * a.KW;
    const int cpt = a.Cin_p >> 3;                                    // 16-byte chunks per tap (a multiple of 4)
    const int ksteps_all = (taps * cpt) >> 2;
    constexpr bool sm = SM;                                          // slice-major K order (ConvArgs.kslice; cpt % 8 == 0): K-tile T = (slice T / taps, tap T % taps)
    // tap-major: the split's range in K-steps of 32 channels; slice-major: in K-tiles of 64 (the same variables, one unit up)
    const int units_all = sm ? ksteps_all >> 1 : ksteps_all;
    const int per_split = (units_all + a.splits - 1) / a.splits;
    const int s_begin = split * per_split;
    const int s_end = min(units_all, s_begin + per_split);
    const int nkt = s_end > s_begin ? (sm ? s_end - s_begin : (s_end - s_begin + 1) >> 1) : 0;   // K-tiles of this split (tap-major: the last may hold one K-step)
    const long Kp = (long)taps * a.Cin_p;
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ wp = (const T*)a.w;

    // ---- loader constants.  One LDS-DMA instruction of a wave fills 16 rows x 64 B: lane -> row (lane >> 2), slot (lane & 3); the slot
    // holds source chunk slot ^ swz(row), swz(row) = (row >> 1) & 3 (conflict-free 16-row fragment reads, as conv_igemm.hip).
    const int lrow = wv * 16 + (lane >> 2);                          // row inside a 128-row half-tile image
    const int kc = (lane & 3) ^ ((lane >> 3) & 3);
    int a_oy[2], a_ox[2]; unsigned voffA[2]; bool a_ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long m = m0 + h * HM + lrow;
        a_ok[h] = m < a.M;
        const long mm = a_ok[h] ? m : 0;
        const int hw = a.H * a.W;
        const int b = (int)(mm / hw), rem = (int)(mm - (long)b * hw);
        a_oy[h] = rem / a.W; a_ox[h] = rem - a_oy[h] * a.W;
        voffA[h] = (unsigned)((mm * a.ldx + kc * 8) * 2);
    }
    // B rows: BN = 256: as A (two instructions, K-half = instruction index); BN = 128: one instruction, waves 0-3 fill K-half 0, 4-7 K-half 1
    const int brow = BN == 256 ? lrow : (wv & 3) * 16 + (lane >> 2);
    unsigned voffB[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * HN + brow;
        voffB[h] = n < a.N ? (unsigned)(((long)n * Kp + kc * 8) * 2) : OOB8;
    }
    // tap state of the even / odd K-steps (K-half 0 / 1 of the K-tile the A stream is at); all scalars wave-uniform.
    //   tap-major  : the two K-halves are consecutive K-steps of the pack order and may sit in different taps (Cin_p % 64 == 32)
    //   slice-major: both halves share the tap; the slice gives their chunk offsets; st_s[kh] = the half's K-step index IN THE PACK (weights), a_t the K-tile
    int st_s[2], st_ty[2], st_tx[2], st_cb[2];
    bool st_live[2];
    int a_t = s_begin, a_slice = 0;
    unsigned voffT[2][2];                                            // [K-half][pixel half]
    auto set_tap = [&](int kh) {
#ifdef MTE_I8_TAP0      // private diagnostic build (tools/igemm8_locality.py): every tap reads the centre pixel -- what perfect L2 locality of the taps would be worth
        const int dy = 0, dx = 0;
#else
        const int dy = st_ty[kh] - pad_h, dx = st_tx[kh] - pad_w;
#endif
        const int delta = (dy * a.W + dx) * (int)a.ldx * 2;         // byte shift of this tap (may be negative)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int iy = a_oy[h] + dy, ix = a_ox[h] + dx;
            const bool ok = a_ok[h] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            voffT[kh][h] = ok ? voffA[h] + (unsigned)delta : OOB8;
        }
    };
    auto sm_place = [&]() {                                          // slice-major: both halves from (a_slice, tap st_ty[0], st_tx[0])
        const int tap = st_ty[0] * a.KW + st_tx[0];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            st_cb[kh] = a_slice * 8 + 4 * kh;
            st_s[kh] = tap * (cpt >> 2) + a_slice * 2 + kh;
            st_live[kh] = a_t < s_end;
        }
        st_ty[1] = st_ty[0]; st_tx[1] = st_tx[0];
        set_tap(0);
        voffT[1][0] = voffT[0][0]; voffT[1][1] = voffT[0][1];
    };
    if (sm) {
        a_slice = s_begin / taps;
        const int tap0 = s_begin - a_slice * taps;
        st_ty[0] = tap0 / a.KW; st_tx[0] = tap0 - st_ty[0] * a.KW;
        sm_place();
    } else {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            st_s[kh] = s_begin + kh;
            st_live[kh] = st_s[kh] < s_end;
            const int q0 = 4 * st_s[kh], tap0 = q0 / cpt;
            st_cb[kh] = q0 - tap0 * cpt; st_ty[kh] = tap0 / a.KW; st_tx[kh] = tap0 - st_ty[kh] * a.KW;
            set_tap(kh);
        }
    }
    auto advance = [&]() {                                           // the A stream to the next K-tile
        if (sm) {                                                    // next tap of the slice; after the last one the next slice's first
            ++a_t;
            if (++st_tx[0] == a.KW) { st_tx[0] = 0; if (++st_ty[0] == a.KH) { st_ty[0] = 0; ++a_slice; } }
            sm_place();
            return;
        }
        if constexpr (ONE) {                                         // both halves: tap of half 0, chunks st_cb[0] and st_cb[0] + 4 (see stageA)
            st_s[0] += 2; st_cb[0] += 8;
            st_live[0] = st_s[0] < s_end;                            // (an even count of K-steps: the halves live and die together)
            if (st_cb[0] >= cpt) {                                   // wave-uniform; cpt % 8 == 0: exactly cpt
                st_cb[0] = 0;
                if (++st_tx[0] == a.KW) { st_tx[0] = 0; ++st_ty[0]; }
                set_tap(0);
            }
            return;
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            st_s[kh] += 2; st_cb[kh] += 8;
            st_live[kh] = st_s[kh] < s_end;
            if (st_cb[kh] >= cpt) {                                  // wave-uniform
                do { st_cb[kh] -= cpt; if (++st_tx[kh] == a.KW) { st_tx[kh] = 0; ++st_ty[kh]; } } while (st_cb[kh] >= cpt);
                set_tap(kh);
            }
        }
    };
    // the B (weight) stream: K-step index in the pack of K-half 0 of the K-tile it stands at (K-half 1 is the next one in both orders)
    int b_t = s_begin, b_tap = 0, b_slice = 0, b_sp0 = s_begin;
    if (sm) { b_slice = s_begin / taps; b_tap = s_begin - b_slice * taps; b_sp0 = b_tap * (cpt >> 2) + b_slice * 2; }
    auto advanceB = [&]() {
        if (sm) {
            ++b_t;
            if (++b_tap == taps) { b_tap = 0; ++b_slice; }
            b_sp0 = b_tap * (cpt >> 2) + b_slice * 2;
        } else { b_t += 2; b_sp0 += 2; }
    };
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lptr_t;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, (int)(((a.M - 1) * a.ldx + a.Cin_p) * 2), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, (int)((long)a.N * Kp * 2), 0x00020000);
#endif
    // half-tile A_h of the K-tile the tap state stands at, into buffer `buf`
    auto stageA = [&](int h, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            if constexpr (ONE) {
                const unsigned vo = st_live[0] ? voffT[0][h] : OOB8;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + buf * BUF + (h * 2 + kh) * (HM * 64) + wv * 1024), 16, vo, (st_cb[0] + 4 * kh) * 16, 0, 0);
            } else {
                const unsigned vo = st_live[kh] ? voffT[kh][h] : OOB8;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(smem + buf * BUF + (h * 2 + kh) * (HM * 64) + wv * 1024), 16, vo, st_cb[kh] * 16, 0, 0);
            }
        }
#endif
    };
    // half-tile B_h of the K-tile the B stream stands at
    auto stageB = [&](int h, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (BN == 256) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bool live = (sm || ONE) ? b_t < s_end : b_t + kh < s_end;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(smem + buf * BUF + A_BYTES + (h * 2 + kh) * (HN * 64) + wv * 1024), 16,
                                                         live ? voffB[h] : OOB8, (b_sp0 + kh) * 64, 0, 0);
            }
        } else {
            const int kh = wv >> 2;
            const bool live = (sm || ONE) ? b_t < s_end : b_t + kh < s_end;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(smem + buf * BUF + A_BYTES + (h * 2 + kh) * (HN * 64) + (wv & 3) * 1024), 16,
                                                     live ? voffB[h] : OOB8, (b_sp0 + kh) * 64, 0, 0);
        }
#endif
    };

    // ---- fragments: lane (r16 = lane & 15, q16 = lane >> 4) reads chunk q16 of row r16 of a 16-row block
    const int r16 = lane & 15, q16 = lane >> 4;
    const int swz = (q16 ^ ((r16 >> 1) & 3)) << 4;
    const int offA = (wm * QA * 16 + r16) * 64 + swz;                                  // + (h * 2 + kh) * HM * 64 + i * 1024
    const int offB = A_BYTES + (wn * QB * 16 + r16) * 64 + swz;                        // + (h * 2 + kh) * HN * 64 + j * 1024
    u32x4_t fa[QA][2], fb0[QB][2], fb1[QB][2];                                         // [block][K-half]; fa is A0, then A1
    f32x4_t acc[2][QA][2][QB];                                                         // [pixel half][block][column half][block]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < QA; ++i)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < QB; ++j) acc[h][i][g][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto readA = [&](int h, const char* pb) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < QA; ++i) fa[i][kh] = *(const u32x4_t*)(pb + offA + (h * 2 + kh) * (HM * 64) + i * 1024);
    };
    auto readB = [&](int g, u32x4_t (&fb)[QB][2], const char* pb) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int j = 0; j < QB; ++j) fb[j][kh] = *(const u32x4_t*)(pb + offB + (g * 2 + kh) * (HN * 64) + j * 1024);
    };
    auto quadrant = [&](int h, int g, const u32x4_t (&fb)[QB][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int i = 0; i < QA; ++i)
#pragma unroll
                for (int j = 0; j < QB; ++j) acc[h][i][g][j] = mma16(fb[j][kh], fa[i][kh], acc[h][i][g][j]);
        __builtin_amdgcn_s_setprio(0);
    };
#define MTE8_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define MTE8_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")
#define MTE8_BARRIER() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    // ---- prologue: stream elements 0..6 = B0 A0 B1 A1 of K-tile 0, B0 A0 B1 of K-tile 1
    stageB(0, 0); stageA(0, 0); stageB(1, 0); stageA(1, 0);
    advance(); advanceB();
    stageB(0, 1); stageA(0, 1); stageB(1, 1);
    advanceB();
    MTE8_WAIT_VM(2 * IB + IA);                                       // K-tile 0 has landed
    MTE8_BARRIER();
    if (grp == 1) MTE8_BARRIER();                                    // group 1 runs one barrier behind from here on
#ifdef MTE_STAMPS
    unsigned long long t_st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_a, t_b;      // [phase][load section | first barrier | MFMAs | second barrier]
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#define ST8(ACC) { __builtin_amdgcn_sched_barrier(0); t_b = __builtin_amdgcn_s_memtime(); t_st[ACC] += t_b - t_a; t_a = t_b; __builtin_amdgcn_sched_barrier(0); }
#else
#define ST8(ACC)
#endif
    MTE_CLOCK_BEGIN()
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const char* pb = smem + buf * BUF;
#ifdef MTE_STAMPS
        t_a = __builtin_amdgcn_s_memtime();
#endif
        // ---- phase 1: quadrant (A0, B0); stage A1 of K-tile kt + 1 (the tap state stands there)
        readB(0, fb0, pb);
        __builtin_amdgcn_sched_barrier(0);                           // B0 first: lgkmcnt(QA * 2) below retires exactly these
        readA(0, pb);
        stageA(1, buf ^ 1);
        MTE8_WAIT_LGKM(QA * 2);
        ST8(0)
        MTE8_BARRIER();
        ST8(1)
        MTE8_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        quadrant(0, 0, fb0);
        ST8(2)
        MTE8_BARRIER();
        ST8(3)
        // ---- phase 2: quadrant (A0, B1); stage B0 of K-tile kt + 2, move the tap state there
        readB(1, fb1, pb);
        stageB(0, buf);
        advance();
        ST8(4)
        MTE8_BARRIER();
        ST8(5)
        MTE8_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        quadrant(0, 1, fb1);
        ST8(6)
        MTE8_BARRIER();
        ST8(7)
        // ---- phase 3: quadrant (A1, B1); stage A0 of K-tile kt + 2
        readA(1, pb);
        stageA(0, buf);
        ST8(8)
        MTE8_BARRIER();
        ST8(9)
        MTE8_WAIT_LGKM(0);
        __builtin_amdgcn_sched_barrier(0);
        quadrant(1, 1, fb1);
        ST8(10)
        MTE8_BARRIER();
        ST8(11)
        // ---- phase 4: quadrant (A1, B0); stage B1 of K-tile kt + 2; K-tile kt + 1 must have landed
        stageB(1, buf);
        advanceB();
        MTE8_WAIT_VM(2 * IB + IA);
        ST8(12)
        MTE8_BARRIER();
        ST8(13)
        quadrant(1, 0, fb0);
        ST8(14)
        MTE8_BARRIER();
        ST8(15)
    }
    MTE_CLOCK_END(igemm8)
    if (grp == 0) MTE8_BARRIER();
#ifdef MTE_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* o = g_igemm8_stamps + ((long)blockIdx.x * 8 + wv) * 24;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = t_st[k];
        o[16] = __builtin_amdgcn_s_memtime() - t_start; o[17] = nkt;
    }
#endif
    MTE8_WAIT_VM(0);                                                 // the zero-filling DMA of the K-tiles past the end must not land in the output tile
    MTE8_BARRIER();

    // ---- epilogue.  acc[h][i][g][j][e]: pixel row h * 128 + (wm * QA + i) * 16 + r16, column g * HN + (wn * QB + j) * 16 + 4 * q16 + e
    if (a.splits > 1) {                                              // partial sums of this split: fp32 slab, 16-byte stores
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < QA; ++i) {
                const long m = m0 + h * HM + (wm * QA + i) * 16 + r16;
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int j = 0; j < QB; ++j) {
                        const int n = n0 + g * HN + (wn * QB + j) * 16 + 4 * q16;
                        if (m < a.M && n < a.N) *(f32x4_t*)(a.ws + ((long)split * a.M + m) * a.N + n) = acc[h][i][g][j];
                    }
            }
        return;
    }
    // [256 pixels][BN columns] tile in bf16 (row = BN * 2 bytes), 16-byte chunk c of row r at slot c ^ (r & 15)
    constexpr int ROWB = BN * 2, CPR = BN / 8;
    static_assert(BM * ROWB <= 2 * BUF, "output tile must fit the ring");
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const int col = g * HN + (wn * QB + j) * 16 + 4 * q16;
            f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && n0 + col < a.N) {                      // (four scalar loads: a bias inside a flat parameter buffer is only 4-byte aligned)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = a.bias[n0 + col + e];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < QA; ++i) {
                    const int row = h * HM + (wm * QA + i) * 16 + r16;
                    const f32x4_t v = acc[h][i][g][j] + bv;
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *(uint2*)(smem + row * ROWB + ((((col >> 3) ^ r16)) << 4) + (col & 7) * 2) = pk;
                }
        }
    __syncthreads();
    const int cvalid = (a.N - n0) >> 3;                              // chunks of this tile inside N (N % 8 == 0)
    const int cc = tid % CPR;
#pragma unroll
    for (int it = 0; it < BM * CPR / 512; ++it) {
        const int row = tid / CPR + it * (512 / CPR);
        const long m = m0 + row;
        if (m < a.M && cc < cvalid) {
            u32x4_t c = *(const u32x4_t*)(smem + row * ROWB + ((cc ^ (row & 15)) << 4));
            T* dst = (T*)a.y + m * a.ldy + n0 + cc * 8;
            if (a.accum) {
                float vn[8], vo[8];
                unpack16<T>(c, vn);
                unpack16<T>(*(const u32x4_t*)dst, vo);
#pragma unroll
                for (int k = 0; k < 8; ++k) vn[k] += vo[k];
                c = pack16<T>(vn);
            }
            *(u32x4_t*)dst = c;
        }
    }
}


template <int BN, bool SM, bool ONE = false> int launch8(const ConvArgs& a, hipStream_t st) {
    constexpr int LDS = 2 * (4 * 128 * 64 + 4 * (BN / 2) * 64);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_igemm8_kernel<BN, SM, ONE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return MTE_ERR_LAUNCH;
        attr = true;
    }
    const long tiles = ((a.M + 255) / 256) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL((conv_igemm8_kernel<BN, SM, ONE>), dim3((unsigned)(tiles * a.splits)), dim3(512), LDS, st, a);
    return MTE_OK;
}

}  // namespace

int g_igemm8_one = 1;                                // development knob (mte_debug_set(28, v)): 0 = the two-state loop everywhere
// a.splits is final (1, or the number of fp32 slabs the caller's finish kernel adds); a.kslice selects the K order (ConvArgs); the caller checks the launch.
int igemm8_launch(ConvArgs a, int bn, hipStream_t st) {
    if (a.out_f32 || a.rows || a.Cin_p % 32 != 0 || a.N % 8 != 0 || (a.splits > 1 && (!a.ws || a.N % 4 != 0))) return MTE_ERR_UNSUPPORTED;
    if (a.kslice && a.Cin_p % 64 != 0) return MTE_ERR_ARG;
    if (((a.M - 1) * a.ldx + a.Cin_p) * 2 >= 0x7ff00000L || (long)a.N * a.KH * a.KW * a.Cin_p * 2 >= 0x7ff00000L || a.M >= 0x7fffff00L) return MTE_ERR_UNSUPPORTED;
#ifdef MTE_DEV
    if (a.kslice) {                                                  // (measured slower: development library only -- tests, tools/igemm8_locality.py)
        if (bn == 256) return launch8<256, true>(a, st);
        if (bn == 128) return launch8<128, true>(a, st);
        return MTE_ERR_UNSUPPORTED;
    }
#else
    if (a.kslice) return MTE_ERR_UNSUPPORTED;
#endif
    // one tap state for both K-halves where they can never straddle a tap: 64-channel granularity and a split range that starts on an even K-step
    const int ksteps = a.KH * a.KW * (a.Cin_p / 32);
    const bool one = g_igemm8_one && a.Cin_p % 64 == 0 && (a.splits == 1 || ((ksteps + a.splits - 1) / a.splits) % 2 == 0);
    if (one) {
        if (bn == 256) return launch8<256, false, true>(a, st);
        if (bn == 128) return launch8<128, false, true>(a, st);
    }
    if (bn == 256) return launch8<256, false>(a, st);
    if (bn == 128) return launch8<128, false>(a, st);
    return MTE_ERR_UNSUPPORTED;
}
