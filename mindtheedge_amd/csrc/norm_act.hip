// GroupNorm(16, C) + ELU (and the residual-sum / Dropout2d variants) for NHWC activations, gfx950.
//
//   forward : z = ELU( gamma_c * (v - mean_{b,g}) * rstd_{b,g} + beta_c ),  v = y1 + scale2[b,c] * y2
//   Replaces torch.nn.GroupNorm(16, C) + nn.ELU(inplace) of Conv2D (layers01.py:32-38) and the
//   residual tail ELU(GN(x_out + Dropout2d(conv1x1(x)))) of ResidualConv (layers01.py:62-73); scale2 is
//   the per-(sample, channel) Dropout2d factor keep/(1-p) (null = no second input scaling).
//
// Two families, both HBM-bound by construction (GroupNorm statistics are per sample, so there is no cross-GPU
// reduction anywhere in this file):
//
// * SLAB kernels (the lowest-resolution layers, 512 channels at 12x40; see GN_SLAB_MAX):
//   one workgroup owns one (sample, group) slab -- HW pixels x C/16 channels, 30-240 KB -- loads it ONCE into
//   registers, reduces on chip and writes the result: the forward is 1 read + 1 write instead of stats pass + apply
//   pass (2 reads + 1 write), the backward 2 reads + 1 write instead of reduce pass + apply pass (4 reads + 1 write),
//   with no global atomics on the data path.  The 16 group-workgroups of a sample are mapped to ONE XCD, so the
//   128-byte lines that neighbouring groups share are fetched into one L2.
// * STREAM kernels (everything larger): a thread owns one 16-byte channel chunk (8 bf16 / 4 f32 channels) and walks
//   pixels, so every wave instruction moves 1 KiB of contiguous NHWC bytes; statistics accumulate in fp32 per
//   thread, are combined per block by a FIXED tree (wave butterflies, then the waves in order) and leave the block as
//   one partial record per (sample, block); the last block of a sample to arrive adds the records up in a fixed
//   order (see gn_stats_kernel) -- the forward pass is bit-reproducible run to run and under HIP-graph replay.  The per-channel arithmetic is folded into affine constants (u = v*ka + kb for the ELU
//   argument, dv = dyh*ka - (c0 + v*c1) for the backward) and every optional input is a template flag: the
//   backward apply kernel went from 208 to <100 VGPRs (2 -> 5 waves per SIMD), which is what an HBM stream needs.
#include "common.hpp"

int g_mte_gn_prezeroed = 0;
int g_mte_handoff_fences = 0;
int g_mte_wgrad_shared = 0;
unsigned* g_mte_err_dev = nullptr;             // device address of the error word (mte_device_error_init)
static volatile unsigned* g_mte_err_host = nullptr;

namespace {

constexpr int GN_GROUPS = 16;

struct GnArgs {
    const void* y1; long ld1;
    const void* y2; long ld2;          // nullable
    const float* scale2;               // [B][C] or null
    double* stats;                     // mte_gn_stats_elems(B) doubles: [B][16][2] final (sum, sumsq), then tickets + per-block partial records (common.hpp)
    const float* gamma; const float* beta;
    void* z; long ldz;                 // forward output
    const void* dz; long lddz;         // backward input
    float* red;                        // [B][C][2]  (sum dyhat, sum dyhat*xhat)
    void* d1; long ldd1;               // backward outputs
    void* d2; long ldd2;
    float* dbias;                      // optional [C]: per-channel sum of d1 (= gradient of the conv bias in front of the norm)
    float* dgamma; float* dbeta;       // [C]
    int B, HW, C;
    float eps;
    int blocks_per_sample;
    int cps_shift;                     // slab kernels: log2(chunks per pixel inside one group)
    int reverse;                       // stream kernels: walk the samples last-to-first (see gn_zigzag)
    int b0, nb, ppl;                   // cluster kernels: this launch covers samples [b0, b0 + nb), nb <= ppl = samples a full launch takes (8, 4, 2 or 1)
    const double* stats_in;            // residual-tail kernel: the statistics of y1 (the inner GroupNorm's sums; `stats` receives those of the sum)
    unsigned* err;                     // device error word (common.hpp) or null
    int fences;                        // MTE_OPT_HANDOFF_FENCES
    unsigned spin_max;                 // bound of the cluster kernels' arrival poll
};

// Pixel rows are processed in batches of GN_U: all 16-byte loads of a batch are issued before any of its stores, so a
// wave keeps GN_U x (2..3) KiB in flight (the output pointers may alias the inputs as far as the compiler knows, which
// otherwise serialises load -> store -> load and leaves these streams latency-bound).
constexpr int GN_U = 4;

template <bool HAS2> struct RawRow;
template <> struct RawRow<false> { u32x4_t c1; };
template <> struct RawRow<true> { u32x4_t c1, c2; };

template <typename T, bool HAS2>
__device__ __forceinline__ void load_raw(const GnArgs& a, long pix, int ch0, RawRow<HAS2>& r) {
    r.c1 = *(const u32x4_t*)((const T*)a.y1 + pix * a.ld1 + ch0);
    if constexpr (HAS2) r.c2 = *(const u32x4_t*)((const T*)a.y2 + pix * a.ld2 + ch0);
}
// v = y1 + sc * y2 (sc = Dropout2d factor of this sample's channels, 1 when absent)
template <typename T, bool HAS2>
__device__ __forceinline__ void finish_v(const RawRow<HAS2>& r, const float* sc, float* v) {
    constexpr int P = Elem<T>::PER16;
    unpack16<T>(r.c1, v);
    if constexpr (HAS2) {
        float w[P];
        unpack16<T>(r.c2, w);
#pragma unroll
        for (int i = 0; i < P; ++i) v[i] += w[i] * sc[i];            // product rounded first, like Dropout2d then add
    }
}
template <typename T>
__device__ __forceinline__ void load_scale2(const GnArgs& a, int b, int ch0, float* sc) {
#pragma unroll
    for (int i = 0; i < Elem<T>::PER16; ++i) sc[i] = a.scale2 ? a.scale2[(long)b * a.C + ch0 + i] : 1.f;
}
__device__ __forceinline__ float elu_grad(float u) { return u > 0.f ? 1.f : __expf(u); }
// The slab kernels sweep their register-resident chunks several times.  Without this the compiler keeps the UNPACKED floats
// of the first sweep alive for the later ones (2x the registers of the packed chunks -> scratch spills at 1024 threads).
template <bool HAS2> __device__ __forceinline__ void keep_packed(RawRow<HAS2>& r) {
    asm volatile("" : "+v"(r.c1));
    if constexpr (HAS2) asm volatile("" : "+v"(r.c2));
}

// =====================================================================================================================
// STREAM kernels
// =====================================================================================================================
// thread -> (chunk column cc, pixel lane prow); block -> (pixel range of one sample)
#define GN_THREAD_MAP()                                                            \
    constexpr int P = Elem<T>::PER16;                                              \
    const int cpr = a.C / P;                                                       \
    const int cc = threadIdx.x % cpr, prow = threadIdx.x / cpr, rstep = 256 / cpr; \
    const int b = a.reverse ? a.B - 1 - (int)blockIdx.y : (int)blockIdx.y;         \
    const int ch0 = cc * P;                                                        \
    const int per_blk = (a.HW + a.blocks_per_sample - 1) / a.blocks_per_sample;    \
    const int p_begin = blockIdx.x * per_blk;                                      \
    const int p_end = min(a.HW, p_begin + per_blk);                                \
    const int gs = a.C / GN_GROUPS;

// The fp64 mean / rstd of the 16 groups of sample b are evaluated once per block and shared through LDS (per-thread
// evaluation -- 8 fp64 divisions + square roots per thread -- used to dominate the short low-resolution launches).
__device__ __forceinline__ void block_group_stats(const double* stats, const GnArgs& a, int b, int gs, float* s_mr) {
    if (threadIdx.x < GN_GROUPS) {
        const double* sp = stats + ((long)b * GN_GROUPS + threadIdx.x) * 2;
        const double n = (double)a.HW * gs;
        const double m = sp[0] / n;
        double var = sp[1] / n - m * m;
        if (var < 0.0) var = 0.0;
        s_mr[2 * threadIdx.x] = (float)m;
        s_mr[2 * threadIdx.x + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
    __syncthreads();
}

// Statistics pass.  Bit-reproducible by construction (GPUTEST_r02: the LDS / fp64 atomics this replaces made two forward passes
// of the same frame differ by 1-2 % in inverse depth -- a one-ulp change of one statistic is amplified by ~60 bf16 layers):
//   thread   : fp32 sums over its pixels of its 16-byte channel chunk (fixed order)
//   wave     : xor butterflies over the lanes that share a chunk column, then over the chunks of one group
//   block    : the waves' group sums added in wave order -> ONE record of 32 doubles, written (returning agent-scope exchange:
//              performed at the memory side before the ticket is drawn) into this block's own slot
//   sample   : the block that draws the last ticket of sample b adds the records in slot order -> stats[b][16][2]
// Grid (blocks_per_sample <= MTE_GN_SLOTS(B), B), NT = 1024 threads so that <= 64 blocks per sample still fill the chip at B = 8.
//
// TAIL = true (round 5): the residual block's tail (layers01.py:62-73).  y1 is the inner Conv2D's convolution output with statistics
// `stats_in` and affine (gamma, beta); the kernel forms t = ELU(GN(y1)) + scale2 * y2 -- the sum the block normalises -- stores it (rounded to the
// storage type) into z and takes the statistics of the STORED values: the inner layer's apply pass, the stand-alone statistics pass over two
// tensors and the second tensor read of the outer apply / both backward passes are gone (see mte_gn_tail_fwd).
template <typename T, bool HAS2, int NT, bool TAIL = false>
__global__ __launch_bounds__(NT) void gn_stats_kernel(GnArgs a) {
    constexpr int P = Elem<T>::PER16;
    constexpr int NW = NT / 64;
    const int cpr = a.C / P;
    const int cc = threadIdx.x % cpr, prow = threadIdx.x / cpr, rstep = NT / cpr;
    const int b = a.reverse ? a.B - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const int ch0 = cc * P;
    const int per_blk = (a.HW + a.blocks_per_sample - 1) / a.blocks_per_sample;
    const int p_begin = blockIdx.x * per_blk;
    const int p_end = min(a.HW, p_begin + per_blk);
    const int gs = a.C / GN_GROUPS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float s_part[NW * GN_GROUPS * 2];
    __shared__ double s_fin[(NT / 32) * GN_GROUPS * 2];
    __shared__ int s_last;
    for (int i = threadIdx.x; i < NW * GN_GROUPS * 2; i += NT) s_part[i] = 0.f;
    float s[P], q[P];
#pragma unroll
    for (int i = 0; i < P; ++i) { s[i] = 0.f; q[i] = 0.f; }
    float sc[P];
    if constexpr (HAS2) load_scale2<T>(a, b, ch0, sc);
    float ka[P], kb[P];
    if constexpr (TAIL) {
        static_assert(HAS2, "the tail adds a second tensor");
        __shared__ float s_mr[GN_GROUPS * 2];
        block_group_stats(a.stats_in, a, b, gs, s_mr);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int g = (ch0 + i) / gs;
            const float mean = s_mr[2 * g], rstd = s_mr[2 * g + 1];
            const float gm = a.gamma[ch0 + i];
            ka[i] = rstd * gm;
            kb[i] = a.beta[ch0 + i] - mean * rstd * gm;
        }
    }
    for (int p = p_begin + prow; p < p_end; p += rstep * GN_U) {
        RawRow<HAS2> raw[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u)
            if (p + u * rstep < p_end) load_raw<T, HAS2>(a, (long)b * a.HW + p + u * rstep, ch0, raw[u]);
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            if (p + u * rstep >= p_end) break;
            float v[P];
            if constexpr (TAIL) {
                float w[P];
                unpack16<T>(raw[u].c1, v);
                unpack16<T>(raw[u].c2, w);
#pragma unroll
                for (int i = 0; i < P; ++i) v[i] = elu1(fmaf(v[i], ka[i], kb[i])) + w[i] * sc[i];
                const u32x4_t pk = pack16<T>(v);
                *(u32x4_t*)((T*)a.z + ((long)b * a.HW + p + u * rstep) * a.ldz + ch0) = pk;
                unpack16<T>(pk, v);                        // the statistics of what every later pass reads back
            } else finish_v<T, HAS2>(raw[u], sc, v);
#pragma unroll
            for (int i = 0; i < P; ++i) { s[i] += v[i]; q[i] = fmaf(v[i], v[i], q[i]); }
        }
    }
    // channels of one group held by this thread: running sums, complete at the group's last channel (`last` below)
#pragma unroll
    for (int i = 0; i + 1 < P; ++i)
        if ((ch0 + i + 1) / gs == (ch0 + i) / gs) { s[i + 1] += s[i]; q[i + 1] += q[i]; }
    // lanes of this wave that hold the same chunk column (other pixel rows): butterfly (every lane ends with the total)
    if (cpr < 64) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            for (int off = cpr; off < 64; off <<= 1) { s[i] += __shfl_xor(s[i], off, 64); q[i] += __shfl_xor(q[i], off, 64); }
    }
    // chunks of one group (gs > P: the group total so far sits in element P - 1 of each of its gs / P chunk columns)
    const int cpg = gs > P ? gs / P : 1;
    for (int off = 1; off < cpg; off <<= 1) { s[P - 1] += __shfl_xor(s[P - 1], off, 64); q[P - 1] += __shfl_xor(q[P - 1], off, 64); }
    __syncthreads();                                       // s_part is zero
    if (lane < (cpr < 64 ? cpr : 64) && cc % cpg == 0) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int g = (ch0 + i) / gs;
            const bool last = (i == P - 1) || ((ch0 + i + 1) / gs != g);
            if (last) { s_part[(wave * GN_GROUPS + g) * 2] = s[i]; s_part[(wave * GN_GROUPS + g) * 2 + 1] = q[i]; }
        }
    }
    __syncthreads();
    const int slots = MTE_GN_SLOTS(a.B);
    double* part = mte_gn_partials(a.stats, a.B) + ((long)b * slots + blockIdx.x) * (GN_GROUPS * 2);
    if (threadIdx.x < GN_GROUPS * 2) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += s_part[w * GN_GROUPS * 2 + threadIdx.x];
        const unsigned long long before = atomicExch((unsigned long long*)part + threadIdx.x, (unsigned long long)__double_as_longlong((double)tot));
        asm volatile("" ::"v"(before));                    // returning: the record is at the memory side once the value is back
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // (TAIL writes its output through this XCD's L2: a release there would also write those lines back, per workgroup -- the records are
        //  returning exchanges, at the memory side already, so the tail keeps the relaxed ticket whatever the option says)
        if (a.fences && !TAIL) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned old = __hip_atomic_fetch_add(mte_gn_tickets(a.stats, a.B) + 2 * b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == gridDim.x - 1;
        if (s_last) {
            // the last arriver puts the ticket back: a statistics buffer can be used again without being cleared (MTE_OPT_GN_PREZEROED is
            // then an optimisation of the FIRST use, not a precondition of every use)
            __hip_atomic_store(mte_gn_tickets(a.stats, a.B) + 2 * b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.fences) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
    }
    __syncthreads();
    if (!s_last) return;
    // every record of sample b is in memory: NT / 32 threads per value take the slots j = k, k + NT/32, ... in order, then the
    // NT / 32 partial sums are added in order
    {
        const int v = threadIdx.x & 31, k = threadIdx.x >> 5;
        const double* rec = mte_gn_partials(a.stats, a.B) + (long)b * slots * (GN_GROUPS * 2) + v;
        double acc = 0.0;
        for (int j = k; j < (int)gridDim.x; j += NT / 32)
            acc += __hip_atomic_load(rec + (long)j * (GN_GROUPS * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_fin[k * 32 + v] = acc;
    }
    __syncthreads();
    if (threadIdx.x < GN_GROUPS * 2) {
        double tot = 0.0;
#pragma unroll 8
        for (int k = 0; k < NT / 32; ++k) tot += s_fin[k * 32 + threadIdx.x];
        a.stats[(long)b * GN_GROUPS * 2 + threadIdx.x] = tot;
    }
}

template <typename T, bool HAS2>
__global__ __launch_bounds__(256) void gn_elu_fwd_kernel(GnArgs a) {
    GN_THREAD_MAP();
    __shared__ float s_mr[GN_GROUPS * 2];
    block_group_stats(a.stats, a, b, gs, s_mr);
    float ka[P], kb[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int g = (ch0 + i) / gs;
        const float mean = s_mr[2 * g], rstd = s_mr[2 * g + 1];
        const float gm = a.gamma[ch0 + i];
        ka[i] = rstd * gm;
        kb[i] = a.beta[ch0 + i] - mean * rstd * gm;
    }
    float sc[P];
    if constexpr (HAS2) load_scale2<T>(a, b, ch0, sc);
    for (int p = p_begin + prow; p < p_end; p += rstep * GN_U) {
        RawRow<HAS2> raw[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u)
            if (p + u * rstep < p_end) load_raw<T, HAS2>(a, (long)b * a.HW + p + u * rstep, ch0, raw[u]);
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            if (p + u * rstep >= p_end) break;
            float v[P];
            finish_v<T, HAS2>(raw[u], sc, v);
#pragma unroll
            for (int i = 0; i < P; ++i) v[i] = elu1(fmaf(v[i], ka[i], kb[i]));
            *(u32x4_t*)((T*)a.z + ((long)b * a.HW + p + u * rstep) * a.ldz + ch0) = pack16<T>(v);
        }
    }
}

// pass 1 of the backward: red[b][c] = (sum_px dyhat, sum_px dyhat * xhat), dyhat = dz * elu'(u), u = v*ka + kb,
// xhat = v*xa + xb (xa = rstd, xb = -mean*rstd of the channel's group)
template <typename T, bool HAS2>
__global__ __launch_bounds__(256) void gn_elu_bwd_reduce_kernel(GnArgs a) {
    GN_THREAD_MAP();
    extern __shared__ float s_red[];                      // [C][2]
    __shared__ float s_mr[GN_GROUPS * 2];
    for (int i = threadIdx.x; i < a.C * 2; i += 256) s_red[i] = 0.f;
    block_group_stats(a.stats, a, b, gs, s_mr);
    float ka[P], kb[P], xa[P], xb[P], r1[P], r2[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const float mean = s_mr[2 * ((ch0 + i) / gs)], rstd = s_mr[2 * ((ch0 + i) / gs) + 1];
        const float gm = a.gamma[ch0 + i];
        xa[i] = rstd; xb[i] = -mean * rstd;
        ka[i] = rstd * gm; kb[i] = a.beta[ch0 + i] - mean * rstd * gm;
        r1[i] = 0.f; r2[i] = 0.f;
    }
    float sc[P];
    if constexpr (HAS2) load_scale2<T>(a, b, ch0, sc);
    for (int p = p_begin + prow; p < p_end; p += rstep * GN_U) {
        RawRow<HAS2> raw[GN_U];
        u32x4_t gr[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u)
            if (p + u * rstep < p_end) {
                const long pix = (long)b * a.HW + p + u * rstep;
                load_raw<T, HAS2>(a, pix, ch0, raw[u]);
                gr[u] = *(const u32x4_t*)((const T*)a.dz + pix * a.lddz + ch0);
            }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            if (p + u * rstep >= p_end) break;
            float v[P], g[P];
            finish_v<T, HAS2>(raw[u], sc, v);
            unpack16<T>(gr[u], g);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float dyh = g[i] * elu_grad(fmaf(v[i], ka[i], kb[i]));
                r1[i] += dyh; r2[i] = fmaf(dyh, fmaf(v[i], xa[i], xb[i]), r2[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) { atomicAdd(&s_red[2 * (ch0 + i)], r1[i]); atomicAdd(&s_red[2 * (ch0 + i) + 1], r2[i]); }
    __syncthreads();
    for (int i = threadIdx.x; i < a.C * 2; i += 256) atomicAdd(&a.red[(long)b * a.C * 2 + i], s_red[i]);
}

// pass 2: dv = rstd * (dyhat*gamma - (S1 + xhat*S2)/n) = dyhat*ka - (c0 + v*c1),
//         S1 = sum_{c in g} gamma_c r1, S2 = sum gamma_c r2, c1 = rstd^2 S2/n, c0 = rstd S1/n - mean c1
// M2: 0 = one input; 1 = v = y1 + scale2 * y2 (HAS2: the second tensor is read); 2 (round 5) = one input, but the gradient leaves twice:
//     d1 = dv and d2 = scale2[b,c] * dv (and dbias = column sums of d2) -- the residual tail after mte_gn_tail_fwd, whose input is the stored sum.
template <typename T, int M2, bool HASDB>
__global__ __launch_bounds__(256, M2 == 1 ? 3 : 4) void gn_elu_bwd_apply_kernel(GnArgs a) {      // (4: left alone, the two-output form M2 = 2 took 144 VGPRs = three waves per SIMD; the two-input form would spill at 128)
    constexpr bool HAS2 = M2 == 1, SCL = M2 != 0;
    GN_THREAD_MAP();
    extern __shared__ float s_db[];                       // [C] when HASDB
    __shared__ float s_mr[GN_GROUPS * 2], s_S[GN_GROUPS * 2];
    if constexpr (HASDB)
        for (int i = threadIdx.x; i < a.C; i += 256) s_db[i] = 0.f;
    if (threadIdx.x < GN_GROUPS * 2) s_S[threadIdx.x] = 0.f;
    block_group_stats(a.stats, a, b, gs, s_mr);
    // S1_g = sum_{c in g} gamma_c r1_c, S2_g likewise: one pass of the block over the C channels of this sample
    for (int c = threadIdx.x; c < a.C; c += 256) {
        const float gc = a.gamma[c];
        const float2 r = *(const float2*)(a.red + ((long)b * a.C + c) * 2);
        atomicAdd(&s_S[2 * (c / gs)], gc * r.x); atomicAdd(&s_S[2 * (c / gs) + 1], gc * r.y);
    }
    __syncthreads();
    if (blockIdx.x == 0 && b == 0) {                      // dgamma_c = sum_b r2, dbeta_c = sum_b r1 (red is complete: pass 1 ran before)
        for (int c = threadIdx.x; c < a.C; c += 256) {
            float g = 0.f, bsum = 0.f;
            for (int bb = 0; bb < a.B; ++bb) {
                const float2 r = *(const float2*)(a.red + ((long)bb * a.C + c) * 2);
                bsum += r.x; g += r.y;
            }
            a.dgamma[c] = g; a.dbeta[c] = bsum;
        }
    }
    float ka[P], kb[P], c0[P], c1[P], sc[P], db[P];
    const float inv_n = 1.f / ((float)a.HW * gs);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int g = (ch0 + i) / gs;
        const float mean = s_mr[2 * g], rstd = s_mr[2 * g + 1];
        const float gm = a.gamma[ch0 + i];
        ka[i] = rstd * gm; kb[i] = a.beta[ch0 + i] - mean * rstd * gm;
        c1[i] = rstd * rstd * (s_S[2 * g + 1] * inv_n);
        c0[i] = rstd * (s_S[2 * g] * inv_n) - mean * c1[i];
        if constexpr (HASDB) db[i] = 0.f;
    }
    if constexpr (SCL) load_scale2<T>(a, b, ch0, sc);
    for (int p = p_begin + prow; p < p_end; p += rstep * GN_U) {
        RawRow<HAS2> raw[GN_U];
        u32x4_t gr[GN_U];
#pragma unroll
        for (int u = 0; u < GN_U; ++u)
            if (p + u * rstep < p_end) {
                const long pix = (long)b * a.HW + p + u * rstep;
                load_raw<T, HAS2>(a, pix, ch0, raw[u]);
                gr[u] = *(const u32x4_t*)((const T*)a.dz + pix * a.lddz + ch0);
            }
#pragma unroll
        for (int u = 0; u < GN_U; ++u) {
            if (p + u * rstep >= p_end) break;
            const long pix = (long)b * a.HW + p + u * rstep;
            float v[P], g[P];
            finish_v<T, HAS2>(raw[u], sc, v);
            unpack16<T>(gr[u], g);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float dyh = g[i] * elu_grad(fmaf(v[i], ka[i], kb[i]));
                v[i] = dyh * ka[i] - fmaf(v[i], c1[i], c0[i]);
                if constexpr (HASDB) db[i] += v[i];
            }
            *(u32x4_t*)((T*)a.d1 + pix * a.ldd1 + ch0) = pack16<T>(v);
            if constexpr (SCL) {
                if (a.d2) {
#pragma unroll
                    for (int i = 0; i < P; ++i) v[i] *= sc[i];
                    *(u32x4_t*)((T*)a.d2 + pix * a.ldd2 + ch0) = pack16<T>(v);
                }
            }
        }
    }
    if constexpr (HASDB) {
        // with a second input the bias gradient asked for is that of the SECOND input's producer (the 1x1 shortcut conv of a residual
        // block): sum of d2 = scale2[b][c] * d1 over the pixels -- the factor is constant per (sample, channel), so it multiplies the sum
#pragma unroll
        for (int i = 0; i < P; ++i) atomicAdd(&s_db[ch0 + i], SCL ? db[i] * sc[i] : db[i]);
        __syncthreads();
        for (int i = threadIdx.x; i < a.C; i += 256) atomicAdd(&a.dbias[i], s_db[i]);
    }
}

// =====================================================================================================================
// SLAB kernels: workgroup = one (sample b, group g) slab held in registers
// =====================================================================================================================
// block id -> (b, g) with the 16 groups of a sample on one XCD (blocks L and L+8 share an XCD): neighbouring groups share
// 128-byte lines (a group is 16-64 bytes of a pixel), so their fetches meet in one L2 instead of eight.
#define GN_SLAB_MAP()                                                                                  \
    constexpr int P = Elem<T>::PER16;                                                                  \
    const int L_ = blockIdx.x, k_ = L_ >> 3;                                                           \
    const int b = (L_ & 7) + 8 * (k_ / GN_GROUPS), g = k_ % GN_GROUPS;                                 \
    if (b >= a.B) return;                                                                              \
    const int gs = a.C / GN_GROUPS;                                                                    \
    const int cps = 1 << a.cps_shift;                      /* 16-byte chunks per pixel in this group */ \
    const int nchunks = a.HW << a.cps_shift;                                                           \
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;                                           \
    const int ch0 = g * gs + (t & (cps - 1)) * P;          /* NT % cps == 0: a thread keeps its column */

// sum over the block of one float per thread -> double, broadcast to all threads (s_w: NT/64 floats of LDS)
template <int NT>
__device__ __forceinline__ double block_sum(float v, float* s_w, int lane, int wave) {
    v = wave_sum(v);
    __syncthreads();                                       // s_w free again
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot += (double)s_w[w];
    return tot;
}

template <typename T, bool HAS2, int NCH, int NT>
__global__ __launch_bounds__(NT) void gn_fwd_slab_kernel(GnArgs a) {
    GN_SLAB_MAP();
    __shared__ float s_w[NT / 64];
    RawRow<HAS2> raw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) load_raw<T, HAS2>(a, (long)b * a.HW + (id >> a.cps_shift), ch0, raw[i]);
    }
    float sc[P];
    if constexpr (HAS2) load_scale2<T>(a, b, ch0, sc);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (t + i * NT < nchunks) {
            float v[P];
            finish_v<T, HAS2>(raw[i], sc, v);
#pragma unroll
            for (int k = 0; k < P; ++k) s += v[k];
        }
    }
    const double n = (double)a.HW * gs;
    const double S = block_sum<NT>(s, s_w, lane, wave);
    const float mean = (float)(S / n);
    float q = 0.f;                                         // centred second moment: the slab is on chip, a second sweep is free
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (t + i * NT < nchunks) {
            float v[P];
            keep_packed<HAS2>(raw[i]);
            finish_v<T, HAS2>(raw[i], sc, v);
#pragma unroll
            for (int k = 0; k < P; ++k) { const float d = v[k] - mean; q = fmaf(d, d, q); }
        }
    }
    const double Q = block_sum<NT>(q, s_w, lane, wave);
    // statistics in the stream kernels' format (sum, sum of squares):
    // sum (v - m)^2 = sum v^2 - 2 m S + n m^2 for ANY m, so sum v^2 = Q + 2 m S - n m^2 exactly
    const double sumsq = Q + 2.0 * (double)mean * S - n * (double)mean * (double)mean;
    if (t < 2) a.stats[((long)b * GN_GROUPS + g) * 2 + t] = t ? sumsq : S;
    const double md = S / n;
    double var = sumsq / n - md * md;                      // what the consumers of `stats` will compute
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)md;
    float ka[P], kb[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float gm = a.gamma[ch0 + k];
        ka[k] = rstd * gm;
        kb[k] = a.beta[ch0 + k] - meanf * rstd * gm;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) {
            float v[P];
            keep_packed<HAS2>(raw[i]);
            finish_v<T, HAS2>(raw[i], sc, v);
#pragma unroll
            for (int k = 0; k < P; ++k) v[k] = elu1(fmaf(v[k], ka[k], kb[k]));
            *(u32x4_t*)((T*)a.z + ((long)b * a.HW + (id >> a.cps_shift)) * a.ldz + ch0) = pack16<T>(v);
        }
    }
}

// per-channel sums over the block: x[k] of threads that share a chunk column (t % cps) -> s_ch[(t % cps) * P + k]
// (one shuffle tree over the lanes of equal column, one LDS slot per wave, then the first gs threads add the waves up)
template <int NT, int P>
__device__ __forceinline__ void block_channel_sums(float* x, int cps, int lane, int wave, float* s_part, float* s_ch, int gs) {
#pragma unroll
    for (int k = 0; k < P; ++k)
        for (int off = cps; off < 64; off <<= 1) x[k] += __shfl_xor(x[k], off, 64);
    __syncthreads();                                       // s_part / s_ch free again
    if (lane < cps) {
#pragma unroll
        for (int k = 0; k < P; ++k) s_part[wave * 32 + lane * P + k] = x[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < gs) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += s_part[w * 32 + threadIdx.x];
        s_ch[threadIdx.x] = tot;
    }
    __syncthreads();
}

template <typename T, int M2, bool HASDB, int NCH, int NT>
__global__ __launch_bounds__(NT) void gn_bwd_slab_kernel(GnArgs a) {
    constexpr bool HAS2 = M2 == 1, SCL = M2 != 0;
    GN_SLAB_MAP();
    __shared__ float s_part[(NT / 64) * 32];               // per-wave channel partials (a group has <= 32 channels)
    __shared__ float s_r1[32], s_r2[32], s_db[32], s_S[2];
    __shared__ double s_st[2];
    if (t < 2) s_st[t] = a.stats[((long)b * GN_GROUPS + g) * 2 + t];      // this group's (sum, sum of squares)
    RawRow<HAS2> raw[NCH];
    u32x4_t gr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) {
            const long pix = (long)b * a.HW + (id >> a.cps_shift);
            load_raw<T, HAS2>(a, pix, ch0, raw[i]);
            gr[i] = *(const u32x4_t*)((const T*)a.dz + pix * a.lddz + ch0);
        }
    }
    __syncthreads();
    const double n = (double)a.HW * gs;
    const double md = s_st[0] / n;
    double var = s_st[1] / n - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float xa = rstd, xb = -mean * rstd;
    float ka[P], kb[P], sc[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float gm = a.gamma[ch0 + k];
        ka[k] = rstd * gm; kb[k] = a.beta[ch0 + k] - mean * rstd * gm;
    }
    if constexpr (SCL) load_scale2<T>(a, b, ch0, sc);
    float r1[P], r2[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { r1[k] = 0.f; r2[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (t + i * NT < nchunks) {
            float v[P], gz[P];
            finish_v<T, HAS2>(raw[i], sc, v);
            unpack16<T>(gr[i], gz);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float dyh = gz[k] * elu_grad(fmaf(v[k], ka[k], kb[k]));
                r1[k] += dyh; r2[k] = fmaf(dyh, fmaf(v[k], xa, xb), r2[k]);
            }
        }
    }
    block_channel_sums<NT, P>(r1, cps, lane, wave, s_part, s_r1, gs);
    block_channel_sums<NT, P>(r2, cps, lane, wave, s_part, s_r2, gs);
    if (t < 64) {                                          // S1 = sum_c gamma_c r1_c, S2 = sum_c gamma_c r2_c over the group's channels
        const float gm = t < gs ? a.gamma[g * gs + t] : 0.f;
        const float S1 = wave_sum(t < gs ? gm * s_r1[t] : 0.f), S2 = wave_sum(t < gs ? gm * s_r2[t] : 0.f);
        if (t == 0) { s_S[0] = S1; s_S[1] = S2; }
        if (t < gs) {                                      // dgamma / dbeta: sums over the batch (zeroed by the launcher)
            atomicAdd(&a.dgamma[g * gs + t], s_r2[t]);
            atomicAdd(&a.dbeta[g * gs + t], s_r1[t]);
        }
    }
    __syncthreads();
    const float inv_n = 1.f / ((float)a.HW * gs);
    const float c1 = rstd * rstd * (s_S[1] * inv_n);
    const float c0 = rstd * (s_S[0] * inv_n) - mean * c1;
    float db[P];
#pragma unroll
    for (int k = 0; k < P; ++k) db[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) {
            const long pix = (long)b * a.HW + (id >> a.cps_shift);
            float v[P], gz[P];
            keep_packed<HAS2>(raw[i]);
            asm volatile("" : "+v"(gr[i]));
            finish_v<T, HAS2>(raw[i], sc, v);
            unpack16<T>(gr[i], gz);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float dyh = gz[k] * elu_grad(fmaf(v[k], ka[k], kb[k]));
                v[k] = dyh * ka[k] - fmaf(v[k], c1, c0);
                if constexpr (HASDB) db[k] += v[k];
            }
            *(u32x4_t*)((T*)a.d1 + pix * a.ldd1 + ch0) = pack16<T>(v);
            if constexpr (SCL) {
                if (a.d2) {
#pragma unroll
                    for (int k = 0; k < P; ++k) v[k] *= sc[k];
                    *(u32x4_t*)((T*)a.d2 + pix * a.ldd2 + ch0) = pack16<T>(v);
                }
            }
        }
    }
    if constexpr (HASDB) {
        block_channel_sums<NT, P>(db, cps, lane, wave, s_part, s_db, gs);
        // (with a scale the bias gradient asked for is that of the scaled copy d2: the factor is constant per (sample, channel))
        if (t < gs) atomicAdd(&a.dbias[g * gs + t], SCL && a.scale2 ? s_db[t] * a.scale2[(long)b * a.C + g * gs + t] : s_db[t]);
    }
}

// =====================================================================================================================
// CLUSTER kernels (round 4): the slab idea for the layers whose (sample, group) slab is too large for ONE workgroup -- 512
// channels at 24x80 (123 KB), 256 at 48x160 (246 KB), 128 at 96x320 (491 KB, forward) -- where the two dependent streaming
// launches of 12-50 us each run at 1.3-3 TB/s.  CL workgroups of 256 threads share a slab, each holds its pixel range in
// registers; the partial sums meet through a few words of global memory:
//   every workgroup: partials -> its record (agent-scope relaxed stores = write-through), s_waitcnt vmcnt(0), ticket += 1;
//                    thread 0 polls the ticket (agent-scope relaxed loads) until all CL arrived, then the records are read
//                    with agent-scope loads and added IN SLOT ORDER by every workgroup (same totals everywhere, bit-reproducible).
// Forward = 1 read + 1 write (was 2 + 1), backward = 2 reads + 1 write (was 4 + 1).  The hand-off needs every workgroup of a
// cluster resident at the same time: a launch has <= 1024 LIVE workgroups of 256 threads with <= 128 VGPRs (4 per CU; a batch whose
// clusters exceed that goes out in 2-4 launches over sample ranges),
// the members of a cluster are 8 block ids apart (one XCD, dispatched together), and the poll is bounded.  The exchange words:
// forward -- this (sample, group)'s share of the statistics buffer's record area (zero at entry: arena / cleared by the
// launcher; the last reader puts the two counters back to zero); backward -- counters in `red` (zero at entry), records (two
// floats per workgroup: its share of the group sums) in the same record area (overwritten, not accumulated).
// =====================================================================================================================
#define GN_CLUSTER_MAP()                                                                               \
    constexpr int P = Elem<T>::PER16;                                                                  \
    constexpr int NT = 256;                                                                            \
    const int L_ = blockIdx.x, k_ = L_ >> 3;                                                           \
    /* a launch covers `ppl` = 8, 4, 2 or 1 samples: sample s owns 8 / ppl XCD labels, each with 16 * ppl / 8 of its groups (whole   */ \
    /* 128-byte lines: neighbouring groups meet in one L2); the CL members of a cluster are 8 block ids apart (one label)             */ \
    const int nx_ = 8 / a.ppl, gp_ = GN_GROUPS / nx_;                                                  \
    const int w = k_ % CL, g = ((L_ & 7) % nx_) * gp_ + (k_ / CL) % gp_, b = a.b0 + (L_ & 7) / nx_;    \
    if (b >= a.b0 + a.nb) return;                                                                      \
    const int gs = a.C / GN_GROUPS;                                                                    \
    const int cps = 1 << a.cps_shift;                                                                  \
    const int ppw = (a.HW + CL - 1) / CL;                  /* pixels per workgroup */                  \
    const int p0 = w * ppw, p1 = min(a.HW, p0 + ppw);                                                  \
    const int nchunks = max(0, p1 - p0) << a.cps_shift;                                                \
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;                                           \
    const int ch0 = g * gs + (t & (cps - 1)) * P;                                                      \
    double* xch = mte_gn_partials(a.stats, a.B) + ((long)b * MTE_GN_SLOTS(a.B) * 32) + (long)g * (MTE_GN_SLOTS(a.B) * 2);

constexpr unsigned GN_SPIN_MAX = 1u << 24;
unsigned g_gn_spin_max = GN_SPIN_MAX;              // development knob (mte_debug_set(25, 1000 + n) -> n polls): tests force the give-up path with it

// thread 0: all CL workgroups of the cluster have published.  Bounded: a hand-off that cannot complete (a member of the cluster not resident --
// the launch geometry keeps them resident, but nothing in HIP promises it) gives up instead of hanging the GPU; the caller then REPORTS it
// through the device error word and the host fails the step (round 4 went on with whatever records there were: wrong statistics, no error).
__device__ __forceinline__ bool cluster_wait(unsigned* ticket, unsigned want, unsigned spin_max) {
    unsigned spins = 0;
    while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (++spins >= spin_max) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}
// thread 0: this workgroup's record, two values.  RETURNING exchanges: a value comes back only after the store has been performed at the memory
// side (edge_loss.hip saw a ticket overtake plain and no-return forms once per few thousand workgroups), then the wait, then -- by the caller --
// the ticket.  Readers use agent-scope atomic loads (never a CU's L1).
template <typename V> __device__ __forceinline__ void cluster_publish(V* rec, V v0, V v1) {
    if constexpr (sizeof(V) == 8) {
        const unsigned long long r0 = atomicExch((unsigned long long*)rec, (unsigned long long)__double_as_longlong((double)v0));
        const unsigned long long r1 = atomicExch((unsigned long long*)rec + 1, (unsigned long long)__double_as_longlong((double)v1));
        asm volatile("" ::"v"(r0), "v"(r1));
    } else {
        const unsigned r0 = atomicExch((unsigned*)rec, __float_as_uint((float)v0));
        const unsigned r1 = atomicExch((unsigned*)rec + 1, __float_as_uint((float)v1));
        asm volatile("" ::"v"(r0), "v"(r1));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// thread 0, after its workgroup has read the records: the last reader of the cluster zeroes both counters (the buffer can be used again)
__device__ __forceinline__ void cluster_done(unsigned* ticket, unsigned* done, unsigned cl) {
    if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cl - 1) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, bool HAS2, int NCH, int CL>
__global__ __launch_bounds__(256, 4) void gn_fwd_cluster_kernel(GnArgs a) {
    GN_CLUSTER_MAP();
    __shared__ float s_w[NT / 64];
    __shared__ double s_tot[2];
    RawRow<HAS2> raw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) load_raw<T, HAS2>(a, (long)b * a.HW + p0 + (id >> a.cps_shift), ch0, raw[i]);
    }
    float sc[P];
    if constexpr (HAS2) load_scale2<T>(a, b, ch0, sc);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (t + i * NT < nchunks) {
            float v[P];
            finish_v<T, HAS2>(raw[i], sc, v);
#pragma unroll
            for (int k = 0; k < P; ++k) s += v[k];
        }
    }
    const double nw = (double)max(0, p1 - p0) * gs;
    const double Sw = block_sum<NT>(s, s_w, lane, wave);
    const float mw = nw > 0.0 ? (float)(Sw / nw) : 0.f;    // this workgroup's mean: its second moment is centred on it
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (t + i * NT < nchunks) {
            float v[P];
            keep_packed<HAS2>(raw[i]);
            finish_v<T, HAS2>(raw[i], sc, v);
#pragma unroll
            for (int k = 0; k < P; ++k) { const float d = v[k] - mw; q = fmaf(d, d, q); }
        }
    }
    const double Qw = block_sum<NT>(q, s_w, lane, wave);
    unsigned* ticket = (unsigned*)xch; unsigned* done = (unsigned*)(xch + 1);
    if (t == 0) {
        cluster_publish(xch + 2 + 2 * w, Sw, Qw);
        __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!cluster_wait(ticket, CL, a.spin_max)) mte_report_device_error(a.err, MTE_DEVERR_GN_CLUSTER_FWD);
        // sum v^2 over the slab from the per-workgroup (S_w, Q_w = sum (v - m_w)^2 about m_w = (float)(S_w / n_w)):
        // sum_w v^2 = Q_w + 2 m_w S_w - n_w m_w^2 for ANY m_w, exactly -- one pass over the records, in slot order
        double S = 0.0, SQ = 0.0;
#pragma unroll
        for (int k = 0; k < CL; ++k) {
            const double sk = __hip_atomic_load(xch + 2 + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double qk = __hip_atomic_load(xch + 3 + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double nk = (double)max(0, min(a.HW, (k + 1) * ppw) - k * ppw) * gs;
            const double mk = nk > 0.0 ? (double)(float)(sk / nk) : 0.0;
            S += sk;
            SQ += qk + 2.0 * mk * sk - nk * mk * mk;
        }
        s_tot[0] = S; s_tot[1] = SQ;                       // (sum, sum of squares): the stream kernels' format
        cluster_done(ticket, done, CL);
    }
    __syncthreads();
    const double n = (double)a.HW * gs, S = s_tot[0], sumsq = s_tot[1];
    if (w == 0 && t < 2) a.stats[((long)b * GN_GROUPS + g) * 2 + t] = t ? sumsq : S;
    const double md = S / n;
    double var = sumsq / n - md * md;                      // what the consumers of `stats` will compute
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float meanf = (float)md;
    float ka[P], kb[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float gm = a.gamma[ch0 + k];
        ka[k] = rstd * gm;
        kb[k] = a.beta[ch0 + k] - meanf * rstd * gm;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) {
            float v[P];
            keep_packed<HAS2>(raw[i]);
            finish_v<T, HAS2>(raw[i], sc, v);
#pragma unroll
            for (int k = 0; k < P; ++k) v[k] = elu1(fmaf(v[k], ka[k], kb[k]));
            *(u32x4_t*)((T*)a.z + ((long)b * a.HW + p0 + (id >> a.cps_shift)) * a.ldz + ch0) = pack16<T>(v);
        }
    }
}

template <typename T, int M2, bool HASDB, int NCH, int CL>
__global__ __launch_bounds__(256, 4) void gn_bwd_cluster_kernel(GnArgs a) {
    constexpr bool HAS2 = M2 == 1, SCL = M2 != 0;
    GN_CLUSTER_MAP();
    __shared__ float s_part[(NT / 64) * 32];
    __shared__ float s_r1[32], s_r2[32], s_db[32], s_S[2];
    __shared__ double s_st[2];
    if (t < 2) s_st[t] = a.stats[((long)b * GN_GROUPS + g) * 2 + t];
    RawRow<HAS2> raw[NCH];
    u32x4_t gr[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) {
            const long pix = (long)b * a.HW + p0 + (id >> a.cps_shift);
            load_raw<T, HAS2>(a, pix, ch0, raw[i]);
            gr[i] = *(const u32x4_t*)((const T*)a.dz + pix * a.lddz + ch0);
        }
    }
    __syncthreads();
    const double n = (double)a.HW * gs;
    const double md = s_st[0] / n;
    double var = s_st[1] / n - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = (float)md, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float xa = rstd, xb = -mean * rstd;
    float ka[P], kb[P], sc[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const float gm = a.gamma[ch0 + k];
        ka[k] = rstd * gm; kb[k] = a.beta[ch0 + k] - mean * rstd * gm;
    }
    if constexpr (SCL) load_scale2<T>(a, b, ch0, sc);
    float r1[P], r2[P];
#pragma unroll
    for (int k = 0; k < P; ++k) { r1[k] = 0.f; r2[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        if (t + i * NT < nchunks) {
            float v[P], gz[P];
            finish_v<T, HAS2>(raw[i], sc, v);
            unpack16<T>(gr[i], gz);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float dyh = gz[k] * elu_grad(fmaf(v[k], ka[k], kb[k]));
                r1[k] += dyh; r2[k] = fmaf(dyh, fmaf(v[k], xa, xb), r2[k]);
            }
        }
    }
    block_channel_sums<NT, P>(r1, cps, lane, wave, s_part, s_r1, gs);
    block_channel_sums<NT, P>(r2, cps, lane, wave, s_part, s_r2, gs);
    // ---- the apply pass needs the GROUP sums S1 = sum_c gamma_c r1_c, S2 = sum_c gamma_c r2_c of the whole slab: every workgroup publishes
    // its two partial sums (record area, 2 floats per workgroup; counters in `red`) and adds the CL records up in slot order; the per-channel
    // sums only feed dgamma / dbeta, which every workgroup adds for itself (sums over the batch: float atomics, as in the slab kernels)
    float* rec = (float*)xch;
    unsigned* ticket = (unsigned*)(a.red + ((long)b * a.C + g * gs) * 2); unsigned* done = ticket + 1;
    if (t < 64) {
        const float gm = t < gs ? a.gamma[g * gs + t] : 0.f;
        const float S1w = wave_sum(t < gs ? gm * s_r1[t] : 0.f), S2w = wave_sum(t < gs ? gm * s_r2[t] : 0.f);
        if (t < gs) {
            atomicAdd(&a.dgamma[g * gs + t], s_r2[t]);
            atomicAdd(&a.dbeta[g * gs + t], s_r1[t]);
        }
        if (t == 0) {
            cluster_publish(rec + 2 * w, S1w, S2w);
            __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!cluster_wait(ticket, CL, a.spin_max)) mte_report_device_error(a.err, MTE_DEVERR_GN_CLUSTER_BWD);
            float S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int k = 0; k < CL; ++k) {
                S1 += __hip_atomic_load(rec + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S2 += __hip_atomic_load(rec + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_S[0] = S1; s_S[1] = S2;
            cluster_done(ticket, done, CL);
        }
    }
    __syncthreads();
    const float inv_n = 1.f / ((float)a.HW * gs);
    const float c1 = rstd * rstd * (s_S[1] * inv_n);
    const float c0 = rstd * (s_S[0] * inv_n) - mean * c1;
    float db[P];
#pragma unroll
    for (int k = 0; k < P; ++k) db[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int id = t + i * NT;
        if (id < nchunks) {
            const long pix = (long)b * a.HW + p0 + (id >> a.cps_shift);
            float v[P], gz[P];
            keep_packed<HAS2>(raw[i]);
            asm volatile("" : "+v"(gr[i]));
            finish_v<T, HAS2>(raw[i], sc, v);
            unpack16<T>(gr[i], gz);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const float dyh = gz[k] * elu_grad(fmaf(v[k], ka[k], kb[k]));
                v[k] = dyh * ka[k] - fmaf(v[k], c1, c0);
                if constexpr (HASDB) db[k] += v[k];
            }
            *(u32x4_t*)((T*)a.d1 + pix * a.ldd1 + ch0) = pack16<T>(v);
            if constexpr (SCL) {
                if (a.d2) {
#pragma unroll
                    for (int k = 0; k < P; ++k) v[k] *= sc[k];
                    *(u32x4_t*)((T*)a.d2 + pix * a.ldd2 + ch0) = pack16<T>(v);
                }
            }
        }
    }
    if constexpr (HASDB) {
        block_channel_sums<NT, P>(db, cps, lane, wave, s_part, s_db, gs);
        if (t < gs) atomicAdd(&a.dbias[g * gs + t], SCL && a.scale2 ? s_db[t] * a.scale2[(long)b * a.C + g * gs + t] : s_db[t]);
    }
}

int g_gn_min_rows = 32, g_gn_target = 2048;         // development knobs (mte_debug_set(2 / 3, v))
int g_gn_slab = 1;                                  // development knob (mte_debug_set(13, v)): 0 = stream kernels only

int g_gn_zigzag = 1;                                // development knob (mte_debug_set(14, v))

// Two-pass kernels read the same tensors twice.  Blocks are dispatched in blockIdx order, i.e. sample after sample: when the
// second pass walks the samples in the OPPOSITE order it starts on the bytes the first pass touched last, which are still in
// the 256 MB Infinity Cache (a full-resolution 32-channel activation is 252 MB per tensor at T8: in the same order nothing
// of the first pass survives to the second).
int gn_blocks(int B, int HW, int rstep, bool forward = false) {
    // ~8 workgroups per CU across the batch, at least min_rows pixels per thread row: a thread's rows are consumed in serial
    // batches of GN_U loads, so long runs leave the short low-resolution launches latency-bound; the backward kernels pay a
    // per-workgroup flush of 2-3 C atomics and prefer fewer, longer workgroups
    const int min_rows = forward ? (g_gn_min_rows < 16 ? g_gn_min_rows : 16) : g_gn_min_rows;
    long want = (g_gn_target + B - 1) / B;
    long maxb = ((long)HW + (long)min_rows * rstep - 1) / ((long)min_rows * rstep);
    if (want > maxb) want = maxb;
    return (int)(want < 1 ? 1 : want);
}

bool gn_shape_ok(int C, int dtype) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    if (C % GN_GROUPS != 0 || C % per16 != 0) return false;
    const int cpr = C / per16;
    return cpr <= 256 && 256 % cpr == 0;
}

// slab geometry: a group must be whole 16-byte chunks (1, 2, 4 or 8 per pixel); -> chunks of one slab, or 0 if not a slab shape
long slab_chunks(int HW, int C, int dtype, int* cps_shift) {
    const int per16 = dtype == MTE_DT_BF16 ? 8 : 4;
    const int gs = C / GN_GROUPS;
    if (!g_gn_slab || gs % per16 != 0 || gs > 32) return 0;
    const int cps = gs / per16;
    if (cps != 1 && cps != 2 && cps != 4 && cps != 8) return 0;
    int sh = 0;
    while ((1 << sh) < cps) ++sh;
    *cps_shift = sh;
    return (long)HW * cps;
}
unsigned slab_grid(int B) { return 8u * GN_GROUPS * ((B + 7) / 8); }
// A slab workgroup runs load -> reduce -> apply -> store back to back with the CU to itself (1024 threads), and only B*16
// of them exist: measured (tools/gn_bench.py, B = 8, bf16) it wins 3x on the 12x40 layers (6.8 vs 19 us forward, 19 vs 37 us
// backward), ties at 24x80 and loses at 48x160, where the streaming kernels overlap their phases across workgroups.
constexpr long GN_SLAB_MAX = 1024L * 4;

template <typename T, bool HAS2> bool launch_fwd_slab(const GnArgs& a, long n, hipStream_t st) {
    const dim3 grid(slab_grid(a.B));
#define GN_FWD_SLAB(NCH)                                                                                       \
    if (n <= 1024L * NCH) { hipLaunchKernelGGL((gn_fwd_slab_kernel<T, HAS2, NCH, 1024>), grid, dim3(1024), 0, st, a); return true; }
    GN_FWD_SLAB(2) GN_FWD_SLAB(4)
#undef GN_FWD_SLAB
    return false;
}
template <typename T, int HAS2, bool HASDB> bool launch_bwd_slab(const GnArgs& a, long n, hipStream_t st) {
    const dim3 grid(slab_grid(a.B));
#define GN_BWD_SLAB(NCH, NT)                                                                                   \
    if (n <= (long)NT * NCH) { hipLaunchKernelGGL((gn_bwd_slab_kernel<T, HAS2, HASDB, NCH, NT>), grid, dim3(NT), 0, st, a); return true; }
    GN_BWD_SLAB(2, 1024) GN_BWD_SLAB(4, 1024)
#undef GN_BWD_SLAB
    return false;
}

int g_gn_cluster = 1;                               // development knob (mte_debug_set(25, v)): 0 = no cluster kernels
void gn_common(GnArgs& a) { a.err = g_mte_err_dev; a.fences = g_mte_handoff_fences; a.spin_max = g_gn_spin_max; }

// cluster geometry for a slab of n chunks that is too large for the slab kernels: CL workgroups of 256 threads, NCH chunks per thread.
// regs = 16-byte registers a thread holds per chunk (forward: 1 + second input; backward: 2 + second input).  0 = not a cluster shape.
int cluster_plan(int B, int HW, int C, long n, int sh, int regs, int* nch, int* per_launch) {
    if (!g_gn_cluster || n <= GN_SLAB_MAX || B < 1) return 0;
    const int slots = MTE_GN_SLOTS(B);
    // the registers decide the workgroups per slab: <= 128 VGPRs without spills (measured on the compiler's report) = 12 sixteen-byte data
    // registers per thread, so four workgroups per CU stay resident
    int nmax = 8;
    while (nmax * regs > 12) nmax >>= 1;
    if (nmax < 2) return 0;
    // (measured, tools/gn_bench.py: clusters of 16 with the batch in two launches of 4 samples -- 256 channels at 48x160 backward, 128 at 96x320
    //  forward -- run 1.4-2.3x SLOWER than the streaming kernels: 66 vs 48 us, 89 vs 38 us; a cluster pays only when ONE launch of <= 8
    //  workgroups per slab covers 8 samples)
    int cl = 2;
    while (cl < 8 && ((long)((HW + cl - 1) / cl) << sh) > 256L * nmax) cl <<= 1;
    const long per_wg = (long)((HW + cl - 1) / cl) << sh;  // chunks of the largest pixel range
    if (per_wg > 256L * nmax || 2 + 2 * cl > slots * 2) return 0;
    int n_ = 2;
    while (256L * n_ < per_wg) n_ <<= 1;
    *nch = n_;
    // every LIVE workgroup of a launch must be resident (1024 of 256 threads): samples per launch; the batch goes out in several launches
    const int pl = 8;                                      // (the block map would also deal 4, 2 or 1 samples to the 8 XCD labels)
    if (pl * GN_GROUPS * cl > 1024) return 0;
    if ((B + pl - 1) / pl > 4) return 0;                   // more than four launches: the streaming kernels are the better form
    *per_launch = pl;
    return cl;
}

template <typename T, bool HAS2, int CL> bool launch_fwd_cluster_cl(GnArgs a, int nch, int per_launch, hipStream_t st) {
    const dim3 grid((unsigned)(per_launch * GN_GROUPS * CL));
    a.ppl = per_launch;
#define GN_FWD_CL(NCH) if (nch == NCH) { if constexpr (NCH * (HAS2 ? 2 : 1) <= 12) { \
        for (a.b0 = 0; a.b0 < a.B; a.b0 += per_launch) { a.nb = a.B - a.b0 < per_launch ? a.B - a.b0 : per_launch;          \
            hipLaunchKernelGGL((gn_fwd_cluster_kernel<T, HAS2, NCH, CL>), grid, dim3(256), 0, st, a); }                      \
        return true; } }
    GN_FWD_CL(2) GN_FWD_CL(4) GN_FWD_CL(8)
#undef GN_FWD_CL
    return false;
}
template <typename T, bool HAS2> bool launch_fwd_cluster(const GnArgs& a, int cl, int nch, int per_launch, hipStream_t st) {
    if (cl == 16) return launch_fwd_cluster_cl<T, HAS2, 16>(a, nch, per_launch, st);
    if (cl == 8) return launch_fwd_cluster_cl<T, HAS2, 8>(a, nch, per_launch, st);
    if (cl == 4) return launch_fwd_cluster_cl<T, HAS2, 4>(a, nch, per_launch, st);
    if (cl == 2) return launch_fwd_cluster_cl<T, HAS2, 2>(a, nch, per_launch, st);
    return false;
}
template <typename T, int HAS2, bool HASDB, int CL> bool launch_bwd_cluster_cl(GnArgs a, int nch, int per_launch, hipStream_t st) {
    const dim3 grid((unsigned)(per_launch * GN_GROUPS * CL));
    a.ppl = per_launch;
#define GN_BWD_CL(NCH) if (nch == NCH) { if constexpr (NCH * (HAS2 == 1 ? 3 : 2) <= 12) { \
        for (a.b0 = 0; a.b0 < a.B; a.b0 += per_launch) { a.nb = a.B - a.b0 < per_launch ? a.B - a.b0 : per_launch;          \
            hipLaunchKernelGGL((gn_bwd_cluster_kernel<T, HAS2, HASDB, NCH, CL>), grid, dim3(256), 0, st, a); }               \
        return true; } }
    GN_BWD_CL(2) GN_BWD_CL(4)
#undef GN_BWD_CL
    return false;
}
template <typename T, int HAS2, bool HASDB> bool launch_bwd_cluster(const GnArgs& a, int cl, int nch, int per_launch, hipStream_t st) {
    if (cl == 16) return launch_bwd_cluster_cl<T, HAS2, HASDB, 16>(a, nch, per_launch, st);
    if (cl == 8) return launch_bwd_cluster_cl<T, HAS2, HASDB, 8>(a, nch, per_launch, st);
    if (cl == 4) return launch_bwd_cluster_cl<T, HAS2, HASDB, 4>(a, nch, per_launch, st);
    if (cl == 2) return launch_bwd_cluster_cl<T, HAS2, HASDB, 2>(a, nch, per_launch, st);
    return false;
}

// statistics from per-tile records (round 5: the LDS-patch forward kernels leave one record of 32 floats per output tile, conv_patch.hip: patch_gn_record):
// stats[b][group][2] (fp64) = the records of sample b added in tile order.  One workgroup per sample; 32 threads per value take tiles k, k + 32, ... in order,
// then the 32 partial sums are added in order (the scheme of the statistics pass's last arriver).
__global__ __launch_bounds__(1024) void gn_stats_from_records_kernel(const float* __restrict__ rec, int nrec, double* __restrict__ stats) {
    __shared__ double s_fin[32 * 32];
    const int b = blockIdx.x, v = threadIdx.x & 31, k = threadIdx.x >> 5;
    const float* r = rec + (long)b * nrec * 32 + v;
    double acc = 0.0;
    int j = k;
    for (; j + 3 * 32 < nrec; j += 4 * 32) {
        const float t0 = r[(long)j * 32], t1 = r[(long)(j + 32) * 32], t2 = r[(long)(j + 64) * 32], t3 = r[(long)(j + 96) * 32];
        acc += (double)t0; acc += (double)t1; acc += (double)t2; acc += (double)t3;
    }
    for (; j < nrec; j += 32) acc += (double)r[(long)j * 32];
    s_fin[k * 32 + v] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += s_fin[i * 32 + threadIdx.x];
        stats[(long)b * 32 + threadIdx.x] = t;
    }
}

template <typename T> int run_stats(GnArgs& a, hipStream_t stream) {
    constexpr int NT = 1024;
    const int rstep = NT / (a.C / Elem<T>::PER16);
    // the same number of threads as 2048 workgroups of 256 would be; one record slot per block
    long want = ((long)g_gn_target * 256 / NT + a.B - 1) / a.B;
    const int min_rows = g_gn_min_rows < 16 ? g_gn_min_rows : 16;
    const long maxb = ((long)a.HW + (long)min_rows * rstep - 1) / ((long)min_rows * rstep);
    if (want > maxb) want = maxb;
    if (want > MTE_GN_SLOTS(a.B)) want = MTE_GN_SLOTS(a.B);
    a.blocks_per_sample = (int)(want < 1 ? 1 : want);
    a.reverse = g_gn_zigzag;                               // the producing conv wrote sample 0 first: start on the freshest bytes
    dim3 grid(a.blocks_per_sample, a.B);
    if (a.y2) hipLaunchKernelGGL((gn_stats_kernel<T, true, NT>), grid, dim3(NT), 0, stream, a);
    else hipLaunchKernelGGL((gn_stats_kernel<T, false, NT>), grid, dim3(NT), 0, stream, a);
    return mte_check_launch();
}

template <typename T> int run_fwd(GnArgs& a, hipStream_t stream, int reverse = 0) {
    a.blocks_per_sample = gn_blocks(a.B, a.HW, 256 / (a.C / Elem<T>::PER16), true);
    a.reverse = reverse;                                   // ... and the statistics pass ended on sample 0
    dim3 grid(a.blocks_per_sample, a.B);
    if (a.y2) hipLaunchKernelGGL((gn_elu_fwd_kernel<T, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gn_elu_fwd_kernel<T, false>), grid, dim3(256), 0, stream, a);
    return mte_check_launch();
}

template <typename T> int run_bwd(GnArgs& a, int dtype, hipStream_t stream) {
    int sh = 0;
    const long n = slab_chunks(a.HW, a.C, dtype, &sh);
    // second-output mode (see gn_elu_bwd_apply_kernel): 1 = second input tensor, 2 = one input whose gradient also leaves scaled (d2 = scale2 * d1)
    const int m2 = a.y2 ? 1 : (a.scale2 && a.d2 ? 2 : 0);
    if (m2 != 1) a.y2 = nullptr;
    if (m2 == 0) { a.scale2 = nullptr; a.d2 = nullptr; }
    if (n > 0 && n <= GN_SLAB_MAX && m2 != 2) {
        a.cps_shift = sh;
        // the slab kernels ADD this sample's part of dgamma / dbeta (the stream kernels overwrite them): clear them first (prezeroed callers did)
        if (!g_mte_gn_prezeroed && (mte_memset_async(a.dgamma, 0, sizeof(float) * a.C, stream) != hipSuccess ||
                                    mte_memset_async(a.dbeta, 0, sizeof(float) * a.C, stream) != hipSuccess)) return MTE_ERR_LAUNCH;
        bool done;
        if (a.y2) done = a.dbias ? launch_bwd_slab<T, 1, true>(a, n, stream) : launch_bwd_slab<T, 1, false>(a, n, stream);
        else if (a.dbias) done = launch_bwd_slab<T, 0, true>(a, n, stream);
        else done = launch_bwd_slab<T, 0, false>(a, n, stream);
        if (done) return mte_check_launch();
    }
    if (n > GN_SLAB_MAX) {                  // (round 5: also the tail with a bias gradient -- the kernels scale the column sums now)
        int nch = 0, pl = 0;
        const int cl = cluster_plan(a.B, a.HW, a.C, n, sh, a.y2 ? 3 : 2, &nch, &pl);
        if (cl) {
            a.cps_shift = sh;
            if (!g_mte_gn_prezeroed && (mte_memset_async(a.dgamma, 0, sizeof(float) * a.C, stream) != hipSuccess ||
                                        mte_memset_async(a.dbeta, 0, sizeof(float) * a.C, stream) != hipSuccess)) return MTE_ERR_LAUNCH;
            bool done;
            if (a.y2) done = a.dbias ? launch_bwd_cluster<T, 1, true>(a, cl, nch, pl, stream) : launch_bwd_cluster<T, 1, false>(a, cl, nch, pl, stream);
            else if (m2 == 2) done = a.dbias ? launch_bwd_cluster<T, 2, true>(a, cl, nch, pl, stream) : launch_bwd_cluster<T, 2, false>(a, cl, nch, pl, stream);
            else if (a.dbias) done = launch_bwd_cluster<T, 0, true>(a, cl, nch, pl, stream);
            else done = launch_bwd_cluster<T, 0, false>(a, cl, nch, pl, stream);
            if (done) return mte_check_launch();
        }
    }
    a.blocks_per_sample = gn_blocks(a.B, a.HW, 256 / (a.C / Elem<T>::PER16));
    dim3 grid(a.blocks_per_sample, a.B);
    const size_t lds = sizeof(float) * a.C * 2;
    GnArgs a2 = a;
    a2.reverse = g_gn_zigzag;                              // the apply pass starts where the reduce pass ended
    const size_t ldb = a.dbias ? sizeof(float) * a.C : 0;
    if (a.y2) {
        hipLaunchKernelGGL((gn_elu_bwd_reduce_kernel<T, true>), grid, dim3(256), lds, stream, a);
        if (a.dbias) hipLaunchKernelGGL((gn_elu_bwd_apply_kernel<T, 1, true>), grid, dim3(256), ldb, stream, a2);
        else hipLaunchKernelGGL((gn_elu_bwd_apply_kernel<T, 1, false>), grid, dim3(256), 0, stream, a2);
    } else {
        hipLaunchKernelGGL((gn_elu_bwd_reduce_kernel<T, false>), grid, dim3(256), lds, stream, a);
        if (m2 == 2) {
            if (a.dbias) hipLaunchKernelGGL((gn_elu_bwd_apply_kernel<T, 2, true>), grid, dim3(256), ldb, stream, a2);
            else hipLaunchKernelGGL((gn_elu_bwd_apply_kernel<T, 2, false>), grid, dim3(256), 0, stream, a2);
        } else if (a.dbias) hipLaunchKernelGGL((gn_elu_bwd_apply_kernel<T, 0, true>), grid, dim3(256), ldb, stream, a2);
        else hipLaunchKernelGGL((gn_elu_bwd_apply_kernel<T, 0, false>), grid, dim3(256), 0, stream, a2);
    }
    return mte_check_launch();
}

// the residual tail's first kernel (gn_stats_kernel<.., TAIL>): the geometry of the statistics pass
template <typename T> int run_tail(GnArgs& a, hipStream_t stream) {
    constexpr int NT = 1024;
    const int rstep = NT / (a.C / Elem<T>::PER16);
    long want = ((long)g_gn_target * 256 / NT + a.B - 1) / a.B;
    const int min_rows = g_gn_min_rows < 16 ? g_gn_min_rows : 16;
    const long maxb = ((long)a.HW + (long)min_rows * rstep - 1) / ((long)min_rows * rstep);
    if (want > maxb) want = maxb;
    if (want > MTE_GN_SLOTS(a.B)) want = MTE_GN_SLOTS(a.B);
    a.blocks_per_sample = (int)(want < 1 ? 1 : want);
    a.reverse = 0;                                         // the inner layer's statistics pass ended on sample 0
    dim3 grid(a.blocks_per_sample, a.B);
    hipLaunchKernelGGL((gn_stats_kernel<T, true, NT, true>), grid, dim3(NT), 0, stream, a);
    return mte_check_launch();
}

}  // namespace

#ifdef MTE_DEV
extern "C" int mtei_set_gn(int which, int value) {
    if (which == 2) { g_gn_slab = value; return MTE_OK; }
    if (which == 3) { g_gn_zigzag = value; return MTE_OK; }
    if (which == 4) { if (value >= 1000) g_gn_spin_max = (unsigned)(value - 1000); else g_gn_cluster = value; return MTE_OK; }
    if (value < 1) return MTE_ERR_ARG;
    if (which == 0) g_gn_min_rows = value; else g_gn_target = value;
    return MTE_OK;
}
#endif

extern "C" {

// 1 if mte_gn_elu_fwd computes the statistics of this shape itself (single-pass slab kernel): the caller then skips mte_gn_stats.
int mte_gn_fwd_is_single_pass(int HW, int C, int has_y2, int dtype) {
    int sh = 0;
    if (!gn_shape_ok(C, dtype)) return 0;
    const long n = slab_chunks(HW, C, dtype, &sh);
    return (n > 0 && n <= GN_SLAB_MAX) ? 1 : 0;
}

// The same question for a batch of B samples: also 1 where a cluster of workgroups holds the slab (norm_act.hip, CLUSTER kernels; the
// cluster's size depends on the batch).  Callers that know B ask this one.
int mte_gn_fwd_is_single_pass_b(int B, int HW, int C, int has_y2, int dtype) {
    int sh = 0, nch = 0, pl = 0;
    if (!gn_shape_ok(C, dtype)) return 0;
    const long n = slab_chunks(HW, C, dtype, &sh);
    if (n <= 0) return 0;
    if (n <= GN_SLAB_MAX) return 1;
    return cluster_plan(B, HW, C, n, sh, has_y2 ? 2 : 1, &nch, &pl) ? 1 : 0;
}

// doubles of a statistics buffer for batch B (final sums + arrival tickets + per-block records, see common.hpp)
long mte_gn_stats_elems(int B) { return B < 1 ? 0 : mte_gn_stats_elems_(B); }

// stats[0 .. B*32) <- per-(sample, group) sum and sum of squares of v = y1 + scale2*y2 (doubles; the rest of the buffer is the
// pass's workspace: tickets -- zeroed here unless MTE_OPT_GN_PREZEROED -- and records).  Bit-reproducible.
int mte_gn_stats(const void* y1, long ld1, const void* y2, long ld2, const float* scale2, double* stats,
                 int B, int HW, int C, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!y1 || !stats || !gn_shape_ok(C, dtype)) return MTE_ERR_ARG;
    if (!g_mte_gn_prezeroed && mte_memset_async(mte_gn_tickets(stats, B), 0, sizeof(double) * ((B + 15) & ~15), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    GnArgs a{}; a.y1 = y1; a.ld1 = ld1; a.y2 = y2; a.ld2 = ld2; a.scale2 = scale2; a.stats = stats; a.B = B; a.HW = HW; a.C = C;
    gn_common(a);
    return dtype == MTE_DT_BF16 ? run_stats<bf16_t>(a, stream) : run_stats<float>(a, stream);
}

int mte_gn_stats_from_records(const float* rec, int tiles_per_sample, double* stats, int B, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!rec || !stats || tiles_per_sample < 1 || B < 1) return MTE_ERR_ARG;
    hipLaunchKernelGGL(gn_stats_from_records_kernel, dim3((unsigned)B), dim3(1024), 0, stream, rec, tiles_per_sample, stats);
    return mte_check_launch();
}

// z = ELU(GN(y1 + scale2*y2)).  stats: the sums mte_gn_stats (or a conv epilogue) accumulated -- or, where
// mte_gn_fwd_is_single_pass(HW, C, y2 != null, dtype) is 1 and stats_ready == 0, an OUTPUT: the kernel computes the
// statistics of the slab it holds on chip and stores them in the same format for the backward pass.
int mte_gn_elu_fwd(const void* y1, long ld1, const void* y2, long ld2, const float* scale2, double* stats, int stats_ready,
                   const float* gamma, const float* beta, void* z, long ldz,
                   int B, int HW, int C, float eps, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!y1 || !stats || !gamma || !beta || !z || !gn_shape_ok(C, dtype)) return MTE_ERR_ARG;
    GnArgs a{}; a.y1 = y1; a.ld1 = ld1; a.y2 = y2; a.ld2 = ld2; a.scale2 = scale2; a.stats = stats;
    a.gamma = gamma; a.beta = beta; a.z = z; a.ldz = ldz; a.B = B; a.HW = HW; a.C = C; a.eps = eps;
    gn_common(a);
    if (!stats_ready) {
        if (!mte_gn_fwd_is_single_pass_b(B, HW, C, y2 != nullptr, dtype)) return MTE_ERR_ARG;
        int sh = 0;
        const long n = slab_chunks(HW, C, dtype, &sh);
        a.cps_shift = sh;
        bool ok;
        if (n > GN_SLAB_MAX) {                             // a cluster of workgroups per (sample, group)
            int nch = 0, pl = 0;
            const int cl = cluster_plan(B, HW, C, n, sh, y2 ? 2 : 1, &nch, &pl);
            // the cluster's exchange words live in the record area of the statistics buffer and must be zero at entry
            if (!g_mte_gn_prezeroed &&
                mte_memset_async(mte_gn_partials(stats, B), 0, sizeof(double) * (size_t)B * MTE_GN_SLOTS(B) * 32, stream) != hipSuccess) return MTE_ERR_LAUNCH;
            if (dtype == MTE_DT_BF16) ok = y2 ? launch_fwd_cluster<bf16_t, true>(a, cl, nch, pl, stream) : launch_fwd_cluster<bf16_t, false>(a, cl, nch, pl, stream);
            else ok = y2 ? launch_fwd_cluster<float, true>(a, cl, nch, pl, stream) : launch_fwd_cluster<float, false>(a, cl, nch, pl, stream);
            return ok ? mte_check_launch() : MTE_ERR_ARG;
        }
        if (dtype == MTE_DT_BF16) ok = y2 ? launch_fwd_slab<bf16_t, true>(a, n, stream) : launch_fwd_slab<bf16_t, false>(a, n, stream);
        else ok = y2 ? launch_fwd_slab<float, true>(a, n, stream) : launch_fwd_slab<float, false>(a, n, stream);
        return ok ? mte_check_launch() : MTE_ERR_ARG;
    }
    return dtype == MTE_DT_BF16 ? run_fwd<bf16_t>(a, stream) : run_fwd<float>(a, stream);
}

// Backward of z = ELU(GN(y1 + scale2*y2)).  red[B][C][2] is scratch (zeroed here).  Writes d1 (grad of y1),
// optionally d2 (grad of y2 = scale2 * d1), dgamma/dbeta [C] (overwritten) and, if non-null, dbias [C] += column sums of d1 -- or, when y2 is
// given, of d2 (the bias gradient of the conv that produced y2: the residual block's 1x1 shortcut).
int mte_gn_elu_bwd(const void* dz, long lddz, const void* y1, long ld1, const void* y2, long ld2, const float* scale2,
                   const double* stats, const float* gamma, const float* beta, float* red,
                   void* d1, long ldd1, void* d2, long ldd2, float* dgamma, float* dbeta, float* dbias,
                   int B, int HW, int C, float eps, int dtype, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!dz || !y1 || !stats || !gamma || !beta || !red || !d1 || !dgamma || !dbeta || !gn_shape_ok(C, dtype)) return MTE_ERR_ARG;
    if (d2 && !y2 && !scale2) return MTE_ERR_ARG;      // a second output needs its factor (with one input d2 = scale2 * d1): never MTE_OK with d2 left unwritten
    if (!g_mte_gn_prezeroed) {
        if (mte_memset_async(red, 0, sizeof(float) * (size_t)B * C * 2, stream) != hipSuccess) return MTE_ERR_LAUNCH;
        if (dbias && mte_memset_async(dbias, 0, sizeof(float) * C, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    }
    GnArgs a{}; a.dbias = dbias; a.dgamma = dgamma; a.dbeta = dbeta; a.y1 = y1; a.ld1 = ld1; a.y2 = y2; a.ld2 = ld2; a.scale2 = scale2; a.stats = (double*)stats;
    a.gamma = gamma; a.beta = beta; a.dz = dz; a.lddz = lddz; a.red = red; a.d1 = d1; a.ldd1 = ldd1; a.d2 = d2; a.ldd2 = ldd2;
    a.B = B; a.HW = HW; a.C = C; a.eps = eps;
    gn_common(a);
    return dtype == MTE_DT_BF16 ? run_bwd<bf16_t>(a, dtype, stream) : run_bwd<float>(a, dtype, stream);
}

// The residual block's tail, ELU(GN_t(ELU(GN_1(y1)) + scale2 * y2)) (ResidualConv.forward, layers01.py:62-73: y1 = conv2's convolution output,
// GN_1 / ELU = the rest of that Conv2D, y2 = the 1x1 shortcut, scale2 = Dropout2d's per-(sample, channel) factor or null), in TWO launches:
//   1. t = ELU(GN_1(y1)) + scale2 * y2, stored in the activation type, and the statistics of the stored t  (stats1: the sums of y1 from mte_gn_stats)
//   2. z = ELU(GN_t(t))
// against four before (inner apply, statistics over two tensors, outer apply over two tensors; + the inner statistics either way): 6 tensor passes
// instead of 8, and the backward pass of the outer norm reads ONE saved tensor (t) -- mte_gn_elu_bwd with y2 = null, scale2 and d2 given.
// stats_t: a statistics buffer (mte_gn_stats_elems(B) doubles, tickets zero at entry); bit-reproducible like mte_gn_stats.
int mte_gn_tail_fwd(const void* y1, long ld1, const double* stats1, const float* gamma1, const float* beta1,
                    const void* y2, long ld2, const float* scale2, void* t, long ldt, double* stats_t,
                    const float* gamma_t, const float* beta_t, void* z, long ldz,
                    int B, int HW, int C, float eps, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (!y1 || !stats1 || !gamma1 || !beta1 || !y2 || !t || !stats_t || !gamma_t || !beta_t || !z || !gn_shape_ok(C, dtype)) return MTE_ERR_ARG;
    if (!g_mte_gn_prezeroed && mte_memset_async(mte_gn_tickets(stats_t, B), 0, sizeof(double) * ((B + 15) & ~15), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    GnArgs a{}; a.y1 = y1; a.ld1 = ld1; a.y2 = y2; a.ld2 = ld2; a.scale2 = scale2; a.stats_in = stats1; a.stats = stats_t;
    a.gamma = gamma1; a.beta = beta1; a.z = t; a.ldz = ldt; a.B = B; a.HW = HW; a.C = C; a.eps = eps;
    gn_common(a);
    int rc = dtype == MTE_DT_BF16 ? run_tail<bf16_t>(a, stream) : run_tail<float>(a, stream);
    if (rc != MTE_OK) return rc;
    GnArgs f{}; f.y1 = t; f.ld1 = ldt; f.stats = stats_t; f.gamma = gamma_t; f.beta = beta_t; f.z = z; f.ldz = ldz; f.B = B; f.HW = HW; f.C = C; f.eps = eps;
    gn_common(f);
    rc = dtype == MTE_DT_BF16 ? run_fwd<bf16_t>(f, stream, g_gn_zigzag) : run_fwd<float>(f, stream, g_gn_zigzag);   // (the tail kernel ended on the last sample)
    return rc;
}

int mte_set_option(int option, int value) {
    if (option == 0) { g_mte_gn_prezeroed = value ? 1 : 0; return MTE_OK; }      // MTE_OPT_GN_PREZEROED
    if (option == 1) { g_mte_loss_prezeroed = value ? 1 : 0; return MTE_OK; }    // MTE_OPT_LOSS_PREZEROED
    if (option == 2) { g_mte_handoff_fences = value ? 1 : 0; return MTE_OK; }    // MTE_OPT_HANDOFF_FENCES
    if (option == 3) { g_mte_wgrad_shared = value ? 1 : 0; return MTE_OK; }      // MTE_OPT_WGRAD_SHARES_CHIP
    return MTE_ERR_ARG;
}

// Device error word (common.hpp).  init: one 32-bit word of pinned, host-coherent memory (call once, outside any stream capture; a second call
// is a no-op).  poll: -> the flags reported since the last poll (0 = none) and clears them; does not synchronise -- a kernel still running may
// report later, so a step's error can surface at the NEXT poll.  Without init the kernels' bounded waits still give up, silently.
int mte_device_error_init(void) {
    if (g_mte_err_host) return MTE_OK;
    unsigned* h = nullptr; unsigned* d = nullptr;
    if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) != hipSuccess || !h) { (void)hipGetLastError(); return MTE_ERR_LAUNCH; }
    *h = 0u;
    if (hipHostGetDevicePointer((void**)&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); return MTE_ERR_LAUNCH; }
    g_mte_err_host = h; g_mte_err_dev = d;
    return MTE_OK;
}
int mte_device_error_poll(void) {
    if (!g_mte_err_host) return 0;
    const unsigned v = *g_mte_err_host;
    if (v) *g_mte_err_host = 0u;
    return (int)v;
}

}  // extern "C"
