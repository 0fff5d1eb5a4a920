// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of mindtheedge_amd.
// Wavefront = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MTE_OK 0
#define MTE_ERR_ARG (-1)
#define MTE_ERR_LAUNCH (-2)
#define MTE_ERR_UNSUPPORTED (-3)

#define MTE_DT_BF16 0
#define MTE_DT_F32 1

typedef unsigned short bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
    static constexpr int PER16 = 8;                      // elements per 16-byte chunk
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <> struct Elem<float> {
    static constexpr int PER16 = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};

// 16-byte chunk <-> 8 floats (bf16) / 4 floats (f32).  Always fills/consumes v[0..PER16).
template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& c, float* v);
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4_t& c, float* v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(c[i] << 16);
        v[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& c, float* v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(c[i]);
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* v);
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* v) {
    u32x4_t c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack2bf(v[2 * i], v[2 * i + 1]);
    return c;
}
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* v) {
    u32x4_t c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(v[i]);
    return c;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float elu1(float u) { return u > 0.f ? u : (__expf(u) - 1.f); }

// XCD-aware bijective remap of a 1-D grid: blocks that share an XCD (bid % 8) get a contiguous
// range of logical tile ids, so neighbouring tiles (shared halos / weight panels) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// In-loop clock of the MFMA loops (diagnostic build only: -DMTE_CLOCK, tools/inloop_clock.py; no stamp executes in the product or the development library).
// MI355X_MICROARCH.md 'DVFS give-back' item 6: shader clock = delta s_memtime / delta s_memrealtime x 100 MHz, stamped ONCE around the main loop; thread 0 of a
// workgroup stores the two deltas into a buffer of the translation unit that no kernel reads.
#ifdef MTE_CLOCK
#define MTE_CLOCK_DEFINE(TAG)                                                                                                          \
    static __device__ unsigned long long g_clk_##TAG[16384 * 2];                                                                       \
    extern "C" int mtei_clk_##TAG(unsigned long long* host, int n) {                                                                   \
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clk_##TAG), sizeof(unsigned long long) * n) != hipSuccess) return -1;               \
        void* p_ = nullptr;                                          /* read and clear: a later launch with fewer workgroups leaves no stale entries */ \
        if (hipGetSymbolAddress(&p_, HIP_SYMBOL(g_clk_##TAG)) != hipSuccess) return -1;                                                \
        return hipMemset(p_, 0, sizeof(g_clk_##TAG)) == hipSuccess ? 0 : -1;                                                           \
    }
#define MTE_CLOCK_BEGIN()                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                                                 \
    const unsigned long long clk_c0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime();                      \
    __builtin_amdgcn_sched_barrier(0);
#define MTE_CLOCK_END(TAG)                                                                                                             \
    {                                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();                          \
        __builtin_amdgcn_sched_barrier(0);                                                                                             \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 16384) {                                            \
            g_clk_##TAG[2 * blockIdx.x] = c1_ - clk_c0_;                                                                               \
            g_clk_##TAG[2 * blockIdx.x + 1] = r1_ - clk_r0_;                                                                           \
        }                                                                                                                              \
    }
#else
#define MTE_CLOCK_DEFINE(TAG)
#define MTE_CLOCK_BEGIN()
#define MTE_CLOCK_END(TAG)
#endif

// mte_set_option(MTE_OPT_GN_PREZEROED, 1): the caller hands over GroupNorm statistics / reduction / bias-gradient buffers
// that are already zero (carved from one arena it clears with a single memset), so the library skips its ~140 tiny
// per-layer hipMemsetAsync launches per training step.  Defined in norm_act.hip.
extern int g_mte_gn_prezeroed;
// mte_set_option(MTE_OPT_LOSS_PREZEROED, 1): the workspaces handed to mte_edge_loss_multi_fwd / mte_edge_loss_fwd are zero on entry (same arena):
// the launch's own fill kernel is skipped.  Defined in edge_loss.hip.
extern int g_mte_loss_prezeroed;

// Device error word (mte_device_error_init / mte_device_error_poll, norm_act.hip): one 32-bit word of host-coherent pinned memory that every
// kernel with a bounded inter-workgroup wait receives as an argument.  A wait that gives up ORs its code into the word (system scope) instead
// of publishing numbers computed from records that never arrived; the host polls the word once per step and raises.  Null = not initialised
// (the kernels then only give up).  Codes are bit flags so that several kernels of a step can report.
#define MTE_DEVERR_GN_CLUSTER_FWD 1u
#define MTE_DEVERR_GN_CLUSTER_BWD 2u
extern unsigned* g_mte_err_dev;
__device__ __forceinline__ void mte_report_device_error(unsigned* err, unsigned code) {
    // (load | store rather than an atomic OR: the word lives in host memory and a lost flag of a second, simultaneous reporter does not matter --
    //  any non-zero value fails the step)
    if (err) __hip_atomic_store(err, __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// mte_set_option(MTE_OPT_HANDOFF_FENCES, v): the last-arriver hand-offs of the read-only kernels (GroupNorm statistics, loss sums) draw their
// ticket with an agent-scope RELEASE and the last arriver issues an agent-scope ACQUIRE before it reads the records.  The records themselves
// are returning atomic exchanges / agent-scope atomic loads (performed at the memory side, never served from a CU's L1), which is the form
// MI355X_MICROARCH.md's visibility table lists as measured-valid; the fences make the protocol independent of that table.  Defined in norm_act.hip.
extern int g_mte_handoff_fences;
// mte_set_option(MTE_OPT_WGRAD_SHARES_CHIP, v): 1 = the caller queues the weight-gradient launches on a stream of their own beside the data-gradient chain (the
// host side of this repository does): the MFMA weight-gradient kernels then aim for HALF a chip of workgroups (profiles/r05_side_queue_width.txt: same-box step
// 23.59 -> 23.1 ms); 0 (default) = nothing runs beside them: one workgroup (group) per CU.  Defined in norm_act.hip.
extern int g_mte_wgrad_shared;

// GroupNorm statistics buffer of a batch of B samples, in doubles (mte_gn_stats_elems(B)):
//   [0, 32 B)                       final (sum, sum of squares) per (sample, group): what every consumer reads
//   [32 B, 32 B + round16(B))       arrival tickets of the statistics pass (one 32-bit counter per sample, 8 bytes apart; must be 0)
//   then [B][MTE_GN_SLOTS(B)][32]   one record per statistics workgroup (written, never accumulated: needs no clearing)
// The statistics pass is bit-reproducible: no floating-point atomics anywhere on the forward path (norm_act.hip).
#define MTE_GN_SLOTS(B) ((B) >= 8 ? 64 : 512 / (B))
static inline long mte_gn_stats_elems_(int B) { return (long)B * 32 + ((B + 15) & ~15) + (long)B * MTE_GN_SLOTS(B) * 32; }
__host__ __device__ static inline unsigned* mte_gn_tickets(double* stats, int B) { return (unsigned*)(stats + (long)B * 32); }
__host__ __device__ static inline double* mte_gn_partials(double* stats, int B) { return stats + (long)B * 32 + ((B + 15) & ~15); }

// Zero / byte-pattern fill as a KERNEL.  hipMemsetAsync must not be used on this library's launch paths: captured into a HIP
// graph (utils/graph.py) its memset node was seen NOT to take effect on replays issued after the device had gone idle
// (ROCm 7.2, MI355X: split-K workspaces kept the previous frame's sums, GroupNorm variances went negative, NaN downstream),
// while fill kernels replay correctly.  16-byte stores when the range allows it, words or bytes otherwise.
static __global__ void mte_fill16_kernel(u32x4_t* __restrict__ p, unsigned v, size_t n16, unsigned* __restrict__ tail, int ntail) {
    const u32x4_t vv = {v, v, v, v};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = vv;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = v;
}
static __global__ void mte_fill1_kernel(unsigned char* __restrict__ p, unsigned char v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static inline hipError_t mte_memset_async(void* p, int value, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    const unsigned b = (unsigned)value & 0xffu, v = b | (b << 8) | (b << 16) | (b << 24);
    if (((uintptr_t)p & 15) == 0 && bytes % 4 == 0) {
        const size_t n16 = bytes / 16;
        size_t g = (n16 + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1;
        hipLaunchKernelGGL(mte_fill16_kernel, dim3((unsigned)g), dim3(256), 0, st, (u32x4_t*)p, v, n16, (unsigned*)p + n16 * 4, (int)((bytes % 16) / 4));
    } else {
        size_t g = (bytes + 255) / 256; if (g > 2048) g = 2048;
        hipLaunchKernelGGL(mte_fill1_kernel, dim3((unsigned)g), dim3(256), 0, st, (unsigned char*)p, (unsigned char)b, bytes);
    }
    return hipGetLastError();
}

static inline int mte_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "[libmte_hip] launch failed: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
    return e == hipSuccess ? MTE_OK : MTE_ERR_LAUNCH;
}
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
