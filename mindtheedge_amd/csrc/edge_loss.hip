// Fused depth-edge loss and silog loss for gfx950 (fp32 NCHW maps with C = 1, i.e. plain [B,H,W]).
//
// Edge loss = GradLoss.forward with edge_loss_type 'cross_entropy' (packnet_sfm/losses/grad_loss.py:122-219)
// fused with inv2depth (utils/depth.py:104-121) and GradLayer (grad_loss.py:20-31,65-95):
//   depth = 1/max(inv,1e-6) -> 4 Sobel responses (v,h,lr,rl; zero pad) -> per-pixel direction chosen by the
//   edge-normal angle (or sqrt(v^2+h^2+1e-6) without normals) -> p = sigmoid(g - thresh)
//   -> pos = -e log(p+1e-3), neg = -(1-e) log(1-p+1e-3) -> per-sample class-balance alpha -> weighted mean.
// Silog = SupervisedLoss 'sparse-silog', one scale (losses/supervised_loss.py:57-69,155-216).
//
// The reference issues 4 conv2d + ~20 masking kernels + torch.unique (host syncs) per scale.  Here ONE forward launch
// covers all four scales (+ the silog sums, which read the same full-resolution inverse depth) and ONE backward launch
// writes all four gradients (+ the silog gradient):
//   * workgroup = one 64 x 32-pixel tile of one (scale, sample); the tile + halo of the prediction is staged in LDS as
//     DEPTH (the reciprocal is taken once per pixel, halo overhead 10 % / 20 %); a thread owns 4 consecutive pixels of a
//     row, so every global access is a 16-byte load / store (1 KiB per wave instruction);
//   * every global load of a workgroup (tile + halo, labels, normals) is issued before its first use, so a workgroup pays
//     one memory latency, not one per phase;
//   * sums: fp32 per thread and per wave, fp64 per workgroup; a workgroup leaves ONE 128-byte record, the last workgroup of a
//     (scale, sample) to arrive (agent-scope ticket) adds that image's <= 240 records in a fixed order, and the LAST of those computes
//     alpha, the loss scalars and the backward coefficients on the device from an LDS copy of the 32 images' sums (one thread per
//     scale): no reduce / finalize launches, no host sync, no floating-point atomics -- losses and coefficients are bit-reproducible
//     (round 4; rounds 1-3 added each value with one fp64 atomic per workgroup: 1,680 adds on one cache line per full-resolution
//     image cost half of the launch, and the single-thread scalar tail another 15 us: forward 73 -> see profiles/README.md).
// HBM-bound: 12 B/pixel forward (inv, edge, normal), 16 B/pixel backward (+4 B gradient write).
#include "common.hpp"
#include "edge_direction.hpp"

int g_mte_loss_prezeroed = 0;

namespace {

constexpr int TW = 64, TH = 32;         // output tile (256 threads x 2 passes x 4 pixels)
constexpr int LS = 72;                  // LDS row stride in floats: image column j of the tile sits at index 4 + j (16-byte aligned interior)
constexpr int MAXS = 4;                 // scales per launch
constexpr int NP = 13;                  // partial sums per workgroup: 6 edge sums, 4 mask statistics, 3 silog sums
constexpr int REC = 16;                 // doubles per workgroup record (one 128-byte line)
#ifndef MTE_EDGE_FWD_TILES
#define MTE_EDGE_FWD_TILES 3
#endif
constexpr int FWD_TILES_PER_WG = MTE_EDGE_FWD_TILES;     // forward: consecutive tiles per workgroup (3: 856 workgroups at T8, one round of the 1024 slots)

struct EdgeScale {
    const float* pred;                  // inv-depth (from_inv), depth, or probability map
    const float* edge; const float* normal; const float* mask;    // normal / mask nullable
    float* gmap;                        // forward: optional edge-strength map output
    float* dpred;                       // backward output
    int H, W, tiles_x, tiles_y, first_block, vec;                // vec: 16-byte accesses are legal (W % 4 == 0, aligned bases)
    int groups;                         // workgroups per image: each takes `tiles_per_wg` consecutive tiles (row-major)
};

struct EdgeMulti {
    EdgeScale s[MAXS];
    int nscales, B, nblocks, tiles_per_wg;
    int from_inv, is_grad, is_sigmoid, finalize;
    float thresh, weight, pos_to_neg;
    double* results;                    // [nscales][B][NP] sums of each (scale, sample) (zeroed by the launcher; written by the image's last workgroup)
    double* records;                    // [nblocks][REC]: one record of partial sums per workgroup
    unsigned* counter;                  // [0] launch ticket, [1 + s*B + b] ticket of (scale, sample) (zeroed by the launcher)
    float* losses;                      // forward out: [nscales]
    float* coef;                        // forward out / backward in: [nscales][2B + 1]
    const float* gout;                  // backward: upstream gradient per scale loss (device, nullable = 1)
    const float* gt_depth;              // optional fused silog on scale 0: metric depth, 0 = invalid
    float* silog_loss; float* silog_aux;          // forward out: loss, (mean, 10/sqrt(S)/n)
    const float* silog_gout;            // backward: upstream gradient of the silog loss (device, nullable = 1)
    int fences;                         // MTE_OPT_HANDOFF_FENCES (common.hpp): release before each ticket, acquire in the last arriver
};

// 1 / x as v_rcp_f32 (1 ulp) + one Newton step: r' = r + r (1 - x r), the two fmas of the IEEE division sequence without its scaling and
// fix-up instructions (3 instructions instead of ~10).  x is in [1e-6, ~1e3] here -- no denormals, overflow or division by zero to fix up -- and
// the result is within 1 ulp of the correctly rounded quotient (almost always equal to it).
__device__ __forceinline__ float rcp_newton(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.f), r);
}
#if defined(MTE_EDGE_ABLATE) && (MTE_EDGE_ABLATE & 4)
__device__ __forceinline__ float to_depth(int from_inv, float v) { return from_inv ? 1.f / fmaxf(v, 1e-6f) : v; }   // diagnostic: the IEEE division of rounds 1-3
#else
__device__ __forceinline__ float to_depth(int from_inv, float v) { return from_inv ? rcp_newton(fmaxf(v, 1e-6f)) : v; }
#endif

// 4 consecutive floats of row `row` starting at column x (x % 4 == 0); zero beyond the image
__device__ __forceinline__ f32x4_t load4(const float* base, long row, int x, int W, int vec) {
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (vec) { if (x < W) v = *(const f32x4_t*)(base + row * W + x); }
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (x + k < W) v[k] = base[row * W + x + k];
    }
    return v;
}
__device__ __forceinline__ void store4(float* base, long row, int x, int W, int vec, const f32x4_t& v) {
    if (vec) { if (x < W) *(f32x4_t*)(base + row * W + x) = v; }
    else {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (x + k < W) base[row * W + x + k] = v[k];
    }
}

// Depth tile: rows y0-R .. y0+TH+R-1, columns x0-R .. x0+TW+R-1 of sample b, staged as DEPTH into sd (row stride LS, column j
// at 4 + j).  Two steps so that the loads are in flight together with the workgroup's other loads: issue -> registers, commit
// -> reciprocal + LDS store.
template <int R> struct DepthTile {
    static constexpr int ROWS = TH + 2 * R;
    static constexpr int NI = (ROWS * (TW / 4) + 255) / 256;      // interior float4 groups per thread
    static_assert(ROWS * 2 * R <= 256, "one halo pixel per thread");
    f32x4_t v[NI];
    float hv;
    __device__ __forceinline__ void issue(const EdgeScale& sc, int vec, int b, int x0, int y0) {
        const float* img = sc.pred + (long)b * sc.H * sc.W;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = threadIdx.x + k * 256;
            const int ly = i >> 4, c4 = (i & 15) * 4;
            const int gy = y0 + ly - R;
            v[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (i < ROWS * (TW / 4) && (unsigned)gy < (unsigned)sc.H) v[k] = load4(img, gy, x0 + c4, sc.W, vec);
        }
        hv = 0.f;
        const int i = threadIdx.x;
        if (i < ROWS * 2 * R) {
            const int ly = i / (2 * R), k = i % (2 * R);
            const int j = k < R ? k - R : TW + (k - R);
            const int gy = y0 + ly - R, gx = x0 + j;
            if ((unsigned)gy < (unsigned)sc.H && (unsigned)gx < (unsigned)sc.W) hv = img[(long)gy * sc.W + gx];
        }
    }
    __device__ __forceinline__ void commit(const EdgeScale& sc, int from_inv, int x0, int y0, float* sd) const {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i >= ROWS * (TW / 4)) break;
            const int ly = i >> 4, c4 = (i & 15) * 4;
            const bool rowok = (unsigned)(y0 + ly - R) < (unsigned)sc.H;
            f32x4_t d;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = (rowok && x0 + c4 + e < sc.W) ? to_depth(from_inv, v[k][e]) : 0.f;
            *(f32x4_t*)(sd + ly * LS + 4 + c4) = d;
        }
        const int i = threadIdx.x;
        if (i < ROWS * 2 * R) {
            const int ly = i / (2 * R), k = i % (2 * R);
            const int j = k < R ? k - R : TW + (k - R);
            const int gy = y0 + ly - R, gx = x0 + j;
            sd[ly * LS + 4 + j] = ((unsigned)gy < (unsigned)sc.H && (unsigned)gx < (unsigned)sc.W) ? to_depth(from_inv, hv) : 0.f;
        }
    }
};

// the 3 x 6 window around 4 consecutive pixels: w[r][0..5] = columns c-1 .. c+4 of LDS rows (ly-1, ly, ly+1); c % 4 == 0
__device__ __forceinline__ void window(const float* sd, int ly, int c, float w[3][6]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float* row = sd + (ly - 1 + r) * LS + 4 + c;
        const f32x4_t m = *(const f32x4_t*)row;
        w[r][0] = row[-1]; w[r][1] = m[0]; w[r][2] = m[1]; w[r][3] = m[2]; w[r][4] = m[3]; w[r][5] = row[4];
    }
}
// Sobel responses of pixel k (0..3) of the window -- kernels of grad_loss.py:20-31
__device__ __forceinline__ void sobel4(const float w[3][6], int k, float& sh, float& sv, float& srl, float& slr) {
    const float n0 = w[0][k], n1 = w[0][k + 1], n2 = w[0][k + 2], n3 = w[1][k], n5 = w[1][k + 2], n6 = w[2][k], n7 = w[2][k + 1], n8 = w[2][k + 2];
    sh = (n2 - n0) + 2.f * (n5 - n3) + (n8 - n6);
    sv = (n6 - n0) + 2.f * (n7 - n1) + (n8 - n2);
    srl = (n1 - n3) + 2.f * (n2 - n6) + (n5 - n7);
    slr = (n5 - n1) + 2.f * (n8 - n0) + (n7 - n3);
}
// 1-ulp reciprocal (v_rcp_f32): enough wherever the result is not differenced against a neighbour (the depth tile keeps IEEE division)
__device__ __forceinline__ float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
// natural log / exp on the transcendental unit without the denormal-range scaling of __logf / __expf (5 extra instructions each): every
// argument here is >= 1e-4 (p + 0.001, 10 (inv + 1e-5), 10 / depth), and an exp that underflows may flush to zero (1 + t follows)
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// wave-wide sum, every lane ends with the total: xor butterfly over quad_perm / row_half_mirror / row_mirror DPP operands (lanes 1, 2, 4, 8 apart)
// and the two lane-swap instructions of gfx950 (rows, then halves) -- 6 adds + 2 swaps instead of 6 ds_bpermute round trips
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);          // {rows 0 0 2 2, rows 1 1 3 3}
    v = __builtin_bit_cast(float, (unsigned)r16[0]) + __builtin_bit_cast(float, (unsigned)r16[1]);
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto r32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);          // {lower half twice, upper half twice}
    return __builtin_bit_cast(float, (unsigned)r32[0]) + __builtin_bit_cast(float, (unsigned)r32[1]);
}

struct BlockId { int s, b, t0, t1; };                            // tiles t0 .. t1 - 1 of sample b of scale s
__device__ __forceinline__ BlockId decode_block(const EdgeMulti& a) {
    BlockId id;
    int s = 0;
#pragma unroll
    for (int k = 1; k < MAXS; ++k) if (k < a.nscales && (int)blockIdx.x >= a.s[k].first_block) s = k;
    const int r = blockIdx.x - a.s[s].first_block;
    const int g = r % a.s[s].groups, tiles = a.s[s].tiles_x * a.s[s].tiles_y;
    id.s = s; id.b = r / a.s[s].groups;
    id.t0 = g * a.tiles_per_wg; id.t1 = id.t0 + a.tiles_per_wg < tiles ? id.t0 + a.tiles_per_wg : tiles;
    return id;
}

// ---------------- forward -----------------------------------------------------------------------------------------

__device__ void finalize_losses(const EdgeMulti& a, double* stage, int stage_elems);
// Sums of one workgroup -> record -> image sums -> losses (shared by the generic and the fast forward kernel).  sd: the depth tile's LDS
// storage ((TH + 2) * LS floats, dead by now), sred / s_last: LDS scratch of the kernel.
__device__ __forceinline__ void forward_tail(const EdgeMulti& a, const BlockId& id, float acc[NP], bool has_mask, bool silog,
                                             float* sd, float (*sred)[NP], int* s_last_p) {
    const EdgeScale& sc = a.s[id.s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    volatile int& s_last = *s_last_p;
    // ---- sums.  fp32 per thread and per wave (xor butterfly in DPP / lane-swap instructions), fp64 from there on.  Round 4: NO atomic
    //      adds.  The workgroup leaves ONE record (its own 128-byte line); the last workgroup of a (scale, sample) to arrive adds that
    //      image's <= 240 records in a FIXED order and writes the image's sums; the last of those finishes the losses.  Every sum of the
    //      launch has a fixed order, so losses and coefficients are bit-reproducible.  (Before: one fp64 atomicAdd per value into the
    //      image's accumulators -- 7 x 240 adds on ONE cache line per full-resolution image, serialised at its L2 channel: 27 us of a
    //      54 us launch.  A first fixed-order attempt in round 1 had ONE workgroup sweep all 2,568 records: ~120 us.)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i >= 4 && i < 10 && !has_mask) continue;
        if (i >= 10 && !silog) continue;
        const float s = wave_sum_dpp(acc[i]);
        if (lane == 0) sred[wave][i] = s;
    }
    __syncthreads();
    const bool live = tid < NP && !((tid >= 4 && tid < 10 && !has_mask) || (tid >= 10 && !silog));
    if (live) {
        // RETURNING exchange: the value comes back only after the store has been performed at the memory side (a plain store, or a
        // no-return atomic, is acknowledged earlier: the ticket below was seen to overtake it about once per few thousand workgroups)
        const double v = (double)sred[0][tid] + (double)sred[1][tid] + (double)sred[2][tid] + (double)sred[3][tid];
        const unsigned long long before = atomicExch((unsigned long long*)(a.records + (long)blockIdx.x * REC) + tid, (unsigned long long)__double_as_longlong(v));
        asm volatile("" ::"v"(before));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int per_image = sc.groups;                               // records (= workgroups) of this image
    if (tid == 0) {
        // (the kernel stores nothing but its records: a release fence here writes back no output lines of its own)
        if (a.fences) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        s_last = __hip_atomic_fetch_add(a.counter + 1 + id.s * a.B + id.b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(per_image - 1);
        if (s_last && a.fences) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();
    if (!s_last) return;
    // ---- last workgroup of this (scale, sample): value v = tid % 16 of records k, k + 16, ... (k = tid / 16), then the 16 part sums in order
    {
        double* s_fin = (double*)sd;                               // the tile is dead (barriers above)
        const int v = tid & 15, k = tid >> 4;
        const double* rec = a.records + ((long)sc.first_block + (long)id.b * per_image) * REC + v;
        double part = 0.0;
        if (v < NP && !((v >= 4 && v < 10 && !has_mask) || (v >= 10 && !silog))) {
#pragma unroll 4
            for (int j = k; j < per_image; j += 16) part += __hip_atomic_load(rec + (long)j * REC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_fin[k * 16 + v] = part;
        __syncthreads();
        if (live) {
            double tot = 0.0;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) tot += s_fin[kk * 16 + tid];
            const unsigned long long before = atomicExch((unsigned long long*)(a.results + ((long)id.s * a.B + id.b) * NP) + tid, (unsigned long long)__double_as_longlong(tot));
            asm volatile("" ::"v"(before));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (a.fences) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            s_last = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(a.nscales * a.B - 1);
            if (s_last && a.fences) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        __syncthreads();
        if (!s_last) return;
    }
    finalize_losses(a, (double*)sd, (TH + 2) * LS / 2);
}

// FAST = the training configuration on every scale (16-byte accesses legal, inverse depth in, Sobel + normals + sigmoid, no mask):
// the flags are compile-time there.  The generic instantiation carries every runtime flag -- its body is ~10k instructions
// (scalar and vector access paths, magnitude / probability / mask branches), several times what the instruction cache likes.
template <bool FAST>
__global__ __launch_bounds__(256, FAST ? 4 : 2) void edge_loss_fwd_kernel(EdgeMulti a) {
    __shared__ __attribute__((aligned(16))) float sd[(TH + 2) * LS];
    __shared__ float sred[4][NP];
    __shared__ int s_last;
    const BlockId id = decode_block(a);
    const EdgeScale& sc = a.s[id.s];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = (tid & 15) * 4, r0 = tid >> 4;                   // this thread's 4 pixels: columns c..c+3 of rows r0 and r0 + 16
    const bool has_mask = FAST ? false : sc.mask != nullptr;
    const bool silog = a.gt_depth != nullptr && id.s == 0;
    const int vec = FAST ? 1 : sc.vec;
    const bool is_grad = FAST ? true : a.is_grad != 0, is_sigmoid = FAST ? true : a.is_sigmoid != 0;
    const bool has_normal = FAST ? true : sc.normal != nullptr;
    const int from_inv = FAST ? 1 : a.from_inv;
    const long img = (long)id.b * sc.H;

    float acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.f;
    // a workgroup walks `tiles_per_wg` consecutive tiles of its image and keeps the sums in registers: the record / ticket round trips
    // at the end (two dependent trips to the memory side, ~5 us with the slot held) are paid once per 3 tiles instead of per tile
    for (int t = id.t0; t < id.t1; ++t) {
    const int x0 = (t % sc.tiles_x) * TW, y0 = (t / sc.tiles_x) * TH;
    if (t != id.t0) __syncthreads();                              // the previous tile's window reads are done
        // every global load of the workgroup first: one memory latency for the tile, the labels and the normals together
        DepthTile<1> tile;
        if (is_grad) tile.issue(sc, vec, id.b, x0, y0);
        f32x4_t e4[2], n4[2], m4[2], d4[2], i4[2];
    #pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int gy = y0 + r0 + 16 * ps;
            const bool ok = gy < sc.H;
            e4[ps] = ok ? load4(sc.edge, img + gy, x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (has_normal && is_grad) n4[ps] = ok ? load4(sc.normal, img + gy, x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (has_mask) m4[ps] = ok ? load4(sc.mask, img + gy, x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (silog) {
                d4[ps] = ok ? load4(a.gt_depth, img + gy, x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                i4[ps] = ok ? load4(sc.pred, img + gy, x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            if (!is_grad) i4[ps] = ok ? load4(sc.pred, img + gy, x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        if (is_grad) tile.commit(sc, from_inv, x0, y0, sd);
        __syncthreads();

#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int ly = r0 + 16 * ps, gy = y0 + ly;
        if (gy >= sc.H) continue;
        float w[3][6];
        if (is_grad) window(sd, ly + 1, c, w);
        f32x4_t g4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x0 + c + k >= sc.W) { g4[k] = 0.f; continue; }
            float g;
            if (is_grad) {
                float sh, sv, srl, slr;
                sobel4(w, k, sh, sv, srl, slr);
                if (has_normal) {
                    const int code = direction_code(n4[ps][k]);
                    g = fabsf(code == 0 ? sh : (code == 1 ? sv : (code == 2 ? srl : slr)));
                } else {
                    g = sqrtf(sv * sv + sh * sh + 1e-6f);
                }
            } else {
                g = i4[ps][k];
            }
            g4[k] = g;
            // p = 1 / (1 + t), 1 - p = t p with t = exp(-(g - thresh)): the reference forms 1 - p by subtraction in float32, which
            // loses everything once p rounds towards 1 (g - thresh > ~8); t p is exact to an ulp and agrees wherever that is defined
            float p, omp;
            if (is_sigmoid) { const float t = fast_exp(-(g - a.thresh)); p = rcpf(1.f + t); omp = t * p; }
            else { p = g; omp = 1.f - g; }
            const float e = e4[ps][k];
#if defined(MTE_EDGE_ABLATE) && (MTE_EDGE_ABLATE & 2)
            const float pos = -e * g, neg = -(1.f - e) * g;          // diagnostic: no exp / rcp / log
#else
            const float pos = -e * fast_log(p + 0.001f), neg = -(1.f - e) * fast_log(omp + 0.001f);
#endif
            acc[2] += pos; acc[3] += neg;
            if (has_mask) {
                const float m = m4[ps][k];
                const float keep = m != 0.f ? 1.f : 0.f;
                acc[0] += e * m; acc[1] += (1.f - e) * m;
                acc[4] += pos * keep; acc[5] += neg * keep;
                acc[6] += m == 0.f ? 1.f : 0.f; acc[7] += m == 1.f ? 1.f : 0.f; acc[8] += (m != 0.f && m != 1.f) ? 1.f : 0.f; acc[9] += m;
            } else {
                acc[0] += e; acc[1] += 1.f - e;
            }
            if (silog) {
                const float d = d4[ps][k];
                if (d > 0.f) {
                    const float gt = rcpf(fmaxf(d, 1e-6f));           // feeds a log: 1 ulp here is 6e-8 absolute there
                    const float dl = fast_log((i4[ps][k] + 1e-5f) * 10.f) - fast_log(gt * 10.f);
                    acc[10] += dl; acc[11] = fmaf(dl, dl, acc[11]); acc[12] += 1.f;
                }
            }
        }
        if (!FAST && sc.gmap) store4(sc.gmap, img + gy, x0 + c, sc.W, vec, g4);
    }
    }   // tiles of this workgroup
#if defined(MTE_EDGE_ABLATE) && (MTE_EDGE_ABLATE & 1)
    { float tt = 0.f; for (int i = 0; i < NP; ++i) tt += acc[i]; if (tt == 123.456f) a.losses[0] = tt; return; }      // diagnostic: no reductions, atomics, tickets
#endif
    forward_tail(a, id, acc, has_mask, silog, sd, sred, &s_last);
}

// Runs in the last workgroup: every accumulator is complete.  The scalar arithmetic of comp_cross_entropy (grad_loss.py:161-219) and
// SilogLoss (supervised_loss.py:57-69) is a few hundred dependent reads of the accumulators; straight from memory (agent-scope loads,
// one L2 round trip each, one thread) that serial tail was ~45 us of a 73 us launch (round 4: forward 73 -> see profiles/README.md).  So the
// whole workgroup first copies the sums into LDS (`stage`: the depth tile's storage, free by now; one round trip); one thread per
// (scale, sample) does the divisions, one per scale the ordered sum (+ one for the silog loss, in another wave).
__device__ __forceinline__ double acc_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// class-balance arithmetic of one scale from its image sums R[b * NP + i] (comp_cross_entropy, grad_loss.py:161-219)
struct ScaleStats { bool binary; double nvalid, wneg_total; };
template <typename V> __device__ __forceinline__ ScaleStats scale_stats(const V& R0, long R, int B, bool has_mask, double numel) {
    double mi[4] = {0.0, 0.0, 0.0, 0.0};
    if (has_mask)
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < 4; ++k) mi[k] += R0(R + b * NP + 6 + k);
    ScaleStats st;
    st.binary = has_mask && mi[2] == 0.0 && mi[0] > 0.0 && mi[1] > 0.0;          // unique(mask) == {0, 1}
    st.nvalid = st.binary ? mi[3] : numel;
    st.wneg_total = 0.0;
    for (int b = 0; b < B; ++b) st.wneg_total += (double)(float)R0(R + b * NP + 1);
    return st;
}
// sample b of a scale: backward coefficients -> coef[2b], coef[2b + 1]; returns its term of the loss sum
template <typename V> __device__ __forceinline__ double sample_term(const EdgeMulti& a, const V& R0, long sums, const ScaleStats& st, float* coef, int b) {
    const float wp = (float)R0(sums), wn = (float)R0(sums + 1);
    const float alpha = st.wneg_total == 0.0 ? 1.f : wn / (wp + wn);
    const double P = st.binary ? R0(sums + 4) : R0(sums + 2), N = st.binary ? R0(sums + 5) : R0(sums + 3);
    coef[2 * b] = (float)((double)a.weight * a.pos_to_neg * alpha / st.nvalid);
    coef[2 * b + 1] = (float)((double)a.weight * (1.f - alpha) / st.nvalid);
    return (double)a.pos_to_neg * alpha * P + (double)(1.f - alpha) * N;
}
template <typename V> __device__ __forceinline__ void silog_finish(const EdgeMulti& a, const V& R0) {   // loss = 10 sqrt(E[d^2] - 0.85 E[d]^2);  aux = (mean, 10/sqrt(S)/n)
    double ss[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < a.B; ++b)
        for (int k = 0; k < 3; ++k) ss[k] += R0((long)b * NP + 10 + k);
    const double n1 = ss[2];
    const double m1 = ss[0] / n1, m2 = ss[1] / n1;
    const double S = m2 - 0.85 * m1 * m1;
    if (a.silog_loss) *a.silog_loss = (float)(sqrt(S) * 10.0);
    if (a.silog_aux) { a.silog_aux[0] = (float)m1; a.silog_aux[1] = (float)(10.0 / sqrt(S) / n1); }
}
struct LdsView { const double* l; __device__ __forceinline__ double operator()(long i) const { return l[i]; } };
struct MemView { const double* g; __device__ __forceinline__ double operator()(long i) const { return acc_load(g + i); } };

__device__ void finalize_losses(const EdgeMulti& a, double* stage, int stage_elems) {
    const int tid = threadIdx.x;
    if (!a.finalize) return;
    const int n = a.nscales * a.B * NP, images = a.nscales * a.B;
    if (n + images <= stage_elems) {
        // the usual case.  (1) sums -> LDS; (2) one thread per (scale, sample): alpha, the two coefficients (fp64 divisions -- done by one
        // thread per SCALE these were ~3 us of serial tail) and the sample's term of the loss; (3) one thread per scale adds the terms in order
        __syncthreads();                                          // every wave is done with the tile
        for (int i = tid; i < n; i += 256) stage[i] = acc_load(a.results + i);
        __syncthreads();
        const LdsView R0{stage};
        double* term = stage + n;
        for (int i = tid; i < images; i += 256) {
            const int s = i / a.B, b = i - s * a.B;
            const long R = (long)s * a.B * NP;
            const ScaleStats st = scale_stats(R0, R, a.B, a.s[s].mask != nullptr, (double)a.B * a.s[s].H * a.s[s].W);
            float* coef = a.coef + (long)s * (2 * a.B + 1);
            term[i] = sample_term(a, R0, R + b * NP, st, coef, b);
            if (b == 0) coef[2 * a.B] = st.binary ? 1.f : 0.f;
        }
        if (a.gt_depth && tid == 255) silog_finish(a, R0);         // another wave than the first threads
        __syncthreads();
        if (tid < a.nscales) {
            const int s = tid;
            const ScaleStats st = scale_stats(R0, (long)s * a.B * NP, a.B, a.s[s].mask != nullptr, (double)a.B * a.s[s].H * a.s[s].W);
            double total = 0.0;
            for (int b = 0; b < a.B; ++b) total += term[s * a.B + b];
            a.losses[s] = (float)((double)a.weight * total / st.nvalid);
        }
        return;
    }
    if (tid != 0) return;                                         // very large batches: one thread, straight from memory
    const MemView R0{a.results};
    for (int s = 0; s < a.nscales; ++s) {
        const long R = (long)s * a.B * NP;
        const ScaleStats st = scale_stats(R0, R, a.B, a.s[s].mask != nullptr, (double)a.B * a.s[s].H * a.s[s].W);
        float* coef = a.coef + (long)s * (2 * a.B + 1);
        double total = 0.0;
        for (int b = 0; b < a.B; ++b) total += sample_term(a, R0, R + b * NP, st, coef, b);
        coef[2 * a.B] = st.binary ? 1.f : 0.f;
        a.losses[s] = (float)((double)a.weight * total / st.nvalid);
    }
    if (a.gt_depth) silog_finish(a, R0);
}

// single-scale finalize (GradLoss called on its own): accumulators [B][NP] -> loss (accumulated into *loss_acc with factor
// out_scale) and backward coefficients
__global__ void edge_loss_finalize_kernel(const double* __restrict__ sums, int B, long numel, float weight, float pos_to_neg,
                                          int has_mask, float out_scale, float* __restrict__ loss_acc, float* __restrict__ loss_this,
                                          float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double mi[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < 4; ++k) mi[k] += sums[b * NP + 6 + k];
    const bool binary = has_mask && mi[2] == 0.0 && mi[0] > 0.0 && mi[1] > 0.0;      // unique(mask) == {0, 1}
    const double nvalid = binary ? mi[3] : (double)numel;
    double wneg_total = 0.0;
    for (int b = 0; b < B; ++b) wneg_total += (double)(float)sums[b * NP + 1];
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        const float wp = (float)sums[b * NP], wn = (float)sums[b * NP + 1];
        const float alpha = wneg_total == 0.0 ? 1.f : wn / (wp + wn);
        const double P = binary ? sums[b * NP + 4] : sums[b * NP + 2], N = binary ? sums[b * NP + 5] : sums[b * NP + 3];
        total += (double)pos_to_neg * alpha * P + (double)(1.f - alpha) * N;
        coef[2 * b] = (float)((double)weight * pos_to_neg * alpha / nvalid);
        coef[2 * b + 1] = (float)((double)weight * (1.f - alpha) / nvalid);
    }
    coef[2 * B] = binary ? 1.f : 0.f;
    const float l = (float)((double)weight * total / nvalid);
    if (loss_this) *loss_this = l;
    if (loss_acc) *loss_acc += out_scale * l;
}

// ---------------- backward ----------------------------------------------------------------------------------------
// Transposed Sobel.  d s_code(p) / d depth(p + t) for tap t = (dy, dx) is  h: dx (2 - |dy|), v: dy (2 - |dx|), rl: dx - dy, lr: dx + dy.  With the
// four tap patterns X[t] = dx, Y[t] = dy, Xc[t] = dx [dy = 0], Yc[t] = dy [dx = 0] these are h = X + Xc, v = Y + Yc, rl = X - Y, lr = X + Y, so
//   d loss / d depth(q) = sum_t  A(q - t) X[t] + B(q - t) Y[t] + C(q - t) Xc[t] + D(q - t) Yc[t]
// with per-pixel planes A = G [code != v], B = G (v: 1, rl: -1, lr: 1, h: 0), C = G [code = h], D = G [code = v].  X and Y are separable
// (column sums of A, row differences of B shared by the 4 pixels of a thread), Xc / Yc touch one row / one column, and C, D need no storage:
// C = A where B = 0, D = B where A = 0 (G = 0 makes all four vanish).  ~80 instructions per 4 pixels; the per-tap select by the neighbour's
// code it replaces (rounds 1-3) took ~420.  Magnitude mode (no normals): A = C = the h part, B = D = the v part.
template <bool FAST>
__global__ __launch_bounds__(256) void edge_loss_bwd_kernel(EdgeMulti a) {
    // depth on the tile + 2-pixel halo; the planes A, B of G = d loss / d s(p) on the tile + 1-pixel halo
    __shared__ __attribute__((aligned(16))) float sd[(TH + 4) * LS];
    __shared__ __attribute__((aligned(16))) float sga[(TH + 2) * LS], sgb[(TH + 2) * LS];        // planes A and B (see above)
    const BlockId id = decode_block(a);                            // one tile per workgroup (tiles_per_wg = 1)
    const EdgeScale& sc = a.s[id.s];
    const int x0 = (id.t0 % sc.tiles_x) * TW, y0 = (id.t0 / sc.tiles_x) * TH;
    const int tid = threadIdx.x;
    const int c = (tid & 15) * 4, r0 = tid >> 4;
    const long img = (long)id.b * sc.H;
    const float go = a.gout ? a.gout[id.s] : 1.f;
    const float* coef = a.coef + (long)id.s * (2 * a.B + 1);
    const float cpos = coef[2 * id.b] * go, cneg = coef[2 * id.b + 1] * go;
    const bool use_keep = FAST ? false : (coef[2 * a.B] != 0.f && sc.mask != nullptr);
    const bool silog = a.gt_depth != nullptr && id.s == 0;
    const bool magnitude = FAST ? false : sc.normal == nullptr;
    const int vec = FAST ? 1 : sc.vec;
    const bool is_grad = FAST ? true : a.is_grad != 0, is_sigmoid = FAST ? true : a.is_sigmoid != 0;
    const int from_inv = FAST ? 1 : a.from_inv;

    // ---- every global load of the workgroup is issued here: depth tile, labels / normals of the G region, inputs of the output phase
    constexpr int GITEMS = (TH + 2) * (TW / 4 + 2), NG = (GITEMS + 255) / 256;     // G region: interior groups of 4 + two halo columns per row
    DepthTile<2> tile;
    f32x4_t ge[NG], gn[NG], gm[NG];
    f32x4_t inv4[2], d4[2], oe4[2], om4[2];
    if (is_grad) {
        tile.issue(sc, vec, id.b, x0, y0);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = tid + k * 256;
            const int ly = i / (TW / 4 + 2), q = i % (TW / 4 + 2);               // q < 16: interior group, 16 / 17: left / right halo column
            const int gy = y0 + ly - 1;
            const bool group = q < TW / 4;
            const int j0 = group ? q * 4 : (q == TW / 4 ? -1 : TW);
            ge[k] = f32x4_t{0.f, 0.f, 0.f, 0.f}; gn[k] = ge[k]; gm[k] = f32x4_t{1.f, 1.f, 1.f, 1.f};
            if (i < GITEMS && (unsigned)gy < (unsigned)sc.H) {
                if (group) {
                    ge[k] = load4(sc.edge, img + gy, x0 + j0, sc.W, vec);
                    if (!magnitude) gn[k] = load4(sc.normal, img + gy, x0 + j0, sc.W, vec);
                    if (use_keep) gm[k] = load4(sc.mask, img + gy, x0 + j0, sc.W, vec);
                } else if ((unsigned)(x0 + j0) < (unsigned)sc.W) {
                    const long idx = (img + gy) * sc.W + x0 + j0;
                    ge[k][0] = sc.edge[idx];
                    if (!magnitude) gn[k][0] = sc.normal[idx];
                    if (use_keep) gm[k][0] = sc.mask[idx];
                }
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int gy = y0 + r0 + 16 * ps;
        const bool ok = gy < sc.H && x0 + c < sc.W;
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        inv4[ps] = ok ? load4(sc.pred, img + gy, x0 + c, sc.W, vec) : z;
        if (silog) d4[ps] = ok ? load4(a.gt_depth, img + gy, x0 + c, sc.W, vec) : z;
        if (!is_grad) {
            oe4[ps] = ok ? load4(sc.edge, img + gy, x0 + c, sc.W, vec) : z;
            om4[ps] = (ok && use_keep) ? load4(sc.mask, img + gy, x0 + c, sc.W, vec) : f32x4_t{1.f, 1.f, 1.f, 1.f};
        }
    }
    if (is_grad) {
        tile.commit(sc, from_inv, x0, y0, sd);
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < NG; ++kq) {
            const int i = tid + kq * 256;
            if (i >= GITEMS) break;
            const int ly = i / (TW / 4 + 2), q = i % (TW / 4 + 2);
            const int gy = y0 + ly - 1;
            const bool group = q < TW / 4;
            const int j0 = group ? q * 4 : (q == TW / 4 ? -1 : TW);
            const int np = group ? 4 : 1;
            const f32x4_t e4 = ge[kq], n4 = gn[kq], m4 = gm[kq];
            const bool rowok = (unsigned)gy < (unsigned)sc.H;
            float w[3][6];
            if (group) window(sd, ly + 1, j0, w);
            else {                                                            // single column: centre it at window column 1
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) w[r][k] = sd[(ly + r) * LS + 4 + j0 - 1 + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= np) break;
                const int gx = x0 + j0 + k;
                float ga = 0.f, gb = 0.f;       // planes A (X coefficient) and B (Y coefficient)
                if (rowok && (unsigned)gx < (unsigned)sc.W) {
                    float sh, sv, srl, slr;
                    sobel4(w, k, sh, sv, srl, slr);
                    float g, da, db = 0.f;        // d g / d s_a, d g / d s_b
                    int code = 0;
                    if (!magnitude) {
                        code = direction_code(n4[k]);
                        const float s = code == 0 ? sh : (code == 1 ? sv : (code == 2 ? srl : slr));
                        g = fabsf(s);
                        da = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
                    } else {
                        g = sqrtf(sv * sv + sh * sh + 1e-6f);
                        da = sv / g; db = sh / g;                  // a = v, b = h
                    }
                    float p, omp;                  // p and 1 - p without cancellation (see the forward kernel)
                    if (is_sigmoid) { const float t = fast_exp(-(g - a.thresh)); p = rcpf(1.f + t); omp = t * p; }
                    else { p = g; omp = 1.f - g; }
                    const float e = e4[k];
                    const float keep = (use_keep && m4[k] == 0.f) ? 0.f : 1.f;
                    const float dp = is_sigmoid ? p * omp : 1.f;
                    const float dg = keep * dp * (-cpos * e * rcpf(p + 0.001f) + cneg * (1.f - e) * rcpf(omp + 0.001f));
                    if (magnitude) { ga = dg * db; gb = dg * da; }             // h part, v part
                    else {
                        const float G = dg * da;
                        ga = code != 1 ? G : 0.f;
                        gb = code == 1 ? G : (code == 2 ? -G : (code == 3 ? G : 0.f));
                    }
                }
                const int o = ly * LS + 4 + j0 + k;
                sga[o] = ga; sgb[o] = gb;
            }
        }
        __syncthreads();
    }
    const float m1 = silog ? a.silog_aux[0] : 0.f, ksl = silog ? a.silog_aux[1] * (a.silog_gout ? a.silog_gout[0] : 1.f) : 0.f;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int ly = r0 + 16 * ps, gy = y0 + ly;
        if (gy >= sc.H || x0 + c >= sc.W) continue;
        f32x4_t out = {0.f, 0.f, 0.f, 0.f};
        if (is_grad) {
            // the 3 x 6 windows of the planes around the 4 pixels; window column j = image column c - 1 + j, pixel k sits at column k + 1
            float wa[3][6], wb[3][6];
            window(sga, ly + 1, c, wa);
            window(sgb, ly + 1, c, wb);
            float ca[6], eb[6], cc1[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                ca[j] = (wa[0][j] + wa[1][j]) + wa[2][j];                                   // column sums of A
                eb[j] = wb[0][j] - wb[2][j];                                                // row above - row below of B
                cc1[j] = magnitude ? wa[1][j] : (wb[1][j] == 0.f ? wa[1][j] : 0.f);         // C on the centre row
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // source p at window (r, k + cc) reaches q = (1, k + 1) with t = q - p = (1 - r, 1 - cc)
                const float d0 = magnitude ? wb[0][k + 1] : (wa[0][k + 1] == 0.f ? wb[0][k + 1] : 0.f);       // D above / below q
                const float d2 = magnitude ? wb[2][k + 1] : (wa[2][k + 1] == 0.f ? wb[2][k + 1] : 0.f);
                float dd = (ca[k] - ca[k + 2]) + ((eb[k] + eb[k + 1]) + eb[k + 2]) + (cc1[k] - cc1[k + 2]) + (d0 - d2);
                if (from_inv) {
                    const float inv = inv4[ps][k];
                    const float d = rcpf(fmaxf(inv, 1e-6f));
                    dd = inv >= 1e-6f ? -dd * d * d : 0.f;
                }
                out[k] = dd;
            }
        } else {
            const f32x4_t e4 = oe4[ps], m4 = om4[ps];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = inv4[ps][k];
                float p, omp;
                if (is_sigmoid) { const float t = fast_exp(-(g - a.thresh)); p = rcpf(1.f + t); omp = t * p; }
                else { p = g; omp = 1.f - g; }
                const float keep = (use_keep && m4[k] == 0.f) ? 0.f : 1.f;
                const float dp = is_sigmoid ? p * omp : 1.f;
                out[k] = keep * dp * (-cpos * e4[k] * rcpf(p + 0.001f) + cneg * (1.f - e4[k]) * rcpf(omp + 0.001f));
            }
        }
        if (silog) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = d4[ps][k];
                if (d > 0.f) {
                    const float gt = rcpf(fmaxf(d, 1e-6f));           // feeds a log: 1 ulp here is 6e-8 absolute there
                    const float pi = inv4[ps][k] + 1e-5f;
                    const float dl = fast_log(pi * 10.f) - fast_log(gt * 10.f);
                    out[k] += ksl * (dl - 0.85f * m1) * rcpf(pi);
                }
            }
        }
        store4(sc.dpred, img + gy, x0 + c, sc.W, vec, out);
    }
}

// ======================= the training configuration: branch-free kernels ============================================================
// Every scale: 16-byte accesses legal (W % 4 == 0), inverse depth in, Sobel + normals + sigmoid, no mask, no edge-map output (`fast_config`).
// Round 4.  The generic kernels above spend a third of their instructions on control flow: per-pixel bounds tests compiled to exec-mask
// branches with their phi copies, divergent loop exits between 4-pixel and 1-pixel items, a load guard per access.  Here
//   * every global address is CLAMPED into the image and the loaded value selected to zero afterwards -- no guard around a load;
//   * a thread stages the depth of ITS OWN 8 pixels (the same load feeds the silog term and the chain rule), the halo rows / columns are
//     single extra loads of the first threads: no index arithmetic by division, no scalar / vector load variants;
//   * validity is one factor per 4-pixel group folded into the sums (1 - e becomes valid - e), the silog mask is a select;
//   * the direction bins are three range tests (edge_direction.hpp) that select the Sobel response and the backward planes directly;
//   * logs stay in base 2 until the sums (the ln 2 is folded into the label factors).
// Per 8 pixels of a thread: ~1,000 instructions forward (generic: 2,450), ~1,500 backward (3,100).
struct OwnGroup { bool ok; long off; };                        // a thread's 4 pixels of one row: inside the image?  offset of the (clamped) group
__device__ __forceinline__ OwnGroup own_group(const EdgeScale& sc, int b, int gy, int gx) {
    OwnGroup g;
    g.ok = (unsigned)gy < (unsigned)sc.H && (unsigned)gx < (unsigned)sc.W;
    const int cy = min(max(gy, 0), sc.H - 1), cx = min(max(gx, 0), sc.W - 4);
    g.off = ((long)b * sc.H + cy) * sc.W + cx;
    return g;
}
__device__ __forceinline__ f32x4_t sel4(bool ok, const f32x4_t& v) { return f32x4_t{ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f}; }
__device__ __forceinline__ f32x4_t depth4(bool ok, const f32x4_t& inv) {
    return f32x4_t{ok ? rcp_newton(fmaxf(inv[0], 1e-6f)) : 0.f, ok ? rcp_newton(fmaxf(inv[1], 1e-6f)) : 0.f,
                   ok ? rcp_newton(fmaxf(inv[2], 1e-6f)) : 0.f, ok ? rcp_newton(fmaxf(inv[3], 1e-6f)) : 0.f};
}
// signed Sobel response of the direction the normal selects
__device__ __forceinline__ float directed_response(const float w[3][6], int k, const DirMasks& m) {
    float sh, sv, srl, slr;
    sobel4(w, k, sh, sv, srl, slr);
    float s = m.v ? sv : sh;
    s = m.k1 ? (m.neg ? slr : srl) : s;
    s = m.k3 ? (m.neg ? srl : slr) : s;
    return s;
}
constexpr float LOG2E = 1.44269504088896341f, LN2 = 0.693147180559945309f;

__global__ __launch_bounds__(256, 4) void edge_fwd_fast_kernel(EdgeMulti a) {
    __shared__ __attribute__((aligned(16))) float sd[(TH + 2) * LS];
    __shared__ float sred[4][NP];
    __shared__ int s_last;
    const BlockId id = decode_block(a);
    const EdgeScale& sc = a.s[id.s];
    const int tid = threadIdx.x;
    const int c = (tid & 15) * 4, r0 = tid >> 4;                   // this thread's 4 pixels: columns c..c+3 of rows r0 and r0 + 16
    const bool silog = a.gt_depth != nullptr && id.s == 0;
    const float tk = a.thresh * LOG2E;
    float acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.f;
    for (int t = id.t0; t < id.t1; ++t) {
        const int x0 = (t % sc.tiles_x) * TW, y0 = (t / sc.tiles_x) * TH;
        if (t != id.t0) __syncthreads();                          // the previous tile's window reads are done
        // ---- loads: own pixels (inverse depth, label, normal, ground truth), then the halo of the depth tile
        OwnGroup og[2];
        f32x4_t inv4[2], e4[2], n4[2], d4[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            og[ps] = own_group(sc, id.b, y0 + r0 + 16 * ps, x0 + c);
            inv4[ps] = *(const f32x4_t*)(sc.pred + og[ps].off);
            e4[ps] = *(const f32x4_t*)(sc.edge + og[ps].off);
            n4[ps] = *(const f32x4_t*)(sc.normal + og[ps].off);
            if (silog) d4[ps] = *(const f32x4_t*)(a.gt_depth + og[ps].off);
        }
        f32x4_t hrow = {0.f, 0.f, 0.f, 0.f}; bool hrow_ok = false;
        float hcol = 0.f; bool hcol_ok = false;
        if (tid < 32) {                                            // halo rows -1 and TH: 2 x 16 groups
            const OwnGroup g = own_group(sc, id.b, y0 + (tid >> 4 ? TH : -1), x0 + c);
            hrow = *(const f32x4_t*)(sc.pred + g.off); hrow_ok = g.ok;
        } else if (tid >= 64 && tid < 64 + 2 * (TH + 2)) {         // halo columns -1 and TW of rows -1 .. TH
            const int i = tid - 64, gy = y0 + (i >> 1) - 1, gx = x0 + (i & 1 ? TW : -1);
            hcol_ok = (unsigned)gy < (unsigned)sc.H && (unsigned)gx < (unsigned)sc.W;
            hcol = sc.pred[((long)id.b * sc.H + min(max(gy, 0), sc.H - 1)) * sc.W + min(max(gx, 0), sc.W - 1)];
        }
        // ---- depth tile
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) *(f32x4_t*)(sd + (r0 + 16 * ps + 1) * LS + 4 + c) = depth4(og[ps].ok, inv4[ps]);
        if (tid < 32) *(f32x4_t*)(sd + (tid >> 4 ? TH + 1 : 0) * LS + 4 + c) = depth4(hrow_ok, hrow);
        else if (tid >= 64 && tid < 64 + 2 * (TH + 2)) {
            const int i = tid - 64;
            sd[(i >> 1) * LS + (i & 1 ? 4 + TW : 3)] = hcol_ok ? rcp_newton(fmaxf(hcol, 1e-6f)) : 0.f;
        }
        __syncthreads();
        // ---- the 8 pixels
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            float w[3][6];
            window(sd, r0 + 16 * ps + 1, c, w);
            const float vm = og[ps].ok ? 1.f : 0.f;
            const f32x4_t e = sel4(og[ps].ok, e4[ps]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = fabsf(directed_response(w, k, direction_masks(n4[ps][k])));
                // p = 1 / (1 + t), 1 - p = t p with t = exp(-(g - thresh)) (see the generic kernel)
                const float tt = __builtin_amdgcn_exp2f(__builtin_fmaf(g, -LOG2E, tk));
                const float p = rcpf(1.f + tt), omp = tt * p;
                const float ne = e[k] * -LN2, nf = (e[k] - vm) * LN2;            // -e ln 2, -(1 - e) ln 2 (0 outside the image)
                acc[0] += e[k]; acc[1] += vm - e[k];
                acc[2] = __builtin_fmaf(ne, __builtin_amdgcn_logf(p + 0.001f), acc[2]);
                acc[3] = __builtin_fmaf(nf, __builtin_amdgcn_logf(omp + 0.001f), acc[3]);
            }
            if (silog) {
                const f32x4_t d = sel4(og[ps].ok, d4[ps]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float gt = rcpf(fmaxf(d[k], 1e-6f));
                    const float dl = (__builtin_amdgcn_logf((inv4[ps][k] + 1e-5f) * 10.f) - __builtin_amdgcn_logf(gt * 10.f)) * LN2;
                    const float m = d[k] > 0.f ? 1.f : 0.f, dlm = d[k] > 0.f ? dl : 0.f;
                    acc[10] += dlm; acc[11] = __builtin_fmaf(dlm, dlm, acc[11]); acc[12] += m;     // dl itself may be NaN where d = 0
                }
            }
        }
    }
#if defined(MTE_EDGE_ABLATE) && (MTE_EDGE_ABLATE & 1)
    { float tt = 0.f; for (int i = 0; i < NP; ++i) tt += acc[i]; if (tt == 123.456f) a.losses[0] = tt; return; }      // diagnostic: no sums / records / tickets
#endif
    forward_tail(a, id, acc, false, silog, sd, sred, &s_last);
}

// one G evaluation: planes A and B of pixel k of a window (see "Transposed Sobel" above); cp / cn = cpos e, cneg (1 - e) (0 outside the image)
__device__ __forceinline__ void g_planes(const float w[3][6], int k, float n, float cp, float cn, float tk, float& A, float& B) {
    const DirMasks m = direction_masks(n);
    const float s = directed_response(w, k, m);
    const float tt = __builtin_amdgcn_exp2f(__builtin_fmaf(fabsf(s), -LOG2E, tk));
    const float p = rcpf(1.f + tt), omp = tt * p;
    const float dg = (p * omp) * (cn * rcpf(omp + 0.001f) - cp * rcpf(p + 0.001f));
    const float G = s > 0.f ? dg : (s < 0.f ? -dg : 0.f);           // d |s| / d s
    const float Gn = m.neg ? G : -G;
    A = m.v ? 0.f : G;
    B = m.v ? G : 0.f;
    B = m.k1 ? Gn : B;                                              // rl above zero (-1), lr below (+1)
    B = m.k3 ? -Gn : B;
}

__global__ __launch_bounds__(256, 4) void edge_bwd_fast_kernel(EdgeMulti a) {
    // depth on the tile + 2-pixel halo; the planes A, B of G = d loss / d s(p) on the tile + 1-pixel halo
    __shared__ __attribute__((aligned(16))) float sd[(TH + 4) * LS];
    __shared__ __attribute__((aligned(16))) float sga[(TH + 2) * LS], sgb[(TH + 2) * LS];
    const BlockId id = decode_block(a);                            // one tile per workgroup
    const EdgeScale& sc = a.s[id.s];
    const int x0 = (id.t0 % sc.tiles_x) * TW, y0 = (id.t0 / sc.tiles_x) * TH;
    const int tid = threadIdx.x;
    const int c = (tid & 15) * 4, r0 = tid >> 4;
    const float go = a.gout ? a.gout[id.s] : 1.f;
    const float* coef = a.coef + (long)id.s * (2 * a.B + 1);
    const float cpos = coef[2 * id.b] * go, cneg = coef[2 * id.b + 1] * go;
    const bool silog = a.gt_depth != nullptr && id.s == 0;
    const float tk = a.thresh * LOG2E;
    const long img = (long)id.b * sc.H;
    // ---- loads.  Own pixels: inverse depth, label, normal (+ ground truth); depth halo: rows -2, -1, TH, TH + 1 (threads 0..63) and columns
    //      -2, -1, TW, TW + 1 of rows -2 .. TH + 1 (threads 64..207); G halo (label + normal of ONE pixel): rows -1 and TH (threads 0..127), columns
    //      -1 and TW of rows -1 .. TH (threads 128..195)
    OwnGroup og[2];
    f32x4_t inv4[2], e4[2], n4[2], d4[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        og[ps] = own_group(sc, id.b, y0 + r0 + 16 * ps, x0 + c);
        inv4[ps] = *(const f32x4_t*)(sc.pred + og[ps].off);
        e4[ps] = *(const f32x4_t*)(sc.edge + og[ps].off);
        n4[ps] = *(const f32x4_t*)(sc.normal + og[ps].off);
        if (silog) d4[ps] = *(const f32x4_t*)(a.gt_depth + og[ps].off);
    }
    f32x4_t hrow = {0.f, 0.f, 0.f, 0.f}; bool hrow_ok = false;
    float hcol = 0.f; bool hcol_ok = false;
    int hrow_ly = 0, hcol_idx = 0;
    if (tid < 64) {
        const int hr = tid >> 4;                                   // 0, 1: rows -2, -1;  2, 3: rows TH, TH + 1
        hrow_ly = hr < 2 ? hr : TH + hr;                           // LDS row (tile row + 2)
        const OwnGroup g = own_group(sc, id.b, y0 + hrow_ly - 2, x0 + c);
        hrow = *(const f32x4_t*)(sc.pred + g.off); hrow_ok = g.ok;
    } else if (tid < 64 + 4 * (TH + 4)) {
        const int i = tid - 64, q = i & 3, j = q < 2 ? q - 2 : TW - 2 + q;       // columns -2, -1, TW, TW + 1
        const int gy = y0 + (i >> 2) - 2, gx = x0 + j;
        hcol_ok = (unsigned)gy < (unsigned)sc.H && (unsigned)gx < (unsigned)sc.W;
        hcol = sc.pred[(img + min(max(gy, 0), sc.H - 1)) * sc.W + min(max(gx, 0), sc.W - 1)];
        hcol_idx = (i >> 2) * LS + 4 + j;
    }
    // the single G pixel of this thread (threads 0..195): LDS position (row gl of the planes, column gj of the image tile) and its labels
    const bool gx_row = tid < 2 * TW;
    const int gl = gx_row ? (tid < TW ? 0 : TH + 1) : (tid - 2 * TW) >> 1;                    // plane row = tile row + 1
    const int gj = gx_row ? (tid & (TW - 1)) : ((tid - 2 * TW) & 1 ? TW : -1);
    const bool g_has = tid < 2 * TW + 2 * (TH + 2);
    float ge = 0.f, gn = 0.f; bool g_ok = false;
    if (g_has) {
        const int gy = y0 + gl - 1, gx = x0 + gj;
        g_ok = (unsigned)gy < (unsigned)sc.H && (unsigned)gx < (unsigned)sc.W;
        const long o = (img + min(max(gy, 0), sc.H - 1)) * sc.W + min(max(gx, 0), sc.W - 1);
        ge = sc.edge[o]; gn = sc.normal[o];
    }
    // ---- depth tile (the own rows' depth stays in registers for the chain rule)
    f32x4_t dep[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        dep[ps] = depth4(og[ps].ok, inv4[ps]);
        *(f32x4_t*)(sd + (r0 + 16 * ps + 2) * LS + 4 + c) = dep[ps];
    }
    if (tid < 64) *(f32x4_t*)(sd + hrow_ly * LS + 4 + c) = depth4(hrow_ok, hrow);
    else if (tid < 64 + 4 * (TH + 4)) sd[hcol_idx] = hcol_ok ? rcp_newton(fmaxf(hcol, 1e-6f)) : 0.f;
    __syncthreads();
    // ---- planes of G: own pixels, then the halo pixel
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        float w[3][6];
        window(sd, r0 + 16 * ps + 2, c, w);
        const f32x4_t e = sel4(og[ps].ok, e4[ps]);
        const float vm = og[ps].ok ? 1.f : 0.f;
        f32x4_t A, B;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float Ak, Bk;
            g_planes(w, k, n4[ps][k], cpos * e[k], cneg * (vm - e[k]), tk, Ak, Bk);
            A[k] = Ak; B[k] = Bk;
        }
        *(f32x4_t*)(sga + (r0 + 16 * ps + 1) * LS + 4 + c) = A;
        *(f32x4_t*)(sgb + (r0 + 16 * ps + 1) * LS + 4 + c) = B;
    }
    if (g_has) {
        float w[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) w[r][k] = sd[(gl + r) * LS + 4 + gj - 1 + k];          // plane row gl = depth rows gl .. gl + 2
        float Ak, Bk;
        const float e = g_ok ? ge : 0.f;
        g_planes(w, 0, gn, cpos * e, cneg * ((g_ok ? 1.f : 0.f) - e), tk, Ak, Bk);
        sga[gl * LS + 4 + gj] = Ak; sgb[gl * LS + 4 + gj] = Bk;
    }
    __syncthreads();
    // ---- transposed Sobel, chain rule through 1 / max(inv, 1e-6), silog gradient
    const float m1 = silog ? a.silog_aux[0] : 0.f, ksl = silog ? a.silog_aux[1] * (a.silog_gout ? a.silog_gout[0] : 1.f) : 0.f;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        float wa[3][6], wb[3][6];
        window(sga, r0 + 16 * ps + 1, c, wa);
        window(sgb, r0 + 16 * ps + 1, c, wb);
        float ca[6], eb[6], cc1[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            ca[j] = (wa[0][j] + wa[1][j]) + wa[2][j];
            eb[j] = wb[0][j] - wb[2][j];
            cc1[j] = wb[1][j] == 0.f ? wa[1][j] : 0.f;
        }
        f32x4_t out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d0 = wa[0][k + 1] == 0.f ? wb[0][k + 1] : 0.f, d2 = wa[2][k + 1] == 0.f ? wb[2][k + 1] : 0.f;
            const float dd = (ca[k] - ca[k + 2]) + ((eb[k] + eb[k + 1]) + eb[k + 2]) + (cc1[k] - cc1[k + 2]) + (d0 - d2);
            const float d = dep[ps][k];
            out[k] = inv4[ps][k] >= 1e-6f ? -dd * d * d : 0.f;
        }
        if (silog) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dgt = d4[ps][k];
                const float gt = rcpf(fmaxf(dgt, 1e-6f));
                const float pi = inv4[ps][k] + 1e-5f;
                const float dl = (__builtin_amdgcn_logf(pi * 10.f) - __builtin_amdgcn_logf(gt * 10.f)) * LN2;
                out[k] += dgt > 0.f ? ksl * (dl - 0.85f * m1) * rcpf(pi) : 0.f;
            }
        }
        if (og[ps].ok) *(f32x4_t*)(sc.dpred + og[ps].off) = out;
    }
}

// ---------------- silog (stand-alone: SupervisedLoss used without the edge loss) -------------------------------------
__global__ __launch_bounds__(256) void silog_fwd_kernel(const float* __restrict__ inv, const float* __restrict__ depth, long n, double* __restrict__ sums) {
    __shared__ double sred[4][3];
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = depth[i];
        if (d > 0.f) {
            const float g = 1.f / fmaxf(d, 1e-6f);
            const float dl = logf((inv[i] + 1e-5f) * 10.f) - logf(g * 10.f);
            s1 += dl; s2 += (double)dl * dl; cnt += 1.0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); cnt = wave_sum_d(cnt);
    if (lane == 0) { sred[wave][0] = s1; sred[wave][1] = s2; sred[wave][2] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&sums[threadIdx.x], sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}
// loss = 10 sqrt(E[d^2] - 0.85 E[d]^2);  aux = (mean, 10/sqrt(S)/n)
__global__ void silog_finalize_kernel(const double* __restrict__ sums, float out_scale, float* __restrict__ loss_acc, float* __restrict__ loss_this, float* __restrict__ aux) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double n = sums[2];
    const double m1 = sums[0] / n, m2 = sums[1] / n;
    const double S = m2 - 0.85 * m1 * m1;
    const float l = (float)(sqrt(S) * 10.0);
    if (loss_this) *loss_this = l;
    if (loss_acc) *loss_acc += out_scale * l;
    aux[0] = (float)m1;
    aux[1] = (float)(10.0 / sqrt(S) / n);
}
__global__ void silog_bwd_kernel(const float* __restrict__ inv, const float* __restrict__ depth, const float* __restrict__ aux,
                                 const float* __restrict__ gout, float* __restrict__ dinv, long n, int accumulate) {
    const float m1 = aux[0], k = aux[1] * (gout ? gout[0] : 1.f);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = depth[i];
        float gr = 0.f;
        if (d > 0.f) {
            const float g = 1.f / fmaxf(d, 1e-6f);
            const float pi = inv[i] + 1e-5f;
            const float dl = logf(pi * 10.f) - logf(g * 10.f);
            gr = k * (dl - 0.85f * m1) / pi;
        }
        dinv[i] = accumulate ? dinv[i] + gr : gr;
    }
}

// public scale record of include/mte_kernels.h
struct mte_edge_scale_t { const float* pred; const float* edge; const float* normal; const float* mask; float* gmap; float* dpred; int H, W; };

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// fills the grid geometry; returns the number of workgroups or -1
// every scale in the training configuration (see FAST)?
bool fast_config(const EdgeMulti& a) {
    if (!a.from_inv || !a.is_grad || !a.is_sigmoid) return false;
    for (int s = 0; s < a.nscales; ++s)
        if (!a.s[s].vec || !a.s[s].normal || a.s[s].mask || a.s[s].gmap) return false;
    return true;
}

int setup_scales(EdgeMulti& a, const mte_edge_scale_t* scales, int nscales, int B, bool backward) {
    const int T = backward ? 1 : FWD_TILES_PER_WG;
    if (!scales || nscales < 1 || nscales > MAXS || B < 1) return -1;
    int blocks = 0;
    for (int s = 0; s < nscales; ++s) {
        const mte_edge_scale_t& in = scales[s];
        if (!in.pred || !in.edge || in.H < 1 || in.W < 1 || (backward && !in.dpred)) return -1;
        EdgeScale& o = a.s[s];
        o.pred = in.pred; o.edge = in.edge; o.normal = in.normal; o.mask = in.mask; o.gmap = backward ? nullptr : in.gmap; o.dpred = in.dpred;
        o.H = in.H; o.W = in.W;
        o.tiles_x = (in.W + TW - 1) / TW; o.tiles_y = (in.H + TH - 1) / TH;
        o.first_block = blocks;
        o.vec = in.W % 4 == 0 && aligned16(in.pred) && aligned16(in.edge) && aligned16(in.normal) && aligned16(in.mask) &&
                aligned16(in.gmap) && aligned16(in.dpred);
        o.groups = (o.tiles_x * o.tiles_y + T - 1) / T;
        blocks += o.groups * B;
    }
    a.nscales = nscales; a.B = B; a.nblocks = blocks; a.tiles_per_wg = T;
    return blocks;
}
long results_elems(int nscales, int B) { return ((long)nscales * B * NP + 1) & ~1L; }
long counter_elems(int nscales, int B) { return (((long)nscales * B + 1) * 4 + 15) / 16 * 2; }      // doubles holding the tickets (16-byte multiple)

}  // namespace

extern "C" {

// doubles of workspace for a forward launch over these scales: [nscales][B][13] sums + the arrival tickets + one 16-double record per workgroup
long mte_edge_loss_work_elems(const void* scales, int nscales, int B) {
    EdgeMulti a{};
    a.fences = g_mte_handoff_fences;
    const int blocks = setup_scales(a, (const mte_edge_scale_t*)scales, nscales, B, false);
    if (blocks < 0) return -1;
    return results_elems(nscales, B) + counter_elems(nscales, B) + (long)blocks * REC;
}

// Forward of `nscales` (<= 4) depth-edge losses in ONE launch, the silog loss of scale 0 fused in when gt_depth != NULL.
// losses[s] <- weight * balanced BCE of scale s; coef [nscales][2B+1] <- backward coefficients; silog_loss / silog_aux[2].
int mte_edge_loss_multi_fwd(const void* scales, int nscales, int B, int from_inv, int is_grad, int is_sigmoid, float thresh,
                            float weight, float pos_to_neg, const float* gt_depth, double* work, float* losses, float* coef,
                            float* silog_loss, float* silog_aux, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    EdgeMulti a{};
    a.fences = g_mte_handoff_fences;
    const int blocks = setup_scales(a, (const mte_edge_scale_t*)scales, nscales, B, false);
    if (blocks < 0 || !work || !losses || !coef || (gt_depth && (!silog_loss || !silog_aux))) return MTE_ERR_ARG;
    if (gt_depth && !aligned16(gt_depth)) a.s[0].vec = 0;
    a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.finalize = 1; a.thresh = thresh;
    a.weight = weight; a.pos_to_neg = pos_to_neg; a.gt_depth = gt_depth; a.losses = losses; a.coef = coef;
    a.silog_loss = silog_loss; a.silog_aux = silog_aux;
    const long r = results_elems(nscales, B);
    a.results = work; a.counter = (unsigned*)(work + r); a.records = work + r + counter_elems(nscales, B);
    if (!g_mte_loss_prezeroed && mte_memset_async(work, 0, sizeof(double) * (r + counter_elems(nscales, B)), stream) != hipSuccess) return MTE_ERR_LAUNCH;     // sums + tickets (the records are overwritten)
    if (fast_config(a)) hipLaunchKernelGGL(edge_fwd_fast_kernel, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(edge_loss_fwd_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// Backward of the same: dpred of every scale <- gout[s] * d loss_s / d pred_s (+ silog_gout * d silog / d pred_0 on scale 0).
int mte_edge_loss_multi_bwd(const void* scales, int nscales, int B, int from_inv, int is_grad, int is_sigmoid, float thresh,
                            const float* coef, const float* gout, const float* gt_depth, const float* silog_aux, const float* silog_gout,
                            hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    EdgeMulti a{};
    a.fences = g_mte_handoff_fences;
    const int blocks = setup_scales(a, (const mte_edge_scale_t*)scales, nscales, B, true);
    if (blocks < 0 || !coef || (gt_depth && !silog_aux)) return MTE_ERR_ARG;
    if (gt_depth && !aligned16(gt_depth)) a.s[0].vec = 0;
    a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.thresh = thresh;
    a.coef = (float*)coef; a.gout = gout; a.gt_depth = gt_depth; a.silog_aux = (float*)silog_aux; a.silog_gout = silog_gout;
    if (fast_config(a)) hipLaunchKernelGGL(edge_bwd_fast_kernel, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(edge_loss_bwd_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// ---- single-scale entry points (GradLoss / GradLayer called directly): the same kernels with one scale
// doubles the caller must provide as `sums`: [B][13] sums (6 class-balance / BCE sums, 4 mask statistics, 3 unused) + the tickets + one 16-double record per workgroup
long mte_edge_loss_sums_elems(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return -1;
    const long tiles = (long)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    return results_elems(1, B) + counter_elems(1, B) + (long)B * ((tiles + FWD_TILES_PER_WG - 1) / FWD_TILES_PER_WG) * REC;
}

// Forward pass of one scale.  sums: mte_edge_loss_sums_elems(B, H, W) doubles (content on entry ignored).  gmap nullable.
int mte_edge_loss_fwd(const float* pred, const float* edge, const float* normal, const float* mask, double* sums, float* gmap,
                      int B, int H, int W, int from_inv, int is_grad, int is_sigmoid, float thresh, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!pred || !edge || !sums) return MTE_ERR_ARG;
    mte_edge_scale_t one{pred, edge, normal, mask, gmap, nullptr, H, W};
    EdgeMulti a{};
    a.fences = g_mte_handoff_fences;
    const int blocks = setup_scales(a, &one, 1, B, false);
    if (blocks < 0) return MTE_ERR_ARG;
    a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.finalize = 0; a.thresh = thresh;
    const long r = results_elems(1, B);
    a.results = sums; a.counter = (unsigned*)(sums + r); a.records = sums + r + counter_elems(1, B);
    if (!g_mte_loss_prezeroed && mte_memset_async(sums, 0, sizeof(double) * (r + counter_elems(1, B)), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    hipLaunchKernelGGL(edge_loss_fwd_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return mte_check_launch();
}
// loss_this (nullable) <- weight * balanced BCE;  *loss_acc (nullable) += out_scale * loss;  coef: [2B + 1] floats
int mte_edge_loss_finalize(const double* sums, int B, long numel, float weight, float pos_to_neg, int has_mask,
                           float out_scale, float* loss_acc, float* loss_this, float* coef, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!sums || !coef) return MTE_ERR_ARG;
    hipLaunchKernelGGL(edge_loss_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, B, numel, weight, pos_to_neg, has_mask, out_scale, loss_acc, loss_this, coef);
    return mte_check_launch();
}
// dpred <- gout * d loss / d pred  (gout: device scalar, nullable = 1)
int mte_edge_loss_bwd(const float* pred, const float* edge, const float* normal, const float* mask, const float* coef, const float* gout,
                      float* dpred, int B, int H, int W, int from_inv, int is_grad, int is_sigmoid, float thresh, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!pred || !edge || !coef || !dpred) return MTE_ERR_ARG;
    mte_edge_scale_t one{pred, edge, normal, mask, nullptr, dpred, H, W};
    return mte_edge_loss_multi_bwd(&one, 1, B, from_inv, is_grad, is_sigmoid, thresh, coef, gout, nullptr, nullptr, nullptr, stream);
}

// sums[3] doubles (zeroed here); aux[2] floats
int mte_silog_fwd(const float* inv, const float* depth, long n, double* sums, float out_scale, float* loss_acc, float* loss_this, float* aux, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!inv || !depth || !sums || !aux) return MTE_ERR_ARG;
    if (mte_memset_async(sums, 0, sizeof(double) * 3, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    long g = (n + 255) / 256; if (g > 1024) g = 1024; if (g < 1) g = 1;
    hipLaunchKernelGGL(silog_fwd_kernel, dim3((unsigned)g), dim3(256), 0, stream, inv, depth, n, sums);
    hipLaunchKernelGGL(silog_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, out_scale, loss_acc, loss_this, aux);
    return mte_check_launch();
}
int mte_silog_bwd(const float* inv, const float* depth, const float* aux, const float* gout, float* dinv, long n, int accumulate, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!inv || !depth || !aux || !dinv) return MTE_ERR_ARG;
    long g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
    hipLaunchKernelGGL(silog_bwd_kernel, dim3((unsigned)g), dim3(256), 0, stream, inv, depth, aux, gout, dinv, n, accumulate);
    return mte_check_launch();
}

}  // extern "C"
