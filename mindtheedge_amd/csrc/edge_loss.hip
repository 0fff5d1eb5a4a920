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
//   * sums accumulate in fp32 per thread and per wave, are combined in fp64 per workgroup and leave it as one fp64 atomic
//     per value into the accumulators of its (scale, sample): <= 240 adds per address, spread over the launch (the
//     mask statistics are skipped when there is no mask).  A fixed-order reduction of per-workgroup partials by the last
//     workgroup was measured first: its serial sweep over 2,568 partial records took ~120 us, six times the stencil;
//   * the LAST workgroup to arrive (one agent-scope release per workgroup, one ticket, one acquire in the last one:
//     cdna_hip_programming.md guideline 16 / in-launch split-K recipe) computes alpha, the loss scalars and the backward
//     coefficients on the device: no reduce / finalize launches, no host sync.
// HBM-bound: 12 B/pixel forward (inv, edge, normal), 16 B/pixel backward (+4 B gradient write).
#include "common.hpp"

namespace {

constexpr int TW = 64, TH = 32;         // output tile (256 threads x 2 passes x 4 pixels)
constexpr int LS = 72;                  // LDS row stride in floats: image column j of the tile sits at index 4 + j (16-byte aligned interior)
constexpr int MAXS = 4;                 // scales per launch
constexpr int NP = 13;                  // partial sums per workgroup: 6 edge sums, 4 mask statistics, 3 silog sums

struct EdgeScale {
    const float* pred;                  // inv-depth (from_inv), depth, or probability map
    const float* edge; const float* normal; const float* mask;    // normal / mask nullable
    float* gmap;                        // forward: optional edge-strength map output
    float* dpred;                       // backward output
    int H, W, tiles_x, tiles_y, first_block, vec;                // vec: 16-byte accesses are legal (W % 4 == 0, aligned bases)
};

struct EdgeMulti {
    EdgeScale s[MAXS];
    int nscales, B, nblocks;
    int from_inv, is_grad, is_sigmoid, finalize;
    float thresh, weight, pos_to_neg;
    double* results;                    // [nscales][B][NP] accumulators (zeroed by the launcher)
    unsigned* counter;                  // [0] launch ticket, [1 + s*B + b] ticket of (scale, sample) (zeroed by the launcher)
    float* losses;                      // forward out: [nscales]
    float* coef;                        // forward out / backward in: [nscales][2B + 1]
    const float* gout;                  // backward: upstream gradient per scale loss (device, nullable = 1)
    const float* gt_depth;              // optional fused silog on scale 0: metric depth, 0 = invalid
    float* silog_loss; float* silog_aux;          // forward out: loss, (mean, 10/sqrt(S)/n)
    const float* silog_gout;            // backward: upstream gradient of the silog loss (device, nullable = 1)
};

__device__ __forceinline__ int direction_code(float n) {
    // thresholds = float32(k*pi/8), half-open bins, later assignments win (grad_loss.py:80-93)
    const float P1 = (float)(1 * 3.14159265358979323846 / 8), P3 = (float)(3 * 3.14159265358979323846 / 8),
                P5 = (float)(5 * 3.14159265358979323846 / 8), P7 = (float)(7 * 3.14159265358979323846 / 8);
    int code = 0;                                                   // 0: h
    if ((n >= -P5 && n < -P3) || (n >= P3 && n < P5)) code = 1;     // v
    if ((n >= -P7 && n < -P5) || (n >= P1 && n < P3)) code = 2;     // rl
    if ((n >= -P3 && n < -P1) || (n >= P5 && n < P7)) code = 3;     // lr
    return code;
}

__device__ __forceinline__ float to_depth(int from_inv, float v) { return from_inv ? 1.f / fmaxf(v, 1e-6f) : v; }

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
__device__ __forceinline__ float sigmoidf(float x) { return rcpf(1.f + __expf(-x)); }

struct BlockId { int s, b, x0, y0; };
__device__ __forceinline__ BlockId decode_block(const EdgeMulti& a) {
    BlockId id;
    int s = 0;
#pragma unroll
    for (int k = 1; k < MAXS; ++k) if (k < a.nscales && (int)blockIdx.x >= a.s[k].first_block) s = k;
    int r = blockIdx.x - a.s[s].first_block;
    const int tx = r % a.s[s].tiles_x; r /= a.s[s].tiles_x;
    const int ty = r % a.s[s].tiles_y;
    id.s = s; id.b = r / a.s[s].tiles_y; id.x0 = tx * TW; id.y0 = ty * TH;
    return id;
}

// ---------------- forward -----------------------------------------------------------------------------------------
__device__ void finalize_losses(const EdgeMulti& a);

// FAST = the training configuration on every scale (16-byte accesses legal, inverse depth in, Sobel + normals + sigmoid, no mask):
// the flags are compile-time there.  The generic instantiation carries every runtime flag -- its body is ~10k instructions
// (scalar and vector access paths, magnitude / probability / mask branches), several times what the instruction cache likes.
template <bool FAST>
__global__ __launch_bounds__(256, 4) void edge_loss_fwd_kernel(EdgeMulti a) {
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

    // every global load of the workgroup first: one memory latency for the tile, the labels and the normals together
    DepthTile<1> tile;
    if (is_grad) tile.issue(sc, vec, id.b, id.x0, id.y0);
    f32x4_t e4[2], n4[2], m4[2], d4[2], i4[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int gy = id.y0 + r0 + 16 * ps;
        const bool ok = gy < sc.H;
        e4[ps] = ok ? load4(sc.edge, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (has_normal && is_grad) n4[ps] = ok ? load4(sc.normal, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (has_mask) m4[ps] = ok ? load4(sc.mask, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (silog) {
            d4[ps] = ok ? load4(a.gt_depth, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            i4[ps] = ok ? load4(sc.pred, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        if (!is_grad) i4[ps] = ok ? load4(sc.pred, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    if (is_grad) tile.commit(sc, from_inv, id.x0, id.y0, sd);
    __syncthreads();

    float acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.f;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int ly = r0 + 16 * ps, gy = id.y0 + ly;
        if (gy >= sc.H) continue;
        float w[3][6];
        if (is_grad) window(sd, ly + 1, c, w);
        f32x4_t g4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (id.x0 + c + k >= sc.W) { g4[k] = 0.f; continue; }
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
            if (is_sigmoid) { const float t = __expf(-(g - a.thresh)); p = rcpf(1.f + t); omp = t * p; }
            else { p = g; omp = 1.f - g; }
            const float e = e4[ps][k];
            const float pos = -e * __logf(p + 0.001f), neg = -(1.f - e) * __logf(omp + 0.001f);
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
                    const float gt = 1.f / fmaxf(d, 1e-6f);
                    const float dl = __logf((i4[ps][k] + 1e-5f) * 10.f) - __logf(gt * 10.f);
                    acc[10] += dl; acc[11] = fmaf(dl, dl, acc[11]); acc[12] += 1.f;
                }
            }
        }
        if (!FAST && sc.gmap) store4(sc.gmap, img + gy, id.x0 + c, sc.W, vec, g4);
    }
    // fp32 within the wave, fp64 across the waves and across workgroups
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i >= 4 && i < 10 && !has_mask) continue;
        if (i >= 10 && !silog) continue;
        const float s = wave_sum(acc[i]);
        if (lane == 0) sred[wave][i] = s;
    }
    __syncthreads();
    if (tid < NP) {
        const bool live = !((tid >= 4 && tid < 10 && !has_mask) || (tid >= 10 && !silog));
        if (live) {
            // RETURNING atomic: the value comes back only after the add has been performed at the memory side, so once this wave
            // has its results every add of this workgroup is globally visible (a no-return add is acknowledged earlier: with it
            // the ticket below was seen to overtake an add about once per few thousand workgroups)
            const double before = atomicAdd(&a.results[((long)id.s * a.B + id.b) * NP + tid],
                                            (double)sred[0][tid] + (double)sred[1][tid] + (double)sred[2][tid] + (double)sred[3][tid]);
            asm volatile("" ::"v"(before));
        }
    }
    // ---- the last workgroup to arrive finishes the losses.  The payload is the fp64 atomics above: device-scope, performed at
    //      the memory side (nothing of it sits in this CU's L1 / this XCD's L2), so there is nothing for a write-back fence to
    //      write back -- the adding wave waits for its returned values (vmcnt), the workgroup meets at a barrier, then one lane
    //      draws a ticket, and the last workgroup reads the accumulators with agent-scope loads.  Two levels of tickets -- per (scale, sample), then one per
    //      launch -- because 2,568 workgroups on ONE word serialise at ~88 tickets/us (29 us, longer than the stencil itself).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        int last = 0;
        const unsigned per_image = (unsigned)(sc.tiles_x * sc.tiles_y);
        const unsigned old = __hip_atomic_fetch_add(a.counter + 1 + id.s * a.B + id.b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == per_image - 1) {
            const unsigned old2 = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old2 == (unsigned)(a.nscales * a.B - 1);
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    finalize_losses(a);
}

// Runs in the last workgroup: every accumulator is complete.  Thread 0 does the scalar arithmetic of comp_cross_entropy
// (grad_loss.py:161-219) and SilogLoss (supervised_loss.py:57-69) once per scale.
__device__ __forceinline__ double acc_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ void finalize_losses(const EdgeMulti& a) {
    const int tid = threadIdx.x;
    if (tid != 0 || !a.finalize) return;
    for (int s = 0; s < a.nscales; ++s) {
        const double* R = a.results + (long)s * a.B * NP;
        double mi[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = 0; b < a.B; ++b)
            for (int k = 0; k < 4; ++k) mi[k] += acc_load(R + b * NP + 6 + k);
        const bool has_mask = a.s[s].mask != nullptr;
        const bool binary = has_mask && mi[2] == 0.0 && mi[0] > 0.0 && mi[1] > 0.0;      // unique(mask) == {0, 1}
        const double nvalid = binary ? mi[3] : (double)a.B * a.s[s].H * a.s[s].W;
        double wneg_total = 0.0;
        for (int b = 0; b < a.B; ++b) wneg_total += (double)(float)acc_load(R + b * NP + 1);
        double total = 0.0;
        float* coef = a.coef + (long)s * (2 * a.B + 1);
        for (int b = 0; b < a.B; ++b) {
            const double* sums = R + b * NP;
            const float wp = (float)acc_load(sums), wn = (float)acc_load(sums + 1);
            const float alpha = wneg_total == 0.0 ? 1.f : wn / (wp + wn);
            const double P = binary ? acc_load(sums + 4) : acc_load(sums + 2), N = binary ? acc_load(sums + 5) : acc_load(sums + 3);
            total += (double)a.pos_to_neg * alpha * P + (double)(1.f - alpha) * N;
            coef[2 * b] = (float)((double)a.weight * a.pos_to_neg * alpha / nvalid);
            coef[2 * b + 1] = (float)((double)a.weight * (1.f - alpha) / nvalid);
        }
        coef[2 * a.B] = binary ? 1.f : 0.f;
        a.losses[s] = (float)((double)a.weight * total / nvalid);
    }
    if (a.gt_depth) {                                             // loss = 10 sqrt(E[d^2] - 0.85 E[d]^2);  aux = (mean, 10/sqrt(S)/n)
        double ss[3] = {0.0, 0.0, 0.0};
        for (int b = 0; b < a.B; ++b)
            for (int k = 0; k < 3; ++k) ss[k] += acc_load(a.results + (long)b * NP + 10 + k);
        const double n = ss[2];
        const double m1 = ss[0] / n, m2 = ss[1] / n;
        const double S = m2 - 0.85 * m1 * m1;
        if (a.silog_loss) *a.silog_loss = (float)(sqrt(S) * 10.0);
        if (a.silog_aux) { a.silog_aux[0] = (float)m1; a.silog_aux[1] = (float)(10.0 / sqrt(S) / n); }
    }
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
// transposed Sobel weights: d s_code(p) / d depth(p + t) for tap t = (dy, dx):  h: dx (2 - |dy|), v: dy (2 - |dx|), rl: dx - dy, lr: dx + dy
__device__ __forceinline__ float tap_weight(int code, int dy, int dx) {
    const float kh = (float)(dx * (2 - (dy < 0 ? -dy : dy))), kv = (float)(dy * (2 - (dx < 0 ? -dx : dx)));
    const float krl = (float)(dx - dy), klr = (float)(dx + dy);
    return code == 0 ? kh : (code == 1 ? kv : (code == 2 ? krl : klr));
}

template <bool FAST>
__global__ __launch_bounds__(256) void edge_loss_bwd_kernel(EdgeMulti a) {
    // depth on the tile + 2-pixel halo; G = d loss / d s(p) (and the direction code) on the tile + 1-pixel halo
    __shared__ __attribute__((aligned(16))) float sd[(TH + 4) * LS];
    __shared__ __attribute__((aligned(16))) float sga[(TH + 2) * LS], sgb[(TH + 2) * LS];
    __shared__ __attribute__((aligned(16))) int scode[(TH + 2) * LS];
    const BlockId id = decode_block(a);
    const EdgeScale& sc = a.s[id.s];
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
        tile.issue(sc, vec, id.b, id.x0, id.y0);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = tid + k * 256;
            const int ly = i / (TW / 4 + 2), q = i % (TW / 4 + 2);               // q < 16: interior group, 16 / 17: left / right halo column
            const int gy = id.y0 + ly - 1;
            const bool group = q < TW / 4;
            const int j0 = group ? q * 4 : (q == TW / 4 ? -1 : TW);
            ge[k] = f32x4_t{0.f, 0.f, 0.f, 0.f}; gn[k] = ge[k]; gm[k] = f32x4_t{1.f, 1.f, 1.f, 1.f};
            if (i < GITEMS && (unsigned)gy < (unsigned)sc.H) {
                if (group) {
                    ge[k] = load4(sc.edge, img + gy, id.x0 + j0, sc.W, vec);
                    if (!magnitude) gn[k] = load4(sc.normal, img + gy, id.x0 + j0, sc.W, vec);
                    if (use_keep) gm[k] = load4(sc.mask, img + gy, id.x0 + j0, sc.W, vec);
                } else if ((unsigned)(id.x0 + j0) < (unsigned)sc.W) {
                    const long idx = (img + gy) * sc.W + id.x0 + j0;
                    ge[k][0] = sc.edge[idx];
                    if (!magnitude) gn[k][0] = sc.normal[idx];
                    if (use_keep) gm[k][0] = sc.mask[idx];
                }
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int gy = id.y0 + r0 + 16 * ps;
        const bool ok = gy < sc.H && id.x0 + c < sc.W;
        const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
        inv4[ps] = ok ? load4(sc.pred, img + gy, id.x0 + c, sc.W, vec) : z;
        if (silog) d4[ps] = ok ? load4(a.gt_depth, img + gy, id.x0 + c, sc.W, vec) : z;
        if (!is_grad) {
            oe4[ps] = ok ? load4(sc.edge, img + gy, id.x0 + c, sc.W, vec) : z;
            om4[ps] = (ok && use_keep) ? load4(sc.mask, img + gy, id.x0 + c, sc.W, vec) : f32x4_t{1.f, 1.f, 1.f, 1.f};
        }
    }
    if (is_grad) {
        tile.commit(sc, from_inv, id.x0, id.y0, sd);
        __syncthreads();
#pragma unroll
        for (int kq = 0; kq < NG; ++kq) {
            const int i = tid + kq * 256;
            if (i >= GITEMS) break;
            const int ly = i / (TW / 4 + 2), q = i % (TW / 4 + 2);
            const int gy = id.y0 + ly - 1;
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
                const int gx = id.x0 + j0 + k;
                float ga = 0.f, gb = 0.f; int code = 0;
                if (rowok && (unsigned)gx < (unsigned)sc.W) {
                    float sh, sv, srl, slr;
                    sobel4(w, k, sh, sv, srl, slr);
                    float g, da, db = 0.f;        // d g / d s_a, d g / d s_b
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
                    if (is_sigmoid) { const float t = __expf(-(g - a.thresh)); p = rcpf(1.f + t); omp = t * p; }
                    else { p = g; omp = 1.f - g; }
                    const float e = e4[k];
                    const float keep = (use_keep && m4[k] == 0.f) ? 0.f : 1.f;
                    const float dp = is_sigmoid ? p * omp : 1.f;
                    const float dg = keep * dp * (-cpos * e * rcpf(p + 0.001f) + cneg * (1.f - e) * rcpf(omp + 0.001f));
                    ga = dg * da; gb = dg * db;
                }
                const int o = ly * LS + 4 + j0 + k;
                sga[o] = ga; scode[o] = code;
                if (magnitude) sgb[o] = gb;
            }
        }
        __syncthreads();
    }
    const float m1 = silog ? a.silog_aux[0] : 0.f, ksl = silog ? a.silog_aux[1] * (a.silog_gout ? a.silog_gout[0] : 1.f) : 0.f;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int ly = r0 + 16 * ps, gy = id.y0 + ly;
        if (gy >= sc.H || id.x0 + c >= sc.W) continue;
        f32x4_t out = {0.f, 0.f, 0.f, 0.f};
        if (is_grad) {
            // d loss / d depth(q) = sum_p G(p) K_code(p)[q - p]: the 3 x 6 windows of G and code around the 4 pixels
            float wa[3][6], wb[3][6]; int wc[3][6];
            window(sga, ly + 1, c, wa);
            if (magnitude) window(sgb, ly + 1, c, wb);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int* row = scode + (ly + r) * LS + 4 + c;
                wc[r][0] = row[-1]; wc[r][1] = row[0]; wc[r][2] = row[1]; wc[r][3] = row[2]; wc[r][4] = row[3]; wc[r][5] = row[4];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float dd = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        // source pixel p at window (r, k + cc); q - p = (dy, dx) = (1 - r, 1 - cc)
                        const int dy = 1 - r, dx = 1 - cc;
                        if (dy == 0 && dx == 0) continue;
                        if (magnitude) dd += tap_weight(1, dy, dx) * wa[r][k + cc] + tap_weight(0, dy, dx) * wb[r][k + cc];
                        else dd += tap_weight(wc[r][k + cc], dy, dx) * wa[r][k + cc];
                    }
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
                if (is_sigmoid) { const float t = __expf(-(g - a.thresh)); p = rcpf(1.f + t); omp = t * p; }
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
                    const float gt = 1.f / fmaxf(d, 1e-6f);
                    const float pi = inv4[ps][k] + 1e-5f;
                    const float dl = __logf(pi * 10.f) - __logf(gt * 10.f);
                    out[k] += ksl * (dl - 0.85f * m1) / pi;
                }
            }
        }
        store4(sc.dpred, img + gy, id.x0 + c, sc.W, vec, out);
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
        blocks += o.tiles_x * o.tiles_y * B;
    }
    a.nscales = nscales; a.B = B; a.nblocks = blocks;
    return blocks;
}
long results_elems(int nscales, int B) { return ((long)nscales * B * NP + 1) & ~1L; }
long counter_elems(int nscales, int B) { return (((long)nscales * B + 1) * 4 + 15) / 16 * 2; }      // doubles holding the tickets (16-byte multiple)

}  // namespace

extern "C" {

// doubles of workspace for a forward launch over these scales: [nscales][B][13] accumulators + the arrival ticket
long mte_edge_loss_work_elems(const void* scales, int nscales, int B) {
    EdgeMulti a{};
    const int blocks = setup_scales(a, (const mte_edge_scale_t*)scales, nscales, B, false);
    if (blocks < 0) return -1;
    return results_elems(nscales, B) + counter_elems(nscales, B);
}

// Forward of `nscales` (<= 4) depth-edge losses in ONE launch, the silog loss of scale 0 fused in when gt_depth != NULL.
// losses[s] <- weight * balanced BCE of scale s; coef [nscales][2B+1] <- backward coefficients; silog_loss / silog_aux[2].
int mte_edge_loss_multi_fwd(const void* scales, int nscales, int B, int from_inv, int is_grad, int is_sigmoid, float thresh,
                            float weight, float pos_to_neg, const float* gt_depth, double* work, float* losses, float* coef,
                            float* silog_loss, float* silog_aux, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    EdgeMulti a{};
    const int blocks = setup_scales(a, (const mte_edge_scale_t*)scales, nscales, B, false);
    if (blocks < 0 || !work || !losses || !coef || (gt_depth && (!silog_loss || !silog_aux))) return MTE_ERR_ARG;
    if (gt_depth && !aligned16(gt_depth)) a.s[0].vec = 0;
    a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.finalize = 1; a.thresh = thresh;
    a.weight = weight; a.pos_to_neg = pos_to_neg; a.gt_depth = gt_depth; a.losses = losses; a.coef = coef;
    a.silog_loss = silog_loss; a.silog_aux = silog_aux;
    const long r = results_elems(nscales, B);
    a.results = work; a.counter = (unsigned*)(work + r);
    if (mte_memset_async(work, 0, sizeof(double) * (r + counter_elems(nscales, B)), stream) != hipSuccess) return MTE_ERR_LAUNCH;     // accumulators + tickets
    if (fast_config(a)) hipLaunchKernelGGL(edge_loss_fwd_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(edge_loss_fwd_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// Backward of the same: dpred of every scale <- gout[s] * d loss_s / d pred_s (+ silog_gout * d silog / d pred_0 on scale 0).
int mte_edge_loss_multi_bwd(const void* scales, int nscales, int B, int from_inv, int is_grad, int is_sigmoid, float thresh,
                            const float* coef, const float* gout, const float* gt_depth, const float* silog_aux, const float* silog_gout,
                            hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    EdgeMulti a{};
    const int blocks = setup_scales(a, (const mte_edge_scale_t*)scales, nscales, B, true);
    if (blocks < 0 || !coef || (gt_depth && !silog_aux)) return MTE_ERR_ARG;
    if (gt_depth && !aligned16(gt_depth)) a.s[0].vec = 0;
    a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.thresh = thresh;
    a.coef = (float*)coef; a.gout = gout; a.gt_depth = gt_depth; a.silog_aux = (float*)silog_aux; a.silog_gout = silog_gout;
    if (fast_config(a)) hipLaunchKernelGGL(edge_loss_bwd_kernel<true>, dim3(blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(edge_loss_bwd_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// ---- single-scale entry points (GradLoss / GradLayer called directly): the same kernels with one scale
// doubles the caller must provide as `sums`: [B][13] accumulators (6 class-balance / BCE sums, 4 mask statistics, 3 unused) + the ticket
long mte_edge_loss_sums_elems(int B, int H, int W) {
    (void)H; (void)W;
    return results_elems(1, B) + counter_elems(1, B);
}

// Forward pass of one scale.  sums: mte_edge_loss_sums_elems(B, H, W) doubles (content on entry ignored).  gmap nullable.
int mte_edge_loss_fwd(const float* pred, const float* edge, const float* normal, const float* mask, double* sums, float* gmap,
                      int B, int H, int W, int from_inv, int is_grad, int is_sigmoid, float thresh, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!pred || !edge || !sums) return MTE_ERR_ARG;
    mte_edge_scale_t one{pred, edge, normal, mask, gmap, nullptr, H, W};
    EdgeMulti a{};
    const int blocks = setup_scales(a, &one, 1, B, false);
    if (blocks < 0) return MTE_ERR_ARG;
    a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.finalize = 0; a.thresh = thresh;
    const long r = results_elems(1, B);
    a.results = sums; a.counter = (unsigned*)(sums + r);
    if (mte_memset_async(sums, 0, sizeof(double) * (r + counter_elems(1, B)), stream) != hipSuccess) return MTE_ERR_LAUNCH;
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
