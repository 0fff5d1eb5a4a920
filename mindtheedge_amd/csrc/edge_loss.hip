// Fused depth-edge loss and silog loss for gfx950 (fp32 NCHW maps with C = 1, i.e. plain [B,H,W]).
//
// Edge loss = GradLoss.forward with edge_loss_type 'cross_entropy' (packnet_sfm/losses/grad_loss.py:122-219)
// fused with inv2depth (utils/depth.py:104-121) and GradLayer (grad_loss.py:20-31,65-95):
//   depth = 1/max(inv,1e-6) -> 4 Sobel responses (v,h,lr,rl; zero pad) -> per-pixel direction chosen by the
//   edge-normal angle (or sqrt(v^2+h^2+1e-6) without normals) -> p = sigmoid(g - thresh)
//   -> pos = -e log(p+1e-3), neg = -(1-e) log(1-p+1e-3) -> per-sample class-balance alpha -> weighted mean.
// The reference issues 4 conv2d + ~20 masking kernels + torch.unique (host syncs) per scale; here one
// LDS-tiled stencil pass produces the per-sample sums, a one-block kernel turns them into the loss scalar
// and the backward coefficients (no host sync), and one stencil pass produces d loss / d inv.
// HBM-bound: 12 B/pixel forward (inv, edge, normal), 16 B/pixel backward (+4 B gradient write).
//
// Silog = SupervisedLoss 'sparse-silog', one scale (losses/supervised_loss.py:57-69,155-216).
#include "common.hpp"

namespace {

constexpr int TX = 64, TY = 4;          // output tile (256 threads, one pixel each)

struct EdgeArgs {
    const float* pred;                  // inv-depth (from_inv), depth, or probability map
    const float* edge; const float* normal; const float* mask;    // normal/mask nullable
    double* sums;                       // [B][6]: w_pos, w_neg, pos_all, neg_all, pos_keep, neg_keep ; then [4] mask info
    float* gmap;                        // optional edge-strength map output
    const float* coef;                  // backward: [B][2] + [1] use_keep flag   (from finalize)
    const float* gout;                  // upstream gradient scalar (device)
    float* dpred;                       // backward output
    int B, H, W;
    int from_inv, is_grad, is_sigmoid;
    float thresh;
};

__constant__ float c_sobel[4][9] = {
    {-1, 0, 1, -2, 0, 2, -1, 0, 1},      // code 0: h
    {-1, -2, -1, 0, 0, 0, 1, 2, 1},      // code 1: v
    {0, 1, 2, -1, 0, 1, -2, -1, 0},      // code 2: rl
    {-2, -1, 0, -1, 0, 1, 0, 1, 2},      // code 3: lr
};

__device__ __forceinline__ int direction_code(float n) {
    // thresholds = float32(k*pi/8), half-open bins, later assignments win (grad_loss.py:80-93)
    const float P1 = (float)(1 * 3.14159265358979323846 / 8), P3 = (float)(3 * 3.14159265358979323846 / 8),
                P5 = (float)(5 * 3.14159265358979323846 / 8), P7 = (float)(7 * 3.14159265358979323846 / 8);
    int code = 0;
    if ((n >= -P5 && n < -P3) || (n >= P3 && n < P5)) code = 1;
    if ((n >= -P7 && n < -P5) || (n >= P1 && n < P3)) code = 2;
    if ((n >= -P3 && n < -P1) || (n >= P5 && n < P7)) code = 3;
    return code;
}

__device__ __forceinline__ float to_depth(const EdgeArgs& a, float v) { return a.from_inv ? 1.f / fmaxf(v, 1e-6f) : v; }

// loads the (TY+2R) x (TX+2R) depth tile around the block's output tile; zero outside the image
template <int R>
__device__ __forceinline__ void load_depth_tile(const EdgeArgs& a, int b, int x0, int y0, float* sd) {
    constexpr int PW = TX + 2 * R, PH = TY + 2 * R;
    for (int i = threadIdx.x; i < PW * PH; i += 256) {
        const int ly = i / PW, lx = i % PW;
        const int gy = y0 + ly - R, gx = x0 + lx - R;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) v = to_depth(a, a.pred[((long)b * a.H + gy) * a.W + gx]);
        sd[i] = v;
    }
}

template <int PW>
__device__ __forceinline__ void sobel4(const float* sd, int ly, int lx, float& sh, float& sv, float& srl, float& slr) {
    float n[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) n[t] = sd[(ly + t / 3 - 1) * PW + lx + t % 3 - 1];
    sh = (n[2] - n[0]) + 2.f * (n[5] - n[3]) + (n[8] - n[6]);
    sv = (n[6] - n[0]) + 2.f * (n[7] - n[1]) + (n[8] - n[2]);
    srl = (n[1] - n[3]) + 2.f * (n[2] - n[6]) + (n[5] - n[7]);
    slr = (n[5] - n[1]) + 2.f * (n[8] - n[0]) + (n[7] - n[3]);
}

constexpr int FWD_TILES = 2;            // output tiles per block (stacked in y)

__global__ __launch_bounds__(256) void edge_loss_fwd_kernel(EdgeArgs a) {
    __shared__ float sd[(TY + 2) * (TX + 2)];
    __shared__ double sred[4][10];
    const int b = blockIdx.z, x0 = blockIdx.x * TX;
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    const int gx = x0 + lx;
    double acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0;
    for (int t = 0; t < FWD_TILES; ++t) {
        const int y0 = (blockIdx.y * FWD_TILES + t) * TY;
        if (y0 >= a.H) break;
        if (a.is_grad) { __syncthreads(); load_depth_tile<1>(a, b, x0, y0, sd); __syncthreads(); }
        const int gy = y0 + ly;
        if (gx < a.W && gy < a.H) {
            const long idx = ((long)b * a.H + gy) * a.W + gx;
            float g;
            if (a.is_grad) {
                float sh, sv, srl, slr;
                sobel4<TX + 2>(sd, ly + 1, lx + 1, sh, sv, srl, slr);
                if (a.normal) {
                    const int code = direction_code(a.normal[idx]);
                    g = fabsf(code == 0 ? sh : (code == 1 ? sv : (code == 2 ? srl : slr)));
                } else {
                    g = sqrtf(sv * sv + sh * sh + 1e-6f);
                }
            } else {
                g = a.pred[idx];
            }
            if (a.gmap) a.gmap[idx] = g;
            const float p = a.is_sigmoid ? 1.f / (1.f + expf(-(g - a.thresh))) : g;
            const float e = a.edge[idx];
            const float m = a.mask ? a.mask[idx] : 1.f;
            const float pos = -e * logf(p + 0.001f), neg = -(1.f - e) * logf(1.f - p + 0.001f);
            const float keep = m != 0.f ? 1.f : 0.f;
            acc[0] += (double)(e * m); acc[1] += (double)((1.f - e) * m); acc[2] += (double)pos; acc[3] += (double)neg;
            acc[4] += (double)(pos * keep); acc[5] += (double)(neg * keep);
            acc[6] += m == 0.f ? 1.0 : 0.0; acc[7] += m == 1.f ? 1.0 : 0.0; acc[8] += (m != 0.f && m != 1.f) ? 1.0 : 0.0; acc[9] += (double)m;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const double s = wave_sum_d(acc[i]);
        if (lane == 0) sred[wave][i] = s;
    }
    __syncthreads();
    // no atomics: thousands of workgroups adding into the same 6 + 4 words serialise (~8 ns each), so every workgroup stores
    // its ten partial sums and edge_sums_reduce_kernel adds them up in a fixed order (deterministic as a bonus)
    if (threadIdx.x < 10) {
        const long blk = ((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.sums[(long)a.B * 6 + 4 + blk * 10 + threadIdx.x] = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
    }
}

// sums[o] for o < 6B: per-image sums of partial k = o % 6; the last 4: mask statistics over the whole launch
__global__ __launch_bounds__(256) void edge_sums_reduce_kernel(double* __restrict__ sums, int B, int per_image) {
    __shared__ double sred[4];
    const int o = blockIdx.x;
    const double* part = sums + (long)B * 6 + 4;
    long first, count;
    int k;
    if (o < B * 6) { first = (long)(o / 6) * per_image; count = per_image; k = o % 6; }
    else { first = 0; count = (long)B * per_image; k = 6 + (o - B * 6); }
    double acc = 0.0;
    for (long t = threadIdx.x; t < count; t += 256) acc += part[(first + t) * 10 + k];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[o] = sred[0] + sred[1] + sred[2] + sred[3];
}

// one block: sums -> loss (accumulated into *loss_acc with factor out_scale) and backward coefficients
__global__ void edge_loss_finalize_kernel(const double* __restrict__ sums, int B, long numel, float weight, float pos_to_neg,
                                          int has_mask, float out_scale, float* __restrict__ loss_acc, float* __restrict__ loss_this,
                                          float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double* mi = sums + (long)B * 6;
    const bool binary = has_mask && mi[2] == 0.0 && mi[0] > 0.0 && mi[1] > 0.0;      // unique(mask) == {0, 1}
    const double nvalid = binary ? mi[3] : (double)numel;
    double wneg_total = 0.0;
    for (int b = 0; b < B; ++b) wneg_total += (double)(float)sums[b * 6 + 1];
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        const float wp = (float)sums[b * 6], wn = (float)sums[b * 6 + 1];
        const float alpha = wneg_total == 0.0 ? 1.f : wn / (wp + wn);
        const double P = binary ? sums[b * 6 + 4] : sums[b * 6 + 2], N = binary ? sums[b * 6 + 5] : sums[b * 6 + 3];
        total += (double)pos_to_neg * alpha * P + (double)(1.f - alpha) * N;
        coef[2 * b] = (float)((double)weight * pos_to_neg * alpha / nvalid);
        coef[2 * b + 1] = (float)((double)weight * (1.f - alpha) / nvalid);
    }
    coef[2 * B] = binary ? 1.f : 0.f;
    const float l = (float)((double)weight * total / nvalid);
    if (loss_this) *loss_this = l;
    if (loss_acc) *loss_acc += out_scale * l;
}

__global__ __launch_bounds__(256) void edge_loss_bwd_kernel(EdgeArgs a) {
    // G tile: (TY+2) x (TX+2) pixels p around the output tile: contribution weights + direction code
    __shared__ float sd[(TY + 4) * (TX + 4)];
    __shared__ float sga[(TY + 2) * (TX + 2)], sgb[(TY + 2) * (TX + 2)];
    __shared__ int scode[(TY + 2) * (TX + 2)];
    const int b = blockIdx.z, x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const float go = a.gout ? a.gout[0] : 1.f;
    const float cpos = a.coef[2 * b] * go, cneg = a.coef[2 * b + 1] * go;
    const bool use_keep = a.coef[2 * a.B] != 0.f;
    if (!a.is_grad) {
        const int lx = threadIdx.x % TX, ly = threadIdx.x / TX, gx = x0 + lx, gy = y0 + ly;
        if (gx < a.W && gy < a.H) {
            const long idx = ((long)b * a.H + gy) * a.W + gx;
            const float g = a.pred[idx];
            const float p = a.is_sigmoid ? 1.f / (1.f + expf(-(g - a.thresh))) : g;
            const float e = a.edge[idx];
            const float keep = (use_keep && a.mask && a.mask[idx] == 0.f) ? 0.f : 1.f;
            const float dp = a.is_sigmoid ? p * (1.f - p) : 1.f;
            a.dpred[idx] = keep * dp * (-cpos * e / (p + 0.001f) + cneg * (1.f - e) / (1.f - p + 0.001f));
        }
        return;
    }
    load_depth_tile<2>(a, b, x0, y0, sd);
    __syncthreads();
    constexpr int GW = TX + 2, GH = TY + 2;
    for (int i = threadIdx.x; i < GW * GH; i += 256) {
        const int ly = i / GW, lx = i % GW;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        float ga = 0.f, gb = 0.f; int code = 0;
        if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
            const long idx = ((long)b * a.H + gy) * a.W + gx;
            float sh, sv, srl, slr;
            sobel4<TX + 4>(sd, ly + 1, lx + 1, sh, sv, srl, slr);
            float g, da, db = 0.f;        // d g / d s_a, d g / d s_b
            if (a.normal) {
                code = direction_code(a.normal[idx]);
                const float s = code == 0 ? sh : (code == 1 ? sv : (code == 2 ? srl : slr));
                g = fabsf(s);
                da = s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f);
            } else {
                code = 4;                  // magnitude: a = v, b = h
                g = sqrtf(sv * sv + sh * sh + 1e-6f);
                da = sv / g; db = sh / g;
            }
            const float p = a.is_sigmoid ? 1.f / (1.f + expf(-(g - a.thresh))) : g;
            const float e = a.edge[idx];
            const float keep = (use_keep && a.mask && a.mask[idx] == 0.f) ? 0.f : 1.f;
            const float dp = a.is_sigmoid ? p * (1.f - p) : 1.f;
            const float dg = keep * dp * (-cpos * e / (p + 0.001f) + cneg * (1.f - e) / (1.f - p + 0.001f));
            ga = dg * da; gb = dg * db;
        }
        sga[i] = ga; sgb[i] = gb; scode[i] = code;
    }
    __syncthreads();
    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX, gx = x0 + lx, gy = y0 + ly;
    if (gx < a.W && gy < a.H) {
        // d loss / d depth(q) = sum_t K_{code(p)}[t] * G(p), p = q - t  (K indexed by t = (dy+1)*3 + (dx+1))
        float dd = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const int i = (ly + 1 - dy) * GW + (lx + 1 - dx);
            const int code = scode[i];
            if (code == 4) dd += c_sobel[1][t] * sga[i] + c_sobel[0][t] * sgb[i];
            else dd += c_sobel[code][t] * sga[i];
        }
        const long idx = ((long)b * a.H + gy) * a.W + gx;
        if (a.from_inv) {
            const float inv = a.pred[idx];
            const float d = 1.f / fmaxf(inv, 1e-6f);
            dd = inv >= 1e-6f ? -dd * d * d : 0.f;
        }
        a.dpred[idx] = dd;
    }
}

// ---------------- silog -------------------------------------------------------------------------
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

}  // namespace

extern "C" {

static inline void edge_fwd_grid(int H, int W, int& gx, int& gy) { gx = (W + TX - 1) / TX; gy = ((H + TY - 1) / TY + FWD_TILES - 1) / FWD_TILES; }

// doubles the caller must provide as `sums`: the B*6 + 4 results followed by the per-workgroup partial sums
long mte_edge_loss_sums_elems(int B, int H, int W) {
    int gx, gy; edge_fwd_grid(H, W, gx, gy);
    return (long)B * 6 + 4 + (long)B * gx * gy * 10;
}

// Forward pass of one scale.  sums: mte_edge_loss_sums_elems(B, H, W) doubles (content on entry ignored).  gmap nullable.
int mte_edge_loss_fwd(const float* pred, const float* edge, const float* normal, const float* mask, double* sums, float* gmap,
                      int B, int H, int W, int from_inv, int is_grad, int is_sigmoid, float thresh, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!pred || !edge || !sums) return MTE_ERR_ARG;
    EdgeArgs a{}; a.pred = pred; a.edge = edge; a.normal = normal; a.mask = mask; a.sums = sums; a.gmap = gmap;
    a.B = B; a.H = H; a.W = W; a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.thresh = thresh;
    int gx, gy; edge_fwd_grid(H, W, gx, gy);
    hipLaunchKernelGGL(edge_loss_fwd_kernel, dim3(gx, gy, B), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(edge_sums_reduce_kernel, dim3(B * 6 + 4), dim3(256), 0, stream, sums, B, gx * gy);
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
    EdgeArgs a{}; a.pred = pred; a.edge = edge; a.normal = normal; a.mask = mask; a.coef = coef; a.gout = gout; a.dpred = dpred;
    a.B = B; a.H = H; a.W = W; a.from_inv = from_inv; a.is_grad = is_grad; a.is_sigmoid = is_sigmoid; a.thresh = thresh;
    hipLaunchKernelGGL(edge_loss_bwd_kernel, dim3((W + TX - 1) / TX, (H + TY - 1) / TY, B), dim3(256), 0, stream, a);
    return mte_check_launch();
}

// sums[3] doubles (zeroed here); aux[2] floats
int mte_silog_fwd(const float* inv, const float* depth, long n, double* sums, float out_scale, float* loss_acc, float* loss_this, float* aux, hipStream_t stream) {
    (void)hipGetLastError();   // drop stale errors left by other runtime users (e.g. event queries)
    if (!inv || !depth || !sums || !aux) return MTE_ERR_ARG;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 3, stream) != hipSuccess) return MTE_ERR_LAUNCH;
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
