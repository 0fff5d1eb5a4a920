// Validation depth metrics on device for gfx950 (depth half).
//
// Replaces, for fp32 [B,1,H,W] maps already resident in HBM:
//   post_process_inv_depth   packnet_sfm/utils/depth.py:230-256  (flip-TTA fusion, fuse_inv_depth :202-227)
//   compute_depth_metrics    packnet_sfm/utils/depth.py:259-325  (scale_depth :328-361, 'resize' = bilinear with
//                            align_corners=True utils/image.py:122-151; garg crop; torch.median scaling; 7 metrics)
// The reference loops over the images on the host, compacts the valid pixels (boolean indexing = a host sync each),
// calls torch.median twice (a sort) and ~25 small reductions per image.  Here nothing is compacted or sorted:
//   * the prediction is sampled at ground-truth resolution on the fly (never materialised),
//   * the two medians (ground truth and prediction over the valid pixels) come from a 3-pass radix select on the
//     order-preserving integer image of the floats (11+11+10 bits, LDS histograms) -- exactly torch.median's element
//     of rank (n-1)/2, no averaging,
//   * one pass accumulates all seven sums in double, one block turns them into the batch-averaged float[7].
// No host synchronisation; 8 B/pixel of HBM traffic per pass (4 passes with median scaling, 1 without).
#include "common.hpp"
#include <algorithm>

namespace {

struct MetricArgs {
    const float* gt;            // [B][H][W]
    const float* pred;          // [B][h][w]
    int B, H, W, h, w;
    int scale_mode;             // 0 'resize' (bilinear, align_corners), 1 'top-center'
    int y1, y2, x1, x2;         // crop window (whole image without crop)
    float lo, hi;               // min_depth, max_depth
    float sy, sx;               // (h-1)/(H-1), (w-1)/(W-1)
    int top, left;
    int use_gt_scale;
};

constexpr int NB = 2048;        // bins per radix pass
constexpr int STATE_U32 = 8;    // prefix[2], krem[2], n, pad, median bits[2]
// workspace per image: uint32 hist[2][NB] | uint32 state[STATE_U32] | double sums[8]
constexpr long WS_IMAGE_BYTES = 2 * NB * 4 + STATE_U32 * 4 + 8 * 8;

__device__ __forceinline__ unsigned* ws_hist(void* ws, int b) { return (unsigned*)((char*)ws + (long)b * WS_IMAGE_BYTES); }
__device__ __forceinline__ unsigned* ws_state(void* ws, int b) { return ws_hist(ws, b) + 2 * NB; }
__device__ __forceinline__ double* ws_sums(void* ws, int b) { return (double*)(ws_state(ws, b) + STATE_U32); }

__device__ __forceinline__ unsigned order_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// prediction at ground-truth pixel (y, x)
__device__ __forceinline__ float sample_pred(const MetricArgs& a, const float* p, int y, int x) {
    if (a.scale_mode == 1) {
        const int py = y - a.top, px = x - a.left;
        return ((unsigned)py < (unsigned)a.h && (unsigned)px < (unsigned)a.w) ? p[(long)py * a.w + px] : 0.f;
    }
    if (a.h == a.H && a.w == a.W) return p[(long)y * a.w + x];
    const float ry = a.sy * (float)y, rx = a.sx * (float)x;
    int y0 = min((int)ry, a.h - 1), x0 = min((int)rx, a.w - 1);
    const int y1 = min(y0 + 1, a.h - 1), x1 = min(x0 + 1, a.w - 1);
    const float ly = ry - (float)y0, lx = rx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float A = p[(long)y0 * a.w + x0], Bv = p[(long)y0 * a.w + x1], C = p[(long)y1 * a.w + x0], D = p[(long)y1 * a.w + x1];
    return hy * (hx * A + lx * Bv) + ly * (hx * C + lx * D);
}

// the cropped window is enumerated densely: i in [0, (y2-y1)*(x2-x1))
__device__ __forceinline__ bool valid_pair(const MetricArgs& a, int b, long i, float& g, float& p) {
    const int cw = a.x2 - a.x1;
    const int y = a.y1 + (int)(i / cw), x = a.x1 + (int)(i % cw);
    g = a.gt[((long)b * a.H + y) * a.W + x];
    if (!(g > a.lo && g < a.hi)) return false;
    p = sample_pred(a, a.pred + (long)b * a.h * a.w, y, x);
    return true;
}

template <int PASS>
__global__ __launch_bounds__(256) void median_hist_kernel(MetricArgs a, void* ws) {
    __shared__ unsigned sh[2][NB];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * NB; i += 256) (&sh[0][0])[i] = 0;
    __syncthreads();
    const unsigned* st = ws_state(ws, b);
    const unsigned pg = st[0], pp = st[1];
    const long total = (long)(a.y2 - a.y1) * (a.x2 - a.x1);
    if (PASS == 0 || st[4] != 0) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            float g, p;
            if (!valid_pair(a, b, i, g, p)) continue;
            const unsigned kg = order_key(g), kp = order_key(p);
            if (PASS == 0) {
                atomicAdd(&sh[0][kg >> 21], 1u);
                atomicAdd(&sh[1][kp >> 21], 1u);
            } else if (PASS == 1) {
                if ((kg >> 21) == pg) atomicAdd(&sh[0][(kg >> 10) & 2047u], 1u);
                if ((kp >> 21) == pp) atomicAdd(&sh[1][(kp >> 10) & 2047u], 1u);
            } else {
                if ((kg >> 10) == pg) atomicAdd(&sh[0][kg & 1023u], 1u);
                if ((kp >> 10) == pp) atomicAdd(&sh[1][kp & 1023u], 1u);
            }
        }
    }
    __syncthreads();
    unsigned* hist = ws_hist(ws, b);
    for (int i = threadIdx.x; i < 2 * NB; i += 256) {
        const unsigned v = (&sh[0][0])[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// one block per (which, image): finds the bin holding the wanted rank, extends the prefix, clears the histogram
template <int PASS>
__global__ __launch_bounds__(256) void median_select_kernel(void* ws) {
    __shared__ unsigned scan[256];
    __shared__ unsigned total_s;
    const int which = blockIdx.x, b = blockIdx.y;
    unsigned* hist = ws_hist(ws, b) + which * NB;
    unsigned* st = ws_state(ws, b);
    unsigned v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = hist[threadIdx.x * 8 + j]; s += v[j]; }
    scan[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                   // inclusive Hillis-Steele scan
        const unsigned add = threadIdx.x >= o ? scan[threadIdx.x - o] : 0u;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    if (threadIdx.x == 255) total_s = scan[255];
    __syncthreads();
    const unsigned n = PASS == 0 ? total_s : st[4];
    unsigned k = PASS == 0 ? (n ? (n - 1) / 2 : 0u) : st[2 + which];
    const unsigned incl = scan[threadIdx.x], excl = incl - s;
    __syncthreads();                                      // every read of st[] above precedes the writes below
    if (n != 0 && k >= excl && k < incl) {
        unsigned c = excl;
        int bin = threadIdx.x * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (k >= c + v[j]) { c += v[j]; bin = threadIdx.x * 8 + j + 1; } else break;
        }
        const unsigned prefix = PASS == 0 ? (unsigned)bin : PASS == 1 ? ((st[which] << 11) | (unsigned)bin) : ((st[which] << 10) | (unsigned)bin);
        st[which] = prefix;
        st[2 + which] = k - c;
        if (PASS == 2) st[6 + which] = __float_as_uint(key_to_float(prefix));
    }
    if (PASS == 0 && which == 0 && threadIdx.x == 0) st[4] = n;
#pragma unroll
    for (int j = 0; j < 8; ++j) hist[threadIdx.x * 8 + j] = 0;
}

__global__ __launch_bounds__(256) void metric_sums_kernel(MetricArgs a, void* ws) {
    __shared__ double sred[4][8];
    const int b = blockIdx.y;
    const unsigned* st = ws_state(ws, b);
    const float mg = __uint_as_float(st[6]), mp = __uint_as_float(st[7]);
    const long total = (long)(a.y2 - a.y1) * (a.x2 - a.x1);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float g, p;
        if (!valid_pair(a, b, i, g, p)) continue;
        if (a.use_gt_scale) p = (p * mg) / mp;
        p = fminf(fmaxf(p, a.lo), a.hi);
        const float thresh = fmaxf(g / p, p / g);
        const float d = g - p;
        const float dl = logf(g) - logf(p);
        acc[0] += (double)(fabsf(d) / g);
        acc[1] += (double)(d * d / g);
        acc[2] += (double)(d * d);
        acc[3] += (double)(dl * dl);
        acc[4] += thresh < 1.25f ? 1.0 : 0.0;
        acc[5] += thresh < 1.5625f ? 1.0 : 0.0;
        acc[6] += thresh < 1.953125f ? 1.0 : 0.0;
        acc[7] += 1.0;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double r = wave_sum_d(acc[j]);
        if (lane == 0) sred[wave][j] = r;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const double r = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
        if (r != 0.0) atomicAdd(&ws_sums(ws, b)[threadIdx.x], r);
    }
}

// images without a valid pixel add nothing but the divisor stays B (depth.py:297-299,322-324)
__global__ void metric_final_kernel(void* ws, int B, float* out7) {
    const int j = threadIdx.x;
    if (j >= 7) return;
    double acc = 0.0;
    for (int b = 0; b < B; ++b) {
        const double* s = ws_sums(ws, b);
        const double n = s[7];
        if (n == 0.0) continue;
        double m;
        if (j == 0) m = s[0] / n;
        else if (j == 1) m = s[1] / n;
        else if (j == 2) m = sqrt(s[2] / n);
        else if (j == 3) m = sqrt(s[3] / n);
        else m = s[j] / n;
        acc += m;
    }
    out7[j] = (float)(acc / (double)B);
}

__global__ __launch_bounds__(256) void post_process_kernel(const float* __restrict__ inv, const float* __restrict__ invf,
                                                           float* __restrict__ out, long rows, int W, int method) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * W) return;
    const int x = (int)(i % W);
    const long row = i - x;
    const float step = 1.f / (float)max(W - 1, 1);
    // torch.linspace(0,1,W) in float32: counted from the nearer end
    auto xs = [&](int t) { return t < W / 2 ? step * (float)t : 1.f - step * (float)(W - 1 - t); };
    auto ramp = [&](int t) { return 1.f - fminf(fmaxf(20.f * (xs(t) - 0.05f), 0.f), 1.f); };
    const float a = inv[i], hat = invf[row + (W - 1 - x)];
    const float fused = method == 0 ? 0.5f * (a + hat) : method == 1 ? fmaxf(a, hat) : fminf(a, hat);
    const float m = ramp(x), mh = ramp(W - 1 - x);
    out[i] = mh * a + m * hat + (1.f - m - mh) * fused;
}

}  // namespace

extern "C" long mte_depth_metrics_workspace_bytes(int B) { return (long)B * WS_IMAGE_BYTES; }

extern "C" int mte_depth_metrics(const float* gt, const float* pred, int B, int H, int W, int h, int w, int scale_mode, int garg_crop,
                                 float min_depth, float max_depth, int use_gt_scale, void* workspace, long workspace_bytes,
                                 float* out7, hipStream_t stream) {
    if (!gt || !pred || !workspace || !out7 || B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return MTE_ERR_ARG;
    if (workspace_bytes < mte_depth_metrics_workspace_bytes(B)) return MTE_ERR_ARG;
    if (scale_mode != 0 && scale_mode != 1) return MTE_ERR_UNSUPPORTED;
    if (scale_mode == 1 && (h > H || w > W)) return MTE_ERR_ARG;
    MetricArgs a;
    a.gt = gt; a.pred = pred; a.B = B; a.H = H; a.W = W; a.h = h; a.w = w; a.scale_mode = scale_mode;
    a.y1 = 0; a.y2 = H; a.x1 = 0; a.x2 = W;
    if (garg_crop) {                                       // depth.py:286-289 (python float arithmetic = double)
        a.y1 = (int)(0.40810811 * H); a.y2 = (int)(0.99189189 * H);
        a.x1 = (int)(0.03594771 * W); a.x2 = (int)(0.96405229 * W);
    }
    a.lo = min_depth; a.hi = max_depth;
    a.sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    a.sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    a.top = H - h; a.left = (W - w) / 2;
    a.use_gt_scale = use_gt_scale;
    if (mte_memset_async(workspace, 0, (size_t)mte_depth_metrics_workspace_bytes(B), stream) != hipSuccess) return MTE_ERR_LAUNCH;
    const long total = (long)(a.y2 - a.y1) * (a.x2 - a.x1);
    if (total > 0) {
        const int bx = (int)std::min<long>(std::max<long>(cdiv(total, 256 * 8), 1), 1024);
        const dim3 grid(bx, B), sel(2, B);
        if (use_gt_scale) {
            hipLaunchKernelGGL(median_hist_kernel<0>, grid, dim3(256), 0, stream, a, workspace);
            hipLaunchKernelGGL(median_select_kernel<0>, sel, dim3(256), 0, stream, workspace);
            hipLaunchKernelGGL(median_hist_kernel<1>, grid, dim3(256), 0, stream, a, workspace);
            hipLaunchKernelGGL(median_select_kernel<1>, sel, dim3(256), 0, stream, workspace);
            hipLaunchKernelGGL(median_hist_kernel<2>, grid, dim3(256), 0, stream, a, workspace);
            hipLaunchKernelGGL(median_select_kernel<2>, sel, dim3(256), 0, stream, workspace);
        }
        hipLaunchKernelGGL(metric_sums_kernel, grid, dim3(256), 0, stream, a, workspace);
    }
    hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, stream, workspace, B, out7);
    return mte_check_launch();
}

extern "C" int mte_post_process_inv_depth(const float* inv_depth, const float* inv_depth_flipped, float* out, int B, int H, int W,
                                          int method, hipStream_t stream) {
    if (!inv_depth || !inv_depth_flipped || !out || B <= 0 || H <= 0 || W <= 0) return MTE_ERR_ARG;
    if (method < 0 || method > 2) return MTE_ERR_UNSUPPORTED;
    const long rows = (long)B * H;
    hipLaunchKernelGGL(post_process_kernel, dim3(cdiv(rows * W, 256)), dim3(256), 0, stream, inv_depth, inv_depth_flipped, out, rows, W, method);
    return mte_check_launch();
}
