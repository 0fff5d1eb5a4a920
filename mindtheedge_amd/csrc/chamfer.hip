// Chamfer edge metrics on device for gfx950 (edge half after the Canny step).
//
// chamfer_distance(im_pred, im_gt) of packnet_sfm/utils/edge.py:19-64 (same body as /root/reference/edge.py:29-71):
//   binarise both images (v/255 > 0.5), Euclidean distance transform of the ground-truth edges
//   (scipy.ndimage.distance_transform_edt), c_dist = mean distance over the predicted edge pixels, percentage = share
//   of predicted edge pixels closer than edge_to_edge_thresh, and the -1/0/1 map used for visualisation.
// The distance transform is exact: squared distances are integers (column scan, then the minimum over the row of
// dx^2 + g^2 in LDS, searched outward from the pixel and cut off once dx^2 can no longer improve), the square root is
// taken in double -- the same values scipy produces.  Only predicted edge pixels need a distance, so the row pass
// does work for ~5 % of the pixels unless the full map is requested.  Integer / byte work, HBM- and LDS-bound.
#include "common.hpp"

namespace {

constexpr unsigned EDT_INF = 1u << 30;

// v/255 > 0.5 in double (edge.py:30-32) <=> v > 127.5 for every float32 v (127.5 is exact, the quotient is monotone)
__device__ __forceinline__ bool is_edge(float v) { return v > 127.5f; }

constexpr int EC = 64, ES = 16;                               // columns per block x row segments per column (1024 threads)

// distance along the column to the nearest ground-truth edge pixel, squared (EDT_INF if none).  A column is cut into ES
// segments scanned in parallel; the segments exchange their first / last edge row through LDS.
__global__ __launch_bounds__(EC * ES) void edt_columns_kernel(const float* __restrict__ gt, unsigned* __restrict__ g2, int H, int W) {
    __shared__ int s_first[ES][EC], s_last[ES][EC];
    const int c = threadIdx.x % EC, seg = threadIdx.x / EC;
    const int x = blockIdx.x * EC + c, b = blockIdx.y;
    const int per = (H + ES - 1) / ES, r0 = seg * per, r1 = min(H, r0 + per);
    const bool live = x < W;
    const float* src = gt + (long)b * H * W + x;
    unsigned* dst = g2 + (long)b * H * W + x;
    int first = -1, last = -1;
    if (live)
        for (int y = r0; y < r1; ++y)
            if (is_edge(src[(long)y * W])) { if (first < 0) first = y; last = y; }
    s_first[seg][c] = first; s_last[seg][c] = last;
    __syncthreads();
    if (!live) return;
    int above = -1, below = -1;                               // nearest edge rows outside this segment
    for (int k = 0; k < seg; ++k) if (s_last[k][c] >= 0) above = s_last[k][c];
    for (int k = ES - 1; k > seg; --k) if (s_first[k][c] >= 0) below = s_first[k][c];
    int near = above;
    for (int y = r0; y < r1; ++y) {                            // nearest edge at or above y
        if (is_edge(src[(long)y * W])) near = y;
        dst[(long)y * W] = near < 0 ? EDT_INF : (unsigned)((y - near) * (y - near));
    }
    near = below;
    for (int y = r1 - 1; y >= r0; --y) {                       // nearest edge at or below y
        if (is_edge(src[(long)y * W])) near = y;
        if (near >= 0) {
            const unsigned d = (unsigned)((near - y) * (near - y));
            if (d < dst[(long)y * W]) dst[(long)y * W] = d;
        }
    }
}

// one block per (row, image): exact squared distance for the predicted edge pixels of the row (all pixels when dist is set)
__global__ __launch_bounds__(256) void edt_rows_chamfer_kernel(const float* __restrict__ pred, const unsigned* __restrict__ g2,
                                                               double* __restrict__ acc, float* __restrict__ dist, float* __restrict__ cond,
                                                               int H, int W, double thresh) {
    extern __shared__ unsigned sg[];                           // g^2 of this row
    __shared__ double sred[4][3];
    const int y = blockIdx.x, b = blockIdx.y;
    const long row = ((long)b * H + y) * W;
    for (int i = threadIdx.x; i < W; i += 256) sg[i] = g2[row + i];
    __syncthreads();
    double sum = 0.0, n = 0.0, nclose = 0.0;
    for (int x = threadIdx.x; x < W; x += 256) {
        const bool e = is_edge(pred[row + x]);
        if (!e && !dist) { if (cond) cond[row + x] = -1.f; continue; }
        unsigned best = sg[x];
        for (int d = 1; d < W; ++d) {
            const unsigned dd = (unsigned)d * (unsigned)d;
            if (dd >= best) break;                            // farther columns cannot beat the current minimum
            if (x - d >= 0) best = min(best, dd + sg[x - d]);
            if (x + d < W) best = min(best, dd + sg[x + d]);
        }
        const double dv = best >= EDT_INF ? __builtin_huge_val() : sqrt((double)best);
        if (dist) dist[row + x] = (float)dv;
        if (e) {
            sum += dv; n += 1.0;
            if (dv < thresh) nclose += 1.0;
        }
        if (cond) cond[row + x] = e ? (dv < thresh ? 1.f : 0.f) : -1.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double r0 = wave_sum_d(sum), r1 = wave_sum_d(n), r2 = wave_sum_d(nclose);
    if (lane == 0) { sred[wave][0] = r0; sred[wave][1] = r1; sred[wave][2] = r2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double s = sred[0][threadIdx.x] + sred[1][threadIdx.x] + sred[2][threadIdx.x] + sred[3][threadIdx.x];
        if (s != 0.0) atomicAdd(&acc[b * 4 + threadIdx.x], s);
    }
}

__global__ void chamfer_final_kernel(const double* __restrict__ acc, double* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    out[b * 2 + 0] = acc[b * 4 + 0] / acc[b * 4 + 1];          // 0/0 = NaN when nothing is predicted, like the reference
    out[b * 2 + 1] = acc[b * 4 + 2] / acc[b * 4 + 1];
}

}  // namespace

extern "C" long mte_chamfer_workspace_bytes(int B, int H, int W) { return (long)B * H * W * 4 + (long)B * 4 * 8; }

extern "C" int mte_chamfer_distance(const float* im_pred, const float* im_gt, int B, int H, int W, double edge_to_edge_thresh,
                                    void* workspace, long workspace_bytes, double* out, float* dist_map, float* cond_map,
                                    hipStream_t stream) {
    if (!im_pred || !im_gt || !workspace || !out || B <= 0 || H <= 0 || W <= 0 || H >= 32768 || W >= 32768) return MTE_ERR_ARG;
    if (workspace_bytes < mte_chamfer_workspace_bytes(B, H, W)) return MTE_ERR_ARG;
    if ((size_t)W * 4 > 60000) return MTE_ERR_UNSUPPORTED;                                  // one row of g^2 lives in LDS
    double* acc = (double*)workspace;                                                       // [B][4], 8-byte aligned first
    unsigned* g2 = (unsigned*)((char*)workspace + (long)B * 4 * 8);
    if (mte_memset_async(acc, 0, (size_t)B * 4 * 8, stream) != hipSuccess) return MTE_ERR_LAUNCH;
    hipLaunchKernelGGL(edt_columns_kernel, dim3(cdiv(W, EC), B), dim3(EC * ES), 0, stream, im_gt, g2, H, W);
    hipLaunchKernelGGL(edt_rows_chamfer_kernel, dim3(H, B), dim3(256), (size_t)W * 4, stream, im_pred, g2, acc, dist_map, cond_map, H, W,
                       edge_to_edge_thresh);
    hipLaunchKernelGGL(chamfer_final_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, acc, out, B);
    return mte_check_launch();
}
