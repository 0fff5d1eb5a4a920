// Training-target preparation on device for gfx950 (data half): what the reference's dataset code does
// on the host, per sample, between reading the annotation PNGs and handing the batch to the model.
//   edges     sample/255 when the map is on the 0..255 scale                   datasets/augmentations.py:186-188,199-201
//   normals   (360 * (v / 255) - 180) * (pi / 180)  from the uint8 PNG          datasets/gta_dataset.py:407-409,417-418
//   resize_depth_preserve: every valid (> 0) pixel of a sparse map is moved to int(y * H/h), int(x * W/w) of the output;
//             when several land on one output pixel the LAST one in raster order wins (numpy fancy assignment), pixels
//             that land outside are dropped                                    datasets/augmentations.py:58-100
// Byte / index work, HBM-bound; double arithmetic where the reference computes in float64 so the float32 results match.
#include "common.hpp"

namespace {

// dst = float( (double(src) * mul / 255 + add) * post )   -- edges: mul 1, add 0, post 1 ; normals: mul 360, add -180, post pi/180
__global__ __launch_bounds__(256) void u8_to_target_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long n,
                                                           double mul, double add, double post) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dst[i] = (float)((mul * ((double)src[i] / 255.0) + add) * post);
}

__global__ __launch_bounds__(256) void rdp_scatter_kernel(const float* __restrict__ src, int* __restrict__ winner, int h, int w, int H, int W,
                                                          double sy, double sx) {
    const int b = blockIdx.y;
    const int n = h * w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (!(src[(long)b * n + i] > 0.f)) continue;
        const int y = i / w, x = i - y * w;
        const int ty = (int)((double)y * sy), tx = (int)((double)x * sx);
        if (ty < H && tx < W) atomicMax(&winner[((long)b * H + ty) * W + tx], i);      // raster order: the largest source index wins
    }
}

__global__ __launch_bounds__(256) void rdp_gather_kernel(const float* __restrict__ src, const int* __restrict__ winner, float* __restrict__ dst,
                                                         int hw, int HW) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const int k = winner[(long)b * HW + i];
        dst[(long)b * HW + i] = k < 0 ? 0.f : src[(long)b * hw + k];
    }
}

inline unsigned grid_for(long total) { long g = (total + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1; return (unsigned)g; }

}  // namespace

extern "C" {

int mte_edge_target_from_u8(const unsigned char* src, float* dst, long n, hipStream_t stream) {
    if (!src || !dst || n <= 0) return MTE_ERR_ARG;
    hipLaunchKernelGGL(u8_to_target_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, n, 1.0, 0.0, 1.0);
    return mte_check_launch();
}

int mte_normal_target_from_u8(const unsigned char* src, float* dst, long n, hipStream_t stream) {
    if (!src || !dst || n <= 0) return MTE_ERR_ARG;
    hipLaunchKernelGGL(u8_to_target_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, n, 360.0, -180.0, 3.141592653589793 / 180.0);
    return mte_check_launch();
}

int mte_resize_depth_preserve(const float* src, int B, int h, int w, float* dst, int H, int W, int* winner_ws, hipStream_t stream) {
    if (!src || !dst || !winner_ws || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || (long)h * w >= (1L << 30) || (long)H * W >= (1L << 30))
        return MTE_ERR_ARG;
    if (mte_memset_async(winner_ws, 0xff, sizeof(int) * (size_t)B * H * W, stream) != hipSuccess) return MTE_ERR_LAUNCH;      // -1
    hipLaunchKernelGGL(rdp_scatter_kernel, dim3(grid_for((long)h * w), B), dim3(256), 0, stream, src, winner_ws, h, w, H, W,
                       (double)H / (double)h, (double)W / (double)w);
    hipLaunchKernelGGL(rdp_gather_kernel, dim3(grid_for((long)H * W), B), dim3(256), 0, stream, src, winner_ws, dst, h * w, H * W);
    return mte_check_launch();
}

}  // extern "C"
