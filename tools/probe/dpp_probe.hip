// development probe (round 4): which lane a DPP row shift / rotate reads from on gfx950.  hipcc --offload-arch=gfx950 dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int l = threadIdx.x;
    out[0 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xF, 0xF, false);   // row_shr:1
    out[1 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x101, 0xF, 0xF, false);   // row_shl:1
    out[2 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x121, 0xF, 0xF, false);   // row_ror:1
    out[3 * 64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x12F, 0xF, 0xF, false);   // row_ror:15
}
int main() {
    int* d; hipMalloc(&d, 4 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"row_shr:1 (0x111)", "row_shl:1 (0x101)", "row_ror:1 (0x121)", "row_ror:15 (0x12F)"};
    for (int i = 0; i < 4; ++i) { printf("%-20s lane 0..17 read from:", names[i]); for (int l = 0; l < 18; ++l) printf(" %d", h[i * 64 + l]); printf("\n"); }
    return 0;
}
