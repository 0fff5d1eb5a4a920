// development probe: semantics of v_dot2c_f32_bf16 on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const unsigned* b, float* o) {
    o[threadIdx.x] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[threadIdx.x]), __builtin_bit_cast(bf16x2, b[threadIdx.x]), 100.f, false);
}
static unsigned bf(float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; }
int main() {
    unsigned ha[4] = {bf(2.f) | (bf(3.f) << 16), bf(1.f) | (bf(0.f) << 16), bf(0.f) | (bf(1.f) << 16), bf(-1.5f) | (bf(4.f) << 16)};
    unsigned hb[4] = {bf(5.f) | (bf(7.f) << 16), bf(10.f) | (bf(20.f) << 16), bf(10.f) | (bf(20.f) << 16), bf(2.f) | (bf(0.5f) << 16)};
    unsigned *a, *b; float* o; float ho[4];
    hipMalloc(&a, 16); hipMalloc(&b, 16); hipMalloc(&o, 16);
    hipMemcpy(a, ha, 16, hipMemcpyHostToDevice); hipMemcpy(b, hb, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, a, b, o);
    hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost);
    printf("expect 131 110 120 99 : got %g %g %g %g\n", ho[0], ho[1], ho[2], ho[3]);
    return 0;
}
