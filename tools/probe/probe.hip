// Hardware probe (development aid, not product code): prints the lane->element maps this
// build relies on: MFMA 32x32x16 bf16 / 32x32x2 f32 operand+accumulator layout and
// ds_read_b64_tr_b16 block-transpose semantics.  Build: hipcc --offload-arch=gfx950 probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ inline unsigned short f2bf(float f){ unsigned u = __float_as_uint(f); return (unsigned short)((u + 0x7FFF + ((u>>16)&1))>>16); }
__device__ inline float bf2f(unsigned short h){ return __uint_as_float(((unsigned)h)<<16); }

// C[32][32] = A[32][16] * B[16][32]; A row-major [m][k], B given as Bt[n][k]
__global__ void k_mfma_bf16(const unsigned short* A, const unsigned short* Bt, float* C){
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  union { bf16x8 v; unsigned short s[8]; } a, b;
  for(int j=0;j<8;j++){ a.s[j] = A[r*16 + 8*h + j]; b.s[j] = Bt[r*16 + 8*h + j]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0,0,0);
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; C[row*32 + r] = c[i]; }
}
// f32: C[32][32] = A[32][2] * B[2][32]
__global__ void k_mfma_f32(const float* A, const float* Bt, float* C){
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r*2+h], Bt[r*2+h], c, 0,0,0);
  for(int i=0;i<16;i++){ int row = (i&3) + 8*(i>>2) + 4*h; C[row*32 + r] = c[i]; }
}
// tr16: LDS image [16 rows][64 cols] of u16 = row*64+col (row stride 128 B). Each 16-lane group g reads
// the 4x16 block at rows 4g..4g+3, cols 16g.. ; lane 4q+p supplies &img[4g+q][16g+4p]
__global__ void k_tr16(unsigned short* out){
  __shared__ __attribute__((aligned(16))) unsigned short img[16*64];
  for(int i=threadIdx.x;i<16*64;i+=64) img[i] = (unsigned short)i;
  __syncthreads();
  int l = threadIdx.x, g = l>>4, i = l&15, q = i>>2, p = i&3;
  const unsigned short* addr = &img[(4*g+q)*64 + 16*g + 4*p];
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for(int e=0;e<4;e++) out[l*4+e] = (unsigned short)t[e];
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n){
  size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x*blockDim.x;
  for(; i<n; i+=st) b[i] = a[i];
}
// buffer load OOB -> 0 ?
__global__ void k_buf(const float* a, float* o, int nbytes){
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
  float v0 = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x*4, 0, 0) == 0 ? 0.f : 1.f;
  unsigned u = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x*4, 0, 0);
  unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rs, 0x7ffffff0u, 0, 0);
  o[threadIdx.x] = __uint_as_float(u); o[64+threadIdx.x] = __uint_as_float(w) + 0*v0;
}
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} }while(0)
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0)); printf("device %s CUs %d gcn %s\n", p.name, p.multiProcessorCount, p.gcnArchName);
  { // bf16 mfma
    std::vector<unsigned short> A(32*16), Bt(32*16); std::vector<float> Af(32*16), Bf(32*16), C(1024), R(1024,0.f);
    for(int i=0;i<512;i++){ float a = (float)((i*7)%13 - 6), b = (float)((i*5)%11 - 5); Af[i]=a; Bf[i]=b;
      unsigned ua; std::memcpy(&ua,&a,4); A[i]=ua>>16; unsigned ub; std::memcpy(&ub,&b,4); Bt[i]=ub>>16; }
    for(int m=0;m<32;m++)for(int n=0;n<32;n++){ float s=0; for(int k=0;k<16;k++) s+=Af[m*16+k]*Bf[n*16+k]; R[m*32+n]=s; }
    unsigned short *dA,*dB; float* dC; CK(hipMalloc(&dA,1024)); CK(hipMalloc(&dB,1024)); CK(hipMalloc(&dC,4096));
    CK(hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,Bt.data(),1024,hipMemcpyHostToDevice));
    k_mfma_bf16<<<1,64>>>(dA,dB,dC); CK(hipMemcpy(C.data(),dC,4096,hipMemcpyDeviceToHost));
    double e=0; for(int i=0;i<1024;i++) e=fmax(e,fabs(C[i]-R[i])); printf("mfma_bf16_32x32x16 layout maxerr %g\n", e);
  }
  { std::vector<float> A(64),Bt(64),C(1024),R(1024);
    for(int i=0;i<64;i++){ A[i]=(float)((i*7)%13-6)+0.25f; Bt[i]=(float)((i*5)%11-5)-0.5f; }
    for(int m=0;m<32;m++)for(int n=0;n<32;n++){ R[m*32+n]=A[m*2]*Bt[n*2]+A[m*2+1]*Bt[n*2+1]; }
    float *dA,*dB,*dC; CK(hipMalloc(&dA,256)); CK(hipMalloc(&dB,256)); CK(hipMalloc(&dC,4096));
    CK(hipMemcpy(dA,A.data(),256,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,Bt.data(),256,hipMemcpyHostToDevice));
    k_mfma_f32<<<1,64>>>(dA,dB,dC); CK(hipMemcpy(C.data(),dC,4096,hipMemcpyDeviceToHost));
    double e=0; for(int i=0;i<1024;i++) e=fmax(e,fabs(C[i]-R[i])); printf("mfma_f32_32x32x2 layout maxerr %g\n", e);
  }
  { unsigned short* d; CK(hipMalloc(&d,64*4*2)); k_tr16<<<1,64>>>(d); std::vector<unsigned short> o(256); CK(hipMemcpy(o.data(),d,512,hipMemcpyDeviceToHost));
    int bad=0; for(int l=0;l<64;l++){ int g=l>>4,i=l&15; for(int e=0;e<4;e++){ int exp=(4*g+e)*64 + 16*g + i; if(o[l*4+e]!=exp) bad++; } }
    printf("tr16_b64 expected-map mismatches %d\n", bad);
    for(int l=0;l<64;l+=5){ printf(" lane %2d:", l); for(int e=0;e<4;e++) printf(" (r%d,c%d)", o[l*4+e]/64, o[l*4+e]%64); printf("\n"); }
  }
  { float *a,*o; CK(hipMalloc(&a,256)); CK(hipMalloc(&o,512)); std::vector<float> h(64); for(int i=0;i<64;i++) h[i]=i+1; CK(hipMemcpy(a,h.data(),256,hipMemcpyHostToDevice));
    k_buf<<<1,64>>>(a,o,128); std::vector<float> r(128); CK(hipMemcpy(r.data(),o,512,hipMemcpyDeviceToHost));
    printf("buffer_load: in-range[3]=%g oob-lane40=%g far-oob=%g\n", r[3], r[40], r[64]); }
  { size_t n = (size_t)1<<30; float4 *a,*b; CK(hipMalloc(&a,n)); CK(hipMalloc(&b,n)); CK(hipMemset(a,1,n));
    hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for(int it=0;it<3;it++){ hipEventRecord(e0); k_copy<<<2048,256>>>(a,b,n/16); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1);
      printf("copy 1GiB: %.3f ms -> %.2f TB/s (r+w)\n", ms, 2.0*n/ms/1e9); } }
  return 0;
}
