// LDS-direct 16-byte loads on gfx950: lane i of a wave lands at M0 + 16 i (prints 16 i + 6 for in[j] = j).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/ldslds.cpp -o /tmp/ldslds && /tmp/ldslds
#include <hip/hip_runtime.h>
__global__ void k(float* o, const float* in) {
  __shared__ float4 buf[256];
  const float* src = in + threadIdx.x * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  float4 v = buf[threadIdx.x];
  o[threadIdx.x] = v.x + v.y + v.z + v.w;
}
int main() {
  float *d, *in, h[64], hi[256];
  for (int i = 0; i < 256; ++i) hi[i] = i;
  hipMalloc(&d, 256); hipMalloc(&in, 1024); hipMemcpy(in, hi, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%g ", h[i]); printf("\n");
}
