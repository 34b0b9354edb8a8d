// Which lane feeds which under the DPP controls the strip kernels use (fx_jacobi_strip.hip lane_up1 / lane_dn1 / row_dn<N>).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/dpp_test.cpp -o /tmp/dpp_test && /tmp/dpp_test
// MI355X prints: wave_shr:1 -> lane i reads lane i-1 (lane 0: 0); wave_shl:1 -> lane i reads lane i+1 (lane 63: 0);
// row_shl:N -> lane i reads lane i+N of its 16-lane row (0 past the row end, bound_ctrl).
#include <hip/hip_runtime.h>
__device__ __forceinline__ float wave_up1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float wave_dn1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false)); }
template <int N> __device__ __forceinline__ float row_shl(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true)); }
__global__ void k(float* o, const float* in) {
  float v = in[threadIdx.x];
  o[threadIdx.x] = wave_up1(v); o[64 + threadIdx.x] = wave_dn1(v); o[128 + threadIdx.x] = row_shl<2>(v); o[192 + threadIdx.x] = row_shl<10>(v);
}
int main() {
  float *d, *in, h[256], hi[64];
  for (int i = 0; i < 64; ++i) hi[i] = i + 1;
  hipMalloc(&d, 1024); hipMalloc(&in, 256); hipMemcpy(in, hi, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in); hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) { for (int i = 0; i < 64; ++i) printf("%g ", h[r * 64 + i]); printf("\n"); }
}
