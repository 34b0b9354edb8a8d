// Where does a 2-byte LDS-DMA load (global_load_lds_ushort) put lane i's value?  in[j] = j (uint16).  Prints the first 40 dwords of the LDS
// buffer after 64 lanes loaded in[lane] each: "lane-linear at 2 B" would show dwords (2k | (2k+1) << 16), "a dword per lane" shows k.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/ldslds2.cpp -o /tmp/ldslds2 && /tmp/ldslds2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* o, const uint16_t* in)
{
	__shared__ uint32_t buf[256];
	buf[threadIdx.x] = 0xdeadbeefu; buf[threadIdx.x + 64] = 0xdeadbeefu;
	__syncthreads();
	const uint16_t* src = in + threadIdx.x;
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)buf, 2, 0, 0);
	__builtin_amdgcn_s_waitcnt(0);
	__syncthreads();
	o[threadIdx.x] = buf[threadIdx.x]; o[threadIdx.x + 64] = buf[threadIdx.x + 64];
}
int main()
{
	uint32_t *d, h[128]; uint16_t *in, hi[256];
	for (int i = 0; i < 256; ++i) hi[i] = (uint16_t)(i + 0x100);
	hipMalloc(&d, 512); hipMalloc(&in, 512); hipMemcpy(in, hi, 512, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, in); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
	for (int i = 0; i < 40; ++i) printf("%08x ", h[i]); printf("\n... dword 64..67: %08x %08x %08x %08x\n", h[64], h[65], h[66], h[67]);
}
