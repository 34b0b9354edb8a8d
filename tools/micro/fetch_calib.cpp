// What FETCH_SIZE / WRITE_SIZE report for the access widths of this library's kernels (MI355X_MICROARCH.md: only 16 B per lane is
// calibrated -- "double it"; "other access widths and WRITE_SIZE are uncalibrated").  Each kernel streams a 1-GiB buffer (4 x the
// Infinity Cache) exactly once:
//   rd4 / rd8 / rd16   global_load_dword / dwordx2 / dwordx4 per lane, consecutive lanes consecutive addresses
//   dma4 / dma16       global_load_lds_dword / dwordx4 (LDS-DMA), the same
//   rd4_rows600        4 B per lane along 600-byte rows (150 floats: the scalar kernels at 150^3), rows back to back
//   wr4 / wr16         stores
//   hipcc -O2 --offload-arch=gfx950 tools/micro/fetch_calib.cpp -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- /tmp/fetch_calib ; same with WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
static const size_t BYTES = (size_t)1 << 30;

__global__ void rd4(const float* p, float* o, size_t n) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 12345.678f) *o = s; }
__global__ void rd8(const float2* p, float* o, size_t n) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float2 v = p[i]; s += v.x + v.y; } if (s == 12345.678f) *o = s; }
__global__ void rd16(const float4* p, float* o, size_t n) { float s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; } if (s == 12345.678f) *o = s; }
__global__ void dma4(const float* p, float* o, size_t n)
{
	__shared__ float buf[256];
	float s = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i), (__attribute__((address_space(3))) void*)(buf + (threadIdx.x & ~63u)), 4, 0, 0);
		__builtin_amdgcn_s_waitcnt(0);
		s += buf[threadIdx.x];
	}
	if (s == 12345.678f) *o = s;
}
__global__ void dma16(const float4* p, float* o, size_t n)
{
	__shared__ float4 buf[256];
	float s = 0;
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i), (__attribute__((address_space(3))) void*)(buf + (threadIdx.x & ~63u)), 16, 0, 0);
		__builtin_amdgcn_s_waitcnt(0);
		s += buf[threadIdx.x].x;
	}
	if (s == 12345.678f) *o = s;
}
// one 64-thread block row = 150 floats: threads 0..63 take x = 0..63, 64..127, 128..149 in three trips (as the scalar 64 x 4 kernels do)
__global__ void rd4_rows600(const float* p, float* o, size_t rows)
{
	float s = 0;
	for (size_t r = blockIdx.x * (size_t)blockDim.y + threadIdx.y; r < rows; r += (size_t)gridDim.x * blockDim.y)
		for (int x = threadIdx.x; x < 150; x += 64) s += p[r * 150 + x];
	if (s == 12345.678f) *o = s;
}
__global__ void wr4(float* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f; }
__global__ void wr16(float4* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1, 2, 3, 4); }

int main()
{
	float *d, *o;
	hipMalloc(&d, BYTES); hipMalloc(&o, 4); hipMemset(d, 0, BYTES);
	const dim3 g(256 * 8), b(256);
	for (int rep = 0; rep < 2; ++rep) {
		hipLaunchKernelGGL(rd4, g, b, 0, 0, d, o, BYTES / 4);
		hipLaunchKernelGGL(rd8, g, b, 0, 0, (const float2*)d, o, BYTES / 8);
		hipLaunchKernelGGL(rd16, g, b, 0, 0, (const float4*)d, o, BYTES / 16);
		hipLaunchKernelGGL(dma4, g, b, 0, 0, d, o, BYTES / 4);
		hipLaunchKernelGGL(dma16, g, b, 0, 0, (const float4*)d, o, BYTES / 16);
		hipLaunchKernelGGL(rd4_rows600, g, dim3(64, 4), 0, 0, d, o, BYTES / 600);
		hipLaunchKernelGGL(wr4, g, b, 0, 0, d, BYTES / 4);
		hipLaunchKernelGGL(wr16, g, b, 0, 0, (float4*)d, BYTES / 16);
	}
	hipDeviceSynchronize();
	printf("bytes per kernel: %zu (rd4_rows600: %zu)\n", BYTES, BYTES / 600 * 600);
	return 0;
}
