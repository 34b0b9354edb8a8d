// Does a kernel run at full speed on a HIGHEST-priority HIP stream?  One streaming copy kernel (256 MiB, 2048 x 256 threads, 16 B per
// thread and trip), alone on the device, on streams of default / highest / lowest priority; and the same while a big kernel of a
// default-priority stream occupies the chip.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/prio_bw.cpp -o /tmp/prio_bw && /tmp/prio_bw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_big(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float v = p[i]; for (int k = 0; k < 4096; ++k) v = v * 1.0001f + 1.0f; p[i] = v; } }
int main()
{
	const size_t bytes = (size_t)256 << 20, n = bytes / 16;
	uint4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
	float* d; const int nb = 1 << 24; hipMalloc(&d, nb * 4); hipMemset(d, 0, nb * 4);
	int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
	hipStream_t s0; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
	const int prios[3] = { 0, hi, lo }; const char* names[3] = { "default", "highest", "lowest" };
	for (int busy = 0; busy < 2; ++busy)
		for (int p = 0; p < 3; ++p) {
			hipStream_t s; hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prios[p]);
			hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
			float best = 1e9f;
			for (int rep = 0; rep < 5; ++rep) {
				hipDeviceSynchronize();
				if (busy) hipLaunchKernelGGL(k_big, dim3(nb / 256), dim3(256), 0, s0, d, nb);
				hipEventRecord(e0, s);
				hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, s, b, a, n);
				hipEventRecord(e1, s);
				hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				if (ms < best) best = ms;
			}
			printf("copy of 256 MiB on a %-7s-priority stream, device %s: %.3f ms = %.0f GB/s (read + write)\n", names[p], busy ? "busy with a long default-priority kernel" : "otherwise idle",
				best, 2.0 * bytes / best / 1e6);
			hipStreamDestroy(s);
		}
	return 0;
}
