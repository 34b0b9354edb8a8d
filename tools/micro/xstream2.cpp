// Why did a HIGH-priority side stream cost ~1 ms per cross-stream dependency inside the slab schedule?
// The schedule's pattern, reduced: per "round" the compute stream s runs big kernels; a side stream cs (default or highest priority)
// picks up behind an event of s, runs small kernels (the face chain / the exchange copies), and s later waits for cs's event.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/xstream2.cpp -o /tmp/xstream2 && /tmp/xstream2
// Prints microseconds per round for: priority of cs x what runs on cs (small kernels | hipMemcpyAsync D2D) x whether s keeps the
// device busy while cs works (overlap) or idles (fork-join).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_big(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float v = p[i]; for (int k = 0; k < 64; ++k) v = v * 1.0001f + 1.0f; p[i] = v; } }
__global__ void k_small(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const int n = 1 << 24;
	float *d, *d2; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4); hipMalloc(&d2, n * 4); hipMemset(d2, 0, n * 4);
	int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
	printf("stream priority range: lowest %d, highest %d\n", lo, hi);
	for (int prio = 0; prio < 3; ++prio) {
		hipStream_t s, cs;
		hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
		hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio == 0 ? 0 : (prio == 1 ? hi : lo));
		hipEvent_t e1, e2; hipEventCreateWithFlags(&e1, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
		for (int what = 0; what < 2; ++what)
			for (int overlap = 0; overlap < 2; ++overlap) {
				const int rounds = 100;
				double best = 1e9;
				for (int rep = 0; rep < 3; ++rep) {
					hipDeviceSynchronize();
					const double t0 = now();
					for (int it = 0; it < rounds; ++it) {
						hipLaunchKernelGGL(k_big, dim3(n / 256), dim3(256), 0, s, d, n);
						hipEventRecord(e1, s); hipStreamWaitEvent(cs, e1, 0);
						for (int j = 0; j < 4; ++j) {
							if (what == 0) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, cs, d2, 256 * 256);
							else hipMemcpyAsync(d2, d2 + (n / 2), 1 << 20, hipMemcpyDeviceToDevice, cs);
						}
						hipEventRecord(e2, cs);
						if (overlap) hipLaunchKernelGGL(k_big, dim3(n / 256), dim3(256), 0, s, d, n);
						hipStreamWaitEvent(s, e2, 0);
						hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d, 256 * 256);
					}
					hipDeviceSynchronize();
					const double us = (now() - t0) / rounds * 1e6;
					if (us < best) best = us;
				}
				printf("side stream priority %-7s  side work %-13s  %-9s : %8.1f us per round\n", prio == 0 ? "default" : (prio == 1 ? "highest" : "lowest"),
					what == 0 ? "4 kernels" : "4 memcpyAsync", overlap ? "overlap" : "fork-join", best);
			}
		hipStreamDestroy(s); hipStreamDestroy(cs);
	}
	// reference: the big kernel alone
	{
		hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
		hipDeviceSynchronize(); const double t0 = now();
		for (int it = 0; it < 100; ++it) hipLaunchKernelGGL(k_big, dim3(n / 256), dim3(256), 0, s, d, n);
		hipDeviceSynchronize(); printf("one big kernel alone: %.1f us\n", (now() - t0) / 100 * 1e6);
	}
	return 0;
}
