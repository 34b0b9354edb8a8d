// cross-stream dependency cost on this platform: s --event--> cs --event--> s per iteration, against the same kernels on one stream
// hipcc -O2 tools/micro/xstream.cpp -o /tmp/xstream && /tmp/xstream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	float* d; const int n = 1 << 22;
	hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4);
	float* d2; hipMalloc(&d2, n * 4); hipMemset(d2, 0, n * 4);
	for (int prio = 0; prio < 2; ++prio) {
		hipStream_t s, cs; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
		hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
		hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio ? hi : 0);
		hipEvent_t e1, e2; hipEventCreateWithFlags(&e1, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
		for (int mode = 0; mode < 3; ++mode) {
			const int iters = 200;
			hipDeviceSynchronize();
			const double t0 = now();
			for (int it = 0; it < iters; ++it) {
				hipLaunchKernelGGL(k_spin, dim3(n / 256), dim3(256), 0, s, d, n);
				if (mode == 0) {                              // same stream
					hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, d2, 64 * 256);
				} else if (mode == 1) {                       // fork-join through events
					hipEventRecord(e1, s); hipStreamWaitEvent(cs, e1, 0);
					hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, cs, d2, 64 * 256);
					hipEventRecord(e2, cs); hipStreamWaitEvent(s, e2, 0);
				} else {                                      // fork, overlap a big kernel, join
					hipEventRecord(e1, s); hipStreamWaitEvent(cs, e1, 0);
					hipMemcpyAsync(d2, d2 + (n / 2), n, hipMemcpyDeviceToDevice, cs);
					hipEventRecord(e2, cs);
					hipLaunchKernelGGL(k_spin, dim3(n / 256), dim3(256), 0, s, d, n);
					hipStreamWaitEvent(s, e2, 0);
				}
			}
			hipDeviceSynchronize();
			printf("comm-stream priority %s, mode %d: %.1f us per iteration\n", prio ? "high" : "default", mode, (now() - t0) / iters * 1e6);
		}
		hipStreamDestroy(s); hipStreamDestroy(cs);
	}
	return 0;
}
