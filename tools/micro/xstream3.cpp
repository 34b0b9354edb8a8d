// The slab schedule's overlapped pressure round (jacobi_overlapped, fx_api.cpp) reduced to its stream / event skeleton, to find
// out why a HIGHEST-priority comm stream made the library's step 2x slower although a plain two-stream
// fork-join shows no priority effect at all (tools/micro/xstream2.cpp):
//   compute stream s   3 big kernels per round (the 2nd waits for the chain's first sweep), then waits for the chain's end, 2 small
//                      copy kernels, records ev_int
//   face stream fs     waits ev_int (previous round of s) and ev_done (previous exchange), 9 small kernels, records ev_face1 after
//                      the first and ev_ready after the last
//   comm stream cs     waits ev_ready, 4 small copy kernels, records ev_done            <- priority varied
//   hipcc -O2 --offload-arch=gfx950 tools/micro/xstream3.cpp -o /tmp/xstream3 && /tmp/xstream3
// Variants: priority of cs (default / highest), priority of fs (default / highest), and whether the events are created with
// hipEventDisableTiming.  Prints microseconds per round.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_big(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float v = p[i]; for (int k = 0; k < 64; ++k) v = v * 1.0001f + 1.0f; p[i] = v; } }
__global__ void k_small(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const int n = 1 << 24;
	float *d, *d2, *d3; hipMalloc(&d, n * 4); hipMemset(d, 0, n * 4); hipMalloc(&d2, n * 4); hipMemset(d2, 0, n * 4); hipMalloc(&d3, n * 4); hipMemset(d3, 0, n * 4);
	int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
	for (int variant = 0; variant < 6; ++variant) {
		const int pcs = (variant == 1 || variant == 3 || variant == 5) ? hi : 0, pfs = (variant == 2 || variant == 3) ? hi : 0;
		const unsigned evflags = variant >= 4 ? hipEventDefault : hipEventDisableTiming;
		hipStream_t s, fs, cs;
		hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
		hipStreamCreateWithPriority(&fs, hipStreamNonBlocking, pfs);
		hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, pcs);
		hipEvent_t ev_int, ev_face1, ev_ready, ev_done;
		hipEventCreateWithFlags(&ev_int, evflags); hipEventCreateWithFlags(&ev_face1, evflags); hipEventCreateWithFlags(&ev_ready, evflags); hipEventCreateWithFlags(&ev_done, evflags);
		double best = 1e9;
		for (int rep = 0; rep < 3; ++rep) {
			const int rounds = 50;
			hipDeviceSynchronize();
			const double t0 = now();
			hipEventRecord(ev_int, s);
			bool in_flight = false;
			for (int it = 0; it < rounds; ++it) {
				hipStreamWaitEvent(fs, ev_int, 0);
				if (in_flight) hipStreamWaitEvent(fs, ev_done, 0);
				for (int j = 0; j < 9; ++j) {
					hipLaunchKernelGGL(k_small, dim3(512), dim3(256), 0, fs, d2, 512 * 256);
					if (j == 0) hipEventRecord(ev_face1, fs);
				}
				hipEventRecord(ev_ready, fs);
				hipStreamWaitEvent(cs, ev_ready, 0);
				for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, cs, d3, 256 * 256);
				hipEventRecord(ev_done, cs);
				in_flight = true;
				for (int j = 0; j < 3; ++j) {
					if (j == 1) hipStreamWaitEvent(s, ev_face1, 0);
					hipLaunchKernelGGL(k_big, dim3(n / 256), dim3(256), 0, s, d, n);
				}
				hipStreamWaitEvent(s, ev_ready, 0);
				for (int j = 0; j < 2; ++j) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, d, 256 * 256);
				hipEventRecord(ev_int, s);
			}
			hipStreamWaitEvent(s, ev_done, 0);
			hipDeviceSynchronize();
			const double us = (now() - t0) / rounds * 1e6;
			if (us < best) best = us;
		}
		printf("comm stream %-7s  face stream %-7s  events %-14s : %8.1f us per round\n", pcs ? "highest" : "default", pfs ? "highest" : "default",
			variant >= 4 ? "with timing" : "disable-timing", best);
		hipStreamDestroy(s); hipStreamDestroy(fs); hipStreamDestroy(cs);
	}
	return 0;
}
