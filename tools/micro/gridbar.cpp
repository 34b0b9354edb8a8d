// Feasibility probe for a register-resident Jacobi: what does one device-wide barrier cost on
// MI355X with one workgroup per CU, and what does a per-sweep face exchange through agent-scope loads/stores add?
//   hipcc -O2 --offload-arch=gfx950 tools/micro/gridbar.cpp -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
	}
	__syncthreads();
}

// mode 0: barriers only.  mode 1: + every workgroup writes `face_dwords` dwords and reads its neighbour's (relaxed agent-scope
// accesses, which bypass the non-coherent caches), so the barrier needs no cache invalidation.  mode 2: same exchange with plain
// stores/loads fenced by __threadfence() (release/acquire of the whole L2).
__global__ __launch_bounds__(256) void k_bar(unsigned* counter, float* faces, int face_dwords, int iters, int mode, float* sink)
{
	const unsigned nwg = gridDim.x;
	const unsigned me = blockIdx.x, nb = (blockIdx.x + 1) % nwg;
	float acc = 0.0f;
	for (int it = 0; it < iters; ++it) {
		float* mine = faces + ((size_t)(it & 1) * nwg + me) * face_dwords;
		const float* theirs = faces + ((size_t)(it & 1) * nwg + nb) * face_dwords;
		if (mode == 1) for (int i = threadIdx.x; i < face_dwords; i += blockDim.x)
			__hip_atomic_store(mine + i, acc + (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (mode == 2) { for (int i = threadIdx.x; i < face_dwords; i += blockDim.x) mine[i] = acc + (float)i; __threadfence(); }
		grid_barrier(counter, (unsigned)(it + 1) * nwg);
		if (mode == 1) for (int i = threadIdx.x; i < face_dwords; i += blockDim.x)
			acc += __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (mode == 2) { __threadfence(); for (int i = threadIdx.x; i < face_dwords; i += blockDim.x) acc += theirs[i]; }
	}
	if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int nwg = prop.multiProcessorCount;
	unsigned* counter; float* faces; float* sink;
	const int max_face = 16384;
	hipMalloc(&counter, 4); hipMalloc(&faces, (size_t)2 * nwg * max_face * 4); hipMalloc(&sink, 4);
	hipMemset(faces, 0, (size_t)2 * nwg * max_face * 4);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	printf("%s: %d CUs, one 256-thread workgroup each (cooperative launch)\n", prop.name, nwg);
	for (int mode = 0; mode < 3; ++mode)
		for (int fd : { 4096, 10240, 16384 }) {
			if (mode == 0 && fd != 4096) continue;
			int iters = 200;
			for (int rep = 0; rep < 2; ++rep) {
				hipMemset(counter, 0, 4);
				void* args[] = { &counter, &faces, &fd, &iters, &mode, &sink };
				hipEventRecord(e0, 0);
				hipError_t e = hipLaunchCooperativeKernel((const void*)k_bar, dim3(nwg), dim3(256), args, 0, 0);
				hipEventRecord(e1, 0);
				if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
				float ms; hipEventElapsedTime(&ms, e0, e1);
				if (rep) printf("mode %d (%s) face %5d dwords/WG: %.2f us per barrier+exchange\n", mode,
					mode == 0 ? "barrier only" : mode == 1 ? "agent-scope relaxed ld/st" : "plain ld/st + __threadfence", mode ? fd : 0, ms * 1e3 / iters);
			}
		}
	return 0;
}
