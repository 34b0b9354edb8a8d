// issue_rate.cpp -- what one wave per SIMD pays per instruction (gfx950): streams of independent instructions of one kind, timed with
// s_memtime.  One 256-thread workgroup per CU (the LDS is declared full so that no second one fits), 256 workgroups.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/issue_rate.cpp -o gpurun_out/issue_rate && gpurun_out/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k(unsigned long long* out, int iters)
{
	__shared__ float big[38 * 1024];                       // 152 KiB: one workgroup per CU
	float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 p0 = { 1, 2 }, p1 = { 3, 4 }, p2 = { 5, 6 }, p3 = { 7, 8 }, p4 = { 1, 1 }, p5 = { 2, 2 }, p6 = { 3, 3 }, p7 = { 4, 4 };
	big[threadIdx.x] = a0;
	__syncthreads();
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) {
		if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) :);) }
		if (KIND == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4));) }
		if (KIND == 2) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
		if (KIND == 3) { REP16(asm volatile("v_pk_mov_b32 %0, %4, %5 op_sel:[1,0]\n v_pk_mov_b32 %1, %4, %5 op_sel:[1,0]\n v_pk_mov_b32 %2, %4, %5 op_sel:[1,0]\n v_pk_mov_b32 %3, %4, %5 op_sel:[1,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4), "v"(p5));) }
		if (KIND == 4) { REP16(asm volatile("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
		if (KIND == 5) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p4));) }
		if (KIND == 6) { REP16(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %1, a0\n v_accvgpr_write_b32 a1, %2\n v_accvgpr_read_b32 %3, a1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "a0", "a1");) }
		if (KIND == 7) { REP16(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "s20", "s21", "s22", "s23", "scc");) }
		if (KIND == 8) { REP16(asm volatile("v_add_f32 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %4\n s_add_u32 s21, s21, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4) : "s20", "s21", "scc");) }
		if (KIND == 9) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_mov_b32 %2, %3\n v_pk_add_f32 %1, %1, %4\n v_mov_b32 %2, %3" : "+v"(p0), "+v"(p1), "+v"(a0) : "v"(a1), "v"(p4));) }
		if (KIND == 10) { REP16(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %2" : "+v"(p0), "+v"(p1) : "v"(p4));) }   // dependent chain
		if (KIND == 11) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(a1));) }   // dependent chain
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
	if (a0 + a1 + a2 + a3 + a4 + p0.x + p1.x + p2.x + p3.x + p0.y == 12345.678f) out[1] = 1;
}

int main()
{
	unsigned long long* d;
	hipMalloc(&d, 16);
	const char* names[] = { "v_add_f32", "v_pk_add_f32", "v_mov_b32", "v_pk_mov_b32", "v_mov_b32_dpp", "v_pk_mul_f32", "v_accvgpr_write+read", "s_add_u32",
		"v_add_f32 + s_add_u32 alternating", "v_pk_add_f32 + v_mov_b32 alternating", "v_pk_add_f32 dependent chain", "v_add_f32 dependent chain" };
	const int iters = 2000;
	for (int threads = 256; threads <= 1024; threads *= 2) {
	printf("---- %d waves per SIMD\n", threads / 256);
	for (int kind = 0; kind < 12; ++kind) {
		float ms = 0;
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0, 0);
			switch (kind) {
#define L(K) case K: if (threads == 256) hipLaunchKernelGGL((k<K, 256>), dim3(256), dim3(256), 0, 0, d, iters); else if (threads == 512) hipLaunchKernelGGL((k<K, 512>), dim3(256), dim3(512), 0, 0, d, iters); else hipLaunchKernelGGL((k<K, 1024>), dim3(256), dim3(1024), 0, 0, d, iters); break;
			L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11)
			}
			hipEventRecord(e1, 0);
			hipDeviceSynchronize();
			hipEventElapsedTime(&ms, e0, e1);
		}
		unsigned long long h[2];
		hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
		printf("%-40s %6.2f counter ticks, %6.3f ns per instruction and wave (kernel %.1f us)\n", names[kind], (double)h[0] / (iters * 64.0), ms * 1e6 / (iters * 64.0), ms * 1e3);
	}
	}
	return 0;
}
