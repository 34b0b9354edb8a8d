// fx_jacobi_strip3.hip -- THREE lock-step Jacobi sweeps per launch: the strip kernel of fx_jacobi_strip.hip with the LDS
// as a second register file.
//
// Restates CSPoisson.hlsli:8-26 (/root/reference/FluidX12/Content/Shaders/) like k_jacobi_v4; per-cell arithmetic and
// association order are unchanged, so three fused sweeps are bit-identical to three single ones.
//
// Why LDS now: a wave64 = one strip of R = 4 full-x rows streaming along z.  Level l (0 = input ... 3 = output) needs a
// 3-plane window of R + 2(3 - l) rows, plus the b planes of three different z offsets: 116 float4 rows = 464 registers per
// lane before temporaries -- the all-register version spilled and ran slower than two sweeps (profiles/archive/r01c_jacobi_strip.txt).
// Here only the windows of levels 1 and 2 (8 + 6 rows x 3 planes, rotated by NAME as in k_jacobi_strip2u) and the plane
// in flight live in registers; the input window's older two planes (2 x 10 rows) and the b planes waiting for sweeps 2
// and 3 (3 x 6 rows) live in the wave's private 38-KiB slice of the LDS (4 waves x 38 KiB = 152 of the CU's 160 KiB; no
// barrier, no sharing: each wave only ever touches its own slice).  A ds_read_b128 / ds_write_b128 moves a whole row
// (16 B per lane) per instruction, where an AGPR spill moves 4 B per instruction.  The LDS slots rotate at run time (three
// byte offsets swapped per step), so the z loop still unrolls by three only.
//
// Per z step q (plane q of the input arrives in registers, prefetched during step q - 1):
//   sweep 1  level-1 plane q-1, rows y0-2 .. y0+R+1, from input planes q-2 (LDS), q-1 (LDS), q (registers) and b[q-1] (registers)
//            then the input plane and rows 1..6 of b[q-1] go to their LDS slots and the prefetch of plane q+1 / b[q] is issued
//   sweep 2  level-2 plane q-2, rows y0-1 .. y0+R, from the level-1 window and b[q-2] (LDS)
//   sweep 3  output plane q-3, rows y0 .. y0+R-1, from the level-2 window and b[q-3] (LDS); stored if inside the chunk
// Traffic: p and b are read once and p''' written once per THREE sweeps.
//
// What it measures (256^3, MI355X): 19.2 us per sweep against 18.5 for the two-sweep kernel -- no gain, and that is the
// instructive part.  Both kernels hold ONE plane in flight per wave (14 / 18 KiB) and their z step lasts about one loaded
// memory round trip: two sweeps 1.85 us per step = exactly the 6.4 TB/s the fabric sustains with 1024 such waves; here the
// prefetch can only be issued after sweep 1 has consumed the arrived plane, so a step costs sweep 1 + a round trip = 2.6 us.
// Issuing it at the top of the step (arrived plane parked in the LDS first, sweep 1 reading it back) needs the in-flight
// 18 rows live across the whole step: 330 registers, and the compiler then keeps that plane in SCRATCH, whose reloads
// serialise with the prefetch (33 us per sweep).  A third input slot in the LDS would fix it and does not fit: 48 rows
// against the 40 a wave can have.  So this kernel serves jacobi_fuse = 3 (23 us -> 19.2 us against the all-register
// version) and the default stays at two sweeps.
//
// Later in round 1 the x-neighbour shuffles (ds_bpermute) became DPP wave shifts: 19.2 -> 17.3 us per sweep, level with
// the two-sweep kernel's 17.8.  The double-buffered variant was tried again on top of that (two register buffers for
// the plane in flight alternating by step parity, z loop unrolled by six, loads issued at the top of the step): it
// compiles without scratch (436 VGPR + AGPR) and is bit-exact, but runs 21.2 us per sweep -- the allocator parks the
// buffer that is live across the whole step in AGPRs, loads into VGPRs and copies over, so an `s_waitcnt vmcnt` lands
// directly behind the loads (1542 v_accvgpr moves per six steps against 240 per three here) and the prefetch distance
// is gone.  Getting the distance back needs loads that target AGPRs directly or land in the LDS
// (global_load_lds_dwordx4) with a fourth input slot, which the 160 KiB do not have at R = 4.
// The first of those was tried too: inline-assembly loads straight into accumulator registers ("+a" operands, one
// s_waitcnt that takes every in-flight register as an in/out operand, AGPR -> VGPR moves at the top of the next step).
// The ISA came out as intended -- loads issued at the top of the step, one wait per step -- and it was SLOWER (56.4
// against 49.8 us per launch; it also broke one bit-exactness test, the allocator renames the in-flight registers between
// the unrolled phases).  So the step is not waiting for memory: at one wave per SIMD it is issue bound.  Per z step the
// wave issues ~1080 instructions (250 packed FP32 adds/multiplies, 150 v_mov that line the x-shifted operands up for
// v_pk_add_f32, 80 v_accvgpr moves, 66 LDS and 24 vector-memory instructions, 50 selects, 36 DPP shifts), i.e. ~4300
// cycles = the 2.25 us it takes.  What this kernel needs next is fewer instructions, not more memory-level parallelism.
// One attempt at that: an even/odd register layout (c0, c2, c1, c3), in which the left neighbours of (c1, c3) and the right
// neighbours of (c0, c2) are aligned register pairs as they stand.  Written at the source level (permute at the global
// loads/stores, relax4 on the permuted vectors) the compiler answers with MORE moves (1184 v_mov + 627 v_accvgpr per three
// steps against 553 + 240): the permutation of a loaded 128-bit tuple costs four moves and the pairs are not kept
// aligned.  It would take hand-scheduled assembly.
// And the LDS-direct route (global_load_lds_dwordx4, checked on the device by tools/micro/ldslds.cpp): three rotating input
// slots + a b landing slot in the same 38 KiB, the rows of plane q-2 and the three b planes in registers, every LDS read in
// inline assembly so that the only `s_waitcnt vmcnt(0)` of a step is the one at its top -- 11 waits per three steps in the
// ISA, no ds_write left, and the SAME 44.2 us per launch (390 registers, three times the v_accvgpr traffic).  Together
// with the AGPR prefetch this settles it: hiding the memory latency buys nothing here, the kernel's time is its ~580
// issued instructions per z step (rocprofv3: 9.9 k VALU + 1.3 k scalar + 1.0 k LDS + 0.45 k vector-memory per wave and
// launch, a wave issuing 61 % of its cycles).
#include "fx_internal.h"
#include "fx_pk.h"
#include <climits>
#include <cstdlib>

namespace fx {

namespace {

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// ((((((L - b) + R) + U) + D) + F) + B) * (1/6) on a float4 column; x neighbours by DPP wave shifts, the x-shifted operand pairs
// built by v_pk_mov_b32 (fx_pk.h: 8 -> 5 instructions of plumbing per 14 of arithmetic; k_jacobi_strip3c 41.2 -> 40.7 us per launch)
__device__ __forceinline__ float4 relax4(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb, bool x_first, bool x_last)
{
	(void)x_first; (void)x_last;
	return relax4_pairs(c, U, D, F, Bk, bb, 0.0f, true, true);
}

__device__ __forceinline__ int xcd_index3(int n, int remap)
{
	int t = (int)blockIdx.x;
	if (remap) {
		const int q = n >> 3, r = n & 7;
		const int xcd = t & 7, j = t >> 3;
		t = xcd * q + min(xcd, r) + j;
	}
	return t;
}

constexpr int R3 = 4;                       // output rows per strip
constexpr int LDS_P0_ROWS = R3 + 6;         // one input plane: rows y0-3 .. y0+R+2
constexpr int LDS_B_ROWS = R3 + 2;          // one stored b plane: rows y0-1 .. y0+R
constexpr int LDS_ROWS_PER_WAVE = 2 * LDS_P0_ROWS + 3 * LDS_B_ROWS;   // 38 rows of 1 KiB

// a 32-bit row offset the optimiser may not widen ahead of time: hoisted out of the z loop, zext(offset) becomes a 64-bit
// register pair and every access a v_lshl_add_u64 + `global_load ... off`; kept 32 bits wide at the use it folds into the
// `global_load v, v_offset, s[base]` form
__device__ __forceinline__ uint32_t opaque32(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

// The step counters of k_jacobi_strip3c's hand-overs live in the LDS and are polled / published with inline-assembly DS
// instructions: as `volatile` C++ accesses they make the compiler put `s_waitcnt vmcnt(0)` in front of every poll, i.e. the wave
// waits for its own prefetch (the next input plane, in flight since sweep 1) in the middle of the step.  A wave's LDS operations
// execute in order, so "data rows, then counter" needs no fence; the asm's "memory" clobber keeps the compiler from moving the
// rows' reads / writes across it.  (Measured neutral for k_jacobi_strip3c, 43.5 us either way, and WORSE for k_jacobi_strip3h --
// 337 -> 359 us: its scalar seam loads share lgkmcnt with the DS poll -- which therefore keeps the volatile form.  Tried again with
// the seam cells fetched by VECTOR loads, so that no wait of the step except the next sweep 1's touches the prefetch: 347 -> 372-381 us
// at 512^3.  The volatile form's `s_waitcnt vmcnt(0)` at hand-over 2 -- the wave stands until its own prefetch has landed, half a
// step after issuing it -- is what keeps the two halves of a row in step; without it they drift and spin at the hand-overs.)
__device__ __forceinline__ int lds_peek(uint32_t lds_byte_addr)
{
	int v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_byte_addr) : "memory");
	return v;
}
__device__ __forceinline__ void lds_post(uint32_t lds_byte_addr, int v)
{
	asm volatile("ds_write_b32 %0, %1" :: "v"(lds_byte_addr), "v"(v) : "memory");
}
// Every hand-over wait of this file is BOUNDED like k_jacobi_strip4o's (a partner is a z step away, ~2 us; 65 536 polls are ~10 ms) and
// LOUD when it runs out: it raises this word, which fx_synchronize reads behind the device (strip3_fault_take) and returns as
// FX_E_DEVICE instead of a hung device or a silently wrong pressure field (these loops span unbounded).
__device__ unsigned g_strip3_fault;
constexpr int kWaitSpins3 = 1 << 16;
// (inline, and a plain atomic without a return value: a CALL in these kernels would cost the windows their registers)
__device__ __forceinline__ void strip3_raise_fault() { (void)__hip_atomic_fetch_or(&g_strip3_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lds_wait_ge(uint32_t lds_byte_addr, int v)
{
	for (int spins = 0; lds_peek(lds_byte_addr) < v; ++spins) {
		if (__builtin_expect(spins > kWaitSpins3, 0)) { strip3_raise_fault(); break; }
		__builtin_amdgcn_s_sleep(1);
	}
}
// ... and the `volatile` form k_jacobi_strip3h keeps (see above)
__device__ __forceinline__ void lds_wait_ge_volatile(const int* flag, int v)
{
	for (int spins = 0; *reinterpret_cast<const volatile int*>(flag) < v; ++spins) {
		if (__builtin_expect(spins > kWaitSpins3, 0)) { strip3_raise_fault(); break; }
		__builtin_amdgcn_s_sleep(1);
	}
}

// counter and row in ONE LDS round trip: both reads are issued back to back (LDS operations of a wave execute in order, and the
// partner wrote the row before the counter: a counter that is high enough vouches for the row read behind it); only a partner that is
// late costs a second trip.  Saves one ~100-cycle LDS latency per hand-over, which one wave per SIMD cannot hide.
typedef float fx_q4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_wait_read(uint32_t flag_byte_addr, int need, uint32_t row_byte_addr)
{
	int f;
	fx_q4 d;
	for (int spins = 0;; ++spins) {
		asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f), "=&v"(d) : "v"(flag_byte_addr), "v"(row_byte_addr) : "memory");
		if (f >= need) break;
		if (__builtin_expect(spins > kWaitSpins3, 0)) { strip3_raise_fault(); break; }
		__builtin_amdgcn_s_sleep(1);
	}
	return make_float4(d.x, d.y, d.z, d.w);
}

// row `r` of an LDS slot whose first row starts `slot` float4s into the wave's slice
#define FX_LDS(slot, r) lds[(slot) + (r) * 64]

#define FX_STRIP3_STEP(PH) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	/* ---- sweep 1: level-1 plane q-1, rows j <-> y0-2+j; input rows i <-> y0-3+i ------------------------------- */ \
	if (q == 0) {                                   /* input plane -1 := plane 0, once (clamped front neighbour) */ \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) FX_LDS(s_ctr, i) = NP[i]; \
	} \
	if (q - 1 == g.Zg) {                            /* level-1 plane Zg := plane Zg-1 */ \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) P1[NEW][j] = P1[CTR][j]; \
	} else { \
		/* all 18 LDS rows first: with one wave per SIMD a ds_read that is issued next to its use costs its whole latency */ \
		float4 C_[R3 + 6], F_[R3 + 4]; \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) C_[i] = FX_LDS(s_ctr, i); \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) F_[j] = FX_LDS(s_old, j + 1); \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) \
			P1[NEW][j] = relax4(C_[j + 1], C_[j], C_[j + 2], F_[j], NP[j + 1], NB[j], x_first, x_last); \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) P1[CTR][j] = P1[NEW][j]; \
		} \
	} \
	/* ---- the b rows sweeps 2 and 3 will need (slots untouched by the writes below), read now so that they arrive behind the \
	   writes and the prefetch instead of in front of each update ---- */ \
	float4 B2_[R3 + 2], B3_[R3]; \
	_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) B2_[k] = FX_LDS(s_b2, k);        /* b[q-2]: becomes s_b3 in the rotation */ \
	_Pragma("unroll") for (int m = 0; m < R3; ++m) B3_[m] = FX_LDS(s_b3, m + 1);        /* b[q-3]: becomes s_bfree */ \
	/* ---- the plane in flight moves to the LDS (over the input plane q-2, dead now); b[q-1] rows 1..6 to the free b slot ---- */ \
	_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) FX_LDS(s_old, i) = NP[i]; \
	_Pragma("unroll") for (int i = 0; i < R3 + 2; ++i) FX_LDS(s_bfree, i) = NB[i + 1]; \
	{ const int t_ = s_old; s_old = s_ctr; s_ctr = t_; }            /* plane q is next step's centre */ \
	{ const int t_ = s_bfree; s_bfree = s_b3; s_b3 = s_b2; s_b2 = t_; } /* after this: s_b2 = b[q-1], s_b3 = b[q-2], s_bfree = b[q-3] */ \
	if (q + 1 <= q_load_last) {                     /* prefetch input plane q+1; past the last plane NP keeps plane zhi */ \
		const char* pb_ = reinterpret_cast<const char*>(p_in + (size_t)g.lz(q + 1) * plane); \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) NP[i] = *reinterpret_cast<const float4*>(pb_ + opaque32(roff[i])); \
	} \
	if (q <= b_load_last) {                         /* b[q] for the next step's sweep 1 */ \
		const char* bb_ = reinterpret_cast<const char*>(b + (size_t)g.lz(q) * plane); \
		_Pragma("unroll") for (int i = 0; i < R3 + 4; ++i) NB[i] = *reinterpret_cast<const float4*>(bb_ + opaque32(roff[i + 1])); \
	} \
	/* ---- sweep 2: level-2 plane q-2, rows k <-> y0-1+k; b[q-2] is s_b3 (rows y0-1 ..) ------------------------------- */ \
	if (q - 2 == g.Zg) { \
		_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) P2[NEW][k] = P2[CTR][k]; \
	} else { \
		_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) { \
			const float4 c_ = P1[CTR][k + 1]; \
			float4 u_ = P1[CTR][k], d_ = P1[CTR][k + 2]; \
			if (k == 1 && y0 == 0) u_ = c_;                             /* rows outside the domain hold no data */ \
			if (k == R3 && y0 + R3 >= g.Y) d_ = c_; \
			P2[NEW][k] = relax4(c_, u_, d_, P1[OLD][k + 1], P1[NEW][k + 1], B2_[k], x_first, x_last); \
		} \
		if (q - 2 == 0) { \
			_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) P2[CTR][k] = P2[NEW][k]; \
		} \
	} \
	/* ---- sweep 3: output plane q-3, rows m <-> y0+m; b[q-3] is s_bfree (its rows 1..4) ------------------------------- */ \
	if (q - 3 >= zb && q - 3 < ze) { \
		char* ob_ = reinterpret_cast<char*>(p_out + (size_t)g.lz(q - 3) * plane); \
		_Pragma("unroll") for (int m = 0; m < R3; ++m) { \
			const float4 c_ = P2[CTR][m + 1]; \
			float4 u_ = P2[CTR][m], d_ = P2[CTR][m + 2]; \
			if (m == 0 && y0 == 0) u_ = c_; \
			if (m == R3 - 1 && y0 + R3 >= g.Y) d_ = c_; \
			const float4 x_ = relax4(c_, u_, d_, P2[OLD][m + 1], P2[NEW][m + 1], B3_[m], x_first, x_last); \
			if (strip_live) *reinterpret_cast<float4*>(ob_ + opaque32(roff[m + 3])) = x_;   /* rows y0 .. y0+3 of a live strip are never clamped */ \
		} \
	} \
} while (0)

__global__ __launch_bounds__(256, 1) void k_jacobi_strip3(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	__shared__ float4 lds_all[4 * LDS_ROWS_PER_WAVE * 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int lx = lane;                                              // X = 256: one row = one wave
	float4* lds = lds_all + wave * (LDS_ROWS_PER_WAVE * 64) + lane;
	const int tile = xcd_index3(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int y0 = (grp * 4 + wave) * R3;
	const bool strip_live = y0 < g.Y;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 3, g.zlo), q_last = ze - 1 + 3, q_load_last = min(q_last, g.zhi);
	const int b_load_last = min(q_last - 1, g.zhi);
	const bool x_first = lx == 0, x_last = lx == 63;
	const size_t plane = g.plane();

	// byte offsets of the strip's rows inside a plane, 32 bits each: the plane base is wave-uniform (SGPR pair), so every access
	// is `global_load/store v, v_offset, s[base]` with no 64-bit address arithmetic per row
	uint32_t roff[R3 + 6];
#pragma unroll
	for (int i = 0; i < R3 + 6; ++i) roff[i] = ((uint32_t)min(max(y0 - 3 + i, 0), g.Y - 1) * (uint32_t)g.X + 4u * (uint32_t)lx) * 4u;

	// LDS slots (float4 offsets into the wave's slice): two input planes, three b planes
	int s_ctr = 0, s_old = LDS_P0_ROWS * 64;
	int s_b2 = 2 * LDS_P0_ROWS * 64, s_b3 = s_b2 + LDS_B_ROWS * 64, s_bfree = s_b3 + LDS_B_ROWS * 64;

	float4 P1[3][R3 + 4], P2[3][R3 + 2], NP[R3 + 6], NB[R3 + 4];
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R3 + 4; ++i) P1[k][i] = zero;
#pragma unroll
		for (int i = 0; i < R3 + 2; ++i) P2[k][i] = zero;
	}
#pragma unroll
	for (int i = 0; i < LDS_ROWS_PER_WAVE; ++i) lds[i * 64] = zero;
	{
		const char* pb = reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(qs, q_load_last)) * plane);
#pragma unroll
		for (int i = 0; i < R3 + 6; ++i) NP[i] = *reinterpret_cast<const float4*>(pb + roff[i]);
		// b[qs - 1] for the first step's sweep 1 (clamped into the present planes: its level-1 plane is never used when qs - 1 < zlo)
		const char* bbase = reinterpret_cast<const char*>(b + (size_t)g.lz(min(max(qs - 1, g.zlo), g.zhi)) * plane);
#pragma unroll
		for (int i = 0; i < R3 + 4; ++i) NB[i] = *reinterpret_cast<const float4*>(bbase + roff[i + 1]);
	}
	int q = qs;
	for (;;) {
		FX_STRIP3_STEP(0);
		if (++q > q_last) break;
		FX_STRIP3_STEP(1);
		if (++q > q_last) break;
		FX_STRIP3_STEP(2);
		if (++q > q_last) break;
	}
}
#undef FX_STRIP3_STEP

// X = 512: the 256-wide recipe on half a row per wave.  The wall side of the wave clamps as before; on the cut side the
// neighbour cell belongs to the partner wave (the other half of the same rows, the next wave of the same workgroup) and
// (round 2: 386 -> 353 us per launch at 512^3 by handing the edge cells over wait-read-publish AFTER the sweep that produced them,
// with a counter per level, instead of waiting for the partner at the top of the step: a wave may now run a step ahead)
// comes in as `edge`: the input level from memory (a wave-uniform scalar load per row), the first- and second-sweep
// levels from a 512-byte LDS mailbox the partner filled one z step earlier; a per-wave LDS counter of the last published
// step orders it (a wave waits only while its partner is more than a step behind; a workgroup barrier per step cost 3 %).
template <bool right_half>
__device__ __forceinline__ float4 relax4_h(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb, float edge)
{
	// compile-time: the step is expanded once per half; the lane at the cut takes `edge`, the lane at the wall its own cell
	return relax4_pairs(c, U, D, F, Bk, bb, edge, !right_half, right_half);
}

#define FX_STRIP3H_STEP(PH, RIGHT, S1, S2, S3) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	/* ---- sweep 1: level-1 plane q-1, rows j <-> y0-2+j; input rows i <-> y0-3+i ------------------------------- */ \
	if (q == 0) {                                   /* input plane -1 := plane 0, once (clamped front neighbour) */ \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) FX_LDS(s_ctr, i) = NP[i]; \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) E0c[i] = E0n[i]; \
	} \
	if (!(S1)) {                                    /* (a fill step: see the kernel) */ \
	} else if (q - 1 == g.Zg) {                     /* level-1 plane Zg := plane Zg-1 */ \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) P1[NEW][j] = P1[CTR][j]; \
	} else { \
		/* all 18 LDS rows first: with one wave per SIMD a ds_read that is issued next to its use costs its whole latency */ \
		float4 C_[R3 + 6], F_[R3 + 4]; \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) C_[i] = FX_LDS(s_ctr, i); \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) F_[j] = FX_LDS(s_old, j + 1); \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) \
			P1[NEW][j] = relax4_h<RIGHT>(C_[j + 1], C_[j], C_[j + 2], F_[j], NP[j + 1], NB[j], E0c[j + 1]); \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) P1[CTR][j] = P1[NEW][j]; \
		} \
	} \
	/* hand-over 1: the partner half-row wave's first-sweep cells next to the cut of ITS step q-1 (plane q-2, for my sweep 2), then \
	   mine of this step (plane q-1).  Order as in k_jacobi_strip3c: wait, read, only then publish data and counter -- a wave that \
	   sees my counter at q knows I have read what it wrote two steps ago into the slot it reuses; a wave may run a step ahead */ \
	float e1_[R3 + 4]; \
	if (S1) { \
		lds_wait_ge_volatile(xflag + (wave ^ 1), q - 1); \
		asm volatile("" ::: "memory"); \
		{ \
			const float* xr_ = xbuf + (((q - 1) & 1) * WPG + (wave ^ 1)) * 16; \
			_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) e1_[j] = xr_[j]; \
		} \
		if (lane == edge_lane) { \
			float* xw_ = xbuf + ((q & 1) * WPG + wave) * 16; \
			_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) xw_[j] = (RIGHT) ? P1[NEW][j].x : P1[NEW][j].w; \
			asm volatile("" ::: "memory"); \
			*reinterpret_cast<volatile int*>(xflag + wave) = q;      /* LDS operations of a wave execute in order: the data is there before the counter */ \
		} \
	} else { \
		_Pragma("unroll") for (int j = 0; j < R3 + 4; ++j) e1_[j] = 0.0f; \
	} \
	/* ---- the b rows sweeps 2 and 3 will need (slots untouched by the writes below), read now so that they arrive behind the \
	   writes and the prefetch instead of in front of each update ---- */ \
	float4 B2_[R3 + 2], B3_[R3]; \
	_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) B2_[k] = FX_LDS(s_b2, k);        /* b[q-2]: becomes s_b3 in the rotation */ \
	_Pragma("unroll") for (int m = 0; m < R3; ++m) B3_[m] = FX_LDS(s_b3, m + 1);        /* b[q-3]: becomes s_bfree */ \
	/* ---- the plane in flight moves to the LDS (over the input plane q-2, dead now); b[q-1] rows 1..6 to the free b slot ---- */ \
	_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) FX_LDS(s_old, i) = NP[i]; \
	_Pragma("unroll") for (int i = 0; i < R3 + 2; ++i) FX_LDS(s_bfree, i) = NB[i + 1]; \
	{ const int t_ = s_old; s_old = s_ctr; s_ctr = t_; }            /* plane q is next step's centre */ \
	{ const int t_ = s_bfree; s_bfree = s_b3; s_b3 = s_b2; s_b2 = t_; } /* after this: s_b2 = b[q-1], s_b3 = b[q-2], s_bfree = b[q-3] */ \
	_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) E0c[i] = E0n[i];              /* plane q is next step's centre */ \
	if (q + 1 <= q_load_last) {                     /* prefetch input plane q+1; past the last plane NP keeps plane zhi */ \
		const char* pb_ = reinterpret_cast<const char*>(p_in + (size_t)g.lz(q + 1) * plane); \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) E0n[i] = *reinterpret_cast<const float*>(pb_ + soff[i]);   /* uniform address: scalar load */ \
		_Pragma("unroll") for (int i = 0; i < R3 + 6; ++i) NP[i] = *reinterpret_cast<const float4*>(pb_ + opaque32(roff[i])); \
	} \
	if (q <= b_load_last) {                         /* b[q] for the next step's sweep 1 */ \
		const char* bb_ = reinterpret_cast<const char*>(b + (size_t)g.lz(q) * plane); \
		_Pragma("unroll") for (int i = 0; i < R3 + 4; ++i) NB[i] = *reinterpret_cast<const float4*>(bb_ + opaque32(roff[i + 1])); \
	} \
	/* ---- sweep 2: level-2 plane q-2, rows k <-> y0-1+k; b[q-2] is s_b3 (rows y0-1 ..) ------------------------------- */ \
	if (!(S2)) { \
	} else if (q - 2 == g.Zg) { \
		_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) P2[NEW][k] = P2[CTR][k]; \
	} else { \
		_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) { \
			const float4 c_ = P1[CTR][k + 1]; \
			float4 u_ = P1[CTR][k], d_ = P1[CTR][k + 2]; \
			if (k == 1 && y0 == 0) u_ = c_;                             /* rows outside the domain hold no data */ \
			if (k == R3 && y0 + R3 >= g.Y) d_ = c_; \
			P2[NEW][k] = relax4_h<RIGHT>(c_, u_, d_, P1[OLD][k + 1], P1[NEW][k + 1], B2_[k], e1_[k + 1]); \
		} \
		if (q - 2 == 0) { \
			_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) P2[CTR][k] = P2[NEW][k]; \
		} \
	} \
	/* hand-over 2: the second-sweep cells (partner's plane q-3 for my sweep 3; mine of plane q-2), same order, their own counter */ \
	float e2_[R3 + 2]; \
	if (S2) { \
		lds_wait_ge_volatile(xflag + WPG + (wave ^ 1), q - 1); \
		asm volatile("" ::: "memory"); \
		{ \
			const float* xr_ = xbuf + (((q - 1) & 1) * WPG + (wave ^ 1)) * 16 + 8; \
			_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) e2_[k] = xr_[k]; \
		} \
		if (lane == edge_lane) { \
			float* xw_ = xbuf + ((q & 1) * WPG + wave) * 16 + 8; \
			_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) xw_[k] = (RIGHT) ? P2[NEW][k].x : P2[NEW][k].w; \
			asm volatile("" ::: "memory"); \
			*reinterpret_cast<volatile int*>(xflag + WPG + wave) = q; \
		} \
	} else { \
		_Pragma("unroll") for (int k = 0; k < R3 + 2; ++k) e2_[k] = 0.0f; \
	} \
	/* ---- sweep 3: output plane q-3, rows m <-> y0+m; b[q-3] is s_bfree (its rows 1..4) ------------------------------- */ \
	if ((S3) && q - 3 >= zb && q - 3 < ze) { \
		char* ob_ = reinterpret_cast<char*>(p_out + (size_t)g.lz(q - 3) * plane); \
		_Pragma("unroll") for (int m = 0; m < R3; ++m) { \
			const float4 c_ = P2[CTR][m + 1]; \
			float4 u_ = P2[CTR][m], d_ = P2[CTR][m + 2]; \
			if (m == 0 && y0 == 0) u_ = c_; \
			if (m == R3 - 1 && y0 + R3 >= g.Y) d_ = c_; \
			const float4 x_ = relax4_h<RIGHT>(c_, u_, d_, P2[OLD][m + 1], P2[NEW][m + 1], B3_[m], e2_[m + 1]); \
			if (strip_live) *reinterpret_cast<float4*>(ob_ + opaque32(roff[m + 3])) = x_;   /* rows y0 .. y0+3 of a live strip are never clamped */ \
		} \
	} \
} while (0)

template <int WPG>
__global__ __launch_bounds__(64 * WPG, 4 / WPG) void k_jacobi_strip3h(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	__shared__ float4 lds_all[WPG * LDS_ROWS_PER_WAVE * 64];
	__shared__ float xbuf[2 * WPG * 16];
	__shared__ int xflag[2 * WPG];                                    // last z step whose first-sweep [0..WPG) / second-sweep [WPG..) edge cells each wave has published                                // [step parity][wave][8 first-sweep + 6 second-sweep edge cells]
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int lx = lane;
	float4* lds = lds_all + wave * (LDS_ROWS_PER_WAVE * 64) + lane;
	const int tile = xcd_index3(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int hs = grp * WPG + wave;                                    // half-strip: the two halves of a row strip are waves 2k, 2k + 1
	const bool right_half = hs & 1;
	const int y0 = (hs >> 1) * R3;
	const int xb = right_half ? 256 : 0, edge_lane = right_half ? 0 : 63;
	const bool strip_live = y0 < g.Y;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 3, g.zlo), q_last = ze - 1 + 3, q_load_last = min(q_last, g.zhi);
	const int b_load_last = min(q_last - 1, g.zhi);
	const size_t plane = g.plane();

	// byte offsets of the strip's rows inside a plane, 32 bits each: the plane base is wave-uniform (SGPR pair), so every access
	// is `global_load/store v, v_offset, s[base]` with no 64-bit address arithmetic per row
	uint32_t roff[R3 + 6];
#pragma unroll
	for (int i = 0; i < R3 + 6; ++i) roff[i] = ((uint32_t)min(max(y0 - 3 + i, 0), g.Y - 1) * (uint32_t)g.X + (uint32_t)xb + 4u * (uint32_t)lx) * 4u;
	// the cell across the cut, per row: a wave-uniform byte offset (scalar loads)
	uint32_t soff[R3 + 6];
#pragma unroll
	for (int i = 0; i < R3 + 6; ++i) soff[i] = ((uint32_t)min(max(y0 - 3 + i, 0), g.Y - 1) * (uint32_t)g.X + (right_half ? 255u : 256u)) * 4u;

	// LDS slots (float4 offsets into the wave's slice): two input planes, three b planes
	int s_ctr = 0, s_old = LDS_P0_ROWS * 64;
	int s_b2 = 2 * LDS_P0_ROWS * 64, s_b3 = s_b2 + LDS_B_ROWS * 64, s_bfree = s_b3 + LDS_B_ROWS * 64;

	float4 P1[3][R3 + 4], P2[3][R3 + 2], NP[R3 + 6], NB[R3 + 4];
	float E0c[R3 + 6], E0n[R3 + 6];                                     // input cells across the cut: centre plane, plane in flight
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R3 + 4; ++i) P1[k][i] = zero;
#pragma unroll
		for (int i = 0; i < R3 + 2; ++i) P2[k][i] = zero;
	}
#pragma unroll
	for (int i = 0; i < LDS_ROWS_PER_WAVE; ++i) lds[i * 64] = zero;
	if (threadIdx.x < 2 * WPG * 16) xbuf[threadIdx.x] = 0.0f;
	if (threadIdx.x < 2 * WPG) xflag[threadIdx.x] = INT_MIN;
	__syncthreads();
	{
		const char* pb = reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(qs, q_load_last)) * plane);
#pragma unroll
		for (int i = 0; i < R3 + 6; ++i) NP[i] = *reinterpret_cast<const float4*>(pb + roff[i]);
#pragma unroll
		for (int i = 0; i < R3 + 6; ++i) { E0n[i] = *reinterpret_cast<const float*>(pb + soff[i]); E0c[i] = 0.0f; }
		// b[qs - 1] for the first step's sweep 1 (clamped into the present planes: its level-1 plane is never used when qs - 1 < zlo)
		const char* bbase = reinterpret_cast<const char*>(b + (size_t)g.lz(min(max(qs - 1, g.zlo), g.zhi)) * plane);
#pragma unroll
		for (int i = 0; i < R3 + 4; ++i) NB[i] = *reinterpret_cast<const float4*>(bbase + roff[i + 1]);
	}
	// the pipeline's fill as six peeled steps without the sweeps that feed nothing stored (see k_jacobi_strip3c); the hand-over
	// counters start where the first active hand-over of each level expects them
	const bool fill = qs == zb - 3;
	if (lane == 0) { xflag[wave] = fill ? zb - 2 : qs - 1; xflag[WPG + wave] = fill ? zb : qs - 1; }   // both waves of a pair share the chunk
	__syncthreads();
	int q = qs;
#define FX_STRIP3H_RUN(RIGHT) do { \
		if (fill) { \
			FX_STRIP3H_STEP(0, RIGHT, false, false, false); ++q; \
			FX_STRIP3H_STEP(1, RIGHT, false, false, false); ++q; \
			FX_STRIP3H_STEP(2, RIGHT, true, false, false); ++q; \
			FX_STRIP3H_STEP(0, RIGHT, true, false, false); ++q; \
			FX_STRIP3H_STEP(1, RIGHT, true, true, false); ++q; \
			FX_STRIP3H_STEP(2, RIGHT, true, true, false); ++q; \
		} \
		for (;;) { \
			FX_STRIP3H_STEP(0, RIGHT, true, true, true); \
			if (++q > q_last) break; \
			FX_STRIP3H_STEP(1, RIGHT, true, true, true); \
			if (++q > q_last) break; \
			FX_STRIP3H_STEP(2, RIGHT, true, true, true); \
			if (++q > q_last) break; \
		} } while (0)
	if (right_half) FX_STRIP3H_RUN(true); else FX_STRIP3H_RUN(false);   // wave-uniform: each half runs its own expansion, free of per-update selects
}
#undef FX_STRIP3H_RUN
#undef FX_STRIP3H_STEP

// ---------------------------------------------------------------------------------------------------------------------------
// X = 256, cooperative pairs.  k_jacobi_strip3 recomputes a strip's y-halo on BOTH sides (18 row updates per z step for 12 useful
// ones: level 1 on 8 rows, level 2 on 6, output 4) so that no wave ever talks to another.  Here the two strips that share a
// workgroup's rows 8k .. 8k+7 are a PAIR: each recomputes only its OUTER side and takes the one row per level it needs across the
// inner boundary from its partner -- the partner's own edge row, handed over through a 1-KiB-per-row LDS mailbox one z step after
// it was computed (the level-1 row of plane q-2 and the level-2 row of plane q-3 are exactly what step q-1 produced), ordered by a
// per-wave step counter in the LDS like k_jacobi_strip3h's: a wave only waits while its partner is more than a step behind.  Per z
// step: 6 + 5 + 4 = 15 row updates instead of 18, 8 + 6 row loads instead of 10 + 8, and 24 fewer float4 of window registers
// (fewer AGPR round trips).  UP = the upper strip of the pair (outer side above, partner below); the lower strip is the mirror image
// -- two expansions of the step, because the stencil's U and D enter the sum in a fixed order and cannot trade places.
// Measured (256^3, rocprofv3 SQ counters): 7.96 k instead of 9.86 k VALU instructions per wave and launch (-19 %), bit-identical --
// and 43.6-44.0 us per launch against 44.2-44.4, because the wave now issues only 52 % of its cycles (62 % before): with less
// arithmetic per z step the step is as long as the prefetch's round trip (it is issued behind sweep 1 and needed at the next
// step's start, two thirds of a step later), which one wave per SIMD cannot hide.  A first version that waited for the partner
// at the top of the step was 4 % SLOWER (46.1 us); the order below (wait, read, then publish) lets a wave run a full step ahead.
// Row indices (top to bottom): input i <-> y = yb + i (8 rows, yb = y0 - 3 | y0 - 1), level 1 j <-> yb + 1 + j (6 rows),
// level 2 k <-> y0 - 1 + k | y0 + k (5 rows), output m <-> y0 + m.
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef FX_STRIP3C_ROWS
#define FX_STRIP3C_ROWS 4
#endif
constexpr int RC = FX_STRIP3C_ROWS;          // output rows per strip of the cooperative kernel.  2 rows (FLUIDX_BUILD_STRIP3C_ROWS=2: 128 strips x 8
                                             // chunks of 32 planes, 226 instead of 249 MB of traffic per launch) measured 47.6-48.0 us per launch against 43.5 for 4
constexpr int C_P0_ROWS = RC + 4;           // input rows per plane
constexpr int C_B_ROWS = RC + 2;            // b rows per plane (= level-1 rows)
constexpr int C_ROWS_PER_WAVE = 2 * C_P0_ROWS + 3 * C_B_ROWS;     // 34 rows of 1 KiB

#define FX_STRIP3C_STEP(PH, UP, S1, S2, S3) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	/* ---- sweep 1: level-1 plane q-1 ------------------------------------------------------------------------------- */ \
	if (q == 0) {                                   /* input plane -1 := plane 0, once (clamped front neighbour) */ \
		_Pragma("unroll") for (int i = 0; i < C_P0_ROWS; ++i) FX_LDS(s_ctr, i) = NP[i]; \
	} \
	if (!(S1)) {                                    /* (a fill step: see the kernel) */ \
	} else if (q - 1 == g.Zg) {                     /* level-1 plane Zg := plane Zg-1 */ \
		_Pragma("unroll") for (int j = 0; j < C_B_ROWS; ++j) P1[NEW][j] = P1[CTR][j]; \
	} else { \
		float4 C_[C_P0_ROWS], F_[C_B_ROWS]; \
		_Pragma("unroll") for (int i = 0; i < C_P0_ROWS; ++i) C_[i] = FX_LDS(s_ctr, i); \
		_Pragma("unroll") for (int j = 0; j < C_B_ROWS; ++j) F_[j] = FX_LDS(s_old, j + 1); \
		_Pragma("unroll") for (int j = 0; j < C_B_ROWS; ++j) \
			P1[NEW][j] = relax4(C_[j + 1], C_[j], C_[j + 2], F_[j], NP[j + 1], NB[j], false, false); \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int j = 0; j < C_B_ROWS; ++j) P1[CTR][j] = P1[NEW][j]; \
		} \
	} \
	float4 B2_[RC + 1], B3_[RC]; \
	_Pragma("unroll") for (int k = 0; k < RC + 1; ++k) B2_[k] = FX_LDS(s_b2, (UP) ? k + 1 : k);        /* b[q-2], rows of level 2 */ \
	_Pragma("unroll") for (int m = 0; m < RC; ++m) B3_[m] = FX_LDS(s_b3, (UP) ? m + 2 : m);            /* b[q-3], output rows */ \
	_Pragma("unroll") for (int i = 0; i < C_P0_ROWS; ++i) FX_LDS(s_old, i) = NP[i]; \
	_Pragma("unroll") for (int i = 0; i < C_B_ROWS; ++i) FX_LDS(s_bfree, i) = NB[i]; \
	{ const int t_ = s_old; s_old = s_ctr; s_ctr = t_; } \
	{ const int t_ = s_bfree; s_bfree = s_b3; s_b3 = s_b2; s_b2 = t_; } \
	/* the plane bases walk along with q (pp: input plane q+1, pbq: b plane q, po: output plane q-3): two scalar adds each instead of \
	   the eleven of a 64-bit `lz(q) * plane` */ \
	if (q + 1 <= q_load_last) { \
		_Pragma("unroll") for (int i = 0; i < C_P0_ROWS; ++i) NP[i] = *reinterpret_cast<const float4*>(pp + opaque32(roff[i])); \
	} \
	if (q <= b_load_last) { \
		_Pragma("unroll") for (int i = 0; i < C_B_ROWS; ++i) NB[i] = *reinterpret_cast<const float4*>(pbq + opaque32(roff[i + 1])); \
	} \
	pp += plane_bytes; pbq += plane_bytes; \
	/* hand-over 1 (level-1 edge rows), BEHIND the prefetch issue: a wait here must not delay the loads.  Order: wait until the \
	   partner has published its step q-1, READ its row, only then publish \
	   mine and my counter -- a wave that sees my counter at q knows I have already read what it wrote two steps ago into the slot \
	   it is about to reuse, so two slots (step parity) suffice; and a wave may run a whole step ahead of its partner */ \
	float4 H1_ = zero; \
	if (S1) { \
		H1_ = lds_wait_read(xf_partner, q - 1, xb0 + 16u * (uint32_t)(((((q - 1) & 1) * 4 + (wave ^ 1)) * 2 + 0) * 64 + lane));   /* partner's level 1, plane q-2 */ \
		xbuf[(((q & 1) * 4 + wave) * 2 + 0) * 64 + lane] = (UP) ? P1[NEW][C_B_ROWS - 1] : P1[NEW][0];   /* mine, plane q-1 */ \
		if (lane == 0) lds_post(xf_mine, q);                           /* LDS operations of a wave execute in order */ \
	} \
	/* ---- sweep 2: level-2 plane q-2 ------------------------------------------------------------------------------- */ \
	if (!(S2)) { \
	} else if (q - 2 == g.Zg) { \
		_Pragma("unroll") for (int k = 0; k < RC + 1; ++k) P2[NEW][k] = P2[CTR][k]; \
	} else { \
		_Pragma("unroll") for (int k = 0; k < RC + 1; ++k) { \
			const int jc_ = (UP) ? k + 1 : k;                            /* level-1 index of this row */ \
			const float4 c_ = P1[CTR][jc_]; \
			float4 u_ = jc_ >= 1 ? P1[CTR][jc_ >= 1 ? jc_ - 1 : 0] : H1_; \
			float4 d_ = jc_ + 1 < C_B_ROWS ? P1[CTR][jc_ + 1 < C_B_ROWS ? jc_ + 1 : 0] : H1_; \
			if ((UP) && k == 1 && y0 == 0) u_ = c_;                       /* rows outside the domain hold no data */ \
			if (!(UP) && k == RC - 1 && y0 + RC >= g.Y) d_ = c_; \
			P2[NEW][k] = relax4(c_, u_, d_, P1[OLD][jc_], P1[NEW][jc_], B2_[k], false, false); \
		} \
		if (q - 2 == 0) { \
			_Pragma("unroll") for (int k = 0; k < RC + 1; ++k) P2[CTR][k] = P2[NEW][k]; \
		} \
	} \
	/* hand-over 2 (level-2 edge rows), same order */ \
	float4 H2_ = zero; \
	if (S2) { \
		H2_ = lds_wait_read(xf_partner + 16, q - 1, xb0 + 16u * (uint32_t)(((((q - 1) & 1) * 4 + (wave ^ 1)) * 2 + 1) * 64 + lane));   /* partner's level 2, plane q-3 */ \
		xbuf[(((q & 1) * 4 + wave) * 2 + 1) * 64 + lane] = (UP) ? P2[NEW][RC] : P2[NEW][0];          /* mine, plane q-2 */ \
		if (lane == 0) lds_post(xf_mine + 16, q); \
	} \
	/* ---- sweep 3: output plane q-3 ------------------------------------------------------------------------------- */ \
	if ((S3) && q - 3 >= zb && q - 3 < ze) { \
		char* ob_ = po; \
		_Pragma("unroll") for (int m = 0; m < RC; ++m) { \
			const int kc_ = (UP) ? m + 1 : m;                            /* level-2 index of this row */ \
			const float4 c_ = P2[CTR][kc_]; \
			float4 u_ = kc_ >= 1 ? P2[CTR][kc_ >= 1 ? kc_ - 1 : 0] : H2_; \
			float4 d_ = kc_ + 1 < RC + 1 ? P2[CTR][kc_ + 1 < RC + 1 ? kc_ + 1 : 0] : H2_; \
			if ((UP) && m == 0 && y0 == 0) u_ = c_; \
			if (!(UP) && m == RC - 1 && y0 + RC >= g.Y) d_ = c_; \
			const float4 x_ = relax4(c_, u_, d_, P2[OLD][kc_], P2[NEW][kc_], B3_[m], false, false); \
			if (strip_live) *reinterpret_cast<float4*>(ob_ + opaque32(roff[(UP) ? m + 3 : m + 1])) = x_; \
		} \
	} \
	po += plane_bytes; \
} while (0)

__global__ __launch_bounds__(256, 1) void k_jacobi_strip3c(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	__shared__ float4 lds_all[4 * C_ROWS_PER_WAVE * 64];
	__shared__ float4 xbuf[2 * 4 * 2 * 64];                           // [step parity][wave][level-1 row, level-2 row][lane]
	__shared__ int xflag[8];                                          // last z step whose level-1 [0..3] / level-2 [4..7] edge row each wave has published
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	float4* lds = lds_all + wave * (C_ROWS_PER_WAVE * 64) + lane;
	const int tile = xcd_index3(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int y0 = (grp * 4 + wave) * RC;                             // (Y % 8 == 0: both strips of every pair exist)
	const bool up = (wave & 1) == 0;
	const bool strip_live = y0 < g.Y;                                 // Y % 16 == 8: the last workgroup's second pair lies outside (computes, never stores)
	const int yb = up ? y0 - 3 : y0 - 1;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 3, g.zlo), q_last = ze - 1 + 3, q_load_last = min(q_last, g.zhi);
	const int b_load_last = min(q_last - 1, g.zhi);
	const size_t plane = g.plane();

	uint32_t roff[C_P0_ROWS];
#pragma unroll
	for (int i = 0; i < C_P0_ROWS; ++i) roff[i] = ((uint32_t)min(max(yb + i, 0), g.Y - 1) * (uint32_t)g.X + 4u * (uint32_t)lane) * 4u;

	int s_ctr = 0, s_old = C_P0_ROWS * 64;
	int s_b2 = 2 * C_P0_ROWS * 64, s_b3 = s_b2 + C_B_ROWS * 64, s_bfree = s_b3 + C_B_ROWS * 64;

	float4 P1[3][C_B_ROWS], P2[3][RC + 1], NP[C_P0_ROWS], NB[C_B_ROWS];
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < C_B_ROWS; ++i) P1[k][i] = zero;
#pragma unroll
		for (int i = 0; i < RC + 1; ++i) P2[k][i] = zero;
	}
#pragma unroll
	for (int i = 0; i < C_ROWS_PER_WAVE; ++i) lds[i * 64] = zero;
	for (int i = (int)threadIdx.x; i < 2 * 4 * 2 * 64; i += 256) xbuf[i] = zero;
	{
		const char* pb = reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(qs, q_load_last)) * plane);
#pragma unroll
		for (int i = 0; i < C_P0_ROWS; ++i) NP[i] = *reinterpret_cast<const float4*>(pb + roff[i]);
		const char* bbase = reinterpret_cast<const char*>(b + (size_t)g.lz(min(max(qs - 1, g.zlo), g.zhi)) * plane);
#pragma unroll
		for (int i = 0; i < C_B_ROWS; ++i) NB[i] = *reinterpret_cast<const float4*>(bbase + roff[i + 1]);
	}
	// The pipeline's fill (a chunk that starts three planes before its first output plane): level-1 planes below zb-2 and level-2
	// planes below zb-1 feed nothing that is stored, so the first two steps only park and prefetch, the next two add sweep 1, the two
	// after that sweep 2 -- six peeled steps (two whole triples of phases) in front of the loop, 32 of a chunk's 330 row updates
	// less; and with the peeled copies the allocator keeps the loop itself free of AGPR moves (105-146 per three steps before).
	// 43.0 -> 41.7 us per launch.  (As branches inside the step: +150 instructions per three steps in the hot loop, a net loss.  With
	// the two load-only steps folded into a prologue that fetches all three planes at once: 41.95 us -- the launch is bound by its
	// bytes and its issue slots, not by the round trips of one workgroup.)  The hand-over counters start where the first active
	// hand-over of each level expects them.
	const bool fill = qs == zb - 3;
	if (threadIdx.x < 8) xflag[threadIdx.x] = !fill ? qs - 1 : threadIdx.x < 4 ? zb - 2 : zb;     // the waves of a workgroup share the chunk
	__syncthreads();
	const uint32_t xf0 = (uint32_t)(size_t)(__attribute__((address_space(3))) int*)xflag;      // LDS byte address of the counters
	const uint32_t xf_mine = xf0 + 4u * (uint32_t)wave, xf_partner = xf0 + 4u * (uint32_t)(wave ^ 1);
	const uint32_t xb0 = (uint32_t)(size_t)(__attribute__((address_space(3))) float4*)xbuf;   // LDS byte address of the mailbox
	int q = qs;
	const size_t plane_bytes = plane * 4;
	const char* pp = reinterpret_cast<const char*>(p_in) + ((ptrdiff_t)g.lz(qs) + 1) * (ptrdiff_t)plane_bytes;
	const char* pbq = reinterpret_cast<const char*>(b) + (ptrdiff_t)g.lz(qs) * (ptrdiff_t)plane_bytes;
	char* po = reinterpret_cast<char*>(p_out) + ((ptrdiff_t)g.lz(qs) - 3) * (ptrdiff_t)plane_bytes;     // (only dereferenced for planes inside the chunk)
#define FX_STRIP3C_RUN(UP) do { \
		if (fill) { \
			FX_STRIP3C_STEP(0, UP, false, false, false); ++q; \
			FX_STRIP3C_STEP(1, UP, false, false, false); ++q; \
			FX_STRIP3C_STEP(2, UP, true, false, false); ++q; \
			FX_STRIP3C_STEP(0, UP, true, false, false); ++q; \
			FX_STRIP3C_STEP(1, UP, true, true, false); ++q; \
			FX_STRIP3C_STEP(2, UP, true, true, false); ++q; \
		} \
		for (;;) { \
			FX_STRIP3C_STEP(0, UP, true, true, true); \
			if (++q > q_last) break; \
			FX_STRIP3C_STEP(1, UP, true, true, true); \
			if (++q > q_last) break; \
			FX_STRIP3C_STEP(2, UP, true, true, true); \
			if (++q > q_last) break; \
		} } while (0)
	if (up) FX_STRIP3C_RUN(true); else FX_STRIP3C_RUN(false);
}
#undef FX_STRIP3C_RUN
#undef FX_STRIP3C_STEP

// (k_jacobi_strip3z -- two z streams per workgroup that meet in the middle: 11 % fewer bytes, 6 % fewer instructions, and 45.9 us per
// launch against k_jacobi_strip3c's 43.0 -- was measured in round 2 (profiles/archive/r03c_strip3z.txt) and removed in round 3.)

}  // namespace

// read-and-clear of the hand-over fault word on the current device (fx_synchronize, behind the device)
hipError_t strip3_fault_take(unsigned* out)
{
	unsigned v = 0;
	hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_strip3_fault), sizeof v);
	if (e == hipSuccess && v) { const unsigned zero = 0; e = hipMemcpyToSymbol(HIP_SYMBOL(g_strip3_fault), &zero, sizeof zero); }
	*out = v;
	return e;
}

bool jacobi_strip3_supported(const Geom& g)
{
	const bool no_h = [] { const char* e = FX_KNOB("STRIP3_NO512"); return e && e[0] == '1'; }();
	return g.Zg > 1 && (g.X == 256 || (g.X == 512 && !no_h)) && (g.Y & 3) == 0 && g.Y >= 8;
}

hipError_t launch_jacobi_strip3(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (!jacobi_strip3_supported(g)) return hipErrorNotSupported;
	const char* ce = FX_KNOB("STRIP3_ZCHUNK");
	const char* re = FX_KNOB("STRIP_REMAP");
	const int forced_chunk = ce && *ce ? atoi(ce) : 0;
	const int remap = re && *re ? atoi(re) : 1;
	const bool halves = g.X == 512;                                     // two half-row waves per strip (k_jacobi_strip3h)
	const int coop = [] { const char* e = FX_KNOB("STRIP3_COOP"); return e && *e ? atoi(e) : 1; }();
	const bool use_coop = !halves && coop && (g.Y % (2 * RC)) == 0;
	const int nstrips = use_coop ? g.Y / RC : (g.Y / R3) * (halves ? 2 : 1);
	const int pair_wg = [] { const char* e = FX_KNOB("STRIP3H_PAIRS"); return e ? atoi(e) : 0; }();
	const int wpg = halves && pair_wg ? 2 : 4;                          // X = 512: a workgroup = one pair of half-row waves (its barrier syncs only them)
	const int ngroups = (nstrips + wpg - 1) / wpg;                      // waves (strips) per workgroup
	const int nzp = z_end - z_begin;
	int nchunks = (1024 / wpg + ngroups - 1) / ngroups;                 // 1024 waves: one per SIMD
	int zchunk = forced_chunk > 0 ? forced_chunk : (nzp + nchunks - 1) / nchunks;
	if (zchunk < 8) zchunk = 8;
	if (zchunk > nzp) zchunk = nzp;
	nchunks = (nzp + zchunk - 1) / zchunk;
	if (halves && wpg == 2) hipLaunchKernelGGL(k_jacobi_strip3h<2>, dim3(ngroups * nchunks), dim3(128), 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else if (halves) hipLaunchKernelGGL(k_jacobi_strip3h<4>, dim3(ngroups * nchunks), dim3(256), 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else {
		// cooperative pairs (k_jacobi_strip3c) where every workgroup holds whole pairs; FLUIDX_STRIP3_COOP=0: every strip on its own
		if (use_coop)
			hipLaunchKernelGGL(k_jacobi_strip3c, dim3(ngroups * nchunks), dim3(256), 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
		else
			hipLaunchKernelGGL(k_jacobi_strip3, dim3(ngroups * nchunks), dim3(256), 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	}
	return hipGetLastError();
}

}  // namespace fx
