// fx_jacobi_stripm.hip -- THREE levels of the reference's own pressure loop (CSPoisson.hlsli:8-26: a sweep, and a cell leaves the loop for
// good once a sweep changes it by less than 1e-3) for EVERY cell, per launch: k_jacobi_strip3's streaming pipeline (fx_jacobi_strip3.hip: a
// wave = a strip of four full-x rows streaming along z, the level-1 and level-2 windows in registers, the older input planes and the
// b planes of the later levels in the wave's slice of the LDS) with the freeze nibbles of fx_jacobi_freeze.hip carried along.
//
// Why: on a developed plume 82-98 % of the tiles still relax after the dense first level, and the sparse solver's first tile launches
// are dense sweeps in all but name -- 43 us per level at 256^3, each tile staging a cone five times its core.  The strip pipeline reads
// a plane once per launch and is bound by its memory round trips, not by its arithmetic (a probe that doubled k_jacobi_strip3c's VALU work
// left the launch at 40.6 us): the freeze test, the select that keeps a frozen cell and the nibble bookkeeping ride along for free.
// fx_schedule.cpp runs it for the levels behind k_freeze_dense while most tiles are expected to relax, then hands over to
// k_freeze_tiles, whose first launch scans the tile marks THIS kernel's last launch left.
//
// Per cell the arithmetic is relax1's of fx_jacobi_freeze.hip, operation for operation: s = (((((L - b) + R) + U) + D) + F) + B,
// x = s * 1/6, frozen |= |fma(s, 1/6, -x0)| < 1e-3, and a frozen cell keeps x0.  Halo rows and planes are recomputed with their freeze
// decisions (deterministic: the owner computes the same).  Input: level L in p_in with its nibbles in m_in.  Output: level L + 3 to BOTH
// p_outA and p_outB and the nibbles to both mask buffers -- the tile launches alternate between two buffers and expect unlisted tiles to
// agree in them -- so the three pressure buffers and three mask buffers of a context rotate (in-place would race: a strip's halo rows
// are another strip's core).  Tile marks: a 32 x 8 x 8 tile with a cell that still relaxes after level L + 3 gets `tag`.  stat: the last
// level that left a cell relaxing (fx_jacobi_freeze.hip's word: atomicMax of stat_hi + level, once per wave).
// X = 256 only (one row = one wave of float4 lanes), single domain.
#include "fx_internal.h"
#include "fx_pk.h"

namespace fx {

namespace {

constexpr int RM = 4;                                    // output rows per strip
constexpr int M_P0_ROWS = RM + 6, M_B_ROWS = RM + 2;
constexpr int M_ROWS_PER_WAVE = 2 * M_P0_ROWS + 3 * M_B_ROWS;   // 38 rows of 1 KiB, as k_jacobi_strip3
constexpr float kBelow = 0.00100000005f;                 // CSPoisson.hlsli:24 as compiled (0x3a83126f)

__device__ __forceinline__ uint32_t opaque_u32(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

// one quad: relax4_pairs' arithmetic (fx_pk.h) with the sum kept for the freeze test; nib = frozen bits on entry, returned updated
__device__ __forceinline__ float4 relax4m(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb, uint32_t nib, uint32_t& nib_out)
{
	const fx_f2 c01 = { c.x, c.y }, c23 = { c.z, c.w };
	fx_f2 lx = pk_mov(c01, c01, 0);
	const fx_f2 mid = pk_mov(c01, c23, 1);
	fx_f2 rx = pk_mov(c23, c23, 2);
	lx.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lx.x), __builtin_bit_cast(int, c.w), 0x138, 0xf, 0xf, false));
	rx.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, rx.y), __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
	const fx_f2 b01 = { bb.x, bb.y }, b23 = { bb.z, bb.w }, U01 = { U.x, U.y }, U23 = { U.z, U.w }, D01 = { D.x, D.y }, D23 = { D.z, D.w };
	const fx_f2 F01 = { F.x, F.y }, F23 = { F.z, F.w }, B01 = { Bk.x, Bk.y }, B23 = { Bk.z, Bk.w };
	const fx_f2 s01 = (((((lx - b01) + mid) + U01) + D01) + F01) + B01;
	const fx_f2 s23 = (((((mid - b23) + rx) + U23) + D23) + F23) + B23;
	const float inv = __uint_as_float(0x3e2aaaabu);
	const bool f0 = fabsf(fmaf(s01.x, inv, -c.x)) < kBelow, f1 = fabsf(fmaf(s01.y, inv, -c.y)) < kBelow;
	const bool f2 = fabsf(fmaf(s23.x, inv, -c.z)) < kBelow, f3 = fabsf(fmaf(s23.y, inv, -c.w)) < kBelow;
	fx_f2 x01 = s01, x23 = s23;
	x01 *= inv; x23 *= inv;
	nib_out = nib | (f0 ? 1u : 0u) | (f1 ? 2u : 0u) | (f2 ? 4u : 0u) | (f3 ? 8u : 0u);
	return make_float4((nib & 1u) ? c.x : x01.x, (nib & 2u) ? c.y : x01.y, (nib & 4u) ? c.z : x23.x, (nib & 8u) ? c.w : x23.y);
}

__device__ __forceinline__ int xcd_index_m(int n)
{
	const int t = (int)blockIdx.x, q = n >> 3, r = n & 7, xcd = t & 7, j = t >> 3;
	return xcd * q + min(xcd, r) + j;
}

#define FXM_LDS(slot, r) lds[(slot) + (r) * 64]
#define FXM_NIB(m, r) (((m) >> (4 * (r))) & 15u)

// One z step; NEW / CTR / OLD name the three planes of the level-1 and level-2 windows (rotated by name, as in FX_STRIP3_STEP).
// M0c: nibbles of input plane q - 1, rows i = 1 .. 8 (row i at bits 4 (i - 1)); NMb: the mask bytes of input plane q, in flight.
#define FXM_STEP(PH) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	/* the input plane's nibbles have arrived with it */ \
	uint32_t M0n = 0u; \
	_Pragma("unroll") for (int i = 0; i < RM + 4; ++i) M0n |= ((uint32_t)NMb[i] & 15u) << (4 * i); \
	if (q >= zb && q < ze && strip_live) rel0 |= ~(M0n >> 8) & 0xFFFFu;          /* rows i = 3 .. 6 = the strip's own, relaxing on entry */ \
	/* ---- level 1 of plane q - 1, rows j <-> y0 - 2 + j; input rows i <-> y0 - 3 + i ---- */ \
	if (q == 0) { \
		_Pragma("unroll") for (int i = 0; i < RM + 6; ++i) FXM_LDS(s_ctr, i) = NP[i]; \
		M0c = M0n; \
	} \
	if (q - 1 == g.Zg) { \
		_Pragma("unroll") for (int j = 0; j < RM + 4; ++j) P1[NEW][j] = P1[CTR][j]; \
		M1[NEW] = M1[CTR]; \
	} else { \
		float4 C_[RM + 6], F_[RM + 4]; \
		_Pragma("unroll") for (int i = 0; i < RM + 6; ++i) C_[i] = FXM_LDS(s_ctr, i); \
		_Pragma("unroll") for (int j = 0; j < RM + 4; ++j) F_[j] = FXM_LDS(s_old, j + 1); \
		uint32_t m1_ = 0u; \
		_Pragma("unroll") for (int j = 0; j < RM + 4; ++j) { \
			uint32_t n_; \
			P1[NEW][j] = relax4m(C_[j + 1], C_[j], C_[j + 2], F_[j], NP[j + 1], NB[j], FXM_NIB(M0c, j), n_); \
			m1_ |= n_ << (4 * j); \
		} \
		M1[NEW] = m1_; \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int j = 0; j < RM + 4; ++j) P1[CTR][j] = P1[NEW][j]; \
			M1[CTR] = M1[NEW]; \
		} \
		if (q - 1 >= zb && q - 1 < ze && strip_live) rel1 |= ~(m1_ >> 8) & 0xFFFFu;   /* rows j = 2 .. 5 */ \
	} \
	float4 B2_[RM + 2], B3_[RM]; \
	_Pragma("unroll") for (int k = 0; k < RM + 2; ++k) B2_[k] = FXM_LDS(s_b2, k); \
	_Pragma("unroll") for (int m = 0; m < RM; ++m) B3_[m] = FXM_LDS(s_b3, m + 1); \
	_Pragma("unroll") for (int i = 0; i < RM + 6; ++i) FXM_LDS(s_old, i) = NP[i]; \
	_Pragma("unroll") for (int i = 0; i < RM + 2; ++i) FXM_LDS(s_bfree, i) = NB[i + 1]; \
	{ const int t_ = s_old; s_old = s_ctr; s_ctr = t_; } \
	{ const int t_ = s_bfree; s_bfree = s_b3; s_b3 = s_b2; s_b2 = t_; } \
	M0c = M0n;                                                                /* plane q is next step's centre */ \
	if (q + 1 <= q_load_last) { \
		const char* pb_ = reinterpret_cast<const char*>(p_in + (size_t)(q + 1) * plane); \
		_Pragma("unroll") for (int i = 0; i < RM + 6; ++i) NP[i] = *reinterpret_cast<const float4*>(pb_ + opaque_u32(roff[i])); \
		const uint8_t* mb_ = m_in + (size_t)(q + 1) * plane4; \
		_Pragma("unroll") for (int i = 0; i < RM + 4; ++i) NMb[i] = mb_[opaque_u32(moff[i + 1])]; \
	} \
	if (q <= b_load_last) { \
		const char* bb_ = reinterpret_cast<const char*>(b + (size_t)q * plane); \
		_Pragma("unroll") for (int i = 0; i < RM + 4; ++i) NB[i] = *reinterpret_cast<const float4*>(bb_ + opaque_u32(roff[i + 1])); \
	} \
	/* ---- level 2 of plane q - 2, rows k <-> y0 - 1 + k ---- */ \
	if (q - 2 == g.Zg) { \
		_Pragma("unroll") for (int k = 0; k < RM + 2; ++k) P2[NEW][k] = P2[CTR][k]; \
		M2[NEW] = M2[CTR]; \
	} else { \
		uint32_t m2_ = 0u; \
		_Pragma("unroll") for (int k = 0; k < RM + 2; ++k) { \
			const float4 c_ = P1[CTR][k + 1]; \
			float4 u_ = P1[CTR][k], d_ = P1[CTR][k + 2]; \
			if (k == 1 && y0 == 0) u_ = c_; \
			if (k == RM && y0 + RM >= g.Y) d_ = c_; \
			uint32_t n_; \
			P2[NEW][k] = relax4m(c_, u_, d_, P1[OLD][k + 1], P1[NEW][k + 1], B2_[k], FXM_NIB(M1[CTR], k + 1), n_); \
			m2_ |= n_ << (4 * k); \
		} \
		M2[NEW] = m2_; \
		if (q - 2 == 0) { \
			_Pragma("unroll") for (int k = 0; k < RM + 2; ++k) P2[CTR][k] = P2[NEW][k]; \
			M2[CTR] = M2[NEW]; \
		} \
		if (q - 2 >= zb && q - 2 < ze && strip_live) rel2 |= ~(m2_ >> 4) & 0xFFFFu;   /* rows k = 1 .. 4 */ \
	} \
	/* ---- level 3 of plane q - 3, rows m <-> y0 + m: stored to both buffers, with the nibbles and the tile mark ---- */ \
	if (q - 3 >= zb && q - 3 < ze) { \
		const size_t po_ = (size_t)(q - 3) * plane * 4u; \
		const size_t mo_ = (size_t)(q - 3) * plane4; \
		uint32_t all_ = 15u; \
		_Pragma("unroll") for (int m = 0; m < RM; ++m) { \
			const float4 c_ = P2[CTR][m + 1]; \
			float4 u_ = P2[CTR][m], d_ = P2[CTR][m + 2]; \
			if (m == 0 && y0 == 0) u_ = c_; \
			if (m == RM - 1 && y0 + RM >= g.Y) d_ = c_; \
			uint32_t n_; \
			const float4 x_ = relax4m(c_, u_, d_, P2[OLD][m + 1], P2[NEW][m + 1], B3_[m], FXM_NIB(M2[CTR], m + 1), n_); \
			all_ &= n_; \
			if (strip_live) { \
				*reinterpret_cast<float4*>(reinterpret_cast<char*>(p_outA) + po_ + opaque_u32(roff[m + 3])) = x_; \
				*reinterpret_cast<float4*>(reinterpret_cast<char*>(p_outB) + po_ + opaque_u32(roff[m + 3])) = x_; \
				m_outA[mo_ + moff[m + 3]] = (uint8_t)n_; \
				m_outB[mo_ + moff[m + 3]] = (uint8_t)n_; \
			} \
		} \
		if (strip_live && all_ != 15u) { rel3 = 1u; tile_act = true; } \
		/* the mark of a tile: one plain store per tile plane group, behind its last plane (no read of the mark: a load answered in \
		   the middle of the step would make the wave wait for its own prefetch) */ \
		if ((((q - 3) & 7) == 7 || q - 3 == ze - 1)) { \
			const unsigned long long bal_ = __ballot(tile_act); \
			if ((lane & 7) == 0 && ((bal_ >> lane) & 0xFFull) != 0ull) tile_mark[(((q - 3) >> 3) * nty + (y0 >> 3)) * ntx + (lane >> 3)] = tag; \
			tile_act = false; \
		} \
	} \
} while (0)

__global__ __launch_bounds__(256, 1) void k_freeze_strip3(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_outA, float* __restrict__ p_outB, const uint8_t* __restrict__ m_in, uint8_t* __restrict__ m_outA, uint8_t* __restrict__ m_outB,
	uint32_t* __restrict__ tile_mark, uint32_t tag, int ntx, int nty, uint32_t* __restrict__ stat, uint32_t stat_hi, int level_in,
	int zchunk, int ngroups, int nchunks)
{
	__shared__ float4 lds_all[4 * M_ROWS_PER_WAVE * 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	float4* lds = lds_all + wave * (M_ROWS_PER_WAVE * 64) + lane;
	const int tile = xcd_index_m(ngroups * nchunks);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int y0 = (grp * 4 + wave) * RM;
	const bool strip_live = y0 < g.Y;
	const int zb = chunk * zchunk, ze = min(zb + zchunk, g.Zg);
	const int qs = max(zb - 3, 0), q_last = ze - 1 + 3, q_load_last = min(q_last, g.Zg - 1);
	const int b_load_last = min(q_last - 1, g.Zg - 1);
	const size_t plane = g.plane(), plane4 = (size_t)64 * g.Y;

	uint32_t roff[RM + 6], moff[RM + 6];
#pragma unroll
	for (int i = 0; i < RM + 6; ++i) {
		const uint32_t y = (uint32_t)min(max(y0 - 3 + i, 0), g.Y - 1);
		roff[i] = (y * (uint32_t)g.X + 4u * (uint32_t)lane) * 4u;                     // bytes
		moff[i] = y * 64u + (uint32_t)lane;
	}
	int s_ctr = 0, s_old = M_P0_ROWS * 64;
	int s_b2 = 2 * M_P0_ROWS * 64, s_b3 = s_b2 + M_B_ROWS * 64, s_bfree = s_b3 + M_B_ROWS * 64;

	float4 P1[3][RM + 4], P2[3][RM + 2], NP[RM + 6], NB[RM + 4];
	uint32_t M1[3] = { 0u, 0u, 0u }, M2[3] = { 0u, 0u, 0u }, M0c = 0u;
	uint8_t NMb[RM + 4];
	uint32_t rel0 = 0u, rel1 = 0u, rel2 = 0u, rel3 = 0u;
	bool tile_act = false;
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < RM + 4; ++i) P1[k][i] = zero;
#pragma unroll
		for (int i = 0; i < RM + 2; ++i) P2[k][i] = zero;
	}
#pragma unroll
	for (int i = 0; i < M_ROWS_PER_WAVE; ++i) lds[i * 64] = zero;
	{
		const int q0 = min(qs, q_load_last);
		const char* pb = reinterpret_cast<const char*>(p_in + (size_t)q0 * plane);
#pragma unroll
		for (int i = 0; i < RM + 6; ++i) NP[i] = *reinterpret_cast<const float4*>(pb + roff[i]);
		const uint8_t* mb = m_in + (size_t)q0 * plane4;
#pragma unroll
		for (int i = 0; i < RM + 4; ++i) NMb[i] = mb[moff[i + 1]];
		const char* bbase = reinterpret_cast<const char*>(b + (size_t)min(max(qs - 1, 0), g.Zg - 1) * plane);
#pragma unroll
		for (int i = 0; i < RM + 4; ++i) NB[i] = *reinterpret_cast<const float4*>(bbase + roff[i + 1]);
	}
	int q = qs;
	for (;;) {
		FXM_STEP(0);
		if (++q > q_last) break;
		FXM_STEP(1);
		if (++q > q_last) break;
		FXM_STEP(2);
		if (++q > q_last) break;
	}
	// the last level that left one of this wave's own cells relaxing (level_in itself: a cell that came in relaxing)
	const int lvl = __any(rel3 != 0u) ? 3 : __any(rel2 != 0u) ? 2 : __any(rel1 != 0u) ? 1 : __any(rel0 != 0u) ? 0 : -1;
	if (lane == 0 && lvl >= 0) atomicMax(stat, stat_hi + (uint32_t)(level_in + lvl));
}
#undef FXM_STEP
#undef FXM_NIB
#undef FXM_LDS

// how many tiles the dense sweep marked (tile_mark == gen), into a host-visible word: fx_schedule.cpp reads it -- a step or two late, without
// waiting -- to decide whether the next solve's first levels go through the strip pipeline or straight to the tile launches
__global__ __launch_bounds__(1024) void k_count_marks(const uint32_t* __restrict__ tile_mark, uint32_t gen, int ntiles, uint32_t* __restrict__ out)
{
	__shared__ uint32_t part[16];
	uint32_t n = 0;
	for (int t = (int)threadIdx.x; t < ntiles; t += 1024) n += tile_mark[t] == gen ? 1u : 0u;
	n = (uint32_t)__popcll(__ballot(n & 1u)) + 2u * (uint32_t)__popcll(__ballot(n & 2u)) + 4u * (uint32_t)__popcll(__ballot(n & 4u)) + 8u * (uint32_t)__popcll(__ballot(n & 8u))
		+ 16u * (uint32_t)__popcll(__ballot(n & 16u)) + 32u * (uint32_t)__popcll(__ballot(n >> 5 & 1u)) + 64u * (uint32_t)__popcll(__ballot(n >> 6 & 1u)) + 128u * (uint32_t)__popcll(__ballot(n >> 7));
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t s = 0; for (int i = 0; i < 16; ++i) s += part[i]; *out = s; }
}

}  // namespace

hipError_t launch_count_marks(const uint32_t* tile_mark, uint32_t gen, int ntiles, uint32_t* out, hipStream_t s)
{
	hipLaunchKernelGGL(k_count_marks, dim3(1), dim3(1024), 0, s, tile_mark, gen, ntiles, out);
	return hipGetLastError();
}

namespace {
}  // namespace

bool jacobi_freeze_strip_supported(const Geom& g)
{
	return g.nz == g.Zg && g.H == 0 && g.X == 256 && (g.Y & 3) == 0 && g.Y >= 8 && g.Zg >= 8 && (uint64_t)g.X * g.Y * (uint64_t)g.Zg < (1u << 30);
}

// levels level_in + 1 .. level_in + 3 for every cell: p_in / m_in -> p_outA = p_outB, m_outA = m_outB; tiles that still relax get `tag`
hipError_t launch_freeze_strip3(const Geom& g, const float* p_in, const float* b, float* p_outA, float* p_outB, const uint8_t* m_in, uint8_t* m_outA, uint8_t* m_outB,
	uint32_t* tile_mark, uint32_t tag, uint32_t* stat, uint32_t stat_hi, int level_in, hipStream_t s)
{
	if (!jacobi_freeze_strip_supported(g)) return hipErrorNotSupported;
	const int nstrips = g.Y / RM, ngroups = (nstrips + 3) / 4;
	int nchunks = (256 + ngroups - 1) / ngroups;                                   // 1024 waves: one per SIMD
	int zchunk = (g.Zg + nchunks - 1) / nchunks;
	if (zchunk < 8) zchunk = 8;
	if (zchunk > g.Zg) zchunk = g.Zg;
	nchunks = (g.Zg + zchunk - 1) / zchunk;
	const int ntx = (g.X + 31) / 32, nty = (g.Y + 7) / 8;
	hipLaunchKernelGGL(k_freeze_strip3, dim3(ngroups * nchunks), dim3(256), 0, s, g, p_in, b, p_outA, p_outB, m_in, m_outA, m_outB, tile_mark, tag, ntx, nty,
		stat, stat_hi, level_in, zchunk, ngroups, nchunks);
	return hipGetLastError();
}

}  // namespace fx
