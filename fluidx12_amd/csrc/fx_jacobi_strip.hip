// fx_jacobi_strip.hip -- two lock-step Jacobi sweeps per launch, register-resident ("strip" kernel).
//
// Restates CSPoisson.hlsli:8-26 (/root/reference/FluidX12/Content/Shaders/) like k_jacobi_v4, two sweeps
// at a time.  Design for CDNA4's 512-VGPR waves rather than for LDS:
//   * one wave64 = one strip of R consecutive full-x rows (thread = 4 x-cells x R rows); it streams along z and
//     keeps a 3-plane window of the input (R+4 rows: a 2-row halo on each side) and of the first sweep's result
//     (R+2 rows) in registers -- about 330 VGPRs, one wave per SIMD;
//   * the halo rows are recomputed, not exchanged: no LDS, no barrier, every wave is independent;
//   * x neighbours come from wave shuffles, y neighbours from the thread's own registers, z neighbours from the
//     register window; p and b are read once (halo rows hit L2) and p'' is written once per TWO sweeps, so the
//     HBM / Infinity-Cache traffic per sweep is roughly halved.
// Per-cell arithmetic and association order are those of k_jacobi_v4: results are bit-identical to two single
// sweeps (tests/test_gpu_sim.py).
#include "fx_internal.h"
#include "fx_pk.h"
#include <cstdlib>

namespace fx {

namespace {

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// x neighbours across lanes as DPP operand modifiers (wave_shr:1 / wave_shl:1 exist on the GFX9 family, which gfx950 is;
// semantics checked on the device by tools/micro/dpp_test.cpp) instead of __shfl_up/__shfl_down, which compile to
// ds_bpermute_b32: a trip through the LDS crossbar whose latency a one-wave-per-SIMD kernel cannot hide.  Lane 0 / lane 63
// receive 0 and are overridden by the callers' edge rules.
__device__ __forceinline__ float lane_up1(float v)   // lane i <- lane i - 1
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_dn1(float v)   // lane i <- lane i + 1
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
template <int N> __device__ __forceinline__ float row_dn(float v)   // lane i <- lane i + N inside its row of 16 lanes
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

// one Jacobi update of a float4 column: ((((((L - b) + R) + U) + D) + F) + B) * (1/6)
__device__ __forceinline__ float4 relax4(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb, bool x_first, bool x_last)
{
	float L = lane_up1(c.w), Rr = lane_dn1(c.x);
	if (x_first) L = c.x;
	if (x_last) Rr = c.w;
	const float4 Lv = make_float4(L, c.x, c.y, c.z), Rv = make_float4(c.y, c.z, c.w, Rr);
	float4 x = add4(add4(add4(add4(add4(sub4(Lv, bb), Rv), U), D), F), Bk);
	const float inv = __uint_as_float(0x3e2aaaabu);
	x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
	return x;
}

// X = 256: the row is the wave, so the lanes without a DPP source (0 for wave_shr:1, 63 for wave_shl:1) are exactly the clamped
// wall cells; they keep the DPP's `old` operand, which is set to the cell itself -- no select per update
__device__ __forceinline__ float4 relax4_row(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb)
{
	return relax4_pairs(c, U, D, F, Bk, bb, 0.0f, true, true);         // (fx_pk.h)
}

// same XCD-aware tile order as fx_sim.hip (see xcd_tile there)
__device__ __forceinline__ int xcd_index(int n, int remap)
{
	int t = (int)blockIdx.x;
	if (remap) {
		const int q = n >> 3, r = n & 7;
		const int xcd = t & 7, j = t >> 3;
		t = xcd * q + min(xcd, r) + j;
	}
	return t;
}

// T sweeps per launch, R output rows per strip.  Level l (0 = input, T = output) carries R + 2(T - l) rows; row i of
// level l is global row y0 - (T - l) + i, i.e. row i + 1 of level l - 1.
template <int T, int R>
__global__ __launch_bounds__(256, 1) void k_jacobi_strip(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	constexpr int NR0 = R + 2 * T;                   // input rows per strip
	constexpr int NRB = R + 2 * (T - 1);             // rows of b the first sweep needs
	const int LX = g.X >> 2;                         // lanes per row
	const int SPW = 64 / LX;                         // strips per wave (1 at X = 256)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int lx = lane % LX, sub = lane / LX;
	const int tile = xcd_index(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int strip = (grp * 4 + wave) * SPW + sub;
	const int y0 = strip * R;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - T, g.zlo), q_last = ze - 1 + T, q_load_last = min(q_last, g.zhi);
	const bool x_first = lx == 0, x_last = lx == LX - 1;
	const size_t plane = g.plane();

	size_t roff[NR0];                                // clamped row offsets (clamp = the clamp-to-edge stencil of the input level)
#pragma unroll
	for (int i = 0; i < NR0; ++i) roff[i] = (size_t)min(max(y0 - T + i, 0), g.Y - 1) * g.X + 4 * lx;

	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float4 P[T][3][NR0];                             // level l uses the first R + 2(T - l) rows; the rest is never touched
	float4 Bq[T + 1][NRB];                           // Bq[k] = b of plane q - k, rows y0 - (T-1) ...
	float4 NP[NR0], NB[NRB];
#pragma unroll
	for (int l = 0; l < T; ++l)
#pragma unroll
		for (int k = 0; k < 3; ++k)
#pragma unroll
			for (int i = 0; i < NR0; ++i) P[l][k][i] = zero;
#pragma unroll
	for (int k = 0; k <= T; ++k)
#pragma unroll
		for (int i = 0; i < NRB; ++i) Bq[k][i] = zero;
#pragma unroll
	for (int i = 0; i < NR0; ++i) NP[i] = zero;
#pragma unroll
	for (int i = 0; i < NRB; ++i) NB[i] = zero;

	if (qs <= q_load_last) {
		const size_t zo = (size_t)g.lz(qs) * plane;
#pragma unroll
		for (int i = 0; i < NR0; ++i) NP[i] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]);
#pragma unroll
		for (int i = 0; i < NRB; ++i) NB[i] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]);
	}

	for (int q = qs; q <= q_last; ++q) {
		// ---- take input plane q, rotate the input window and the b delay line ----------------------------------------
#pragma unroll
		for (int i = 0; i < NR0; ++i) { P[0][0][i] = P[0][1][i]; P[0][1][i] = P[0][2][i]; P[0][2][i] = NP[i]; }
#pragma unroll
		for (int k = T; k >= 1; --k)
#pragma unroll
			for (int i = 0; i < NRB; ++i) Bq[k][i] = Bq[k - 1][i];
#pragma unroll
		for (int i = 0; i < NRB; ++i) Bq[0][i] = NB[i];
		if (q + 1 <= q_load_last) {                 // prefetch plane q+1 (one whole z-step of latency cover)
			const size_t zo = (size_t)g.lz(q + 1) * plane;
#pragma unroll
			for (int i = 0; i < NR0; ++i) NP[i] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]);
#pragma unroll
			for (int i = 0; i < NRB; ++i) NB[i] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]);
		}

		// ---- sweep s produces level s, plane q - s, from level s - 1 (planes q-s-1, q-s, q-s+1) ---------------------------
#pragma unroll
		for (int sw = 1; sw <= T; ++sw) {
			constexpr int dummy = 0; (void)dummy;
			const int m = q - sw;
			const bool m_first = m == 0, m_last = m == g.Zg - 1;
			const bool store_plane = sw == T && m >= zb && m < ze;
			const size_t zo2 = store_plane ? (size_t)g.lz(m) * plane : 0;
			const int nr = R + 2 * (T - sw);
			if (sw < T) {
#pragma unroll
				for (int i = 0; i < NR0; ++i)
					if (i < nr) { P[sw < T ? sw : 0][0][i] = P[sw < T ? sw : 0][1][i]; P[sw < T ? sw : 0][1][i] = P[sw < T ? sw : 0][2][i]; }
			}
#pragma unroll
			for (int i = 0; i < NR0; ++i) {
				if (i < nr) {
					const int y = y0 - (T - sw) + i;
					const float4 c = P[sw - 1][1][i + 1];
					const float4 U = y <= 0 ? c : P[sw - 1][1][i];            // rows outside the domain never hold data
					const float4 D = y >= g.Y - 1 ? c : P[sw - 1][1][i + 2];
					const float4 x = relax4(c, U, D, m_first ? c : P[sw - 1][0][i + 1], m_last ? c : P[sw - 1][2][i + 1],
						Bq[sw][i + sw - 1], x_first, x_last);
					if (sw < T) P[sw < T ? sw : 0][2][i] = x;
					else if (store_plane && y < g.Y) *reinterpret_cast<float4*>(p_out + zo2 + (size_t)y * g.X + 4 * lx) = x;
				}
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Two sweeps, tuned: the generic kernel above spends most of its VALU issue on rotating the register windows
// (v_mov / v_accvgpr) and on per-use boundary selects (rocprofv3: SQ_ACTIVE_INST_VALU = 64 % of wave cycles, only a
// quarter of it arithmetic).  Here the three window slots rotate by NAME (the z loop is unrolled by three, slot
// indices are compile-time), and the z boundaries are handled by duplicating a plane into the neighbouring slot once,
// under a wave-uniform branch, instead of a select at every use.  Needs Y % R == 0.
// ---------------------------------------------------------------------------------------------------------------
// one z step with the window slots named at compile time (expanded three times in the kernel: plain local arrays with
// constant indices stay in registers; a struct passed by reference was demoted to scratch by the compiler)
#define FX_STRIP2_STEP(PH) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	_Pragma("unroll") for (int i = 0; i < R + 4; ++i) P0[NEW][i] = NP[i]; \
	_Pragma("unroll") for (int i = 0; i < R + 2; ++i) Bq[NEW][i] = NB[i]; \
	if (q == 0) {                                   /* plane -1 := plane 0 (clamped front neighbour), once */ \
		_Pragma("unroll") for (int i = 0; i < R + 4; ++i) P0[CTR][i] = NP[i]; \
	} \
	if (q + 1 <= q_load_last) {                     /* prefetch plane q+1; past the last plane NP keeps plane zhi (clamped back) */ \
		const size_t zo = (size_t)g.lz(q + 1) * plane; \
		_Pragma("unroll") for (int i = 0; i < R + 4; ++i) NP[i] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]); \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) NB[i] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]); \
	} \
	/* sweep 1: plane q-1, rows y0-1 .. y0+R; the clamped row loads already encode the y boundary */ \
	if (q - 1 == g.Zg) {                            /* plane Zg := plane Zg-1 of the first sweep's result */ \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) P1[NEW][i] = P1[CTR][i]; \
	} else { \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) \
			P1[NEW][i] = FULLROW ? relax4_row(P0[CTR][i + 1], P0[CTR][i], P0[CTR][i + 2], P0[OLD][i + 1], P0[NEW][i + 1], Bq[CTR][i]) \
				: relax4(P0[CTR][i + 1], P0[CTR][i], P0[CTR][i + 2], P0[OLD][i + 1], P0[NEW][i + 1], Bq[CTR][i], x_first, x_last); \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int i = 0; i < R + 2; ++i) P1[CTR][i] = P1[NEW][i]; \
		} \
	} \
	/* sweep 2: plane q-2, rows y0 .. y0+R-1 */ \
	if (q - 2 >= zb && q - 2 < ze) { \
		const size_t zo2 = (size_t)g.lz(q - 2) * plane; \
		_Pragma("unroll") for (int j = 0; j < R; ++j) { \
			const float4 c = P1[CTR][j + 1]; \
			float4 U = P1[CTR][j], D = P1[CTR][j + 2]; \
			if (j == 0 && y0 == 0) U = c;                                /* rows outside the domain hold no data */ \
			if (j == R - 1 && y0 + R >= g.Y) D = c; \
			const float4 x = FULLROW ? relax4_row(c, U, D, P1[OLD][j + 1], P1[NEW][j + 1], Bq[OLD][j + 1]) \
				: relax4(c, U, D, P1[OLD][j + 1], P1[NEW][j + 1], Bq[OLD][j + 1], x_first, x_last); \
			if (strip_live) *reinterpret_cast<float4*>(p_out + zo2 + (size_t)(y0 + j) * g.X + 4 * lx) = x; \
		} \
	} \
} while (0)

template <int R, int MINW, bool FULLROW>
__global__ __launch_bounds__(256, MINW) void k_jacobi_strip2u(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	const int LX = g.X >> 2, SPW = 64 / LX;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int lx = lane % LX, sub = lane / LX;
	const int tile = xcd_index(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int strip = (grp * 4 + wave) * SPW + sub;
	const int y0 = strip * R;
	const bool strip_live = y0 < g.Y;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 2, g.zlo), q_last = ze - 1 + 2, q_load_last = min(q_last, g.zhi);
	const bool x_first = lx == 0, x_last = lx == LX - 1;
	const size_t plane = g.plane();

	size_t roff[R + 4];
#pragma unroll
	for (int i = 0; i < R + 4; ++i) roff[i] = (size_t)min(max(y0 - 2 + i, 0), g.Y - 1) * g.X + 4 * lx;

	float4 P0[3][R + 4], P1[3][R + 2], Bq[3][R + 2], NP[R + 4], NB[R + 2];
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R + 4; ++i) P0[k][i] = zero;
#pragma unroll
		for (int i = 0; i < R + 2; ++i) { P1[k][i] = zero; Bq[k][i] = zero; }
	}
	{
		const size_t zo = (size_t)g.lz(min(qs, q_load_last)) * plane;
#pragma unroll
		for (int i = 0; i < R + 4; ++i) NP[i] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]);
#pragma unroll
		for (int i = 0; i < R + 2; ++i) NB[i] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]);
	}
	int q = qs;
	for (;;) {
		FX_STRIP2_STEP(0);
		if (++q > q_last) break;
		FX_STRIP2_STEP(1);
		if (++q > q_last) break;
		FX_STRIP2_STEP(2);
		if (++q > q_last) break;
	}
}
#undef FX_STRIP2_STEP

// ---------------------------------------------------------------------------------------------------------------
// The same two-sweep kernel for X = 512: a row is two coalesced 1-KiB halves, lane l owns float4 l of the left half
// (A) and float4 l of the right half (B).  The x neighbours across the middle of the row travel by v_readlane
// (lane 63's A.w <-> lane 0's B.x), everything else is as above.  R = 2 rows per strip keeps the windows at ~420
// registers; 512^2 planes still give 256 strips, i.e. long z chunks.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 relax4_lr(float4 c, float L, float Rr, float4 U, float4 D, float4 F, float4 Bk, float4 bb)
{
	const float4 Lv = make_float4(L, c.x, c.y, c.z), Rv = make_float4(c.y, c.z, c.w, Rr);
	float4 x = add4(add4(add4(add4(add4(sub4(Lv, bb), Rv), U), D), F), Bk);
	const float inv = __uint_as_float(0x3e2aaaabu);
	x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
	return x;
}

// relax both halves of one row: inputs are [2]-arrays (A, B)
#define FX_RELAX_W(out, c, U, D, F, Bk, bb) do { \
	const float4 cA = (c)[0], cB = (c)[1]; \
	float LA = lane_up1(cA.w), RA = lane_dn1(cA.x), LB = lane_up1(cB.w), RB = lane_dn1(cB.x); \
	const float midA = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cA.w), 63)), midB = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cB.x), 0)); \
	if (lane == 0) { LA = cA.x; LB = midA; } \
	if (lane == 63) { RA = midB; RB = cB.w; } \
	(out)[0] = relax4_lr(cA, LA, RA, (U)[0], (D)[0], (F)[0], (Bk)[0], (bb)[0]); \
	(out)[1] = relax4_lr(cB, LB, RB, (U)[1], (D)[1], (F)[1], (Bk)[1], (bb)[1]); \
} while (0)

#define FX_STRIPW_STEP(PH) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	_Pragma("unroll") for (int i = 0; i < R + 4; ++i) { P0[NEW][i][0] = NP[i][0]; P0[NEW][i][1] = NP[i][1]; } \
	_Pragma("unroll") for (int i = 0; i < R + 2; ++i) { Bq[NEW][i][0] = NB[i][0]; Bq[NEW][i][1] = NB[i][1]; } \
	if (q == 0) { \
		_Pragma("unroll") for (int i = 0; i < R + 4; ++i) { P0[CTR][i][0] = NP[i][0]; P0[CTR][i][1] = NP[i][1]; } \
	} \
	if (q + 1 <= q_load_last) { \
		const size_t zo = (size_t)g.lz(q + 1) * plane; \
		_Pragma("unroll") for (int i = 0; i < R + 4; ++i) { \
			NP[i][0] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]); \
			NP[i][1] = *reinterpret_cast<const float4*>(p_in + zo + roff[i] + 256); } \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) { \
			NB[i][0] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]); \
			NB[i][1] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1] + 256); } \
	} \
	if (q - 1 == g.Zg) { \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) { P1[NEW][i][0] = P1[CTR][i][0]; P1[NEW][i][1] = P1[CTR][i][1]; } \
	} else { \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) \
			FX_RELAX_W(P1[NEW][i], P0[CTR][i + 1], P0[CTR][i], P0[CTR][i + 2], P0[OLD][i + 1], P0[NEW][i + 1], Bq[CTR][i]); \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int i = 0; i < R + 2; ++i) { P1[CTR][i][0] = P1[NEW][i][0]; P1[CTR][i][1] = P1[NEW][i][1]; } \
		} \
	} \
	if (q - 2 >= zb && q - 2 < ze) { \
		const size_t zo2 = (size_t)g.lz(q - 2) * plane; \
		_Pragma("unroll") for (int j = 0; j < R; ++j) { \
			float4 U[2] = { P1[CTR][j][0], P1[CTR][j][1] }, D[2] = { P1[CTR][j + 2][0], P1[CTR][j + 2][1] }, x[2]; \
			if (j == 0 && y0 == 0) { U[0] = P1[CTR][j + 1][0]; U[1] = P1[CTR][j + 1][1]; } \
			if (j == R - 1 && y0 + R >= g.Y) { D[0] = P1[CTR][j + 1][0]; D[1] = P1[CTR][j + 1][1]; } \
			FX_RELAX_W(x, P1[CTR][j + 1], U, D, P1[OLD][j + 1], P1[NEW][j + 1], Bq[OLD][j + 1]); \
			if (strip_live) { \
				float* dst = p_out + zo2 + (size_t)(y0 + j) * g.X + 4 * lane; \
				*reinterpret_cast<float4*>(dst) = x[0]; \
				*reinterpret_cast<float4*>(dst + 256) = x[1]; \
			} \
		} \
	} \
} while (0)

template <int R>
__global__ __launch_bounds__(256, 1) void k_jacobi_strip2w(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int tile = xcd_index(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int y0 = (grp * 4 + wave) * R;
	const bool strip_live = y0 < g.Y;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 2, g.zlo), q_last = ze - 1 + 2, q_load_last = min(q_last, g.zhi);
	const size_t plane = g.plane();

	size_t roff[R + 4];
#pragma unroll
	for (int i = 0; i < R + 4; ++i) roff[i] = (size_t)min(max(y0 - 2 + i, 0), g.Y - 1) * g.X + 4 * lane;

	float4 P0[3][R + 4][2], P1[3][R + 2][2], Bq[3][R + 2][2], NP[R + 4][2], NB[R + 2][2];
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R + 4; ++i) { P0[k][i][0] = zero; P0[k][i][1] = zero; }
#pragma unroll
		for (int i = 0; i < R + 2; ++i) { P1[k][i][0] = zero; P1[k][i][1] = zero; Bq[k][i][0] = zero; Bq[k][i][1] = zero; }
	}
	{
		const size_t zo = (size_t)g.lz(min(qs, q_load_last)) * plane;
#pragma unroll
		for (int i = 0; i < R + 4; ++i) {
			NP[i][0] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]);
			NP[i][1] = *reinterpret_cast<const float4*>(p_in + zo + roff[i] + 256);
		}
#pragma unroll
		for (int i = 0; i < R + 2; ++i) {
			NB[i][0] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]);
			NB[i][1] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1] + 256);
		}
	}
	int q = qs;
	for (;;) {
		FX_STRIPW_STEP(0);
		if (++q > q_last) break;
		FX_STRIPW_STEP(1);
		if (++q > q_last) break;
		FX_STRIPW_STEP(2);
		if (++q > q_last) break;
	}
}
#undef FX_STRIPW_STEP
#undef FX_RELAX_W

// ---------------------------------------------------------------------------------------------------------------
// X = 512, second design: the row is cut in the middle and every wave runs the X = 256 recipe (R = 4 rows, one float4
// per lane) on ONE half.  What the 256-wide kernel gets from the wall clamp at its outer lane, the half-row wave gets
// from a "seam": the two columns just across the cut (s, the neighbour of the wave's edge cell, and t, the one
// behind it).  The seam lives transposed in three scalar-sized window registers -- lane i < 8 holds row i of column s,
// lane 8 + i row i of column t (one masked load instruction per plane for p, one for b) -- and the first sweep's value
// of column s (needed by the second sweep's edge cell) is computed for the four strip rows in lanes 0..3 with the same
// association order as relax4.  Edge-lane neighbours come out of those registers by v_readlane.  Cost over the 256-wide
// kernel: two loads, a dozen lane reads and one scalar-width relax per z step; unlike k_jacobi_strip2w it keeps R = 4
// (2.5 instead of 3 row updates per output row, 14 instead of 20 float4 loads per 4 x 256 outputs).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 relax4_edge(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb,
	bool x_first, bool x_last, float edge_l, float edge_r)
{
	float L = lane_up1(c.w), Rr = lane_dn1(c.x);
	if (x_first) L = edge_l;
	if (x_last) Rr = edge_r;
	const float4 Lv = make_float4(L, c.x, c.y, c.z), Rv = make_float4(c.y, c.z, c.w, Rr);
	float4 x = add4(add4(add4(add4(add4(sub4(Lv, bb), Rv), U), D), F), Bk);
	const float inv = __uint_as_float(0x3e2aaaabu);
	x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
	return x;
}

// lane value -> wave-uniform scalar
#define FX_RL(v, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (v)), (l)))

#define FX_STRIPH_STEP(PH) do { \
	constexpr int NEW = (PH) % 3, CTR = ((PH) + 2) % 3, OLD = ((PH) + 1) % 3; \
	_Pragma("unroll") for (int i = 0; i < R + 4; ++i) P0[NEW][i] = NP[i]; \
	_Pragma("unroll") for (int i = 0; i < R + 2; ++i) Bq[NEW][i] = NB[i]; \
	SP[NEW] = SNP; SB[NEW] = SNB; \
	if (q == 0) { \
		_Pragma("unroll") for (int i = 0; i < R + 4; ++i) P0[CTR][i] = NP[i]; \
		SP[CTR] = SNP; \
	} \
	if (q + 1 <= q_load_last) { \
		const size_t zo = (size_t)g.lz(q + 1) * plane; \
		_Pragma("unroll") for (int i = 0; i < R + 4; ++i) NP[i] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]); \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) NB[i] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]); \
		if (lane < 16) SNP = p_in[zo + seam_off]; \
		if (lane < 8) SNB = b[zo + seam_off]; \
	} \
	/* sweep 1, plane q-1 */ \
	if (q - 1 == g.Zg) { \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) P1[NEW][i] = P1[CTR][i]; \
		S1[NEW] = S1[CTR]; \
	} else { \
		/* column s of the first sweep, strip rows j = 0..R-1 in lanes 0..R-1 (row j = seam row j + 2) */ \
		{ \
			const float c = row_dn<2>(SP[CTR]), U = row_dn<1>(SP[CTR]), D = row_dn<3>(SP[CTR]); \
			const float F = row_dn<2>(SP[OLD]), Bk = row_dn<2>(SP[NEW]), far = row_dn<10>(SP[CTR]); \
			const float bb = row_dn<2>(SB[CTR]); \
			float own = 0.0f;                       /* the wave's edge cell of the same row: x = 255 (.w of lane 63) or x = 256 (.x of lane 0) */ \
			_Pragma("unroll") for (int j = 0; j < R; ++j) { \
				const float e = right_half ? FX_RL(P0[CTR][j + 2].x, 0) : FX_RL(P0[CTR][j + 2].w, 63); \
				if (lane == j) own = e; \
			} \
			const float Lc = right_half ? far : own, Rc = right_half ? own : far; \
			S1[NEW] = ((((((Lc - bb) + Rc) + U) + D) + F) + Bk) * __uint_as_float(0x3e2aaaabu); \
		} \
		_Pragma("unroll") for (int i = 0; i < R + 2; ++i) { \
			const float e = FX_RL(SP[CTR], i + 1); \
			P1[NEW][i] = relax4_edge(P0[CTR][i + 1], P0[CTR][i], P0[CTR][i + 2], P0[OLD][i + 1], P0[NEW][i + 1], Bq[CTR][i], \
				x_first, x_last, right_half ? e : P0[CTR][i + 1].x, right_half ? P0[CTR][i + 1].w : e); \
		} \
		if (q - 1 == 0) { \
			_Pragma("unroll") for (int i = 0; i < R + 2; ++i) P1[CTR][i] = P1[NEW][i]; \
			S1[CTR] = S1[NEW]; \
		} \
	} \
	/* sweep 2, plane q-2 */ \
	if (q - 2 >= zb && q - 2 < ze) { \
		const size_t zo2 = (size_t)g.lz(q - 2) * plane; \
		_Pragma("unroll") for (int j = 0; j < R; ++j) { \
			const float4 c = P1[CTR][j + 1]; \
			float4 U = P1[CTR][j], D = P1[CTR][j + 2]; \
			if (j == 0 && y0 == 0) U = c; \
			if (j == R - 1 && y0 + R >= g.Y) D = c; \
			const float e = FX_RL(S1[CTR], j); \
			const float4 x = relax4_edge(c, U, D, P1[OLD][j + 1], P1[NEW][j + 1], Bq[OLD][j + 1], x_first, x_last, \
				right_half ? e : c.x, right_half ? c.w : e); \
			if (strip_live) *reinterpret_cast<float4*>(p_out + zo2 + (size_t)(y0 + j) * g.X + xb + 4 * lane) = x; \
		} \
	} \
} while (0)

template <int R>
__global__ __launch_bounds__(256, 1) void k_jacobi_strip2h(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: scalar branches on the half
	const int tile = xcd_index(ngroups * nchunks, remap);
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int hs = grp * 4 + wave;                   // half-strip index: the two halves of a row strip are neighbours in a workgroup
	const bool right_half = hs & 1;
	const int y0 = (hs >> 1) * R;
	const int xb = right_half ? 256 : 0;
	const bool strip_live = y0 < g.Y;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 2, g.zlo), q_last = ze - 1 + 2, q_load_last = min(q_last, g.zhi);
	const bool x_first = lane == 0, x_last = lane == 63;
	const size_t plane = g.plane();

	size_t roff[R + 4];
#pragma unroll
	for (int i = 0; i < R + 4; ++i) roff[i] = (size_t)min(max(y0 - 2 + i, 0), g.Y - 1) * g.X + xb + 4 * lane;
	// seam: lanes 0..7 = column s (first across the cut), lanes 8..15 = column t (second), row (lane & 7) of the R + 4 rows
	const size_t seam_off = (size_t)min(max(y0 - 2 + (lane & 7), 0), g.Y - 1) * g.X + (right_half ? (lane < 8 ? 255 : 254) : (lane < 8 ? 256 : 257));

	float4 P0[3][R + 4], P1[3][R + 2], Bq[3][R + 2], NP[R + 4], NB[R + 2];
	float SP[3] = { 0.0f, 0.0f, 0.0f }, SB[3] = { 0.0f, 0.0f, 0.0f }, S1[3] = { 0.0f, 0.0f, 0.0f }, SNP = 0.0f, SNB = 0.0f;
	const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R + 4; ++i) P0[k][i] = zero;
#pragma unroll
		for (int i = 0; i < R + 2; ++i) { P1[k][i] = zero; Bq[k][i] = zero; }
	}
	{
		const size_t zo = (size_t)g.lz(min(qs, q_load_last)) * plane;
#pragma unroll
		for (int i = 0; i < R + 4; ++i) NP[i] = *reinterpret_cast<const float4*>(p_in + zo + roff[i]);
#pragma unroll
		for (int i = 0; i < R + 2; ++i) NB[i] = *reinterpret_cast<const float4*>(b + zo + roff[i + 1]);
		if (lane < 16) SNP = p_in[zo + seam_off];
		if (lane < 8) SNB = b[zo + seam_off];
	}
	int q = qs;
	for (;;) {
		FX_STRIPH_STEP(0);
		if (++q > q_last) break;
		FX_STRIPH_STEP(1);
		if (++q > q_last) break;
		FX_STRIPH_STEP(2);
		if (++q > q_last) break;
	}
}
#undef FX_STRIPH_STEP
#undef FX_RL


}  // namespace

bool jacobi_strip_supported(const Geom& g)
{
	const int LX = g.X >> 2;
	if (g.Zg > 1 && g.X == 512 && (g.Y & 1) == 0) return true;        // wide kernel: two sweeps only (k_jacobi_strip2w)
	return g.Zg > 1 && (g.X & 3) == 0 && (LX == 16 || LX == 32 || LX == 64) && g.Y >= 8;
}

bool jacobi_strip_wide(const Geom& g) { return g.X == 512; }

hipError_t launch_jacobi_strip(const Geom& g, const float* p_in, const float* b, float* p_out, int sweeps, int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (!jacobi_strip_supported(g)) return hipErrorNotSupported;
	const int forced_chunk = FX_KNOB_INT("STRIP_ZCHUNK", 0);   // measurement knobs
	const int remap = FX_KNOB_INT("STRIP_REMAP", 1);
	const int Rsel = FX_KNOB_INT("STRIP_R", 0);
	const bool wide = jacobi_strip_wide(g);
	if (wide && sweeps != 2) return hipErrorNotSupported;
	// X = 512: half-row waves with a seam (k_jacobi_strip2h, R = 4) unless Y % 4 != 0 or FLUIDX_STRIP_WIDE=1 asks for the
	// two-float4-per-lane kernel (k_jacobi_strip2w, R = 2): the A/B switch of DESIGN.md section 6
	const int force_w = FX_KNOB_INT("STRIP_WIDE", 0);
	const bool halves = wide && (g.Y & 3) == 0 && !force_w;
	const int rows = halves ? 4 : wide ? 2 : (Rsel == 2 || Rsel == 4 ? Rsel : (sweeps == 3 ? 2 : 4));
	const int R = rows;
	const int LX = g.X >> 2, SPW = wide ? 1 : 64 / LX;
	const int nstrips = ((g.Y + R - 1) / R) * (halves ? 2 : 1);
	const int ngroups = (nstrips + 4 * SPW - 1) / (4 * SPW);            // 4 waves per workgroup
	const int nzp = z_end - z_begin;
	const int wg_target = FX_KNOB_INT("STRIP_WGS", 256);
	int nchunks = (wg_target + ngroups - 1) / ngroups;                  // one workgroup (= 1 wave per SIMD) per CU
	int zchunk = forced_chunk > 0 ? forced_chunk : (nzp + nchunks - 1) / nchunks;
	if (zchunk < 8) zchunk = 8;
	if (zchunk > nzp) zchunk = nzp;
	nchunks = (nzp + zchunk - 1) / zchunk;
	const dim3 grid(ngroups * nchunks), block(256);
#define FX_STRIP(T_, R_) hipLaunchKernelGGL((k_jacobi_strip<T_, R_>), grid, block, 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap)
	const int generic = FX_KNOB_INT("STRIP_GENERIC", 0);
	if (halves)
		hipLaunchKernelGGL(k_jacobi_strip2h<4>, grid, block, 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else if (wide)
		hipLaunchKernelGGL(k_jacobi_strip2w<2>, grid, block, 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else if (sweeps == 2 && R == 4 && (g.Y & 3) == 0 && !generic && g.X == 256)
		hipLaunchKernelGGL((k_jacobi_strip2u<4, 1, true>), grid, block, 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else if (sweeps == 2 && R == 4 && (g.Y & 3) == 0 && !generic)
		hipLaunchKernelGGL((k_jacobi_strip2u<4, 1, false>), grid, block, 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else if (sweeps == 2 && R == 2 && (g.Y & 1) == 0 && !generic)   // two waves per SIMD (<= 256 registers): FLUIDX_STRIP_R=2
		hipLaunchKernelGGL((k_jacobi_strip2u<2, 2, false>), grid, block, 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
	else if (sweeps == 2) { if (R == 2) FX_STRIP(2, 2); else FX_STRIP(2, 4); }
	else if (sweeps == 3) { if (R == 4) FX_STRIP(3, 4); else FX_STRIP(3, 2); }
	else return hipErrorNotSupported;
#undef FX_STRIP
	return hipGetLastError();
}

}  // namespace fx
