// fx_jacobi_block.hip -- two lock-step Jacobi sweeps per launch for X = 128 (BASELINE configs[1], the reference's default grid,
// FluidX12/FluidX12.cpp:44): one wave64 = one 128-cell row, a lane = two cells, and a wave owns a 4 x 4 (y, z) block of rows
// OUTRIGHT -- it loads the 8 x 8 rows of p and 6 x 6 rows of b the block's dependency cone covers in one burst (100 independent
// 512-byte row loads in flight), relaxes 6 x 6 rows to the first level and 4 x 4 to the second in registers, stores 16 rows.
// No streaming along z, no LDS, no barrier.
//
// Why not the strip kernels (fx_jacobi_strip.hip): they stream along z and need ~16 planes per wave to amortise the pipeline's
// fill; 128^3 then has 256 waves for 1024 SIMDs and a launch lasts 14 dependent z steps (7.6 us per sweep against 5.6 for one
// sweep per launch).  At 2.1 M cells the three fields (24 MiB) sit in L2 / Infinity Cache, a single-sweep launch is a 1.7-us
// kernel boundary plus one round trip, and forty of them are the whole 0.226 ms of the step's Jacobi phase.  A block per wave
// gives 1024 waves x one round trip x two sweeps: half the launches, 1.6 x the arithmetic (52 row updates for 32 useful).
//
// Measured (MI355X, 128^3, 40 sweeps): Jacobi phase 0.224 -> 0.131 ms (twenty 6.6-us launches instead of forty 5.6-us ones), step
// 0.2975 -> 0.204 ms = 7.05 -> 10.3 G voxel-updates/s.  Other block shapes (FLUIDX_BLOCK_SHAPE = rows * 10 + planes): 4 x 2 7.1 us,
// 2 x 4 7.3, 2 x 2 8.1, 4 x 3 8.3 per launch -- 4 x 4 (224 VGPRs, one wave per SIMD, 1024 waves at 128^3) it is.
//
// Restates CSPoisson.hlsli:8-26 (/root/reference/FluidX12/Content/Shaders/) like k_jacobi_v4; per-cell arithmetic and
// association order are unchanged, so the result is bit-identical to two single sweeps (tests/test_gpu_sim.py).
#include "fx_internal.h"
#include "fx_pk.h"
#include <cstdlib>

namespace fx {

namespace {

// ((((((L - b) + R) + U) + D) + F) + B) * (1/6) on a float2 column; the row IS the wave, so the lanes without a DPP source (0 for
// wave_shr:1, 63 for wave_shl:1) are the clamped wall cells and keep the `old` operand = the cell itself
__device__ __forceinline__ float2 relax2(float2 c, float2 U, float2 D, float2 F, float2 Bk, float2 bb)
{
	// (L, c.x) and (c.y, R) as register pairs: (c.x, c.x) / (c.y, c.y) by one v_pk_mov_b32 each (fx_pk.h), the DPP shift lands in one half
	const fx_f2 cc = { c.x, c.y };
	fx_f2 lx = pk_mov(cc, cc, 0), rx = pk_mov(cc, cc, 2);
	lx.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lx.x), __builtin_bit_cast(int, c.y), 0x138, 0xf, 0xf, false));
	rx.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, rx.y), __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
	const fx_f2 b2 = { bb.x, bb.y }, U2 = { U.x, U.y }, D2 = { D.x, D.y }, F2 = { F.x, F.y }, B2 = { Bk.x, Bk.y };
	fx_f2 x = (((((lx - b2) + rx) + U2) + D2) + F2) + B2;
	x *= __uint_as_float(0x3e2aaaabu);
	return make_float2(x.x, x.y);
}


// uniform base + 32-bit byte offset: the `global_load v, v_offset, s[base]` form, no 64-bit address arithmetic per row
__device__ __forceinline__ float2 ld_row(const float* base, uint32_t byte_off)
{
	return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + byte_off);
}

// BY x BZ = rows x planes a wave produces
template <int BY, int BZ>
__global__ __launch_bounds__(256, 1) void k_jacobi_block2(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, int z_begin, int z_end, int nby, int nbz, int remap)
{
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int nblk = nby * nbz;
	int t = (int)blockIdx.x;
	if (remap) {                                            // XCD k walks the k-th contiguous eighth of the workgroup sequence
		const int n = (nblk + 3) >> 2, q = n >> 3, r = n & 7, xcd = t & 7, j = t >> 3;
		t = xcd * q + min(xcd, r) + j;
	}
	const int blk = t * 4 + wave;
	if (blk >= nblk) return;                                // uniform per wave; no barrier in this kernel
	const int by = blk % nby, bz = blk / nby;
	const int y0 = by * BY, z0 = z_begin + bz * BZ;
	const uint32_t plane = (uint32_t)g.plane();
	const int zmin = max(g.zlo, 0), zmax = min(g.zhi, g.Zg - 1);

	// ---- the block's dependency cone, clamped loads (the clamp IS the stencil's clamp-to-edge at the input level) -------------
	float2 P0[BZ + 4][BY + 4], Bv[BZ + 2][BY + 2];
	uint32_t roff[BY + 4];
#pragma unroll
	for (int i = 0; i < BY + 4; ++i) roff[i] = ((uint32_t)min(max(y0 - 2 + i, 0), g.Y - 1) * (uint32_t)g.X + 2u * lane) * 4u;   // bytes
#pragma unroll
	for (int k = 0; k < BZ + 4; ++k) {
		const uint32_t zo = (uint32_t)g.lz(min(max(z0 - 2 + k, zmin), zmax)) * plane * 4u;
#pragma unroll
		for (int i = 0; i < BY + 4; ++i) P0[k][i] = ld_row(p_in, zo + roff[i]);
	}
#pragma unroll
	for (int k = 0; k < BZ + 2; ++k) {
		const uint32_t zo = (uint32_t)g.lz(min(max(z0 - 1 + k, zmin), zmax)) * plane * 4u;
#pragma unroll
		for (int j = 0; j < BY + 2; ++j) Bv[k][j] = ld_row(b, zo + roff[j + 1]);
	}

	// ---- first sweep: planes z0-1 .. z0+BZ (k1), rows y0-1 .. y0+BY (j) -----------------------------------------------------------
	float2 P1[BZ + 2][BY + 2];
#pragma unroll
	for (int k1 = 0; k1 < BZ + 2; ++k1)
#pragma unroll
		for (int j = 0; j < BY + 2; ++j)
			P1[k1][j] = relax2(P0[k1 + 1][j + 1], P0[k1 + 1][j], P0[k1 + 1][j + 2], P0[k1][j + 1], P0[k1 + 2][j + 1], Bv[k1][j]);

	// ---- second sweep: planes z0 .. z0+BZ-1, rows y0 .. y0+BY-1.  Rows / planes outside the domain hold no first-level data:
	// there the neighbour is the cell itself.  Blocks that touch no face of the domain (wave-uniform) run without the selects.
	const bool edge = y0 == 0 || y0 + BY >= g.Y || z0 == 0 || z0 + BZ >= g.Zg;
	if (!edge) {
#pragma unroll
		for (int k2 = 0; k2 < BZ; ++k2) {
			const int z = z0 + k2;
			if (z >= z_end) break;
			const uint32_t zo = (uint32_t)g.lz(z) * plane;
#pragma unroll
			for (int r = 0; r < BY; ++r) {
				const float2 x = relax2(P1[k2 + 1][r + 1], P1[k2 + 1][r], P1[k2 + 1][r + 2], P1[k2][r + 1], P1[k2 + 2][r + 1], Bv[k2 + 1][r + 1]);
				*reinterpret_cast<float2*>(reinterpret_cast<char*>(p_out) + (zo + (uint32_t)(y0 + r) * (uint32_t)g.X + 2u * lane) * 4u) = x;
			}
		}
		return;
	}
#pragma unroll
	for (int k2 = 0; k2 < BZ; ++k2) {
		const int z = z0 + k2;
		if (z >= z_end) break;
		const bool zfirst = z == 0, zlast = z == g.Zg - 1;
		const uint32_t zo = (uint32_t)g.lz(z) * plane;
#pragma unroll
		for (int r = 0; r < BY; ++r) {
			const int y = y0 + r;
			const float2 c = P1[k2 + 1][r + 1];
			const float2 U = y == 0 ? c : P1[k2 + 1][r], D = y == g.Y - 1 ? c : P1[k2 + 1][r + 2];
			const float2 F = zfirst ? c : P1[k2][r + 1], Bk = zlast ? c : P1[k2 + 2][r + 1];
			const float2 x = relax2(c, U, D, F, Bk, Bv[k2 + 1][r + 1]);
			*reinterpret_cast<float2*>(reinterpret_cast<char*>(p_out) + (zo + (uint32_t)y * (uint32_t)g.X + 2u * lane) * 4u) = x;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same block-per-wave scheme for any row that fits one wave: X <= 256 with X = CPL x (active lanes), CPL = 1 .. 4 cells per
// lane.  Written for the reference's own GI preset, 150^3 (Bin/FluidGI.bat:1: X = 150 = 3 x 50 lanes): it ran forty launches of the
// scalar k_jacobi_generic, 19.3 us each = 0.76 of its 0.89-ms step.  Rows are not 16-byte aligned there (600 bytes), so the loads are
// 4-byte-aligned vector loads (global_load_dwordx3: fine in the global address space); lanes beyond the row work on the last
// lane's cells and store nothing; the row's last lane takes its right neighbour from itself (the DPP shift would hand it a copy
// from a lane outside the row).  Partial blocks at the y / z ends: clamped loads, guarded stores.  Arithmetic and association
// order per cell as above: bit-identical to two single sweeps.
// ---------------------------------------------------------------------------------------------------------------------------
template <int CPL> struct RowVec { typedef float __attribute__((ext_vector_type(CPL))) aligned_t; };
template <> struct RowVec<1> { typedef float aligned_t; };

template <int CPL> struct Cells { float v[CPL]; };

template <int CPL>
__device__ __forceinline__ Cells<CPL> ld_cells(const float* base, uint32_t byte_off)
{
	Cells<CPL> c;
	if constexpr (CPL == 1) {
		c.v[0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
	} else {
		typedef typename RowVec<CPL>::aligned_t __attribute__((aligned(4))) vec_t;
		const vec_t t = *reinterpret_cast<const vec_t*>(reinterpret_cast<const char*>(base) + byte_off);
#pragma unroll
		for (int i = 0; i < CPL; ++i) c.v[i] = t[i];
	}
	return c;
}

template <int CPL>
__device__ __forceinline__ void st_cells(float* base, uint32_t byte_off, const Cells<CPL>& c)
{
	if constexpr (CPL == 1) {
		*reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = c.v[0];
	} else {
		typedef typename RowVec<CPL>::aligned_t __attribute__((aligned(4))) vec_t;
		vec_t t;
#pragma unroll
		for (int i = 0; i < CPL; ++i) t[i] = c.v[i];
		*reinterpret_cast<vec_t*>(reinterpret_cast<char*>(base) + byte_off) = t;
	}
}

template <int CPL>
__device__ __forceinline__ Cells<CPL> relax_cells(const Cells<CPL>& c, const Cells<CPL>& U, const Cells<CPL>& D, const Cells<CPL>& F,
	const Cells<CPL>& Bk, const Cells<CPL>& bb, bool last_lane)
{
	const float first = c.v[0], last = c.v[CPL - 1];
	const float L = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, first), __builtin_bit_cast(int, last), 0x138, 0xf, 0xf, false));
	float R = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, last), __builtin_bit_cast(int, first), 0x130, 0xf, 0xf, false));
	if (last_lane) R = last;                                  // the wall cell: its right neighbour is itself
	const float inv = __uint_as_float(0x3e2aaaabu);
	Cells<CPL> x;
#pragma unroll
	for (int i = 0; i < CPL; ++i) {
		const float l = i == 0 ? L : c.v[i > 0 ? i - 1 : 0], r = i == CPL - 1 ? R : c.v[i < CPL - 1 ? i + 1 : 0];
		x.v[i] = ((((((l - bb.v[i]) + r) + U.v[i]) + D.v[i]) + F.v[i]) + Bk.v[i]) * inv;
	}
	return x;
}

template <int CPL, int BY, int BZ>
__global__ __launch_bounds__(256, 1) void k_jacobi_blockg(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, int z_begin, int z_end, int nby, int nbz, int remap)
{
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int nblk = nby * nbz;
	int t = (int)blockIdx.x;
	if (remap) {
		const int n = (nblk + 3) >> 2, q = n >> 3, r = n & 7, xcd = t & 7, j = t >> 3;
		t = xcd * q + min(xcd, r) + j;
	}
	const int blk = t * 4 + wave;
	if (blk >= nblk) return;
	const int by = blk % nby, bz = blk / nby;
	const int y0 = by * BY, z0 = z_begin + bz * BZ;
	const int nl = g.X / CPL;                               // active lanes
	const bool live = lane < nl, last_lane = lane == nl - 1;
	const uint32_t xoff = (uint32_t)min(lane, nl - 1) * (uint32_t)(CPL * 4);
	const uint32_t plane = (uint32_t)g.plane();
	const int zmin = max(g.zlo, 0), zmax = min(g.zhi, g.Zg - 1);

	Cells<CPL> P0[BZ + 4][BY + 4], Bv[BZ + 2][BY + 2];
	uint32_t roff[BY + 4];
#pragma unroll
	for (int i = 0; i < BY + 4; ++i) roff[i] = (uint32_t)min(max(y0 - 2 + i, 0), g.Y - 1) * (uint32_t)g.X * 4u + xoff;
#pragma unroll
	for (int k = 0; k < BZ + 4; ++k) {
		const uint32_t zo = (uint32_t)g.lz(min(max(z0 - 2 + k, zmin), zmax)) * plane * 4u;
#pragma unroll
		for (int i = 0; i < BY + 4; ++i) P0[k][i] = ld_cells<CPL>(p_in, zo + roff[i]);
	}
#pragma unroll
	for (int k = 0; k < BZ + 2; ++k) {
		const uint32_t zo = (uint32_t)g.lz(min(max(z0 - 1 + k, zmin), zmax)) * plane * 4u;
#pragma unroll
		for (int j = 0; j < BY + 2; ++j) Bv[k][j] = ld_cells<CPL>(b, zo + roff[j + 1]);
	}

	Cells<CPL> P1[BZ + 2][BY + 2];
#pragma unroll
	for (int k1 = 0; k1 < BZ + 2; ++k1)
#pragma unroll
		for (int j = 0; j < BY + 2; ++j)
			P1[k1][j] = relax_cells<CPL>(P0[k1 + 1][j + 1], P0[k1 + 1][j], P0[k1 + 1][j + 2], P0[k1][j + 1], P0[k1 + 2][j + 1], Bv[k1][j], last_lane);

#pragma unroll
	for (int k2 = 0; k2 < BZ; ++k2) {
		const int z = z0 + k2;
		if (z >= z_end) break;
		const bool zfirst = z == 0, zlast = z == g.Zg - 1;
		const uint32_t zo = (uint32_t)g.lz(z) * plane * 4u;
#pragma unroll
		for (int r = 0; r < BY; ++r) {
			const int y = y0 + r;
			if (y >= g.Y) break;
			const Cells<CPL> c = P1[k2 + 1][r + 1];
			const Cells<CPL> U = y == 0 ? c : P1[k2 + 1][r], D = y == g.Y - 1 ? c : P1[k2 + 1][r + 2];
			const Cells<CPL> F = zfirst ? c : P1[k2][r + 1], Bk = zlast ? c : P1[k2 + 2][r + 1];
			const Cells<CPL> x = relax_cells<CPL>(c, U, D, F, Bk, Bv[k2 + 1][r + 1], last_lane);
			if (live) st_cells<CPL>(p_out, zo + (uint32_t)y * (uint32_t)g.X * 4u + xoff, x);
		}
	}
}


}  // namespace

bool jacobi_block2_supported(const Geom& g)
{
	const int on = FX_KNOB_INT("JACOBI_BLOCK", 1);
	return on && g.Zg > 1 && g.X == 128 && (g.Y & 3) == 0 && g.cells_local() < ((size_t)1 << 30);
}

// cells per lane of the general kernel: the smallest CPL <= 4 with X = CPL x (at most 64 lanes); 0 = none
static int blockg_cpl(const Geom& g)
{
	for (int c = 1; c <= 4; ++c)
		if (g.X % c == 0 && g.X / c <= 64) return c;
	return 0;
}

bool jacobi_blockg_supported(const Geom& g)
{
	const int on = FX_KNOB_INT("JACOBI_BLOCKG", 1);
	return on && g.Zg > 1 && g.X >= 8 && blockg_cpl(g) != 0 && g.cells_local() < ((size_t)1 << 30);
}

hipError_t launch_jacobi_blockg(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (!jacobi_blockg_supported(g)) return hipErrorNotSupported;
	const int remap = FX_KNOB_INT("BLOCK_REMAP", 1);
	const int cpl = blockg_cpl(g);
	// 3 rows x 2 planes per wave.  Measured in round 2 (Jacobi phase of a step, ms; single sweeps / 4 x 4 / 4 x 2 / 2 x 4 / 2 x 2 / 3 x 3 /
	// 3 x 2 / 2 x 3 rows x planes per wave): 150^3 0.767 / 0.319 / 0.277 / 0.285 / 0.250 / 0.299 / 0.237 / 0.236, 192^3 0.581 / 0.533 / 0.485 /
	// 0.512 / 0.440 / 0.478 / 0.388 / 0.387, 160^3 0.386 / 0.339 / 0.365 / 0.376 / 0.304 / 0.327 / 0.303 / 0.303, 100^3 0.167 / 0.136 / 0.139 /
	// 0.142 / 0.131 / 0.146 / 0.125 / 0.128: six rows per wave everywhere -- unlike X = 128, whose float2 rows leave the 4 x 4 block at 224
	// registers.  The other shapes are no longer compiled.
	const int nby = (g.Y + 2) / 3, nbz = (z_end - z_begin + 1) / 2;
#define FX_BLKG(C_) hipLaunchKernelGGL((k_jacobi_blockg<C_, 3, 2>), dim3((nby * nbz + 3) / 4), dim3(256), 0, s, g, p_in, b, p_out, z_begin, z_end, nby, nbz, remap)
	switch (cpl) {
	case 1: FX_BLKG(1); break;
	case 2: FX_BLKG(2); break;
	case 3: FX_BLKG(3); break;
	default: FX_BLKG(4); break;
	}
#undef FX_BLKG
	return hipGetLastError();
}

hipError_t launch_jacobi_block2(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (!jacobi_block2_supported(g)) return hipErrorNotSupported;
	const int remap = FX_KNOB_INT("BLOCK_REMAP", 1);
	// 4 rows x 4 planes per wave (224 VGPRs, one wave per SIMD, 1024 waves at 128^3); 4 x 2 7.1 us, 2 x 4 7.3, 2 x 2 8.1, 4 x 3 8.3 per launch
	// against 6.6 -- the other shapes are no longer compiled
	const int nby = g.Y / 4, nbz = (z_end - z_begin + 3) / 4;
	hipLaunchKernelGGL((k_jacobi_block2<4, 4>), dim3((nby * nbz + 3) / 4), dim3(256), 0, s, g, p_in, b, p_out, z_begin, z_end, nby, nbz, remap);
	return hipGetLastError();
}

}  // namespace fx
