// fx_jacobi_freeze.hip -- the reference's OWN pressure solve as a sparse solver: CSPoisson.hlsli:8-26 under CSProject3D.hlsl:13
// (/root/reference/FluidX12/Content/Shaders/): at most ITER = 64 sweeps, and a cell leaves the loop for good as soon as one
// sweep changes it by less than 1e-3 (`if (abs(x - x0) < 0.001) break`, :24) -- FX_JACOBI_FAITHFUL, the configuration
// Fluid.hpp's FluidOptions default to.  Lock-step schedule: sweep k reads level k - 1 everywhere; a cell
// that broke out ("frozen") keeps the value it stored last, its neighbours keep reading that value.
//
// What the flow looks like (oracle, 128^3 / 64^3, steps 1..60): 72-99 % of the cells freeze in the FIRST sweep (the pressure is
// warm-started and only moves near the plume), the rest in a shrinking region that needs 22-64 sweeps; summed over a step the
// cells still relaxing amount to 1.0-2.0 sweeps of the grid, not 64.  So:
//
//   launch 1   k_freeze_dense   level 1 for every cell, streaming like k_jacobi_v4 (thread = 4 x-cells): p1 goes to TWO buffers
//                               (A and B; the input lives in a third), the freeze mask (one byte per 4-cell quad, low nibble) to
//                               two as well, and every 32 x 8 x 8-cell TILE that still has a relaxing cell is marked for launch 2.
//                               In whole steps it also computes the divergence from the advected velocity (and stores it for the
//                               tile launches) instead of reading it back from a launch of its own.
//   launch L   k_freeze_tiles<T> T more levels, A -> B -> A ..., only on the marked tiles: a workgroup stages the tile's cone
//                               (40 x (8 + 2T)^2 cells of p, b and mask) in the LDS, relaxes level by level there (halo cells are
//                               RECOMPUTED, their freeze decisions included -- the same deterministic arithmetic the owner tile
//                               does), stores its 32 x 8 x 8 core and marks itself for launch L + 1 while a core cell relaxes.
//                               A tile whose core froze completely during launch L holds its final values in that launch's output
//                               buffer only: launch L + 1 copies them across (core only, no LDS), after which both buffers
//                               agree and the tile is never touched again -- its neighbours read either buffer alike.
// Nothing is read back by the host: all 1 + ceil((N - 1) / T) launches are enqueued, and launches whose tile marks name nobody
// find nothing to do.  Results are bit-identical to N lock-step sweeps with the freeze mask (oracle: orc_jacobi mode 1); once
// every cell is frozen further sweeps change nothing, so the oracle's early exit at "active == 0" is not a different result.
// The number of sweeps the reference's loop would have executed (1 + the last level that left a cell relaxing, capped at N) is
// kept in a device word per step (`stat`), for fx_timing / bench.py's byte count.
//
// Scope: 3-D contexts (rows of any length >= 4: X % 4 != 0 -- 150^3, the reference's GI preset -- takes the cell-wise path of ldq / stq
// for a row's last, short quad).  A z-slab rank (round 4) runs the same kernels on a VIEW of its local arrays: the planes it holds (owned
// + exchanged halo) as a grid of their own -- jacobi_freeze_view -- whose ends are either the global boundary (the true clamp) or the
// outermost halo plane (a cell T planes from an owned tile is only ever read); the dense sweep covers the owned planes, the cones of the
// tile launches reach into the halo, and T planes of pressure + mask travel to the neighbours behind every launch (fx_schedule.cpp).
// 2-D grids keep k_jacobi_generic.
#include "fx_internal.h"
#include <algorithm>
#include <cstdlib>

namespace fx {

namespace {

// -DFX_FREEZE_PROF (tools/micro/freeze_prof.py): thread 0 of every workgroup adds the shader clocks it spent in each phase of a tile pass
#ifdef FX_FREEZE_PROF
__device__ unsigned long long fz_prof[16];
__device__ unsigned int fz_passes[2][80];      // tile passes / copy-only entries per first level of a launch
#define PROF(i) do { if (tid == 0) { const unsigned long long now = __builtin_readcyclecounter(); atomicAdd(&fz_prof[i], now - tprev); tprev = now; } } while (0)
#define PROF_INIT unsigned long long tprev = __builtin_readcyclecounter()
#define PROF_COUNT do { if (tid == 0) { atomicAdd(&fz_prof[15], 1ull); atomicAdd(&fz_passes[0][level_base], 1u); } } while (0)
#define PROF_COPY do { if (tid == 0) atomicAdd(&fz_passes[1][level_base], 1u); } while (0)
#else
#define PROF(i) do { } while (0)
#define PROF_INIT do { } while (0)
#define PROF_COUNT do { } while (0)
#define PROF_COPY do { } while (0)
#endif
typedef _Float16 fz_h16;
typedef _Float16 fz_h16x4 __attribute__((ext_vector_type(4)));
// a velocity quad from its storage (FUSE 1: fp32, 2: binary16 -- the reference's RGBA16F), cell offset `off` of component plane `comp`
template <int FUSE>
__device__ __forceinline__ float4 ldv4(const void* vel, uint32_t comp_cells, int comp, uint32_t off)
{
	const size_t cell = (size_t)comp * comp_cells + off;
	if (FUSE == 2) { const fz_h16x4 h = *reinterpret_cast<const fz_h16x4*>(static_cast<const fz_h16*>(vel) + cell); return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w); }
	return *reinterpret_cast<const float4*>(static_cast<const float*>(vel) + cell);
}
template <int FUSE>
__device__ __forceinline__ float ldv1(const void* vel, uint32_t off)
{
	return FUSE == 2 ? (float)static_cast<const fz_h16*>(vel)[off] : static_cast<const float*>(vel)[off];
}

constexpr int TCX = 32, TCY = 8, TCZ = 8;     // tile core (cells)
constexpr int TQ = 10;                        // quads (4 x-cells) per staged row: the core's 8 + one halo quad per side
constexpr float kFreezeBelow = 0.00100000005f;   // CSPoisson.hlsli:24 as compiled (0x3a83126f)

__device__ __forceinline__ float inv6() { return __uint_as_float(0x3e2aaaabu); }

// one cell: s = ((((((L - b) + R) + U) + D) + F) + B), x = s * 1/6, freezes when |fma(s, 1/6, -x0)| < 1e-3 (the DXBC's mad)
__device__ __forceinline__ float relax1(float L, float R, float U, float D, float F, float Bk, float bb, float x0, bool& fr)
{
	float s = L - bb;
	s = R + s; s = U + s; s = D + s; s = F + s; s = Bk + s;
	fr = fabsf(fmaf(s, inv6(), -x0)) < kFreezeBelow;
	return s * inv6();
}

// a quad; `m` = frozen nibble on entry (a frozen cell keeps x0), returns the new nibble
__device__ __forceinline__ uint32_t relax_quad(float4 c, float L, float R, float4 U, float4 D, float4 F, float4 Bk, float4 bb, uint32_t m, float4& o)
{
	bool f0, f1, f2, f3;
	const float x0 = relax1(L, c.y, U.x, D.x, F.x, Bk.x, bb.x, c.x, f0);
	const float x1 = relax1(c.x, c.z, U.y, D.y, F.y, Bk.y, bb.y, c.y, f1);
	const float x2 = relax1(c.y, c.w, U.z, D.z, F.z, Bk.z, bb.z, c.z, f2);
	const float x3 = relax1(c.z, R, U.w, D.w, F.w, Bk.w, bb.w, c.w, f3);
	o.x = (m & 1u) ? c.x : x0;
	o.y = (m & 2u) ? c.y : x1;
	o.z = (m & 4u) ? c.z : x2;
	o.w = (m & 8u) ? c.w : x3;
	return m | (f0 ? 1u : 0u) | (f1 ? 2u : 0u) | (f2 ? 4u : 0u) | (f3 ? 8u : 0u);
}

// A quad = cells 4 q .. 4 q + 3 of a row.  AL: X % 4 == 0, rows of whole 16-byte-aligned quads (128, 256, ...).  Otherwise (150^3, the
// reference's GI preset) a row's quads sit at any 4-byte alignment and its last quad has nv = X - 4 (X4 - 1) < 4 cells: loaded /
// stored cell by cell (the vector form would touch the next row, or run off the field's end), its missing cells carry the wall
// cell's value -- which makes the clamped right neighbour of the wall cell fall out of the quad arithmetic -- and count as frozen.
typedef float fz_v4 __attribute__((ext_vector_type(4)));
typedef fz_v4 __attribute__((aligned(4))) fz_v4u;

__device__ __forceinline__ float4 wall_patch(float4 c, int nv)
{
	if (nv < 2) c.y = c.x;
	if (nv < 3) c.z = c.y;
	if (nv < 4) c.w = c.z;
	return c;
}

// (cell offsets are 32-bit: uniform base + 32-bit byte offset is the `global_load v, v_offset, s[base]` form, and the 64-bit multiply-adds
// of a size_t offset run at quarter rate -- jacobi_freeze_supported() keeps the fields below 4 GiB)
template <bool AL>
__device__ __forceinline__ float4 ldq(const float* __restrict__ base_, uint32_t off, int nv)
{
	const float* __restrict__ base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base_) + ((size_t)off << 2));
	off = 0u;
	if (AL) return *reinterpret_cast<const float4*>(base);
	if (nv == 4) { const fz_v4u t = *reinterpret_cast<const fz_v4u*>(base + off); return make_float4(t.x, t.y, t.z, t.w); }
	float4 c;
	c.x = base[off];
	c.y = nv > 1 ? base[off + 1] : c.x;
	c.z = nv > 2 ? base[off + 2] : c.y;
	c.w = c.z;
	return c;
}

template <bool AL>
__device__ __forceinline__ void stq(float* __restrict__ base_, uint32_t off, int nv, float4 v)
{
	float* __restrict__ base = reinterpret_cast<float*>(reinterpret_cast<char*>(base_) + ((size_t)off << 2));
	off = 0u;
	if (AL) { *reinterpret_cast<float4*>(base) = v; return; }
	if (nv == 4) { fz_v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *reinterpret_cast<fz_v4u*>(base + off) = t; return; }
	base[off] = v.x;
	if (nv > 1) base[off + 1] = v.y;
	if (nv > 2) base[off + 2] = v.z;
}

// a velocity quad of a row of any length (AL = false: cell by cell where the quad is short or unaligned; the cells a short last quad
// does not have repeat the wall cell, which is what the clamped stencil reads there)
template <bool AL, int FUSE>
__device__ __forceinline__ float4 ldvq(const void* vel, uint32_t comp_cells, int comp, uint32_t off, int nv)
{
	if (AL) return ldv4<FUSE>(vel, comp_cells, comp, off);
	if (FUSE == 2) {
		const fz_h16* h = static_cast<const fz_h16*>(vel) + (size_t)comp * comp_cells + off;
		float4 c;
		if (nv == 4) {
			typedef fz_h16x4 __attribute__((aligned(2))) fz_h16x4u;
			const fz_h16x4u t = *reinterpret_cast<const fz_h16x4u*>(h);
			return make_float4((float)t.x, (float)t.y, (float)t.z, (float)t.w);
		}
		c.x = (float)h[0];
		c.y = nv > 1 ? (float)h[1] : c.x;
		c.z = nv > 2 ? (float)h[2] : c.y;
		c.w = nv > 3 ? (float)h[3] : c.z;
		return c;
	}
	return ldq<false>(static_cast<const float*>(vel) + (size_t)comp * comp_cells, off, nv);
}

// OR over the wave (every lane active), returned in every lane: four row rotations, then the four rows through SGPRs
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
	int x = (int)v;
	x |= __builtin_amdgcn_update_dpp(0, x, 0x121, 0xf, 0xf, false);      // row_ror:1
	x |= __builtin_amdgcn_update_dpp(0, x, 0x122, 0xf, 0xf, false);      // row_ror:2
	x |= __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, false);      // row_ror:4
	x |= __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, false);      // row_ror:8
	return (uint32_t)(__builtin_amdgcn_readlane(x, 0) | __builtin_amdgcn_readlane(x, 16) | __builtin_amdgcn_readlane(x, 32) | __builtin_amdgcn_readlane(x, 48));
}

// Work lists.  An entry = { tile | copy_only << 31, box, dirty, 0 }.  box = the core quads / rows / planes that still relax, as
// q0 | q1 << 3 | y0 << 6 | y1 << 9 | z0 << 12 | z1 << 15 (each 0..7): within T levels they depend on nothing further than one quad /
// T rows / T planes outside it, so only that much is staged and relaxed.  dirty = the box the PREVIOUS launch was given: the only
// core cells whose values can differ between the two pressure buffers, i.e. what this launch has to store (the relaxing set only
// shrinks, so dirty contains box).  Eight lists per launch, one per XCD (workgroup b runs on XCD b % 8): a tile stays with the XCD
// whose L2 holds what it stored last, and eight counters share the appends.
// floor(i / d) = (i * kMagic[d]) >> 20, exact for i < 2^20 / 16 and d <= 16 (kMagic[d] = ceil(2^20 / d)): the thread -> (quad, row,
// plane) decomposition over a box whose extents are only known at run time
__constant__ uint32_t kMagic[17] = { 0u, 1048576u, 524288u, 349526u, 262144u, 209716u, 174763u, 149797u, 131072u, 116509u, 104858u, 95326u,
	87382u, 80660u, 74899u, 69906u, 65536u };
__device__ __forceinline__ int div_magic(int i, uint32_t m) { return (int)(__umul24((uint32_t)i, m) >> 20); }   // (i < 2^12, m <= 2^20: the 24-bit multiply is exact and full rate)

constexpr uint32_t kCopyOnly = 0x80000000u;
constexpr uint32_t kFullBox = 0u | 7u << 3 | 0u << 6 | 7u << 9 | 0u << 12 | 7u << 15;
constexpr int kShards = 8;

// ---------------------------------------------------------------------------------------------------------------------------
// level 1, every cell.  Block = (bx quads, by rows), one plane per blockIdx slice, as k_jacobi_v4 (x neighbours by DPP).
// ---------------------------------------------------------------------------------------------------------------------------
// FUSE != 0 (whole steps): the divergence is computed here from the advected velocity -- k_divergence_v4's
// arithmetic, CSProject3D.hlsl:68-86 -- and written to `b_out` for the tile launches, instead of being read back from a launch of its own.
template <bool AL, int FUSE>
__global__ __launch_bounds__(256) void k_freeze_dense(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ pA, float* __restrict__ pB, uint8_t* __restrict__ mA, uint8_t* __restrict__ mB,
	uint32_t* __restrict__ tile_mark, uint32_t gen, uint32_t* __restrict__ cnt_clear, int n_clear, int ntx, int nty, int rows_per_block,
	const void* __restrict__ vel, float* __restrict__ b_out, int z_begin, int nzp, uint32_t vel_comp_cells)
{
	const int X4 = (g.X + 3) >> 2;
	const int lane = threadIdx.x;
	const int gx = (X4 + (int)blockDim.x - 1) / (int)blockDim.x, gy = (g.Y + rows_per_block - 1) / rows_per_block;
	if (blockIdx.x == 0)                                                 // the next solve's counters (this one's were cleared by the previous solve)
		for (int i = (int)(threadIdx.y * blockDim.x + threadIdx.x); i < n_clear; i += (int)(blockDim.x * blockDim.y)) cnt_clear[i] = 0u;
	// XCD k walks the k-th contiguous eighth of the (x, y, z)-ordered block sequence (see xcd_tile in fx_sim.hip)
	int t = (int)blockIdx.x;
	{
		const int n = gx * gy * nzp, q = n >> 3, r = n & 7, xcd = t & 7, j = t >> 3;
		t = xcd * q + min(xcd, r) + j;
	}
	const int x4 = (t % gx) * blockDim.x + lane;
	const int y = ((t / gx) % gy) * rows_per_block + threadIdx.y;
	const int z = z_begin + t / (gx * gy);                               // (a slab view: the owned planes only; z_begin = 0, nzp = Zg otherwise)
	const int wl = (int)((threadIdx.y * blockDim.x + threadIdx.x) & 63);
	const bool in = x4 < X4 && y < g.Y;
	uint32_t nib = 0xFu;
	int tile = -1;
	uint32_t qi = 0;
	if (in) {
		const uint32_t plane = (uint32_t)g.plane();
		const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
		const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
		const uint32_t zrow = (uint32_t)z * plane;
		const uint32_t c_off = zrow + (uint32_t)y * g.X + 4 * x4;
		const int nv = x4 == X4 - 1 ? g.X - 4 * (X4 - 1) : 4;
		const float4 c = ldq<AL>(p_in, c_off, nv);
		const float4 U = ldq<AL>(p_in, zrow + (uint32_t)yu * g.X + 4 * x4, nv);
		const float4 D = ldq<AL>(p_in, zrow + (uint32_t)yd * g.X + 4 * x4, nv);
		const float4 F = ldq<AL>(p_in, (uint32_t)zf * plane + (uint32_t)y * g.X + 4 * x4, nv);
		const float4 Bk = ldq<AL>(p_in, (uint32_t)zb * plane + (uint32_t)y * g.X + 4 * x4, nv);
		float4 bb;
		if (FUSE) {
			const uint32_t vcells = vel_comp_cells;
			const float4 cx = ldvq<AL, FUSE>(vel, vcells, 0, c_off, nv);
			const float4 vU = ldvq<AL, FUSE>(vel, vcells, 1, zrow + (uint32_t)yu * g.X + 4 * x4, nv);
			const float4 vD = ldvq<AL, FUSE>(vel, vcells, 1, zrow + (uint32_t)yd * g.X + 4 * x4, nv);
			const float4 vF = ldvq<AL, FUSE>(vel, vcells, 2, (uint32_t)zf * plane + (uint32_t)y * g.X + 4 * x4, nv);
			const float4 vB = ldvq<AL, FUSE>(vel, vcells, 2, (uint32_t)zb * plane + (uint32_t)y * g.X + 4 * x4, nv);
			float vL = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cx.w), 0x138, 0xf, 0xf, false));
			float vR = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cx.x), 0x130, 0xf, 0xf, false));
			if (x4 == 0) vL = cx.x; else if (wl == 0 || lane == 0) vL = ldv1<FUSE>(vel, c_off - 1);
			if (x4 == X4 - 1) vR = cx.w; else if (wl == 63 || lane == (int)blockDim.x - 1) vR = ldv1<FUSE>(vel, c_off + 4);
			bb.x = 0.5f * ((-vF.x + vB.x) + ((-vU.x + vD.x) + (-vL + cx.y)));
			bb.y = 0.5f * ((-vF.y + vB.y) + ((-vU.y + vD.y) + (-cx.x + cx.z)));
			bb.z = 0.5f * ((-vF.z + vB.z) + ((-vU.z + vD.z) + (-cx.y + cx.w)));
			bb.w = 0.5f * ((-vF.w + vB.w) + ((-vU.w + vD.w) + (-cx.z + vR)));
			stq<AL>(b_out, c_off, nv, bb);
		} else bb = ldq<AL>(b, c_off, nv);
		// x neighbours: the adjacent quad sits in the adjacent lane (DPP wave_shr:1 / wave_shl:1); only a wave's first / last lane
		// inside a row still loads them
		float L = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.w), 0x138, 0xf, 0xf, false));
		float R = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
		if (x4 == 0) L = c.x; else if (wl == 0 || lane == 0) L = p_in[c_off - 1];
		if (x4 == X4 - 1) R = c.w; else if (wl == 63 || lane == (int)blockDim.x - 1) R = p_in[c_off + 4];
		float4 o;
		nib = relax_quad(c, L, R, U, D, F, Bk, bb, (0xFu << nv) & 0xFu, o);     // (cells a short last quad does not have: frozen)
		stq<AL>(pA, c_off, nv, o);
		if (pB) stq<AL>(pB, c_off, nv, o);                                // (null: a masked strip launch follows, which reads one copy and writes both)
		qi = ((uint32_t)z * g.Y + y) * X4 + x4;
		tile = ((z >> 3) * nty + (y >> 3)) * ntx + (x4 >> 3);
	}
	// the mask bytes of four adjacent quads leave as one dword where the row allows it (X4 % 4 == 0: every aligned lane quartet
	// lies inside one row, inside the grid or outside it as a whole)
	if ((X4 & 3) == 0) {
		const int v = (int)nib;
		const uint32_t packed = nib | (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, false) << 8 |
			(uint32_t)__builtin_amdgcn_update_dpp(0, v, 0xAA, 0xf, 0xf, false) << 16 | (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0xFF, 0xf, 0xf, false) << 24;
		if (in && (wl & 3) == 0) {
			*reinterpret_cast<uint32_t*>(mA + qi) = packed;
			if (mB) *reinterpret_cast<uint32_t*>(mB + qi) = packed;
		}
	} else if (in) { mA[qi] = (uint8_t)nib; if (mB) mB[qi] = (uint8_t)nib; }
	// a tile with a relaxing cell is flagged for the first tile launch: one plain store per run of lanes that share a tile (every
	// writer stores the same tag; no atomics, no list here -- with them the sweep took 72 us on a young plume and 95 us on a developed
	// one, the difference being the marking)
	const bool active = nib != 0xFu;
	const int mine = active ? tile : -1;
	const int prev = __shfl_up(mine, 1);
	if (active && (wl == 0 || prev != mine) && tile_mark[tile] != gen) tile_mark[tile] = gen;   // (the cached read keeps most of the ~64 stores per tile away from its one line)
	// (that a second sweep runs at all is reported by the first tile launch, per flagged tile: a returning device-scope access per wave
	// HERE was a 17-us floor under the sweep at 128^3 -- 24.5 against 7.2 us)
}

// ---------------------------------------------------------------------------------------------------------------------------
// T more levels on the listed tiles
// ---------------------------------------------------------------------------------------------------------------------------
template <int T, int NT, bool AL>
__global__ __launch_bounds__(NT, NT >= 1024 ? 4 : NT / 128) void k_freeze_tiles(const Geom g, const float* __restrict__ p_src, const float* __restrict__ b,
	float* __restrict__ p_dst, const uint8_t* __restrict__ m_src, uint8_t* __restrict__ m_dst,
	const uint4* __restrict__ list_in, const uint32_t* __restrict__ cnt_in, uint4* __restrict__ list_out, uint32_t* __restrict__ cnt_out, int cap,
	int ntx, int nty, int level_base, uint32_t* __restrict__ stat, uint32_t stat_hi, const uint32_t* __restrict__ tile_flag, uint32_t flag_gen, int ntiles, int scan_limit,
	int zo0, int zo1)                             // the planes whose cells count as this context's (a slab view: the owned ones; else all)
{
	constexpr int E = 8 + 2 * T;                 // staged rows per plane = staged planes
	constexpr int NQ = E * E * TQ;               // staged quads
	constexpr int EB = E - 2, NB = EB * EB * TQ; // b is needed one cell less deep
	constexpr int NW = NT / 64;
	__shared__ __attribute__((aligned(16))) float4 Pq[NQ + 2];    // [1 + idx]: the x neighbours of a row's first / last quad stay inside the array
	__shared__ __attribute__((aligned(16))) float4 Bq[NB];
	__shared__ uint8_t Mq[NQ];
	__shared__ uint32_t wave_bits[NW];
	const int tid = threadIdx.x;
	PROF_INIT;
	const int X4 = (g.X + 3) >> 2, nv_last = g.X - 4 * (X4 - 1);   // quads per row; cells of a row's last quad
	const uint32_t plane4 = (uint32_t)(X4 * g.Y);   // quads per plane (mask bytes)
	const uint32_t plane = (uint32_t)g.plane();
	// a quad's first cell / its mask byte (24-bit multiplies: full rate; X * Y < 2^24 is part of jacobi_freeze_supported())
	auto foff = [&](int x4, int y, int z) -> uint32_t { return __umul24((uint32_t)z, plane) + __umul24((uint32_t)y, (uint32_t)g.X) + 4u * (uint32_t)x4; };
	auto moff = [&](int x4, int y, int z) -> uint32_t { return __umul24((uint32_t)z, plane4) + __umul24((uint32_t)y, (uint32_t)X4) + (uint32_t)x4; };
	const int shard = (int)(blockIdx.x & (kShards - 1)), wg = (int)(blockIdx.x >> 3), nwg = (int)(gridDim.x >> 3);
	// The solve's FIRST tile launch (tile_flag != null) has no list yet: every workgroup looks at its share of the tile flags the dense
	// sweep set (tiles w, w + grid, ...: neighbours in the grid go to different workgroups) and takes the flagged ones whole.  Later
	// launches walk the list their predecessor appended to, one of eight per-XCD sub-lists per workgroup; the first entry is loaded
	// together with the count it is checked against.  (Leaving the tiles at fixed list positions -- no counters per launch, no
	// returning atomic at a tile's end -- measured 4-10 % SLOWER than appending the survivors to a fresh list: 256^3 0.847 against
	// 0.815 ms per step, 128^3 0.311 against 0.283, same box.)
	const uint4* my_list = list_in + (size_t)shard * cap;
	uint4 entry = make_uint4(0u, 0u, 0u, 0u);
	uint32_t first, limit, stride, raised = 0u;
	if (tile_flag) { first = blockIdx.x; limit = (uint32_t)scan_limit; stride = gridDim.x; }
	else { if (wg < cap) entry = my_list[wg]; first = (uint32_t)wg; limit = cnt_in[shard]; stride = (uint32_t)nwg; }

	for (uint32_t e = first; e < limit; e += stride) {
		if (tile_flag) {
			uint32_t t = e;
			{   // XCD k (= e & 7) takes the tile planes k, k + 8, ...: x and y neighbours share its L2, and a plume in the middle of the grid is spread
			    // over all eight (tile index % 8 alone hands each XCD ONE x-column of tiles at 256^3: 0.489 against 0.340 ms of Jacobi per step)
				const uint32_t per_plane = (uint32_t)(ntx * nty), j = e >> 3, pl = j / per_plane;
				t = (pl * 8u + (e & 7u)) * per_plane + (j - pl * per_plane);
				if (t >= (uint32_t)ntiles) continue;
			}
			if (tile_flag[t] != flag_gen) continue;
			entry = make_uint4(t, kFullBox, kFullBox, 0u);
		} else if (e != first) entry = my_list[e];
		const int t = (int)(entry.x & ~kCopyOnly);
		const int tx = t % ntx, ty = (t / ntx) % nty, tz = t / (ntx * nty);
		// core quads / rows / planes the previous launch may have changed (core indices 0..7)
		const int dq0 = (int)(entry.z & 7u), dq1 = (int)((entry.z >> 3) & 7u), dy0 = (int)((entry.z >> 6) & 7u), dy1 = (int)((entry.z >> 9) & 7u);
		const int dz0 = (int)((entry.z >> 12) & 7u), dz1 = (int)((entry.z >> 15) & 7u);
		// ---- a tile that froze completely in the previous launch: carry its core across, then it is settled ---------------------
		if (entry.x & kCopyOnly) {
			PROF_COPY;
#pragma unroll
			for (int j = 0; j < (512 + NT - 1) / NT; ++j) {
				const int i = tid + NT * j;                               // 512 core quads
				if (i >= 512) break;
				const int q = i & 7, yy = (i >> 3) & 7, zz = i >> 6;
				const int x4 = tx * 8 + q, y = ty * TCY + yy, z = tz * TCZ + zz;
				if (x4 < X4 && y < g.Y && z < g.Zg && q >= dq0 && q <= dq1 && yy >= dy0 && yy <= dy1 && zz >= dz0 && zz <= dz1) {
					const uint32_t qi = moff(x4, y, z);
					const int nv = x4 == X4 - 1 ? nv_last : 4;
					stq<AL>(p_dst, foff(x4, y, z), nv, ldq<AL>(p_src, foff(x4, y, z), nv));
					m_dst[qi] = m_src[qi];
				}
			}
			continue;
		}
		PROF(0); PROF_COUNT;
		// every tile that gets here appends exactly one entry (itself, or itself as copy-only): its slot is reserved now, so that the
		// returning atomic's round trip passes behind the staging loads instead of at the tile's end (3.5 of a tile pass's 20 us)
		uint32_t slot;
		asm volatile("" : "=v"(slot));                                  // (no value: a `= 0` here would become a register copy at the join, i.e. a wait for the atomic on the spot)
		const int x40 = tx * 8 - 1, y0 = ty * TCY - T, z0 = tz * TCZ - T;   // quad / row / plane of staged index 0
		// the box of core cells that still relax (core indices) ...
		const int aq0 = (int)(entry.y & 7u), aq1 = (int)((entry.y >> 3) & 7u), ay0 = (int)((entry.y >> 6) & 7u), ay1 = (int)((entry.y >> 9) & 7u);
		const int az0 = (int)((entry.y >> 12) & 7u), az1 = (int)((entry.y >> 15) & 7u);
		// ... and what has to be staged, in staged indices: that box grown by one quad / T rows / T planes (all its cells can depend on
		// within T levels) and the dirty box (stored from the LDS)
		const int rq0 = min(aq0, dq0 + 1), rq1 = max(aq1 + 2, dq1 + 1);
		const int ry0 = min(ay0, dy0 + T), ry1 = max(ay1 + 2 * T, dy1 + T);
		const int rz0 = min(az0, dz0 + T), rz1 = max(az1 + 2 * T, dz1 + T);
		int tl = tid;
		asm volatile("" : "+v"(tl));                                     // (keeps the index arithmetic below inside the tile loop: hoisted, it spills)
		// ---- stage the cone ---------------------------------------------------------------------------------------------------
		__syncthreads(); PROF(1);
		if (tid == 0) slot = atomicAdd(cnt_out + shard, 1u);                                                 // the previous tile of this workgroup is done with the LDS
		{
			// every load is issued before the first LDS store; lanes beyond the box repeat its last quad.  Cells of the box outside
			// the grid are marked frozen: no cell inside reads them (clamped taps).  What lies outside the box keeps whatever the LDS
			// held: further than T cells from every relaxing core cell, it cannot reach one within T levels.
			constexpr int SJ = (NQ + NT - 1) / NT;
			const int nqs = rq1 - rq0 + 1, nys = ry1 - ry0 + 1, total_s = nqs * nys * (rz1 - rz0 + 1);
			const uint32_t mq = kMagic[nqs], my = kMagic[nys];
			float4 sv[SJ], sb[SJ];
			uint32_t sm[SJ];
			int si[SJ];
#pragma unroll
			for (int j = 0; j < SJ; ++j) {
				// no branch around the loads: a join after them makes the compiler move the loaded registers, i.e. wait for each of the
				// SJ rounds in turn (measured: 7.5 of a tile pass's 20 us).  Rounds beyond the box re-read its last quad (one cached line).
				const int i = min(tl + NT * j, total_s - 1);
				const int r = div_magic(i, mq), q = rq0 + i - r * nqs, zr = div_magic(r, my), yy = ry0 + r - zr * nys, zz = rz0 + zr;
				const int x4 = x40 + q, y = y0 + yy, z = z0 + zz;
				const int xc = min(max(x4, 0), X4 - 1), yc = min(max(y, 0), g.Y - 1), zc = min(max(z, 0), g.Zg - 1);
				const int nv = xc == X4 - 1 ? nv_last : 4;
				sv[j] = ldq<AL>(p_src, foff(xc, yc, zc), nv);
				sb[j] = ldq<AL>(b, foff(xc, yc, zc), nv);
				sm[j] = m_src[moff(xc, yc, zc)];
				// LDS index | "outside the grid" | "no b row here"; -1 = this lane has no quad in this round
				const bool in = x4 >= 0 && x4 < X4 && y >= 0 && y < g.Y && z >= 0 && z < g.Zg;
				const bool brow = yy >= 1 && yy < E - 1 && zz >= 1 && zz < E - 1;
				const int code = ((zz * E + yy) * TQ + q) | (in ? 0 : 0x20000000) | (brow ? 0 : 0x40000000);
				si[j] = tl + NT * j < total_s ? code : -1;
			}
			PROF(10);
			// (the compiler otherwise sinks each load into the guarded store below and waits for them one by one)
#pragma unroll
			for (int j = 0; j < SJ; ++j)
				asm volatile("" : "+v"(sv[j].x), "+v"(sv[j].y), "+v"(sv[j].z), "+v"(sv[j].w), "+v"(sb[j].x), "+v"(sb[j].y), "+v"(sb[j].z), "+v"(sb[j].w), "+v"(sm[j]));
			PROF(2);
#pragma unroll
			for (int j = 0; j < SJ; ++j) {
				if (si[j] >= 0) {
					const int idx = si[j] & 0xFFFFF;
					Pq[1 + idx] = sv[j];
					Mq[idx] = (uint8_t)((si[j] & 0x20000000) ? 0xFu : sm[j]);
					if (!(si[j] & 0x40000000)) {
						const int q = idx % TQ, r = idx / TQ, yy = r % E, zz = r / E;
						Bq[((zz - 1) * EB + (yy - 1)) * TQ + q] = sb[j];
					}
				}
			}
		}
		__syncthreads(); PROF(3);
		// ---- T levels in the LDS: level k on the box grown by T - k rows / planes (one quad, until the last level) -----------------
		int last_active = 0;                                             // last level (1..T) that left a core cell relaxing
		uint32_t core_bits = 0;                                          // quads | rows << 8 | planes << 16 of the core that still relax
#pragma unroll
		for (int k = 1; k <= T; ++k) {
			const int cq0 = k == T ? aq0 + 1 : aq0, nq = aq1 - aq0 + (k == T ? 1 : 3);
			const int cy0 = ay0 + k, ny = ay1 - ay0 + 1 + 2 * (T - k), cz0 = az0 + k, nz = az1 - az0 + 1 + 2 * (T - k);
			const int total = nq * ny * nz;
			const uint32_t mq = kMagic[nq], my = kMagic[ny];
			constexpr int MAXJ = ((E - 2) * (E - 2) * TQ + NT - 1) / NT;
			float4 nv[MAXJ];
			uint32_t nm[MAXJ];
			uint32_t bits = 0;
#pragma unroll
			for (int j = 0; j < MAXJ; ++j) {
				const int i = tl + NT * j;
				nm[j] = 0x100u;                                            // "nothing to write"
				if (NT * j < total) {                                      // uniform
					if (i < total) {
						const int r = div_magic(i, mq), q = cq0 + i - r * nq, zr = div_magic(r, my), yy = cy0 + r - zr * ny, zz = cz0 + zr;
						const int idx = (zz * E + yy) * TQ + q;
						const uint32_t m = Mq[idx];
						if (m != 0xFu) {
							const int x4 = x40 + q, y = y0 + yy, z = z0 + zz;
							float4 c = Pq[1 + idx];
							if (!AL && x4 == X4 - 1) c = wall_patch(c, nv_last);   // the missing cells of a short last quad follow the wall cell
							const float4 U = Pq[1 + (y == 0 ? idx : idx - TQ)];
							const float4 D = Pq[1 + (y == g.Y - 1 ? idx : idx + TQ)];
							const float4 F = Pq[1 + (z == 0 ? idx : idx - E * TQ)];
							const float4 Bk = Pq[1 + (z == g.Zg - 1 ? idx : idx + E * TQ)];
							const float L = x4 == 0 ? c.x : reinterpret_cast<const float*>(Pq)[4 * idx + 3];          // .w of quad idx - 1
							const float R = x4 == X4 - 1 ? c.w : reinterpret_cast<const float*>(Pq)[4 * (idx + 2)];    // .x of quad idx + 1
							const float4 bb = Bq[((zz - 1) * EB + (yy - 1)) * TQ + q];
							nm[j] = relax_quad(c, L, R, U, D, F, Bk, bb, m, nv[j]) | (uint32_t)(idx << 9);
							const bool core = q >= 1 && q <= 8 && yy >= T && yy < T + TCY && zz >= T && zz < T + TCZ && z >= zo0 && z < zo1;
							if (core && (nm[j] & 0xFu) != 0xFu) bits |= 1u << (q - 1) | 0x100u << (yy - T) | 0x10000u << (zz - T);
						}
					}
				}
			}
			__syncthreads();                                               // every read of level k - 1 is done
#pragma unroll
			for (int j = 0; j < MAXJ; ++j) {
				if (nm[j] != 0x100u) {
					const int idx = (int)(nm[j] >> 9);
					Pq[1 + idx] = nv[j];
					Mq[idx] = (uint8_t)(nm[j] & 0xFu);
				}
			}
			const uint32_t wb = wave_or(bits);
			if ((tid & 63) == 0) wave_bits[tid >> 6] = wb;
			__syncthreads();
			core_bits = 0;
#pragma unroll
			for (int w = 0; w < NW; ++w) core_bits |= wave_bits[w];
			if (core_bits) last_active = k;
			PROF(3 + k);
			if (!core_bits) break;                                         // the core is frozen: deeper levels cannot change it
		}
		// ---- store the core ---------------------------------------------------------------------------------------------------
#pragma unroll
		for (int j = 0; j < (512 + NT - 1) / NT; ++j) {
			const int i = tl + NT * j;
			if (i >= 512) break;
			const int q = 1 + (i & 7), yy = T + ((i >> 3) & 7), zz = T + (i >> 6);
			const int x4 = x40 + q, y = y0 + yy, z = z0 + zz;
			if (x4 < X4 && y < g.Y && z < g.Zg && q - 1 >= dq0 && q - 1 <= dq1 && yy - T >= dy0 && yy - T <= dy1 && zz - T >= dz0 && zz - T <= dz1) {
				const int idx = (zz * E + yy) * TQ + q;
				const uint32_t qi = moff(x4, y, z);
				stq<AL>(p_dst, foff(x4, y, z), x4 == X4 - 1 ? nv_last : 4, Pq[1 + idx]);
				m_dst[qi] = Mq[idx];
			}
		}
		PROF(8);
		if (tid == 0) {
			uint4 next = make_uint4((uint32_t)t | kCopyOnly, 0u, entry.y, 0u);     // what this launch was given is what the next one must store
			if (core_bits) {
				const uint32_t qm = core_bits & 0xFFu, ym = (core_bits >> 8) & 0xFFu, zm = (core_bits >> 16) & 0xFFu;
				next.x = (uint32_t)t;
				next.y = (uint32_t)(__ffs(qm) - 1) | (uint32_t)(31 - __clz(qm)) << 3 | (uint32_t)(__ffs(ym) - 1) << 6 |
					(uint32_t)(31 - __clz(ym)) << 9 | (uint32_t)(__ffs(zm) - 1) << 12 | (uint32_t)(31 - __clz(zm)) << 15;
			}
			list_out[(size_t)shard * cap + slot] = next;
			// the last level that left a cell relaxing; a tile the dense sweep flagged had one after level `level_base` itself
			if ((last_active > 0 || tile_flag) && stat_hi + (uint32_t)(level_base + last_active) > raised) {
				raised = stat_hi + (uint32_t)(level_base + last_active);       // (one returning access per workgroup and level, not per tile)
				atomicMax(stat, raised);                                      // (no return value: nobody waits for it)
			}
		}
		PROF(9);
	}
}

}
#ifdef FX_FREEZE_PROF
extern "C" void fx_debug_freeze_prof(unsigned long long* out, int reset) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fz_prof), sizeof fz_prof); if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fz_prof), z, sizeof z); } }
extern "C" void fx_debug_freeze_passes(unsigned int* out, int reset) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fz_passes), sizeof fz_passes); if (reset) { static unsigned int z[2][80]; (void)hipMemcpyToSymbol(HIP_SYMBOL(fz_passes), z, sizeof z); } }
#endif

// The planes a context holds, as a grid of their own (the header's "view"): `first` = local plane index of the view's plane 0 (what the
// field pointers are advanced by), `own0` = the view plane of the first owned plane.  A single-domain context is its own view.
Geom jacobi_freeze_view(const Geom& g, int* first, int* own0)
{
	const int n = g.zhi - g.zlo + 1;
	if (first) *first = g.lz(g.zlo);
	if (own0) *own0 = g.z0 - g.zlo;
	return Geom{ g.X, g.Y, n, 0, n, 0, 0, n - 1 };
}

bool jacobi_freeze_supported(const Geom& g)
{
	const bool slab = g.nz != g.Zg;
	// a slab's cones reach T <= 4 planes into the halo; its halo planes must be exchanged ones on every interior side
	if (slab && (g.H < 4 || (g.z0 > 0 && g.z0 - g.zlo < 4) || (g.z0 + g.nz < g.Zg && g.zhi - (g.z0 + g.nz - 1) < 4))) return false;
	const uint64_t planes = (uint64_t)g.nzl();
	return FX_KNOB_INT("FREEZE_FAST", 1) != 0 && g.Zg > 1 && g.X >= 4 &&
		(uint64_t)g.X * g.Y < (1u << 24) && (uint64_t)g.X * g.Y * planes < (1u << 30);   // 24-bit row / plane multiplies, 32-bit byte offsets
}

int jacobi_freeze_tiles(const Geom& g)                                // (sized for every plane the context allocates)
{
	return ((g.X + TCX - 1) / TCX) * ((g.Y + TCY - 1) / TCY) * ((g.nzl() + TCZ - 1) / TCZ);
}

size_t jacobi_freeze_mask_bytes(const Geom& g) { return (size_t)((g.X + 3) / 4) * g.Y * (size_t)g.nzl(); }
size_t jacobi_freeze_list_bytes(const Geom& g) { return (size_t)kShards * (size_t)jacobi_freeze_tiles(g) * sizeof(uint4); }
size_t jacobi_freeze_count_words() { return (size_t)kFreezeSlots * kShards; }

int jacobi_freeze_levels_per_launch()
{
	const int v = FX_KNOB_INT("FREEZE_T", 4);        // read per call: the tests switch it inside one process
	return v < 1 ? 1 : (v > 4 ? 4 : v);
}

hipError_t launch_freeze_dense(const Geom& g, const float* p_in, const float* b, float* pA, float* pB, uint8_t* mA, uint8_t* mB,
	const FreezeWork& w, hipStream_t s, const void* vel, int vel_half, int z_begin, int nzp, size_t vel_comp_cells)
{
	if (nzp <= 0) { z_begin = 0; nzp = g.Zg; }
	if (!vel_comp_cells) vel_comp_cells = g.cells_local();
	const int X4 = (g.X + 3) >> 2;
	const int bx = X4 < 64 ? X4 : 64;
	int by = 256 / bx; if (by < 1) by = 1; if (by > g.Y) by = g.Y;
	const int ntx = (g.X + TCX - 1) / TCX, nty = (g.Y + TCY - 1) / TCY;
	const dim3 block(bx, by, 1), grid(((X4 + bx - 1) / bx) * ((g.Y + by - 1) / by) * nzp, 1, 1);
	float* b_out = const_cast<float*>(b);
#define FX_DENSE(AL_, F_) hipLaunchKernelGGL((k_freeze_dense<AL_, F_>), grid, block, 0, s, g, p_in, b, pA, pB, mA, mB, w.tile_mark, w.gen, \
		w.counts_next, kFreezeSlots * kShards, ntx, nty, by, vel, b_out, z_begin, nzp, (uint32_t)vel_comp_cells)
	if ((g.X & 3) != 0) { if (!vel) FX_DENSE(false, 0); else if (vel_half) FX_DENSE(false, 2); else FX_DENSE(false, 1); }
	else if (!vel) FX_DENSE(true, 0);
	else if (vel_half) FX_DENSE(true, 2);
	else FX_DENSE(true, 1);
#undef FX_DENSE
	return hipGetLastError();
}

bool jacobi_freeze_can_fuse_divergence(const Geom& g) { (void)g; return FX_KNOB_INT("FREEZE_FUSE_DIV", 1) != 0; }

// launch number `n` (0, 1, ...) of a solve reads list[n & 1] and writes list[(n + 1) & 1]
hipError_t launch_freeze_tiles(const Geom& g, const float* p_src, const float* b, float* p_dst, const uint8_t* m_src, uint8_t* m_dst,
	const FreezeWork& w, int n, int levels, int level_base, uint32_t* stat, uint32_t stat_hi, hipStream_t s, int z_begin, int nzp, uint32_t flag_tag)
{
	if (nzp <= 0) { z_begin = 0; nzp = g.Zg; }
	if (!flag_tag) flag_tag = w.gen;
	int max_wgs = FX_KNOB_INT("FREEZE_WGS", 2048);
	// the relaxing set only shrinks: later launches of a solve get smaller grids (an empty or nearly empty launch of 2048 workgroups of
	// 512 threads costs 4-8 us just to start and retire them; a workgroup walks its list, so fewer workgroups still cover every entry)
	if (FX_KNOB_INT("FREEZE_SHRINK", 1)) max_wgs = std::max(256, max_wgs >> std::min(n / 4, 3));
	max_wgs = max_wgs < kShards ? kShards : (max_wgs & ~(kShards - 1));
	const int ntx = (g.X + TCX - 1) / TCX, nty = (g.Y + TCY - 1) / TCY, ntiles = ntx * nty * ((g.Zg + TCZ - 1) / TCZ);
	const int want = (ntiles + kShards - 1) / kShards * kShards;
	const int nt = FX_KNOB_INT("FREEZE_NT", 512);
	const dim3 block((g.X & 3) != 0 ? 512 : (nt == 256 ? 256 : (nt == 1024 ? 1024 : 512)), 1, 1), grid(want < max_wgs ? want : max_wgs, 1, 1);
	const int ntz = (g.Zg + TCZ - 1) / TCZ;
	const int scan_limit = (ntz + 7) / 8 * 8 * ntx * nty;   // the first launch's index space: (tile plane % 8 = XCD, plane / 8, tile in plane)
	const uint4* lin = (const uint4*)w.list[n & 1];
	uint4* lout = (uint4*)w.list[(n + 1) & 1];
	const uint32_t* cin = w.counts + (size_t)n * kShards;
	uint32_t* cout = w.counts + (size_t)(n + 1) * kShards;
#define FX_FREEZE_ARGS grid, block, 0, s, g, p_src, b, p_dst, m_src, m_dst, lin, cin, lout, cout, w.cap, ntx, nty, level_base, stat, stat_hi, n == 0 ? w.tile_mark : nullptr, flag_tag, ntiles, scan_limit, z_begin, z_begin + nzp
#define FX_FREEZE_LAUNCH(T) if ((g.X & 3) != 0) hipLaunchKernelGGL((k_freeze_tiles<T, 512, false>), FX_FREEZE_ARGS); \
	else if (nt == 256) hipLaunchKernelGGL((k_freeze_tiles<T, 256, true>), FX_FREEZE_ARGS); \
	else if (nt == 1024) hipLaunchKernelGGL((k_freeze_tiles<T, 1024, true>), FX_FREEZE_ARGS); \
	else hipLaunchKernelGGL((k_freeze_tiles<T, 512, true>), FX_FREEZE_ARGS)
	switch (levels) {
	case 1: FX_FREEZE_LAUNCH(1); break;
	case 2: FX_FREEZE_LAUNCH(2); break;
	case 3: FX_FREEZE_LAUNCH(3); break;
	case 4: FX_FREEZE_LAUNCH(4); break;
	default: return hipErrorInvalidValue;
	}
#undef FX_FREEZE_LAUNCH
#undef FX_FREEZE_ARGS
	return hipGetLastError();
}

}  // namespace fx
