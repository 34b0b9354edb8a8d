// fx_render_accel.hip -- the ray marches as they run by default: bit-identical to the plain kernels of fx_render.hip (and thereby to
// CSRayMarchL.hlsl:15-80, CSRayMarch.hlsl:98-196, CSRayMarchV.hlsl:5-7, PSRayCast(V).hlsl -- the arithmetic is fx_march.h's for both),
// organised around what the plain kernels measured as their bounds ("render"):
//
//   * the view march is a chain of DEPENDENT memory round trips (at frame 132 of the 256^3 run 19 M of its 19.3 M samples fall into
//     empty space, each a global look-up the next step waits for): the occupancy of the volume is kept as bit masks -- one bit per
//     4^3 block, 32 KiB at 256^3 -- that every workgroup copies into its LDS, so an empty sample costs arithmetic and one ds_read;
//     the samples that do gather fetch colour and light map in one round trip.
//   * the light pass read alpha as one float of a 16-byte texel -- four useful bytes per sixteen through the vector L1, which was
//     its bound (30 M L1 accesses in 199 us): k_occupancy_blocks, which reads every alpha anyway, writes an alpha-only fp32 side
//     volume; every density tap of every march reads that.
//   * 98 % of the light-map voxels are empty and receive a constant: k_light_classify writes it and appends the lit voxels to the
//     list of their z plane (one counter per plane, a cache line apart: a single list head serialises at ~90 atomics per us,
//     which alone cost 170 us); k_light_march runs full waves over the lists, chunks of 64 voxels dealt out round-robin.
//
// Scratch (RenderAccel, owned by the context): alpha volume, fine occupancy grid, the masks, the voxel list.
#include "fx_march.h"
#include <algorithm>

namespace fx {

#ifndef FX_MASK_BUDGET_BITS
#define FX_MASK_BUDGET_BITS 262144      // 32 KiB of LDS per mask
#endif
static const uint32_t kMaskBudgetBits = FX_MASK_BUDGET_BITS;
// samples a gathering lane fetches per round trip (fx_march.h)
#ifndef FX_VIEW_AHEAD
#define FX_VIEW_AHEAD 2
#endif
#ifndef FX_LIGHT_AHEAD
#define FX_LIGHT_AHEAD 2
#endif
static const int kViewAhead = FX_VIEW_AHEAD, kLightAhead = FX_LIGHT_AHEAD;
#ifndef FX_LIGHT_RAY_UNROLL
#define FX_LIGHT_RAY_UNROLL 2           // samples a live lane takes between two looks at the refill state
#endif
#ifndef FX_LIGHT_RAY_WGS
#define FX_LIGHT_RAY_WGS 1024
#endif
static const size_t kLightRayWorkgroups = FX_LIGHT_RAY_WGS;   // persistent workgroups of the refilling shadow-ray march
// RenderAccel::ctr, one 128-byte line per counter: [Zg] lengths of the per-plane lists of lit voxels, [8] work heads of the view march,
// [CZ] lengths of the per-layer lists of occupied 4^3 cells
static const int kCntStride = 32;
__host__ __device__ static inline size_t ctr_heads(const Geom& g) { return (size_t)g.Zg * kCntStride; }
__host__ __device__ static inline size_t ctr_cells(const Geom& g) { return ((size_t)g.Zg + 8) * kCntStride; }
// (two sets, RenderAccel::frame & 1: every build pass clears the list lengths of the next render's set, so that a pass which appends
// while it sweeps -- k_build_fill -- finds its own set empty)
static size_t ctr_set_words(const Geom& g) { return ctr_cells(g) + (size_t)((g.Zg + 3) >> 2) * kCntStride; }
size_t render_accel_ctr_words(const Geom& g) { return 2 * ctr_set_words(g); }
static uint32_t* ctr_now(const RenderAccel& a, const Geom& g) { return a.ctr + (a.frame & 1u) * ctr_set_words(g); }
static uint32_t* ctr_next(const RenderAccel& a, const Geom& g) { return a.ctr + ((a.frame & 1u) ^ 1u) * ctr_set_words(g); }

void render_accel_layout(const Geom& g, RenderAccel* a)
{
	a->CX = (g.X + 3) >> 2; a->CY = (g.Y + 3) >> 2; a->CZ = (g.Zg + 3) >> 2;
	const size_t n = (size_t)a->CX * a->CY * a->CZ;
	a->fine_words = (uint32_t)(((n + 127) / 128) * 4);            // whole 16-byte groups: the LDS fill moves uint4
	int sh = 0;
	size_t m = n;
	int MX = a->CX, MY = a->CY, MZ = a->CZ;
	while (m > kMaskBudgetBits) {
		++sh;
		MX = (a->CX + (1 << sh) - 1) >> sh; MY = (a->CY + (1 << sh) - 1) >> sh; MZ = (a->CZ + (1 << sh) - 1) >> sh;
		m = (size_t)MX * MY * MZ;
	}
	a->msh = sh; a->MX = MX; a->MY = MY; a->MZ = MZ;
	a->mask_words = sh ? (uint32_t)(((m + 127) / 128) * 4) : a->fine_words;
}

size_t render_accel_bits_words(const RenderAccel& a) { return 2 * (size_t)a.fine_words + (a.msh ? 2 * (size_t)a.mask_words : 0); }

static const uint32_t* mask_pos(const RenderAccel& a) { return a.msh ? a.bits + 2 * (size_t)a.fine_words : a.bits; }
static const uint32_t* mask_vis(const RenderAccel& a) { return a.msh ? a.bits + 2 * (size_t)a.fine_words + a.mask_words : a.bits + a.fine_words; }

// ---- acceleration structures ----------------------------------------------------------------------------------------------
// occupancy grid: entry c bounds the alpha of the voxels [4c, 4c + 4] per axis -- everything a trilinear sample whose base tap
// lies in block c can touch.  k_occupancy_blocks reads every alpha once, coalesced along x (lane = x, each thread folds a
// 1 x 4 x 4 column, four lanes fold into one 4^3 block: no atomics) and writes it to the alpha side volume on the way;
// k_occupancy_dilate takes the max over the 2 x 2 x 2 blocks c .. c + 1 (a superset of [4c, 4c + 4]: conservative, which only
// skips less) and stores the two masks of the fine level by wave ballot.
// FROM_ALPHA: the side volume holds this colour field's alpha already (the advection that made the field wrote it, fx_advect_lds.hip
// <ALPHA>): 4 bytes per voxel to read instead of the texel, nothing to write but the block maxima.
template <bool HALF, bool FROM_ALPHA>
__global__ __launch_bounds__(256) void k_occupancy_blocks(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, float* __restrict__ blk,
	float* __restrict__ alpha, uint32_t* __restrict__ cnt, uint32_t* __restrict__ cnt_next)
{
	const int CX = (g.X + 3) >> 2, CY = (g.Y + 3) >> 2;
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4 && 4 * (int)blockIdx.z + (int)threadIdx.x < g.Zg) {
		cnt[(4 * blockIdx.z + threadIdx.x) * kCntStride] = 0u;                     // the light-voxel lists of this frame start empty
		cnt_next[(4 * blockIdx.z + threadIdx.x) * kCntStride] = 0u;                // ... and so will the next frame's (its build pass may append while it sweeps)
	}
	if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x >= 64 && threadIdx.x < 72)
		cnt[ctr_heads(g) + (threadIdx.x - 64) * kCntStride] = 0u;                  // ... the eight work heads of the view march at zero
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 72) cnt[ctr_cells(g) + blockIdx.z * kCntStride] = 0u;   // ... and the cell lists empty
	const int x = blockIdx.x * 64 + (threadIdx.x & 63);
	const int cy = blockIdx.y * 4 + (threadIdx.x >> 6), cz = blockIdx.z;
	auto fetch = [&](size_t i) -> float { return FROM_ALPHA ? alpha[i] : ColTex<HALF>::ldw(col, i); };
	float m = 0.0f;
	if (x < g.X && cy < CY) {
		if (4 * cz + 4 <= g.Zg && 4 * cy + 4 <= g.Y) {                             // the whole column: 16 independent loads in flight
			float a[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) a[k] = fetch(((size_t)(4 * cz + (k >> 2)) * g.Y + (4 * cy + (k & 3))) * g.X + x);
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				if (!FROM_ALPHA) alpha[((size_t)(4 * cz + (k >> 2)) * g.Y + (4 * cy + (k & 3))) * g.X + x] = a[k];
				m = fmaxf(m, a[k]);
			}
		} else {
			for (int z = 4 * cz; z < min(4 * cz + 4, g.Zg); ++z)
				for (int y = 4 * cy; y < min(4 * cy + 4, g.Y); ++y) {
					const size_t i = ((size_t)z * g.Y + y) * g.X + x;
					const float a = fetch(i);
					if (!FROM_ALPHA) alpha[i] = a;
					m = fmaxf(m, a);
				}
		}
	}
	m = fmaxf(m, __shfl_xor(m, 1));
	m = fmaxf(m, __shfl_xor(m, 2));
	if (x < g.X && cy < CY && (x & 3) == 0) blk[((size_t)cz * CY + cy) * CX + (x >> 2)] = m;
}

// the same from the side volume for rows of whole quads: a lane folds a whole 4^3 block from 16 float4s
__global__ __launch_bounds__(256) void k_occupancy_blocks_a4(const Geom g, float* __restrict__ blk, const float* __restrict__ alpha, uint32_t* __restrict__ cnt,
	uint32_t* __restrict__ cnt_next)
{
	const int CX = g.X >> 2, CY = (g.Y + 3) >> 2;
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4 && 4 * (int)blockIdx.z + (int)threadIdx.x < g.Zg) {
		cnt[(4 * blockIdx.z + threadIdx.x) * kCntStride] = 0u;
		cnt_next[(4 * blockIdx.z + threadIdx.x) * kCntStride] = 0u;
	}
	if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x >= 64 && threadIdx.x < 72)
		cnt[ctr_heads(g) + (threadIdx.x - 64) * kCntStride] = 0u;
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 72) cnt[ctr_cells(g) + blockIdx.z * kCntStride] = 0u;
	const int cx = blockIdx.x * 64 + (threadIdx.x & 63);
	const int cy = blockIdx.y * 4 + (threadIdx.x >> 6), cz = blockIdx.z;
	if (cx >= CX || cy >= CY) return;
	const int ny = min(4, g.Y - 4 * cy), nz = min(4, g.Zg - 4 * cz);
	float4 a[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		const int dz = min(k >> 2, nz - 1), dy = min(k & 3, ny - 1);             // (short blocks at the far faces fold a voxel twice)
		a[k] = *reinterpret_cast<const float4*>(alpha + ((size_t)(4 * cz + dz) * g.Y + (4 * cy + dy)) * g.X + 4 * cx);
	}
	float m = 0.0f;
#pragma unroll
	for (int k = 0; k < 16; ++k) m = fmaxf(m, fmaxf(fmaxf(a[k].x, a[k].y), fmaxf(a[k].z, a[k].w)));
	blk[((size_t)cz * CY + cy) * CX + cx] = m;
}

// Build pass and light-map fill in one sweep over the side volume (grids whose extents are powers of two: a voxel's centre sample --
// CSRayMarchL.hlsl:36-37 -- has all filter weights 0 and IS the voxel's alpha): a lane folds a 4^3 cell from 16 float4s like
// k_occupancy_blocks_a4, stores the unlit light-map value over the whole cell (the ray kernels overwrite the lit voxels behind it) and
// appends the voxels with alpha >= 0.01 (:44) to the list of their z plane -- what k_light_cells and k_light_classify did in two more
// passes.  The lists of this frame's counter set were cleared by the previous frame's build pass.
__global__ __launch_bounds__(256) void k_build_fill(const Geom g, float* __restrict__ blk, const float* __restrict__ alpha, uint32_t* __restrict__ cnt,
	uint32_t* __restrict__ cnt_next, uint32_t* __restrict__ list, uint32_t* __restrict__ lightmap, const FrameConsts fc, int has_sh,
	const unsigned long long* __restrict__ lit_prev, unsigned long long* __restrict__ lit_now)
{
	const int CX = g.X >> 2, CY = g.Y >> 2;
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4 && 4 * (int)blockIdx.z + (int)threadIdx.x < g.Zg)
		cnt_next[(4 * blockIdx.z + threadIdx.x) * kCntStride] = 0u;
	if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x >= 64 && threadIdx.x < 72)
		cnt[ctr_heads(g) + (threadIdx.x - 64) * kCntStride] = 0u;
	const int cx = blockIdx.x * 64 + (threadIdx.x & 63);
	const int cy = blockIdx.y * 4 + (threadIdx.x >> 6), cz = blockIdx.z;
	const uint32_t lane = threadIdx.x & 63u;
	const bool valid = cx < CX && cy < CY;                                         // (uniform per wave in y; x only on rows shorter than 256)
	const uint32_t X = (uint32_t)g.X, XY = X * (uint32_t)g.Y;
	float4 a[16];
	uint32_t lit[4] = { 0u, 0u, 0u, 0u };                                          // per plane of the cell: 16 bits, x fastest
	if (valid) {
#pragma unroll
		for (int k = 0; k < 16; ++k) a[k] = *reinterpret_cast<const float4*>(alpha + (uint32_t)(4 * cz + (k >> 2)) * XY + (uint32_t)(4 * cy + (k & 3)) * X + 4u * (uint32_t)cx);
		float m = 0.0f;
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			m = fmaxf(m, fmaxf(fmaxf(a[k].x, a[k].y), fmaxf(a[k].z, a[k].w)));
			const uint32_t b4 = (a[k].x >= 0.00999999978f ? 1u : 0u) | (a[k].y >= 0.00999999978f ? 2u : 0u) | (a[k].z >= 0.00999999978f ? 4u : 0u) | (a[k].w >= 0.00999999978f ? 8u : 0u);
			lit[k >> 2] |= b4 << (4 * (k & 3));
		}
		blk[((size_t)cz * CY + cy) * CX + cx] = m;
	}
	// The unlit value over the whole cell.  Frame after frame that value is the same and only the ray kernels write anything else, into
	// lit voxels: a cell without a lit voxel last time still holds it everywhere.  So each pass leaves a bit per cell "holds a lit voxel"
	// (a 64-bit word per wave = 64 cells of a row) and the next one rewrites only those cells -- 92 % of the light map's 67 MB stay
	// untouched at frame 132.  lit_prev null: no such record (first pass, another light path in between, other constants): every cell.
	const uint32_t wordi = ((uint32_t)cz * (uint32_t)CY + (uint32_t)cy) * gridDim.x + blockIdx.x;
	const unsigned long long had = lit_prev && cy < CY ? lit_prev[wordi] : ~0ull;
	const unsigned long long has = __ballot(valid && (lit[0] | lit[1] | lit[2] | lit[3]) != 0u);
	if (lane == 0 && cy < CY) lit_now[wordi] = has;
	if (valid && ((had >> lane) & 1ull)) {
		const float irr[3] = { 0.0f, 0.0f, 0.0f };
		const uint32_t e = light_value(fc, has_sh != 0, 1.0f, 1.0f, irr);
		const uint4 e4 = make_uint4(e, e, e, e);
#pragma unroll
		for (int k = 0; k < 16; ++k) *reinterpret_cast<uint4*>(lightmap + (uint32_t)(4 * cz + (k >> 2)) * XY + (uint32_t)(4 * cy + (k & 3)) * X + 4u * (uint32_t)cx) = e4;
	}
	if (has == 0) return;                                                          // most waves: nothing lit
#pragma unroll
	for (int p = 0; p < 4; ++p) {
		const uint32_t n = (uint32_t)__popc(lit[p]);
		if (__ballot(n != 0u) == 0) continue;
		uint32_t incl = n;                                                         // inclusive scan over the lanes
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)incl, d); if ((int)lane >= d) incl += up; }
		const uint32_t total = (uint32_t)__shfl((int)incl, 63);
		const uint32_t z = (uint32_t)(4 * cz + p);
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd(&cnt[z * kCntStride], total);
		base = (uint32_t)__shfl((int)base, 0) + incl - n;
		uint32_t bits = lit[p];
		while (bits) {
			const int j = __ffs((int)bits) - 1;
			bits &= bits - 1u;
			list[z * XY + base++] = z * XY + (uint32_t)(4 * cy + (j >> 2)) * X + 4u * (uint32_t)cx + (uint32_t)(j & 3);
		}
	}
}

__global__ __launch_bounds__(256) void k_occupancy_dilate(int CX, int CY, int CZ, const float* __restrict__ blk, float* __restrict__ occ,
	unsigned long long* __restrict__ pos64, unsigned long long* __restrict__ vis64, uint32_t words64)
{
	const int c = blockIdx.x * 256 + threadIdx.x;
	const bool valid = c < CX * CY * CZ;
	float m = 0.0f;
	if (valid) {
		const int cx = c % CX, cy = (c / CX) % CY, cz = c / (CX * CY);
		const int x1 = min(cx + 1, CX - 1), y1 = min(cy + 1, CY - 1), z1 = min(cz + 1, CZ - 1);
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int xx = (k & 1) ? x1 : cx, yy = (k & 2) ? y1 : cy, zz = (k & 4) ? z1 : cz;
			m = fmaxf(m, blk[((size_t)zz * CY + yy) * CX + xx]);
		}
		occ[c] = m;
	}
	// the same comparisons the marches would make on occ[c] (fx_march.h): NaN counts as occupied
	const unsigned long long bp = __ballot(valid && !(m == 0.0f)), bv = __ballot(valid && !(m <= 0.00999999978f));
	if ((threadIdx.x & 63) == 0 && (uint32_t)(c >> 6) < words64) { pos64[c >> 6] = bp; vis64[c >> 6] = bv; }
}

// masks of a coarser level (grids whose 4^3 blocks exceed the LDS budget): a bit per (4 << sh)^3 block = OR over its fine blocks
__global__ __launch_bounds__(256) void k_mask_coarsen(int CX, int CY, int CZ, int sh, int MX, int MY, int MZ, const float* __restrict__ occ,
	unsigned long long* __restrict__ pos64, unsigned long long* __restrict__ vis64, uint32_t words64)
{
	const int c = blockIdx.x * 256 + threadIdx.x;
	const bool valid = c < MX * MY * MZ;
	bool p = false, v = false;
	if (valid) {
		const int mx = c % MX, my = (c / MX) % MY, mz = c / (MX * MY);
		const int n = 1 << sh;
		for (int dz = 0; dz < n; ++dz)
			for (int dy = 0; dy < n; ++dy)
				for (int dx = 0; dx < n; ++dx) {
					const int x = (mx << sh) + dx, y = (my << sh) + dy, z = (mz << sh) + dz;
					if (x >= CX || y >= CY || z >= CZ) continue;
					const float m = occ[((size_t)z * CY + y) * CX + x];
					p |= !(m == 0.0f);
					v |= !(m <= 0.00999999978f);
				}
	}
	const unsigned long long bp = __ballot(p), bv = __ballot(v);
	if ((threadIdx.x & 63) == 0 && (uint32_t)(c >> 6) < words64) { pos64[c >> 6] = bp; vis64[c >> 6] = bv; }
}

hipError_t launch_accel_build(const Geom& g, int half_store, const void* color, const RenderAccel& a, hipStream_t s, bool alpha_current, const LightFill* fill, bool* filled)
{
	const int n = a.CX * a.CY * a.CZ;
	float* blk = a.occ + n;
	const dim3 grid((g.X + 63) / 64, (a.CY + 3) / 4, a.CZ), block(256);
	uint32_t* cnt = ctr_now(a, g);
	uint32_t* nxt = ctr_next(a, g);
	auto pow2 = [](int v) { return v >= 4 && (v & (v - 1)) == 0; };
	const bool fused = fill && alpha_current && pow2(g.X) && pow2(g.Y) && pow2(g.Zg) && (size_t)((a.CX + 63) / 64) * a.CY * a.CZ * 4 <= (size_t)n && FX_KNOB_INT("LIGHT_FILL", 1);
	if (filled) *filled = fused;
	if (fused) {
		// (the per-cell "holds a lit voxel" words live in the cell list of the three-pass path, which this path does not use: two sets)
		const size_t words = (size_t)((a.CX + 63) / 64) * a.CY * a.CZ;
		unsigned long long* sets = reinterpret_cast<unsigned long long*>(a.cells);
		unsigned long long* now = sets + (a.fill_frame & 1u) * words;
		const unsigned long long* prev = fill->incremental && FX_KNOB_INT("LIGHT_FILL_DIRTY", 1) ? sets + ((a.fill_frame & 1u) ^ 1u) * words : nullptr;
		hipLaunchKernelGGL(k_build_fill, dim3((a.CX + 63) / 64, (a.CY + 3) / 4, a.CZ), block, 0, s, g, blk, a.alpha, cnt, nxt, a.list, fill->lightmap, *fill->fc, fill->has_sh, prev, now);
	}
	else if (alpha_current && (g.X & 3) == 0) hipLaunchKernelGGL(k_occupancy_blocks_a4, dim3((a.CX + 63) / 64, (a.CY + 3) / 4, a.CZ), block, 0, s, g, blk, a.alpha, cnt, nxt);
	else if (alpha_current) hipLaunchKernelGGL((k_occupancy_blocks<false, true>), grid, block, 0, s, g, (const float4*)color, blk, a.alpha, cnt, nxt);
	else if (half_store) hipLaunchKernelGGL((k_occupancy_blocks<true, false>), grid, block, 0, s, g, (const h16x4*)color, blk, a.alpha, cnt, nxt);
	else hipLaunchKernelGGL((k_occupancy_blocks<false, false>), grid, block, 0, s, g, (const float4*)color, blk, a.alpha, cnt, nxt);
	hipLaunchKernelGGL(k_occupancy_dilate, dim3((n + 255) / 256), dim3(256), 0, s, a.CX, a.CY, a.CZ, blk, a.occ,
		reinterpret_cast<unsigned long long*>(a.bits), reinterpret_cast<unsigned long long*>(a.bits + a.fine_words), a.fine_words / 2);
	if (a.msh) {
		const int m = a.MX * a.MY * a.MZ;
		hipLaunchKernelGGL(k_mask_coarsen, dim3((m + 255) / 256), dim3(256), 0, s, a.CX, a.CY, a.CZ, a.msh, a.MX, a.MY, a.MZ, a.occ,
			reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(mask_pos(a))), reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(mask_vis(a))), a.mask_words / 2);
	}
	return hipGetLastError();
}

// whole 16-byte groups of a mask into the LDS, by every thread of the workgroup; eight loads in flight per thread (32 KiB = one
// round trip for 256 threads)
__device__ __forceinline__ void fill_lds(uint32_t* dst, const uint32_t* __restrict__ src, uint32_t words)
{
	const uint4* s4 = reinterpret_cast<const uint4*>(src);
	uint4* d4 = reinterpret_cast<uint4*>(dst);
	const uint32_t n4 = words / 4, bd = blockDim.x;
	uint32_t base = threadIdx.x;
	for (; base + 7 * bd < n4; base += 8 * bd) {
		uint4 r[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) r[k] = s4[base + k * bd];
#pragma unroll
		for (int k = 0; k < 8; ++k) d4[base + k * bd] = r[k];
	}
	for (; base < n4; base += bd) d4[base] = s4[base];
}

// ---- light volume (CSRayMarchL.hlsl:15-80) ----------------------------------------------------------------------------------
// Every voxel takes its centre sample (:37); the empty ones (density < 0.01, :44) get the constant `light colour + ambient`
// (shadow = 1; with the light probe: ao * irradiance = 1 * 0), the others cast rays.  98 % are empty, most of them inside empty
// 4^3 cells: pass 1a looks at cells -- a voxel's centre sample has its base tap in the voxel's cell or (by rounding, on grids that
// are no power of two) in a lower neighbour, so where the pre-dilation maxima of the 27 cells around a cell are all +0 every voxel
// of the cell samples +0: sixteen 16-byte stores of the constant; the other cells go on a list (per cell layer, counters a cache
// line apart).  Pass 1b takes the centre samples of the listed cells, a wave per cell, writes the constant or appends the voxel to
// the list of its z plane: list[z * X * Y + k], k < cnt[z * kCntStride].
__global__ __launch_bounds__(256) void k_light_cells(const Geom g, int CX, int CY, int CZ, const float* __restrict__ blk, uint32_t* __restrict__ cells,
	uint32_t* __restrict__ cnt, uint32_t* __restrict__ lightmap, const FrameConsts fc, int has_sh)
{
	const int c = blockIdx.x * 256 + threadIdx.x;
	const bool valid = c < CX * CY * CZ;
	const int cx = c % CX, cy = (c / CX) % CY, cz = c / (CX * CY);
	bool occupied = false;
	if (valid) {
		float m = 0.0f;
		for (int dz = max(cz - 1, 0); dz <= min(cz + 1, CZ - 1); ++dz)
			for (int dy = max(cy - 1, 0); dy <= min(cy + 1, CY - 1); ++dy)
				for (int dx = max(cx - 1, 0); dx <= min(cx + 1, CX - 1); ++dx) {
					const float v = blk[((size_t)dz * CY + dy) * CX + dx];
					m = v == 0.0f ? m : 1.0f;                                      // (NaN counts as occupied)
				}
		occupied = m != 0.0f;
		if (!occupied) {
			const float irr[3] = { 0.0f, 0.0f, 0.0f };
			const uint32_t e = light_value(fc, has_sh != 0, 1.0f, 1.0f, irr);
			const bool whole_rows = 4 * cx + 4 <= g.X && (g.X & 3) == 0;
			for (int z = 4 * cz; z < min(4 * cz + 4, g.Zg); ++z)
				for (int y = 4 * cy; y < min(4 * cy + 4, g.Y); ++y) {
					uint32_t* row = lightmap + ((size_t)z * g.Y + y) * g.X + 4 * cx;
					if (whole_rows) *reinterpret_cast<uint4*>(row) = make_uint4(e, e, e, e);
					else for (int x = 0; x < min(4, g.X - 4 * cx); ++x) row[x] = e;
				}
		}
	}
	// the occupied cells of a wave go on the list of their layer (one atomic per wave and layer; a wave spans layers only on tiny grids)
	const unsigned long long b = __ballot(occupied);
	if (b) {
		const uint32_t lane = threadIdx.x & 63u;
		const int first = __ffsll((long long)b) - 1;
		const int cz0 = __shfl(cz, first);
		const unsigned long long same = __ballot(occupied && cz == cz0);
		uint32_t* cc = cnt + ctr_cells(g);
		if (same == b) {
			uint32_t base = 0;
			if ((int)lane == first) base = atomicAdd(&cc[cz0 * kCntStride], (uint32_t)__popcll(b));
			base = (uint32_t)__shfl((int)base, first);
			if (occupied) cells[(size_t)cz * CX * CY + base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)c;
		} else if (occupied) cells[(size_t)cz * CX * CY + atomicAdd(&cc[cz * kCntStride], 1u)] = (uint32_t)c;
	}
}

// exclusive scan of n counters (stride kCntStride) into pre[0 .. n], counts shifted right by `sh` and rounded up; by wave 0 of the workgroup
__device__ __forceinline__ void scan_counters(uint32_t* pre, const uint32_t* __restrict__ cnt, uint32_t n, int sh)
{
	const uint32_t lane = threadIdx.x & 63u;
	if ((threadIdx.x >> 6) == 0) {
		uint32_t carry = 0;
		for (uint32_t z0 = 0; z0 < n; z0 += 64) {
			const uint32_t z = z0 + lane;
			uint32_t v = z < n ? (cnt[z * kCntStride] + ((1u << sh) - 1u)) >> sh : 0u;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t up = (uint32_t)__shfl_up((int)v, d);
				if ((int)lane >= d) v += up;
			}
			if (z < n) pre[z + 1] = carry + v;
			carry += (uint32_t)__shfl((int)v, 63);
		}
		if (lane == 0) pre[0] = 0;
	}
	__syncthreads();
}

// the segment of item c: pre[lo] <= c < pre[lo + 1]
__device__ __forceinline__ uint32_t find_segment(const uint32_t* pre, uint32_t n, uint32_t c)
{
	uint32_t lo = 0, hi = n;
	while (hi - lo > 1) {
		const uint32_t mid = (lo + hi) >> 1;
		if (pre[mid] <= c) lo = mid; else hi = mid;
	}
	return lo;
}

__global__ __launch_bounds__(256) void k_light_classify(const Geom g, const float* __restrict__ alpha, const uint32_t* __restrict__ pos_fine, int CX, int CY, int CZ,
	const uint32_t* __restrict__ cells, uint32_t* __restrict__ list, uint32_t* __restrict__ cnt, uint32_t* __restrict__ lightmap, const FrameConsts fc, int has_sh)
{
	extern __shared__ uint32_t lds[];
	uint32_t* pre = lds;
	scan_counters(pre, cnt + ctr_cells(g), (uint32_t)CZ, 0);
	const uint32_t T = pre[CZ];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t XY = (uint32_t)g.X * (uint32_t)g.Y;
	const AccelVol<false, false> vol{ nullptr, alpha, nullptr, pos_fine, pos_fine, 0, CX, CY, CX, CY };
	for (uint32_t n = blockIdx.x * 4u + wave; n < T; n += gridDim.x * 4u) {
		const uint32_t layer = find_segment(pre, (uint32_t)CZ, n);
		const uint32_t c = cells[(size_t)layer * CX * CY + (n - pre[layer])];
		const int cx = (int)(c % (uint32_t)CX), cy = (int)((c / (uint32_t)CX) % (uint32_t)CY), cz = (int)(c / ((uint32_t)CX * (uint32_t)CY));
		const int x = 4 * cx + (int)(lane & 3u), y = 4 * cy + (int)((lane >> 2) & 3u), z = 4 * cz + (int)(lane >> 4);
		const bool valid = x < g.X && y < g.Y && z < g.Zg;
		const uint32_t id = (uint32_t)z * XY + (uint32_t)y * (uint32_t)g.X + (uint32_t)x;
		bool lit = false, exact = true;
		uint32_t centre = 0;
		float pu = 0.0f, pv = 0.0f, pw = 0.0f;
		if (valid) {
			const float ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);    // CSRayMarchL.hlsl:22
			const float oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
			const float oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
			// :36-37.  Where the voxel centre's texel coordinate comes out exact (extents that are powers of two) all filter weights are 0 and
			// the trilinear sample IS its base tap (fma(0, b - a, a) = a): one coalesced load per voxel instead of a mask word and 8 taps
			pu = fmaf(ox, 0.5f, 0.5f); pv = fmaf(oy, 0.5f, 0.5f); pw = fmaf(oz, 0.5f, 0.5f);
			const Base bc = make_base(g, pu, pv, pw);
			exact = bc.fx == 0.0f && bc.fy == 0.0f && bc.fz == 0.0f;
			centre = mad24(mad24((uint32_t)bc.z0, (uint32_t)g.Y, (uint32_t)bc.y0), (uint32_t)g.X, (uint32_t)bc.x0);
		}
		const bool all_exact = __ballot(valid && !exact) == 0;
		if (valid) {
			float density;
			if (all_exact) density = alpha[centre]; else density = density_at(vol, g, pu, pv, pw);
			lit = density >= 0.00999999978f;                                       // :44
			if (!lit) {
				const float irr[3] = { 0.0f, 0.0f, 0.0f };
				lightmap[id] = light_value(fc, has_sh != 0, 1.0f, 1.0f, irr);
			}
		}
		// lanes 16 q .. 16 q + 15 = plane 4 cz + q: one atomic per plane of the cell that holds a lit voxel
		const unsigned long long b = __ballot(lit);
		const unsigned long long mine = (b >> (lane & 48u)) & 0xFFFFull;
		if (mine) {
			const int first = __ffsll((long long)mine) - 1 + (int)(lane & 48u);
			uint32_t base = 0;
			if ((int)lane == first) base = atomicAdd(&cnt[z * kCntStride], (uint32_t)__popcll(mine));
			base = (uint32_t)__shfl((int)base, first);
			if (lit) list[(uint32_t)z * XY + base + (uint32_t)__popcll(mine & ((1ull << (lane & 15u)) - 1ull))] = id;
		}
	}
}

// pass 2: the shadow ray (:55) and the GI term (:59-68) of the listed voxels.  Every workgroup scans the plane counters into chunk
// offsets (LDS), then wave w of the launch takes chunks w, w + W, ...: neighbours in the list -- rays of similar length -- go to
// different waves, and nothing is dealt out through memory.
template <bool COARSE>
__global__ __launch_bounds__(256) void k_light_march(const Geom g, const float* __restrict__ alpha, const float* __restrict__ occ,
	const uint32_t* __restrict__ pos_mask, uint32_t mask_words, int msh, int MX, int MY, int CX, int CY,
	const uint32_t* __restrict__ list, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ lightmap, const FrameConsts fc,
	const float* __restrict__ sh, uint32_t numSamples, unsigned long long* __restrict__ counters)
{
	extern __shared__ uint32_t lds[];
	uint32_t* pre = lds + mask_words;                                              // pre[z] = chunks of the planes below z; pre[Zg] = all
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t Z = (uint32_t)g.Zg;
	scan_counters(pre, cnt, Z, 6);
	const uint32_t T = pre[Z];
	if (blockIdx.x * 4u >= T) return;                                              // (uniform) more waves than chunks
	fill_lds(lds, pos_mask, mask_words);
	__syncthreads();
	const AccelVol<false, COARSE> vol{ nullptr, alpha, occ, lds, lds, msh, MX, MY, CX, CY };
	const float stepScale = 3.46410155f / (float)numSamples;                       // RayMarch.hlsli:29-30
	float lx, ly, lz;
	light_dir_local(fc, lx, ly, lz);
	const uint32_t XY = (uint32_t)g.X * (uint32_t)g.Y;
	for (uint32_t c = blockIdx.x * 4u + wave; c < T; c += gridDim.x * 4u) {
		const uint32_t lo = find_segment(pre, Z, c);                               // the plane of chunk c
		const uint32_t k = (c - pre[lo]) * 64u + lane;
		const bool has = k < cnt[lo * kCntStride];
		uint32_t ns = 0;
		if (has) {
			const uint32_t id = list[lo * XY + k];
			const int z = (int)lo, y = (int)((id - lo * XY) / (uint32_t)g.X), x = (int)(id - lo * XY - (uint32_t)y * (uint32_t)g.X);
			const float ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);    // CSRayMarchL.hlsl:22
			const float oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
			const float oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
			float shadow = 1.0f, ao = 1.0f, irr[3] = { 0.0f, 0.0f, 0.0f };
			cast_light_ray<kLightAhead>(shadow, g, vol, ox, oy, oz, lx, ly, lz, stepScale, numSamples, ns);   // :55
			if (sh) gi_term<kLightAhead>(irr, ao, g, vol, fc, sh, ox, oy, oz, fmaf(ox, 0.5f, 0.5f), fmaf(oy, 0.5f, 0.5f), fmaf(oz, 0.5f, 0.5f), stepScale, numSamples, ns);   // :59-68
			lightmap[id] = light_value(fc, sh != nullptr, shadow, ao, irr);
		}
		flush_counts(counters, 0u, ns, 0u);
	}
}

// pass 2 without the light probe: shadow rays only (:55) -- and a different way to fill the machine.  The lit voxels are few
// (340 k of 16.8 M at frame 132 of the 256^3 run: 5 300 waves) and their rays take 1 .. 64 samples, so a wave that marches 64 rays
// to the end runs as long as its longest ray with a third of its lanes busy (54 M instructions for 6.8 M samples).  Here a wave owns
// every W-th block of 64 entries of the (flattened) lists and keeps refilling: whenever a quarter of its lanes have finished their
// rays, those lanes write their results and take the next entries.  Every iteration each live lane takes exactly one sample.
// With the light probe (:59-68) the voxel casts a second ray, along its density gradient: three launches -- the shadow rays
// (RAYS_SHADOW_KEEP: the transmittance is parked in the voxel's light-map word), k_light_gi_dirs (GetDensityGradient for every
// listed voxel, full waves), the occlusion rays (RAYS_AO: direction per lane; the finished lane evaluates the irradiance, picks the
// shadow up again and writes the light-map value).
enum { RAYS_SHADOW = 0, RAYS_SHADOW_KEEP = 1, RAYS_AO = 2 };

template <bool COARSE, int MODE>
__global__ __launch_bounds__(1024) void k_light_rays(const Geom g, const float* __restrict__ alpha, const float* __restrict__ occ,
	const uint32_t* __restrict__ pos_mask, uint32_t mask_words, int msh, int MX, int MY, int CX, int CY,
	const uint32_t* __restrict__ list, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ lightmap, const FrameConsts fc,
	const float* __restrict__ gi, const float* __restrict__ sh, uint32_t numSamples, unsigned long long* __restrict__ counters)
{
	extern __shared__ uint32_t lds[];
	uint32_t* pre = lds + mask_words;                                              // pre[z] = lit voxels of the planes below z; pre[Zg] = all
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t Z = (uint32_t)g.Zg;
	scan_counters(pre, cnt, Z, 0);
	const uint32_t N = pre[Z];
	const uint32_t wpg = blockDim.x >> 6;                                           // waves per workgroup (they share one copy of the mask)
	if (blockIdx.x * blockDim.x >= N) return;                                      // (uniform) more waves than blocks
	fill_lds(lds, pos_mask, mask_words);
	__syncthreads();
	const AccelVol<false, COARSE> vol{ nullptr, alpha, occ, lds, lds, msh, MX, MY, CX, CY };
	const float stepScale = 3.46410155f / (float)numSamples;                       // RayMarch.hlsli:29-30
	float lx, ly, lz;                                                              // the ray's direction: the light's, or (RAYS_AO) the lane's own
	light_dir_local(fc, lx, ly, lz);
	float rx = 0.0f, ry = 0.0f, rz = 0.0f;                                         // RAYS_AO: the direction before it was normalised
	const uint32_t XY = (uint32_t)g.X * (uint32_t)g.Y;
	const uint32_t W = gridDim.x * wpg;
	uint32_t cur = blockIdx.x * wpg + (threadIdx.x >> 6);                          // the block being handed out, entries [64 cur + pos, 64 cur + pos + avail)
	uint32_t pos = 0, avail = 64u * cur < N ? min(64u, N - 64u * cur) : 0u;
	bool live = false, pending = false;
	uint32_t id = 0, i = 0, ns = 0;
	float ox = 0.0f, oy = 0.0f, oz = 0.0f, t = 0.0f, prev = 0.0f, transm = 1.0f;
	// what a finished ray leaves in the voxel's light-map word (:72-79)
	auto finish = [&](uint32_t vid, float tr, float ax, float ay, float az) -> uint32_t {
		if (MODE == RAYS_SHADOW_KEEP) return __float_as_uint(tr);
		float irr[3] = { 0.0f, 0.0f, 0.0f };
		if (MODE == RAYS_SHADOW) return light_value(fc, false, tr, 1.0f, irr);
		float wx = dot3(ax, ay, az, fc.world[0], fc.world[1], fc.world[2]);       // RayMarch.hlsli:280 / CSRayMarchL.hlsl:65
		float wy = dot3(ax, ay, az, fc.world[4], fc.world[5], fc.world[6]);
		float wz = dot3(ax, ay, az, fc.world[8], fc.world[9], fc.world[10]);
		const float rw = rsqf(dot3(wx, wy, wz, wx, wy, wz));
		wx *= rw; wy *= rw; wz *= rw;
		sh_irradiance(irr, sh, wx, wy, wz);
		return light_value(fc, true, __uint_as_float(lightmap[vid]), tr, irr);
	};
	for (;;) {
		const unsigned long long idle = __ballot(!live);
		const uint32_t nidle = (uint32_t)__popcll(idle);
		if (nidle == 64u && avail == 0u) break;
		if (avail != 0u && nidle >= 16u) {
			if (pending) {                                                         // (pending implies !live)
				lightmap[id] = finish(id, transm, rx, ry, rz);
				pending = false;
			}
			const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
			if (!live && rank < avail) {
				const uint32_t flat = 64u * cur + pos + rank;
				const uint32_t z = find_segment(pre, Z, flat);
				id = list[z * XY + (flat - pre[z])];
				const uint32_t y = (id - z * XY) / (uint32_t)g.X, x = id - z * XY - y * (uint32_t)g.X;
				ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);            // CSRayMarchL.hlsl:22
				oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
				oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
				t = stepScale; prev = 0.0f; transm = 1.0f; i = 0;                  // CastLightRay, RayMarch.hlsli:215-221
				if (MODE == RAYS_AO) {
					rx = gi[3 * (size_t)flat]; ry = gi[3 * (size_t)flat + 1]; rz = gi[3 * (size_t)flat + 2];
					const float rd = rsqf(dot3(rx, ry, rz, rx, ry, rz));           // CSRayMarchL.hlsl:66
					lx = rx * rd; ly = ry * rd; lz = rz * rd;
				}
				live = true;
			}
			const uint32_t n = min(nidle, avail);
			pos += n; avail -= n;
			if (avail == 0u) { cur += W; pos = 0; avail = 64u * cur < N ? min(64u, N - 64u * cur) : 0u; }
		}
#pragma unroll
		for (int rep = 0; rep < FX_LIGHT_RAY_UNROLL; ++rep)
		if (live) {                                                                // one sample of the loop :222-246
			const float px = fmaf(lx, t, ox), py = fmaf(ly, t, oy), pz = fmaf(lz, t, oz);
			bool on = i < numSamples && !outside(px, py, pz);
			if (on) {
				++ns;
				const Base b = make_base(g, fmaf(px, 0.5f, 0.5f), fmaf(py, 0.5f, 0.5f), fmaf(pz, 0.5f, 0.5f));
				float density = 0.0f;
				if (vol.dense(b)) density = vol.density(make_taps(g, b));
				float fac;
				on = light_step(density, stepScale, t, prev, transm, i, fac);
			}
			if (!on) { live = false; pending = true; }
		}
	}
	if (pending) lightmap[id] = finish(id, transm, rx, ry, rz);
	flush_counts(counters, 0u, ns, 0u);
}

// GetDensityGradient (RayMarch.hlsli:73-95) of every listed voxel: the occlusion ray's direction before normalisation -- minus the
// gradient, or the voxel's position where the gradient vanishes (CSRayMarchL.hlsl:61-64) -- to gi[flat index][3]
__global__ __launch_bounds__(256) void k_light_gi_dirs(const Geom g, const float* __restrict__ alpha, int CX, int CY,
	const uint32_t* __restrict__ list, const uint32_t* __restrict__ cnt, float* __restrict__ gi, unsigned long long* __restrict__ counters)
{
	extern __shared__ uint32_t lds[];
	uint32_t* pre = lds;
	const uint32_t Z = (uint32_t)g.Zg;
	scan_counters(pre, cnt, Z, 0);
	const uint32_t N = pre[Z];
	const uint32_t XY = (uint32_t)g.X * (uint32_t)g.Y;
	const AccelVol<false, false> vol{ nullptr, alpha, nullptr, nullptr, nullptr, 0, CX, CY, CX, CY };
	uint32_t ns = 0;
	for (uint32_t flat = blockIdx.x * 256u + threadIdx.x; flat < N; flat += gridDim.x * 256u) {
		const uint32_t z = find_segment(pre, Z, flat);
		const uint32_t id = list[z * XY + (flat - pre[z])];
		const uint32_t y = (id - z * XY) / (uint32_t)g.X, x = id - z * XY - y * (uint32_t)g.X;
		const float ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);        // CSRayMarchL.hlsl:22
		const float oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
		const float oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
		const float u = fmaf(ox, 0.5f, 0.5f), v = fmaf(oy, 0.5f, 0.5f), w = fmaf(oz, 0.5f, 0.5f);
		ns += 6;
		float qxm, qxp, qym, qyp, qzm, qzp;
		const Base c = make_base(g, u, v, w);
		if (__builtin_amdgcn_ballot_w64(c.fx != 0.0f || c.fy != 0.0f || c.fz != 0.0f) == 0) {
			// The six samples sit at the voxel's own centre, shifted by whole texels: where the centre's texel coordinate comes out exact
			// (extents that are powers of two: (x + 0.5) / X * X - 0.5 = x) all three filter weights are 0 and every lerp returns its first
			// operand, fma(0, b - a, a) = a for the finite alphas of a saturated field -- the sample IS its base tap: 6 loads, not 48.
			const uint32_t X = (uint32_t)g.X, XY = X * (uint32_t)g.Y;
			auto tap = [&](int ox_, int oy_, int oz_) -> float {
				const int x0 = min(max(c.ix + ox_, 0), g.X - 1), y0 = min(max(c.iy + oy_, 0), g.Y - 1), z0 = min(max(c.iz + oz_, 0), g.Zg - 1);
				return alpha[(uint32_t)z0 * XY + (uint32_t)y0 * X + (uint32_t)x0];
			};
			qxm = tap(-1, 0, 0); qxp = tap(1, 0, 0); qym = tap(0, -1, 0); qyp = tap(0, 1, 0); qzm = tap(0, 0, -1); qzp = tap(0, 0, 1);
		} else {
			// (the voxel is lit: its neighbourhood is smoke, so all 48 taps go out at once instead of asking the masks first -- same values)
			qxm = vol.density(make_taps(g, make_base(g, u, v, w, -1, 0, 0))); qxp = vol.density(make_taps(g, make_base(g, u, v, w, 1, 0, 0)));
			qym = vol.density(make_taps(g, make_base(g, u, v, w, 0, -1, 0))); qyp = vol.density(make_taps(g, make_base(g, u, v, w, 0, 1, 0)));
			qzm = vol.density(make_taps(g, make_base(g, u, v, w, 0, 0, -1))); qzp = vol.density(make_taps(g, make_base(g, u, v, w, 0, 0, 1)));
		}
		const float gx = -qxm + qxp, gy = -qym + qyp, gz = -qzm + qzp;
		const bool any = fabsf(gx) > 0.0f || fabsf(gy) > 0.0f || fabsf(gz) > 0.0f;
		gi[3 * (size_t)flat] = any ? -gx : ox; gi[3 * (size_t)flat + 1] = any ? -gy : oy; gi[3 * (size_t)flat + 2] = any ? -gz : oz;
	}
	flush_counts(counters, 0u, ns, 0u);
}

hipError_t launch_accel_light(const Geom& g, const RenderAccel& a, uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	uint32_t num_samples, hipStream_t s, unsigned long long* counters, bool filled)
{
	const int ncell = a.CX * a.CY * a.CZ;
	uint32_t* ctr = ctr_now(a, g);
	if (!filled) {                                                                 // (else k_build_fill has written the constants and the lists)
	hipLaunchKernelGGL(k_light_cells, dim3((ncell + 255) / 256), dim3(256), 0, s, g, a.CX, a.CY, a.CZ, a.occ + ncell, a.cells, ctr, lightmap, fc, sh ? 1 : 0);
	// (2048 persistent workgroups: 30 us at 256^3 / frame 132; 8192 -- a wave per listed cell -- 37 us.  Of today's 25.6 us: 4 the
	// prologue, 12.5 list entry + alpha + arithmetic, 2.4 the stores, 6.6 the list atomics -- measured by leaving each out; two cells in
	// flight per wave changed nothing: more than half of the 262 k cells of frame 132 are listed, the pass moves ~75 MB)
	hipLaunchKernelGGL(k_light_classify, dim3((unsigned)std::min(ncell / 4 + 1, 2048)), dim3(256), ((size_t)a.CZ + 1) * 4, s, g, a.alpha, a.bits, a.CX, a.CY, a.CZ,
		a.cells, a.list, ctr, lightmap, fc, sh ? 1 : 0);
	}
	const size_t cells = (size_t)g.X * g.Y * g.Zg;
	const size_t lds = (size_t)a.mask_words * 4 + ((size_t)g.Zg + 1) * 4;
	// (workgroups of 8 waves around one copy of the mask: 8 instead of 5 waves per SIMD fit beside it.  256^3 frame 132, light pass by
	// threads x workgroups: 256 x 2048 0.118, 512 x 1024 0.113, 1024 x 512 0.118 ms)
	const int ray_nt = FX_KNOB_INT("LIGHT_RAY_NT", 512);
	const unsigned wgs = (unsigned)std::min<size_t>((cells + ray_nt - 1) / ray_nt, (size_t)FX_KNOB_INT("LIGHT_RAY_WGS", (int)kLightRayWorkgroups));
#define FX_RAYS(C, M) hipLaunchKernelGGL((k_light_rays<C, M>), dim3(wgs), dim3(ray_nt), lds, s, g, a.alpha, a.occ, mask_pos(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY, \
	a.list, ctr, lightmap, fc, a.gi, sh, num_samples, counters)
	if (!sh) {
		if (a.msh) FX_RAYS(true, RAYS_SHADOW); else FX_RAYS(false, RAYS_SHADOW);
		return hipGetLastError();
	}
	if (a.gi) {
		if (a.msh) FX_RAYS(true, RAYS_SHADOW_KEEP); else FX_RAYS(false, RAYS_SHADOW_KEEP);
		hipLaunchKernelGGL(k_light_gi_dirs, dim3((unsigned)std::min<size_t>((cells + 255) / 256, 2048)), dim3(256), ((size_t)g.Zg + 1) * 4, s, g, a.alpha, a.CX, a.CY,
			a.list, ctr, a.gi, counters);
		if (a.msh) FX_RAYS(true, RAYS_AO); else FX_RAYS(false, RAYS_AO);
		return hipGetLastError();
	}
#undef FX_RAYS
	// (no scratch for the occlusion rays' directions: the chunked march, a wave per 64 listed voxels)
	if (a.msh) hipLaunchKernelGGL(k_light_march<true>, dim3(wgs), dim3(256), lds, s, g, a.alpha, a.occ, mask_pos(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY,
		a.list, ctr, lightmap, fc, sh, num_samples, counters);
	else hipLaunchKernelGGL(k_light_march<false>, dim3(wgs), dim3(256), lds, s, g, a.alpha, a.occ, mask_pos(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY,
		a.list, ctr, lightmap, fc, sh, num_samples, counters);
	return hipGetLastError();
}



// ---- view marches ---------------------------------------------------------------------------------------------------------
// A workgroup = TX x TY tiles of 8 x 8 rays (one tile per wave, so a wave's taps stay spatially coherent) sharing one copy of the
// masks.  Lanes without a ray keep running to the barriers and sit the march out.
struct MaskArgs { const uint32_t* pos; const uint32_t* vis; uint32_t words; int msh, MX, MY, CX, CY; };

template <bool HALF, bool SEPARATE, bool COARSE>
__device__ __forceinline__ AccelVol<HALF, COARSE> view_volume(uint32_t* lds, const typename ColTex<HALF>::T* col, const float* alpha, const float* occ,
	const MaskArgs& m, bool any)
{
	if (any) {                                                                     // (uniform over the workgroup)
		fill_lds(lds, m.vis, m.words);
		if (!SEPARATE) fill_lds(lds + m.words, m.pos, m.words);                    // the nested light rays of the merged march
	}
	__syncthreads();
	return AccelVol<HALF, COARSE>{ col, alpha, occ, SEPARATE ? lds : lds + m.words, lds, m.msh, m.MX, m.MY, m.CX, m.CY };
}

template <bool HALF, bool SEPARATE, bool COARSE>
__global__ __launch_bounds__(256) void k_view_march(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, const float* __restrict__ alpha,
	const float* __restrict__ occ, const MaskArgs m, const uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh,
	int size, uint32_t mask, uint32_t numSamples, uint32_t numLightSamples, uint32_t* __restrict__ cube, unsigned long long* __restrict__ counters)
{
	extern __shared__ uint32_t lds[];
	const int face = blockIdx.z;
	if (!((mask >> face) & 1u)) return;                                            // CSRayMarch.hlsl:102 (uniform)
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7), y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
	float o[3] = { 0.0f, 0.0f, 0.0f }, d[3] = { 0.0f, 0.0f, 1.0f }, tMax = 0.0f;
	const bool go = x < size && y < size && cube_texel_ray(fc, face, x, y, size, o, d, tMax);   // :116
	const AccelVol<HALF, COARSE> vol = view_volume<HALF, SEPARATE, COARSE>(lds, col, alpha, occ, m, __syncthreads_or(go) != 0);
	float sr, sg, sb, sa;
	uint32_t nv = 0, nl = 0, nm = 0;
	march_ray<AccelVol<HALF, COARSE>, SEPARATE, SEPARATE ? kViewAhead : 1>(g, vol, lightmap, fc, sh, o, d, tMax, numSamples, numLightSamples, go, sr, sg, sb, sa, nv, nl, nm);
	flush_counts(counters, nv, nl, nm);
	if (!go) return;
	sr *= 0.159154937f; sg *= 0.159154937f; sb *= 0.159154937f;                    // :192
	cube[((size_t)face * size + y) * size + x] =
		to_unorm8(sr) | (to_unorm8(sg) << 8) | (to_unorm8(sb) << 16) | (to_unorm8(sa) << 24);   // :195
}

// direct screen-space march (row f-2; PSRayCast.hlsl:44-127 / PSRayCastV.hlsl): 16 x 16 pixels per workgroup
template <bool HALF, bool SEPARATE, bool COARSE>
__global__ __launch_bounds__(256) void k_direct_march(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, const float* __restrict__ alpha,
	const float* __restrict__ occ, const MaskArgs m, const uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh,
	int W, int H, uint32_t numSamples, uint32_t numLightSamples, uint32_t* __restrict__ target, float4* __restrict__ out_float,
	unsigned long long* __restrict__ counters)
{
	extern __shared__ uint32_t lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7), py = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
	const bool in = px < W && py < H;
	const size_t pix = (size_t)py * W + px;
	if (in && out_float) out_float[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float o[3] = { 0.0f, 0.0f, 0.0f }, d[3] = { 0.0f, 0.0f, 1.0f };
	const bool go = in && pixel_ray(fc, px, py, W, H, o, d);                       // PSRayCast.hlsl:50 discard
	const AccelVol<HALF, COARSE> vol = view_volume<HALF, SEPARATE, COARSE>(lds, col, alpha, occ, m, __syncthreads_or(go) != 0);
	float sr, sg, sb, sa;
	uint32_t nv = 0, nl = 0, nm = 0;
	march_ray<AccelVol<HALF, COARSE>, SEPARATE, SEPARATE ? kViewAhead : 1>(g, vol, lightmap, fc, sh, o, d, 3.40282347e+38f, numSamples, numLightSamples, go, sr, sg, sb, sa, nv, nl, nm);
	flush_counts(counters, nv, nl, nm);
	if (!go) return;
	sr *= 0.159154937f; sg *= 0.159154937f; sb *= 0.159154937f;                    // :124
	if (out_float) out_float[pix] = make_float4(sr, sg, sb, sa);
	if (target) target[pix] = blend_premultiplied(target[pix], sr, sg, sb, sa);
}

// ---- separate-pass view marches, eight samples of a ray per step of the wave ------------------------------------------------
// The march of a ray is a chain: sample k + 1 sits where GetStep of sample k says.  One lane per ray therefore walks ~500
// instructions and a memory round trip per gathered sample, one after the other, and the launch lasts as long as its longest ray
// (256^3, frame 132: 210 us for a wave with ~190 gathered samples per lane, while the average wave needs 48).  But the expensive
// part of a sample -- sixteen taps, two trilinear blends, the R11G11B10 decode -- depends on its POSITION only, and GetStep
// returns the plain step wherever the smoke is thin or varied, which is where rays get long.  So a ray gets eight lanes (a wave =
// a 4 x 2 block of cube-map texels): lane s evaluates the sample the ray reaches after s plain steps, all lanes of the ray then replay the
// reference's sequential loop over the eight results (exchanged through the LDS) and stop at the first sample whose predecessor
// took another step than the plain one; the rest is dropped and the next round starts from the true position.  Arithmetic and
// order per sample are the reference's, so the pictures stay bit-identical; the chain shrinks from one round per sample to one per
// up to eight.  Workgroups are persistent (the masks are copied into the LDS once), ray groups are dealt out round-robin.
static const int kSlots = 8;
static const size_t kViewWorkgroups = 1024;                    // persistent: 256 CUs x 4 (LDS: masks 32 KiB + exchange 5 KiB each)
static const int kXchgFloats = 5 * kSlots * (64 / kSlots);       // per wave: [ray][kind, alpha, r, g, b][slot]

template <bool HALF, bool COARSE>
#ifndef FX_VIEW_NT
#define FX_VIEW_NT 256
#endif
#ifndef FX_VIEW_WPE
#define FX_VIEW_WPE 0
#endif
#if FX_VIEW_WPE
#define FX_VIEW_ATTR __attribute__((amdgpu_waves_per_eu(FX_VIEW_WPE, FX_VIEW_WPE)))
#else
#define FX_VIEW_ATTR
#endif
__global__ __launch_bounds__(FX_VIEW_NT) FX_VIEW_ATTR void k_view_slots(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, const float* __restrict__ alpha,
	const float* __restrict__ occ, const MaskArgs m, const uint32_t* __restrict__ lightmap, const FrameConsts fc,
	int size, uint32_t mask, uint32_t numSamples, uint32_t* __restrict__ cube, uint32_t* __restrict__ heads,
	unsigned long long* __restrict__ counters, int order)
{
	extern __shared__ uint32_t lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 3, s = lane & 7;
	// ray groups: 4 x 2 blocks of cube texels of the visible faces
	const uint32_t gx = (uint32_t)(size + 3) >> 2, gy = (uint32_t)(size + 1) >> 1, gpf = gx * gy;
	const uint32_t total = gpf * (uint32_t)__popc(mask & 63u);
	if (blockIdx.x >= total) return;                                               // (uniform; a surplus workgroup would only find its head empty)
	fill_lds(lds, m.vis, m.words);
	__syncthreads();
	const AccelVol<HALF, COARSE> vol{ col, alpha, occ, lds, lds, m.msh, m.MX, m.MY, m.CX, m.CY };
	float* xch = reinterpret_cast<float*>(lds + m.words) + wave * kXchgFloats + r * (5 * kSlots);
	const float stepScale = 3.46410155f / (float)numSamples;
	uint32_t nv = 0, nm = 0;

	// ray groups are dealt out dynamically (the rays through the plume cost a hundred times the rays beside it, and they sit together):
	// eight heads, one per residue of the group index, each pulled by the workgroups of one residue of blockIdx -- on this part that is
	// one XCD per head (for speed only: any placement is correct), ~50 atomics per us and head instead of 400 on one word.  A wave
	// asks for its next group before it marches the current one.
	// (`run` groups per ticket on very large cube maps keep that rate)
	uint32_t* head = heads + (blockIdx.x & 7u) * 32u;
	const uint32_t run = total > 65536u ? 8u : 1u;
	uint32_t next = 0, cur = 0;
	if (lane == 0) next = atomicAdd(head, 1u);
	for (uint32_t sub = run;; ++sub) {
		if (sub == run) {
			sub = 0;
			cur = (uint32_t)__shfl((int)next, 0);
			if (((blockIdx.x & 7u) + 8u * cur) * run >= total) break;
			if (lane == 0) next = atomicAdd(head, 1u);
		}
		const uint32_t grp = ((blockIdx.x & 7u) + 8u * cur) * run + sub;
		if (grp >= total) break;
		// Which group a ticket stands for.  The launch ends with its last wave, and a group of rays through the plume costs a hundred times
		// a group beside it: the plume rises around the volume's vertical axis (the source sits at (0.5, 0.1, 0.5), CSAdvect.hlsl:11-14), i.e.
		// behind the middle columns of a cube face, so the columns are dealt out from the middle outwards -- the expensive groups start
		// first and the cheap ones fill the tail (view pass 0.117 -> 0.102 ms at frame 132; any order is correct).  order 0: as numbered.
		uint32_t g2 = grp;
		if (order) {
			const uint32_t per_col = total / gx, cr = grp / per_col, rest = grp - cr * per_col;     // rest = face * gy + row
			const uint32_t c0 = gx >> 1, col = (cr & 1u) ? c0 - ((cr + 1u) >> 1) : c0 + (cr >> 1);
			const uint32_t fj = rest / gy, row = rest - fj * gy;
			g2 = fj * gpf + row * gx + col;
		}
		uint32_t j = g2 / gpf, mm = mask & 63u;
		const uint32_t rem = g2 - j * gpf;
		for (; j; --j) mm &= mm - 1u;                                              // the j-th visible face (CSRayMarch.hlsl:102)
		const int face = __ffs((int)mm) - 1;
		const int x = (int)(rem % gx) * 4 + (r & 3), y = (int)(rem / gx) * 2 + (r >> 2);
		const size_t pix = ((size_t)face * size + y) * size + x;
		float o[3] = { 0.0f, 0.0f, 0.0f }, d[3] = { 0.0f, 0.0f, 1.0f }, tMax = 3.40282347e+38f;
		const bool go = x < size && y < size && cube_texel_ray(fc, face, x, y, size, o, d, tMax);   // CSRayMarch.hlsl:116

		float sr = 0.0f, sg = 0.0f, sb = 0.0f, sa = 0.0f, t = 0.0f, prev = 0.0f;
		uint32_t i = 0;
		bool live = go;
		while (__any(live)) {
			// the sample this lane evaluates: s plain steps ahead (t advances by rounded additions, exactly like the loop's own)
			float ts = t;
#pragma unroll
			for (int j = 0; j < kSlots - 1; ++j) { const float nx = ts + stepScale; ts = j < s ? nx : ts; }
			const float qx = fmaf(d[0], ts, o[0]), qy = fmaf(d[1], ts, o[1]), qz = fmaf(d[2], ts, o[2]);
			// kind 0: the loop does not get here (sample count :146, target passed :189, cube left :149); 1: cannot be seen; 2: fetch
			int kind = 0;
			Base b;
			if (live && i + (uint32_t)s < numSamples && (s == 0 || !(tMax < ts)) && !outside(qx, qy, qz)) {
				b = make_base(g, fmaf(qx, 0.5f, 0.5f), fmaf(qy, 0.5f, 0.5f), fmaf(qz, 0.5f, 0.5f));
				kind = vol.visible(b) ? 2 : 1;
			}
			if (__any(kind == 2)) {
				float cw = 0.0f, mr = 0.0f, mg = 0.0f, mb = 0.0f;
				if (kind == 2) {
					const Taps tp = make_taps(g, b);
					float4 c8[8];
					uint32_t l8[8];
					vol.color_taps(tp, c8);                                        // :157
					vol.light_taps(lightmap, tp, l8);                                  // RayMarch.hlsli:253-258 (used only behind :161)
					const float4 c = blend8x4(c8, tp);
					cw = c.w;
					if (0.00999999978f < c.w) {                                    // :161
						const float3 l = blend_light(l8, tp);
						mr = l.x * c.x; mg = l.y * c.y; mb = l.z * c.z;            // :180 (light * rgb precedes the transmittance)
					}
				}
				xch[0 * kSlots + s] = __int_as_float(kind);
				xch[1 * kSlots + s] = cw;
				xch[2 * kSlots + s] = mr; xch[3 * kSlots + s] = mg; xch[4 * kSlots + s] = mb;
				__builtin_amdgcn_wave_barrier();
				float kk[kSlots], ca[kSlots], cr[kSlots], cg[kSlots], cb[kSlots];
#pragma unroll
				for (int k = 0; k < kSlots; ++k) { kk[k] = xch[k]; ca[k] = xch[kSlots + k]; cr[k] = xch[2 * kSlots + k]; cg[k] = xch[3 * kSlots + k]; cb[k] = xch[4 * kSlots + k]; }
				__builtin_amdgcn_wave_barrier();
				// the reference's loop over the eight samples, as far as the march really arrives at them
				bool act = live;
#pragma unroll
				for (int k = 0; k < kSlots; ++k) {
					if (act) {
						const int kd = __float_as_int(kk[k]);
						if (kd == 0) { live = false; act = false; }
						else {
							++nv;
							float newStep = stepScale;
							if (kd == 2 && 0.00999999978f < ca[k]) {               // :161
								++nm;
								const float transm = -sa + 1.0f;                   // :170
								newStep = step_factor(-prev + ca[k], transm, ca[k]) * stepScale;   // :172
								sr = fmaf(transm * cr[k], 0.800000012f, sr);       // :180-181
								sg = fmaf(transm * cg[k], 0.800000012f, sg);
								sb = fmaf(transm * cb[k], 0.800000012f, sb);
								sa = fmaf(0.800000012f * ca[k], transm, sa);
								if (transm < 0.00999999978f) live = false;         // :183
								prev = ca[k];
							}
							if (live) {
								++i;
								t = t + newStep;                                   // :187-188
								if (tMax < t) live = false;                        // :189
							}
							if (!live || newStep != stepScale) act = false;
						}
					}
				}
			} else {
				// nobody fetches: every sample of the round is decided (kind 0 / 1).  A ray takes its leading run of unseen samples at
				// once -- and where a ray's whole octet is unseen, the lanes look at the three octets behind it as well
				uint32_t ones = (uint32_t)(__ballot(kind == 1) >> (r * kSlots)) & 0xFFu, twos = 0u;
				float tsm[4] = { ts, ts, ts, ts };
				if (__any(live && ones == 0xFFu)) {
#pragma unroll
					for (int mo = 1; mo < 4; ++mo) {
						float tq = tsm[mo - 1];
#pragma unroll
						for (int j = 0; j < kSlots; ++j) tq = tq + stepScale;
						tsm[mo] = tq;
						const float ux = fmaf(d[0], tq, o[0]), uy = fmaf(d[1], tq, o[1]), uz = fmaf(d[2], tq, o[2]);
						int kd = 0;
						if (live && i + (uint32_t)(mo * kSlots + s) < numSamples && !(tMax < tq) && !outside(ux, uy, uz))
							kd = vol.visible(make_base(g, fmaf(ux, 0.5f, 0.5f), fmaf(uy, 0.5f, 0.5f), fmaf(uz, 0.5f, 0.5f))) ? 2 : 1;
						ones |= ((uint32_t)(__ballot(kd == 1) >> (r * kSlots)) & 0xFFu) << (mo * kSlots);
						twos |= ((uint32_t)(__ballot(kd == 2) >> (r * kSlots)) & 0xFFu) << (mo * kSlots);
					}
				}
				const int n1 = ones == 0xFFFFFFFFu ? 32 : __ffs((int)~ones) - 1;   // 0 .. 32 unseen samples, then sample n1 of kind 0 or 2
				const int src = n1 < 32 ? n1 : 31;
				float tn = tsm[0];
				tn = (src >> 3) == 1 ? tsm[1] : tn; tn = (src >> 3) == 2 ? tsm[2] : tn; tn = (src >> 3) == 3 ? tsm[3] : tn;
				tn = __shfl(tn, (lane & ~(kSlots - 1)) + (src & (kSlots - 1)));    // the ray parameter after n1 plain steps, as the lane there added it up
				if (live) {
					nv += (uint32_t)n1;
					i += (uint32_t)n1;
					t = tn;
					if (n1 == 32) { t = t + stepScale; if (tMax < t) live = false; }    // :187-189 behind the 32nd
					else if (!((twos >> n1) & 1u)) live = false;                   // sample n1 is of kind 0: the loop ends there
				}
			}
		}
		if (go && s == 0) {
			sr *= 0.159154937f; sg *= 0.159154937f; sb *= 0.159154937f;            // CSRayMarch.hlsl:192
			cube[pix] = to_unorm8(sr) | (to_unorm8(sg) << 8) | (to_unorm8(sb) << 16) | (to_unorm8(sa) << 24);   // :195
		}
	}
	if (s == 0) flush_counts(counters, nv, 0u, nm);
}

static MaskArgs mask_args(const RenderAccel& a) { return MaskArgs{ mask_pos(a), mask_vis(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY }; }

hipError_t launch_accel_view(const Geom& g, int half_store, const void* color, const uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	int cube_size, uint32_t mask, uint32_t num_samples, uint32_t num_light_samples, int separate, uint8_t* cube, const RenderAccel& a, hipStream_t s,
	unsigned long long* counters)
{
	const dim3 grid((cube_size + 15) / 16, (cube_size + 15) / 16, 6), block(256);
	const size_t lds = (size_t)a.mask_words * 4 * (separate ? 1 : 2);
	uint32_t* out = reinterpret_cast<uint32_t*>(cube);
	const MaskArgs m = mask_args(a);
	if (separate) {
		const size_t groups = (size_t)((cube_size + 3) / 4) * ((cube_size + 1) / 2) * 6;
		const int vw = FX_VIEW_NT / 64;
		const dim3 pgrid((unsigned)std::min<size_t>((groups + vw - 1) / vw, (size_t)FX_KNOB_INT("VIEW_WGS", (int)kViewWorkgroups)));
		const size_t plds = (size_t)a.mask_words * 4 + vw * kXchgFloats * sizeof(float);
		// middle-out (see the kernel) where a wave gets through many groups; with two or three groups per wave (cube maps of 128 / 150 texels)
		// the order as numbered measured better: 128^3 0.094 against 0.116 ms, 150^3 0.102 / 0.111; 256-texel cubes 0.116 / 0.102
		const size_t visible_groups = groups / 6 * (size_t)__builtin_popcount(mask & 63u);
		const int view_order = FX_KNOB_INT("VIEW_ORDER", visible_groups >= 6 * (size_t)pgrid.x * vw ? 1 : 0);
#define FX_SLOTS(H, C) hipLaunchKernelGGL((k_view_slots<H, C>), pgrid, dim3(FX_VIEW_NT), plds, s, g, (const typename ColTex<H>::T*)color, a.alpha, a.occ, m, \
	lightmap, fc, cube_size, mask, num_samples, out, ctr_now(a, g) + ctr_heads(g), counters, view_order)
		if (half_store) { if (a.msh) FX_SLOTS(true, true); else FX_SLOTS(true, false); }
		else { if (a.msh) FX_SLOTS(false, true); else FX_SLOTS(false, false); }
#undef FX_SLOTS
		return hipGetLastError();
	}
#define FX_LAUNCH(H, S, C) hipLaunchKernelGGL((k_view_march<H, S, C>), grid, block, lds, s, g, (const typename ColTex<H>::T*)color, a.alpha, a.occ, m, \
	lightmap, fc, sh, cube_size, mask, num_samples, num_light_samples, out, counters)
#define FX_PICK(H, S) do { if (a.msh) FX_LAUNCH(H, S, true); else FX_LAUNCH(H, S, false); } while (0)
	if (half_store) FX_PICK(true, false); else FX_PICK(false, false);              // the merged march: its samples cast rays of their own
#undef FX_LAUNCH
	return hipGetLastError();
}

hipError_t launch_accel_direct(const Geom& g, int half_store, const void* color, const uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	int W, int H, uint32_t num_samples, uint32_t num_light_samples, int separate, uint8_t* target, float* out_float, const RenderAccel& a, hipStream_t s,
	unsigned long long* counters)
{
	const dim3 grid((W + 15) / 16, (H + 15) / 16, 1), block(256);
	const size_t lds = (size_t)a.mask_words * 4 * (separate ? 1 : 2);
	const MaskArgs m = mask_args(a);
#define FX_LAUNCH(HF, S, C) hipLaunchKernelGGL((k_direct_march<HF, S, C>), grid, block, lds, s, g, (const typename ColTex<HF>::T*)color, a.alpha, a.occ, m, \
	lightmap, fc, sh, W, H, num_samples, num_light_samples, reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float), counters)
	// one lane per pixel for both variants: two million rays, most of them beside the volume, keep every SIMD busy without the
	// eight-lanes-per-ray scheme (measured at 1920x1080 / 256^3: 0.33 ms either way, 0.47 ms with it)
	if (half_store) { if (separate) FX_PICK(true, true); else FX_PICK(true, false); }
	else { if (separate) FX_PICK(false, true); else FX_PICK(false, false); }
#undef FX_LAUNCH
#undef FX_PICK
	return hipGetLastError();
}

}  // namespace fx
