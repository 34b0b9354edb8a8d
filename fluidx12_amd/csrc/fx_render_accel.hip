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
template <bool HALF>
__global__ __launch_bounds__(256) void k_occupancy_blocks(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, float* __restrict__ blk,
	float* __restrict__ alpha, uint32_t* __restrict__ cnt)
{
	const int CX = (g.X + 3) >> 2, CY = (g.Y + 3) >> 2;
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 4 && 4 * (int)blockIdx.z + (int)threadIdx.x < g.Zg)
		cnt[(4 * blockIdx.z + threadIdx.x) * 32] = 0u;                             // the light-voxel lists of this frame start empty (kCntStride)
	const int x = blockIdx.x * 64 + (threadIdx.x & 63);
	const int cy = blockIdx.y * 4 + (threadIdx.x >> 6), cz = blockIdx.z;
	float m = 0.0f;
	if (x < g.X && cy < CY) {
		if (4 * cz + 4 <= g.Zg && 4 * cy + 4 <= g.Y) {                             // the whole column: 16 independent loads in flight
			float a[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) a[k] = ColTex<HALF>::ldw(col, ((size_t)(4 * cz + (k >> 2)) * g.Y + (4 * cy + (k & 3))) * g.X + x);
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				alpha[((size_t)(4 * cz + (k >> 2)) * g.Y + (4 * cy + (k & 3))) * g.X + x] = a[k];
				m = fmaxf(m, a[k]);
			}
		} else {
			for (int z = 4 * cz; z < min(4 * cz + 4, g.Zg); ++z)
				for (int y = 4 * cy; y < min(4 * cy + 4, g.Y); ++y) {
					const size_t i = ((size_t)z * g.Y + y) * g.X + x;
					const float a = ColTex<HALF>::ldw(col, i);
					alpha[i] = a;
					m = fmaxf(m, a);
				}
		}
	}
	m = fmaxf(m, __shfl_xor(m, 1));
	m = fmaxf(m, __shfl_xor(m, 2));
	if (x < g.X && cy < CY && (x & 3) == 0) blk[((size_t)cz * CY + cy) * CX + (x >> 2)] = m;
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

hipError_t launch_accel_build(const Geom& g, int half_store, const void* color, const RenderAccel& a, hipStream_t s)
{
	const int n = a.CX * a.CY * a.CZ;
	float* blk = a.occ + n;
	const dim3 grid((g.X + 63) / 64, (a.CY + 3) / 4, a.CZ), block(256);
	if (half_store) hipLaunchKernelGGL(k_occupancy_blocks<true>, grid, block, 0, s, g, (const h16x4*)color, blk, a.alpha, a.ctr);
	else hipLaunchKernelGGL(k_occupancy_blocks<false>, grid, block, 0, s, g, (const float4*)color, blk, a.alpha, a.ctr);
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
// pass 1: every voxel takes its centre sample (:37); the empty ones (density < 0.01, :44) get the constant `light colour + ambient`
// (shadow = 1; with the light probe: ao * irradiance = 1 * 0), the others go on the list of their z plane: list[z * X * Y + k],
// k < cnt[z * kCntStride].  Masks and alpha are read from global memory here (one coalesced look-up per wave, nothing depends on it).
static const int kCntStride = 32;                   // words between two plane counters: one 128-byte line each

__global__ __launch_bounds__(256) void k_light_classify(const Geom g, const float* __restrict__ alpha, const uint32_t* __restrict__ pos_fine, int CX, int CY,
	uint32_t* __restrict__ list, uint32_t* __restrict__ cnt, uint32_t* __restrict__ lightmap, const FrameConsts fc, int has_sh)
{
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	const int z = blockIdx.z;
	const bool valid = x < g.X && y < g.Y;
	bool lit = false;
	const uint32_t XY = (uint32_t)g.X * (uint32_t)g.Y;
	const uint32_t id = (uint32_t)z * XY + (uint32_t)y * (uint32_t)g.X + (uint32_t)x;
	if (valid) {
		const AccelVol<false, false> vol{ nullptr, alpha, nullptr, pos_fine, pos_fine, 0, CX, CY, CX, CY };
		const float ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);        // CSRayMarchL.hlsl:22
		const float oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
		const float oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
		const float density = density_at(vol, g, fmaf(ox, 0.5f, 0.5f), fmaf(oy, 0.5f, 0.5f), fmaf(oz, 0.5f, 0.5f));   // :36-37
		lit = density >= 0.00999999978f;                                           // :44
		if (!lit) {
			const float irr[3] = { 0.0f, 0.0f, 0.0f };
			lightmap[id] = light_value(fc, has_sh != 0, 1.0f, 1.0f, irr);
		}
	}
	const unsigned long long b = __ballot(lit);
	if (b) {                                                                       // one atomic per wave; x order survives inside the wave's run
		const uint32_t lane = threadIdx.x & 63u;
		const int first = __ffsll((long long)b) - 1;
		uint32_t base = 0;
		if ((int)lane == first) base = atomicAdd(&cnt[z * kCntStride], (uint32_t)__popcll(b));
		base = (uint32_t)__shfl((int)base, first);
		if (lit) list[(uint32_t)z * XY + base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = id;
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
	if (wave == 0) {
		uint32_t carry = 0;
		for (uint32_t z0 = 0; z0 < Z; z0 += 64) {
			const uint32_t z = z0 + lane;
			uint32_t v = z < Z ? (cnt[z * kCntStride] + 63u) >> 6 : 0u;
#pragma unroll
			for (int d = 1; d < 64; d <<= 1) {
				const uint32_t up = (uint32_t)__shfl_up((int)v, d);
				if ((int)lane >= d) v += up;
			}
			if (z < Z) pre[z + 1] = carry + v;
			carry += (uint32_t)__shfl((int)v, 63);
		}
		if (lane == 0) pre[0] = 0;
	}
	__syncthreads();
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
		uint32_t lo = 0, hi = Z;                                                   // plane of chunk c: pre[lo] <= c < pre[lo + 1]
		while (hi - lo > 1) {
			const uint32_t mid = (lo + hi) >> 1;
			if (pre[mid] <= c) lo = mid; else hi = mid;
		}
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

hipError_t launch_accel_light(const Geom& g, const RenderAccel& a, uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	uint32_t num_samples, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((g.X + 63) / 64, (g.Y + 3) / 4, g.Zg), block(64, 4, 1);
	hipLaunchKernelGGL(k_light_classify, grid, block, 0, s, g, a.alpha, a.bits, a.CX, a.CY, a.list, a.ctr, lightmap, fc, sh ? 1 : 0);
	const size_t cells = (size_t)g.X * g.Y * g.Zg;
	const unsigned wgs = (unsigned)std::min<size_t>((cells + 255) / 256, 2048);
	const size_t lds = (size_t)a.mask_words * 4 + ((size_t)g.Zg + 1) * 4;
	if (a.msh) hipLaunchKernelGGL(k_light_march<true>, dim3(wgs), dim3(256), lds, s, g, a.alpha, a.occ, mask_pos(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY,
		a.list, a.ctr, lightmap, fc, sh, num_samples, counters);
	else hipLaunchKernelGGL(k_light_march<false>, dim3(wgs), dim3(256), lds, s, g, a.alpha, a.occ, mask_pos(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY,
		a.list, a.ctr, lightmap, fc, sh, num_samples, counters);
	return hipGetLastError();
}

size_t render_accel_ctr_words(const Geom& g) { return (size_t)g.Zg * kCntStride; }

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

static MaskArgs mask_args(const RenderAccel& a) { return MaskArgs{ mask_pos(a), mask_vis(a), a.mask_words, a.msh, a.MX, a.MY, a.CX, a.CY }; }

hipError_t launch_accel_view(const Geom& g, int half_store, const void* color, const uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	int cube_size, uint32_t mask, uint32_t num_samples, uint32_t num_light_samples, int separate, uint8_t* cube, const RenderAccel& a, hipStream_t s,
	unsigned long long* counters)
{
	const dim3 grid((cube_size + 15) / 16, (cube_size + 15) / 16, 6), block(256);
	const size_t lds = (size_t)a.mask_words * 4 * (separate ? 1 : 2);
	uint32_t* out = reinterpret_cast<uint32_t*>(cube);
	const MaskArgs m = mask_args(a);
#define FX_LAUNCH(H, S, C) hipLaunchKernelGGL((k_view_march<H, S, C>), grid, block, lds, s, g, (const typename ColTex<H>::T*)color, a.alpha, a.occ, m, \
	lightmap, fc, sh, cube_size, mask, num_samples, num_light_samples, out, counters)
#define FX_PICK(H, S) do { if (a.msh) FX_LAUNCH(H, S, true); else FX_LAUNCH(H, S, false); } while (0)
	if (half_store) { if (separate) FX_PICK(true, true); else FX_PICK(true, false); }
	else { if (separate) FX_PICK(false, true); else FX_PICK(false, false); }
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
	if (half_store) { if (separate) FX_PICK(true, true); else FX_PICK(true, false); }
	else { if (separate) FX_PICK(false, true); else FX_PICK(false, false); }
#undef FX_LAUNCH
#undef FX_PICK
	return hipGetLastError();
}

}  // namespace fx
