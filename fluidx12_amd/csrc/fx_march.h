// fx_march.h -- device-side building blocks of the ray marches, shared by the plain kernels (fx_render.hip: every sample gathers its
// taps) and the accelerated ones (fx_render_accel.hip: occupancy masks in the LDS, an alpha-only side volume, compacted light voxels).
//
//   CastLightRay        RayMarch.hlsli:215-247        cast_light_ray
//   GetStep             RayMarch.hlsli:200-210        step_factor
//   GetDensityGradient  RayMarch.hlsli:73-95          gi_term (with CSRayMarchL.hlsl:59-68 / RayMarch.hlsli:275-283)
//   EvaluateSHIrradiance  XUSG/Shaders/SHIrradianceTypeless.hlsli:16-37   sh_irradiance
//   ComputeRayOrigin    RayMarch.hlsli:146-173        compute_ray_origin
//   the view loop       CSRayMarch.hlsl:140-190 == PSRayCast.hlsl:72-122   march_ray
// (paths relative to /root/reference/FluidX12/Content/Shaders/).  Association order follows the shipped DXBC; a DXBC `mad` is
// fmaf(); rsq is 1/sqrtf.  The volume is filtered manually with fp32 weights (hardware filtering uses ~8-bit weights and would
// break parity).
//
// Every march is written against a *volume policy* V:
//   V::dense(b)    false only where the trilinear alpha at base tap b is known to be +0      (light / AO rays, density gradient)
//   V::visible(b)  false only where it is known not to exceed the view march's 0.01 threshold (view rays)
//   V::density(t)  trilinear alpha                V::color(t)  trilinear rgba
// A sample that is not dense IS +0 (lerp(0, 0, f) = fma(f, 0, 0)), a sample that is not visible cannot pass `alpha > 0.01` (a convex
// combination of values <= m rounds to <= m), so in both cases the march continues with exactly the arithmetic it would have done.
// The loops are written in two phases -- every lane first walks through the samples it can decide without memory, then the lanes
// that need taps gather together -- so that a wave pays one memory round trip per gathered sample of its busiest lane, not one per
// loop iteration.  And a gathering lane fetches K samples per round trip, not one: a march is a chain of DEPENDENT fetches (the
// next position follows from this sample's GetStep), the longest ray of a launch is the launch's duration (measured: one wave of
// the 256^3 view march ran 1.06 M cycles for ~190 gathers while the average wave ran 0.12 M), and in the thin smoke where chains
// get that long GetStep returns 1 -- the plain step.  So the taps of the K - 1 samples that follow IF every step in between is the
// plain one are requested together with the first; each is used only when the march really arrives there (same t, bit for bit),
// otherwise dropped.  What is computed is unchanged, only when its operands are fetched.
#pragma once
#include "fx_internal.h"

namespace fx {

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

template <bool HALF> struct ColTex;
template <> struct ColTex<false> {
	typedef float4 T;
	static __device__ __forceinline__ float4 ld(const T* p, size_t i) { return p[i]; }
	static __device__ __forceinline__ float ldw(const T* p, size_t i) { return reinterpret_cast<const float*>(p)[4 * i + 3]; }
};
template <> struct ColTex<true> {
	typedef h16x4 T;
	static __device__ __forceinline__ float4 ld(const T* p, size_t i)
	{
		const h16x4 h = p[i];
		return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
	}
	static __device__ __forceinline__ float ldw(const T* p, size_t i) { return (float)reinterpret_cast<const h16*>(p)[4 * i + 3]; }
};

__device__ __forceinline__ float lerp1(float a, float b, float f) { return fmaf(f, b - a, a); }
__device__ __forceinline__ float rsqf(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
	return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

// ---- R11G11B10_FLOAT, round-to-nearest-even, negatives -> 0 --------------------------------------
__device__ __forceinline__ uint32_t pack_uf(float f, int mbits)
{
	const uint32_t x = __float_as_uint(f);
	const uint32_t maxfinite = (31u << mbits) - 1u;
	if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (31u << mbits) | 1u;
	if (x & 0x80000000u) return 0u;
	if (x == 0x7F800000u) return 31u << mbits;
	const int drop = 23 - mbits;
	if (x < 0x38800000u) return (uint32_t)rintf(f * __uint_as_float((uint32_t)(127 + 14 + mbits) << 23));   // subnormal, RNE
	uint32_t v = x - (112u << 23);
	v += ((1u << (drop - 1)) - 1u) + ((v >> drop) & 1u);
	v >>= drop;
	return v > maxfinite ? maxfinite : v;
}
// branch-free (the view march decodes 24 of these per gathered sample, on the critical path of its longest rays): the field
// shifted into place is a float with exponent e - 127 (a float denormal for e = 0), and 2^112 rebiases both cases exactly
__device__ __forceinline__ float unpack_uf(uint32_t b, int mbits)
{
	const float v = __uint_as_float(b << (23 - mbits)) * __uint_as_float((127u + 112u) << 23);
	const uint32_t special = (b & ((1u << mbits) - 1u)) ? 0x7FC00000u : 0x7F800000u;      // e = 31: NaN / +inf
	return b >= (31u << mbits) ? __uint_as_float(special) : v;
}
__device__ __forceinline__ uint32_t pack_r11g11b10(float r, float g, float b)
{
	return pack_uf(r, 6) | (pack_uf(g, 6) << 11) | (pack_uf(b, 5) << 22);
}
__device__ __forceinline__ float3 unpack_r11g11b10(uint32_t v)
{
	return make_float3(unpack_uf(v & 0x7FFu, 6), unpack_uf((v >> 11) & 0x7FFu, 6), unpack_uf(v >> 22, 5));
}

// ---- trilinear taps (LINEAR_CLAMP, Fluid.cpp:475) ---------------------------------------------------
// Base = where a sample's 2 x 2 x 2 footprint starts: clamped base texel + the fp32 filter weights.  The 4^3 occupancy block of
// (x0, y0, z0) bounds the alpha of the voxels [4c, 4c + 4] per axis, which contains all eight taps whatever the clamping did.
struct Base { int ix, iy, iz, x0, y0, z0; float fx, fy, fz; };
struct Taps { uint32_t i[8]; float fx, fy, fz; uint32_t dx, xs; };   // dx: tap 1 is tap 0's right neighbour (not clamped onto it); xs: tap 0 is its row's last voxel

__device__ __forceinline__ Base make_base(const Geom& g, float u, float v, float w, int ox = 0, int oy = 0, int oz = 0)
{
	const float tx = u * (float)g.X - 0.5f, ty = v * (float)g.Y - 0.5f, tz = w * (float)g.Zg - 0.5f;
	const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
	Base b;
	b.fx = tx - flx; b.fy = ty - fly; b.fz = tz - flz;
	b.ix = (int)flx + ox; b.iy = (int)fly + oy; b.iz = (int)flz + oz;
	b.x0 = min(max(b.ix, 0), g.X - 1); b.y0 = min(max(b.iy, 0), g.Y - 1); b.z0 = min(max(b.iz, 0), g.Zg - 1);
	return b;
}

// Voxel indices of the eight taps.  Index arithmetic is most of what a sample costs besides its loads, so: 24-bit multiplies (full rate;
// v_mul_lo_u32 / v_mad_u64_u32 issue at a quarter of it) -- fx_render refuses grids with Y (Z + 1) >= 2^24 -- and the clamped "+ 1"
// neighbours as a conditional stride: clamp(i + 1) differs from clamp(i) exactly where 0 <= i < N - 1, one unsigned compare.
// (inline assembly: where the compiler can bound the operands it turns __umul24(a, b) + c back into a plain multiply-add and selects
// v_mad_u64_u32 for it)
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
	uint32_t r;
	asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}

__device__ __forceinline__ Taps make_taps(const Geom& g, const Base& b)
{
	Taps t;
	t.fx = b.fx; t.fy = b.fy; t.fz = b.fz;
	const uint32_t X = (uint32_t)g.X, XY = (uint32_t)g.X * (uint32_t)g.Y;
	const uint32_t dx = (uint32_t)b.ix < (uint32_t)(g.X - 1) ? 1u : 0u;
	const uint32_t dy = (uint32_t)b.iy < (uint32_t)(g.Y - 1) ? X : 0u;
	const uint32_t dz = (uint32_t)b.iz < (uint32_t)(g.Zg - 1) ? XY : 0u;
	t.dx = dx; t.xs = b.x0 == g.X - 1 ? 1u : 0u;
	t.i[0] = mad24(mad24((uint32_t)b.z0, (uint32_t)g.Y, (uint32_t)b.y0), X, (uint32_t)b.x0);
	t.i[1] = t.i[0] + dx; t.i[2] = t.i[0] + dy; t.i[3] = t.i[2] + dx;
	t.i[4] = t.i[0] + dz; t.i[5] = t.i[4] + dx; t.i[6] = t.i[2] + dz; t.i[7] = t.i[6] + dx;
	return t;
}

// a load at a 32-bit byte offset from a uniform base: `global_load v, v_offset, s[base:base+1]` -- no 64-bit address to put together per
// tap (a v_mov for the high half and a v_lshl_add_u64 each).  For volumes below 4 GiB: the accelerated path's (fx_create gives a grid
// its scratch only up to 2^28 voxels)
template <typename T>
__device__ __forceinline__ T ld_off32(const void* base, uint32_t byte_off)
{
	return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ float blend8(const float q[8], const Taps& t)
{
	return lerp1(lerp1(lerp1(q[0], q[1], t.fx), lerp1(q[2], q[3], t.fx), t.fy),
		lerp1(lerp1(q[4], q[5], t.fx), lerp1(q[6], q[7], t.fx), t.fy), t.fz);
}

__device__ __forceinline__ float4 blend8x4(const float4 c[8], const Taps& t)
{
	float q[8];
	float4 r;
#define FX_CH(m) { _Pragma("unroll") for (int k = 0; k < 8; ++k) q[k] = c[k].m; r.m = blend8(q, t); }
	FX_CH(x) FX_CH(y) FX_CH(z) FX_CH(w)
#undef FX_CH
	return r;
}

__device__ __forceinline__ void light_taps(const uint32_t* __restrict__ lm, const Taps& t, uint32_t raw[8])
{
#pragma unroll
	for (int k = 0; k < 8; ++k) raw[k] = lm[t.i[k]];
}

__device__ __forceinline__ float3 blend_light(const uint32_t raw[8], const Taps& t)
{
	float3 c[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) c[k] = unpack_r11g11b10(raw[k]);
	float q[8];
	float3 r;
#define FX_CH(m) { _Pragma("unroll") for (int k = 0; k < 8; ++k) q[k] = c[k].m; r.m = blend8(q, t); }
	FX_CH(x) FX_CH(y) FX_CH(z)
#undef FX_CH
	return r;
}

// ---- volume policies ----------------------------------------------------------------------------------
// every sample gathers: the reference's own behaviour, and the in-product yardstick of the accelerated path
template <bool HALF> struct PlainVol {
	const typename ColTex<HALF>::T* __restrict__ col;
	__device__ __forceinline__ bool dense(const Base&) const { return true; }
	__device__ __forceinline__ bool visible(const Base&) const { return true; }
	__device__ __forceinline__ void density_taps(const Taps& t, float q[8]) const
	{
#pragma unroll
		for (int k = 0; k < 8; ++k) q[k] = ColTex<HALF>::ldw(col, t.i[k]);
	}
	__device__ __forceinline__ void color_taps(const Taps& t, float4 c[8]) const
	{
#pragma unroll
		for (int k = 0; k < 8; ++k) c[k] = ColTex<HALF>::ld(col, t.i[k]);
	}
	__device__ __forceinline__ void light_taps(const uint32_t* __restrict__ lm, const Taps& t, uint32_t raw[8]) const { fx::light_taps(lm, t, raw); }
	__device__ __forceinline__ float density(const Taps& t) const { float q[8]; density_taps(t, q); return blend8(q, t); }
	__device__ __forceinline__ float4 color(const Taps& t) const { float4 c[8]; color_taps(t, c); return blend8x4(c, t); }
};

// Occupancy masks (one bit per 4^3 << L block; `pos`: some alpha of the block's footprint is not +0, `vis`: some alpha exceeds the
// 0.01 of CSRayMarch.hlsl:161) -- in the LDS where the kernel put them there -- and the alpha-only fp32 side volume that
// k_occupancy_blocks writes: a density tap costs 4 bytes of a dense array instead of 4 of a 16-byte (8-byte) texel.
// COARSE (grids whose 4^3 blocks do not fit the LDS budget): a set bit is confirmed against the fine grid `occ` before gathering.
template <bool HALF, bool COARSE> struct AccelVol {
	const typename ColTex<HALF>::T* __restrict__ col;
	const float* __restrict__ alpha;
	const float* __restrict__ occ;         // fine grid: max alpha over the footprint of each 4^3 block
	const uint32_t* pos;                   // bit masks at block size 4 << msh
	const uint32_t* vis;
	int msh, MX, MY, CX, CY;
	__device__ __forceinline__ uint32_t mcell(const Base& b) const
	{
		return mad24(mad24((uint32_t)b.z0 >> (2 + msh), (uint32_t)MY, (uint32_t)b.y0 >> (2 + msh)), (uint32_t)MX, (uint32_t)b.x0 >> (2 + msh));
	}
	__device__ __forceinline__ uint32_t fcell(const Base& b) const { return mad24(mad24((uint32_t)b.z0 >> 2, (uint32_t)CY, (uint32_t)b.y0 >> 2), (uint32_t)CX, (uint32_t)b.x0 >> 2); }
	__device__ __forceinline__ bool dense(const Base& b) const
	{
		const uint32_t c = mcell(b);
		if (!((pos[c >> 5] >> (c & 31u)) & 1u)) return false;
		if (COARSE) return !(occ[fcell(b)] == 0.0f);
		return true;
	}
	__device__ __forceinline__ bool visible(const Base& b) const
	{
		const uint32_t c = mcell(b);
		if (!((vis[c >> 5] >> (c & 31u)) & 1u)) return false;
		if (COARSE) return !(occ[fcell(b)] <= 0.00999999978f);
		return true;
	}
	// The two x taps of a row as ONE 8-byte load (any 4-byte alignment): the marches' gathers are bound by the vector L1's tag look-ups
	// (0.6 per cycle and CU measured on every gathering kernel, lanes mostly in different lines), and the pair shares its line 31 times out
	// of 32.  At the row's last voxel the pair starts one to the left (both taps are that voxel: the clamp); rows have >= 2 voxels.
	__device__ __forceinline__ void density_taps(const Taps& t, float q[8]) const
	{
		typedef float pair_t __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const pair_t p = ld_off32<pair_t>(alpha, (t.i[2 * k] - t.xs) << 2);
			q[2 * k] = t.xs ? p.y : p.x;
			q[2 * k + 1] = t.dx ? p.y : q[2 * k];
		}
	}
	__device__ __forceinline__ void color_taps(const Taps& t, float4 c[8]) const
	{
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			if (HALF) { const h16x4 h = ld_off32<h16x4>(col, t.i[k] << 3); c[k] = make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w); }
			else c[k] = ld_off32<float4>(col, t.i[k] << 4);
		}
		// (two 8-byte fp16 texels of a row as one 16-byte load: view pass 0.120 -> 0.144 ms, direct march 0.23 -> 0.42 -- the march reads a
		// texel's alpha first and the rest only where that passes 0.01, which a paired load forfeits)
	}
	__device__ __forceinline__ void light_taps(const uint32_t* __restrict__ lm, const Taps& t, uint32_t raw[8]) const
	{
		typedef uint32_t pair_t __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const pair_t p = ld_off32<pair_t>(lm, (t.i[2 * k] - t.xs) << 2);
			raw[2 * k] = t.xs ? p.y : p.x;
			raw[2 * k + 1] = t.dx ? p.y : raw[2 * k];
		}
	}
	__device__ __forceinline__ float density(const Taps& t) const { float q[8]; density_taps(t, q); return blend8(q, t); }
	__device__ __forceinline__ float4 color(const Taps& t) const { float4 c[8]; color_taps(t, c); return blend8x4(c, t); }
};

__device__ __forceinline__ bool outside(float x, float y, float z) { return fabsf(x) > 1.0f || fabsf(y) > 1.0f || fabsf(z) > 1.0f; }

// GetStep (RayMarch.hlsli:200-210) as compiled
__device__ __forceinline__ float step_factor(float dDensity, float transm, float density)
{
	// (a wave marching through empty space -- every lane's density difference 0 -- skips the division: 0.0039 / 0 = +inf, min(inf, 2) = 2)
	float ev = 2.0f;
	if (__builtin_amdgcn_ballot_w64(dDensity != 0.0f) != 0) ev = fminf(0.00390625f / fabsf(dDensity), 2.0f);
	const float ui = fminf(-density + 1.0f, 1.0f);
	const float th = -transm + 1.0f;
	return fmaxf(th * (ui * (ev * 1.5f)), 1.0f);
}

// one trilinear alpha, decided without memory where the policy can
template <class V>
__device__ __forceinline__ float density_at(const V& vol, const Geom& g, float u, float v, float w, int ox = 0, int oy = 0, int oz = 0)
{
	const Base b = make_base(g, u, v, w, ox, oy, oz);
	if (!vol.dense(b)) return 0.0f;
	return vol.density(make_taps(g, b));
}

// one sample of CastLightRay's loop body behind the fetch (RayMarch.hlsli:226-245); false = the ray ends here; fac = the step taken
__device__ __forceinline__ bool light_step(float density, float stepScale, float& t, float& prev, float& transm, uint32_t& i, float& fac)
{
	const float nt = fmaf(-density, 0.800000012f, 1.0f) * transm;
	if (nt < 0.00999999978f) { transm = nt; return false; }
	fac = step_factor(-prev + density, transm, density);
	t = fmaf(stepScale, fac, t);
	transm = nt;
	prev = density;
	++i;
	return true;
}

// CastLightRay (RayMarch.hlsli:215-247); ns = density samples taken (FX_OPT_COUNT_SAMPLES; a dead register otherwise);
// K = samples fetched per round trip (see the head of the file)
template <int K, class V>
__device__ void cast_light_ray(float& transm, const Geom& g, const V& vol,
	float ox, float oy, float oz, float dx, float dy, float dz, float stepScale, uint32_t numSamples, uint32_t& ns)
{
	float t = stepScale, prev = 0.0f;
	uint32_t i = 0;
	bool live = true;
	while (live) {
		// phase 1: samples this lane can decide without memory (outside the occupied blocks the density is +0)
		Base b0;
		bool gather = false;
		while (true) {
			if (i >= numSamples) { live = false; break; }
			const float px = fmaf(dx, t, ox), py = fmaf(dy, t, oy), pz = fmaf(dz, t, oz);
			if (outside(px, py, pz)) { live = false; break; }
			++ns;
			b0 = make_base(g, fmaf(px, 0.5f, 0.5f), fmaf(py, 0.5f, 0.5f), fmaf(pz, 0.5f, 0.5f));
			if (vol.dense(b0)) { gather = true; break; }
			float fac;
			if (!light_step(0.0f, stepScale, t, prev, transm, i, fac)) { live = false; break; }
		}
		// phase 2: the lanes that stopped on an occupied block gather together -- this sample and the K - 1 that follow it if
		// GetStep keeps returning 1 (kind: 0 = the ray would have ended before, 1 = decided without memory, 2 = taps requested)
		if (gather) {
			Taps tp[K];
			int kind[K];
			tp[0] = make_taps(g, b0);
			kind[0] = 2;
			{
				float tk = t;
				uint32_t ik = i;
				bool chain = true;
#pragma unroll
				for (int k = 1; k < K; ++k) {
					kind[k] = 0;
					if (chain) {
						tk = fmaf(stepScale, 1.0f, tk);
						++ik;
						const float px = fmaf(dx, tk, ox), py = fmaf(dy, tk, oy), pz = fmaf(dz, tk, oz);
						if (ik >= numSamples || outside(px, py, pz)) chain = false;
						else {
							const Base b = make_base(g, fmaf(px, 0.5f, 0.5f), fmaf(py, 0.5f, 0.5f), fmaf(pz, 0.5f, 0.5f));
							if (vol.dense(b)) { kind[k] = 2; tp[k] = make_taps(g, b); } else kind[k] = 1;
						}
					}
				}
			}
			float q[K][8];
#pragma unroll
			for (int k = 0; k < K; ++k) if (kind[k] == 2) vol.density_taps(tp[k], q[k]);
			float fac = 1.0f;
			bool cont = true;
#pragma unroll
			for (int k = 0; k < K; ++k) {
				if (k > 0 && cont) {
					if (!live || fac != 1.0f) cont = false;                        // the march did not arrive at the predicted sample
					else if (kind[k] == 0) { live = false; cont = false; }         // it did, and ends there (sample count or cube left)
					else ++ns;
				}
				if (cont) {
					const float density = kind[k] == 2 ? blend8(q[k], tp[k]) : 0.0f;
					live = light_step(density, stepScale, t, prev, transm, i, fac);
				}
			}
		}
	}
}

// EvaluateSHIrradiance (SHIrradianceTypeless.hlsli:16-37), compiled association order; sh = 9 x float3
__device__ __forceinline__ void sh_irradiance(float out[3], const float* __restrict__ sh, float nx, float ny, float nz)
{
	const float c1 = 0.429042757f, c3 = 0.247707963f, c4 = 0.886226952f, c1x2 = 0.858085513f, c2x2 = 1.02332675f;
	const float a = fmaf(nx, nx, -(ny * ny)) * c1;
	const float b = fmaf(nz * nz, 3.0f, -1.0f) * c3;
	const float mx = -nx, my = -ny;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const float* L = sh + k;
		float r = L[18] * b;
		r = fmaf(a, L[24], r);
		r = fmaf(L[0], c4, r);
		float q = (L[21] * mx) * nz;
		q = fmaf(L[12] * mx, my, q);
		q = fmaf(L[15] * my, nz, q);
		r = fmaf(q, c1x2, r);
		float l = L[3] * my;
		l = fmaf(L[9], mx, l);
		l = fmaf(L[6], nz, l);
		r = fmaf(l, c2x2, r);
		out[k] = fmaxf(r, 0.0f);
	}
}

// GI branch of CSRayMarchL.hlsl:59-68 / RayMarch.hlsli:275-283
template <int K, class V>
__device__ void gi_term(float irr[3], float& ao, const Geom& g, const V& vol, const FrameConsts& fc,
	const float* __restrict__ sh, float px, float py, float pz, float u, float v, float w, float stepScale, uint32_t numSamples, uint32_t& ns)
{
	ns += 6;
	// GetDensityGradient (RayMarch.hlsli:73-95)
	const float qxm = density_at(vol, g, u, v, w, -1, 0, 0);
	const float qxp = density_at(vol, g, u, v, w, 1, 0, 0);
	const float qym = density_at(vol, g, u, v, w, 0, -1, 0);
	const float qyp = density_at(vol, g, u, v, w, 0, 1, 0);
	const float qzm = density_at(vol, g, u, v, w, 0, 0, -1);
	const float qzp = density_at(vol, g, u, v, w, 0, 0, 1);
	const float gx = -qxm + qxp, gy = -qym + qyp, gz = -qzm + qzp;
	const bool any = fabsf(gx) > 0.0f || fabsf(gy) > 0.0f || fabsf(gz) > 0.0f;
	float dx = any ? -gx : px, dy = any ? -gy : py, dz = any ? -gz : pz;
	float wx = dot3(dx, dy, dz, fc.world[0], fc.world[1], fc.world[2]);
	float wy = dot3(dx, dy, dz, fc.world[4], fc.world[5], fc.world[6]);
	float wz = dot3(dx, dy, dz, fc.world[8], fc.world[9], fc.world[10]);
	const float rw = rsqf(dot3(wx, wy, wz, wx, wy, wz));
	wx *= rw; wy *= rw; wz *= rw;
	sh_irradiance(irr, sh, wx, wy, wz);
	const float rd = rsqf(dot3(dx, dy, dz, dx, dy, dz));
	dx *= rd; dy *= rd; dz *= rd;
	ao = 1.0f;
	cast_light_ray<K>(ao, g, vol, px, py, pz, dx, dy, dz, stepScale, numSamples, ns);
}

__device__ __forceinline__ void light_dir_local(const FrameConsts& fc, float& lx, float& ly, float& lz)
{
	lx = dot3(fc.light_pt[0], fc.light_pt[1], fc.light_pt[2], fc.world_i[0], fc.world_i[1], fc.world_i[2]);
	ly = dot3(fc.light_pt[0], fc.light_pt[1], fc.light_pt[2], fc.world_i[4], fc.world_i[5], fc.world_i[6]);
	lz = dot3(fc.light_pt[0], fc.light_pt[1], fc.light_pt[2], fc.world_i[8], fc.world_i[9], fc.world_i[10]);
	const float r = rsqf(dot3(lx, ly, lz, lx, ly, lz));
	lx *= r; ly *= r; lz *= r;
}

// the light-map value of one voxel (CSRayMarchL.hlsl:22-79) given its centre sample; `lit` = density >= 0.01 (:44)
__device__ __forceinline__ uint32_t light_value(const FrameConsts& fc, bool has_sh, float shadow, float ao, const float irr[3])
{
	float out[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float lc = fc.light_color[3] * fc.light_color[a];
		const float amb = has_sh ? ao * irr[a] : fc.ambient[3] * fc.ambient[a];        // :72-76
		out[a] = fmaf(shadow, lc, amb);                                                // :79
	}
	return pack_r11g11b10(out[0], out[1], out[2]);
}

// FX_OPT_COUNT_SAMPLES: counters = [64 shards][3] { colour samples of view rays, density samples of light / AO rays, light-map fetches };
// null in every timed launch (the per-thread counts are then dead registers)
__device__ __forceinline__ void flush_counts(unsigned long long* __restrict__ counters, uint32_t view, uint32_t light, uint32_t lm)
{
	if (!counters) return;
	unsigned long long* c = counters + 3 * ((blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u + (threadIdx.x >> 6)) & 63u);
	if (view) atomicAdd(c + 0, (unsigned long long)view);
	if (light) atomicAdd(c + 1, (unsigned long long)light);
	if (lm) atomicAdd(c + 2, (unsigned long long)lm);
}

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool compute_ray_origin(float o[3], const float d[3])   // RayMarch.hlsli:146-173
{
	if (fabsf(o[0]) <= 1.0f && fabsf(o[1]) <= 1.0f && fabsf(o[2]) <= 1.0f) return true;
	float U = 3.40282347e+38f;
	bool hit = false;
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const float sgn = (float)((0.0f < d[i]) - (d[i] < 0.0f));
		const float u = (-o[i] + -sgn) / d[i];
		if (!(u >= 0.0f)) continue;
		const int j = (i + 1) % 3, k = (i + 2) % 3;
		if (!(1.0f >= fabsf(fmaf(d[j], u, o[j])))) continue;
		if (!(1.0f >= fabsf(fmaf(d[k], u, o[k])))) continue;
		if (u < U) { U = u; hit = true; }
	}
#pragma unroll
	for (int a = 0; a < 3; ++a) o[a] = fminf(fmaxf(fmaf(d[a], U, o[a]), -1.0f), 1.0f);
	return hit;
}

__device__ __forceinline__ uint32_t to_unorm8(float v)
{
	if (!(v > 0.0f)) return 0u;
	if (v >= 1.0f) return 255u;
	return (uint32_t)(v * 255.0f + 0.5f);
}

// the march of one view ray (CSRayMarch.hlsl:140-190 == PSRayCast.hlsl:72-122): o = origin on/in the cube, d = unit
// direction, tMax = ray parameter at the cube-map target (the direct pixel march has none: FLT_MAX).  `go` = this lane has a ray.
// K = samples fetched per round trip (the merged march, whose samples cast rays of their own, takes one).
template <class V, bool SEPARATE, int K>
__device__ __forceinline__ void march_ray(const Geom& g, const V& vol, const uint32_t* __restrict__ lightmap, const FrameConsts& fc,
	const float* __restrict__ sh, const float o[3], const float d[3], float tMax, uint32_t numSamples, uint32_t numLightSamples, bool go,
	float& sr, float& sg, float& sb, float& sa, uint32_t& nv, uint32_t& nl, uint32_t& nm)
{
	static_assert(SEPARATE || K == 1, "the merged march fetches one sample per round trip");
	const float stepScale = 3.46410155f / (float)numSamples;
	const float lightStep = 3.46410155f / (float)numLightSamples;
	float lx = 0.0f, ly = 0.0f, lz = 0.0f;
	if (!SEPARATE) light_dir_local(fc, lx, ly, lz);

	sr = 0.0f; sg = 0.0f; sb = 0.0f; sa = 0.0f;
	float t = 0.0f, prev = 0.0f;
	uint32_t i = 0;
	bool live = go;
	while (live) {
		// phase 1: samples that cannot be seen (alpha <= 0.01 everywhere in the footprint) only advance the ray      :146-161,187-189
		Base b0;
		float qx = 0.0f, qy = 0.0f, qz = 0.0f, u = 0.0f, v = 0.0f, w = 0.0f;
		bool gather = false;
		while (true) {
			if (i >= numSamples) { live = false; break; }                          // :146
			qx = fmaf(d[0], t, o[0]); qy = fmaf(d[1], t, o[1]); qz = fmaf(d[2], t, o[2]);
			if (outside(qx, qy, qz)) { live = false; break; }                      // :149
			u = fmaf(qx, 0.5f, 0.5f); v = fmaf(qy, 0.5f, 0.5f); w = fmaf(qz, 0.5f, 0.5f);
			b0 = make_base(g, u, v, w);
			++nv;
			if (vol.visible(b0)) { gather = true; break; }
			++i;
			t = t + stepScale;                                                     // :187-188 with newStep = g_step
			if (tMax < t) { live = false; break; }                                 // :189
		}
		// phase 2: the lanes with a sample to look at fetch colour and light map of this sample and of the K - 1 that follow it
		// if every step in between is the plain one (kind: 0 = the ray would have ended before, 1 = cannot be seen, 2 = taps requested)
		if (gather) {
			Taps tp[K];
			int kind[K];
			tp[0] = make_taps(g, b0);
			kind[0] = 2;
			{
				float tk = t;
				uint32_t ik = i;
				bool chain = true;
#pragma unroll
				for (int k = 1; k < K; ++k) {
					kind[k] = 0;
					if (chain) {
						tk = tk + stepScale;
						++ik;
						const float px = fmaf(d[0], tk, o[0]), py = fmaf(d[1], tk, o[1]), pz = fmaf(d[2], tk, o[2]);
						if (tMax < tk || ik >= numSamples || outside(px, py, pz)) chain = false;
						else {
							const Base b = make_base(g, fmaf(px, 0.5f, 0.5f), fmaf(py, 0.5f, 0.5f), fmaf(pz, 0.5f, 0.5f));
							if (vol.visible(b)) { kind[k] = 2; tp[k] = make_taps(g, b); } else kind[k] = 1;
						}
					}
				}
			}
			float4 c8[K][8];
			uint32_t l8[K][8];
#pragma unroll
			for (int k = 0; k < K; ++k)
				if (kind[k] == 2) {
					vol.color_taps(tp[k], c8[k]);                                  // :157
					if (SEPARATE) vol.light_taps(lightmap, tp[k], l8[k]);              // RayMarch.hlsli:253-258 (used only behind :161)
				}
			float newStep = stepScale;
			bool cont = true;
#pragma unroll
			for (int k = 0; k < K; ++k) {
				if (k > 0 && cont) {
					if (!live || newStep != stepScale) cont = false;               // the march did not arrive at the predicted sample
					else if (kind[k] == 0) { live = false; cont = false; }         // it did, and ends there (:146, :149)
					else ++nv;
				}
				if (cont) {
					newStep = stepScale;
					if (kind[k] == 2) {
						const float4 c = blend8x4(c8[k], tp[k]);
						if (0.00999999978f < c.w) {                                // :161
							float light[3];
							if (SEPARATE) {
								const float3 l = blend_light(l8[k], tp[k]);
								++nm;
								light[0] = l.x; light[1] = l.y; light[2] = l.z;
							} else {                                               // RayMarch.hlsli:260-294
								float shadow = 1.0f, ao = 1.0f, irr[3] = { 0.0f, 0.0f, 0.0f };
								cast_light_ray<1>(shadow, g, vol, qx, qy, qz, lx, ly, lz, lightStep, numLightSamples, nl);
								if (sh) gi_term<1>(irr, ao, g, vol, fc, sh, qx, qy, qz, u, v, w, lightStep, numLightSamples, nl);
#pragma unroll
								for (int a = 0; a < 3; ++a) {
									const float amb = sh ? ao * irr[a] : fc.ambient[3] * fc.ambient[a];
									light[a] = fmaf(fc.light_color[3] * fc.light_color[a], shadow, amb);
								}
							}
							const float transm = -sa + 1.0f;                       // :170
							newStep = step_factor(-prev + c.w, transm, c.w) * stepScale;   // :172
							sr = fmaf(transm * (light[0] * c.x), 0.800000012f, sr);    // :180-181
							sg = fmaf(transm * (light[1] * c.y), 0.800000012f, sg);
							sb = fmaf(transm * (light[2] * c.z), 0.800000012f, sb);
							sa = fmaf(0.800000012f * c.w, transm, sa);
							if (transm < 0.00999999978f) live = false;             // :183
							prev = c.w;
						}
					}
					if (live) {
						++i;
						t = t + newStep;                                           // :187-188
						if (tMax < t) live = false;                                // :189
					}
				}
			}
		}
	}
}

// cube-map texel -> ray (CSRayMarch.hlsl:107-118, GetLocalPos :39-64); false = no ray (the texel is left untouched, :116)
__device__ __forceinline__ bool cube_texel_ray(const FrameConsts& fc, int face, int x, int y, int size, float o[3], float d[3], float& tMax)
{
#pragma unroll
	for (int a = 0; a < 3; ++a) {                                                  // :107
		const float* r = fc.world_i + 4 * a;
		o[a] = fmaf(r[3], 1.0f, fmaf(fc.eye_pt[2], r[2], fmaf(fc.eye_pt[1], r[1], fc.eye_pt[0] * r[0])));
	}
	const float px = fmaf(((float)x + 0.5f) / (float)size, 2.0f, -1.0f);
	const float py = -fmaf(((float)y + 0.5f) / (float)size, 2.0f, -1.0f);
	float tg[3];
	switch (face) {
	case 0: tg[0] = 1.0f;  tg[1] = py;    tg[2] = -px;   break;
	case 1: tg[0] = -1.0f; tg[1] = py;    tg[2] = px;    break;
	case 2: tg[0] = px;    tg[1] = 1.0f;  tg[2] = -py;   break;
	case 3: tg[0] = px;    tg[1] = -1.0f; tg[2] = py;    break;
	case 4: tg[0] = px;    tg[1] = py;    tg[2] = 1.0f;  break;
	default: tg[0] = -px;  tg[1] = py;    tg[2] = -1.0f; break;
	}
	d[0] = -o[0] + tg[0]; d[1] = -o[1] + tg[1]; d[2] = -o[2] + tg[2];
	const float rl = rsqf(dot3(d[0], d[1], d[2], d[0], d[1], d[2]));               // :115
	d[0] *= rl; d[1] *= rl; d[2] *= rl;
	if (!compute_ray_origin(o, d)) return false;                                   // :116
	tMax = fmaxf((tg[2] + -o[2]) / d[2], fmaxf((tg[1] + -o[1]) / d[1], (tg[0] + -o[0]) / d[0]));   // :118
	return true;
}

// screen pixel -> ray (PSRayCast.hlsl:17-26,47-50); false = discard
__device__ __forceinline__ bool pixel_ray(const FrameConsts& fc, int px, int py, int W, int H, float o[3], float d[3])
{
	const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
	const float qx = fmaf(u, 2.0f, -1.0f), qy = fmaf(v, -2.0f, 1.0f);
	const float* M = fc.wvp_i;
	const float h0 = dot3(qx, qy, 1.0f, M[0], M[1], M[3]), h1 = dot3(qx, qy, 1.0f, M[4], M[5], M[7]);
	const float h2 = dot3(qx, qy, 1.0f, M[8], M[9], M[11]), h3 = dot3(qx, qy, 1.0f, M[12], M[13], M[15]);
	o[0] = h0 / h3; o[1] = h1 / h3; o[2] = h2 / h3;
#pragma unroll
	for (int a = 0; a < 3; ++a) {                                                  // :47-49
		const float* r = fc.world_i + 4 * a;
		const float e = fmaf(r[3], 1.0f, fmaf(fc.eye_pt[2], r[2], fmaf(fc.eye_pt[1], r[1], fc.eye_pt[0] * r[0])));
		d[a] = o[a] + -e;
	}
	const float rl = rsqf(dot3(d[0], d[1], d[2], d[0], d[1], d[2]));
	d[0] *= rl; d[1] *= rl; d[2] *= rl;
	return compute_ray_origin(o, d);                                               // :50 discard
}

__device__ __forceinline__ uint32_t blend_premultiplied(uint32_t dd, float sr, float sg, float sb, float sa)
{
	const float ia = 1.0f - sa;
	return to_unorm8(fmaf((float)(dd & 255u) / 255.0f, ia, sr)) | (to_unorm8(fmaf((float)((dd >> 8) & 255u) / 255.0f, ia, sg)) << 8)
		| (to_unorm8(fmaf((float)((dd >> 16) & 255u) / 255.0f, ia, sb)) << 16) | (to_unorm8(fmaf((float)(dd >> 24) / 255.0f, ia, sa)) << 24);
}

}  // namespace fx
