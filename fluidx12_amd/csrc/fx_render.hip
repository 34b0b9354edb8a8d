// fx_render.hip -- gfx950 kernels of the cube-map-space ray march.
//
//   k_raymarch_light  <- CSRayMarchL.hlsl:15-80   one thread per light-map voxel: shadow ray (+ GI/AO ray)
//   k_raymarch_view   <- CSRayMarch.hlsl:98-196   one thread per cube-map texel of mip `LOD`
//                        SEPARATE = CSRayMarchV.hlsl:5-7 (light = light-map fetch),
//                        otherwise the merged variant with the nested light march (RayMarch.hlsli:260-294)
// (paths relative to /root/reference/FluidX12/Content/Shaders/; helpers from RayMarch.hlsli and
// XUSG/Shaders/SHIrradianceTypeless.hlsli:16-37).  Association order follows the shipped DXBC; a DXBC
// `mad` is fmaf(); rsq is 1/sqrtf.  The volume is read through L2 / Infinity Cache with manual fp32
// trilinear filtering (hardware filtering uses ~8-bit weights and would break parity).
// A wave64 = one 8x8 texel tile (view) or 64 consecutive x (light), so the taps of a wave are
// spatially coherent.  Dependent gather chains: latency/cache bound, no LDS staging, no MFMA.
// Light map = packed R11G11B10_FLOAT like the reference (Fluid.cpp:226), cube map = R8G8B8A8_UNORM.
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
__device__ __forceinline__ float unpack_uf(uint32_t b, int mbits)
{
	const uint32_t e = b >> mbits, m = b & ((1u << mbits) - 1u);
	if (e == 0) return (float)m * __uint_as_float((uint32_t)(127 - 14 - mbits) << 23);
	if (e == 31) return m ? __uint_as_float(0x7FC00000u) : __uint_as_float(0x7F800000u);
	return __uint_as_float(((e + 112u) << 23) | (m << (23 - mbits)));
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
// `cell` = the 4^3 occupancy block of the base tap; the block's entry in the occupancy grid is the largest alpha of voxels
// [4c, 4c + 4] per axis, which contains all eight taps whatever the clamping did
struct Taps { size_t i[8]; float fx, fy, fz; uint32_t cell; };

__device__ __forceinline__ Taps make_taps(const Geom& g, float u, float v, float w, int ox = 0, int oy = 0, int oz = 0)
{
	const float tx = u * (float)g.X - 0.5f, ty = v * (float)g.Y - 0.5f, tz = w * (float)g.Zg - 0.5f;
	const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
	Taps t;
	t.fx = tx - flx; t.fy = ty - fly; t.fz = tz - flz;
	const int ix = (int)flx + ox, iy = (int)fly + oy, iz = (int)flz + oz;
	const int x0 = min(max(ix, 0), g.X - 1), x1 = min(max(ix + 1, 0), g.X - 1);
	const int y0 = min(max(iy, 0), g.Y - 1), y1 = min(max(iy + 1, 0), g.Y - 1);
	const int z0 = min(max(iz, 0), g.Zg - 1), z1 = min(max(iz + 1, 0), g.Zg - 1);
	const size_t X = g.X, XY = g.plane();
	t.i[0] = z0 * XY + y0 * X + x0; t.i[1] = z0 * XY + y0 * X + x1;
	t.i[2] = z0 * XY + y1 * X + x0; t.i[3] = z0 * XY + y1 * X + x1;
	t.i[4] = z1 * XY + y0 * X + x0; t.i[5] = z1 * XY + y0 * X + x1;
	t.i[6] = z1 * XY + y1 * X + x0; t.i[7] = z1 * XY + y1 * X + x1;
	t.cell = (uint32_t)(((z0 >> 2) * ((g.Y + 3) >> 2) + (y0 >> 2)) * ((g.X + 3) >> 2) + (x0 >> 2));
	return t;
}

__device__ __forceinline__ float blend8(const float q[8], const Taps& t)
{
	return lerp1(lerp1(lerp1(q[0], q[1], t.fx), lerp1(q[2], q[3], t.fx), t.fy),
		lerp1(lerp1(q[4], q[5], t.fx), lerp1(q[6], q[7], t.fx), t.fy), t.fz);
}

// Empty-space skipping that changes no bit: where the occupancy grid says every tap has alpha == 0 the trilinear result IS +0
// (lerp(0, 0, f) = fma(f, 0, 0)), so the eight gathers are replaced by one 4-byte look-up of an L2-resident 1/64-size grid.
template <bool HALF>
__device__ __forceinline__ float sample_density(const typename ColTex<HALF>::T* col, const Taps& t, const float* __restrict__ occ = nullptr)
{
	if (occ && occ[t.cell] == 0.0f) return 0.0f;
	float q[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) q[k] = ColTex<HALF>::ldw(col, t.i[k]);
	return blend8(q, t);
}

// The view march only looks at a sample whose alpha exceeds 0.01 (CSRayMarch.hlsl:161): if no tap does, neither does their
// convex combination (round-to-nearest is monotonic), and the sample can be reported as empty without fetching it.
template <bool HALF>
__device__ __forceinline__ float4 sample_color(const typename ColTex<HALF>::T* col, const Taps& t, const float* __restrict__ occ = nullptr)
{
	if (occ && occ[t.cell] <= 0.00999999978f) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float4 c[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) c[k] = ColTex<HALF>::ld(col, t.i[k]);
	float q[8];
	float4 r;
#define FX_CH(m) { _Pragma("unroll") for (int k = 0; k < 8; ++k) q[k] = c[k].m; r.m = blend8(q, t); }
	FX_CH(x) FX_CH(y) FX_CH(z) FX_CH(w)
#undef FX_CH
	return r;
}

__device__ __forceinline__ float3 sample_light(const uint32_t* lm, const Taps& t)
{
	float3 c[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) c[k] = unpack_r11g11b10(lm[t.i[k]]);
	float q[8];
	float3 r;
#define FX_CH(m) { _Pragma("unroll") for (int k = 0; k < 8; ++k) q[k] = c[k].m; r.m = blend8(q, t); }
	FX_CH(x) FX_CH(y) FX_CH(z)
#undef FX_CH
	return r;
}

__device__ __forceinline__ bool outside(float x, float y, float z) { return fabsf(x) > 1.0f || fabsf(y) > 1.0f || fabsf(z) > 1.0f; }

// GetStep (RayMarch.hlsli:200-210) as compiled
__device__ __forceinline__ float step_factor(float dDensity, float transm, float density)
{
	const float ev = fminf(0.00390625f / fabsf(dDensity), 2.0f);
	const float ui = fminf(-density + 1.0f, 1.0f);
	const float th = -transm + 1.0f;
	return fmaxf(th * (ui * (ev * 1.5f)), 1.0f);
}

// CastLightRay (RayMarch.hlsli:215-247)
template <bool HALF>
__device__ void cast_light_ray(float& transm, const Geom& g, const typename ColTex<HALF>::T* col,
	float ox, float oy, float oz, float dx, float dy, float dz, float stepScale, uint32_t numSamples, const float* __restrict__ occ, uint32_t& ns)
{
	float t = stepScale, prev = 0.0f;
	for (uint32_t i = 0; i < numSamples; ++i) {
		const float px = fmaf(dx, t, ox), py = fmaf(dy, t, oy), pz = fmaf(dz, t, oz);
		if (outside(px, py, pz)) break;
		++ns;                                                  // density samples taken (FX_OPT_COUNT_SAMPLES; a register increment otherwise)
		const Taps tp = make_taps(g, fmaf(px, 0.5f, 0.5f), fmaf(py, 0.5f, 0.5f), fmaf(pz, 0.5f, 0.5f));
		const float density = sample_density<HALF>(col, tp, occ);
		const float nt = fmaf(-density, 0.800000012f, 1.0f) * transm;
		if (nt < 0.00999999978f) { transm = nt; break; }
		const float fac = step_factor(-prev + density, transm, density);
		t = fmaf(stepScale, fac, t);
		transm = nt;
		prev = density;
	}
}

// EvaluateSHIrradiance (SHIrradianceTypeless.hlsli:16-37), compiled association order; sh = 9 x float3
__device__ void sh_irradiance(float out[3], const float* __restrict__ sh, float nx, float ny, float nz)
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
template <bool HALF>
__device__ void gi_term(float irr[3], float& ao, const Geom& g, const typename ColTex<HALF>::T* col, const FrameConsts& fc,
	const float* __restrict__ sh, float px, float py, float pz, float u, float v, float w, float stepScale, uint32_t numSamples,
	const float* __restrict__ occ, uint32_t& ns)
{
	ns += 6;
	// GetDensityGradient (RayMarch.hlsli:73-95)
	const float qxm = sample_density<HALF>(col, make_taps(g, u, v, w, -1, 0, 0), occ);
	const float qxp = sample_density<HALF>(col, make_taps(g, u, v, w, 1, 0, 0), occ);
	const float qym = sample_density<HALF>(col, make_taps(g, u, v, w, 0, -1, 0), occ);
	const float qyp = sample_density<HALF>(col, make_taps(g, u, v, w, 0, 1, 0), occ);
	const float qzm = sample_density<HALF>(col, make_taps(g, u, v, w, 0, 0, -1), occ);
	const float qzp = sample_density<HALF>(col, make_taps(g, u, v, w, 0, 0, 1), occ);
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
	cast_light_ray<HALF>(ao, g, col, px, py, pz, dx, dy, dz, stepScale, numSamples, occ, ns);
}

__device__ __forceinline__ void light_dir_local(const FrameConsts& fc, float& lx, float& ly, float& lz)
{
	lx = dot3(fc.light_pt[0], fc.light_pt[1], fc.light_pt[2], fc.world_i[0], fc.world_i[1], fc.world_i[2]);
	ly = dot3(fc.light_pt[0], fc.light_pt[1], fc.light_pt[2], fc.world_i[4], fc.world_i[5], fc.world_i[6]);
	lz = dot3(fc.light_pt[0], fc.light_pt[1], fc.light_pt[2], fc.world_i[8], fc.world_i[9], fc.world_i[10]);
	const float r = rsqf(dot3(lx, ly, lz, lx, ly, lz));
	lx *= r; ly *= r; lz *= r;
}

// FX_OPT_COUNT_SAMPLES: counters = [64 shards][3] { colour samples of view rays, density samples of light / AO rays, light-map fetches };
// null in every timed launch (the per-thread counts are then dead registers)
__device__ __forceinline__ void flush_counts(unsigned long long* __restrict__ counters, uint32_t view, uint32_t light, uint32_t lm)
{
	if (!counters) return;
	unsigned long long* c = counters + 3 * ((blockIdx.x + blockIdx.y * 7u + blockIdx.z * 13u) & 63u);
	if (view) atomicAdd(c + 0, (unsigned long long)view);
	if (light) atomicAdd(c + 1, (unsigned long long)light);
	if (lm) atomicAdd(c + 2, (unsigned long long)lm);
}

// ---------------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void k_raymarch_light(const Geom g, const typename ColTex<HALF>::T* __restrict__ col,
	uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh, uint32_t numSamples, const float* __restrict__ occ,
	unsigned long long* __restrict__ counters)
{
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	const int z = blockIdx.z;
	if (x >= g.X || y >= g.Y) return;
	const float ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);            // CSRayMarchL.hlsl:22
	const float oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
	const float oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
	const float u = fmaf(ox, 0.5f, 0.5f), v = fmaf(oy, 0.5f, 0.5f), w = fmaf(oz, 0.5f, 0.5f);   // :36
	const float density = sample_density<HALF>(col, make_taps(g, u, v, w), occ);   // :37
	float shadow = 1.0f, ao = 1.0f, irr[3] = { 0.0f, 0.0f, 0.0f };
	uint32_t ns = 0;                                                               // (the voxel's own density sample is counted on the host: X Y Z of them)
	if (density >= 0.00999999978f) {                                               // :44
		const float stepScale = 3.46410155f / (float)numSamples;                   // RayMarch.hlsli:29-30
		float lx, ly, lz;
		light_dir_local(fc, lx, ly, lz);
		cast_light_ray<HALF>(shadow, g, col, ox, oy, oz, lx, ly, lz, stepScale, numSamples, occ, ns);   // :55
		if (sh) gi_term<HALF>(irr, ao, g, col, fc, sh, ox, oy, oz, u, v, w, stepScale, numSamples, occ, ns);   // :59-68
	}
	float out[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float lc = fc.light_color[3] * fc.light_color[a];
		const float amb = sh ? ao * irr[a] : fc.ambient[3] * fc.ambient[a];        // :72-76
		out[a] = fmaf(shadow, lc, amb);                                            // :79
	}
	lightmap[((size_t)z * g.Y + y) * g.X + x] = pack_r11g11b10(out[0], out[1], out[2]);
	flush_counts(counters, 0u, ns, 0u);
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
// direction, tMax = ray parameter at the cube-map target (the direct pixel march has none: FLT_MAX)
template <bool HALF, bool SEPARATE>
__device__ __forceinline__ void march_ray(const Geom& g, const typename ColTex<HALF>::T* __restrict__ col,
	const uint32_t* __restrict__ lightmap, const FrameConsts& fc, const float* __restrict__ sh, const float o[3], const float d[3],
	float tMax, uint32_t numSamples, uint32_t numLightSamples, float& sr, float& sg, float& sb, float& sa, const float* __restrict__ occ,
	uint32_t& nv, uint32_t& nl, uint32_t& nm)
{
	const float stepScale = 3.46410155f / (float)numSamples;
	const float lightStep = 3.46410155f / (float)numLightSamples;
	float lx = 0.0f, ly = 0.0f, lz = 0.0f;
	if (!SEPARATE) light_dir_local(fc, lx, ly, lz);

	sr = 0.0f; sg = 0.0f; sb = 0.0f; sa = 0.0f;
	float t = 0.0f, prev = 0.0f;
	for (uint32_t i = 0; i < numSamples; ++i) {                                    // :146
		const float qx = fmaf(d[0], t, o[0]), qy = fmaf(d[1], t, o[1]), qz = fmaf(d[2], t, o[2]);
		if (outside(qx, qy, qz)) break;                                            // :149
		const float u = fmaf(qx, 0.5f, 0.5f), v = fmaf(qy, 0.5f, 0.5f), w = fmaf(qz, 0.5f, 0.5f);
		const Taps tp = make_taps(g, u, v, w);
		const float4 c = sample_color<HALF>(col, tp, occ);                         // :157
		++nv;
		float newStep = stepScale;
		if (0.00999999978f < c.w) {                                                // :161
			float light[3];
			if (SEPARATE) {                                                        // RayMarch.hlsli:253-258
				const float3 l = sample_light(lightmap, tp);
				++nm;
				light[0] = l.x; light[1] = l.y; light[2] = l.z;
			} else {                                                               // RayMarch.hlsli:260-294
				float shadow = 1.0f, ao = 1.0f, irr[3] = { 0.0f, 0.0f, 0.0f };
				cast_light_ray<HALF>(shadow, g, col, qx, qy, qz, lx, ly, lz, lightStep, numLightSamples, occ, nl);
				if (sh) gi_term<HALF>(irr, ao, g, col, fc, sh, qx, qy, qz, u, v, w, lightStep, numLightSamples, occ, nl);
#pragma unroll
				for (int a = 0; a < 3; ++a) {
					const float amb = sh ? ao * irr[a] : fc.ambient[3] * fc.ambient[a];
					light[a] = fmaf(fc.light_color[3] * fc.light_color[a], shadow, amb);
				}
			}
			const float transm = -sa + 1.0f;                                       // :170
			newStep = step_factor(-prev + c.w, transm, c.w) * stepScale;           // :172
			sr = fmaf(transm * (light[0] * c.x), 0.800000012f, sr);                // :180-181
			sg = fmaf(transm * (light[1] * c.y), 0.800000012f, sg);
			sb = fmaf(transm * (light[2] * c.z), 0.800000012f, sb);
			sa = fmaf(0.800000012f * c.w, transm, sa);
			if (transm < 0.00999999978f) break;                                    // :183
			prev = c.w;
		}
		t = t + newStep;                                                           // :187-188
		if (tMax < t) break;                                                       // :189
	}
}

template <bool HALF, bool SEPARATE>
__global__ __launch_bounds__(64) void k_raymarch_view(const Geom g, const typename ColTex<HALF>::T* __restrict__ col,
	const uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh, int size, uint32_t mask,
	uint32_t numSamples, uint32_t numLightSamples, uint32_t* __restrict__ cube, const float* __restrict__ occ, unsigned long long* __restrict__ counters)
{
	const int face = blockIdx.z;
	if (!((mask >> face) & 1u)) return;                                            // CSRayMarch.hlsl:102
	const int x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
	if (x >= size || y >= size) return;

	float o[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {                                                  // :107
		const float* r = fc.world_i + 4 * a;
		o[a] = fmaf(r[3], 1.0f, fmaf(fc.eye_pt[2], r[2], fmaf(fc.eye_pt[1], r[1], fc.eye_pt[0] * r[0])));
	}
	// GetLocalPos (:39-64)
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
	float d[3] = { -o[0] + tg[0], -o[1] + tg[1], -o[2] + tg[2] };
	const float rl = rsqf(dot3(d[0], d[1], d[2], d[0], d[1], d[2]));               // :115
	d[0] *= rl; d[1] *= rl; d[2] *= rl;
	if (!compute_ray_origin(o, d)) return;                                         // :116
	const float tMax = fmaxf((tg[2] + -o[2]) / d[2], fmaxf((tg[1] + -o[1]) / d[1], (tg[0] + -o[0]) / d[0]));   // :118

	float sr, sg, sb, sa;
	uint32_t nv = 0, nl = 0, nm = 0;
	march_ray<HALF, SEPARATE>(g, col, lightmap, fc, sh, o, d, tMax, numSamples, numLightSamples, sr, sg, sb, sa, occ, nv, nl, nm);
	flush_counts(counters, nv, nl, nm);
	sr *= 0.159154937f; sg *= 0.159154937f; sb *= 0.159154937f;                    // :192
	cube[((size_t)face * size + y) * size + x] =
		to_unorm8(sr) | (to_unorm8(sg) << 8) | (to_unorm8(sb) << 16) | (to_unorm8(sa) << 24);   // :195
}

// ---------------------------------------------------------------------------------------------------
// direct screen-space march (row f-2): PSRayCast.hlsl:44-127 (merged) / PSRayCastV.hlsl (SEPARATE: light-map fetch),
// Fluid::rayCastDirect / rayCastVDirect (Fluid.cpp:932-972).  One thread per pixel, 8x8-pixel tile per wave so that the
// taps of a wave stay spatially coherent; output = the shader's premultiplied SV_TARGET, merged into the RGBA8 target
// with the PREMULTIPLIED blend (Fluid.cpp:670,685) and optionally kept as float4 (parity tests).
// ---------------------------------------------------------------------------------------------------
template <bool HALF, bool SEPARATE>
__global__ __launch_bounds__(64) void k_raycast_direct(const Geom g, const typename ColTex<HALF>::T* __restrict__ col,
	const uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh, int W, int H,
	uint32_t numSamples, uint32_t numLightSamples, uint32_t* __restrict__ target, float4* __restrict__ out_float, const float* __restrict__ occ,
	unsigned long long* __restrict__ counters)
{
	const int px = blockIdx.x * 8 + threadIdx.x, py = blockIdx.y * 8 + threadIdx.y;
	if (px >= W || py >= H) return;
	const size_t pix = (size_t)py * W + px;
	if (out_float) out_float[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	// TexcoordToLocalPos (PSRayCast.hlsl:17-26)
	const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
	const float qx = fmaf(u, 2.0f, -1.0f), qy = fmaf(v, -2.0f, 1.0f);
	const float* M = fc.wvp_i;
	const float h0 = dot3(qx, qy, 1.0f, M[0], M[1], M[3]), h1 = dot3(qx, qy, 1.0f, M[4], M[5], M[7]);
	const float h2 = dot3(qx, qy, 1.0f, M[8], M[9], M[11]), h3 = dot3(qx, qy, 1.0f, M[12], M[13], M[15]);
	float o[3] = { h0 / h3, h1 / h3, h2 / h3 }, d[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {                                                  // :47-49
		const float* r = fc.world_i + 4 * a;
		const float e = fmaf(r[3], 1.0f, fmaf(fc.eye_pt[2], r[2], fmaf(fc.eye_pt[1], r[1], fc.eye_pt[0] * r[0])));
		d[a] = o[a] + -e;
	}
	const float rl = rsqf(dot3(d[0], d[1], d[2], d[0], d[1], d[2]));
	d[0] *= rl; d[1] *= rl; d[2] *= rl;
	if (!compute_ray_origin(o, d)) return;                                         // :50 discard
	float sr, sg, sb, sa;
	uint32_t nv = 0, nl = 0, nm = 0;
	march_ray<HALF, SEPARATE>(g, col, lightmap, fc, sh, o, d, 3.40282347e+38f, numSamples, numLightSamples, sr, sg, sb, sa, occ, nv, nl, nm);
	flush_counts(counters, nv, nl, nm);
	sr *= 0.159154937f; sg *= 0.159154937f; sb *= 0.159154937f;                    // :124
	if (out_float) out_float[pix] = make_float4(sr, sg, sb, sa);
	if (target) {
		const uint32_t dd = target[pix];
		const float ia = 1.0f - sa;
		target[pix] = to_unorm8(fmaf((float)(dd & 255u) / 255.0f, ia, sr)) | (to_unorm8(fmaf((float)((dd >> 8) & 255u) / 255.0f, ia, sg)) << 8)
			| (to_unorm8(fmaf((float)((dd >> 16) & 255u) / 255.0f, ia, sb)) << 16) | (to_unorm8(fmaf((float)(dd >> 24) / 255.0f, ia, sa)) << 24);
	}
}

// 2-D visualiser: PSVisualizeColor.hlsl:24-33 (Fluid::visualizeColor, Fluid.cpp:811-823), PREMULTIPLIED blend
template <bool HALF>
__global__ __launch_bounds__(256) void k_visualize_color(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, int W, int H,
	uint32_t* __restrict__ target, float4* __restrict__ out_float)
{
	const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y * blockDim.y + threadIdx.y;
	if (px >= W || py >= H) return;
	const size_t pix = (size_t)py * W + px;
	const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
	const Taps tp = make_taps(g, fmaf(u, 1.0f, 0.0f), fmaf(v, -1.0f, 1.0f), 0.5f);
	float4 c = sample_color<HALF>(col, tp);
	c.x = c.x / (c.x + 0.5f); c.y = c.y / (c.y + 0.5f); c.z = c.z / (c.z + 0.5f);
	if (out_float) out_float[pix] = c;
	if (target) {
		const uint32_t dd = target[pix];
		const float ia = 1.0f - c.w;
		target[pix] = to_unorm8(fmaf((float)(dd & 255u) / 255.0f, ia, c.x)) | (to_unorm8(fmaf((float)((dd >> 8) & 255u) / 255.0f, ia, c.y)) << 8)
			| (to_unorm8(fmaf((float)((dd >> 16) & 255u) / 255.0f, ia, c.z)) << 16) | (to_unorm8(fmaf((float)(dd >> 24) / 255.0f, ia, c.w)) << 24);
	}
}

__global__ __launch_bounds__(256) void k_lightmap_decode(const uint32_t* __restrict__ lm, float* __restrict__ out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float3 c = unpack_r11g11b10(lm[i]);
		out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
	}
}

// occupancy grid of the ray marches: entry c bounds the alpha of the voxels [4c, 4c + 4] per axis -- everything a trilinear
// sample whose base tap lies in block c can touch.  Two passes: k_occupancy_blocks reads every alpha once, coalesced along
// x (lane = x, each thread folds a 1 x 4 x 4 column, four lanes fold into one 4^3 block: no atomics), k_occupancy_dilate
// takes the max over the 2 x 2 x 2 blocks c .. c + 1 (a superset of [4c, 4c + 4]: conservative, which only skips less).
template <bool HALF>
__global__ __launch_bounds__(256) void k_occupancy_blocks(const Geom g, const typename ColTex<HALF>::T* __restrict__ col, float* __restrict__ blk)
{
	const int CX = (g.X + 3) >> 2, CY = (g.Y + 3) >> 2;
	const int x = blockIdx.x * 64 + (threadIdx.x & 63);
	const int cy = blockIdx.y * 4 + (threadIdx.x >> 6), cz = blockIdx.z;
	float m = 0.0f;
	if (x < g.X && cy < CY) {
		for (int z = 4 * cz; z < min(4 * cz + 4, g.Zg); ++z)
			for (int y = 4 * cy; y < min(4 * cy + 4, g.Y); ++y)
				m = fmaxf(m, ColTex<HALF>::ldw(col, ((size_t)z * g.Y + y) * g.X + x));
	}
	m = fmaxf(m, __shfl_xor(m, 1));
	m = fmaxf(m, __shfl_xor(m, 2));
	if (x < g.X && cy < CY && (x & 3) == 0) blk[((size_t)cz * CY + cy) * CX + (x >> 2)] = m;
}

__global__ __launch_bounds__(256) void k_occupancy_dilate(int CX, int CY, int CZ, const float* __restrict__ blk, float* __restrict__ occ)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= CX * CY * CZ) return;
	const int cx = c % CX, cy = (c / CX) % CY, cz = c / (CX * CY);
	const int x1 = min(cx + 1, CX - 1), y1 = min(cy + 1, CY - 1), z1 = min(cz + 1, CZ - 1);
	float m = 0.0f;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int xx = (k & 1) ? x1 : cx, yy = (k & 2) ? y1 : cy, zz = (k & 4) ? z1 : cz;
		m = fmaxf(m, blk[((size_t)zz * CY + yy) * CX + xx]);
	}
	occ[c] = m;
}

// occ: CX*CY*CZ floats, scratch: as many again
hipError_t launch_occupancy(const Geom& g, int half_store, const void* color, float* occ, hipStream_t s)
{
	const int CX = (g.X + 3) >> 2, CY = (g.Y + 3) >> 2, CZ = (g.Zg + 3) >> 2;
	const int n = CX * CY * CZ;
	float* blk = occ + n;
	const dim3 grid((g.X + 63) / 64, (CY + 3) / 4, CZ), block(256);
	if (half_store) hipLaunchKernelGGL(k_occupancy_blocks<true>, grid, block, 0, s, g, (const h16x4*)color, blk);
	else hipLaunchKernelGGL(k_occupancy_blocks<false>, grid, block, 0, s, g, (const float4*)color, blk);
	hipLaunchKernelGGL(k_occupancy_dilate, dim3((n + 255) / 256), dim3(256), 0, s, CX, CY, CZ, blk, occ);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
hipError_t launch_raymarch_light(const Geom& g, int half_store, const void* color, uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, uint32_t num_samples, const float* occ, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((g.X + 63) / 64, (g.Y + 3) / 4, g.Zg), block(64, 4, 1);
	if (half_store) hipLaunchKernelGGL(k_raymarch_light<true>, grid, block, 0, s, g, (const h16x4*)color, lightmap, fc, sh, num_samples, occ, counters);
	else hipLaunchKernelGGL(k_raymarch_light<false>, grid, block, 0, s, g, (const float4*)color, lightmap, fc, sh, num_samples, occ, counters);
	return hipGetLastError();
}

hipError_t launch_raymarch_view(const Geom& g, int half_store, const void* color, const uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, int cube_size, uint32_t mask, uint32_t num_samples,
	uint32_t num_light_samples, int separate, uint8_t* cube, const float* occ, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((cube_size + 7) / 8, (cube_size + 7) / 8, 6), block(8, 8, 1);
	uint32_t* out = reinterpret_cast<uint32_t*>(cube);
#define FX_LAUNCH(H, S) hipLaunchKernelGGL((k_raymarch_view<H, S>), grid, block, 0, s, g, \
	(const typename ColTex<H>::T*)color, lightmap, fc, sh, cube_size, mask, num_samples, num_light_samples, out, occ, counters)
	if (half_store) { if (separate) FX_LAUNCH(true, true); else FX_LAUNCH(true, false); }
	else { if (separate) FX_LAUNCH(false, true); else FX_LAUNCH(false, false); }
#undef FX_LAUNCH
	return hipGetLastError();
}

hipError_t launch_raycast_direct(const Geom& g, int half_store, const void* color, const uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, int W, int H, uint32_t num_samples, uint32_t num_light_samples, int separate,
	uint8_t* target, float* out_float, const float* occ, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((W + 7) / 8, (H + 7) / 8, 1), block(8, 8, 1);
#define FX_LAUNCH(HF, S) hipLaunchKernelGGL((k_raycast_direct<HF, S>), grid, block, 0, s, g, \
	(const typename ColTex<HF>::T*)color, lightmap, fc, sh, W, H, num_samples, num_light_samples, \
	reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float), occ, counters)
	if (half_store) { if (separate) FX_LAUNCH(true, true); else FX_LAUNCH(true, false); }
	else { if (separate) FX_LAUNCH(false, true); else FX_LAUNCH(false, false); }
#undef FX_LAUNCH
	return hipGetLastError();
}

hipError_t launch_visualize_color(const Geom& g, int half_store, const void* color, int W, int H, uint8_t* target, float* out_float, hipStream_t s)
{
	const dim3 grid((W + 63) / 64, (H + 3) / 4, 1), block(64, 4, 1);
	if (half_store) hipLaunchKernelGGL(k_visualize_color<true>, grid, block, 0, s, g, (const h16x4*)color, W, H,
		reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float));
	else hipLaunchKernelGGL(k_visualize_color<false>, grid, block, 0, s, g, (const float4*)color, W, H,
		reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float));
	return hipGetLastError();
}

hipError_t launch_lightmap_decode(const uint32_t* lightmap, float* out, size_t n, hipStream_t s)
{
	if (!n) return hipSuccess;
	const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
	hipLaunchKernelGGL(k_lightmap_decode, dim3(grid), dim3(256), 0, s, lightmap, out, n);
	return hipGetLastError();
}

}  // namespace fx
