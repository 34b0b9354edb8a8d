// oracle/orc_render.cpp -- CPU oracle for the cube-map-space ray march (TEST INFRASTRUCTURE ONLY,
// see orc_common.h).  Restates, in scalar C++:
//   CSRayMarchL::main   /root/reference/FluidX12/Content/Shaders/CSRayMarchL.hlsl:15-80
//   CSRayMarch::main    CSRayMarch.hlsl:98-196  (merged: nested light march)
//   CSRayMarchV         CSRayMarchV.hlsl:5-7 = CSRayMarch with _LIGHT_PASS_ (light-map fetch)
//   helpers             RayMarch.hlsli:62-68 GetSample, :73-95 GetDensityGradient, :146-173
//                       ComputeRayOrigin, :178-183 ComputeTargetHit, :188-195 LocalToTex3DSpace,
//                       :200-210 GetStep, :215-247 CastLightRay, :253-295 GetLight;
//                       CSRayMarch.hlsl:39-64 GetLocalPos;
//                       XUSG/Shaders/SHIrradianceTypeless.hlsli:16-37 EvaluateSHIrradiance
// following the operation order of the shipped Bin/CSRayMarch{,L,V}.cso (tools/dxbc.py).
// min16float is a precision *hint* in SM5.0 bytecode; everything is fp32 (the .cso carry no
// reduced-precision arithmetic).  Sampler = LINEAR_CLAMP (Fluid.cpp:475).
//
// Layouts: colour float[Z][Y][X][4]; light map float[Z][Y][X][3] (values already rounded to the
// selected storage format); cube map: float[6][S][S][4] (pre-quantisation) + uint8[6][S][S][4]
// (R8G8B8A8_UNORM, Fluid.cpp:231).  Texels of culled faces / missed rays are left untouched
// (the reference `return`s without writing, CSRayMarch.hlsl:102,116).
#include "orc_common.h"
#include "fx_oracle.h"

using namespace orc;

namespace {

// ---- R11G11B10_FLOAT (light map, Fluid.cpp:226), round-to-nearest-even ----------------------
uint32_t pack_ufloat(float f, int mbits)
{
	const uint32_t x = f2bits(f);
	const uint32_t maxfinite = (31u << mbits) - 1u;
	if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (31u << mbits) | 1u;       // NaN
	if (x & 0x80000000u) return 0u;                                         // negative -> 0
	if (x == 0x7F800000u) return 31u << mbits;                              // +inf
	const int drop = 23 - mbits;
	if (x < 0x38800000u) {                                                  // < 2^-14: subnormal
		const float q = std::nearbyintf(f * std::ldexp(1.0f, 14 + mbits)); // RNE
		return (uint32_t)q;
	}
	uint32_t v = x - (112u << 23);                                          // rebias 127 -> 15
	v += ((1u << (drop - 1)) - 1u) + ((v >> drop) & 1u);
	v >>= drop;
	return v > maxfinite ? maxfinite : v;
}

float unpack_ufloat(uint32_t b, int mbits)
{
	const uint32_t e = b >> mbits, m = b & ((1u << mbits) - 1u);
	if (e == 0) return std::ldexp((float)m, -14 - mbits);
	if (e == 31) return m ? NAN : INFINITY;
	return std::ldexp(1.0f + std::ldexp((float)m, -mbits), (int)e - 15);
}

inline float store_light(float v, int fmt, int chan)
{
	switch (fmt) {
	case 1: return quant_half(v);
	case 2: { const int mb = chan == 2 ? 5 : 6; return unpack_ufloat(pack_ufloat(v, mb), mb); }
	default: return v;
	}
}

struct Vol { const float* color; const float* light; int dims[3]; };

inline bool outside(const float p[3]) { return std::fabs(p[0]) > 1.0f || std::fabs(p[1]) > 1.0f || std::fabs(p[2]) > 1.0f; }

// GetStep (RayMarch.hlsli:200-210) as compiled: max(((min(1/256/|dD|, 2) * 1.5) * min(1-rho,1)) * (1-transm), 1)
inline float step_factor(float dDensity, float transm, float density)
{
	const float ev = std::fmin(0.00390625f / std::fabs(dDensity), 2.0f);   // x/0 = inf -> 2
	const float ui = std::fmin(-density + 1.0f, 1.0f);
	const float th = -transm + 1.0f;
	return std::fmax(th * (ui * (ev * 1.5f)), 1.0f);
}

// CastLightRay (RayMarch.hlsli:215-247), mipLevel = 0.  The dead averaging at :236 is dropped, the
// position update is the compiled t = fma(stepScale, factor, t).
void cast_light_ray(float& transm, const Vol& v, const float origin[3], const float dir[3], float stepScale, uint32_t numSamples)
{
	float t = stepScale, prevDensity = 0.0f;
	for (uint32_t i = 0; i < numSamples; ++i) {
		float pos[3], uvw[3];
		for (int a = 0; a < 3; ++a) pos[a] = std::fmaf(dir[a], t, origin[a]);
		if (outside(pos)) break;
		for (int a = 0; a < 3; ++a) uvw[a] = std::fmaf(pos[a], 0.5f, 0.5f);
		const Taps tp = make_taps(uvw, v.dims, ADDR_CLAMP);
		const float density = sample_chan(v.color, 4, 3, v.dims, tp);
		const float nt = std::fmaf(-density, 0.800000012f, 1.0f) * transm;
		if (nt < 0.00999999978f) { transm = nt; break; }
		const float fac = step_factor(-prevDensity + density, transm, density);   // old transm (:235 precedes :240)
		t = std::fmaf(stepScale, fac, t);
		transm = nt;
		prevDensity = density;
	}
}

// GetDensityGradient (RayMarch.hlsli:73-95): integer texel offsets on a trilinear fetch
void density_gradient(float out[3], const Vol& v, const float uvw[3])
{
	static const int offs[6][3] = { {-1,0,0},{1,0,0},{0,-1,0},{0,1,0},{0,0,-1},{0,0,1} };
	float q[6];
	for (int i = 0; i < 6; ++i) {
		const Taps tp = make_taps(uvw, v.dims, ADDR_CLAMP, offs[i]);
		q[i] = sample_chan(v.color, 4, 3, v.dims, tp);
	}
	out[0] = -q[0] + q[1]; out[1] = -q[2] + q[3]; out[2] = -q[4] + q[5];
}

inline void normalize3(float v[3])
{
	const float r = 1.0f / std::sqrt(dp3(v, v));     // DXBC rsq
	v[0] *= r; v[1] *= r; v[2] *= r;
}

// EvaluateSHIrradiance (SHIrradianceTypeless.hlsli:16-37) in the compiled association order.
// sh: 9 x float3.  n = normalised direction (already in world space).
void sh_irradiance(float out[3], const float* sh, const float n[3])
{
	const float c1 = 0.429042757f, c3 = 0.247707963f, c4 = 0.886226952f, c1x2 = 0.858085513f, c2x2 = 1.02332675f;
	const float yy = n[1] * n[1], zz = n[2] * n[2];
	const float a = std::fmaf(n[0], n[0], -yy) * c1;            // c1 (x^2 - y^2)
	const float b = std::fmaf(zz, 3.0f, -1.0f) * c3;            // c3 (3 z^2 - 1)
	const float mx = -n[0], my = -n[1], z = n[2];                // x = -n.x, y = -n.y (:23-25)
	for (int k = 0; k < 3; ++k) {
		const float* L = sh + k;                                 // L[i*3]
		float r = L[6 * 3] * b;
		r = std::fmaf(a, L[8 * 3], r);
		r = std::fmaf(L[0], c4, r);
		float q = (L[7 * 3] * mx) * z;                           // L21 x z
		q = std::fmaf(L[4 * 3] * mx, my, q);                     // + L2-2 x y
		q = std::fmaf(L[5 * 3] * my, z, q);                      // + L2-1 y z
		r = std::fmaf(q, c1x2, r);
		float l = L[1 * 3] * my;                                 // L1-1 y
		l = std::fmaf(L[3 * 3], mx, l);                          // + L11 x
		l = std::fmaf(L[2 * 3], z, l);                           // + L10 z
		r = std::fmaf(l, c2x2, r);
		out[k] = std::fmax(r, 0.0f);
	}
}

// the GI branch shared by CSRayMarchL.hlsl:59-68 and RayMarch.hlsli:275-283
void gi_term(float irradiance[3], float& ao, const Vol& v, const orc_frame* fc, const float pos[3], const float uvw[3],
	float stepScale, uint32_t numSamples)
{
	float grad[3], dir[3];
	density_gradient(grad, v, uvw);
	const bool any = std::fabs(grad[0]) > 0.0f || std::fabs(grad[1]) > 0.0f || std::fabs(grad[2]) > 0.0f;
	for (int a = 0; a < 3; ++a) dir[a] = any ? -grad[a] : pos[a];
	float wdir[3];
	for (int a = 0; a < 3; ++a) wdir[a] = dp3(dir, fc->world + 4 * a);     // mul(rayDir, (float3x3)g_world)
	normalize3(wdir);
	sh_irradiance(irradiance, fc->sh, wdir);
	normalize3(dir);
	ao = 1.0f;
	cast_light_ray(ao, v, pos, dir, stepScale, numSamples);
}

inline void light_dir_local(float out[3], const orc_frame* fc)
{
	for (int a = 0; a < 3; ++a) out[a] = dp3(fc->light_pt, fc->world_i + 4 * a);   // mul(g_lightPt, (float3x3)g_worldI)
	normalize3(out);
}

// GetLocalPos (CSRayMarch.hlsl:39-64)
void cube_texel_to_local(float out[3], int x, int y, int face, int size)
{
	const float px = std::fmaf(((float)x + 0.5f) / (float)size, 2.0f, -1.0f);
	const float py = -std::fmaf(((float)y + 0.5f) / (float)size, 2.0f, -1.0f);
	switch (face) {
	case 0: out[0] = 1.0f;  out[1] = py;   out[2] = -px;  break;
	case 1: out[0] = -1.0f; out[1] = py;   out[2] = px;   break;
	case 2: out[0] = px;    out[1] = 1.0f; out[2] = -py;  break;
	case 3: out[0] = px;    out[1] = -1.0f; out[2] = py;  break;
	case 4: out[0] = px;    out[1] = py;   out[2] = 1.0f; break;
	default: out[0] = -px;  out[1] = py;   out[2] = -1.0f; break;
	}
}

// ComputeRayOrigin (RayMarch.hlsli:146-173)
bool compute_ray_origin(float o[3], const float d[3])
{
	if (std::fabs(o[0]) <= 1.0f && std::fabs(o[1]) <= 1.0f && std::fabs(o[2]) <= 1.0f) return true;
	float U = 3.40282347e+38f;
	bool hit = false;
	for (int i = 0; i < 3; ++i) {
		const float sgn = (float)((0.0f < d[i]) - (d[i] < 0.0f));
		const float u = (-o[i] + -sgn) / d[i];                      // (-sign(dir) - origin) / dir
		if (!(u >= 0.0f)) continue;
		const int j = (i + 1) % 3, k = (i + 2) % 3;
		if (!(1.0f >= std::fabs(std::fmaf(d[j], u, o[j])))) continue;
		if (!(1.0f >= std::fabs(std::fmaf(d[k], u, o[k])))) continue;
		if (u < U) { U = u; hit = true; }
	}
	for (int a = 0; a < 3; ++a) o[a] = std::fmin(std::fmax(std::fmaf(d[a], U, o[a]), -1.0f), 1.0f);
	return hit;
}

// the march of one view ray (CSRayMarch.hlsl:140-190 == PSRayCast.hlsl:72-122): o = ray origin on/in the cube, d = unit
// direction, tMax = parameter at the cube-map target (the direct pixel-shader march has none: pass FLT_MAX)
void march(float scatter[4], const Vol& v, const orc_frame* fc, const float o[3], const float d[3], float tMax,
	const float ldir[3], const float lightColor[3], const float ambient[3], float stepScale, float lightStep,
	uint32_t numSamples, uint32_t numLightSamples, int hasSH, int separate)
{
	scatter[0] = scatter[1] = scatter[2] = scatter[3] = 0.0f;
	float t = 0.0f, prevDensity = 0.0f;
	for (uint32_t i = 0; i < numSamples; ++i) {            // :146
		float pos[3], uvw[3];
		for (int a = 0; a < 3; ++a) pos[a] = std::fmaf(d[a], t, o[a]);
		if (outside(pos)) break;                          // :149
		for (int a = 0; a < 3; ++a) uvw[a] = std::fmaf(pos[a], 0.5f, 0.5f);
		const Taps tp = make_taps(uvw, v.dims, ADDR_CLAMP);
		float c[4];
		for (int a = 0; a < 4; ++a) c[a] = sample_chan(v.color, 4, a, v.dims, tp);   // :157
		float newStep = stepScale;
		if (0.00999999978f < c[3]) {                      // :161
			float light[3];
			if (separate) {
				for (int a = 0; a < 3; ++a) light[a] = sample_chan(v.light, 3, a, v.dims, tp);
			} else {
				float shadow = 1.0f, ao = 1.0f, irr[3] = { 0, 0, 0 };
				cast_light_ray(shadow, v, pos, ldir, lightStep, numLightSamples);        // RayMarch.hlsli:270
				if (hasSH) gi_term(irr, ao, v, fc, pos, uvw, lightStep, numLightSamples);   // :275-283
				for (int a = 0; a < 3; ++a) {
					const float amb = hasSH ? ao * irr[a] : ambient[a];
					light[a] = std::fmaf(lightColor[a], shadow, amb);                   // :293
				}
			}
			const float transm = -scatter[3] + 1.0f;                                    // :170
			newStep = step_factor(-prevDensity + c[3], transm, c[3]) * stepScale;       // :172
			for (int a = 0; a < 3; ++a)
				scatter[a] = std::fmaf(transm * (light[a] * c[a]), 0.800000012f, scatter[a]);   // :180-181
			scatter[3] = std::fmaf(0.800000012f * c[3], transm, scatter[3]);
			if (transm < 0.00999999978f) break;           // :183
			prevDensity = c[3];                           // :174
		}
		t = t + newStep;                                  // :187-188
		if (tMax < t) break;                              // :189
	}
}

inline uint8_t to_unorm8(float v)
{
	if (!(v > 0.0f)) return 0;                     // NaN, negatives
	if (v >= 1.0f) return 255;
	return (uint8_t)(v * 255.0f + 0.5f);           // D3D FLOAT -> UNORM: scale, +0.5, truncate
}

}  // namespace

extern "C" {

uint32_t orc_pack_r11g11b10(float r, float g, float b)
{
	return pack_ufloat(r, 6) | (pack_ufloat(g, 6) << 11) | (pack_ufloat(b, 5) << 22);
}

void orc_unpack_r11g11b10(uint32_t v, float* rgb)
{
	rgb[0] = unpack_ufloat(v & 0x7FFu, 6);
	rgb[1] = unpack_ufloat((v >> 11) & 0x7FFu, 6);
	rgb[2] = unpack_ufloat(v >> 22, 5);
}

// ---------------------------------------------------------------------------------------------
// CSRayMarchL.hlsl:15-80.  g_numSamples := maxLightSamples, step = g_step (Fluid.cpp:872-873).
// light_fmt: 0 fp32, 1 half, 2 R11G11B10F.
// ---------------------------------------------------------------------------------------------
void orc_raymarch_light(const float* color, float* lightmap, int X, int Y, int Z, const orc_frame* fc,
	uint32_t numSamples, int hasSH, int light_fmt)
{
	const Vol v{ color, nullptr, { X, Y, Z } };
	const float fdims[3] = { (float)X, (float)Y, (float)Z };
	float ldir[3];
	light_dir_local(ldir, fc);
	float lightColor[3], ambient[3];
	for (int a = 0; a < 3; ++a) { lightColor[a] = fc->light_color[3] * fc->light_color[a]; ambient[a] = fc->ambient[3] * fc->ambient[a]; }
	const float stepScale = 3.46410155f / (float)numSamples;         // RayMarch.hlsli:29-30

#pragma omp parallel for schedule(dynamic, 1) collapse(2)
	for (int z = 0; z < Z; ++z)
		for (int y = 0; y < Y; ++y)
			for (int x = 0; x < X; ++x) {
				const int cell[3] = { x, y, z };
				float o[3], uvw[3];
				for (int a = 0; a < 3; ++a) {
					o[a] = std::fmaf(((float)cell[a] + 0.5f) / fdims[a], 2.0f, -1.0f);     // :22
					uvw[a] = std::fmaf(o[a], 0.5f, 0.5f);                                    // :36
				}
				const Taps tp = make_taps(uvw, v.dims, ADDR_CLAMP);
				const float density = sample_chan(color, 4, 3, v.dims, tp);                // :37
				float shadow = 1.0f, ao = 1.0f, irr[3] = { 0.0f, 0.0f, 0.0f };
				if (density >= 0.00999999978f) {                                            // :44
					cast_light_ray(shadow, v, o, ldir, stepScale, numSamples);              // :55
					if (hasSH) gi_term(irr, ao, v, fc, o, uvw, stepScale, numSamples);      // :59-68
				}
				float* out = lightmap + (((size_t)z * Y + y) * X + x) * 3;
				for (int a = 0; a < 3; ++a) {
					const float amb = hasSH ? ao * irr[a] : ambient[a];                    // :76
					out[a] = store_light(std::fmaf(shadow, lightColor[a], amb), light_fmt, a);   // :79
				}
			}
}

// ---------------------------------------------------------------------------------------------
// CSRayMarch.hlsl:98-196.  separate != 0: CSRayMarchV (light = light-map fetch, RayMarch.hlsli:
// 253-258); separate == 0: merged (nested CastLightRay + optional GI, RayMarch.hlsli:260-294).
// size = cube-map edge at the bound mip (X >> LOD).
// ---------------------------------------------------------------------------------------------
void orc_raymarch_view(const float* color, const float* lightmap, int X, int Y, int Z, const orc_frame* fc,
	int size, uint32_t mask, uint32_t numSamples, uint32_t numLightSamples, int hasSH, int separate,
	float* cube_f32, uint8_t* cube_u8)
{
	const Vol v{ color, lightmap, { X, Y, Z } };
	float eye[3];
	for (int a = 0; a < 3; ++a) {                                     // mul(float4(g_eyePt,1), g_worldI)  :107 (dp4)
		const float* r = fc->world_i + 4 * a;
		eye[a] = std::fmaf(r[3], 1.0f, std::fmaf(fc->eye_pt[2], r[2], std::fmaf(fc->eye_pt[1], r[1], fc->eye_pt[0] * r[0])));
	}
	float ldir[3];
	light_dir_local(ldir, fc);
	float lightColor[3], ambient[3];
	for (int a = 0; a < 3; ++a) { lightColor[a] = fc->light_color[3] * fc->light_color[a]; ambient[a] = fc->ambient[3] * fc->ambient[a]; }
	const float stepScale = 3.46410155f / (float)numSamples;
	const float lightStep = 3.46410155f / (float)numLightSamples;

#pragma omp parallel for schedule(dynamic, 1) collapse(2)
	for (int face = 0; face < 6; ++face)
		for (int y = 0; y < size; ++y) {
			if (!(mask >> face & 1u)) continue;                       // :102
			for (int x = 0; x < size; ++x) {
				float target[3], o[3] = { eye[0], eye[1], eye[2] }, d[3];
				cube_texel_to_local(target, x, y, face, size);        // :114
				for (int a = 0; a < 3; ++a) d[a] = -o[a] + target[a];
				normalize3(d);                                        // :115
				if (!compute_ray_origin(o, d)) continue;              // :116
				float tq[3];
				for (int a = 0; a < 3; ++a) tq[a] = (target[a] + -o[a]) / d[a];   // ComputeTargetHit :178-183
				const float tMax = std::fmax(tq[2], std::fmax(tq[1], tq[0]));

				float scatter[4];
				march(scatter, v, fc, o, d, tMax, ldir, lightColor, ambient, stepScale, lightStep, numSamples, numLightSamples, hasSH, separate);
				const size_t o4 = (((size_t)face * size + y) * size + x) * 4;
				for (int a = 0; a < 3; ++a) scatter[a] *= 0.159154937f;                    // :192
				for (int a = 0; a < 4; ++a) {
					if (cube_f32) cube_f32[o4 + a] = scatter[a];
					if (cube_u8) cube_u8[o4 + a] = to_unorm8(scatter[a]);                  // :195
				}
			}
		}
}

// ---------------------------------------------------------------------------------------------
// PSRayCast.hlsl:44-127 (separate == 0: nested light march, cb2 = {maxRaySamples, hasSH, maxLightSamples},
// Fluid.cpp:932-951) and PSRayCastV.hlsl (separate != 0: light-map fetch, cb2 = {raySampleCount}, Fluid.cpp:953-972):
// the direct screen-space march, one ray per pixel from the near plane through the volume; the pixel's UV is the
// screen-quad interpolant (px + .5) / W.  out_rgba float[H][W][4] premultiplied, zeros + covered = 0 where discarded.
// ---------------------------------------------------------------------------------------------
void orc_raycast_direct(const float* color, const float* lightmap, int X, int Y, int Z, const orc_frame* fc,
	const float* wvp_i, int W, int H, uint32_t numSamples, uint32_t numLightSamples, int hasSH, int separate,
	float* out_rgba, uint8_t* covered)
{
	const Vol v{ color, lightmap, { X, Y, Z } };
	float eye[3];
	for (int a = 0; a < 3; ++a) {
		const float* r = fc->world_i + 4 * a;
		eye[a] = std::fmaf(r[3], 1.0f, std::fmaf(fc->eye_pt[2], r[2], std::fmaf(fc->eye_pt[1], r[1], fc->eye_pt[0] * r[0])));
	}
	float ldir[3];
	light_dir_local(ldir, fc);
	float lightColor[3], ambient[3];
	for (int a = 0; a < 3; ++a) { lightColor[a] = fc->light_color[3] * fc->light_color[a]; ambient[a] = fc->ambient[3] * fc->ambient[a]; }
	const float stepScale = 3.46410155f / (float)numSamples;
	const float lightStep = 3.46410155f / (float)numLightSamples;
#pragma omp parallel for schedule(dynamic, 1)
	for (int py = 0; py < H; ++py)
		for (int px = 0; px < W; ++px) {
			float* out = out_rgba + ((size_t)py * W + px) * 4;
			out[0] = out[1] = out[2] = out[3] = 0.0f;
			covered[(size_t)py * W + px] = 0;
			const float u = ((float)px + 0.5f) / (float)W, vv = ((float)py + 0.5f) / (float)H;
			const float q[3] = { std::fmaf(u, 2.0f, -1.0f), std::fmaf(vv, -2.0f, 1.0f), 1.0f };   // TexcoordToLocalPos :17-26
			float h[4];
			for (int r = 0; r < 4; ++r) {
				const float col[3] = { wvp_i[4 * r + 0], wvp_i[4 * r + 1], wvp_i[4 * r + 3] };
				h[r] = dp3(q, col);
			}
			float o[3] = { h[0] / h[3], h[1] / h[3], h[2] / h[3] }, d[3];
			for (int a = 0; a < 3; ++a) d[a] = o[a] + -eye[a];
			normalize3(d);                                                     // :49
			if (!compute_ray_origin(o, d)) continue;                          // :50 discard
			float scatter[4];
			march(scatter, v, fc, o, d, 3.40282347e+38f, ldir, lightColor, ambient, stepScale, lightStep, numSamples,
				numLightSamples, hasSH, separate);
			for (int a = 0; a < 3; ++a) out[a] = scatter[a] * 0.159154937f;   // :124
			out[3] = scatter[3];
			covered[(size_t)py * W + px] = 1;
		}
}

// ---------------------------------------------------------------------------------------------
// PSVisualizeColor.hlsl:24-33 (Fluid::visualizeColor, Fluid.cpp:811-823): the 2-D grid on the screen.  uvw = (u, 1 - v, .5),
// linear CLAMP fetch of colour[parity], rgb / (rgb + .5); premultiplied SV_TARGET float[H][W][4].
// ---------------------------------------------------------------------------------------------
void orc_visualize_color(const float* color, int X, int Y, int W, int H, float* out_rgba)
{
	const int dims[3] = { X, Y, 1 };
#pragma omp parallel for schedule(static)
	for (int py = 0; py < H; ++py)
		for (int px = 0; px < W; ++px) {
			const float u = ((float)px + 0.5f) / (float)W, vv = ((float)py + 0.5f) / (float)H;
			const float uvw[3] = { std::fmaf(u, 1.0f, 0.0f), std::fmaf(vv, -1.0f, 1.0f), 0.5f };
			const Taps tp = make_taps(uvw, dims, ADDR_CLAMP);
			float* o = out_rgba + ((size_t)py * W + px) * 4;
			for (int a = 0; a < 4; ++a) o[a] = sample_chan(color, 4, a, dims, tp);
			for (int a = 0; a < 3; ++a) o[a] = o[a] / (o[a] + 0.5f);
		}
}

}  // extern "C"
