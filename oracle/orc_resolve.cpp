// orc_resolve.cpp -- TEST INFRASTRUCTURE (CPU oracle), never linked into the product.
//
// Cube map -> screen resolve, the raster-free formulation of the reference's cube pass:
//   /root/reference/FluidX12/Content/Shaders/PSRayCastCube.hlsl:20-113  (TexcoordToLocalPos, ComputeRayHit,
//       ComputeCubeTexcoord, main)
//   /root/reference/FluidX12/Content/Shaders/PSCube.hlsli:41-122        (GetDomain, CubeCast; _USE_PURE_ARRAY_ = 0:
//       the shipped Bin/PSRayCastCube.cso samples a TextureCube)
// in the operation order of the shipped DXBC (mad == fmaf, dp3/dp4 = mul then fma chain, rsq = 1/sqrt), plus the
// fixed-function pieces the shader leans on, restated from the D3D11 functional spec:
//   * TextureCube addressing: major axis with ties Z > Y > X; (sc, tc) per face table; u = 0.5 * (sc / |ma|) + 0.5
//   * gather4: the 2x2 bilinear footprint at floor(u * N - 0.5), returned as x = (i0, j1), y = (i1, j1),
//     z = (i1, j0), w = (i0, j0); a texel that falls off ONE edge is the adjacent face's edge texel at the same position
//     along the edge (seamless cube filtering); a texel off a corner (no face there) is the mean of the other three
//   * RGBA8_UNORM texels read as byte / 255; PREMULTIPLIED blend (ONE, INV_SRC_ALPHA) into an R8G8B8A8_UNORM target
//     (Fluid.cpp:653, FluidX12.cpp:31), FLOAT -> UNORM as floor(255 * sat(x) + 0.5)
// The pixel's UV is the screen-quad interpolant at the pixel centre, ((px + .5) / W, (py + .5) / H) (VSScreenQuad.hlsl:17-26).
#include "orc_common.h"
#include "fx_oracle.h"

namespace {
using namespace orc;

struct CubeTex { const uint8_t* t; int N; };

// point on face f at face coordinates (sc, tc) in [-1, 1] (D3D cube face table)
inline void face_point(float p[3], int f, float sc, float tc)
{
	switch (f) {
	case 0: p[0] = 1.0f;  p[1] = -tc; p[2] = -sc; break;
	case 1: p[0] = -1.0f; p[1] = -tc; p[2] = sc;  break;
	case 2: p[0] = sc;  p[1] = 1.0f;  p[2] = tc;  break;
	case 3: p[0] = sc;  p[1] = -1.0f; p[2] = -tc; break;
	case 4: p[0] = sc;  p[1] = -tc; p[2] = 1.0f;  break;
	default: p[0] = -sc; p[1] = -tc; p[2] = -1.0f; break;
	}
}

// (sc, tc) of a direction/point on face f, NOT divided by the major axis
inline void face_coords(const float p[3], int f, float& sc, float& tc)
{
	switch (f) {
	case 0: sc = -p[2]; tc = -p[1]; break;
	case 1: sc = p[2];  tc = -p[1]; break;
	case 2: sc = p[0];  tc = p[2];  break;
	case 3: sc = p[0];  tc = -p[2]; break;
	case 4: sc = p[0];  tc = -p[1]; break;
	default: sc = -p[0]; tc = -p[1]; break;
	}
}

inline int major_face(const float d[3])
{
	const float ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
	if (az >= ax && az >= ay) return d[2] < 0.0f ? 5 : 4;          // ties: Z, then Y, then X
	if (ay >= ax) return d[1] < 0.0f ? 3 : 2;
	return d[0] < 0.0f ? 1 : 0;
}

inline void texel(const CubeTex& c, int f, int i, int j, float out[4])
{
	const uint8_t* q = c.t + (((size_t)f * c.N + j) * c.N + i) * 4;
	for (int k = 0; k < 4; ++k) out[k] = (float)q[k] / 255.0f;
}

inline float texel_centre(int i, int N) { return (2.0f * (float)i + 1.0f) / (float)N - 1.0f; }

// texel (i, j) of face f where exactly one of i, j is off the face: the adjacent face's edge texel
inline void texel_across_edge(const CubeTex& c, int f, int i, int j, float out[4])
{
	const int N = c.N;
	const float sc = i < 0 ? -1.0f : i >= N ? 1.0f : texel_centre(i, N);
	const float tc = j < 0 ? -1.0f : j >= N ? 1.0f : texel_centre(j, N);
	float P[3];
	face_point(P, f, sc, tc);                       // on the shared edge, at the texel's position along it
	const int fa = f >> 1;
	int g = -1;
	for (int a = 0; a < 3; ++a)
		if (a != fa && std::fabs(P[a]) == 1.0f) g = 2 * a + (P[a] < 0.0f ? 1 : 0);
	float s2, t2;
	face_coords(P, g, s2, t2);
	const int i2 = std::min(std::max((int)std::floor((0.5f * s2 + 0.5f) * (float)N), 0), N - 1);
	const int j2 = std::min(std::max((int)std::floor((0.5f * t2 + 0.5f) * (float)N), 0), N - 1);
	texel(c, g, i2, j2, out);
}

// the four texels of the bilinear footprint in gather order x (i0, j1), y (i1, j1), z (i1, j0), w (i0, j0)
void footprint(const CubeTex& c, const float d[3], float s[4][4], float& fu, float& fv)
{
	const int N = c.N;
	const int f = major_face(d);
	float sc, tc;
	face_coords(d, f, sc, tc);
	const float ma = std::fabs(d[f >> 1]);
	const float tu = std::fmaf(0.5f * (sc / ma) + 0.5f, (float)N, -0.5f);
	const float tv = std::fmaf(0.5f * (tc / ma) + 0.5f, (float)N, -0.5f);
	const float flu = std::floor(tu), flv = std::floor(tv);
	fu = tu - flu; fv = tv - flv;
	const int i0 = (int)flu, j0 = (int)flv;
	const int ii[4] = { i0, i0 + 1, i0 + 1, i0 }, jj[4] = { j0 + 1, j0 + 1, j0, j0 };
	int missing = -1;
	for (int k = 0; k < 4; ++k) {
		const bool oi = ii[k] < 0 || ii[k] >= N, oj = jj[k] < 0 || jj[k] >= N;
		if (oi && oj) { missing = k; continue; }
		if (oi || oj) texel_across_edge(c, f, ii[k], jj[k], s[k]);
		else texel(c, f, ii[k], jj[k], s[k]);
	}
	if (missing >= 0)
		for (int ch = 0; ch < 4; ++ch) {
			float acc = 0.0f;
			for (int k = 0; k < 4; ++k) if (k != missing) acc += s[k][ch];
			s[missing][ch] = acc / 3.0f;
		}
}

inline float dp4(const float a[4], const float b[4])
{
	return std::fmaf(a[3], b[3], std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0])));
}

}  // namespace

extern "C" {

// out_rgba: float[H][W][4] premultiplied result (zeros where the pixel is discarded), covered: uint8[H][W]
void orc_resolve_cube(const uint8_t* cube, int N, const orc_frame* fc, const float* wvp_i /* 4 rows as stored in the CB */,
	int W, int H, float* out_rgba, uint8_t* covered)
{
	const CubeTex ct{ cube, N };
#pragma omp parallel for schedule(static)
	for (int py = 0; py < H; ++py)
		for (int px = 0; px < W; ++px) {
			float* o = out_rgba + ((size_t)py * W + px) * 4;
			o[0] = o[1] = o[2] = o[3] = 0.0f;
			covered[(size_t)py * W + px] = 0;
			const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
			// TexcoordToLocalPos (PSRayCastCube.hlsl:20-29): (x, y, 0, 1) * worldViewProjI, perspective divide
			const float q[3] = { std::fmaf(u, 2.0f, -1.0f), std::fmaf(v, -2.0f, 1.0f), 1.0f };
			float h[4];
			for (int r = 0; r < 4; ++r) {
				const float col[3] = { wvp_i[4 * r + 0], wvp_i[4 * r + 1], wvp_i[4 * r + 3] };
				h[r] = dp3(q, col);
			}
			float pos[3] = { h[0] / h[3], h[1] / h[3], h[2] / h[3] };
			// local-space eye and ray (:98-100)
			const float e4[4] = { fc->eye_pt[0], fc->eye_pt[1], fc->eye_pt[2], 1.0f };
			float dir[3];
			for (int a = 0; a < 3; ++a) dir[a] = pos[a] + -dp4(e4, fc->world_i + 4 * a);
			const float inv = 1.0f / std::sqrt(dp3(dir, dir));
			for (int a = 0; a < 3; ++a) dir[a] = inv * dir[a];
			// ComputeRayHit (:34-61): the far (interior) face the ray leaves through
			float t[3];
			for (int a = 0; a < 3; ++a) {
				const float sgn = (float)((int)(0.0f < dir[a]) - (int)(dir[a] < 0.0f));
				t[a] = (-pos[a] + sgn) / dir[a];
			}
			float U = 3.40282347e+38f;
			int hit = -1;
			for (int i = 0; i < 3; ++i) {
				const int j = (i + 1) % 3, k = (i + 2) % 3;
				if (!(t[i] >= 0.0f)) continue;
				if (!(1.0f >= std::fabs(std::fmaf(dir[j], t[i], pos[j])))) continue;
				if (1.0f < std::fabs(std::fmaf(dir[k], t[i], pos[k]))) continue;
				if (t[i] < U) { U = t[i]; hit = i; }
			}
			if (hit < 0) continue;                                    // discard
			float P[3];
			for (int a = 0; a < 3; ++a) P[a] = std::fmaf(dir[a], U, pos[a]);
			// ComputeCubeTexcoord (:66-91)
			float uvx, uvy;
			if (hit == 0) { uvx = P[2] * -P[0]; uvy = P[1]; }
			else if (hit == 1) { uvx = P[0]; uvy = P[2] * -P[1]; }
			else { uvx = P[0] * P[2]; uvy = P[1]; }
			uvx = std::fmaf(uvx, 0.5f, 0.5f);
			uvy = std::fmaf(uvy, 0.5f, 0.5f);
			// CubeCast (PSCube.hlsli:66-122)
			float s[4][4], fu, fv;
			footprint(ct, P, s, fu, fv);
			const float g = (float)N;
			const float vf = -uvy + 1.0f;
			const float vN = vf * g, uN = uvx * g;
			float dv = std::fmaf(vf, g, 0.5f), du = std::fmaf(uvx, g, 0.5f);
			dv = dv - std::floor(dv); du = du - std::floor(du);      // frc
			// GetDomain (:41-61): clamp the exterior edge
			const float bound = g + -1.0f;
			bool ext = false;
			for (int a = 0; a < 3; ++a) {
				const float ax = P[a] * g;
				ext = ext || ((bound < std::fabs(ax)) && (dir[a] * ax < 0.0f));
			}
			if (ext) {
				dv = std::fmin(vN, g + -0.5f) < 0.5f ? 1.0f : 0.0f;
				du = std::fmin(uN, g + -0.5f) < 0.5f ? 1.0f : 0.0f;
			}
			const float idu = -du + 1.0f, idv = -dv + 1.0f;
			const float wy = dv * du, wx = dv * idu, wz = du * idv, ww = idv * idu;
			float ws = std::fmaf(idu, dv, wy);
			ws = std::fmaf(idv, du, ws);
			ws = std::fmaf(idu, idv, ws);
			float res[4];
			for (int ch = 0; ch < 4; ++ch) {
				float r = wy * s[1][ch];
				r = std::fmaf(s[0][ch], wx, r);
				r = std::fmaf(s[2][ch], wz, r);
				r = std::fmaf(s[3][ch], ww, r);
				res[ch] = r / ws;
			}
			if (!(0.0f < ws)) {                                        // SampleLevel fallback: bilinear over the same footprint
				for (int ch = 0; ch < 4; ++ch)
					res[ch] = lerpf(lerpf(s[3][ch], s[2][ch], fu), lerpf(s[0][ch], s[1][ch], fu), fv);
			}
			if (0.0f >= res[3]) continue;                             // discard
			for (int ch = 0; ch < 4; ++ch) o[ch] = res[ch];
			covered[(size_t)py * W + px] = 1;
		}
}

// PREMULTIPLIED blend of the resolve output over an R8G8B8A8_UNORM target, in place
void orc_blend_premultiplied(const float* src_rgba, const uint8_t* covered, uint8_t* target, int W, int H)
{
	for (size_t p = 0; p < (size_t)W * H; ++p) {
		if (!covered[p]) continue;
		const float* s = src_rgba + 4 * p;
		const float ia = 1.0f - s[3];
		for (int ch = 0; ch < 4; ++ch) {
			const float d = (float)target[4 * p + ch] / 255.0f;
			const float v = std::fmaf(d, ia, s[ch]);
			target[4 * p + ch] = !(v > 0.0f) ? 0 : v >= 1.0f ? 255 : (uint8_t)(v * 255.0f + 0.5f);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// PSEnvironment.hlsl (LightProbe::RenderEnvironment, LightProbe.cpp:70-97): the sky behind the volume.  Per pixel:
// pos = (x, y, 1, 1) * screenToWorld, perspective divide, dir = normalize(eyePt - pos), sample the radiance cube at -dir
// with level 0 (bilinear, seamless edges; float texels [6][N][N][3]).  s2w = the four constant-buffer rows of
// transpose(inverse(view * proj)); out rgb + alpha 0.
// ---------------------------------------------------------------------------------------------
void orc_environment(const float* cube, int N, const float eye[3], const float* s2w, int W, int H, float* out_rgba)
{
#pragma omp parallel for schedule(static)
	for (int py = 0; py < H; ++py)
		for (int px = 0; px < W; ++px) {
			const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
			const float q[4] = { std::fmaf(u, 2.0f, -1.0f), std::fmaf(v, -2.0f, 1.0f), 1.0f, 1.0f };
			float h[4];
			for (int r = 0; r < 4; ++r) h[r] = dp4(q, s2w + 4 * r);
			float d[3];
			for (int a = 0; a < 3; ++a) d[a] = -(h[a] / h[3]) + eye[a];
			const float inv = 1.0f / std::sqrt(dp3(d, d));
			for (int a = 0; a < 3; ++a) d[a] = -(inv * d[a]);
			// bilinear footprint of the float cube (same addressing rules as the RGBA8 cube map above)
			const int f = major_face(d);
			float sc, tc;
			face_coords(d, f, sc, tc);
			const float ma = std::fabs(d[f >> 1]);
			const float tu = std::fmaf(0.5f * (sc / ma) + 0.5f, (float)N, -0.5f);
			const float tv = std::fmaf(0.5f * (tc / ma) + 0.5f, (float)N, -0.5f);
			const float flu = std::floor(tu), flv = std::floor(tv), fu = tu - flu, fv = tv - flv;
			const int i0 = (int)flu, j0 = (int)flv;
			const int ii[4] = { i0, i0 + 1, i0 + 1, i0 }, jj[4] = { j0 + 1, j0 + 1, j0, j0 };
			float s[4][3];
			int missing = -1;
			for (int k = 0; k < 4; ++k) {
				const bool oi = ii[k] < 0 || ii[k] >= N, oj = jj[k] < 0 || jj[k] >= N;
				if (oi && oj) { missing = k; continue; }
				int g = f, i2 = ii[k], j2 = jj[k];
				if (oi || oj) {
					const float se = ii[k] < 0 ? -1.0f : ii[k] >= N ? 1.0f : texel_centre(ii[k], N);
					const float te = jj[k] < 0 ? -1.0f : jj[k] >= N ? 1.0f : texel_centre(jj[k], N);
					float P[3];
					face_point(P, f, se, te);
					for (int a = 0; a < 3; ++a)
						if (a != (f >> 1) && std::fabs(P[a]) == 1.0f) g = 2 * a + (P[a] < 0.0f ? 1 : 0);
					float s2, t2;
					face_coords(P, g, s2, t2);
					i2 = std::min(std::max((int)std::floor((0.5f * s2 + 0.5f) * (float)N), 0), N - 1);
					j2 = std::min(std::max((int)std::floor((0.5f * t2 + 0.5f) * (float)N), 0), N - 1);
				}
				for (int c = 0; c < 3; ++c) s[k][c] = cube[(((size_t)g * N + j2) * N + i2) * 3 + c];
			}
			if (missing >= 0)
				for (int c = 0; c < 3; ++c) {
					float acc = 0.0f;
					for (int k = 0; k < 4; ++k) if (k != missing) acc += s[k][c];
					s[missing][c] = acc / 3.0f;
				}
			float* o = out_rgba + ((size_t)py * W + px) * 4;
			for (int c = 0; c < 3; ++c) o[c] = lerpf(lerpf(s[3][c], s[2][c], fu), lerpf(s[0][c], s[1][c], fu), fv);
			o[3] = 0.0f;
		}
}

}  // extern "C"
