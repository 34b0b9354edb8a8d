// oracle/orc_host.cpp -- CPU oracle for the per-frame host rules (TEST INFRASTRUCTURE ONLY).
// Restates Fluid::UpdateFrame and its helpers:
//   /root/reference/FluidX12/Content/Fluid.cpp:283-346 UpdateFrame, :141-166 EstimateCubeMapLOD,
//   :86-106 ProjectToViewport, :108-139 EstimateCubeEdgePixelSize, :49-61 GenVisibilityMask,
//   :40-45 IsCubeFaceVisible, :168-183 constructor constants;
//   camera set-up of the demo driver: FluidX12/FluidX12.cpp:243-253.
// DirectXMath semantics restated in scalar fp32: row-vector convention (v * M), LookAtLH /
// PerspectiveFovLH / Inverse / Vector3TransformCoord.
#include "orc_common.h"
#include "fx_oracle.h"

namespace {

struct M4 { float m[4][4]; };

M4 mul(const M4& a, const M4& b)
{
	M4 r;
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j) {
			float s = a.m[i][0] * b.m[0][j];
			for (int k = 1; k < 4; ++k) s = std::fmaf(a.m[i][k], b.m[k][j], s);
			r.m[i][j] = s;
		}
	return r;
}

// cofactor inverse, result = adj * (1 / det) like XMMatrixInverse
M4 inverse(const M4& a)
{
	const float* m = &a.m[0][0];
	float inv[16];
	inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
	inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
	inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
	inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
	inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
	inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
	inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
	inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
	inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
	inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
	inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
	inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
	inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
	inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
	inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
	inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
	const float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
	const float rdet = 1.0f / det;
	M4 r;
	for (int i = 0; i < 16; ++i) (&r.m[0][0])[i] = inv[i] * rdet;
	return r;
}

// XMStoreFloat3x4: three rows of four = the transpose's first three rows
void store3x4(float out[12], const M4& a)
{
	for (int r = 0; r < 3; ++r)
		for (int c = 0; c < 4; ++c) out[r * 4 + c] = a.m[c][r];
}

void transform_coord(float out[3], const float v[3], const M4& a)   // XMVector3TransformCoord
{
	float h[4];
	for (int j = 0; j < 4; ++j)
		h[j] = std::fmaf(v[2], a.m[2][j], std::fmaf(v[1], a.m[1][j], v[0] * a.m[0][j])) + a.m[3][j];
	for (int j = 0; j < 3; ++j) out[j] = h[j] / h[3];
}

}  // namespace

extern "C" {

void orc_look_at_lh(const float eye[3], const float focus[3], const float up[3], float out16[16])
{
	float z[3] = { focus[0] - eye[0], focus[1] - eye[1], focus[2] - eye[2] };
	float l = std::sqrt(orc::dp3(z, z));
	for (int a = 0; a < 3; ++a) z[a] /= l;
	float x[3] = { up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0] };
	l = std::sqrt(orc::dp3(x, x));
	for (int a = 0; a < 3; ++a) x[a] /= l;
	const float y[3] = { z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0] };
	const float d[3] = { -orc::dp3(x, eye), -orc::dp3(y, eye), -orc::dp3(z, eye) };
	const float m[16] = { x[0], y[0], z[0], 0, x[1], y[1], z[1], 0, x[2], y[2], z[2], 0, d[0], d[1], d[2], 1 };
	std::memcpy(out16, m, sizeof m);
}

// CBPerObject.WorldViewProjI (Fluid.cpp:318): transpose(inverse(world * view * proj)), world = scale 10 (Fluid.cpp:182);
// out16 = the four float4 rows of the constant buffer
void orc_world_view_proj_inverse(const float view[16], const float proj[16], float out16[16])
{
	M4 V, P, W{};
	std::memcpy(V.m, view, 64);
	std::memcpy(P.m, proj, 64);
	W.m[0][0] = W.m[1][1] = W.m[2][2] = 10.0f; W.m[3][3] = 1.0f;
	const M4 I = inverse(mul(W, mul(V, P)));
	for (int r = 0; r < 4; ++r)
		for (int c = 0; c < 4; ++c) out16[r * 4 + c] = I.m[c][r];
}

void orc_perspective_fov_lh(float fovy, float aspect, float zn, float zf, float out16[16])
{
	const float h = std::cos(0.5f * fovy) / std::sin(0.5f * fovy);
	const float w = h / aspect;
	const float q = zf / (zf - zn);
	const float m[16] = { w, 0, 0, 0, 0, h, 0, 0, 0, 0, q, 1, 0, 0, -q * zn, 0 };
	std::memcpy(out16, m, sizeof m);
}

void orc_update_frame(const float view[16], const float proj[16], const float eye[3],
	uint32_t viewport_w, uint32_t viewport_h, uint32_t grid_x, uint32_t max_ray_samples,
	orc_frame* fc, uint32_t* lod, uint32_t* ray_samples, uint32_t* mask, float* edge_px)
{
	M4 V, P, W{};
	std::memcpy(V.m, view, 64);
	std::memcpy(P.m, proj, 64);
	W.m[0][0] = W.m[1][1] = W.m[2][2] = 10.0f; W.m[3][3] = 1.0f;       // Fluid.cpp:182
	const M4 WI = inverse(W);
	const M4 WVP = mul(W, mul(V, P));                                  // Fluid.cpp:298,315
	store3x4(fc->world_i, WI);
	store3x4(fc->world, W);
	for (int a = 0; a < 3; ++a) fc->eye_pt[a] = eye[a];
	fc->light_pt[0] = 75.0f; fc->light_pt[1] = 75.0f; fc->light_pt[2] = -75.0f;
	const float pi = 3.141592654f;                                     // XM_PI
	fc->light_color[0] = 1.0f; fc->light_color[1] = 0.7f; fc->light_color[2] = 0.3f; fc->light_color[3] = pi * 3.0f;
	fc->ambient[0] = fc->ambient[1] = fc->ambient[2] = 1.0f; fc->ambient[3] = pi * 1.5f;

	// EstimateCubeMapLOD (Fluid.cpp:141-166)
	static const float corner[8][3] = { {1,1,1},{-1,1,1},{1,-1,1},{-1,-1,1},{-1,1,-1},{1,1,-1},{-1,-1,-1},{1,-1,-1} };
	static const uint8_t ei[12][2] = { {0,1},{3,2},{1,3},{2,0},{4,5},{7,6},{5,7},{6,4},{1,4},{6,3},{5,0},{2,7} };
	float px[8][2];
	for (int i = 0; i < 8; ++i) {                                      // ProjectToViewport :86-106
		float p[3];
		transform_coord(p, corner[i], WVP);
		px[i][0] = (p[0] * 0.5f + 0.5f) * (float)viewport_w;
		px[i][1] = (p[1] * -0.5f + 0.5f) * (float)viewport_h;
	}
	float s = 0.0f;
	for (int i = 0; i < 12; ++i) {                                     // EstimateCubeEdgePixelSize :108-139
		const float ex = px[ei[i][1]][0] - px[ei[i][0]][0], ey = px[ei[i][1]][1] - px[ei[i][0]][1];
		s = std::max(std::sqrt(ex * ex + ey * ey), s);
	}
	if (edge_px) *edge_px = s;
	s = s / 2.0f;                                                      // upscale = 2
	float amt = 2.0f * s / std::sqrt(3.0f);                            // raySampleCountScale = 2
	const uint32_t cnt = (uint32_t)std::ceil(amt);
	const uint32_t count = std::min(cnt, max_ray_samples);
	amt = std::min(amt, (float)count);
	s = amt / 2.0f * std::sqrt(3.0f);
	const uint8_t level = (uint8_t)std::max(std::log2((float)grid_x / s), 0.0f);
	*lod = std::min<uint8_t>(level, 5 - 1);                            // numMips = 5 (Fluid.cpp:229)
	*ray_samples = count;

	// GenVisibilityMask (Fluid.cpp:49-61): XMVector3Transform(eye, worldI)
	uint32_t mk = 0;
	for (uint32_t f = 0; f < 6; ++f) {
		const int a = f >> 1;
		const float c = std::fmaf(eye[2], WI.m[2][a], std::fmaf(eye[1], WI.m[1][a], eye[0] * WI.m[0][a])) + WI.m[3][a];
		const bool vis = (f & 1) ? c > -1.0f : c < 1.0f;
		mk |= (vis ? 1u : 0u) << f;
	}
	*mask = mk;
}

}  // extern "C"
