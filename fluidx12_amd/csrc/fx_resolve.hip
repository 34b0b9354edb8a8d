// fx_resolve.hip -- cube map -> screen resolve, gfx950.
//
//   k_resolve_cube  <- PSRayCastCube.hlsl:20-113 (TexcoordToLocalPos, ComputeRayHit, ComputeCubeTexcoord, main)
//                      + PSCube.hlsli:41-122 (GetDomain, CubeCast), the raster-free formulation of Fluid::renderCube
//                      (Fluid.cpp:910-931), output merger = PREMULTIPLIED blend into an R8G8B8A8_UNORM target
//                      (Fluid.cpp:653, FluidX12.cpp:31)
// (paths relative to /root/reference/FluidX12/Content/Shaders/).  One thread per screen pixel; a wave = 64 x 1 pixels,
// so the target is written as coalesced 256-byte runs and neighbouring rays hit neighbouring cube texels (the cube
// mip is <= 1.5 MiB: L2 resident).  The fixed-function TextureCube pieces (face selection, gather4 footprint, seamless
// edges) are restated in code: hardware filtering would bring its own fixed-point weights and break parity.
// Operation order follows the shipped DXBC (mad = fmaf, rsq = 1/sqrtf).  Bytes: W*H*4 read + written for the blend,
// 4 texels x 4 B per covered pixel from L2 -- a launch-latency-sized pass (2 M pixels), not a roofline subject.
#include "fx_internal.h"

namespace fx {

namespace {

__device__ __forceinline__ float rdot3(float ax, float ay, float az, float bx, float by, float bz)
{
	return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

// point on face f at (sc, tc), D3D cube face table
__device__ __forceinline__ void face_point(float p[3], int f, float sc, float tc)
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

__device__ __forceinline__ void face_coords(const float p[3], int f, float& sc, float& tc)
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

__device__ __forceinline__ float4 texel(const uint32_t* __restrict__ cube, int N, int f, int i, int j)
{
	const uint32_t q = cube[((size_t)f * N + j) * N + i];
	return make_float4((float)(q & 255u) / 255.0f, (float)((q >> 8) & 255u) / 255.0f,
		(float)((q >> 16) & 255u) / 255.0f, (float)(q >> 24) / 255.0f);
}

// texel (i, j) of face f with exactly one coordinate off the face: the adjacent face's edge texel at the same
// position along the shared edge (seamless cube filtering)
__device__ float4 texel_across_edge(const uint32_t* __restrict__ cube, int N, int f, int i, int j)
{
	const float sc = i < 0 ? -1.0f : i >= N ? 1.0f : (2.0f * (float)i + 1.0f) / (float)N - 1.0f;
	const float tc = j < 0 ? -1.0f : j >= N ? 1.0f : (2.0f * (float)j + 1.0f) / (float)N - 1.0f;
	float P[3];
	face_point(P, f, sc, tc);
	const int fa = f >> 1;
	int g = 0;
#pragma unroll
	for (int a = 0; a < 3; ++a)
		if (a != fa && fabsf(P[a]) == 1.0f) g = 2 * a + (P[a] < 0.0f ? 1 : 0);
	float s2, t2;
	face_coords(P, g, s2, t2);
	const int i2 = min(max((int)floorf((0.5f * s2 + 0.5f) * (float)N), 0), N - 1);
	const int j2 = min(max((int)floorf((0.5f * t2 + 0.5f) * (float)N), 0), N - 1);
	return texel(cube, N, g, i2, j2);
}

__device__ __forceinline__ uint32_t unorm8(float v)
{
	if (!(v > 0.0f)) return 0u;
	if (v >= 1.0f) return 255u;
	return (uint32_t)(v * 255.0f + 0.5f);
}

}  // namespace

// wvp_i: the four constant-buffer rows of CBPerObject.WorldViewProjI.  out_float (optional): the shader's SV_TARGET
// before the output merger, float4 per pixel, zeros where discarded (parity tests); target (optional): RGBA8, blended in place.
__global__ __launch_bounds__(256) void k_resolve_cube(const uint32_t* __restrict__ cube, int N, const FrameConsts fc,
	int W, int H, uint32_t* __restrict__ target, float4* __restrict__ out_float)
{
	const int px = blockIdx.x * blockDim.x + threadIdx.x;
	const int py = blockIdx.y * blockDim.y + threadIdx.y;
	if (px >= W || py >= H) return;
	const size_t pix = (size_t)py * W + px;
	if (out_float) out_float[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

	const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;      // VSScreenQuad.hlsl:22
	// TexcoordToLocalPos (PSRayCastCube.hlsl:20-29)
	const float qx = fmaf(u, 2.0f, -1.0f), qy = fmaf(v, -2.0f, 1.0f);
	const float* M = fc.wvp_i;
	const float h0 = rdot3(qx, qy, 1.0f, M[0], M[1], M[3]), h1 = rdot3(qx, qy, 1.0f, M[4], M[5], M[7]);
	const float h2 = rdot3(qx, qy, 1.0f, M[8], M[9], M[11]), h3 = rdot3(qx, qy, 1.0f, M[12], M[13], M[15]);
	const float pos[3] = { h0 / h3, h1 / h3, h2 / h3 };
	float dir[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {                                                          // :98-100
		const float* r = fc.world_i + 4 * a;
		const float e = fmaf(1.0f, r[3], fmaf(fc.eye_pt[2], r[2], fmaf(fc.eye_pt[1], r[1], fc.eye_pt[0] * r[0])));
		dir[a] = pos[a] + -e;
	}
	const float inv = 1.0f / sqrtf(rdot3(dir[0], dir[1], dir[2], dir[0], dir[1], dir[2]));
#pragma unroll
	for (int a = 0; a < 3; ++a) dir[a] = inv * dir[a];
	// ComputeRayHit (:34-61)
	float t[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float sgn = (float)((int)(0.0f < dir[a]) - (int)(dir[a] < 0.0f));
		t[a] = (-pos[a] + sgn) / dir[a];
	}
	float U = 3.40282347e+38f;
	int hit = -1;
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const int j = (i + 1) % 3, k = (i + 2) % 3;
		if (!(t[i] >= 0.0f)) continue;
		if (!(1.0f >= fabsf(fmaf(dir[j], t[i], pos[j])))) continue;
		if (1.0f < fabsf(fmaf(dir[k], t[i], pos[k]))) continue;
		if (t[i] < U) { U = t[i]; hit = i; }
	}
	if (hit < 0) return;                                                                   // discard
	float P[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) P[a] = fmaf(dir[a], U, pos[a]);
	// ComputeCubeTexcoord (:66-91)
	float uvx, uvy;
	if (hit == 0) { uvx = P[2] * -P[0]; uvy = P[1]; }
	else if (hit == 1) { uvx = P[0]; uvy = P[2] * -P[1]; }
	else { uvx = P[0] * P[2]; uvy = P[1]; }
	uvx = fmaf(uvx, 0.5f, 0.5f);
	uvy = fmaf(uvy, 0.5f, 0.5f);

	// TextureCube footprint of direction P: major axis (ties Z > Y > X), gather order x (i0,j1) y (i1,j1) z (i1,j0) w (i0,j0)
	int f;
	{
		const float ax = fabsf(P[0]), ay = fabsf(P[1]), az = fabsf(P[2]);
		if (az >= ax && az >= ay) f = P[2] < 0.0f ? 5 : 4;
		else if (ay >= ax) f = P[1] < 0.0f ? 3 : 2;
		else f = P[0] < 0.0f ? 1 : 0;
	}
	float sc, tc;
	face_coords(P, f, sc, tc);
	const float ma = fabsf(P[f >> 1]);
	const float tu = fmaf(0.5f * (sc / ma) + 0.5f, (float)N, -0.5f);
	const float tv = fmaf(0.5f * (tc / ma) + 0.5f, (float)N, -0.5f);
	const float flu = floorf(tu), flv = floorf(tv);
	const int i0 = (int)flu, j0 = (int)flv;
	float4 s[4];
	int missing = -1;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int ii = i0 + ((k == 1 || k == 2) ? 1 : 0), jj = j0 + (k < 2 ? 1 : 0);
		const bool oi = ii < 0 || ii >= N, oj = jj < 0 || jj >= N;
		if (oi && oj) { missing = k; s[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
		else if (oi || oj) s[k] = texel_across_edge(cube, N, f, ii, jj);
		else s[k] = texel(cube, N, f, ii, jj);
	}
	if (missing >= 0) {                                 // off a corner: no face there, mean of the other three
		float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
		for (int k = 0; k < 4; ++k)
			if (k != missing) { acc.x += s[k].x; acc.y += s[k].y; acc.z += s[k].z; acc.w += s[k].w; }
		const float4 m3 = make_float4(acc.x / 3.0f, acc.y / 3.0f, acc.z / 3.0f, acc.w / 3.0f);
#pragma unroll
		for (int k = 0; k < 4; ++k) if (k == missing) s[k] = m3;
	}

	// CubeCast / GetDomain (PSCube.hlsli:41-122)
	const float g = (float)N;
	const float vf = -uvy + 1.0f;
	const float vN = vf * g, uN = uvx * g;
	float dv = fmaf(vf, g, 0.5f), du = fmaf(uvx, g, 0.5f);
	dv = dv - floorf(dv); du = du - floorf(du);
	const float bound = g + -1.0f;
	bool ext = false;
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float ax = P[a] * g;
		ext = ext || ((bound < fabsf(ax)) && (dir[a] * ax < 0.0f));
	}
	if (ext) {                                          // clamp the exterior edge
		dv = fminf(vN, g + -0.5f) < 0.5f ? 1.0f : 0.0f;
		du = fminf(uN, g + -0.5f) < 0.5f ? 1.0f : 0.0f;
	}
	const float idu = -du + 1.0f, idv = -dv + 1.0f;
	const float wy = dv * du, wx = dv * idu, wz = du * idv, ww = idv * idu;
	float ws = fmaf(idu, dv, wy);
	ws = fmaf(idv, du, ws);
	ws = fmaf(idu, idv, ws);
	float res[4];
#define FX_CH(c, i) { float r = wy * s[1].c; r = fmaf(s[0].c, wx, r); r = fmaf(s[2].c, wz, r); r = fmaf(s[3].c, ww, r); res[i] = r / ws; }
	FX_CH(x, 0) FX_CH(y, 1) FX_CH(z, 2) FX_CH(w, 3)
#undef FX_CH
	if (!(0.0f < ws)) {                                 // SampleLevel fallback: bilinear over the same footprint
		const float fu = tu - flu, fv = tv - flv;
#define FX_CH(c, i) res[i] = fmaf(fv, fmaf(fu, s[1].c - s[0].c, s[0].c) - fmaf(fu, s[2].c - s[3].c, s[3].c), fmaf(fu, s[2].c - s[3].c, s[3].c));
		FX_CH(x, 0) FX_CH(y, 1) FX_CH(z, 2) FX_CH(w, 3)
#undef FX_CH
	}
	if (0.0f >= res[3]) return;                                                            // discard
	if (out_float) out_float[pix] = make_float4(res[0], res[1], res[2], res[3]);
	if (target) {                                       // PREMULTIPLIED: src + dst * (1 - src.a), FLOAT -> UNORM
		const uint32_t d = target[pix];
		const float ia = 1.0f - res[3];
		const uint32_t r8 = unorm8(fmaf((float)(d & 255u) / 255.0f, ia, res[0]));
		const uint32_t g8 = unorm8(fmaf((float)((d >> 8) & 255u) / 255.0f, ia, res[1]));
		const uint32_t b8 = unorm8(fmaf((float)((d >> 16) & 255u) / 255.0f, ia, res[2]));
		const uint32_t a8 = unorm8(fmaf((float)(d >> 24) / 255.0f, ia, res[3]));
		target[pix] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
	}
}

// sky pass: PSEnvironment.hlsl (LightProbe::RenderEnvironment, LightProbe.cpp:85-97).  pos = (x, y, 1, 1) * screenToWorld / w,
// dir = normalize(eyePt - pos), the radiance cube (float3 texels, mip 0) sampled bilinearly at -dir with seamless edges;
// the pass draws without blending: the target takes rgb and alpha 0.
__global__ __launch_bounds__(256) void k_environment(const float* __restrict__ cube, int N, const FrameConsts fc, int W, int H,
	uint32_t* __restrict__ target, float4* __restrict__ out_float)
{
	const int px = blockIdx.x * blockDim.x + threadIdx.x;
	const int py = blockIdx.y * blockDim.y + threadIdx.y;
	if (px >= W || py >= H) return;
	const size_t pix = (size_t)py * W + px;
	const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
	const float qx = fmaf(u, 2.0f, -1.0f), qy = fmaf(v, -2.0f, 1.0f);
	const float* M = fc.s2w;
	float h[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) h[r] = fmaf(1.0f, M[4 * r + 3], fmaf(1.0f, M[4 * r + 2], fmaf(qy, M[4 * r + 1], qx * M[4 * r + 0])));
	float d[3];
#pragma unroll
	for (int a = 0; a < 3; ++a) d[a] = -(h[a] / h[3]) + fc.eye_pt[a];
	const float inv = 1.0f / sqrtf(rdot3(d[0], d[1], d[2], d[0], d[1], d[2]));
#pragma unroll
	for (int a = 0; a < 3; ++a) d[a] = -(inv * d[a]);
	int f;
	{
		const float ax = fabsf(d[0]), ay = fabsf(d[1]), az = fabsf(d[2]);
		if (az >= ax && az >= ay) f = d[2] < 0.0f ? 5 : 4;
		else if (ay >= ax) f = d[1] < 0.0f ? 3 : 2;
		else f = d[0] < 0.0f ? 1 : 0;
	}
	float sc, tc;
	face_coords(d, f, sc, tc);
	const float ma = fabsf(d[f >> 1]);
	const float tu = fmaf(0.5f * (sc / ma) + 0.5f, (float)N, -0.5f);
	const float tv = fmaf(0.5f * (tc / ma) + 0.5f, (float)N, -0.5f);
	const float flu = floorf(tu), flv = floorf(tv), fu = tu - flu, fv = tv - flv;
	const int i0 = (int)flu, j0 = (int)flv;
	float s[4][3];
	int missing = -1;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int ii = i0 + ((k == 1 || k == 2) ? 1 : 0), jj = j0 + (k < 2 ? 1 : 0);
		const bool oi = ii < 0 || ii >= N, oj = jj < 0 || jj >= N;
		if (oi && oj) { missing = k; s[k][0] = s[k][1] = s[k][2] = 0.0f; continue; }
		int g = f, i2 = ii, j2 = jj;
		if (oi || oj) {                                 // across one edge: the adjacent face's edge texel at the same place along it
			const float se = ii < 0 ? -1.0f : ii >= N ? 1.0f : (2.0f * (float)ii + 1.0f) / (float)N - 1.0f;
			const float te = jj < 0 ? -1.0f : jj >= N ? 1.0f : (2.0f * (float)jj + 1.0f) / (float)N - 1.0f;
			float P[3];
			face_point(P, f, se, te);
#pragma unroll
			for (int a = 0; a < 3; ++a)
				if (a != (f >> 1) && fabsf(P[a]) == 1.0f) g = 2 * a + (P[a] < 0.0f ? 1 : 0);
			float s2, t2;
			face_coords(P, g, s2, t2);
			i2 = min(max((int)floorf((0.5f * s2 + 0.5f) * (float)N), 0), N - 1);
			j2 = min(max((int)floorf((0.5f * t2 + 0.5f) * (float)N), 0), N - 1);
		}
		const float* q = cube + (((size_t)g * N + j2) * N + i2) * 3;
		s[k][0] = q[0]; s[k][1] = q[1]; s[k][2] = q[2];
	}
	if (missing >= 0) {
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			float acc = 0.0f;
#pragma unroll
			for (int k = 0; k < 4; ++k) if (k != missing) acc += s[k][c];
			const float m3 = acc / 3.0f;
#pragma unroll
			for (int k = 0; k < 4; ++k) if (k == missing) s[k][c] = m3;
		}
	}
	float o[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const float top = fmaf(fu, s[2][c] - s[3][c], s[3][c]), bot = fmaf(fu, s[1][c] - s[0][c], s[0][c]);
		o[c] = fmaf(fv, bot - top, top);
	}
	if (out_float) out_float[pix] = make_float4(o[0], o[1], o[2], 0.0f);
	if (target) target[pix] = unorm8(o[0]) | (unorm8(o[1]) << 8) | (unorm8(o[2]) << 16);
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

hipError_t launch_resolve_cube(const uint8_t* cube_mip, int N, const FrameConsts& fc, int W, int H, uint8_t* target,
	float* out_float, hipStream_t s)
{
	const dim3 block(64, 4, 1), grid((W + 63) / 64, (H + 3) / 4, 1);
	hipLaunchKernelGGL(k_resolve_cube, grid, block, 0, s, reinterpret_cast<const uint32_t*>(cube_mip), N, fc, W, H,
		reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float));
	return hipGetLastError();
}

hipError_t launch_environment(const float* cube, int n, const FrameConsts& fc, int W, int H, uint8_t* target, float* out_float, hipStream_t s)
{
	const dim3 block(64, 4, 1), grid((W + 63) / 64, (H + 3) / 4, 1);
	hipLaunchKernelGGL(k_environment, grid, block, 0, s, cube, n, fc, W, H, reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float));
	return hipGetLastError();
}

// ClearRenderTargetView: FLOAT -> UNORM per channel
hipError_t launch_clear_target(uint8_t* target, int W, int H, const float rgba[4], hipStream_t s)
{
	uint32_t v = 0;
	for (int c = 0; c < 4; ++c) {
		const float x = rgba[c];
		const uint32_t b = !(x > 0.0f) ? 0u : x >= 1.0f ? 255u : (uint32_t)(x * 255.0f + 0.5f);
		v |= b << (8 * c);
	}
	const size_t n = (size_t)W * H;
	hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, s,
		reinterpret_cast<uint32_t*>(target), v, n);
	return hipGetLastError();
}

}  // namespace fx
