// fx_sh.hip -- order-3 spherical-harmonics projection of a radiance cube map (light probe, config 5).
//
//   k_sh_cubemap    <- XUSG/Shaders/CSSHCubeMap.hlsl:33-96   texel -> weighted SH basis, 32-texel partial sums
//   k_sh_sum        <- XUSG/Shaders/CSSHSum.hlsl:28-59       32:1 tree passes
//   k_sh_normalize  <- XUSG/Shaders/CSSHNormalize.hlsl:11-18 x 4 pi / sum(dOmega)
//   host loop       <- Content/LightProbeEZ.cpp:183-278
// (paths relative to /root/reference/FluidX12/).  The reference reduces 32-lane groups through an LDS tree
// (WaveOpTypeless.hlsli:26-44: g[l] += g[l+s], s = 16..1).  Here one wave64 carries two such groups and the
// tree runs on cross-lane shuffles inside each 32-lane half -- the same pairing, hence the same fp32 sums.
// Deviation (documented in DESIGN.md): every sum pass sees its own element count; the reference's EZ path
// binds constant-buffer slice 0 for all passes (LightProbeEZ.cpp:245-246) and so re-adds 20 stale partials
// in its third pass.
#include "fx_internal.h"

namespace fx {

static const int kGroup = 32;    // SH_GROUP_SIZE / SH_WAVE_SIZE (XUSGSHSharedConsts.h:7-8)

// value of lane (l & ~31) = tree sum of the 32-lane half
__device__ __forceinline__ float tree32(float v)
{
#pragma unroll
	for (int s = 16; s >= 1; s >>= 1) v = __shfl_down(v, s, 32) + v;
	return v;
}

__global__ __launch_bounds__(256) void k_sh_cubemap(const float* __restrict__ cube, int N, int total,
	float* __restrict__ sh_out, float* __restrict__ w_out)
{
	const int id = blockIdx.x * blockDim.x + threadIdx.x;
	const int group = id / kGroup;
	const bool live = id < total;
	const float size = (float)N;
	float diffSolid = 0.0f, cw[3] = { 0.0f, 0.0f, 0.0f }, basis[9];
#pragma unroll
	for (int i = 0; i < 9; ++i) basis[i] = 0.0f;
	if (live) {
		const int face = id / (N * N), xy = id % (N * N), ix = xy % N, iy = xy / N;
		// GetCubeTexcoord (CubeMap.hlsli:26-35, 5-24)
		const float px = fmaf(-size, 0.5f, (float)ix) + 0.5f;
		const float py = -(fmaf(-size, 0.5f, (float)iy) + 0.5f);
		const float pz = size * 0.5f;
		float dx, dy, dz;
		switch (face) {
		case 0: dx = pz;  dy = py;  dz = -px; break;
		case 1: dx = -pz; dy = py;  dz = px;  break;
		case 2: dx = px;  dy = pz;  dz = -py; break;
		case 3: dx = px;  dy = -pz; dz = py;  break;
		case 4: dx = px;  dy = py;  dz = pz;  break;
		default: dx = -px; dy = py; dz = -pz; break;
		}
		const float r = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));        // CSSHCubeMap.hlsl:47
		const float x = dx * r, y = dy * r, z = dz * r;
		const float inv = 1.0f / size;
		const float bb = inv + -1.0f;                                             // :54
		const float a1 = -inv + 1.0f;
		const float ss = N > 1 ? (a1 + a1) / (size + -1.0f) : 0.0f;               // :55
		const float u = fmaf((float)ix, ss, bb), v = fmaf((float)iy, ss, bb);     // :56
		const float diff = fmaf(v, v, u * u) + 1.0f;                              // :57
		diffSolid = 4.0f / (sqrtf(diff) * diff);                                  // :58
		// sh_eval_basis_2 (SHMath.hlsli:37-66) as compiled
		basis[0] = 0.282094806f;
		basis[1] = y * -0.488602519f;
		basis[2] = z * 0.488602519f;
		basis[3] = x * -0.488602519f;
		basis[4] = (y * x) * 1.09254849f;
		const float p21 = z * -1.09254849f;
		basis[5] = y * p21;
		basis[6] = fmaf(z * z, 0.946174681f, -0.31539157f);
		basis[7] = x * p21;
		basis[8] = fmaf(x, x, -(y * y)) * 0.546274245f;
		const float* col = cube + (size_t)id * 3;                                 // :46 (texel-centre fetch)
		cw[0] = diffSolid * col[0]; cw[1] = diffSolid * col[1]; cw[2] = diffSolid * col[2];   // :78
	}
	const bool lead = (threadIdx.x & (kGroup - 1)) == 0 && group * kGroup < total;
	const float wsum = tree32(diffSolid);                                         // :59
	if (lead) w_out[group] = wsum;                                                // :71
#pragma unroll
	for (int i = 0; i < 9; ++i)
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			const float s = tree32(cw[k] * basis[i]);                             // :80-82
			if (lead) sh_out[(size_t)group * 27 + i * 3 + k] = s;                 // :94
		}
}

__global__ __launch_bounds__(256) void k_sh_sum(const float* __restrict__ sh_in, const float* __restrict__ w_in, int count,
	float* __restrict__ sh_out, float* __restrict__ w_out)
{
	const int id = blockIdx.x * blockDim.x + threadIdx.x;
	const int group = id / kGroup;
	const bool live = id < count;                                                 // CSSHSum.hlsl:34
	const bool lead = (threadIdx.x & (kGroup - 1)) == 0 && group * kGroup < count;
	const float w = tree32(live ? w_in[id] : 0.0f);                               // :37,40
	if (lead) w_out[group] = w;                                                   // :57
	for (int k = 0; k < 27; ++k) {
		const float s = tree32(live ? sh_in[(size_t)id * 27 + k] : 0.0f);
		if (lead) sh_out[(size_t)group * 27 + k] = s;                             // :56
	}
}

__global__ void k_sh_normalize(const float* __restrict__ sh_in, const float* __restrict__ w_in, float* __restrict__ out)
{
	const int k = threadIdx.x;
	if (k >= 27) return;
	const float wt = w_in[0];
	const float norm = 0.0f < wt ? 12.566371f / wt : 0.0f;                        // CSSHNormalize.hlsl:14-15
	out[k] = norm * sh_in[k];                                                     // :17
}

size_t sh_scratch_floats(int n, int which)
{
	const size_t total = (size_t)6 * n * n;
	const size_t g0 = (total + kGroup - 1) / kGroup, g1 = (g0 + kGroup - 1) / kGroup;
	switch (which) {
	case 0: return g0 * 27;
	case 1: return g1 * 27 + 27;
	case 2: return g0;
	default: return g1 + 1;
	}
}

hipError_t launch_sh_transform(const float* cube, int n, float* s0, float* s1, float* w0, float* w1, float* out27, hipStream_t st)
{
	const int total = 6 * n * n;
	hipLaunchKernelGGL(k_sh_cubemap, dim3((total + 255) / 256), dim3(256), 0, st, cube, n, total, s0, w0);   // LightProbeEZ.cpp:183-211
	float* S[2] = { s0, s1 };
	float* W[2] = { w0, w1 };
	int src = 0;
	for (int cnt = (total + kGroup - 1) / kGroup; cnt > 1; cnt = (cnt + kGroup - 1) / kGroup) {             // :213-252
		hipLaunchKernelGGL(k_sh_sum, dim3((cnt + 255) / 256), dim3(256), 0, st, S[src], W[src], cnt, S[src ^ 1], W[src ^ 1]);
		src ^= 1;
	}
	hipLaunchKernelGGL(k_sh_normalize, dim3(1), dim3(32), 0, st, S[src], W[src], out27);                    // :254-278
	return hipGetLastError();
}

}  // namespace fx
