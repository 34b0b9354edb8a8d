// oracle/orc_sh.cpp -- CPU oracle for the order-3 spherical-harmonics light probe (TEST
// INFRASTRUCTURE ONLY).  Restates
//   CSSHCubeMap::main   /root/reference/FluidX12/XUSG/Shaders/CSSHCubeMap.hlsl:33-96
//   CSSHSum::main       XUSG/Shaders/CSSHSum.hlsl:28-59
//   CSSHNormalize::main XUSG/Shaders/CSSHNormalize.hlsl:11-18
//   sh_eval_basis_2     XUSG/Shaders/SHMath.hlsli:37-66;  GetCubeTexcoord CubeMap.hlsli:5-35
//   WaveLanesSum        XUSG/Shaders/WaveOpTypeless.hlsli:26-44 (SM5 LDS tree, wave = group = 32,
//                       XUSG/Advanced/XUSGSHSharedConsts.h:5-8)
//   host loop           Content/LightProbeEZ.cpp:183-278 (shCubeMap / shSum / shNormalize)
// in the operation order of the shipped Bin/CSSH{CubeMap,Sum,Normalize}.cso (tools/dxbc.py).
// The radiance fetch (SampleLevel at the texel-centre direction, LINEAR_WRAP) is restated as the
// texel itself (bilinear weights are exactly 1/0 there).
#include "orc_common.h"
#include "fx_oracle.h"
#include <vector>

using namespace orc;

namespace {

const int GROUP = 32;    // SH_GROUP_SIZE = SH_WAVE_SIZE = 32

// g[lane] += g[lane + s] for s = 16, 8, 4, 2, 1 (WaveOpTypeless.hlsli:31-37); result in v[0]
float tree32(float v[GROUP])
{
	for (int s = GROUP / 2; s >= 1; s >>= 1)
		for (int l = 0; l < s; ++l) v[l] = v[l + s] + v[l];
	return v[0];
}

void basis_order3(float b[9], const float n[3])   // sh_eval_basis_2 as compiled (constants folded to fp32)
{
	const float x = n[0], y = n[1], z = n[2];
	b[0] = 0.282094806f;
	b[1] = y * -0.488602519f;
	b[2] = z * 0.488602519f;
	b[3] = x * -0.488602519f;
	b[4] = (y * x) * 1.09254849f;                                 // p_2_2 * (x y + y x) folded to 2 p_2_2
	const float p21 = z * -1.09254849f;
	b[5] = y * p21;
	b[6] = std::fmaf(z * z, 0.946174681f, -0.31539157f);
	b[7] = x * p21;
	b[8] = std::fmaf(x, x, -(y * y)) * 0.546274245f;
}

}  // namespace

extern "C" {

void orc_sh_transform(const float* cube, int N, float* out27, int quirk)
{
	const int total = 6 * N * N;
	const int nGroups = (total + GROUP - 1) / GROUP;
	const float size = (float)N;
	std::vector<float> sh((size_t)nGroups * 27, 0.0f), wt(nGroups, 0.0f);

	// ---- CSSHCubeMap (one group of 32 consecutive texels -> one partial) ------------------
	const float inv = 1.0f / size;
	const float bb = inv + -1.0f;                                           // :54
	const float a1 = -inv + 1.0f;
	const float ss = N > 1 ? (a1 + a1) / (size + -1.0f) : 0.0f;            // :55
	for (int g = 0; g < nGroups; ++g) {
		float w[GROUP], c[27][GROUP];
		for (int l = 0; l < GROUP; ++l) {
			const int id = g * GROUP + l;
			if (id >= total) { w[l] = 0.0f; for (int k = 0; k < 27; ++k) c[k][l] = 0.0f; continue; }
			const int face = id / (N * N), xy = id % (N * N), ix = xy % N, iy = xy / N;
			// GetCubeTexcoord (CubeMap.hlsli:26-35, 5-24)
			const float px = std::fmaf(-size, 0.5f, (float)ix) + 0.5f;
			const float py = -(std::fmaf(-size, 0.5f, (float)iy) + 0.5f);
			const float pz = size * 0.5f;
			float d[3];
			switch (face) {
			case 0: d[0] = pz;  d[1] = py; d[2] = -px; break;
			case 1: d[0] = -pz; d[1] = py; d[2] = px;  break;
			case 2: d[0] = px;  d[1] = pz; d[2] = -py; break;
			case 3: d[0] = px;  d[1] = -pz; d[2] = py; break;
			case 4: d[0] = px;  d[1] = py; d[2] = pz;  break;
			default: d[0] = -px; d[1] = py; d[2] = -pz; break;
			}
			const float* col = cube + (size_t)id * 3;                       // :46
			const float r = 1.0f / std::sqrt(dp3(d, d));                    // :47
			const float n[3] = { d[0] * r, d[1] * r, d[2] * r };
			const float u = std::fmaf((float)ix, ss, bb), v = std::fmaf((float)iy, ss, bb);   // :56
			const float diff = std::fmaf(v, v, u * u) + 1.0f;               // :57
			const float diffSolid = 4.0f / (std::sqrt(diff) * diff);        // :58
			w[l] = diffSolid;
			float b[9];
			basis_order3(b, n);
			for (int k = 0; k < 3; ++k) {
				const float cw = diffSolid * col[k];                        // :78
				for (int i = 0; i < 9; ++i) c[i * 3 + k][l] = cw * b[i];
			}
		}
		wt[g] = tree32(w);                                                  // :59,71
		for (int k = 0; k < 27; ++k) sh[(size_t)g * 27 + k] = tree32(c[k]); // :80-95
	}

	// ---- CSSHSum passes (LightProbeEZ.cpp:213-252) -------------------------------------------
	// ping-pong buffers like m_coeffSH[0/1]; buffer 0 keeps its stale tail, which `quirk` re-reads
	std::vector<float> sh1((size_t)((nGroups + GROUP - 1) / GROUP) * 27, 0.0f), wt1((nGroups + GROUP - 1) / GROUP, 0.0f);
	std::vector<float>* S[2] = { &sh, &sh1 };
	std::vector<float>* W[2] = { &wt, &wt1 };
	int src = 0;
	const int firstCount = nGroups;
	for (int n = nGroups; n > 1; n = (n + GROUP - 1) / GROUP) {
		const int dst = !src;
		const int groups = (n + GROUP - 1) / GROUP;
		const int count = quirk ? firstCount : n;       // g_pixelCount as the shader sees it
		for (int g = 0; g < groups; ++g) {
			float v[4][GROUP];
			for (int k = 0; k < 27; k += 3) {           // Gid.y = coefficient index k/3
				for (int l = 0; l < GROUP; ++l) {
					const int id = g * GROUP + l;
					const bool ok = id < count && (size_t)id * 27 + k + 2 < S[src]->size();
					for (int ch = 0; ch < 3; ++ch) v[ch][l] = ok ? (*S[src])[(size_t)id * 27 + k + ch] : 0.0f;
					v[3][l] = (ok && k == 0 && (size_t)id < W[src]->size()) ? (*W[src])[id] : 0.0f;
				}
				for (int ch = 0; ch < 3; ++ch) (*S[dst])[(size_t)g * 27 + k + ch] = tree32(v[ch]);
				if (k == 0) (*W[dst])[g] = tree32(v[3]);
			}
		}
		src = dst;
	}

	// ---- CSSHNormalize ------------------------------------------------------------------------
	const float w0 = (*W[src])[0];
	const float norm = 0.0f < w0 ? 12.566371f / w0 : 0.0f;
	for (int k = 0; k < 27; ++k) out27[k] = norm * (*S[src])[k];
}

void orc_sh_irradiance(const float* sh27, const float n[3], float out[3])
{
	// SHIrradianceTypeless.hlsli:16-37 -- same association order as orc_render.cpp's copy
	const float c1 = 0.429042757f, c3 = 0.247707963f, c4 = 0.886226952f, c1x2 = 0.858085513f, c2x2 = 1.02332675f;
	const float a = std::fmaf(n[0], n[0], -(n[1] * n[1])) * c1;
	const float b = std::fmaf(n[2] * n[2], 3.0f, -1.0f) * c3;
	const float mx = -n[0], my = -n[1], z = n[2];
	for (int k = 0; k < 3; ++k) {
		const float* L = sh27 + k;
		float r = L[18] * b;
		r = std::fmaf(a, L[24], r);
		r = std::fmaf(L[0], c4, r);
		float q = (L[21] * mx) * z;
		q = std::fmaf(L[12] * mx, my, q);
		q = std::fmaf(L[15] * my, z, q);
		r = std::fmaf(q, c1x2, r);
		float l = L[3] * my;
		l = std::fmaf(L[9], mx, l);
		l = std::fmaf(L[6], z, l);
		r = std::fmaf(l, c2x2, r);
		out[k] = std::fmax(r, 0.0f);
	}
}

}  // extern "C"
