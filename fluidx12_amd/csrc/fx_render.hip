// fx_render.hip -- gfx950 kernels of the cube-map-space ray march: the PLAIN path (every sample gathers its taps, like the
// reference's shaders) + the 2-D visualiser and the light-map decoder.  The accelerated path (occupancy masks, alpha side volume,
// compacted light voxels: fx_render_accel.hip) is the default; this one stays as its in-product yardstick (FX_OPT_RENDER_ACCEL 0:
// tests/test_gpu_render.py::test_empty_space_skipping_changes_no_bit compares the two bit for bit) and serves contexts whose
// acceleration structures could not be allocated.
//
//   k_raymarch_light  <- CSRayMarchL.hlsl:15-80   one thread per light-map voxel: shadow ray (+ GI/AO ray)
//   k_raymarch_view   <- CSRayMarch.hlsl:98-196   one thread per cube-map texel of mip `LOD`
//                        SEPARATE = CSRayMarchV.hlsl:5-7 (light = light-map fetch),
//                        otherwise the merged variant with the nested light march (RayMarch.hlsli:260-294)
//   k_raycast_direct  <- PSRayCast.hlsl:44-127 / PSRayCastV.hlsl (row f-2)
// (paths relative to /root/reference/FluidX12/Content/Shaders/; the march itself lives in fx_march.h).
// A wave64 = one 8x8 texel tile (view) or 64 consecutive x (light), so the taps of a wave are spatially coherent.
// Light map = packed R11G11B10_FLOAT like the reference (Fluid.cpp:226), cube map = R8G8B8A8_UNORM.
#include "fx_march.h"

namespace fx {

// ---------------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void k_raymarch_light(const Geom g, const typename ColTex<HALF>::T* __restrict__ col,
	uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh, uint32_t numSamples,
	unsigned long long* __restrict__ counters)
{
	const int x = blockIdx.x * 64 + threadIdx.x;
	const int y = blockIdx.y * 4 + threadIdx.y;
	const int z = blockIdx.z;
	if (x >= g.X || y >= g.Y) return;
	const PlainVol<HALF> vol{ col };
	const float ox = fmaf(((float)x + 0.5f) / (float)g.X, 2.0f, -1.0f);            // CSRayMarchL.hlsl:22
	const float oy = fmaf(((float)y + 0.5f) / (float)g.Y, 2.0f, -1.0f);
	const float oz = fmaf(((float)z + 0.5f) / (float)g.Zg, 2.0f, -1.0f);
	const float u = fmaf(ox, 0.5f, 0.5f), v = fmaf(oy, 0.5f, 0.5f), w = fmaf(oz, 0.5f, 0.5f);   // :36
	const float density = density_at(vol, g, u, v, w);                             // :37
	float shadow = 1.0f, ao = 1.0f, irr[3] = { 0.0f, 0.0f, 0.0f };
	uint32_t ns = 0;                                                               // (the voxel's own density sample is counted on the host: X Y Z of them)
	if (density >= 0.00999999978f) {                                               // :44
		const float stepScale = 3.46410155f / (float)numSamples;                   // RayMarch.hlsli:29-30
		float lx, ly, lz;
		light_dir_local(fc, lx, ly, lz);
		cast_light_ray<1>(shadow, g, vol, ox, oy, oz, lx, ly, lz, stepScale, numSamples, ns);   // :55
		if (sh) gi_term<1>(irr, ao, g, vol, fc, sh, ox, oy, oz, u, v, w, stepScale, numSamples, ns);   // :59-68
	}
	lightmap[((size_t)z * g.Y + y) * g.X + x] = light_value(fc, sh != nullptr, shadow, ao, irr);
	flush_counts(counters, 0u, ns, 0u);
}

template <bool HALF, bool SEPARATE>
__global__ __launch_bounds__(64) void k_raymarch_view(const Geom g, const typename ColTex<HALF>::T* __restrict__ col,
	const uint32_t* __restrict__ lightmap, const FrameConsts fc, const float* __restrict__ sh, int size, uint32_t mask,
	uint32_t numSamples, uint32_t numLightSamples, uint32_t* __restrict__ cube, unsigned long long* __restrict__ counters)
{
	const int face = blockIdx.z;
	if (!((mask >> face) & 1u)) return;                                            // CSRayMarch.hlsl:102
	const int x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
	if (x >= size || y >= size) return;
	float o[3], d[3], tMax;
	if (!cube_texel_ray(fc, face, x, y, size, o, d, tMax)) return;                 // :116
	const PlainVol<HALF> vol{ col };
	float sr, sg, sb, sa;
	uint32_t nv = 0, nl = 0, nm = 0;
	march_ray<PlainVol<HALF>, SEPARATE, 1>(g, vol, lightmap, fc, sh, o, d, tMax, numSamples, numLightSamples, true, sr, sg, sb, sa, nv, nl, nm);
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
	uint32_t numSamples, uint32_t numLightSamples, uint32_t* __restrict__ target, float4* __restrict__ out_float,
	unsigned long long* __restrict__ counters)
{
	const int px = blockIdx.x * 8 + threadIdx.x, py = blockIdx.y * 8 + threadIdx.y;
	if (px >= W || py >= H) return;
	const size_t pix = (size_t)py * W + px;
	if (out_float) out_float[pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	float o[3], d[3];
	if (!pixel_ray(fc, px, py, W, H, o, d)) return;                                // PSRayCast.hlsl:50 discard
	const PlainVol<HALF> vol{ col };
	float sr, sg, sb, sa;
	uint32_t nv = 0, nl = 0, nm = 0;
	march_ray<PlainVol<HALF>, SEPARATE, 1>(g, vol, lightmap, fc, sh, o, d, 3.40282347e+38f, numSamples, numLightSamples, true, sr, sg, sb, sa, nv, nl, nm);
	flush_counts(counters, nv, nl, nm);
	sr *= 0.159154937f; sg *= 0.159154937f; sb *= 0.159154937f;                    // :124
	if (out_float) out_float[pix] = make_float4(sr, sg, sb, sa);
	if (target) target[pix] = blend_premultiplied(target[pix], sr, sg, sb, sa);
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
	const PlainVol<HALF> vol{ col };
	float4 c = vol.color(make_taps(g, make_base(g, fmaf(u, 1.0f, 0.0f), fmaf(v, -1.0f, 1.0f), 0.5f)));
	c.x = c.x / (c.x + 0.5f); c.y = c.y / (c.y + 0.5f); c.z = c.z / (c.z + 0.5f);
	if (out_float) out_float[pix] = c;
	if (target) target[pix] = blend_premultiplied(target[pix], c.x, c.y, c.z, c.w);
}

__global__ __launch_bounds__(256) void k_lightmap_decode(const uint32_t* __restrict__ lm, float* __restrict__ out, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		const float3 c = unpack_r11g11b10(lm[i]);
		out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
	}
}

// ---------------------------------------------------------------------------------------------------
hipError_t launch_raymarch_light(const Geom& g, int half_store, const void* color, uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, uint32_t num_samples, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((g.X + 63) / 64, (g.Y + 3) / 4, g.Zg), block(64, 4, 1);
	if (half_store) hipLaunchKernelGGL(k_raymarch_light<true>, grid, block, 0, s, g, (const h16x4*)color, lightmap, fc, sh, num_samples, counters);
	else hipLaunchKernelGGL(k_raymarch_light<false>, grid, block, 0, s, g, (const float4*)color, lightmap, fc, sh, num_samples, counters);
	return hipGetLastError();
}

hipError_t launch_raymarch_view(const Geom& g, int half_store, const void* color, const uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, int cube_size, uint32_t mask, uint32_t num_samples,
	uint32_t num_light_samples, int separate, uint8_t* cube, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((cube_size + 7) / 8, (cube_size + 7) / 8, 6), block(8, 8, 1);
	uint32_t* out = reinterpret_cast<uint32_t*>(cube);
#define FX_LAUNCH(H, S) hipLaunchKernelGGL((k_raymarch_view<H, S>), grid, block, 0, s, g, \
	(const typename ColTex<H>::T*)color, lightmap, fc, sh, cube_size, mask, num_samples, num_light_samples, out, counters)
	if (half_store) { if (separate) FX_LAUNCH(true, true); else FX_LAUNCH(true, false); }
	else { if (separate) FX_LAUNCH(false, true); else FX_LAUNCH(false, false); }
#undef FX_LAUNCH
	return hipGetLastError();
}

hipError_t launch_raycast_direct(const Geom& g, int half_store, const void* color, const uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, int W, int H, uint32_t num_samples, uint32_t num_light_samples, int separate,
	uint8_t* target, float* out_float, hipStream_t s, unsigned long long* counters)
{
	const dim3 grid((W + 7) / 8, (H + 7) / 8, 1), block(8, 8, 1);
#define FX_LAUNCH(HF, S) hipLaunchKernelGGL((k_raycast_direct<HF, S>), grid, block, 0, s, g, \
	(const typename ColTex<HF>::T*)color, lightmap, fc, sh, W, H, num_samples, num_light_samples, \
	reinterpret_cast<uint32_t*>(target), reinterpret_cast<float4*>(out_float), counters)
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
