// oracle/fx_oracle.h -- C entry points of the CPU oracle (liborc.so).
//
// TEST INFRASTRUCTURE ONLY: a scalar C++ restatement of the reference's HLSL/host arithmetic for
// the hot path (see orc_common.h for the conventions and oracle/README.md for how it is pinned).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
// product (fluidx12_amd/, include/fluidx_hip.h) never does.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

// ---- simulation (orc_sim.cpp) -------------------------------------------------------------
// layouts: velocity float[3][Z][Y][X]; colour float[Z][Y][X][4]; scalars float[Z][Y][X]
// address_mode: 0 = CLAMP (FluidEZ.cpp:406, the reference's default path), 1 = MIRROR (Fluid.cpp:452)
void orc_advect(const float* vel_in, const float* col_in, float* vel_out, float* col_out,
	int X, int Y, int Z, float dt, int address_mode, int half_storage);
void orc_divergence(const float* vel, float* b, int X, int Y, int Z);
long long orc_jacobi_sweep(const float* p_in, const float* b, float* p_out, uint8_t* frozen, int X, int Y, int Z);
int orc_jacobi(float* p, const float* b, float* tmp, int X, int Y, int Z, int iters, int mode);
void orc_project(const float* vel_in, const float* p, float* vel_out, int X, int Y, int Z, int half_storage);
void orc_simulate(float* vel0, float* vel1, const float* col_src, float* col_dst, float* p, float* b, float* tmp,
	int X, int Y, int Z, float dt, int iters, int mode, int address_mode, int half_storage);
void orc_quantize_half(const float* in, float* out, long long n);
uint16_t orc_f32_to_f16(float f);
float orc_f16_to_f32(uint16_t h);

// ---- per-frame host rules (orc_host.cpp) --------------------------------------------------
// DirectXMath conventions: row vectors (v * M), row-major float[16].
typedef struct orc_frame {
	float world_i[12];      // XMFLOAT3X4 of inverse(world): 3 rows of 4 (Fluid.cpp:320)
	float world[12];        // XMFLOAT3X4 of world (Fluid.cpp:321), world = scale 10 (Fluid.cpp:182)
	float eye_pt[3];        // world-space eye (CBPerFrame.EyePos, Fluid.cpp:304)
	float light_pt[3];      // (75, 75, -75) (Fluid.cpp:169)
	float light_color[4];   // (1, .7, .3, 3 pi) (Fluid.cpp:170)
	float ambient[4];       // (1, 1, 1, 1.5 pi) (Fluid.cpp:173)
	float sh[27];           // 9 x float3 SH coefficients (only read when hasSH)
} orc_frame;

void orc_look_at_lh(const float eye[3], const float focus[3], const float up[3], float out16[16]);
void orc_perspective_fov_lh(float fovy, float aspect, float zn, float zf, float out16[16]);
// Fluid::UpdateFrame (Fluid.cpp:283-346): fills the frame constants, cube-map LOD (EstimateCubeMapLOD
// :141-166), clamped ray sample count and the face visibility mask (GenVisibilityMask :49-61).
void orc_update_frame(const float view[16], const float proj[16], const float eye[3],
	uint32_t viewport_w, uint32_t viewport_h, uint32_t grid_x, uint32_t max_ray_samples,
	orc_frame* frame, uint32_t* lod, uint32_t* ray_samples, uint32_t* mask, float* edge_px);

// ---- ray march (orc_render.cpp) -----------------------------------------------------------
// light map float[Z][Y][X][3]; light_fmt 0 fp32 / 1 half / 2 R11G11B10F (reference, Fluid.cpp:226)
void orc_raymarch_light(const float* color, float* lightmap, int X, int Y, int Z, const orc_frame* fc,
	uint32_t numSamples, int hasSH, int light_fmt);
// cube maps [6][size][size][4]; either output may be null
void orc_raymarch_view(const float* color, const float* lightmap, int X, int Y, int Z, const orc_frame* fc,
	int size, uint32_t mask, uint32_t numSamples, uint32_t numLightSamples, int hasSH, int separate,
	float* cube_f32, uint8_t* cube_u8);
// direct screen-space march (row f-2; PSRayCast.hlsl / PSRayCastV.hlsl), one ray per pixel
void orc_raycast_direct(const float* color, const float* lightmap, int X, int Y, int Z, const orc_frame* fc,
	const float* wvp_i, int W, int H, uint32_t numSamples, uint32_t numLightSamples, int hasSH, int separate,
	float* out_rgba, uint8_t* covered);
// 2-D visualiser (PSVisualizeColor.hlsl:24-33): color float[Y][X][4] -> premultiplied float[H][W][4]
void orc_visualize_color(const float* color, int X, int Y, int W, int H, float* out_rgba);
uint32_t orc_pack_r11g11b10(float r, float g, float b);
void orc_unpack_r11g11b10(uint32_t v, float* rgb);

// ---- cube map -> screen resolve (orc_resolve.cpp; PSRayCastCube.hlsl:20-113, PSCube.hlsli:41-122) ----------
// cube: RGBA8 [6][N][N][4] (the mip the view pass wrote); wvp_i: orc_world_view_proj_inverse; out_rgba float[H][W][4]
// premultiplied (zeros where discarded), covered uint8[H][W]
void orc_world_view_proj_inverse(const float view[16], const float proj[16], float out16[16]);
void orc_resolve_cube(const uint8_t* cube, int N, const orc_frame* fc, const float* wvp_i, int W, int H,
	float* out_rgba, uint8_t* covered);
// sky pass (PSEnvironment.hlsl): float radiance cube [6][N][N][3], world-space eye, screenToWorld rows -> float[H][W][4]
void orc_environment(const float* cube, int N, const float eye[3], const float* s2w, int W, int H, float* out_rgba);
// PREMULTIPLIED blend (Fluid.cpp:653) of a resolve result over an R8G8B8A8_UNORM target (FluidX12.cpp:31), in place
void orc_blend_premultiplied(const float* src_rgba, const uint8_t* covered, uint8_t* target_rgba8, int W, int H);

// ---- BC6H_UF16 + DDS cube container (orc_bc6h.cpp): the radiance asset path of LightProbe::Init (LightProbe.cpp:41-46) ----
int orc_bc6h_decode_block(const uint8_t* block16, uint16_t* out_half_rgb /* [16][3] */);
int orc_dds_bc6h_cube_face(const uint8_t* dds, size_t bytes, int face, int mip, float* out_rgb, int* mode_hist);

// ---- spherical harmonics light probe (orc_sh.cpp) -----------------------------------------
// cube float[6][N][N][3]; out float[9][3].  quirk != 0 reproduces LightProbeEZ.cpp:245-246
// (every sum pass sees the first pass's element count, so pass 3 re-adds 20 stale partials).
void orc_sh_transform(const float* cube, int N, float* out27, int quirk);
void orc_sh_irradiance(const float* sh27, const float n[3], float out[3]);

#ifdef __cplusplus
}
#endif
