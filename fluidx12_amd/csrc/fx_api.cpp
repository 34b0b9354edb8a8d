// fx_api.cpp -- the per-frame entry points of the C ABI declared in include/fluidx_hip.h: the HIP re-statement of class Fluid's
// host side (/root/reference/FluidX12/Content/Fluid.cpp): UpdateFrame, Simulate, Render, the caller-side cube resolve, the SH probe,
// and the slab-group set-up.  The command list becomes a HIP stream; the 3-slot upload constant buffers (Fluid.cpp:239-252) become
// kernel arguments passed by value (no host/device hazard, so `frame_index` only needs range-checking).  Contexts and fields:
// fx_context.cpp; the step's schedule: fx_schedule.cpp.
//
// No CPU fallback and nothing from oracle/: every field operation is a HIP kernel.
#include "fx_host.h"
#include <cstring>
#include "fx_hostmath.h"

using namespace fx;
using namespace fxh;

// ===================================================================================================
extern "C" {

int fx_abi_version(void) { return FX_ABI_VERSION; }

int fx_set_max_samples(fx_ctx* ctx, uint32_t max_ray, uint32_t max_light)
{
	if (!ctx || !max_ray || !max_light) return FX_E_INVALID;
	ctx->max_ray_samples = max_ray;
	ctx->max_light_samples = max_light;
	return FX_OK;
}

int fx_set_sh(fx_ctx* ctx, const float* coeffs27)
{
	if (!ctx) return FX_E_INVALID;
	if (!ctx->sh_dev) return FX_E_STATE;
	if (!coeffs27) { ctx->has_sh = false; return FX_OK; }
	DeviceGuard dg(ctx->device);
	FX_HIP(hipMemcpy(ctx->sh_dev, coeffs27, 27 * sizeof(float), hipMemcpyHostToDevice));
	if (ctx->accel_ok && !ctx->accel.gi)               // scratch of the occlusion rays (fx_render_accel.hip); without it the render takes the chunked march
		if (hipMalloc((void**)&ctx->accel.gi, 3 * ctx->g.cells_owned() * sizeof(float)) != hipSuccess) { ctx->accel.gi = nullptr; (void)hipGetLastError(); }
	ctx->has_sh = true;
	return FX_OK;
}

int fx_update_frame(fx_ctx* ctx, float time_step, uint8_t frame_index,
	const float view[16], const float proj[16], const float eye[3])
{
	if (!ctx || frame_index >= FX_FRAME_COUNT) return FX_E_INVALID;
	std::vector<fx_ctx*> M;
	for_members(ctx, M);
	if (!is_driver(ctx) || group_broken(ctx)) return FX_E_STATE;
	for (fx_ctx* c : M) {
		if (c->g.Zg > 1 && view && proj && eye && c->desc.viewport_w && c->desc.viewport_h) {   // a context without a viewport only simulates
			// Fluid.cpp:296-334
			const Mat4 world = Mat4::scaling(10.0f, 10.0f, 10.0f);              // m_volumeWorld, Fluid.cpp:182
			const Mat4 worldI = world.inverse();
			const Mat4 wvp = world * (Mat4::from(view) * Mat4::from(proj));
			worldI.store3x4(c->fc.world_i);
			world.store3x4(c->fc.world);
			const Mat4 vpI = (Mat4::from(view) * Mat4::from(proj)).inverse();   // LightProbe::UpdateFrame (LightProbe.cpp:70-76)
			for (int r = 0; r < 4; ++r)
				for (int q = 0; q < 4; ++q) c->fc.s2w[r * 4 + q] = vpI.m[q][r];
			const Mat4 wvpI = wvp.inverse();                                    // stored transposed (Fluid.cpp:318)
			for (int r = 0; r < 4; ++r)
				for (int q = 0; q < 4; ++q) c->fc.wvp_i[r * 4 + q] = wvpI.m[q][r];
			for (int a = 0; a < 3; ++a) c->fc.eye_pt[a] = eye[a];
			const float pi = 3.141592654f;
			const float lp[3] = { 75.0f, 75.0f, -75.0f };                        // Fluid.cpp:169-173
			const float lc[4] = { 1.0f, 0.7f, 0.3f, pi * 3.0f }, am[4] = { 1.0f, 1.0f, 1.0f, pi * 1.5f };
			std::memcpy(c->fc.light_pt, lp, sizeof lp);
			std::memcpy(c->fc.light_color, lc, sizeof lc);
			std::memcpy(c->fc.ambient, am, sizeof am);

			// EstimateCubeMapLOD (Fluid.cpp:141-166)
			static const Vec3 corners[8] = { {1,1,1},{-1,1,1},{1,-1,1},{-1,-1,1},{-1,1,-1},{1,1,-1},{-1,-1,-1},{1,-1,-1} };
			static const uint8_t edges[12][2] = { {0,1},{3,2},{1,3},{2,0},{4,5},{7,6},{5,7},{6,4},{1,4},{6,3},{5,0},{2,7} };
			float sx[8], sy[8];
			for (int i = 0; i < 8; ++i) {
				const Vec3 q = wvp.transform_coord(corners[i]);
				sx[i] = (q.x * 0.5f + 0.5f) * (float)c->desc.viewport_w;
				sy[i] = (q.y * -0.5f + 0.5f) * (float)c->desc.viewport_h;
			}
			float edge = 0.0f;
			for (const auto& e : edges) {
				const float ex = sx[e[1]] - sx[e[0]], ey = sy[e[1]] - sy[e[0]];
				edge = std::max(std::sqrt(ex * ex + ey * ey), edge);
			}
			c->edge_pixels = edge;
			float s = edge / 2.0f;
			float amount = 2.0f * s / std::sqrt(3.0f);
			const uint32_t wanted = (uint32_t)std::ceil(amount);
			c->ray_samples = std::min(wanted, c->max_ray_samples);
			amount = std::min(amount, (float)c->ray_samples);
			s = amount / 2.0f * std::sqrt(3.0f);
			const uint8_t level = (uint8_t)std::max(std::log2((float)c->g.X / s), 0.0f);
			c->cube_lod = std::min<uint32_t>(level, kNumMips - 1);

			// GenVisibilityMask (Fluid.cpp:49-61)
			uint32_t mask = 0;
			for (uint32_t f = 0; f < 6; ++f) {
				const float v = worldI.transform_comp(eye, (int)(f >> 1));
				mask |= ((f & 1u) ? v > -1.0f : v < 1.0f) ? 1u << f : 0u;
			}
			c->visibility_mask = mask;
			c->view_valid = true;
		}
		c->time_step = time_step;
		if (time_step > 0.0f && !(c->desc.flags & FX_FLAG_RENDER_ONLY)) c->frame_parity ^= 1;   // Fluid.cpp:345
		c->frame_valid = true;
	}
	return FX_OK;
}

int fx_simulate(fx_ctx* ctx, void* stream, uint8_t frame_index)
{
	if (!ctx || frame_index >= FX_FRAME_COUNT) return FX_E_INVALID;
	if (!ctx->frame_valid || (ctx->desc.flags & FX_FLAG_RENDER_ONLY) || group_broken(ctx)) return FX_E_STATE;
	if (!is_driver(ctx)) return FX_OK;             // loop-back group: rank 0 drives every member
	ctx->last_step_stream = pick_stream(ctx, stream);
	return simulate_impl(ctx, ctx->last_step_stream);
}

// the zero fill is ordered on the stream the target is about to be used on (the context's streams do not synchronise
// with the NULL stream)
static int ensure_target(fx_ctx* ctx, hipStream_t s)
{
	if (ctx->target) return FX_OK;
	const size_t n = (size_t)ctx->desc.viewport_w * ctx->desc.viewport_h;
	if (!n) return FX_E_INVALID;
	FX_HIP(hipMalloc((void**)&ctx->target, n * 4));
	FX_HIP(hipMemsetAsync(ctx->target, 0, n * 4, s));
	FX_HIP(hipMalloc((void**)&ctx->target_float, n * 16));
	FX_HIP(hipMemsetAsync(ctx->target_float, 0, n * 16, s));
	return FX_OK;
}

// The marches of one fx_render.  FX_OPT_RENDER_ACCEL (default): the acceleration structures of this frame's colour field are built
// first, inside the first pass's timing mark -- their cost belongs to the frame.
struct Marches {
	fx_ctx* c; const void* color; hipStream_t s; unsigned long long* cnt; bool accel, built, filled;
	Marches(fx_ctx* ctx, const void* col, hipStream_t st, unsigned long long* counters)
		: c(ctx), color(col), s(st), cnt(counters), accel(ctx->opt_render_accel && ctx->accel_ok), built(false), filled(false) {}
	const float* sh() const { return c->has_sh ? c->sh_dev : nullptr; }
	hipError_t build(bool for_light = false)
	{
		if (!accel || built) return hipSuccess;
		built = true;
		// (the side volume counts as current only when the advection that made this field wrote it: a full build also leaves it current,
		// but a render loop over a paused field would then time a cheaper build than a frame behind its own step gets -- not worth the ambiguity)
		const bool current = c->accel_alpha_of == color;
		if (!current) c->accel_alpha_of = nullptr;                   // the build overwrites the side volume with this field's alpha
		c->rendered_since_step = true; c->rendered_on = s;
		c->accel.frame += 1;                                         // this render's set of counters
		// the filling pass may skip the cells whose light-map words nobody has touched since the last one wrote the same constant there
		float key[9];
		for (int i = 0; i < 4; ++i) { key[i] = c->fc.light_color[i]; key[4 + i] = c->fc.ambient[i]; }
		key[8] = c->has_sh ? 1.0f : 0.0f;
		const bool same = c->lightmap_filled && std::memcmp(key, c->lightmap_key, sizeof key) == 0;
		const LightFill lf{ c->lightmap, &c->fc, c->has_sh ? 1 : 0, same };
		if (for_light) c->accel.fill_frame += 1;
		const hipError_t e = launch_accel_build(c->g, c->half, color, c->accel, s, current, for_light ? &lf : nullptr, &filled);
		if (for_light) {
			if (!filled) c->accel.fill_frame -= 1;
			c->lightmap_filled = filled && e == hipSuccess;          // (another light path writes every word: no record)
			std::memcpy(c->lightmap_key, key, sizeof key);
		}
		return e;
	}
	hipError_t light()                                  // Fluid.cpp:857-878
	{
		hipError_t e = build(true);
		if (e != hipSuccess) return e;
		if (accel) return launch_accel_light(c->g, c->accel, c->lightmap, c->fc, sh(), c->max_light_samples, s, cnt, filled);
		c->lightmap_filled = false;
		return launch_raymarch_light(c->g, c->half, color, c->lightmap, c->fc, sh(), c->max_light_samples, s, cnt);
	}
	hipError_t view(int size, uint8_t* cube, bool separate)   // Fluid.cpp:880-908 (separate) / :825-855 (merged)
	{
		hipError_t e = build();
		if (e != hipSuccess) return e;
		const uint32_t ns = c->ray_samples;
		if (accel) return launch_accel_view(c->g, c->half, color, separate ? c->lightmap : nullptr, c->fc, separate ? nullptr : sh(), size,
			c->visibility_mask, ns, c->max_light_samples, separate, cube, c->accel, s, cnt);
		return launch_raymarch_view(c->g, c->half, color, separate ? c->lightmap : nullptr, c->fc, separate ? nullptr : sh(), size,
			c->visibility_mask, ns, c->max_light_samples, separate, cube, s, cnt);
	}
	hipError_t direct(int W, int H, bool separate)      // rayCastVDirect Fluid.cpp:953-972 / rayCastDirect :932-951
	{
		hipError_t e = build();
		if (e != hipSuccess) return e;
		const uint32_t ns = separate ? c->ray_samples : c->max_ray_samples;
		if (accel) return launch_accel_direct(c->g, c->half, color, separate ? c->lightmap : nullptr, c->fc, separate ? nullptr : sh(), W, H,
			ns, c->max_light_samples, separate, c->target, c->target_float, c->accel, s, cnt);
		return launch_raycast_direct(c->g, c->half, color, separate ? c->lightmap : nullptr, c->fc, separate ? nullptr : sh(), W, H,
			ns, c->max_light_samples, separate, c->target, c->target_float, s, cnt);
	}
};

int fx_render(fx_ctx* ctx, void* stream, uint8_t frame_index, uint8_t flags)
{
	if (!ctx || frame_index >= FX_FRAME_COUNT) return FX_E_INVALID;
	if (ctx->g.Zg <= 1) {                          // Fluid.cpp:445: else visualizeColor(pCommandList)
		if (!ctx->frame_valid) return FX_E_STATE;
		DeviceGuard dg2(ctx->device);
		hipStream_t s2 = pick_stream(ctx, stream);
		int rc = ensure_target(ctx, s2);
		if (rc) return rc;
		ScopedMark mk(ctx, s2, MK_VIEW);
		FX_HIP(launch_visualize_color(ctx->g, ctx->half, ctx->col[ctx->frame_parity], (int)ctx->desc.viewport_w,
			(int)ctx->desc.viewport_h, ctx->target, ctx->target_float, s2));
		if (ctx->timing_on) ctx->acc.renders += 1;
		return FX_OK;
	}
	if (!ctx->desc.viewport_w || !ctx->desc.viewport_h) return FX_E_INVALID;   // created without a viewport: nothing to project the cube onto
	if (!ctx->view_valid) return FX_E_STATE;
	if (ctx->g.nz != ctx->g.Zg) return FX_E_INVALID;            // rays cross slabs: multi-GPU rendering is row f-3
	// (tap indices are 32 bits wide and put together by 24-bit multiplies, fx_march.h make_taps)
	if ((uint64_t)ctx->g.Y * ((uint64_t)ctx->g.Zg + 1) >= (1u << 24) || ctx->g.cells_owned() >= ((size_t)1 << 32)) return FX_E_INVALID;
	unsigned long long* cnt = ctx->opt_count_samples ? ctx->sample_counters : nullptr;
	if (cnt && (flags & FX_SEPARATE_LIGHT_PASS)) ctx->acc.light_samples += (uint64_t)ctx->g.cells_owned();   // the light pass's own fetch per voxel
	DeviceGuard dg(ctx->device);
	hipStream_t s = pick_stream(ctx, stream);
	Marches m(ctx, ctx->col[ctx->frame_parity], s, cnt);
	const bool separate = (flags & FX_SEPARATE_LIGHT_PASS) != 0;
	if (!(flags & FX_RAY_MARCH_CUBEMAP)) {
		// direct screen-space marching (Fluid.cpp:432-443): one ray per pixel, straight onto the render target
		int rc = ensure_target(ctx, s);
		if (rc) return rc;
		if (separate) {
			ScopedMark mk(ctx, s, MK_LIGHT);
			FX_HIP(m.light());
		}
		ScopedMark mk(ctx, s, MK_VIEW);
		FX_HIP(m.direct((int)ctx->desc.viewport_w, (int)ctx->desc.viewport_h, separate));
		if (ctx->timing_on) ctx->acc.renders += 1;
		return FX_OK;
	}
	const int size = ctx->g.X >> ctx->cube_lod;
	uint8_t* cube = ctx->cube + ctx->cube_mip_offset[ctx->cube_lod];
	if (separate) {
		ScopedMark mk(ctx, s, MK_LIGHT);
		FX_HIP(m.light());
	}
	{
		ScopedMark mk(ctx, s, MK_VIEW);
		FX_HIP(m.view(size, cube, separate));
	}
	if (ctx->timing_on) ctx->acc.renders += 1;
	return FX_OK;
}

int fx_clear_render_target(fx_ctx* ctx, void* stream, const float rgba[4])
{
	if (!ctx || !rgba) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	int rc = ensure_target(ctx, pick_stream(ctx, stream));
	if (rc) return rc;
	FX_HIP(launch_clear_target(ctx->target, (int)ctx->desc.viewport_w, (int)ctx->desc.viewport_h, rgba, pick_stream(ctx, stream)));
	return FX_OK;
}

int fx_render_cube(fx_ctx* ctx, void* stream, uint8_t frame_index)
{
	if (!ctx || frame_index >= FX_FRAME_COUNT) return FX_E_INVALID;
	if (ctx->g.Zg <= 1 || !ctx->cube) return FX_E_INVALID;
	if (!ctx->view_valid) return FX_E_STATE;
	if (ctx->g.nz != ctx->g.Zg) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	hipStream_t s = pick_stream(ctx, stream);
	int rc = ensure_target(ctx, s);
	if (rc) return rc;
	ScopedMark mk(ctx, s, MK_RESOLVE);
	FX_HIP(launch_resolve_cube(ctx->cube + ctx->cube_mip_offset[ctx->cube_lod], ctx->g.X >> ctx->cube_lod, ctx->fc,
		(int)ctx->desc.viewport_w, (int)ctx->desc.viewport_h, ctx->target, ctx->target_float, s));
	return FX_OK;
}

int fx_set_environment(fx_ctx* ctx, const float* cube, uint32_t n)
{
	if (!ctx || (cube && (!n || n > 8192))) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	FX_HIP(hipDeviceSynchronize());
	if (ctx->env) { FX_HIP(hipFree(ctx->env)); ctx->env = nullptr; ctx->env_n = 0; }
	if (!cube) return FX_OK;
	const size_t bytes = (size_t)6 * n * n * 3 * sizeof(float);
	FX_HIP(hipMalloc((void**)&ctx->env, bytes));
	FX_HIP(hipMemcpy(ctx->env, cube, bytes, hipMemcpyHostToDevice));
	ctx->env_n = n;
	return FX_OK;
}

int fx_render_environment(fx_ctx* ctx, void* stream, uint8_t frame_index)
{
	if (!ctx || frame_index >= FX_FRAME_COUNT) return FX_E_INVALID;
	if (!ctx->env || !ctx->view_valid) return FX_E_STATE;
	DeviceGuard dg(ctx->device);
	hipStream_t s = pick_stream(ctx, stream);
	int rc = ensure_target(ctx, s);
	if (rc) return rc;
	ScopedMark mk(ctx, s, MK_RESOLVE);
	FX_HIP(launch_environment(ctx->env, (int)ctx->env_n, ctx->fc, (int)ctx->desc.viewport_w, (int)ctx->desc.viewport_h,
		ctx->target, ctx->target_float, s));
	return FX_OK;
}

int fx_get_frame_info(fx_ctx* ctx, fx_frame_info* out)
{
	if (!ctx || !out) return FX_E_INVALID;
	out->cube_lod = ctx->cube_lod;
	out->cube_size = (uint32_t)ctx->g.X >> ctx->cube_lod;
	out->ray_samples = ctx->ray_samples;
	out->visibility_mask = ctx->visibility_mask;
	out->frame_parity = ctx->frame_parity;
	out->edge_pixels = ctx->edge_pixels;
	out->time_step = ctx->time_step;
	std::memcpy(out->world_view_proj_i, ctx->fc.wvp_i, sizeof out->world_view_proj_i);
	std::memcpy(out->screen_to_world, ctx->fc.s2w, sizeof out->screen_to_world);
	return FX_OK;
}

// ---- individual stages (parity tests, micro-benchmarks) ---------------------------------------------
int fx_advect(fx_ctx* ctx, void* stream)
{
	if (!ctx) return FX_E_INVALID;
	if (ctx->desc.flags & FX_FLAG_RENDER_ONLY) return FX_E_STATE;
	if (!ctx->frame_valid) return FX_E_STATE;
	if (ctx->nranks > 1) return FX_E_INVALID;
	hipStream_t s = pick_stream(ctx, stream);
	ScopedMark mk(ctx, s, MK_ADVECT);
	return advect_range(ctx, s, owned(ctx), false);
}

int fx_divergence(fx_ctx* ctx, void* stream)
{
	if (!ctx || ctx->nranks > 1) return FX_E_INVALID;
	if (ctx->desc.flags & FX_FLAG_RENDER_ONLY) return FX_E_STATE;
	return divergence_phase(ctx, pick_stream(ctx, stream));
}

int fx_jacobi(fx_ctx* ctx, void* stream, uint32_t iters)
{
	if (!ctx || !iters || ctx->nranks > 1) return FX_E_INVALID;
	if (ctx->desc.flags & FX_FLAG_RENDER_ONLY) return FX_E_STATE;
	std::vector<fx_ctx*> M{ ctx };
	return jacobi_all(ctx, M, pick_stream(ctx, stream), iters);
}

int fx_project(fx_ctx* ctx, void* stream)
{
	if (!ctx || ctx->nranks > 1) return FX_E_INVALID;
	if (!ctx->frame_valid || (ctx->desc.flags & FX_FLAG_RENDER_ONLY)) return FX_E_STATE;
	return project_phase(ctx, pick_stream(ctx, stream));
}

// ---- SH light probe -----------------------------------------------------------------------------------
int fx_sh_transform(fx_ctx* ctx, const float* cube, uint32_t n, float* out27)
{
	if (!ctx || !cube || !out27 || !n || n > 4096) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	FX_HIP(hipDeviceSynchronize());
	const size_t cubeBytes = (size_t)6 * n * n * 3 * sizeof(float);
	int rc = ensure_stage(ctx, cubeBytes);
	if (rc) return rc;
	if (ctx->sh_scratch_n != n) {
		for (int i = 0; i < 4; ++i) {
			if (ctx->sh_scratch[i]) { FX_HIP(hipFree(ctx->sh_scratch[i])); ctx->sh_scratch[i] = nullptr; }
			FX_HIP(hipMalloc((void**)&ctx->sh_scratch[i], sh_scratch_floats((int)n, i) * sizeof(float)));
		}
		ctx->sh_scratch_n = n;
	}
	float* d_out = nullptr;
	FX_HIP(hipMalloc((void**)&d_out, 27 * sizeof(float)));
	FX_HIP(hipMemcpy(ctx->stage, cube, cubeBytes, hipMemcpyHostToDevice));
	hipError_t e = launch_sh_transform(ctx->stage, (int)n, ctx->sh_scratch[0], ctx->sh_scratch[1], ctx->sh_scratch[2],
		ctx->sh_scratch[3], d_out, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess) e = hipMemcpy(out27, d_out, 27 * sizeof(float), hipMemcpyDeviceToHost);
	(void)hipFree(d_out);
	if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return FX_E_DEVICE; }
	return FX_OK;
}

// ---- timing ---------------------------------------------------------------------------------------------
int fx_dds_cube_info(const void* dds, size_t bytes, uint32_t* size, uint32_t* mips)
{
	DdsCube d;
	if (!dds_cube_layout(dds, bytes, &d)) return FX_E_INVALID;
	if (size) *size = d.size;
	if (mips) *mips = d.mips;
	return FX_OK;
}

int fx_dds_decode_cube(fx_ctx* ctx, const void* dds, size_t bytes, uint32_t mip, float* out_cube, size_t out_floats)
{
	if (!ctx || !out_cube) return FX_E_INVALID;
	DdsCube d;
	if (!dds_cube_layout(dds, bytes, &d) || mip >= d.mips) return FX_E_INVALID;
	const size_t* fo = d.face_offset; const size_t* mo = d.mip_offset;
	const uint32_t n = std::max<uint32_t>(d.size >> mip, 1), nb = (n + 3) / 4;
	const size_t face_floats = (size_t)n * n * 3, face_blocks = (size_t)nb * nb * 16;
	if (out_floats != 6 * face_floats) return FX_E_INVALID;
	if (d.kind != DDS_BC6H_UF16) {                                          // uncompressed: a host-side format conversion
		for (int f = 0; f < 6; ++f) dds_linear_face_to_rgb(static_cast<const char*>(dds) + fo[f] + mo[mip], d.kind, n, out_cube + f * face_floats);
		return FX_OK;
	}
	DeviceGuard dg(ctx->device);
	const size_t float_bytes = (6 * face_floats * 4 + 15) & ~(size_t)15;     // the blocks are read as 16-byte words
	int rc = ensure_stage(ctx, float_bytes + 6 * face_blocks);
	if (rc) return rc;
	char* blocks_dev = reinterpret_cast<char*>(ctx->stage) + float_bytes;
	for (int f = 0; f < 6; ++f)
		FX_HIP(hipMemcpyAsync(blocks_dev + f * face_blocks, static_cast<const char*>(dds) + fo[f] + mo[mip], face_blocks, hipMemcpyHostToDevice, ctx->stream));
	for (int f = 0; f < 6; ++f)
		FX_HIP(launch_bc6h_decode(blocks_dev + f * face_blocks, (int)nb, (int)nb, (int)n, ctx->stage + f * face_floats, ctx->stream));
	FX_HIP(hipMemcpyAsync(out_cube, ctx->stage, 6 * face_floats * 4, hipMemcpyDeviceToHost, ctx->stream));
	FX_HIP(hipStreamSynchronize(ctx->stream));
	return FX_OK;
}

// ---- multi-GPU ---------------------------------------------------------------------------------------------
size_t fx_comm_id_bytes(void) { return rccl_id_bytes(); }

int fx_comm_get_unique_id(void* id_out, size_t bytes)
{
	if (!id_out) return FX_E_INVALID;
	std::string err;
	const int rc = rccl_get_unique_id(id_out, bytes, &err);
	if (rc) std::fprintf(stderr, "fluidx: %s\n", err.c_str());
	return rc;
}

static int check_slab_chain(fx_ctx* c, int rank, int nranks)
{
	if (nranks < 1 || rank < 0 || rank >= nranks) return FX_E_INVALID;
	if (c->group) return FX_E_STATE;
	if (nranks > 1 && c->g.nz == c->g.Zg) return FX_E_INVALID;     // a slab context is required
	if (rank == 0 && c->g.z0 != 0) return FX_E_INVALID;
	if (rank == nranks - 1 && c->g.z0 + c->g.nz != c->g.Zg) return FX_E_INVALID;
	return FX_OK;
}

// a lane: side streams + ordering events of one rank (fx_context.h); `compute` = the member's own stream in peer groups, else null
static int make_lane(fx_comm_group* g, int device, hipStream_t compute)
{
	fx_lane l{};
	l.device = device;
	l.compute = compute;
	DeviceGuard dg(device);
	// DEFAULT priority.  A high-priority side stream made every dependency between it and the compute stream cost about a
	// millisecond for the first group of a process (18 ms per step instead of 7.7, docs/LAB.md); the switch COMM_PRIORITY = 1
	// asks for the highest priority anyway (measurement).
	int lo = 0, hi = 0, prio = 0;
	const char* pe = FX_KNOB("COMM_PRIORITY");
	if (pe && pe[0] == '1' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) prio = hi;
	bool ok = hipStreamCreateWithPriority(&l.comm, hipStreamNonBlocking, prio) == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&l.face, hipStreamNonBlocking) == hipSuccess;
	for (hipEvent_t* e : { &l.ev_ready, &l.ev_done, &l.ev_col_ready, &l.ev_col_done, &l.ev_int, &l.ev_face1, &l.x_ready, &l.x_done })
		ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
	g->lanes.push_back(l);                             // (pushed even when incomplete: the group's teardown frees what exists)
	return ok ? FX_OK : FX_E_DEVICE;
}

// buffers of the per-step record (fx_context.h); nrec = records the host copy holds (RCCL: every rank's; loop-back member: its own)
static int make_step_record(fx_ctx* ctx, int nrec, bool gathered)
{
	DeviceGuard dg(ctx->device);
	if (ctx->step_rec) return FX_OK;
	FX_HIP(hipMalloc((void**)&ctx->step_rec, 4 * sizeof(int)));
	FX_HIP(hipMemset(ctx->step_rec, 0, 4 * sizeof(int)));
	if (gathered) FX_HIP(hipMalloc((void**)&ctx->gath_dev, 4 * sizeof(int) * (size_t)nrec));
	FX_HIP(hipHostMalloc((void**)&ctx->rec_host, 4 * sizeof(int) * (size_t)nrec, hipHostMallocDefault));
	std::memset(ctx->rec_host, 0, 4 * sizeof(int) * (size_t)nrec);
	FX_HIP(hipEventCreateWithFlags(&ctx->rec_ev, hipEventDisableTiming));
	return FX_OK;
}

int fx_comm_init_rank(fx_ctx* ctx, const void* id, size_t bytes, int rank, int nranks)
{
	if (!ctx || !id) return FX_E_INVALID;
	int rc = check_slab_chain(ctx, rank, nranks);
	if (rc) return rc;
	Transport* t = make_rccl_transport(id, bytes, rank, nranks, ctx->device, &ctx->last_error);
	if (!t) { std::fprintf(stderr, "fluidx: %s\n", ctx->last_error.c_str()); return FX_E_COMM; }
	fx_comm_group* g = new fx_comm_group();
	g->members.push_back(ctx);
	g->transport = t;
	g->refs = 1;
	g->per_member = false; g->shared_stream = nullptr; g->broken = false;
	if ((rc = make_lane(g, ctx->device, nullptr))) { destroy_lanes(g); delete t; delete g; return rc; }
	if ((rc = t->min_over_ranks(ctx->g.nz, ctx->stream, &g->min_nz))) { destroy_lanes(g); delete t; delete g; return rc; }
	{	// every rank must run the same schedule: same grid, halos, sweep count, Jacobi mode and storage (min == max of a digest)
		const fx_desc& d = ctx->desc;
		uint32_t h = 2166136261u;
		// ... and the same schedule: what selects the exchange sequence (FX_FLAG_NO_OVERLAP / FX_OPT_OVERLAP, FX_OPT_JACOBI_ROUND,
		// FX_OPT_ADAPTIVE_HALO) is part of the digest; later changes go through fx_set_option, which checks them across the chain
		for (uint32_t v : { d.grid_x, d.grid_y, d.grid_z, d.halo_advect, d.halo_jacobi, d.jacobi_iters, d.jacobi_mode, d.storage, d.advect_address, (uint32_t)nranks,
				d.flags & (FX_FLAG_NO_OVERLAP | FX_FLAG_JACOBI_FUSE_MASK), (uint32_t)ctx->opt_overlap, (uint32_t)ctx->opt_round, (uint32_t)ctx->opt_adaptive })
			h = (h ^ v) * 16777619u;
		const int digest = (int)(h & 0x3FFFFFFFu);
		int lo = 0, hi = 0;
		if ((rc = t->min_over_ranks(digest, ctx->stream, &lo)) || (rc = t->min_over_ranks(-digest, ctx->stream, &hi))) { destroy_lanes(g); delete t; delete g; return rc; }
		if (lo != digest || -hi != digest) {
			ctx->last_error = "fx_comm_init_rank: the ranks were created with different descriptors";
			std::fprintf(stderr, "fluidx: %s\n", ctx->last_error.c_str());
			destroy_lanes(g); delete t; delete g;
			return FX_E_INVALID;
		}
	}
	if (!ctx->step_rec && (rc = make_step_record(ctx, nranks, true))) { destroy_lanes(g); delete t; delete g; return rc; }   // (kept from a refused earlier attempt)
	{	// the slabs must tile the grid in rank order: rank 0 starts at plane 0, the last ends at Zg (check_slab_chain), and every
		// slab starts where its lower neighbour ends -- gaps or overlaps between middle slabs would exchange the wrong planes
		DeviceGuard dg(ctx->device);
		const int mine[4] = { ctx->g.z0, ctx->g.nz, 0, 0 };
		bool ok = hipMemcpy(ctx->step_rec, mine, sizeof mine, hipMemcpyHostToDevice) == hipSuccess &&
			t->allgather(ctx->step_rec, 4, ctx->gath_dev, ctx->stream) == FX_OK &&
			hipMemcpyAsync(ctx->rec_host, ctx->gath_dev, 4 * sizeof(int) * (size_t)nranks, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
			hipStreamSynchronize(ctx->stream) == hipSuccess;
		bool tiles = ok;
		for (int r = 0; ok && r + 1 < nranks; ++r) tiles = tiles && ctx->rec_host[4 * r] + ctx->rec_host[4 * r + 1] == ctx->rec_host[4 * (r + 1)];
		if (ok) ok = hipMemset(ctx->step_rec, 0, 4 * sizeof(int)) == hipSuccess;
		if (!ok || !tiles) {
			ctx->last_error = ok ? "fx_comm_init_rank: the slabs of the ranks do not tile the grid in rank order" : "fx_comm_init_rank: exchanging the slab ranges failed";
			std::fprintf(stderr, "fluidx: %s\n", ctx->last_error.c_str());
			destroy_lanes(g); delete t; delete g;
			return ok ? FX_E_INVALID : FX_E_COMM;
		}
	}
	ctx->group = g; ctx->rank = rank; ctx->nranks = nranks;
	return FX_OK;
}

int fx_comm_gather_color(fx_ctx* ctx, void* stream, fx_ctx* full, int root, const uint32_t* slab_z0, const uint32_t* slab_nz)
{
	if (!ctx || !ctx->group || root < 0 || root >= ctx->nranks) return FX_E_INVALID;
	if (group_broken(ctx)) return FX_E_STATE;
	if (!is_driver(ctx)) return FX_OK;
	const bool local = ctx->group->transport->is_local();
	const bool am_root = local || ctx->rank == root;
	if (am_root) {
		if (!full || full->g.X != ctx->g.X || full->g.Y != ctx->g.Y || full->g.Zg != ctx->g.Zg || full->g.nz != full->g.Zg ||
			full->half != ctx->half) return FX_E_INVALID;
		if (!local && full->device != ctx->device) return FX_E_INVALID;
		if (local && full->device != ctx->group->members[(size_t)root]->device) return FX_E_INVALID;   // the whole-grid context lives where the root does
	}
	if (!local && (!slab_z0 || !slab_nz)) return FX_E_INVALID;
	const size_t plane_bytes = ctx->g.plane() * 4 * elem_size(ctx);
	std::vector<GatherPart> parts((size_t)ctx->nranks);
	for (int r = 0; r < ctx->nranks; ++r) {
		const fx_ctx* m = local ? ctx->group->members[r] : (r == ctx->rank ? ctx : nullptr);
		const uint32_t z0 = local ? (uint32_t)m->g.z0 : slab_z0[r], nz = local ? (uint32_t)m->g.nz : slab_nz[r];
		if (z0 > (uint32_t)ctx->g.Zg || nz > (uint32_t)ctx->g.Zg - z0) return FX_E_INVALID;   // (z0 + nz would wrap for absurd input)
		if (!local && r == ctx->rank && (z0 != (uint32_t)ctx->g.z0 || nz != (uint32_t)ctx->g.nz)) return FX_E_INVALID;   // what this rank sends is its own slab
		parts[r].rank = r;
		parts[r].bytes = (size_t)nz * plane_bytes;
		parts[r].src = m ? (const char*)m->col[m->frame_parity] + (size_t)m->g.H * plane_bytes : nullptr;
		parts[r].dst = am_root ? (char*)full->col[full->frame_parity] + (size_t)z0 * plane_bytes : nullptr;
		if (am_root) full->accel_alpha_of = nullptr;
	}
	DeviceGuard dg(ctx->device);
	hipStream_t s = pick_stream(ctx, stream);
	if (ctx->halo_fault) return FX_E_HALO;             // a colour field that is known to be off is not gathered into a picture
	std::vector<hipStream_t> streams;                  // one; a peer group: every member's own, the copies ordered on the root's
	if (ctx->group->per_member) for (fx_ctx* m : ctx->group->members) streams.push_back(m->stream);
	else streams.push_back(s);
	hipStream_t done_on = ctx->group->per_member ? streams[(size_t)root] : s;
	ScopedMark mk(ctx, s, MK_EXCH);
	int rc = ctx->group->transport->gather(ctx->group, parts, root, streams);
	if (rc) return rc;
	if (am_root && full->stream != done_on) {          // the render context's own stream must see the planes
		fx_lane& L = ctx->group->per_member ? ctx->group->lanes[(size_t)root] : ctx->group->lanes[0];
		DeviceGuard dgr(L.device);
		FX_HIP(hipEventRecord(L.ev_done, done_on));
		DeviceGuard dgf(full->device);
		FX_HIP(hipStreamWaitEvent(full->stream, L.ev_done, 0));
	}
	return FX_OK;
}

// the checks both kinds of in-process group share; one_device: every member must live on the first one's device
static int check_local_members(fx_ctx** ctxs, int nranks, bool one_device)
{
	if (!ctxs || nranks < 1) return FX_E_INVALID;
	int zexp = 0;
	for (int r = 0; r < nranks; ++r) {
		if (!ctxs[r]) return FX_E_INVALID;
		int rc = check_slab_chain(ctxs[r], r, nranks);
		if (rc) return rc;
		if (ctxs[r]->g.z0 != zexp || (one_device && ctxs[r]->device != ctxs[0]->device)) return FX_E_INVALID;   // contiguous chain
		// every member must describe the same run (the RCCL path checks a digest of the same fields across the ranks): members
		// of different grids would exchange planes of different sizes
		const fx_desc &a = ctxs[r]->desc, &b = ctxs[0]->desc;
		if (a.grid_x != b.grid_x || a.grid_y != b.grid_y || a.grid_z != b.grid_z || a.halo_advect != b.halo_advect ||
			a.halo_jacobi != b.halo_jacobi || a.jacobi_iters != b.jacobi_iters || a.jacobi_mode != b.jacobi_mode ||
			a.storage != b.storage || a.advect_address != b.advect_address)
			return FX_E_INVALID;
		for (int q = 0; q < r; ++q) if (ctxs[q] == ctxs[r]) return FX_E_INVALID;
		zexp += ctxs[r]->g.nz;
	}
	return FX_OK;
}

static int init_in_process_group(fx_ctx** ctxs, int nranks, bool peer)
{
	int rc = check_local_members(ctxs, nranks, !peer);
	if (rc) return rc;
	if (peer) {
		// a receiver copies out of its neighbour's memory: across devices that needs peer access (one process owns them all: no IPC
		// handles, which this pool's driver does not give out between processes)
		for (int r = 0; r + 1 < nranks; ++r) {
			const int a = ctxs[r]->device, b = ctxs[r + 1]->device;
			if (a == b) continue;
			int ab = 0, ba = 0;
			if (hipDeviceCanAccessPeer(&ab, a, b) != hipSuccess || hipDeviceCanAccessPeer(&ba, b, a) != hipSuccess || !ab || !ba) {
				ctxs[0]->last_error = "fx_comm_init_peer: two neighbouring slabs live on devices without peer access";
				return FX_E_DEVICE;
			}
			for (int k = 0; k < 2; ++k) {
				DeviceGuard dg(k ? b : a);
				const hipError_t e = hipDeviceEnablePeerAccess(k ? a : b, 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return FX_E_DEVICE; }
				(void)hipGetLastError();
			}
		}
	}
	for (int r = 0; r < nranks; ++r)
		if (!ctxs[r]->step_rec) { if ((rc = make_step_record(ctxs[r], 1, false))) return rc; }
	fx_comm_group* g = new fx_comm_group();
	g->transport = make_local_transport();
	g->refs = nranks;
	g->per_member = peer; g->shared_stream = nullptr; g->broken = false;
	for (int r = 0; r < (peer ? nranks : 1); ++r)
		if ((rc = make_lane(g, ctxs[r]->device, peer ? ctxs[r]->stream : nullptr))) { destroy_lanes(g); delete g->transport; delete g; return rc; }
	g->min_nz = ctxs[0]->g.nz;
	for (int r = 1; r < nranks; ++r) g->min_nz = std::min(g->min_nz, ctxs[r]->g.nz);
	for (int r = 0; r < nranks; ++r) {
		g->members.push_back(ctxs[r]);
		ctxs[r]->group = g; ctxs[r]->rank = r; ctxs[r]->nranks = nranks;
		if (peer) continue;                                // every member keeps (and owns) its stream
		// one stream for the whole shared-stream group: phases of different members are ordered by it.  The group owns it, so
		// that the members can be destroyed in any order.
		if (r > 0) {
			if (ctxs[r]->owns_stream) (void)hipStreamDestroy(ctxs[r]->stream);
			ctxs[r]->stream = ctxs[0]->stream;
		}
		ctxs[r]->owns_stream = false;
		g->shared_stream = ctxs[0]->stream;
	}
	return FX_OK;
}

int fx_comm_init_local(fx_ctx** ctxs, int nranks) { return init_in_process_group(ctxs, nranks, false); }
int fx_comm_init_peer(fx_ctx** ctxs, int nranks) { return init_in_process_group(ctxs, nranks, true); }

}  // extern "C"

