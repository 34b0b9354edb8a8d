// fx_context.cpp -- contexts of the C ABI (include/fluidx_hip.h): what Fluid::Fluid / Fluid::Init (/root/reference/FluidX12/Content/
// Fluid.cpp:168-270) set up -- XUSG resources become hipMalloc'd fields owned by the context -- plus field read-back / upload,
// HIP-event timing and the option switches.  No CPU fallback: without a HIP device fx_create returns FX_E_DEVICE.
#include "fx_host.h"
#include <time.h>

using namespace fx;
using namespace fxh;

namespace fxh {

size_t ev_record(fx_ctx* c, hipStream_t s)
{
	if (c->ev_used == c->ev.size()) {
		hipEvent_t e;
		if (hipEventCreate(&e) != hipSuccess) return (size_t)-1;
		c->ev.push_back(e);
	}
	(void)hipEventRecord(c->ev[c->ev_used], s);
	return c->ev_used++;
}

int drain_timing(fx_ctx* c)
{
	for (const auto& m : c->marks) {
		float ms = 0.0f;
		if (hipEventSynchronize(c->ev[m.e1]) != hipSuccess) return FX_E_DEVICE;
		if (hipEventElapsedTime(&ms, c->ev[m.e0], c->ev[m.e1]) != hipSuccess) return FX_E_DEVICE;
		switch (m.kind) {
		case MK_ADVECT: c->acc.advect_ms += ms; break;
		case MK_DIV: c->acc.divergence_ms += ms; break;
		case MK_JACOBI: c->acc.jacobi_ms += ms; c->acc.jacobi_launches += m.launches; c->acc.jacobi_sweeps += m.sweeps;
			c->acc.jacobi_main_ms += ms; c->acc.jacobi_main_launches += m.launches; c->acc.jacobi_main_sweeps += m.sweeps; break;
		case MK_JACOBI_TAIL: c->acc.jacobi_ms += ms; c->acc.jacobi_launches += m.launches; c->acc.jacobi_sweeps += m.sweeps; break;
		case MK_PROJECT: c->acc.project_ms += ms; break;
		case MK_LIGHT: c->acc.light_ms += ms; break;
		case MK_VIEW: c->acc.view_ms += ms; break;
		case MK_EXCH: c->acc.exchange_ms += ms; break;
		case MK_RESOLVE: c->acc.resolve_ms += ms; break;
		case MK_CHAIN: c->acc.chain_ms += ms; break;
		}
	}
	c->marks.clear();
	c->ev_used = 0;
	return FX_OK;
}

// ---- staging ----------------------------------------------------------------------------------
int ensure_stage(fx_ctx* ctx, size_t bytes)
{
	if (ctx->stage_bytes >= bytes) return FX_OK;
	if (ctx->stage) { FX_HIP(hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
	FX_HIP(hipMalloc((void**)&ctx->stage, bytes));
	ctx->stage_bytes = bytes;
	return FX_OK;
}

void destroy_lanes(fx_comm_group* g)
{
	for (fx_lane& l : g->lanes) {
		DeviceGuard dg(l.device);
		if (l.comm) (void)hipStreamDestroy(l.comm);
		if (l.face) (void)hipStreamDestroy(l.face);
		for (hipEvent_t e : { l.ev_ready, l.ev_done, l.ev_col_ready, l.ev_col_done, l.ev_int, l.ev_face1, l.x_ready, l.x_done }) if (e) (void)hipEventDestroy(e);
	}
	g->lanes.clear();
}

void group_release(fx_comm_group* g)                // the last member is gone (fx_destroy)
{
	{
		DeviceGuard dg(g->lanes.empty() ? 0 : g->lanes[0].device);
		if (g->shared_stream) (void)hipStreamDestroy(g->shared_stream);
	}
	destroy_lanes(g);
	delete g->transport;
	delete g;
}

void free_all(fx_ctx* c)
{
	for (int i = 0; i < 2; ++i) {
		if (c->vel[i]) (void)hipFree(c->vel[i]);
		if (c->col[i]) (void)hipFree(c->col[i]);
		if (c->p[i]) (void)hipFree(c->p[i]);
	}
	void* others[] = { c->env, c->accel.occ, c->accel.alpha, c->accel.bits, c->accel.list, c->accel.cells, c->accel.gi, c->accel.ctr, c->target, c->target_float, c->p_face[0], c->p_face[1], c->b, c->frozen, c->frozen_alt, c->lightmap, c->cube, c->sh_dev, c->halo_overflow, c->stage,
		c->sh_scratch[0], c->sh_scratch[1], c->sh_scratch[2], c->sh_scratch[3], c->p_aux, c->fz_mask[0], c->fz_mask[1], c->fz_mask[2], c->fz_tile_next, c->fz_stat, c->fz_list[0], c->fz_list[1], c->fz_counts, c->sample_counters, c->adv_far };
	for (void* q : others) if (q) (void)hipFree(q);
	for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
	if (c->step_rec) (void)hipFree(c->step_rec);
	if (c->gath_dev) (void)hipFree(c->gath_dev);
	if (c->rec_host) (void)hipHostFree(c->rec_host);
	if (c->fz_active_host) (void)hipHostFree(c->fz_active_host);
	if (c->fz_active_ev) (void)hipEventDestroy(c->fz_active_ev);
	if (c->rec_ev) (void)hipEventDestroy(c->rec_ev);
	if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
}

}  // namespace fxh

extern "C" {

const char* fx_error_string(int status)
{
	switch (status) {
	case FX_OK: return "ok";
	case FX_E_INVALID: return "invalid argument";
	case FX_E_DEVICE: return "HIP device/runtime error";
	case FX_E_NOMEM: return "out of memory";
	case FX_E_STATE: return "invalid call order";
	case FX_E_COMM: return "RCCL communication error";
	case FX_E_HALO: return "advection back-trace left the exchanged halo";
	default: return "unknown status";
	}
}

int fx_create(fx_ctx** out, const fx_desc* d)
{
	if (!out || !d || d->struct_size != sizeof(fx_desc)) return FX_E_INVALID;
	*out = nullptr;
	if (!d->grid_x || !d->grid_y || !d->grid_z) return FX_E_INVALID;
	if (d->grid_x != d->grid_y) return FX_E_INVALID;                         // assert at Fluid.cpp:201
	if (d->grid_x > 65535 || d->grid_z > 65535) return FX_E_INVALID;         // Texture3D extents are uint16 (XUSG.h:1805)
	if (d->storage > FX_STORAGE_FP16 || d->jacobi_mode > FX_JACOBI_FAITHFUL || d->advect_address > FX_ADDRESS_MIRROR) return FX_E_INVALID;
	if (!d->jacobi_iters) return FX_E_INVALID;
	uint32_t z0 = d->slab_z0, nz = d->slab_nz ? d->slab_nz : d->grid_z;
	if (z0 + nz > d->grid_z) return FX_E_INVALID;
	const bool slab = nz != d->grid_z;

	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FX_E_DEVICE;     // fail loudly: no CPU path
	int dev = d->device;
	if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return FX_E_DEVICE;
	if (dev >= ndev) return FX_E_INVALID;

	fx_ctx* ctx = new (std::nothrow) fx_ctx();
	if (!ctx) return FX_E_NOMEM;
	ctx->desc = *d;
	ctx->desc.slab_z0 = z0; ctx->desc.slab_nz = nz;
	if (!ctx->desc.halo_advect) ctx->desc.halo_advect = kDefaultAdvectHalo;
	if (!ctx->desc.halo_jacobi) ctx->desc.halo_jacobi = kDefaultJacobiHalo;
	if (!slab) { ctx->desc.halo_advect = 0; }
	const int H = slab ? (int)std::max(ctx->desc.halo_advect, ctx->desc.halo_jacobi) : 0;
	if (slab && (int)nz < H) { delete ctx; return FX_E_INVALID; }             // a halo may only span the direct neighbour
	if (slab && (d->flags & FX_FLAG_RENDER_ONLY)) { delete ctx; return FX_E_INVALID; }   // rays cross slabs: render contexts are whole grids
	ctx->g = Geom{ (int)d->grid_x, (int)d->grid_y, (int)d->grid_z, (int)z0, (int)nz, H,
		std::max((int)z0 - H, 0), std::min((int)(z0 + nz) + H, (int)d->grid_z) - 1 };
	ctx->half = d->storage == FX_STORAGE_FP16;
	ctx->device = dev;
	ctx->max_ray_samples = 192; ctx->max_light_samples = 64;                  // Fluid.cpp:174-175
	ctx->rank = 0; ctx->nranks = 1;
	ctx->opt_overlap = (d->flags & FX_FLAG_NO_OVERLAP) ? 0 : 2;
	ctx->opt_round = (int)ctx->desc.halo_jacobi;

	DeviceGuard dg(dev);
	if (!dg.ok) { delete ctx; return FX_E_DEVICE; }
	int rc = [&]() -> int {
		FX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
		ctx->owns_stream = true;
		const size_t cells = ctx->g.cells_local();
		const size_t es = elem_size(ctx);
		if (d->flags & FX_FLAG_RENDER_ONLY) {                                // colour only; parity never flips
			FX_HIP(hipMalloc(&ctx->col[0], 4 * cells * es));
			FX_HIP(hipMemsetAsync(ctx->col[0], 0, 4 * cells * es, ctx->stream));
		}
		for (int i = 0; i < 2 && !(d->flags & FX_FLAG_RENDER_ONLY); ++i) {
			FX_HIP(hipMalloc(&ctx->vel[i], 3 * cells * es));
			FX_HIP(hipMalloc(&ctx->col[i], 4 * cells * es));
			FX_HIP(hipMalloc((void**)&ctx->p[i], cells * 4));
			FX_HIP(hipMemsetAsync(ctx->vel[i], 0, 3 * cells * es, ctx->stream));
			FX_HIP(hipMemsetAsync(ctx->col[i], 0, 4 * cells * es, ctx->stream));
			FX_HIP(hipMemsetAsync(ctx->p[i], 0, cells * 4, ctx->stream));
		}
		if (slab && !(d->flags & FX_FLAG_RENDER_ONLY))                       // scratch levels of the face chains (jacobi_overlapped)
			for (int i = 0; i < 2; ++i) {
				FX_HIP(hipMalloc((void**)&ctx->p_face[i], cells * 4));
				FX_HIP(hipMemsetAsync(ctx->p_face[i], 0, cells * 4, ctx->stream));
			}
		if (!(d->flags & FX_FLAG_RENDER_ONLY)) {
			FX_HIP(hipMalloc((void**)&ctx->b, cells * 4));
			FX_HIP(hipMemsetAsync(ctx->b, 0, cells * 4, ctx->stream));
		}
		if (d->jacobi_mode == FX_JACOBI_FAITHFUL && !(d->flags & FX_FLAG_RENDER_ONLY)) {
			FX_HIP(hipMalloc((void**)&ctx->frozen, cells));
			FX_HIP(hipMemsetAsync(ctx->frozen, 0, cells, ctx->stream));
			if (ctx->g.Zg == 1) {                                            // 2-D: the LDS-tile kernel double-buffers the mask
				FX_HIP(hipMalloc((void**)&ctx->frozen_alt, cells));
				FX_HIP(hipMemsetAsync(ctx->frozen_alt, 0, cells, ctx->stream));
			}
			if (jacobi_freeze_supported(ctx->g)) {                           // the sparse solver of fx_jacobi_freeze.hip
				const size_t mb = jacobi_freeze_mask_bytes(ctx->g), nt = (size_t)jacobi_freeze_tiles(ctx->g);
				FX_HIP(hipMalloc((void**)&ctx->p_aux, cells * 4));
				FX_HIP(hipMemsetAsync(ctx->p_aux, 0, cells * 4, ctx->stream));
				for (int i = 0; i < 3; ++i) {
					FX_HIP(hipMalloc((void**)&ctx->fz_mask[i], mb));
					FX_HIP(hipMemsetAsync(ctx->fz_mask[i], 0, mb, ctx->stream));
				}
				FX_HIP(hipMalloc((void**)&ctx->fz_tile_next, nt * sizeof(uint32_t)));
				FX_HIP(hipMemsetAsync(ctx->fz_tile_next, 0, nt * sizeof(uint32_t), ctx->stream));
				for (int i = 0; i < 2; ++i) FX_HIP(hipMalloc(&ctx->fz_list[i], jacobi_freeze_list_bytes(ctx->g)));
				FX_HIP(hipMalloc((void**)&ctx->fz_counts, 2 * jacobi_freeze_count_words() * sizeof(uint32_t)));
				FX_HIP(hipMemsetAsync(ctx->fz_counts, 0, 2 * jacobi_freeze_count_words() * sizeof(uint32_t), ctx->stream));
				FX_HIP(hipMalloc((void**)&ctx->fz_stat, kFreezeStatRing * sizeof(uint32_t)));
				FX_HIP(hipMemsetAsync(ctx->fz_stat, 0, kFreezeStatRing * sizeof(uint32_t), ctx->stream));
				ctx->fz_iters.assign(kFreezeStatRing, 0);
				if (jacobi_freeze_strip_supported(ctx->g) && hipHostMalloc((void**)&ctx->fz_active_host, 64, hipHostMallocMapped) == hipSuccess) {
					ctx->fz_active_host[0] = 0u;
					if (hipHostGetDevicePointer((void**)&ctx->fz_active_dev, ctx->fz_active_host, 0) != hipSuccess) ctx->fz_active_dev = nullptr;
				} else (void)hipGetLastError();
			}
		}
		FX_HIP(hipMalloc((void**)&ctx->halo_overflow, sizeof(unsigned)));
		FX_HIP(hipMemsetAsync(ctx->halo_overflow, 0, sizeof(unsigned), ctx->stream));
		if (!(d->flags & FX_FLAG_RENDER_ONLY)) {
			// scratch the staged advection puts far-tracing voxels aside in (fx_advect_lds.hip), for the owned planes: allocated here, not
			// inside the first step (an allocation there is a device synchronisation on the step path).  Without it -- no staged path for
			// this geometry, or no memory -- the staged kernel gathers those voxels itself.
			ctx->adv_far_tried = true;
			const size_t words = advect_far_words(ctx->g, ctx->g.nz);
			if (words) {
				if (hipMalloc((void**)&ctx->adv_far, words * sizeof(uint32_t)) == hipSuccess) {
					FX_HIP(hipMemsetAsync(ctx->adv_far, 0, 2 * sizeof(uint32_t), ctx->stream));   // the two alternating totals
					ctx->adv_far_words = words;
				} else { (void)hipGetLastError(); ctx->adv_far = nullptr; }
			}
		}
		if (d->grid_z > 1) {                                                 // rendering resources (Fluid.cpp:222-232)
			FX_HIP(hipMalloc((void**)&ctx->lightmap, ctx->g.cells_owned() * 4));
			FX_HIP(hipMemsetAsync(ctx->lightmap, 0, ctx->g.cells_owned() * 4, ctx->stream));
			size_t off = 0;
			for (uint32_t m = 0; m < kNumMips; ++m) {
				ctx->cube_mip_offset[m] = off;
				const size_t sz = std::max<uint32_t>(d->grid_x >> m, 1);
				off += 6 * sz * sz * 4;
			}
			FX_HIP(hipMalloc((void**)&ctx->cube, off));
			FX_HIP(hipMemsetAsync(ctx->cube, 0, off, ctx->stream));
			// rays cross slabs: only whole grids render.  The accelerated marches address their volumes by 32-bit byte offsets (16-byte texels:
			// 2^28 voxels, 645^3); larger grids keep the plain kernels
			if (!slab && ctx->g.cells_owned() <= ((size_t)1 << 28) && ctx->g.X >= 4) {        // (rows of >= 2 voxels: the x taps travel in pairs)
				RenderAccel& A = ctx->accel;
				render_accel_layout(ctx->g, &A);
				const size_t ncell = (size_t)A.CX * A.CY * A.CZ, vox = ctx->g.cells_owned(), bw = render_accel_bits_words(A);
				FX_HIP(hipMalloc((void**)&A.occ, 2 * ncell * sizeof(float)));     // the grid + the per-block maxima it is dilated from
				FX_HIP(hipMalloc((void**)&A.alpha, vox * sizeof(float)));
				FX_HIP(hipMalloc((void**)&A.bits, bw * sizeof(uint32_t)));
				FX_HIP(hipMemsetAsync(A.bits, 0, bw * sizeof(uint32_t), ctx->stream));
				FX_HIP(hipMalloc((void**)&A.list, vox * sizeof(uint32_t)));
				FX_HIP(hipMalloc((void**)&A.cells, ncell * sizeof(uint32_t)));
				FX_HIP(hipMalloc((void**)&A.ctr, render_accel_ctr_words(ctx->g) * sizeof(uint32_t)));
				FX_HIP(hipMemsetAsync(A.ctr, 0, render_accel_ctr_words(ctx->g) * sizeof(uint32_t), ctx->stream));
				ctx->accel_ok = true;
			}
			FX_HIP(hipMalloc((void**)&ctx->sh_dev, 27 * sizeof(float)));
			FX_HIP(hipMemsetAsync(ctx->sh_dev, 0, 27 * sizeof(float), ctx->stream));
		}
		FX_HIP(hipStreamSynchronize(ctx->stream));
		return FX_OK;
	}();
	if (rc != FX_OK) { free_all(ctx); delete ctx; return rc; }
	*out = ctx;
	return FX_OK;
}

int fx_destroy(fx_ctx* ctx)
{
	if (!ctx) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	(void)hipDeviceSynchronize();
	if (ctx->group) {
		fx_comm_group* g = ctx->group;
		for (auto& m : g->members) if (m == ctx) m = nullptr;
		g->broken = true;
		if (--g->refs == 0) group_release(g);
	}
	free_all(ctx);
	delete ctx;
	return FX_OK;
}

// after a device synchronisation: has an advection of this context left its exchanged planes (and nobody acknowledged it yet)?
// The fields are then not the single-domain run's any more: whatever reads them back or stores them says so.
static int halo_fault_status(fx_ctx* c)
{
	if (!c->halo_overflow) return FX_OK;
	unsigned flag = 0;
	if (hipMemcpy(&flag, c->halo_overflow, sizeof flag, hipMemcpyDeviceToHost) != hipSuccess) return FX_E_DEVICE;
	if (flag) c->halo_fault = true;
	return c->halo_fault ? FX_E_HALO : FX_OK;
}

// An RCCL rank waits for its device in polls: a neighbour that died leaves this rank's receive kernels spinning for ever, and
// hipDeviceSynchronize behind them would never return.  Between polls the communicators' asynchronous error is read
// (Transport::poll_error); on a failure they are aborted -- RCCL's kernels end -- and the call returns FX_E_COMM.
static int wait_for_device(fx_ctx* c)
{
	Transport* t = c->group && c->group->transport && !c->group->transport->is_local() ? c->group->transport : nullptr;
	if (!t || !t->can_poll()) return hipDeviceSynchronize() == hipSuccess ? FX_OK : FX_E_DEVICE;
	std::vector<hipStream_t> streams{ c->stream, c->last_step_stream };
	for (const fx_lane& L : c->group->lanes) { streams.push_back(L.comm); streams.push_back(L.face); }
	std::vector<hipEvent_t> evs;
	int rc = FX_OK;
	for (hipStream_t s : streams) {
		if (!s) continue;
		hipEvent_t e;
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rc = FX_E_DEVICE; break; }
		evs.push_back(e);                                                   // (owned from here on: destroyed below whatever happens)
		if (hipEventRecord(e, s) != hipSuccess) { rc = FX_E_DEVICE; break; }
	}
	for (size_t i = 0; i < evs.size() && rc == FX_OK;) {
		const hipError_t q = hipEventQuery(evs[i]);
		if (q == hipSuccess) { ++i; continue; }
		if (q != hipErrorNotReady) { rc = FX_E_DEVICE; break; }
		if ((rc = t->poll_error(&c->last_error))) break;
		struct timespec ts = { 0, 200000 };                                 // 0.2 ms between polls
		nanosleep(&ts, nullptr);
	}
	for (hipEvent_t e : evs) (void)hipEventDestroy(e);
	if (rc == FX_OK && (rc = t->poll_error(&c->last_error)) == FX_OK && hipDeviceSynchronize() != hipSuccess) rc = FX_E_DEVICE;
	return rc;
}

const char* fx_last_error(fx_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int fx_synchronize(fx_ctx* ctx)
{
	if (!ctx) return FX_E_INVALID;
	std::vector<fx_ctx*> M;
	for_members(ctx, M);
	int rc = FX_OK;
	for (fx_ctx* c : M) {
		DeviceGuard dg(c->device);
		if (const int w = wait_for_device(c)) return w;
		// a strip kernel whose LDS hand-over timed out has left a pressure field that is not the solver's: loud, never silent
		unsigned f3 = 0, f4 = 0;
		if (strip3_fault_take(&f3) != hipSuccess || strip4_fault_take(&f4) != hipSuccess) return FX_E_DEVICE;
		if (f3 | f4) {
			c->last_error = f4 ? "k_jacobi_strip4o / k_freeze_strip4o: an LDS hand-over wait ran out (pressure field invalid)"
			                   : "k_jacobi_strip3c / k_jacobi_strip3h: an LDS hand-over wait ran out (pressure field invalid)";
			return FX_E_DEVICE;
		}
		const int st = halo_fault_status(c);
		if (st == FX_E_DEVICE) return st;
		if (st == FX_E_HALO) {                 // reported here, and acknowledged: the next step starts clean
			(void)hipMemset(c->halo_overflow, 0, sizeof(unsigned));
			(void)hipDeviceSynchronize();      // the context's streams do not order against the NULL stream
			c->halo_fault = false;
			rc = FX_E_HALO;
		}
	}
	return rc;
}

// ---- field access ----------------------------------------------------------------------------------
static int field_info(fx_ctx* c, int field, size_t* host_bytes)
{
	const size_t n = c->g.cells_owned();
	switch (field) {
	case FX_FIELD_VELOCITY: case FX_FIELD_VELOCITY1: if (!c->vel[0]) return FX_E_STATE; *host_bytes = 3 * n * 4; return FX_OK;
	case FX_FIELD_COLOR_PREV: if (!c->col[1]) return FX_E_STATE;   /* fall through */
	case FX_FIELD_COLOR: *host_bytes = 4 * n * 4; return FX_OK;
	case FX_FIELD_PRESSURE: case FX_FIELD_DIVERGENCE: if (!c->b) return FX_E_STATE; *host_bytes = n * 4; return FX_OK;
	case FX_FIELD_LIGHTMAP: if (!c->lightmap) return FX_E_INVALID; *host_bytes = 3 * n * 4; return FX_OK;
	case FX_FIELD_CUBEMAP: {
		if (!c->cube) return FX_E_INVALID;
		const size_t s = (size_t)c->g.X >> c->cube_lod;
		*host_bytes = 6 * s * s * 4;
		return FX_OK;
	}
	case FX_FIELD_TARGET: case FX_FIELD_TARGET_FLOAT:
		if (!c->target) return FX_E_STATE;
		*host_bytes = (size_t)c->desc.viewport_w * c->desc.viewport_h * (field == FX_FIELD_TARGET ? 4 : 16);
		return FX_OK;
	}
	return FX_E_INVALID;
}

size_t fx_field_bytes(fx_ctx* ctx, int field)
{
	size_t b = 0;
	if (!ctx || field_info(ctx, field, &b) != FX_OK) return 0;
	return b;
}

int fx_field_digest(fx_ctx* ctx, int field, uint32_t z_begin, uint32_t z_count, uint64_t out[2])
{
	if (!ctx || !out) return FX_E_INVALID;
	size_t need = 0;
	int rc = field_info(ctx, field, &need);
	if (rc) return rc;
	if (field > FX_FIELD_DIVERGENCE) return FX_E_INVALID;
	const Geom& g = ctx->g;
	if (!z_count) { z_begin = (uint32_t)g.z0; z_count = (uint32_t)g.nz; }
	if ((int)z_begin < g.z0 || z_begin + z_count > (uint32_t)(g.z0 + g.nz)) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	FX_HIP(hipDeviceSynchronize());
	if ((rc = halo_fault_status(ctx))) return rc;
	if ((rc = ensure_stage(ctx, 16))) return rc;
	unsigned long long* acc = reinterpret_cast<unsigned long long*>(ctx->stage);
	FX_HIP(hipMemsetAsync(acc, 0, 16, ctx->stream));
	const size_t plane = g.plane(), cl = g.cells_local(), first = (size_t)g.lz((int)z_begin) * plane, n = (size_t)z_count * plane;
	const size_t es = elem_size(ctx);
	const unsigned long long gfirst = (unsigned long long)z_begin * plane, gcells = (unsigned long long)g.Zg * plane;
	switch (field) {
	case FX_FIELD_VELOCITY: case FX_FIELD_VELOCITY1: {
		const char* src = (const char*)ctx->vel[field == FX_FIELD_VELOCITY1];
		for (int a = 0; a < 3; ++a) FX_HIP(launch_digest(src + (a * cl + first) * es, n, (int)es, a * gcells + gfirst, acc, ctx->stream));
		break;
	}
	case FX_FIELD_COLOR: case FX_FIELD_COLOR_PREV: {
		const char* src = (const char*)ctx->col[field == FX_FIELD_COLOR ? ctx->frame_parity : 1 - ctx->frame_parity];
		FX_HIP(launch_digest(src + first * 4 * es, 4 * n, (int)es, 4 * gfirst, acc, ctx->stream));
		break;
	}
	case FX_FIELD_PRESSURE: FX_HIP(launch_digest(ctx->p[ctx->p_cur] + first, n, 4, gfirst, acc, ctx->stream)); break;
	default: FX_HIP(launch_digest(ctx->b + first, n, 4, gfirst, acc, ctx->stream)); break;
	}
	FX_HIP(hipStreamSynchronize(ctx->stream));
	FX_HIP(hipMemcpy(out, acc, 16, hipMemcpyDeviceToHost));
	return FX_OK;
}

int fx_upload(fx_ctx* ctx, int field, const void* host, size_t bytes)
{
	if (!ctx || !host) return FX_E_INVALID;
	size_t need = 0;
	int rc = field_info(ctx, field, &need);
	if (rc) return rc;
	if (bytes != need) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	FX_HIP(hipDeviceSynchronize());
	// a state that comes from outside (a test's fields, a checkpoint): what the sparse solver had learnt about the OLD state's relaxing
	// tiles -- how many masked strip launches a solve takes, a count still on its way -- starts over, so that the launch sequence is a
	// function of (the uploaded state, the steps since) on every run
	if (field <= FX_FIELD_DIVERGENCE) { ctx->fz_dense_n = 0; ctx->fz_active_pending = false; }
	const size_t n = ctx->g.cells_owned(), off = (size_t)ctx->g.H * ctx->g.plane(), cl = ctx->g.cells_local();
	const size_t es = elem_size(ctx);
	switch (field) {
	case FX_FIELD_VELOCITY: case FX_FIELD_VELOCITY1: {
		if (field == FX_FIELD_VELOCITY && ctx->group) {
			// the next advection exchange is sized from a measurement of THIS buffer (FX_OPT_ADAPTIVE_HALO).  A loop-back group
			// simply exchanges the whole halo once; the neighbours of an RCCL rank could not know, so the upload is refused
			// while a measurement is out (switch the option off on every rank first, or upload before the first step)
			// (fx_checkpoint_load is made by every rank: each drops its measurement, and all fall back to halo_advect planes together)
			if (!ctx->group->transport->is_local() && ctx->opt_adaptive && ctx->rec_pending && !ctx->collective_upload) return FX_E_STATE;
			ctx->need_valid = false;
		}
		char* dst = (char*)ctx->vel[field == FX_FIELD_VELOCITY1];
		if ((rc = ensure_stage(ctx, need))) return rc;
		FX_HIP(hipMemcpy(ctx->stage, host, need, hipMemcpyHostToDevice));
		for (int a = 0; a < 3; ++a)
			FX_HIP(launch_to_storage(ctx->stage + a * n, dst + (a * cl + off) * es, n, ctx->half, ctx->stream));
		break;
	}
	case FX_FIELD_COLOR: case FX_FIELD_COLOR_PREV: {
		// FX_OPT_OVERLAP 3: the neighbours already hold this context's colour border planes for the next step.  In a loop-back
		// group the flag can simply be dropped for everyone (the next step exchanges the colour again); across processes the
		// neighbours cannot know, so the upload is refused (set FX_OPT_OVERLAP <= 2 before the step that precedes it).
		if (ctx->col_halo_buf >= 0 && ctx->group) {
			if (!ctx->group->transport->is_local() && !ctx->collective_upload) return FX_E_STATE;
			if (ctx->group->transport->is_local()) { for (fx_ctx* m : ctx->group->members) if (m) m->col_halo_buf = -1; }
			else ctx->col_halo_buf = -1;                   // collective load: every rank forgets the early halo, the next step exchanges the colour again
		}
		char* dst = (char*)ctx->col[field == FX_FIELD_COLOR ? ctx->frame_parity : 1 - ctx->frame_parity];
		if (ctx->accel_alpha_of == dst) ctx->accel_alpha_of = nullptr;      // the render's side volume no longer mirrors this buffer
		if ((rc = ensure_stage(ctx, need))) return rc;
		FX_HIP(hipMemcpy(ctx->stage, host, need, hipMemcpyHostToDevice));
		FX_HIP(launch_to_storage(ctx->stage, dst + off * 4 * es, 4 * n, ctx->half, ctx->stream));
		break;
	}
	case FX_FIELD_PRESSURE:
		FX_HIP(hipMemcpy(ctx->p[ctx->p_cur] + off, host, need, hipMemcpyHostToDevice));
		break;
	case FX_FIELD_DIVERGENCE:
		FX_HIP(hipMemcpy(ctx->b + off, host, need, hipMemcpyHostToDevice));
		break;
	case FX_FIELD_CUBEMAP:       // mip `cube_lod`: lets the resolve be driven with a known cube map (parity tests, replays)
		FX_HIP(hipMemcpy(ctx->cube + ctx->cube_mip_offset[ctx->cube_lod], host, need, hipMemcpyHostToDevice));
		break;
	default:
		return FX_E_INVALID;     // light map / render target are outputs
	}
	FX_HIP(hipStreamSynchronize(ctx->stream));
	return FX_OK;
}

int fx_download(fx_ctx* ctx, int field, void* host, size_t bytes)
{
	if (!ctx || !host) return FX_E_INVALID;
	size_t need = 0;
	int rc = field_info(ctx, field, &need);
	if (rc) return rc;
	if (bytes != need) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	FX_HIP(hipDeviceSynchronize());
	if (field <= FX_FIELD_DIVERGENCE && (rc = halo_fault_status(ctx))) return rc;     // simulation fields of a faulted slab run are not handed out as if nothing had happened
	const size_t n = ctx->g.cells_owned(), off = (size_t)ctx->g.H * ctx->g.plane(), cl = ctx->g.cells_local();
	const size_t es = elem_size(ctx);
	switch (field) {
	case FX_FIELD_VELOCITY: case FX_FIELD_VELOCITY1: {
		const char* src = (const char*)ctx->vel[field == FX_FIELD_VELOCITY1];
		if ((rc = ensure_stage(ctx, need))) return rc;
		for (int a = 0; a < 3; ++a)
			FX_HIP(launch_from_storage(src + (a * cl + off) * es, ctx->stage + a * n, n, ctx->half, ctx->stream));
		FX_HIP(hipStreamSynchronize(ctx->stream));
		FX_HIP(hipMemcpy(host, ctx->stage, need, hipMemcpyDeviceToHost));
		break;
	}
	case FX_FIELD_COLOR: case FX_FIELD_COLOR_PREV: {
		const char* src = (const char*)ctx->col[field == FX_FIELD_COLOR ? ctx->frame_parity : 1 - ctx->frame_parity];
		if ((rc = ensure_stage(ctx, need))) return rc;
		FX_HIP(launch_from_storage(src + off * 4 * es, ctx->stage, 4 * n, ctx->half, ctx->stream));
		FX_HIP(hipStreamSynchronize(ctx->stream));
		FX_HIP(hipMemcpy(host, ctx->stage, need, hipMemcpyDeviceToHost));
		break;
	}
	case FX_FIELD_PRESSURE:
		FX_HIP(hipMemcpy(host, ctx->p[ctx->p_cur] + off, need, hipMemcpyDeviceToHost));
		break;
	case FX_FIELD_DIVERGENCE:
		FX_HIP(hipMemcpy(host, ctx->b + off, need, hipMemcpyDeviceToHost));
		break;
	case FX_FIELD_LIGHTMAP:
		if ((rc = ensure_stage(ctx, need))) return rc;
		FX_HIP(launch_lightmap_decode(ctx->lightmap, ctx->stage, n, ctx->stream));
		FX_HIP(hipStreamSynchronize(ctx->stream));
		FX_HIP(hipMemcpy(host, ctx->stage, need, hipMemcpyDeviceToHost));
		break;
	case FX_FIELD_CUBEMAP:
		FX_HIP(hipMemcpy(host, ctx->cube + ctx->cube_mip_offset[ctx->cube_lod], need, hipMemcpyDeviceToHost));
		break;
	case FX_FIELD_TARGET:
		FX_HIP(hipMemcpy(host, ctx->target, need, hipMemcpyDeviceToHost));
		break;
	case FX_FIELD_TARGET_FLOAT:
		FX_HIP(hipMemcpy(host, ctx->target_float, need, hipMemcpyDeviceToHost));
		break;
	default:
		return FX_E_INVALID;
	}
	return FX_OK;
}

int fx_timing_enable(fx_ctx* ctx, int enable)
{
	if (!ctx) return FX_E_INVALID;
	std::vector<fx_ctx*> M;
	for_members(ctx, M);
	for (fx_ctx* c : M) {
		c->timing_on = enable != 0;
		if (enable) {                                  // the events of a few hundred steps exist before the timed region starts
			DeviceGuard dg(c->device);
			while (c->ev.size() < 4096) {
				hipEvent_t e;
				if (hipEventCreate(&e) != hipSuccess) return FX_E_DEVICE;
				c->ev.push_back(e);
			}
		}
	}
	return FX_OK;
}

int fx_timing_read(fx_ctx* ctx, fx_timing* out, int reset)
{
	if (!ctx || !out) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	int rc = drain_timing(ctx);
	if (rc) return rc;
	if (ctx->sample_counters) {                         // FX_OPT_COUNT_SAMPLES: fold the device shards into the accumulators
		unsigned long long h[kSampleShards * 3];
		FX_HIP(hipDeviceSynchronize());
		FX_HIP(hipMemcpy(h, ctx->sample_counters, sizeof h, hipMemcpyDeviceToHost));
		FX_HIP(hipMemset(ctx->sample_counters, 0, sizeof h));
		for (int i = 0; i < kSampleShards; ++i) { ctx->acc.view_samples += h[3 * i]; ctx->acc.light_samples += h[3 * i + 1]; ctx->acc.lightmap_fetches += h[3 * i + 2]; }
	}
	*out = ctx->acc;
	// faithful mode, sparse solver: sweeps the reference's loop would have executed, per solve since the last reset (the device
	// keeps the last level that left a cell relaxing, one word per solve; older solves than the ring holds are not counted)
	if (ctx->fz_stat && ctx->fz_gen > ctx->fz_gen_mark) {
		std::vector<uint32_t> ring(kFreezeStatRing);
		FX_HIP(hipDeviceSynchronize());
		FX_HIP(hipMemcpy(ring.data(), ctx->fz_stat, kFreezeStatRing * sizeof(uint32_t), hipMemcpyDeviceToHost));
		const uint32_t first = std::max(ctx->fz_gen_mark + 1, ctx->fz_gen >= kFreezeStatRing ? ctx->fz_gen - kFreezeStatRing + 1 : 1u);
		for (uint32_t gtag = first; gtag <= ctx->fz_gen; ++gtag) {
			const uint32_t w = ring[gtag % kFreezeStatRing], lvl = (w >> 8) == gtag ? (w & 0xFFu) : 0u;
			out->freeze_sweeps += std::min(ctx->fz_iters[gtag % kFreezeStatRing], 1u + lvl);
			out->freeze_solves += 1;
		}
	}
	if (reset) { std::memset(&ctx->acc, 0, sizeof ctx->acc); ctx->fz_gen_mark = ctx->fz_gen; }
	return FX_OK;
}

int fx_set_option(fx_ctx* ctx, uint32_t option, uint32_t value)
{
	if (!ctx) return FX_E_INVALID;
	int* slot = nullptr;
	switch (option) {
	case FX_OPT_OVERLAP: if (value > 3) return FX_E_INVALID; slot = &ctx->opt_overlap; break;
	case FX_OPT_JACOBI_ROUND: if (value < 1 || value > ctx->desc.halo_jacobi) return FX_E_INVALID; slot = &ctx->opt_round; break;
	case FX_OPT_ADAPTIVE_HALO: if (value > 1) return FX_E_INVALID; slot = &ctx->opt_adaptive; break;
	case FX_OPT_RENDER_ACCEL:                          // local to the context: which kernels its renders run
		if (value > 1) return FX_E_INVALID;
		ctx->opt_render_accel = (int)value;
		return FX_OK;
	case FX_OPT_COUNT_SAMPLES: {                       // local to the context: statistics of its own renders
		if (value > 1) return FX_E_INVALID;
		DeviceGuard dgc(ctx->device);
		if (value && !ctx->sample_counters) {
			FX_HIP(hipMalloc((void**)&ctx->sample_counters, kSampleShards * 3 * sizeof(unsigned long long)));
			FX_HIP(hipMemset(ctx->sample_counters, 0, kSampleShards * 3 * sizeof(unsigned long long)));
		}
		ctx->opt_count_samples = (int)value;
		return FX_OK;
	}
	default: return FX_E_INVALID;
	}
	// These options select the exchange sequence and the exchanged byte counts: ranks that disagree would hang RCCL or corrupt
	// halos.  On an RCCL chain the call is therefore collective -- every rank makes it, with the same arguments, between two
	// steps -- and the values are compared across the chain (min == max) before any of them takes effect.
	if (ctx->group && !ctx->group->transport->is_local() && ctx->nranks > 1) {
		DeviceGuard dg(ctx->device);
		const int key = (int)(((option & 0xFu) << 8) | (value & 0xFFu));
		int lo = 0, hi = 0, rc;
		if ((rc = ctx->group->transport->min_over_ranks(key, ctx->stream, &lo)) || (rc = ctx->group->transport->min_over_ranks(-key, ctx->stream, &hi))) return rc;
		if (lo != key || -hi != key) {
			ctx->last_error = "fx_set_option: the ranks of the chain asked for different options";
			return FX_E_INVALID;
		}
	}
	*slot = (int)value;
	if (option == FX_OPT_ADAPTIVE_HALO) {
		// Whether the measured need may size the next exchange is decided per rank from `need_valid`; both sides of a face must
		// decide alike.  The one way to make the ranks differ was: option off, velocity upload into ONE rank, option on.  Setting the
		// option -- a call every rank makes -- therefore drops the measurement everywhere: the next step exchanges halo_advect planes
		// on all ranks and measures afresh.
		std::vector<fx_ctx*> M;
		for_members(ctx, M);
		for (fx_ctx* m : M) { m->need_valid = false; if (m != ctx) m->opt_adaptive = (int)value; }
	}
	return FX_OK;
}

}  // extern "C"
