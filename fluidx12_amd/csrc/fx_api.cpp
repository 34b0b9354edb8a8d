// fx_api.cpp -- implementation of the C ABI declared in include/fluidx_hip.h: the HIP re-statement
// of class Fluid's host side (/root/reference/FluidX12/Content/Fluid.cpp).  XUSG resources become
// hipMalloc'd fields, the command list becomes a HIP stream, the 3-slot upload constant buffers
// (Fluid.cpp:239-252) become kernel arguments passed by value (no host/device hazard, so
// `frame_index` only needs range-checking).
//
// No CPU fallback and nothing from oracle/: every field operation is a HIP kernel.
#include "fx_context.h"
#include "fx_hostmath.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

using namespace fx;

// A failed HIP call also leaves its code in the runtime's sticky "last error": the launch helpers end in hipGetLastError(), and a
// stale out-of-memory from one context's failed fx_create would otherwise surface as the status of the next, unrelated launch
// (found by the descriptor fuzz).  Reading the last error here clears it.
#define FX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); \
	ctx->last_error = std::string(#call) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? FX_E_NOMEM : FX_E_DEVICE; } } while (0)

namespace {

const uint32_t kNumMips = 5;            // Fluid.cpp:229
const uint32_t kDefaultAdvectHalo = 6;  // measured z back-trace reach at 256^3: <= 3.5 cells over 400 steps (tools/reach_probe.py)
const uint32_t kFreezeStatRing = 1024;   // per-step statistics words of the sparse faithful solver kept on the device
const uint32_t kDefaultJacobiHalo = 8;   // sweeps per pressure exchange: 5 messages per 40 sweeps, +11% halo sweeps at 64 planes/rank

hipStream_t pick_stream(fx_ctx* ctx, void* s) { return s ? (hipStream_t)s : ctx->stream; }

size_t elem_size(const fx_ctx* c) { return c->half ? 2 : 4; }

struct DeviceGuard {
	int prev = -1;
	bool ok = true;
	explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; }
	~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- timing ---------------------------------------------------------------------------------
enum MarkKind { MK_ADVECT, MK_DIV, MK_JACOBI, MK_PROJECT, MK_LIGHT, MK_VIEW, MK_EXCH, MK_RESOLVE, MK_JACOBI_TAIL, MK_CHAIN };

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

struct ScopedMark {
	fx_ctx* c; hipStream_t s; int kind; size_t e0; uint64_t launches, sweeps;
	ScopedMark(fx_ctx* c_, hipStream_t s_, int kind_) : c(c_), s(s_), kind(kind_), e0((size_t)-1), launches(0), sweeps(0)
	{
		if (c->timing_on) e0 = ev_record(c, s);
	}
	// close the mark here and continue as `new_kind` from the same event (one event more, no gap)
	void split(int new_kind)
	{
		if (c->timing_on && e0 != (size_t)-1) {
			const size_t e1 = ev_record(c, s);
			if (e1 != (size_t)-1) { c->marks.push_back(fx_ctx::Mark{ kind, e0, e1, launches, sweeps }); e0 = e1; }
		}
		kind = new_kind; launches = 0; sweeps = 0;
	}
	~ScopedMark()
	{
		if (c->timing_on && e0 != (size_t)-1) {
			const size_t e1 = ev_record(c, s);
			if (e1 != (size_t)-1) c->marks.push_back(fx_ctx::Mark{ kind, e0, e1, launches, sweeps });
		}
	}
};

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

void free_all(fx_ctx* c)
{
	for (int i = 0; i < 2; ++i) {
		if (c->vel[i]) (void)hipFree(c->vel[i]);
		if (c->col[i]) (void)hipFree(c->col[i]);
		if (c->p[i]) (void)hipFree(c->p[i]);
	}
	void* others[] = { c->env, c->occ, c->target, c->target_float, c->p_face[0], c->p_face[1], c->b, c->frozen, c->lightmap, c->cube, c->sh_dev, c->halo_overflow, c->stage,
		c->sh_scratch[0], c->sh_scratch[1], c->sh_scratch[2], c->sh_scratch[3], c->p_aux, c->fz_mask[0], c->fz_mask[1], c->fz_tile_next, c->fz_stat, c->fz_list[0], c->fz_list[1], c->fz_counts, c->sample_counters };
	for (void* q : others) if (q) (void)hipFree(q);
	for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
	if (c->step_rec) (void)hipFree(c->step_rec);
	if (c->gath_dev) (void)hipFree(c->gath_dev);
	if (c->rec_host) (void)hipHostFree(c->rec_host);
	if (c->rec_ev) (void)hipEventDestroy(c->rec_ev);
	if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
}

// planes of the local array a stage may compute / read, as global z ranges
struct Range { int lo, hi; };   // [lo, hi)
Range owned(const fx_ctx* c) { return Range{ c->g.z0, c->g.z0 + c->g.nz }; }
Range grown(const fx_ctx* c, int by)
{
	return Range{ std::max(c->g.z0 - by, 0), std::min(c->g.z0 + c->g.nz + by, c->g.Zg) };
}

// ---- the simulation step, phase by phase, over a group of slab contexts ----------------------------
// (Fluid::Simulate, Fluid.cpp:348-410; the phase structure is what lets one code path serve the
// single-GPU case, the RCCL slabs and the in-process loop-back slabs)
//
// Multi-rank schedule of one step (k = sweeps per pressure exchange, Ha = advect halo):
//   1  [comm] exchange Ha planes of velocity + colour      || [compute] advect the planes >= Ha away from a slab face
//      then advect the 2 x Ha face planes
//   2  exchange 1 plane of the advected uz ; divergence on the owned planes
//   3  exchange k-1 planes of b and k planes of p (one message group)
//   4  per round of k sweeps: the k planes next to each face are brought to the round's last level first
//      (thin single-sweep launches over both face zones, from the exchanged halo), [comm] they travel to the
//      neighbour || [compute] the interior follows with the fused-sweep kernels (see jacobi_overlapped)
//   5  projection (reads the 1st halo plane of the last exchange)
// Every cell is computed with the arithmetic of the single-domain run, so results are bit-identical.
int for_members(fx_ctx* ctx, std::vector<fx_ctx*>& out)
{
	out.clear();
	if (ctx->group && ctx->group->transport->is_local() && !ctx->group->broken) out = ctx->group->members;
	else out.push_back(ctx);                           // (a broken loop-back group: only this context, and only for teardown)
	return FX_OK;
}

bool multi_rank(const fx_ctx* c) { return c->group && c->nranks > 1; }

// 0 = no side stream, 1 = advection halo overlapped, 2 = pressure rounds overlapped as well, 3 = and the colour half of the
// next step's advection halo travels behind this step's pressure phase (fx_set_option)
int overlap_level(const fx_ctx* lead)
{
	if (!multi_rank(lead) || !lead->group->comm_stream) return 0;
	return lead->opt_overlap;
}

struct ExchSpec { int set, k, pidx; };

int do_exchange(fx_ctx* ctx, const std::vector<fx_ctx*>& M, const ExchSpec* specs, int nspec, hipStream_t s, int channel = 0)
{
	if (!multi_rank(ctx)) return FX_OK;
	DeviceGuard dg(ctx->device);
	ScopedMark mk(ctx, s, MK_EXCH);
	std::vector<std::vector<Seg>> segs(M.size());
	size_t total = 0;
	for (size_t i = 0; i < M.size(); ++i) {
		for (int j = 0; j < nspec; ++j) {
			if (specs[j].k <= 0) continue;
			ExchItem it[4];
			const int n = exchange_items(M[i], specs[j].set, specs[j].k, specs[j].pidx, it);
			halo_segments(M[i], it, n, segs[i]);
		}
		total += segs[i].size();
		if (M[i]->timing_on) for (const Seg& sg : segs[i]) M[i]->acc.exchange_bytes += sg.bytes;     // what this rank sends
	}
	if (!total) return FX_OK;
	for (fx_ctx* m : M) if (m->timing_on) m->acc.exchange_calls += 1;  // one group call (ncclGroupStart .. End) per exchange
	return ctx->group->transport->exchange(ctx->group, segs, s, channel);
}

// comm stream picks up after everything queued on the compute stream so far
int comm_fork(fx_ctx* ctx, hipStream_t s)
{
	fx_comm_group* g = ctx->group;
	FX_HIP(hipEventRecord(g->ev_ready, s));
	FX_HIP(hipStreamWaitEvent(g->comm_stream, g->ev_ready, 0));
	return FX_OK;
}
int comm_mark_done(fx_ctx* ctx) { FX_HIP(hipEventRecord(ctx->group->ev_done, ctx->group->comm_stream)); return FX_OK; }
int comm_join(fx_ctx* ctx, hipStream_t s) { FX_HIP(hipStreamWaitEvent(s, ctx->group->ev_done, 0)); return FX_OK; }

bool has_lower(const fx_ctx* c) { return c->nranks > 1 && c->rank > 0; }
bool has_upper(const fx_ctx* c) { return c->nranks > 1 && c->rank + 1 < c->nranks; }

// advect planes [r.lo, r.hi); own_only: back-traces must stay inside the owned planes (the halo is still in flight)
int advect_range(fx_ctx* ctx, hipStream_t s, Range r, bool own_only)
{
	if (r.hi <= r.lo) return FX_OK;
	DeviceGuard dg(ctx->device);
	const SimParams sp{ ctx->time_step, (int)ctx->desc.advect_address, ctx->g.Zg > 1 ? 1 : 0 };
	const int par = ctx->frame_parity;
	Geom g = ctx->g;
	// only halo_advect planes per side were refreshed by EX_ADVECT_IN; the allocation may be wider (max with halo_jacobi), and
	// a tap into those stale planes must count as "left the exchanged halo", not as present data
	// (with FX_OPT_ADAPTIVE_HALO: only the planes this step's exchange carried, adv_w_lo / adv_w_hi <= halo_advect)
	g.zlo = std::max(g.zlo, g.z0 - ctx->adv_w_lo); g.zhi = std::min(g.zhi, g.z0 + g.nz - 1 + ctx->adv_w_hi);
	if (own_only) { g.zlo = std::max(g.zlo, g.z0); g.zhi = std::min(g.zhi, g.z0 + g.nz - 1); }
	FX_HIP(launch_advect(g, sp, ctx->half, ctx->vel[0], ctx->col[1 - par], ctx->vel[1], ctx->col[par],
		r.lo, r.hi, ctx->halo_overflow, s));
	return FX_OK;
}

// ---- the per-step record (fx_context.h): written behind the projection, read by the next step ------------------------------
int options_digest(const fx_ctx* c)
{
	uint32_t h = 2166136261u;
	uint32_t dt_bits;
	std::memcpy(&dt_bits, &c->time_step, 4);               // the time step the record's needs were measured with: the ranks must agree on it too
	for (uint32_t v : { (uint32_t)c->opt_overlap, (uint32_t)c->opt_round, (uint32_t)c->opt_adaptive, dt_bits }) h = (h ^ v) * 16777619u;
	return (int)(h & 0x3FFFFFFFu);
}

int record_step(fx_ctx* ctx, std::vector<fx_ctx*>& M, hipStream_t s)
{
	if (!multi_rank(ctx) || !ctx->step_rec) return FX_OK;
	for (fx_ctx* m : M) {
		if (m->rec_in_project) { m->rec_in_project = false; continue; }       // k_project_v4 has already written it
		DeviceGuard dg(m->device);
		FX_HIP(launch_face_need(m->g, m->half, m->vel[0], m->time_step, (int)m->desc.advect_address, options_digest(m), m->halo_overflow, m->step_rec, s));
	}
	DeviceGuard dg(ctx->device);
	if (ctx->group->transport->is_local()) {
		for (fx_ctx* m : M) FX_HIP(hipMemcpyAsync(m->rec_host, m->step_rec, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
		FX_HIP(hipEventRecord(ctx->rec_ev, s));
	} else {
		// off the compute stream when there is a side stream: the next step's interior advection need not wait for the gather
		hipStream_t cs = s;
		if (overlap_level(ctx) >= 1) { int rc = comm_fork(ctx, s); if (rc) return rc; cs = ctx->group->comm_stream; }
		if (ctx->group->transport->allgather(ctx->step_rec, 4, ctx->gath_dev, cs) != FX_OK) { ctx->last_error = "rccl: all-gather of the step record failed"; return FX_E_COMM; }
		FX_HIP(hipMemcpyAsync(ctx->rec_host, ctx->gath_dev, 4 * sizeof(int) * (size_t)ctx->nranks, hipMemcpyDeviceToHost, cs));
		FX_HIP(hipEventRecord(ctx->rec_ev, cs));
	}
	for (fx_ctx* m : M) { m->rec_pending = true; m->need_valid = true; m->rec_dt = m->time_step; }
	return FX_OK;
}

// Waits for the previous step's record and takes the step's decisions from it -- every rank holds the same records and therefore
// decides alike: FX_E_HALO if ANY rank's advection left its exchanged planes (or the next one would need more than halo_advect),
// FX_E_STATE if the ranks disagree about the schedule options; else the planes this step's advection exchange carries per face
// (FX_OPT_ADAPTIVE_HALO: the measured need of the two slabs that share the face; otherwise, or when the measurement does not
// cover this step -- first step, velocity uploaded since, larger dt -- the whole halo_advect).
int consume_record(fx_ctx* ctx, std::vector<fx_ctx*>& M)
{
	const int Ha = (int)ctx->desc.halo_advect;
	for (fx_ctx* m : M) { m->adv_w_lo = has_lower(m) ? Ha : 0; m->adv_w_hi = has_upper(m) ? Ha : 0; }
	if (!multi_rank(ctx) || !ctx->rec_pending) return FX_OK;
	{
		DeviceGuard dg(ctx->device);
		FX_HIP(hipEventSynchronize(ctx->rec_ev));
	}
	const bool local = ctx->group->transport->is_local();
	const int n = ctx->nranks;
	auto rec = [&](int r) -> const int* { return local ? M[(size_t)r]->rec_host : ctx->rec_host + 4 * r; };
	bool fault = false, mismatch = false, usable = ctx->opt_adaptive != 0;
	for (int r = 0; r < n; ++r) { fault = fault || rec(r)[3] != 0; mismatch = mismatch || rec(r)[2] != rec(0)[2]; }
	for (fx_ctx* m : M) { usable = usable && m->need_valid && m->time_step <= m->rec_dt; m->rec_pending = false; }
	if (fault) {
		// This return IS the chain-wide notice of the fault: every rank gets it once, from the same gathered record, and the step
		// after it starts clean.  It does not block read-back again: the rank whose own advection overflowed still has its device
		// flag up until its fx_synchronize acknowledges it (before or after this call), the other ranks have nothing to acknowledge.
		// (Before: this set halo_fault on every rank, so the order fx_synchronize -> fx_simulate reported the same fault three times.)
		ctx->last_error = "the previous step's advection left the exchanged halo on at least one rank";
		return FX_E_HALO;
	}
	if (mismatch) { ctx->last_error = "the ranks of the chain run with different schedule options (fx_set_option) or time steps"; return FX_E_STATE; }
	if (!usable) return FX_OK;
	for (int r = 0; r + 1 < n; ++r)
		if (std::max(rec(r)[1], rec(r + 1)[0]) > Ha) {
			ctx->last_error = "the next advection needs more planes across a slab face than halo_advect provides";
			return FX_E_HALO;                            // the step is abandoned; its INPUTS (velocity[0], colour[!parity], pressure) are untouched -- in the
			                                             // overlapped schedule the interior advection has already written part of its outputs (velocity[1], colour[parity])
		}
	for (fx_ctx* m : M) {
		const int r = m->rank;
		m->adv_w_lo = r > 0 ? std::max(rec(r - 1)[1], rec(r)[0]) : 0;
		m->adv_w_hi = r + 1 < n ? std::max(rec(r)[1], rec(r + 1)[0]) : 0;
	}
	return FX_OK;
}

int advect_all(fx_ctx* ctx, std::vector<fx_ctx*>& M, hipStream_t s)
{
	int rc;
	const int Ha = (int)ctx->desc.halo_advect;
	// FX_OPT_OVERLAP 3: the previous step already sent the colour planes this advection gathers from (simulate_impl); only
	// the velocity, which the projection has just finished, travels now.  Every rank of a chain takes the same branch: the
	// flag follows from the option level and the step history alone (a colour upload in between is refused, fx_upload).
	bool col_ready = multi_rank(ctx);
	for (fx_ctx* m : M) col_ready = col_ready && m->col_halo_buf == 1 - (int)m->frame_parity;
	for (fx_ctx* m : M) m->col_halo_buf = -1;
	bool ov = overlap_level(ctx) >= 1;
	if (ov && ctx->group->min_nz <= 2 * Ha) ov = false;        // decided on the thinnest slab of the chain: the same on every rank
	if (!ov) {
		if ((rc = consume_record(ctx, M))) return rc;
		if (col_ready) FX_HIP(hipStreamWaitEvent(s, ctx->group->ev_col_done, 0));
		const ExchSpec spec{ col_ready ? EX_ADVECT_VEL : EX_ADVECT_IN, Ha, 0 };
		if ((rc = do_exchange(ctx, M, &spec, 1, s))) return rc;
		for (fx_ctx* m : M) {
			ScopedMark mk(m, s, MK_ADVECT);
			if (m->timing_on) m->acc.advect_halo_planes += (uint64_t)(m->adv_w_lo + m->adv_w_hi);
			if ((rc = advect_range(m, s, owned(m), false))) return rc;
		}
		return FX_OK;
	}
	// the interior first (it reads owned planes only, whatever the exchange will carry): the device is busy while the host waits
	// for the previous step's record, which sizes the exchange
	if ((rc = comm_fork(ctx, s))) return rc;                   // the comm stream picks up behind the previous step
	for (fx_ctx* m : M) {
		ScopedMark mk(m, s, MK_ADVECT);
		const Range o = owned(m);
		if ((rc = advect_range(m, s, Range{ o.lo + (has_lower(m) ? Ha : 0), o.hi - (has_upper(m) ? Ha : 0) }, true))) return rc;
	}
	if ((rc = consume_record(ctx, M))) return rc;
	if (col_ready) FX_HIP(hipStreamWaitEvent(ctx->group->comm_stream, ctx->group->ev_col_done, 0));
	const ExchSpec spec{ col_ready ? EX_ADVECT_VEL : EX_ADVECT_IN, Ha, 0 };
	if ((rc = do_exchange(ctx, M, &spec, 1, ctx->group->comm_stream))) return rc;
	if ((rc = comm_mark_done(ctx))) return rc;
	if ((rc = comm_join(ctx, s))) return rc;
	for (fx_ctx* m : M) {
		ScopedMark mk(m, s, MK_ADVECT);
		if (m->timing_on) m->acc.advect_halo_planes += (uint64_t)(m->adv_w_lo + m->adv_w_hi);
		const Range o = owned(m);
		if (has_lower(m) && (rc = advect_range(m, s, Range{ o.lo, o.lo + Ha }, false))) return rc;
		if (has_upper(m) && (rc = advect_range(m, s, Range{ o.hi - Ha, o.hi }, false))) return rc;
	}
	return FX_OK;
}

int divergence_phase(fx_ctx* ctx, hipStream_t s)
{
	DeviceGuard dg(ctx->device);
	ScopedMark mk(ctx, s, MK_DIV);
	const Range r = owned(ctx);
	FX_HIP(launch_divergence(ctx->g, ctx->half, ctx->vel[1], ctx->b, r.lo, r.hi, s));
	return FX_OK;
}

// t lock-step sweeps p[src] -> p[src ^ 1] on planes [r.lo, r.hi) in ONE launch
int jacobi_launch(fx_ctx* ctx, hipStream_t s, int src, int t, Range r, ScopedMark* mk)
{
	r.lo = std::max(r.lo, 0); r.hi = std::min(r.hi, ctx->g.Zg);
	if (r.hi <= r.lo) return FX_OK;
	DeviceGuard dg(ctx->device);
	if (t > 1) {
		FX_HIP(launch_jacobi_fused(ctx->g, ctx->p[src], ctx->b, ctx->p[src ^ 1], t, r.lo, r.hi, s));
	} else {
		FX_HIP(launch_jacobi_sweep(ctx->g, ctx->p[src], ctx->b, ctx->p[src ^ 1], ctx->frozen, r.lo, r.hi, s));
	}
	if (mk) { mk->launches += 1; mk->sweeps += t; }
	return FX_OK;
}

int fused_sweeps(const fx_ctx* c)
{
	return c->frozen ? 1 : jacobi_fused_max_sweeps(c->g, (int)(c->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), c->g.nz);
}

// `count` lock-step sweeps whose first one may read `count` exchanged halo planes; the planes swept shrink by
// one per sweep towards the owned range (redundant halo work instead of an exchange per sweep)
int jacobi_round(fx_ctx* ctx, hipStream_t s, int count, ScopedMark* mk)
{
	int done = 0;
	while (done < count) {
		const int left = count - done;
		int t = std::min(left, fused_sweeps(ctx));
		if (!ctx->frozen && jacobi_prefers_three(ctx->g, (int)(ctx->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), ctx->g.nz))
			t = left == 4 ? 2 : std::min(left, 3);           // threes, and a remainder of 4 as 2 + 2 rather than 3 + 1
		if (mk && mk->kind == MK_JACOBI && mk->launches && t * mk->launches < mk->sweeps) mk->split(MK_JACOBI_TAIL);   // shorter launches from here on
		const int rc = jacobi_launch(ctx, s, ctx->p_cur, t, grown(ctx, multi_rank(ctx) ? left - t : 0), mk);
		if (rc) return rc;
		ctx->p_cur ^= 1;
		done += t;
	}
	return FX_OK;
}

int clear_freeze_masks(std::vector<fx_ctx*>& M, hipStream_t s)
{
	for (fx_ctx* m : M)
		if (m->frozen) { DeviceGuard dg(m->device); if (hipMemsetAsync(m->frozen, 0, m->g.cells_local(), s) != hipSuccess) return FX_E_DEVICE; }
	return FX_OK;
}

// FX_JACOBI_FAITHFUL on a single domain: the sparse solver of fx_jacobi_freeze.hip.  Level 1 everywhere (into p[other] AND p_aux;
// the input buffer becomes the spare), then ceil((iters - 1) / T) launches over the tiles that still relax, all enqueued; the
// result is in the last launch's output buffer (settled tiles agree in both).  Bit-identical to `iters` generic sweeps with the
// byte mask (tests/test_gpu_sim.py::test_freeze_fast_path_*).
int jacobi_freeze(fx_ctx* ctx, hipStream_t s, uint32_t iters)
{
	DeviceGuard dg(ctx->device);
	ScopedMark mk(ctx, s, MK_JACOBI);
	if (++ctx->fz_gen >= (1u << 23)) {                                  // the tag (gen << 8 | level) of the stat words stays below 2^32: start over
		FX_HIP(hipMemsetAsync(ctx->fz_tile_next, 0, (size_t)jacobi_freeze_tiles(ctx->g) * sizeof(uint32_t), s));
		FX_HIP(hipMemsetAsync(ctx->fz_stat, 0, kFreezeStatRing * sizeof(uint32_t), s));
		FX_HIP(hipMemsetAsync(ctx->fz_counts, 0, 2 * jacobi_freeze_count_words() * sizeof(uint32_t), s));
		ctx->fz_gen = 2; ctx->fz_gen_mark = 0;
	}
	const uint32_t gen = ctx->fz_gen, stat_hi = gen << 8;
	uint32_t* stat = ctx->fz_stat + gen % kFreezeStatRing;
	ctx->fz_iters[gen % kFreezeStatRing] = iters;
	const size_t cw = jacobi_freeze_count_words();
	const FreezeWork w{ ctx->fz_tile_next, gen, { ctx->fz_list[0], ctx->fz_list[1] }, jacobi_freeze_tiles(ctx->g),
		ctx->fz_counts + (gen & 1u) * cw, ctx->fz_counts + ((gen & 1u) ^ 1u) * cw };
	float* src = ctx->p[ctx->p_cur];
	float* a = ctx->p[ctx->p_cur ^ 1];
	float* d = ctx->p_aux;
	uint8_t* ma = ctx->fz_mask[0];
	uint8_t* md = ctx->fz_mask[1];
	// (The dense sweep writes level 1 to BOTH buffers the tile launches alternate between.  Writing one and letting the first tile
	// launch carry the unlisted tiles' border cells across was built and measured level: the dense sweep 75 -> 46 us at 256^3, the
	// first tile launch slower by as much -- the shell of a 4-deep cone around ~3000 listed tiles is more bytes than the second copy.)
	FX_HIP(launch_freeze_dense(ctx->g, src, ctx->b, a, d, ma, md, w, stat, stat_hi, s));
	mk.launches = 1; mk.sweeps = 1;
	if (iters > 1) mk.split(MK_JACOBI_TAIL);                            // fx_timing books the dense sweep as the "main" launch, the tile launches beside it
	const int T = jacobi_freeze_levels_per_launch();
	int level = 1, n = 0;
	for (uint32_t left = iters - 1; left > 0; ++n) {
		const int t = (int)std::min<uint32_t>((uint32_t)T, left);
		FX_HIP(launch_freeze_tiles(ctx->g, a, ctx->b, d, ma, md, w, n, t, level, stat, stat_hi, s));
		std::swap(a, d); std::swap(ma, md);
		left -= (uint32_t)t; level += t;
		mk.launches += 1; mk.sweeps += (uint64_t)t;
	}
	ctx->p[0] = a; ctx->p[1] = d; ctx->p_aux = src; ctx->p_cur = 0;
	return FX_OK;
}

// exchange, then k sweeps, exchange, ... on one stream
int jacobi_serial(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters)
{
	const bool multi = multi_rank(lead);
	const int k = multi ? lead->opt_round : (int)iters;
	int rc;
	if (!multi && lead->frozen && lead->fz_tile_next && jacobi_freeze_supported(lead->g) && iters <= 255 && (int)iters / jacobi_freeze_levels_per_launch() + 3 < kFreezeSlots)   // a level fits the stat word's low byte, every launch has its counters
		return jacobi_freeze(lead, s, iters);
	if ((rc = clear_freeze_masks(M, s))) return rc;
	const ExchSpec bspec{ EX_DIV, k - 1, 0 };
	if ((rc = do_exchange(lead, M, &bspec, 1, s))) return rc;
	uint32_t done = 0;
	while (done < iters) {
		const int cnt = (int)std::min<uint32_t>(k, iters - done);
		const ExchSpec pspec{ EX_PRESSURE, cnt, lead->p_cur };
		if ((rc = do_exchange(lead, M, &pspec, 1, s))) return rc;
		for (fx_ctx* m : M) {
			ScopedMark mk(m, s, MK_JACOBI);
			if ((rc = jacobi_round(m, s, cnt, &mk))) return rc;
		}
		done += cnt;
	}
	const ExchSpec last{ EX_PRESSURE, 1, lead->p_cur };          // the projection's z-gradient reads one plane across the face
	return do_exchange(lead, M, &last, 1, s);
}

// Rounds of up to k sweeps with the pressure exchange of a round hidden behind its interior sweeps, on three streams.
// With lo/hi = the owned planes, src = the buffer holding the round's level 0 (k halo planes valid), cnt <= k sweeps in the
// round, done as m launches of t_1 <= t_2 = ... = t_m fused sweeps (c_j = t_1 + ... + t_j, rem_j = cnt - c_j):
//   face stream   the FACE CHAIN: cnt single sweeps over both face zones per launch, level s on [lo - (cnt - s), lo + k + (cnt - s))
//                 (mirrored at hi), entirely in two scratch buffers (only its first sweep reads src): thin, latency-bound
//                 launches that run BESIDE the interior instead of in front of it.  Its last level holds the k planes the
//                 neighbour needs.
//   comm stream   after the chain: those k planes leave from the scratch buffer, the neighbour's land in the halo of the
//                 round's last buffer (which no interior launch touches)
//   compute       the INTERIOR, self-sufficient: launch j brings [lo + k - rem_j, hi - k + rem_j) from level c_(j-1) to c_j,
//                 i.e. it recomputes the rem_j planes per side the chain also computes instead of waiting for them (it never
//                 reads below lo + k - cnt >= lo, so it needs no halo).  Launch 2 overwrites src and therefore waits until
//                 the chain's first sweep has read it; after the last launch the chain's k final planes are copied from the
//                 scratch buffer into [lo, lo + k) of the round's last buffer, which completes the owned planes.
// Per round the critical path is max(interior, chain + link) instead of chain + max(interior, link).  Every cell gets the
// arithmetic of the single-domain sweep; (cell, level) pairs of the zone borders are computed twice, which is why the
// faithful mode (its freeze mask is a side effect) takes the serial schedule instead.
int jacobi_overlapped(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters, int t, int k)
{
	fx_comm_group* grp = lead->group;
	hipStream_t fs = grp->face_stream, cs = grp->comm_stream;
	int rc;
	fx_ctx* ctx = lead;                                    // FX_HIP reports through `ctx`
	// Sweeps per launch of the face chain.  Default 1: one single-sweep launch (k_jacobi_v4: 60 registers, both faces) per level.
	// FLUIDX_CHAIN_FUSE=2|3 runs the chain in groups of fused sweeps with the interior's register-strip kernels instead (3 + 3 + 3
	// for a round of nine; identical results) -- measured SLOWER (loop-back N = 4, 256^3 per rank, rounds of 9: 6.30 against
	// 5.67 ms per step; rounds of 6: 6.81 / 5.79; of 3: 6.93 / 6.32; profiles/r02c_chain_fuse_loopback4.txt): on a 9..27-plane zone
	// the strips have 64..192 waves whose 310 registers shut the interior's waves out of their SIMDs for a whole 14-step
	// pipeline, where a single-sweep launch is a few microseconds of small waves beside them.
	int tc = 3;
	{
		static const int forced = [] { const char* e = std::getenv("FLUIDX_CHAIN_FUSE"); return e && *e ? std::atoi(e) : 1; }();
		for (fx_ctx* mctx : M) {
			const int cap = jacobi_strip3_supported(mctx->g) ? 3 : (jacobi_fused_max_sweeps(mctx->g, 2, 2 * k) >= 2 ? 2 : 1);
			tc = std::min(tc, cap);
		}
		if (forced >= 1 && forced <= 3) tc = std::min(tc, forced);
	}
	const ExchSpec first[2] = { { EX_DIV, k - 1, 0 }, { EX_PRESSURE, k, lead->p_cur } };
	if ((rc = do_exchange(lead, M, first, 2, s))) return rc;
	FX_HIP(hipEventRecord(grp->ev_int, s));
	bool in_flight = false;
	uint32_t done = 0;
	while (done < iters) {
		const int cnt = (int)std::min<uint32_t>(k, iters - done);
		const int m = (cnt + t - 1) / t, t_first = cnt - (m - 1) * t;
		const int src = lead->p_cur, fin = src ^ (m & 1);
		int fbuf = 0;                                      // which scratch buffer holds the chain's last level (set below)
		// ---- face stream: the chain (needs the previous round's interior + face copy, and its exchange)
		FX_HIP(hipStreamWaitEvent(fs, grp->ev_int, 0));
		if (in_flight) FX_HIP(hipStreamWaitEvent(fs, grp->ev_done, 0));
		ScopedMark chain_mark(lead, fs, MK_CHAIN);         // one mark per chain (loop-back: all members' chains, booked on the first)
		{
			// the chain in groups of tc sweeps over the two thin face zones (tc = 1 by default, see above)
			int c = 0, grp_i = 0;
			while (c < cnt) {
				const int left = cnt - c;
				int tg = std::min(left, tc);
				if (tc == 3 && left == 4) tg = 2;              // 2 + 2 rather than 3 + 1
				c += tg;
				const int rem = cnt - c, ob = (grp_i + 1) & 1;
				for (fx_ctx* mctx : M) {
					if (!has_lower(mctx) && !has_upper(mctx)) continue;
					DeviceGuard dg(mctx->device);
					const Range o = owned(mctx);
					const float* in = grp_i == 0 ? mctx->p[src] : mctx->p_face[grp_i & 1];
					const Range lo{ o.lo - rem, has_lower(mctx) ? o.lo + k + rem : o.lo - rem };
					const Range hi{ has_upper(mctx) ? o.hi - k - rem : o.hi + rem, o.hi + rem };
					if (tg == 1) {
						FX_HIP(launch_jacobi_sweep2(mctx->g, in, mctx->b, mctx->p_face[ob], nullptr, lo.lo, lo.hi, hi.lo, hi.hi, fs));
					} else {
						if (lo.hi > lo.lo) FX_HIP(launch_jacobi_fused(mctx->g, in, mctx->b, mctx->p_face[ob], tg, lo.lo, lo.hi, fs));
						if (hi.hi > hi.lo) FX_HIP(launch_jacobi_fused(mctx->g, in, mctx->b, mctx->p_face[ob], tg, hi.lo, hi.hi, fs));
					}
				}
				if (grp_i == 0) FX_HIP(hipEventRecord(grp->ev_face1, fs));
				++grp_i;
			}
			fbuf = grp_i & 1;                                  // the buffer the last group wrote
		}
		FX_HIP(hipEventRecord(grp->ev_ready, fs));
		// ---- comm stream: the k final planes of the chain travel, the neighbour's land in the halo of p[fin]
		FX_HIP(hipStreamWaitEvent(cs, grp->ev_ready, 0));
		const ExchSpec pspec{ EX_PRESSURE_FACE, k, (fbuf << 1) | fin };
		if ((rc = do_exchange(lead, M, &pspec, 1, cs))) return rc;
		if ((rc = comm_mark_done(lead))) return rc;
		in_flight = true;
		// ---- compute stream: the interior
		for (fx_ctx* mctx : M) {
			ScopedMark mk(mctx, s, MK_JACOBI);
			const Range o = owned(mctx);
			int lvl = 0, cur = src;
			for (int j = 0; j < m; ++j) {
				const int tj = j == 0 ? t_first : t;
				lvl += tj;
				const int rem = cnt - lvl;
				const Range in{ has_lower(mctx) ? o.lo + k - rem : o.lo, has_upper(mctx) ? o.hi - k + rem : o.hi };
				if (j == 1 && mctx == M.front()) FX_HIP(hipStreamWaitEvent(s, grp->ev_face1, 0));   // launch 2 overwrites the chain's input
				if ((rc = jacobi_launch(mctx, s, cur, tj, in, &mk))) return rc;
				cur ^= 1;
			}
		}
		// the chain's final planes complete the owned range of p[fin]
		FX_HIP(hipStreamWaitEvent(s, grp->ev_ready, 0));
		for (fx_ctx* mctx : M) {
			DeviceGuard dg(mctx->device);
			const size_t pl = mctx->g.plane(), kb = (size_t)k * pl * 4;
			const Range o = owned(mctx);
			if (has_lower(mctx))
				FX_HIP(launch_copy_bytes(mctx->p[fin] + (size_t)mctx->g.lz(o.lo) * pl, mctx->p_face[fbuf] + (size_t)mctx->g.lz(o.lo) * pl, kb, s));
			if (has_upper(mctx))
				FX_HIP(launch_copy_bytes(mctx->p[fin] + (size_t)mctx->g.lz(o.hi - k) * pl, mctx->p_face[fbuf] + (size_t)mctx->g.lz(o.hi - k) * pl, kb, s));
			mctx->p_cur = fin;
		}
		FX_HIP(hipEventRecord(grp->ev_int, s));
		done += cnt;
	}
	if (in_flight) rc = comm_join(lead, s);
	return rc;
}

int jacobi_all(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters)
{
	if (overlap_level(lead) >= 2) {
		int t = fused_sweeps(lead);
		bool three = true;
		for (fx_ctx* m : M) {
			t = std::min(t, fused_sweeps(m));
			three = three && !m->frozen && jacobi_prefers_three(m->g, (int)(m->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), m->g.nz);
		}
		if (three) t = 3;                              // the interior launches of a round as threes (k = 9: 3 + 3 + 3); local choice, the exchanges do not depend on it
		const int k = lead->opt_round;
		// two face zones (<= 2k - 1 planes each) and an interior; decided on the thinnest slab of the chain and on the (chain-wide)
		// Jacobi mode, so that every rank takes the same branch -- the two schedules exchange different things
		const bool ok = lead->group->face_stream != nullptr && lead->group->min_nz >= 4 * k && lead->p_face[0] && !lead->frozen;
		if (ok) return jacobi_overlapped(lead, M, s, iters, t, k);
	}
	return jacobi_serial(lead, M, s, iters);
}

int options_digest(const fx_ctx* c);

int project_phase(fx_ctx* ctx, hipStream_t s)
{
	DeviceGuard dg(ctx->device);
	const SimParams sp{ ctx->time_step, (int)ctx->desc.advect_address, ctx->g.Zg > 1 ? 1 : 0 };
	ScopedMark mk(ctx, s, MK_PROJECT);
	const Range r = owned(ctx);
	int* rec = multi_rank(ctx) ? ctx->step_rec : nullptr;           // slab ranks: the projection also measures the next advection's need
	ctx->rec_in_project = false;
	FX_HIP(launch_project(ctx->g, sp, ctx->half, ctx->vel[1], ctx->p[ctx->p_cur], ctx->vel[0], r.lo, r.hi, s,
		rec, rec ? options_digest(ctx) : 0, ctx->halo_overflow, &ctx->rec_in_project));
	return FX_OK;
}

int simulate_impl(fx_ctx* ctx, hipStream_t s)
{
	std::vector<fx_ctx*> M;
	for_members(ctx, M);
	int rc;
	if ((rc = advect_all(ctx, M, s))) return rc;
	if (overlap_level(ctx) >= 3) {
		// colour[parity] is final for this step: its halo planes -- four of the seven plane-units the next advection needs --
		// leave now on the side stream, behind divergence / pressure / projection
		fx_comm_group* g = ctx->group;
		FX_HIP(hipEventRecord(g->ev_col_ready, s));
		FX_HIP(hipStreamWaitEvent(g->comm_stream, g->ev_col_ready, 0));
		const ExchSpec cs{ EX_COLOR_CUR, (int)ctx->desc.halo_advect, 0 };
		if ((rc = do_exchange(ctx, M, &cs, 1, g->comm_stream, 1))) return rc;       // side channel: not queued with the step's own exchanges
		FX_HIP(hipEventRecord(g->ev_col_done, g->comm_stream));
		for (fx_ctx* m : M) m->col_halo_buf = (int)m->frame_parity;
	}
	if (ctx->time_step > 0.0f) {                       // CSProject3D.hlsl:88
		const ExchSpec uz{ EX_UZ1, 1, 0 };
		if ((rc = do_exchange(ctx, M, &uz, 1, s))) return rc;
		for (fx_ctx* m : M) if ((rc = divergence_phase(m, s))) return rc;
		if ((rc = jacobi_all(ctx, M, s, ctx->desc.jacobi_iters))) return rc;
		for (fx_ctx* m : M) if ((rc = project_phase(m, s))) return rc;
	} else {
		for (fx_ctx* m : M) {
			DeviceGuard dg(m->device);
			m->rec_in_project = false;
			if (launch_copy_velocity(m->g, m->half, m->vel[1], m->vel[0], s) != hipSuccess) return FX_E_DEVICE;
		}
	}
	if ((rc = record_step(ctx, M, s))) return rc;
	for (fx_ctx* m : M) { if (m->timing_on) m->acc.steps += 1; if (ctx->time_step > 0.0f) m->steps_simulated += 1; }
	return FX_OK;
}

bool is_driver(const fx_ctx* c) { return !c->group || !c->group->transport->is_local() || c->group->members[0] == c; }
bool group_broken(const fx_ctx* c) { return c->group && c->group->transport->is_local() && c->group->broken; }

}  // namespace

// ===================================================================================================
extern "C" {

int fx_abi_version(void) { return FX_ABI_VERSION; }

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
			if (jacobi_freeze_supported(ctx->g)) {                           // the sparse solver of fx_jacobi_freeze.hip
				const size_t mb = jacobi_freeze_mask_bytes(ctx->g), nt = (size_t)jacobi_freeze_tiles(ctx->g);
				FX_HIP(hipMalloc((void**)&ctx->p_aux, cells * 4));
				FX_HIP(hipMemsetAsync(ctx->p_aux, 0, cells * 4, ctx->stream));
				for (int i = 0; i < 2; ++i) {
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
			}
		}
		FX_HIP(hipMalloc((void**)&ctx->halo_overflow, sizeof(unsigned)));
		FX_HIP(hipMemsetAsync(ctx->halo_overflow, 0, sizeof(unsigned), ctx->stream));
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
			const size_t ncell = (size_t)((d->grid_x + 3) / 4) * ((d->grid_y + 3) / 4) * ((d->grid_z + 3) / 4);
			FX_HIP(hipMalloc((void**)&ctx->occ, 2 * ncell * sizeof(float)));       // the grid + the per-block maxima it is dilated from
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
		if (--g->refs == 0) {
			if (g->shared_stream) (void)hipStreamDestroy(g->shared_stream);
			if (g->comm_stream) (void)hipStreamDestroy(g->comm_stream);
			if (g->face_stream) (void)hipStreamDestroy(g->face_stream);
			if (g->ev_int) (void)hipEventDestroy(g->ev_int);
			if (g->ev_face1) (void)hipEventDestroy(g->ev_face1);
			if (g->ev_col_ready) (void)hipEventDestroy(g->ev_col_ready);
			if (g->ev_col_done) (void)hipEventDestroy(g->ev_col_done);
			if (g->ev_ready) (void)hipEventDestroy(g->ev_ready);
			if (g->ev_done) (void)hipEventDestroy(g->ev_done);
			delete g->transport;
			delete g;
		}
	}
	free_all(ctx);
	delete ctx;
	return FX_OK;
}

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
	return simulate_impl(ctx, pick_stream(ctx, stream));
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

// the occupancy grid of this frame's colour field (FLUIDX_RENDER_OCCUPANCY=0 switches the empty-space skipping off: A/B runs);
// its cost is booked on the light/view pass that follows
static const float* render_occupancy(fx_ctx* ctx, const void* color, hipStream_t s)
{
	const char* e = std::getenv("FLUIDX_RENDER_OCCUPANCY");
	if ((e && e[0] == '0') || !ctx->occ) return nullptr;
	if (launch_occupancy(ctx->g, ctx->half, color, ctx->occ, s) != hipSuccess) return nullptr;
	return ctx->occ;
}

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
	unsigned long long* cnt = ctx->opt_count_samples ? ctx->sample_counters : nullptr;
	if (cnt && (flags & FX_SEPARATE_LIGHT_PASS)) ctx->acc.light_samples += (uint64_t)ctx->g.cells_owned();   // the light pass's own fetch per voxel
	if (!(flags & FX_RAY_MARCH_CUBEMAP)) {
		// direct screen-space marching (Fluid.cpp:432-443): one ray per pixel, straight onto the render target
		DeviceGuard dgd(ctx->device);
		hipStream_t sd = pick_stream(ctx, stream);
		int rc = ensure_target(ctx, sd);
		if (rc) return rc;
		const void* colord = ctx->col[ctx->frame_parity];
		const int W = (int)ctx->desc.viewport_w, H = (int)ctx->desc.viewport_h;
		const float* occd = nullptr;                        // built inside the first pass's timing mark: its cost belongs to the frame
		if (flags & FX_SEPARATE_LIGHT_PASS) {
			{
				ScopedMark mk(ctx, sd, MK_LIGHT);
				occd = render_occupancy(ctx, colord, sd);
				FX_HIP(launch_raymarch_light(ctx->g, ctx->half, colord, ctx->lightmap, ctx->fc,
					ctx->has_sh ? ctx->sh_dev : nullptr, ctx->max_light_samples, occd, sd, cnt));
			}
			ScopedMark mk(ctx, sd, MK_VIEW);
			FX_HIP(launch_raycast_direct(ctx->g, ctx->half, colord, ctx->lightmap, ctx->fc, nullptr, W, H,
				ctx->ray_samples, ctx->max_light_samples, 1, ctx->target, ctx->target_float, occd, sd, cnt));   // rayCastVDirect :953-972
		} else {
			ScopedMark mk(ctx, sd, MK_VIEW);
			occd = render_occupancy(ctx, colord, sd);
			FX_HIP(launch_raycast_direct(ctx->g, ctx->half, colord, nullptr, ctx->fc, ctx->has_sh ? ctx->sh_dev : nullptr, W, H,
				ctx->max_ray_samples, ctx->max_light_samples, 0, ctx->target, ctx->target_float, occd, sd, cnt));   // rayCastDirect :932-951
		}
		if (ctx->timing_on) ctx->acc.renders += 1;
		return FX_OK;
	}
	hipStream_t s = pick_stream(ctx, stream);
	DeviceGuard dg(ctx->device);
	const void* color = ctx->col[ctx->frame_parity];
	const int size = ctx->g.X >> ctx->cube_lod;
	uint8_t* cube = ctx->cube + ctx->cube_mip_offset[ctx->cube_lod];
	const float* occ = nullptr;
	if (flags & FX_SEPARATE_LIGHT_PASS) {
		{
			ScopedMark mk(ctx, s, MK_LIGHT);
			occ = render_occupancy(ctx, color, s);
			FX_HIP(launch_raymarch_light(ctx->g, ctx->half, color, ctx->lightmap, ctx->fc,
				ctx->has_sh ? ctx->sh_dev : nullptr, ctx->max_light_samples, occ, s, cnt));     // Fluid.cpp:857-878
		}
		ScopedMark mk(ctx, s, MK_VIEW);
		FX_HIP(launch_raymarch_view(ctx->g, ctx->half, color, ctx->lightmap, ctx->fc, nullptr, size,
			ctx->visibility_mask, ctx->ray_samples, ctx->max_light_samples, 1, cube, occ, s, cnt));   // Fluid.cpp:880-908
	} else {
		ScopedMark mk(ctx, s, MK_VIEW);
		occ = render_occupancy(ctx, color, s);
		FX_HIP(launch_raymarch_view(ctx->g, ctx->half, color, nullptr, ctx->fc, ctx->has_sh ? ctx->sh_dev : nullptr,
			size, ctx->visibility_mask, ctx->ray_samples, ctx->max_light_samples, 0, cube, occ, s, cnt));   // Fluid.cpp:825-855
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

int fx_synchronize(fx_ctx* ctx)
{
	if (!ctx) return FX_E_INVALID;
	std::vector<fx_ctx*> M;
	for_members(ctx, M);
	int rc = FX_OK;
	for (fx_ctx* c : M) {
		DeviceGuard dg(c->device);
		if (hipDeviceSynchronize() != hipSuccess) return FX_E_DEVICE;
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

int fx_upload(fx_ctx* ctx, int field, const void* host, size_t bytes)
{
	if (!ctx || !host) return FX_E_INVALID;
	size_t need = 0;
	int rc = field_info(ctx, field, &need);
	if (rc) return rc;
	if (bytes != need) return FX_E_INVALID;
	DeviceGuard dg(ctx->device);
	FX_HIP(hipDeviceSynchronize());
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
	size_t fo[6], mo[16];
	uint32_t sz = 0, nm = 0;
	if (!dds_bc6h_cube_layout(dds, bytes, &sz, &nm, fo, mo)) return FX_E_INVALID;
	if (size) *size = sz;
	if (mips) *mips = nm;
	return FX_OK;
}

int fx_dds_decode_cube(fx_ctx* ctx, const void* dds, size_t bytes, uint32_t mip, float* out_cube, size_t out_floats)
{
	if (!ctx || !out_cube) return FX_E_INVALID;
	size_t fo[6], mo[16];
	uint32_t sz = 0, nm = 0;
	if (!dds_bc6h_cube_layout(dds, bytes, &sz, &nm, fo, mo) || mip >= nm) return FX_E_INVALID;
	const uint32_t n = std::max<uint32_t>(sz >> mip, 1), nb = (n + 3) / 4;
	const size_t face_floats = (size_t)n * n * 3, face_blocks = (size_t)nb * nb * 16;
	if (out_floats != 6 * face_floats) return FX_E_INVALID;
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

// side stream (highest priority: the exchange kernels must get CUs while the interior sweeps fill the chip) + ordering events
static int make_comm_stream(fx_comm_group* g, int device)
{
	g->comm_stream = nullptr; g->ev_ready = nullptr; g->ev_done = nullptr;
	g->shared_stream = nullptr; g->broken = false;
	g->face_stream = nullptr; g->ev_int = nullptr; g->ev_face1 = nullptr; g->ev_col_ready = nullptr; g->ev_col_done = nullptr;
	DeviceGuard dg(device);
	// DEFAULT priority.  A high-priority side stream made every dependency between it and the compute stream cost about a
	// millisecond for the first group of a process (18 ms per step instead of 7.7, profiles/r01d_slab_schedule_loopback.txt);
	// FLUIDX_COMM_PRIORITY=1 asks for the highest priority anyway (measurement knob).
	int lo = 0, hi = 0, prio = 0;
	const char* pe = std::getenv("FLUIDX_COMM_PRIORITY");
	if (pe && pe[0] == '1' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) prio = hi;
	if (hipStreamCreateWithPriority(&g->comm_stream, hipStreamNonBlocking, prio) != hipSuccess) return FX_E_DEVICE;
	if (hipEventCreateWithFlags(&g->ev_ready, hipEventDisableTiming) != hipSuccess) return FX_E_DEVICE;
	if (hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming) != hipSuccess) return FX_E_DEVICE;
	if (hipStreamCreateWithFlags(&g->face_stream, hipStreamNonBlocking) != hipSuccess) return FX_E_DEVICE;
	if (hipEventCreateWithFlags(&g->ev_int, hipEventDisableTiming) != hipSuccess) return FX_E_DEVICE;
	if (hipEventCreateWithFlags(&g->ev_face1, hipEventDisableTiming) != hipSuccess) return FX_E_DEVICE;
	if (hipEventCreateWithFlags(&g->ev_col_ready, hipEventDisableTiming) != hipSuccess) return FX_E_DEVICE;
	if (hipEventCreateWithFlags(&g->ev_col_done, hipEventDisableTiming) != hipSuccess) return FX_E_DEVICE;
	return FX_OK;
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
	if ((rc = make_comm_stream(g, ctx->device))) { delete t; delete g; return rc; }
	if ((rc = t->min_over_ranks(ctx->g.nz, ctx->stream, &g->min_nz))) { delete t; delete g; return rc; }
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
		if ((rc = t->min_over_ranks(digest, ctx->stream, &lo)) || (rc = t->min_over_ranks(-digest, ctx->stream, &hi))) { delete t; delete g; return rc; }
		if (lo != digest || -hi != digest) {
			ctx->last_error = "fx_comm_init_rank: the ranks were created with different descriptors";
			std::fprintf(stderr, "fluidx: %s\n", ctx->last_error.c_str());
			delete t; delete g;
			return FX_E_INVALID;
		}
	}
	if (!ctx->step_rec && (rc = make_step_record(ctx, nranks, true))) { delete t; delete g; return rc; }   // (kept from a refused earlier attempt)
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
			delete t; delete g;
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
	}
	DeviceGuard dg(ctx->device);
	hipStream_t s = pick_stream(ctx, stream);
	if (ctx->halo_fault) return FX_E_HALO;             // a colour field that is known to be off is not gathered into a picture
	ScopedMark mk(ctx, s, MK_EXCH);
	int rc = ctx->group->transport->gather(ctx->group, parts, root, s);
	if (rc) return rc;
	if (am_root && full->stream != s) {                // the render context's own stream must see the planes
		FX_HIP(hipEventRecord(ctx->group->ev_done, s));
		FX_HIP(hipStreamWaitEvent(full->stream, ctx->group->ev_done, 0));
	}
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

int fx_comm_init_local(fx_ctx** ctxs, int nranks)
{
	if (!ctxs || nranks < 1) return FX_E_INVALID;
	int zexp = 0;
	for (int r = 0; r < nranks; ++r) {
		if (!ctxs[r]) return FX_E_INVALID;
		int rc = check_slab_chain(ctxs[r], r, nranks);
		if (rc) return rc;
		if (ctxs[r]->g.z0 != zexp || ctxs[r]->device != ctxs[0]->device) return FX_E_INVALID;   // contiguous chain, one device
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
	for (int r = 0; r < nranks; ++r)
		if (!ctxs[r]->step_rec) { if (int rc = make_step_record(ctxs[r], 1, false)) return rc; }
	fx_comm_group* g = new fx_comm_group();
	g->transport = make_local_transport();
	g->refs = nranks;
	if (int rc = make_comm_stream(g, ctxs[0]->device)) { delete g->transport; delete g; return rc; }
	g->min_nz = ctxs[0]->g.nz;
	for (int r = 1; r < nranks; ++r) g->min_nz = std::min(g->min_nz, ctxs[r]->g.nz);
	for (int r = 0; r < nranks; ++r) {
		g->members.push_back(ctxs[r]);
		ctxs[r]->group = g; ctxs[r]->rank = r; ctxs[r]->nranks = nranks;
		// one stream for the whole loop-back group: phases of different members are ordered by it.  The group owns it, so
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

}  // extern "C"

