// fx_schedule.cpp -- Fluid::Simulate (/root/reference/FluidX12/Content/Fluid.cpp:348-410) as a schedule of kernel launches and halo
// exchanges over a group of z-slab contexts: one code path serves the single-GPU case, the RCCL slabs and the in-process loop-back
// slabs.  The C ABI entry points that drive it are in fx_api.cpp; contexts, fields and options in fx_context.cpp.
#include "fx_host.h"
#include <memory>

using namespace fx;

namespace fxh {

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


// ---- lanes (fx_context.h): where a member's work is enqueued.  One lane serves a whole shared-stream loop-back group and an RCCL rank;
// a peer group has a lane per member, each with the member's own compute stream.
static inline bool peer_group(const fx_ctx* c) { return c->group && c->group->per_member; }
// the compute stream of member m in a step driven on stream s
static inline hipStream_t CS(const fx_ctx* m, hipStream_t s) { return peer_group(m) ? m->group->lane_of(m).compute : s; }
struct LaneRef { fx_lane* lane; hipStream_t compute; };
static std::vector<LaneRef> lanes_of(fx_ctx* lead, const std::vector<fx_ctx*>& M, hipStream_t s)
{
	std::vector<LaneRef> out;
	if (!lead->group) return out;
	if (lead->group->per_member) for (fx_ctx* m : M) out.push_back(LaneRef{ &lead->group->lane_of(m), lead->group->lane_of(m).compute });
	else out.push_back(LaneRef{ &lead->group->lanes[0], s });
	return out;
}
enum StreamKind { ON_COMPUTE = 0, ON_COMM = 1 };

// 0 = no side stream, 1 = advection halo overlapped, 2 = pressure rounds overlapped as well, 3 = and the colour half of the
// next step's advection halo travels behind this step's pressure phase (fx_set_option)
int overlap_level(const fx_ctx* lead)
{
	if (!multi_rank(lead) || lead->group->lanes.empty() || !lead->group->lanes[0].comm) return 0;
	return lead->opt_overlap;
}

struct ExchSpec { int set, k, pidx; };

// one exchange of the chain, ordered on every lane's compute or comm stream
static int do_exchange(fx_ctx* ctx, const std::vector<fx_ctx*>& M, const ExchSpec* specs, int nspec, StreamKind kind, hipStream_t s, int channel = 0)
{
	if (!multi_rank(ctx)) return FX_OK;
	DeviceGuard dg(ctx->device);
	std::vector<hipStream_t> streams;
	for (const LaneRef& L : lanes_of(ctx, M, s)) streams.push_back(kind == ON_COMM ? L.lane->comm : L.compute);
	ScopedMark mk(ctx, peer_group(ctx) ? streams[(size_t)ctx->rank] : streams[0], MK_EXCH);
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
	return ctx->group->transport->exchange(ctx->group, segs, streams, channel);
}

// on every lane: `to` picks up after everything queued on `from` so far
#define FX_LANES(body) do { for (const LaneRef& L : lanes_of(ctx, M, s)) { DeviceGuard dgl_(L.lane->device); body } } while (0)
// comm stream picks up after everything queued on the compute stream so far
static int comm_fork(fx_ctx* ctx, const std::vector<fx_ctx*>& M, hipStream_t s)
{
	FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_ready, L.compute)); FX_HIP(hipStreamWaitEvent(L.lane->comm, L.lane->ev_ready, 0)););
	return FX_OK;
}
static int comm_mark_done(fx_ctx* ctx, const std::vector<fx_ctx*>& M, hipStream_t s) { FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_done, L.lane->comm));); return FX_OK; }
static int comm_join(fx_ctx* ctx, const std::vector<fx_ctx*>& M, hipStream_t s) { FX_LANES(FX_HIP(hipStreamWaitEvent(L.compute, L.lane->ev_done, 0));); return FX_OK; }


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
	// the context lends the staged kernel scratch to put far-tracing voxels aside (allocated by fx_create for its owned planes -- the lazy
	// branch below only serves contexts made before that; every advection of a context runs on its compute stream, one after the other,
	// so one scratch serves all its ranges)
	if (!ctx->adv_far && !ctx->adv_far_tried) {
		ctx->adv_far_tried = true;
		const size_t words = advect_far_words(g, g.nz);
		if (words) {
			if (hipMalloc((void**)&ctx->adv_far, words * sizeof(uint32_t)) == hipSuccess) {
				FX_HIP(hipMemsetAsync(ctx->adv_far, 0, 2 * sizeof(uint32_t), s));   // the two alternating totals
				ctx->adv_far_words = words;
			} else {                                         // no room for the scratch: the staged kernel gathers far-tracing voxels itself
				(void)hipGetLastError();
				ctx->adv_far = nullptr;
			}
		}
	}
	const bool lend = ctx->adv_far != nullptr;
	bool far_used = false;
	// the render's alpha-only side volume (fx_render_accel.hip) is written by the launch that makes the colour field, where that is one
	// staged launch over the whole grid: the render's build pass then reads 4 bytes per voxel instead of the whole texel
	const bool frame_was_rendered = ctx->rendered_since_step && ctx->rendered_on == s;
	ctx->rendered_since_step = false;
	AdvectAlpha aa{ frame_was_rendered && ctx->accel_ok && ctx->opt_render_accel && FX_KNOB_INT("ADVECT_ALPHA", 1) ? ctx->accel.alpha : nullptr, false };
	if (ctx->accel_alpha_of == ctx->col[par]) ctx->accel_alpha_of = nullptr;          // that buffer is about to change
	FX_HIP(launch_advect(g, sp, ctx->half, ctx->vel[0], ctx->col[1 - par], ctx->vel[1], ctx->col[par],
		r.lo, r.hi, ctx->halo_overflow, s, lend ? ctx->adv_far : nullptr, lend ? ctx->adv_far_words : 0, (int)(ctx->adv_far_turn & 1u), &far_used, &aa));
	if (aa.written) ctx->accel_alpha_of = ctx->col[par];
	if (far_used) ++ctx->adv_far_turn;
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

static int record_step(fx_ctx* ctx, std::vector<fx_ctx*>& M, hipStream_t s)
{
	if (!multi_rank(ctx) || !ctx->step_rec) return FX_OK;
	for (fx_ctx* m : M) {
		if (m->rec_in_project) { m->rec_in_project = false; continue; }       // k_project_v4 has already written it
		DeviceGuard dg(m->device);
		FX_HIP(launch_face_need(m->g, m->half, m->vel[0], m->time_step, (int)m->desc.advect_address, options_digest(m), m->halo_overflow, m->step_rec, CS(m, s)));
	}
	DeviceGuard dg(ctx->device);
	if (ctx->group->transport->is_local()) {
		for (fx_ctx* m : M) {                                                  // every member's record on its own stream, behind its own event
			DeviceGuard dgm(m->device);
			FX_HIP(hipMemcpyAsync(m->rec_host, m->step_rec, 4 * sizeof(int), hipMemcpyDeviceToHost, CS(m, s)));
			FX_HIP(hipEventRecord(m->rec_ev, CS(m, s)));
		}
	} else {
		// off the compute stream when there is a side stream: the next step's interior advection need not wait for the gather
		hipStream_t cs = s;
		if (overlap_level(ctx) >= 1) { int rc = comm_fork(ctx, M, s); if (rc) return rc; cs = ctx->group->lanes[0].comm; }
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
static int consume_record(fx_ctx* ctx, std::vector<fx_ctx*>& M)
{
	const int Ha = (int)ctx->desc.halo_advect;
	for (fx_ctx* m : M) { m->adv_w_lo = has_lower(m) ? Ha : 0; m->adv_w_hi = has_upper(m) ? Ha : 0; }
	if (!multi_rank(ctx) || !ctx->rec_pending) return FX_OK;
	const bool local = ctx->group->transport->is_local();
	for (fx_ctx* m : M) {                                  // (in-process groups: every member's record; an RCCL rank: the gathered one)
		DeviceGuard dg(m->device);
		FX_HIP(hipEventSynchronize(m->rec_ev));
		if (!local) break;
	}
	const int n = ctx->nranks;
	auto rec = [&](int r) -> const int* { return local ? M[(size_t)r]->rec_host : ctx->rec_host + 4 * r; };
	bool fault = false, mismatch = false, usable = ctx->opt_adaptive != 0;
	for (int r = 0; r < n; ++r) { fault = fault || rec(r)[3] != 0; mismatch = mismatch || rec(r)[2] != rec(0)[2]; }
	for (fx_ctx* m : M) { usable = usable && m->need_valid && m->time_step <= m->rec_dt; m->rec_pending = false; }
	if (fault) {
		// This return IS the chain-wide notice of the fault: every rank gets it once, from the same gathered record, and the step
		// after it starts clean -- on every rank, whether or not anybody calls fx_synchronize in between: a rank whose own flag is
		// still up takes it down here (nothing of the previous step is in flight any more: its record has arrived) and remembers the
		// fault on the host instead, so that its read-back and checkpoints keep refusing until fx_synchronize acknowledges it.
		// Left up, the flag would be copied into the record of every later step and the chain would alternate between one executed and
		// one refused step for ever.
		for (fx_ctx* m : M) {
			if (!rec(m->rank)[3] || !m->halo_overflow) continue;
			DeviceGuard dg(m->device);
			FX_HIP(hipDeviceSynchronize());                            // (the overlapped schedule has enqueued this step's interior advection already: it may raise the flag again)
			unsigned flag = 0;
			FX_HIP(hipMemcpy(&flag, m->halo_overflow, sizeof flag, hipMemcpyDeviceToHost));
			if (!flag) continue;                                       // acknowledged by an fx_synchronize since
			FX_HIP(hipMemset(m->halo_overflow, 0, sizeof(unsigned)));
			FX_HIP(hipDeviceSynchronize());                            // the context's streams do not order against the NULL stream
			m->halo_fault = true;
		}
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
		if (col_ready) FX_LANES(FX_HIP(hipStreamWaitEvent(L.compute, L.lane->ev_col_done, 0)););
		const ExchSpec spec{ col_ready ? EX_ADVECT_VEL : EX_ADVECT_IN, Ha, 0 };
		if ((rc = do_exchange(ctx, M, &spec, 1, ON_COMPUTE, s))) return rc;
		for (fx_ctx* m : M) {
			ScopedMark mk(m, CS(m, s), MK_ADVECT);
			if (m->timing_on) m->acc.advect_halo_planes += (uint64_t)(m->adv_w_lo + m->adv_w_hi);
			if ((rc = advect_range(m, CS(m, s), owned(m), false))) return rc;
		}
		return FX_OK;
	}
	// the interior first (it reads owned planes only, whatever the exchange will carry): the device is busy while the host waits
	// for the previous step's record, which sizes the exchange
	if ((rc = comm_fork(ctx, M, s))) return rc;                // the comm stream picks up behind the previous step
	for (fx_ctx* m : M) {
		ScopedMark mk(m, CS(m, s), MK_ADVECT);
		const Range o = owned(m);
		if ((rc = advect_range(m, CS(m, s), Range{ o.lo + (has_lower(m) ? Ha : 0), o.hi - (has_upper(m) ? Ha : 0) }, true))) return rc;
	}
	if ((rc = consume_record(ctx, M))) return rc;
	if (col_ready) FX_LANES(FX_HIP(hipStreamWaitEvent(L.lane->comm, L.lane->ev_col_done, 0)););
	const ExchSpec spec{ col_ready ? EX_ADVECT_VEL : EX_ADVECT_IN, Ha, 0 };
	if ((rc = do_exchange(ctx, M, &spec, 1, ON_COMM, s))) return rc;
	if ((rc = comm_mark_done(ctx, M, s))) return rc;
	if ((rc = comm_join(ctx, M, s))) return rc;
	for (fx_ctx* m : M) {
		ScopedMark mk(m, CS(m, s), MK_ADVECT);
		if (m->timing_on) m->acc.advect_halo_planes += (uint64_t)(m->adv_w_lo + m->adv_w_hi);
		const Range o = owned(m);
		if (has_lower(m) && (rc = advect_range(m, CS(m, s), Range{ o.lo, o.lo + Ha }, false))) return rc;
		if (has_upper(m) && (rc = advect_range(m, CS(m, s), Range{ o.hi - Ha, o.hi }, false))) return rc;
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
// 2-D grids relax on LDS tiles (fx_jacobi2d.hip) unless the caller asked for one sweep per launch (jacobi_fuse = 1: the plainest kernels,
// what the kernel-against-kernel parity tests compare with)
static bool takes_2d_tiles(const fx_ctx* c) { return jacobi2d_max_sweeps(c->g) > 0 && (c->desc.flags & FX_FLAG_JACOBI_FUSE_MASK) != 1 && (!c->frozen || c->frozen_alt); }

static int jacobi_launch(fx_ctx* ctx, hipStream_t s, int src, int t, Range r, ScopedMark* mk)
{
	r.lo = std::max(r.lo, 0); r.hi = std::min(r.hi, ctx->g.Zg);
	if (r.hi <= r.lo) return FX_OK;
	DeviceGuard dg(ctx->device);
	if (takes_2d_tiles(ctx)) {                             // 2-D grids: up to eight sweeps per launch on LDS tiles, freeze bytes included
		FX_HIP(launch_jacobi2d(ctx->g, ctx->p[src], ctx->b, ctx->p[src ^ 1], ctx->frozen, ctx->frozen ? ctx->frozen_alt : nullptr, t, s));
		if (ctx->frozen) std::swap(ctx->frozen, ctx->frozen_alt);       // the mask ping-pongs with the pressure
	} else if (t > 1) {
		FX_HIP(launch_jacobi_fused(ctx->g, ctx->p[src], ctx->b, ctx->p[src ^ 1], t, r.lo, r.hi, s));
	} else {
		FX_HIP(launch_jacobi_sweep(ctx->g, ctx->p[src], ctx->b, ctx->p[src ^ 1], ctx->frozen, r.lo, r.hi, s));
	}
	if (mk) { mk->launches += 1; mk->sweeps += t; }
	return FX_OK;
}

// a launch of `t` sweeps the geometry has a kernel for: threes exist for X = 256 / 512 only, twos wherever a fused kernel serves the rows
static int legal_sweeps(const fx_ctx* c, int t)
{
	if (c->frozen || takes_2d_tiles(c)) return t;
	if (t == 3 && !jacobi_strip3_supported(c->g)) t = 2;
	if (t == 2 && jacobi_fused_max_sweeps(c->g, 2, c->g.nz) < 2) t = 1;
	return t;
}

static int fused_sweeps(const fx_ctx* c)
{
	if (takes_2d_tiles(c)) return jacobi2d_max_sweeps(c->g);
	return c->frozen ? 1 : jacobi_fused_max_sweeps(c->g, (int)(c->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), c->g.nz);
}

// `count` lock-step sweeps whose first one may read `count` exchanged halo planes; the planes swept shrink by
// one per sweep towards the owned range (redundant halo work instead of an exchange per sweep)
static int jacobi_round(fx_ctx* ctx, hipStream_t s, int count, ScopedMark* mk)
{
	int done = 0;
	while (done < count) {
		const int left = count - done;
		int t = std::min(left, fused_sweeps(ctx));
		if (!ctx->frozen && !takes_2d_tiles(ctx) && jacobi_prefers_three(ctx->g, (int)(ctx->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), ctx->g.nz))
			t = left == 4 ? 2 : std::min(left, 3);           // threes, and a remainder of 4 as 2 + 2 rather than 3 + 1
		if (!ctx->frozen && !takes_2d_tiles(ctx) && jacobi_prefers_four(ctx->g, (int)(ctx->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), ctx->g.nz))
			t = jacobi_strip3_supported(ctx->g) ? (left >= 7 || left == 4 ? 4 : std::min(left, 3))   // fours; a remainder of 5 / 6 as 3 + 2 / 3 + 3
				: std::min(left, 4);                                    // (x tiles: no threes -- 4 + 2, 4 + 1)
		t = legal_sweeps(ctx, t);
		if (mk && mk->kind == MK_JACOBI && mk->launches && t * mk->launches < mk->sweeps) mk->split(MK_JACOBI_TAIL);   // shorter launches from here on
		const int rc = jacobi_launch(ctx, s, ctx->p_cur, t, grown(ctx, multi_rank(ctx) ? left - t : 0), mk);
		if (rc) return rc;
		ctx->p_cur ^= 1;
		done += t;
	}
	return FX_OK;
}

static int clear_freeze_masks(std::vector<fx_ctx*>& M, hipStream_t s)
{
	for (fx_ctx* m : M)
		if (m->frozen) { DeviceGuard dg(m->device); if (hipMemsetAsync(m->frozen, 0, m->g.cells_local(), s) != hipSuccess) return FX_E_DEVICE; }
	return FX_OK;
}

// FX_JACOBI_FAITHFUL: the sparse solver of fx_jacobi_freeze.hip.  Level 1 everywhere (into p[other] AND p_aux; the input buffer
// becomes the spare), then ceil((iters - 1) / T) launches over the tiles that still relax, all enqueued; the result is in the last
// launch's output buffer (settled tiles agree in both).  Bit-identical to `iters` generic sweeps with the byte mask
// (tests/test_gpu_freeze.py).  A slab rank (round 4) runs it on the view of the planes it holds (jacobi_freeze_view): the dense sweep
// over its owned planes, the tile cones reaching into the halo, and kFreezeHalo planes of pressure + mask travelling to the
// neighbours behind every launch -- every rank enqueues the same launches and exchanges whatever its tiles do.
// (a level fits the stat word's low byte, every launch has its counters)
static const int kFreezeHalo = 4;                          // = the most levels a tile launch takes (jacobi_freeze_levels_per_launch)
static bool takes_sparse_solver(const fx_ctx* c, uint32_t iters)
{
	if (!(c->frozen && c->fz_tile_next && jacobi_freeze_supported(c->g) && iters <= 255 && (int)iters / jacobi_freeze_levels_per_launch() + 3 < kFreezeSlots)) return false;
	if (!multi_rank(c)) return true;
	// chain-wide facts only (every rank must take the same branch): the grid, the halo, the thinnest slab
	const uint64_t planes = (uint64_t)c->g.Zg + 2 * (uint64_t)c->g.H;
	return c->g.H >= kFreezeHalo && c->group->min_nz >= kFreezeHalo && (uint64_t)c->g.X * c->g.Y * planes < (1u << 30);
}

static int jacobi_freeze(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters)
{
	fx_ctx* ctx = lead;                                                 // FX_HIP reports through `ctx`
	const bool multi = multi_rank(lead);
	struct Run { Geom v; size_t off; int own0; FreezeWork w; float *src, *a, *d; uint8_t *ma, *md, *mx; uint32_t* stat; uint32_t stat_hi; };
	std::vector<Run> R(M.size());
	std::vector<std::unique_ptr<ScopedMark>> mk(M.size());
	const bool fuse = lead->fz_fuse_div;
	lead->fz_fuse_div = false;
	int strip_want = 0;                                                 // levels wanted from the masked strip pipelines (single domain, X = 256)
	bool strip_four = false, count_marks_now = false;
	for (size_t i = 0; i < M.size(); ++i) {
		fx_ctx* m = M[i];
		DeviceGuard dg(m->device);
		hipStream_t ms = CS(m, s);
		if (++m->fz_gen >= (1u << 23)) {                                // the tag (gen << 8 | level) of the stat words stays below 2^32: start over
			FX_HIP(hipMemsetAsync(m->fz_tile_next, 0, (size_t)jacobi_freeze_tiles(m->g) * sizeof(uint32_t), ms));
			FX_HIP(hipMemsetAsync(m->fz_stat, 0, kFreezeStatRing * sizeof(uint32_t), ms));
			FX_HIP(hipMemsetAsync(m->fz_counts, 0, 2 * jacobi_freeze_count_words() * sizeof(uint32_t), ms));
			m->fz_gen = 2; m->fz_gen_mark = 0;
		}
		Run& r = R[i];
		int first = 0;
		r.v = jacobi_freeze_view(m->g, &first, &r.own0);
		r.off = (size_t)first * m->g.plane();
		const uint32_t gen = m->fz_gen;
		r.stat_hi = gen << 8;
		r.stat = m->fz_stat + gen % kFreezeStatRing;
		m->fz_iters[gen % kFreezeStatRing] = iters;
		const size_t cw = jacobi_freeze_count_words();
		r.w = FreezeWork{ m->fz_tile_next, gen, { m->fz_list[0], m->fz_list[1] }, jacobi_freeze_tiles(m->g),
			m->fz_counts + (gen & 1u) * cw, m->fz_counts + ((gen & 1u) ^ 1u) * cw };
		r.src = m->p[m->p_cur]; r.a = m->p[m->p_cur ^ 1]; r.d = m->p_aux;
		r.ma = m->fz_mask[0]; r.md = m->fz_mask[1]; r.mx = m->fz_mask[2];
		// (The dense sweep writes level 1 to BOTH buffers the tile launches alternate between.  Writing one and letting the first tile
		// launch carry the unlisted tiles' border cells across was built and measured level: the dense sweep 75 -> 46 us at 256^3, the
		// first tile launch slower by as much -- the shell of a 4-deep cone around ~3000 listed tiles is more bytes than the second copy.)
		// whole steps (simulate_impl) leave the divergence to this launch: it computes b from the advected velocity and stores it for the
		// tile launches, instead of reading it back from a launch of its own
		const size_t moff = (size_t)first * (size_t)((m->g.X + 3) / 4) * m->g.Y, es = elem_size(m);
		// fx_timing books the dense sweep as the "main" launch and the tile launches beside it.  A single domain keeps ONE mark open over
		// all of them (an event record between two launches is a 2-3 us gap, 17 of them a solve); slab ranks close it at every exchange
		// While most tiles still relax the tile launches are dense sweeps in all but name: the levels right behind the dense sweep then go
		// through a masked strip pipeline (see below), decided HERE because the dense sweep itself depends on it -- a strip launch reads one
		// copy of level 1 and writes both buffers, so the dense sweep in front of it leaves its second copy (and second mask copy) away.
		// FREEZE_DENSE_LEVELS: how many levels (0, 3, 4, 6, 7, 8 ...); default -1 = by what the dense sweeps of the last steps left relaxing
		// -- the count k_count_marks put into a host-visible word behind an earlier solve's dense sweep (thresholds below).  Young plumes
		// stay on the tile launches, where a masked launch would cost 0.08 ms for nothing.
		if (!multi && jacobi_freeze_strip_supported(m->g) && m->fz_mask[2]) {
			strip_want = FX_KNOB_INT("FREEZE_DENSE_LEVELS", -1);
			strip_four = FX_KNOB_INT("FREEZE_STRIP4", 1) && jacobi_freeze_strip4_supported(m->g);
			if (strip_want < 0) {
				strip_want = 0;
				if (m->fz_active_dev) {
					// every fourth solve counts (the plume changes slowly, the count is a 5-us launch); the solve TWO after it takes the count over,
					// behind the event recorded with it -- by then it has long arrived, so nothing waits, and which solve switches is a function
					// of the step sequence, not of when the host happened to look (the unsynchronised read made launch sequences,
					// step times and kernel statistics vary from run to run around the threshold)
					if ((gen & 3u) == 2u && m->fz_active_pending) {
						FX_HIP(hipEventSynchronize(m->fz_active_ev));
						m->fz_active_pending = false;
						const uint32_t active = *(volatile uint32_t*)m->fz_active_host, tiles = (uint32_t)jacobi_freeze_tiles(m->g);
						// how many strip launches, with hysteresis.  Four levels per launch (k_freeze_strip4o behind a one-copy dense sweep): one
						// launch pays from a fifth of the tiles (256^3 plume: frame ~70, 0.600 -> 0.585 ms per step; at a half, frame 110: 0.747 ->
						// 0.671), a second one from two thirds (frame ~140; frame 190: 0.790 -> 0.759).  Three levels per launch
						// (k_freeze_strip3, 30 us per level): one launch from half of the tiles.
						int n = m->fz_dense_n;
						if (strip_four) {
							if (n == 0 && 5u * active >= tiles) n = 1;
							else if (n >= 1 && 6u * active < tiles) n = 0;
							if (n == 1 && 16u * active >= 11u * tiles) n = 2;
							else if (n == 2 && 8u * active < 5u * tiles) n = 1;
						} else {
							if (n == 0 && 2u * active >= tiles) n = 1;
							else if (n >= 1 && 5u * active < 2u * tiles) n = 0;
							if (n > 1) n = 1;
						}
						m->fz_dense_n = n;
					}
					strip_want = m->fz_dense_n * (strip_four ? 4 : 3);
					count_marks_now = (gen & 3u) == 0u;
				}
			}
		}
		const bool one_copy = strip_want >= 3 && iters - 1 > 3 && FX_KNOB_INT("FREEZE_DENSE_ONE", 1);   // a strip launch follows (the loop below)
		mk[i].reset(new ScopedMark(m, ms, MK_JACOBI));
		FX_HIP(launch_freeze_dense(r.v, r.src + r.off, m->b + r.off, r.a + r.off, one_copy ? nullptr : r.d + r.off, r.ma + moff, one_copy ? nullptr : r.md + moff, r.w, ms,
			fuse ? (const char*)m->vel[1] + r.off * es : nullptr, m->half, r.own0, m->g.nz, m->g.cells_local()));
		mk[i]->launches = 1; mk[i]->sweeps = 1;
		if (multi || iters == 1) mk[i].reset(); else mk[i]->split(MK_JACOBI_TAIL);
	}
	auto exchange = [&](bool with_b) -> int {
		if (!multi) return FX_OK;
		for (size_t i = 0; i < M.size(); ++i) { M[i]->fz_x_p = R[i].a; M[i]->fz_x_m = R[i].ma; }
		const ExchSpec specs[2] = { { EX_FREEZE, kFreezeHalo, 0 }, { EX_DIV, kFreezeHalo, 0 } };
		return do_exchange(lead, M, specs, with_b ? 2 : 1, ON_COMPUTE, s);
	};
	int rc = exchange(true);                                            // level 1 and the divergence across the faces
	if (rc) return rc;
	const int T = jacobi_freeze_levels_per_launch();
	int level = 1, n = 0;
	uint32_t left = iters - 1, flag_tag = 0;
	// While most tiles still relax the tile launches are dense sweeps in all but name (43 us per level at 256^3, a cone five times the
	// core per tile): the levels right behind the dense sweep go through a masked strip pipeline instead -- k_freeze_strip4o
	// (fx_jacobi_strip4.hip: four levels per launch for every cell, 22 us per level) or k_freeze_strip3 (fx_jacobi_stripm.hip: three, 30 us
	// per level) -- which leaves its last level in two of the three pressure buffers -- they rotate -- and the tile marks for the first
	// tile launch.  Single domain, X = 256; how many levels was decided in front of the dense sweep.
	if (!multi && jacobi_freeze_strip_supported(M[0]->g) && M[0]->fz_mask[2]) {
		fx_ctx* m = M[0];
		Run& r = R[0];
		int want = strip_want;
		const bool strip4 = strip_four;
		if (count_marks_now) {
			FX_HIP(launch_count_marks(r.w.tile_mark, r.w.gen, jacobi_freeze_tiles(m->g), m->fz_active_dev, CS(m, s)));
			if (!m->fz_active_ev) FX_HIP(hipEventCreateWithFlags(&m->fz_active_ev, hipEventDisableTiming));
			FX_HIP(hipEventRecord(m->fz_active_ev, CS(m, s)));
			m->fz_active_pending = true;
		}
		// a launch takes FOUR levels where the octet serves the grid (k_freeze_strip4o, fx_jacobi_strip4.hip; FREEZE_STRIP4=0: never), else three
		const bool four = strip4;
		for (int k = 0; want >= 3 && left > 3; ++k) {
			DeviceGuard dg(m->device);
			const int lv = four && want >= 4 && left > 4 ? 4 : 3;
			flag_tag = r.w.gen | ((uint32_t)(k + 1) << 24);
			if (lv == 4) { FX_HIP(launch_freeze_strip4(r.v, r.a, m->b, r.d, r.src, r.ma, r.md, r.mx, r.w.tile_mark, flag_tag, r.stat, r.stat_hi, level, CS(m, s))); }
			else { FX_HIP(launch_freeze_strip3(r.v, r.a, m->b, r.d, r.src, r.ma, r.md, r.mx, r.w.tile_mark, flag_tag, r.stat, r.stat_hi, level, CS(m, s))); }
			if (mk[0]) { mk[0]->launches += 1; mk[0]->sweeps += (uint64_t)lv; }
			m->acc.freeze_strip_launches += 1;                                  // (counted like the solves: with or without the timing marks)
			float* na = r.d; r.d = r.src; r.src = r.a; r.a = na;                  // level + lv now sits in (a, d); the buffer it was read from is the spare
			uint8_t* nm = r.md; r.md = r.mx; r.mx = r.ma; r.ma = nm;
			level += lv; left -= (uint32_t)lv; want -= lv;
		}
	}
	for (; left > 0; ++n) {
		const int t = (int)std::min<uint32_t>((uint32_t)T, left);
		for (size_t i = 0; i < M.size(); ++i) {
			fx_ctx* m = M[i];
			Run& r = R[i];
			DeviceGuard dg(m->device);
			const size_t moff = r.off / m->g.plane() * (size_t)((m->g.X + 3) / 4) * m->g.Y;
			if (!mk[i]) mk[i].reset(new ScopedMark(m, CS(m, s), MK_JACOBI_TAIL));
			FX_HIP(launch_freeze_tiles(r.v, r.a + r.off, m->b + r.off, r.d + r.off, r.ma + moff, r.md + moff, r.w, n, t, level, r.stat, r.stat_hi, CS(m, s), r.own0, m->g.nz, n == 0 ? flag_tag : 0u));
			mk[i]->launches += 1; mk[i]->sweeps += (uint64_t)t;
			if (multi) mk[i].reset();
			std::swap(r.a, r.d); std::swap(r.ma, r.md);
		}
		left -= (uint32_t)t; level += t;
		if ((rc = exchange(false))) return rc;                          // the levels just made, kFreezeHalo planes deep (the last one serves the projection)
	}
	mk.clear();
	for (size_t i = 0; i < M.size(); ++i) { fx_ctx* m = M[i]; m->p[0] = R[i].a; m->p[1] = R[i].d; m->p_aux = R[i].src; m->fz_mask[0] = R[i].ma; m->fz_mask[1] = R[i].md; m->fz_mask[2] = R[i].mx; m->p_cur = 0; }
	return FX_OK;
}

// exchange, then k sweeps, exchange, ... on one stream
static int jacobi_serial(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters)
{
	const bool multi = multi_rank(lead);
	const int k = multi ? lead->opt_round : (int)iters;
	int rc;
	if (takes_sparse_solver(lead, iters)) return jacobi_freeze(lead, M, s, iters);
	for (fx_ctx* m : M) { std::vector<fx_ctx*> one{ m }; if ((rc = clear_freeze_masks(one, CS(m, s)))) return rc; }
	const ExchSpec bspec{ EX_DIV, k - 1, 0 };
	if ((rc = do_exchange(lead, M, &bspec, 1, ON_COMPUTE, s))) return rc;
	uint32_t done = 0;
	while (done < iters) {
		const int cnt = (int)std::min<uint32_t>(k, iters - done);
		const ExchSpec pspec{ EX_PRESSURE, cnt, lead->p_cur };
		if ((rc = do_exchange(lead, M, &pspec, 1, ON_COMPUTE, s))) return rc;
		for (fx_ctx* m : M) {
			ScopedMark mk(m, CS(m, s), MK_JACOBI);
			if ((rc = jacobi_round(m, CS(m, s), cnt, &mk))) return rc;
		}
		done += cnt;
	}
	const ExchSpec last{ EX_PRESSURE, 1, lead->p_cur };          // the projection's z-gradient reads one plane across the face
	return do_exchange(lead, M, &last, 1, ON_COMPUTE, s);
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
static int jacobi_overlapped(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters, int t, int k)
{
	fx_comm_group* grp = lead->group;
	int rc;
	fx_ctx* ctx = lead;                                    // FX_HIP / FX_LANES report through `ctx`
	// The face chain runs one single-sweep launch (k_jacobi_v4: 60 registers, both faces) per level.  Groups of fused sweeps with the
	// interior's register-strip kernels were built and measured slower (loop-back N = 4, 256^3 per rank, rounds of 9: 6.30 against 5.67 ms
	// per step; docs/LAB.md): on a 9..27-plane zone the strips have 64..192 waves whose 310 registers shut
	// the interior's waves out of their SIMDs for a whole 14-step pipeline -- removed in round 3.
	const ExchSpec first[2] = { { EX_DIV, k - 1, 0 }, { EX_PRESSURE, k, lead->p_cur } };
	if ((rc = do_exchange(lead, M, first, 2, ON_COMPUTE, s))) return rc;
	FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_int, L.compute)););
	bool in_flight = false;
	uint32_t done = 0;
	while (done < iters) {
		const int cnt = (int)std::min<uint32_t>(k, iters - done);
		// the interior's launches: the remainder first, then whole t's -- each one a launch the geometry has a kernel for (the same list on
		// every member: X and Y are the chain's)
		std::vector<int> parts;
		for (int left = cnt - (cnt / t) * t; left > 0;) { const int p = legal_sweeps(lead, left); parts.push_back(p); left -= p; }
		for (int j = 0; j < cnt / t; ++j) parts.push_back(t);
		const int m = (int)parts.size();
		// (every member from ITS OWN current buffer: members that ran serial rounds before -- the schedule is an option at run time -- may
		// hold their pressure in different buffers; SRC / FIN below are per member)
		const int flip = m & 1;
#define FX_SRC(c_) ((c_)->p_cur)
#define FX_FIN(c_) ((c_)->p_cur ^ flip)
		int fbuf = 0;                                      // which scratch buffer holds the chain's last level (set below)
		// ---- face stream: the chain (needs the previous round's interior + face copy, and its exchange)
		FX_LANES(FX_HIP(hipStreamWaitEvent(L.lane->face, L.lane->ev_int, 0)); if (in_flight) FX_HIP(hipStreamWaitEvent(L.lane->face, L.lane->ev_done, 0)););
		ScopedMark chain_mark(lead, grp->lane_of(lead).face, MK_CHAIN);         // one mark per chain (shared stream: all members' chains, booked on the first)
		{
			// the chain: one sweep per launch over the two thin face zones
			int grp_i = 0;
			for (int c = 1; c <= cnt; ++c, ++grp_i) {
				const int rem = cnt - c, ob = (grp_i + 1) & 1;
				for (fx_ctx* mctx : M) {
					if (!has_lower(mctx) && !has_upper(mctx)) continue;
					DeviceGuard dg(mctx->device);
					const Range o = owned(mctx);
					const float* in = grp_i == 0 ? mctx->p[FX_SRC(mctx)] : mctx->p_face[grp_i & 1];
					const Range lo{ o.lo - rem, has_lower(mctx) ? o.lo + k + rem : o.lo - rem };
					const Range hi{ has_upper(mctx) ? o.hi - k - rem : o.hi + rem, o.hi + rem };
					FX_HIP(launch_jacobi_sweep2(mctx->g, in, mctx->b, mctx->p_face[ob], nullptr, lo.lo, lo.hi, hi.lo, hi.hi, grp->lane_of(mctx).face));
				}
				if (grp_i == 0) FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_face1, L.lane->face)););
			}
			fbuf = grp_i & 1;                                  // the buffer the last group wrote
		}
		// ---- comm stream: the k final planes of the chain travel, the neighbour's land in the halo of p[fin]
		FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_ready, L.lane->face)); FX_HIP(hipStreamWaitEvent(L.lane->comm, L.lane->ev_ready, 0)););
		const ExchSpec pspec{ EX_PRESSURE_FACE, k, (fbuf << 1) | flip };      // (bit 0: the receiving buffer relative to the member's current one)
		if ((rc = do_exchange(lead, M, &pspec, 1, ON_COMM, s))) return rc;
		if ((rc = comm_mark_done(lead, M, s))) return rc;
		in_flight = true;
		// ---- compute stream: the interior
		for (fx_ctx* mctx : M) {
			ScopedMark mk(mctx, CS(mctx, s), MK_JACOBI);
			const Range o = owned(mctx);
			int lvl = 0, cur = FX_SRC(mctx);
			for (int j = 0; j < m; ++j) {
				const int tj = parts[j];
				lvl += tj;
				const int rem = cnt - lvl;
				const Range in{ has_lower(mctx) ? o.lo + k - rem : o.lo, has_upper(mctx) ? o.hi - k + rem : o.hi };
				if (j == 1 && (grp->per_member || mctx == M.front())) {      // launch 2 overwrites the chain's input
					DeviceGuard dg(mctx->device);
					FX_HIP(hipStreamWaitEvent(CS(mctx, s), grp->lane_of(mctx).ev_face1, 0));
				}
				if ((rc = jacobi_launch(mctx, CS(mctx, s), cur, tj, in, &mk))) return rc;
				cur ^= 1;
			}
		}
		// the chain's final planes complete the owned range of p[fin]
		FX_LANES(FX_HIP(hipStreamWaitEvent(L.compute, L.lane->ev_ready, 0)););
		for (fx_ctx* mctx : M) {
			DeviceGuard dg(mctx->device);
			const size_t pl = mctx->g.plane(), kb = (size_t)k * pl * 4;
			const Range o = owned(mctx);
			if (has_lower(mctx))
				FX_HIP(launch_copy_bytes(mctx->p[FX_FIN(mctx)] + (size_t)mctx->g.lz(o.lo) * pl, mctx->p_face[fbuf] + (size_t)mctx->g.lz(o.lo) * pl, kb, CS(mctx, s)));
			if (has_upper(mctx))
				FX_HIP(launch_copy_bytes(mctx->p[FX_FIN(mctx)] + (size_t)mctx->g.lz(o.hi - k) * pl, mctx->p_face[fbuf] + (size_t)mctx->g.lz(o.hi - k) * pl, kb, CS(mctx, s)));
			mctx->p_cur = FX_FIN(mctx);
		}
		FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_int, L.compute)););
		done += cnt;
	}
	if (in_flight) rc = comm_join(lead, M, s);
	return rc;
#undef FX_SRC
#undef FX_FIN
}

int jacobi_all(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters)
{
	if (overlap_level(lead) >= 2) {
		int t = fused_sweeps(lead);
		bool three = true, four = true;
		for (fx_ctx* m : M) {
			t = std::min(t, fused_sweeps(m));
			three = three && !m->frozen && jacobi_prefers_three(m->g, (int)(m->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), m->g.nz);
			four = four && !m->frozen && jacobi_prefers_four(m->g, (int)(m->desc.flags & FX_FLAG_JACOBI_FUSE_MASK), m->g.nz);
		}
		if (three) t = 3;                              // the interior launches of a round as threes (k = 9: 3 + 3 + 3); local choice, the exchanges do not depend on it
		if (four) t = 4;                               // ... as fours where the four-sweep kernel serves the slab (k = 9: 1 + 4 + 4)
		const int k = lead->opt_round;
		// two face zones (<= 2k - 1 planes each) and an interior; decided on the thinnest slab of the chain and on the (chain-wide)
		// Jacobi mode, so that every rank takes the same branch -- the two schedules exchange different things
		const bool ok = lead->group->lanes[0].face != nullptr && lead->group->min_nz >= 4 * k && lead->p_face[0] && !lead->frozen;
		if (ok) return jacobi_overlapped(lead, M, s, iters, t, k);
	}
	return jacobi_serial(lead, M, s, iters);
}


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
		FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_col_ready, L.compute)); FX_HIP(hipStreamWaitEvent(L.lane->comm, L.lane->ev_col_ready, 0)););
		const ExchSpec cs{ EX_COLOR_CUR, (int)ctx->desc.halo_advect, 0 };
		if ((rc = do_exchange(ctx, M, &cs, 1, ON_COMM, s, 1))) return rc;           // side channel: not queued with the step's own exchanges
		FX_LANES(FX_HIP(hipEventRecord(L.lane->ev_col_done, L.lane->comm)););
		for (fx_ctx* m : M) m->col_halo_buf = (int)m->frame_parity;
	}
	if (ctx->time_step > 0.0f) {                       // CSProject3D.hlsl:88
		const ExchSpec uz{ EX_UZ1, 1, 0 };
		if ((rc = do_exchange(ctx, M, &uz, 1, ON_COMPUTE, s))) return rc;
		if (takes_sparse_solver(ctx, ctx->desc.jacobi_iters) && jacobi_freeze_can_fuse_divergence(ctx->g)) ctx->fz_fuse_div = true;   // k_freeze_dense computes it
		else for (fx_ctx* m : M) if ((rc = divergence_phase(m, CS(m, s)))) return rc;
		rc = jacobi_all(ctx, M, s, ctx->desc.jacobi_iters);
		ctx->fz_fuse_div = false;                      // (consumed by the dense sweep; never left standing for a later stage call)
		if (rc) return rc;
		for (fx_ctx* m : M) if ((rc = project_phase(m, CS(m, s)))) return rc;
	} else {
		for (fx_ctx* m : M) {
			DeviceGuard dg(m->device);
			m->rec_in_project = false;
			if (launch_copy_velocity(m->g, m->half, m->vel[1], m->vel[0], CS(m, s)) != hipSuccess) return FX_E_DEVICE;
		}
	}
	if ((rc = record_step(ctx, M, s))) return rc;
	for (fx_ctx* m : M) { if (m->timing_on) m->acc.steps += 1; if (ctx->time_step > 0.0f) m->steps_simulated += 1; }
	return FX_OK;
}

}  // namespace fxh
