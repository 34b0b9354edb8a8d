// fx_host.h -- what the three host-side units of the C ABI share (fx_context.cpp: contexts, fields, timing, options;
// fx_schedule.cpp: the simulation step over a group of slab contexts; fx_api.cpp: frame constants, rendering, stage calls, slab
// groups): status macro, device guard, timing marks, plane ranges.  Product code: nothing from oracle/.
#pragma once
#include "fx_context.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

// A failed HIP call also leaves its code in the runtime's sticky "last error": the launch helpers end in hipGetLastError(), and a
// stale out-of-memory from one context's failed fx_create would otherwise surface as the status of the next, unrelated launch
// (found by the descriptor fuzz).  Reading the last error here clears it.
#define FX_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (void)hipGetLastError(); \
	ctx->last_error = std::string(#call) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? FX_E_NOMEM : FX_E_DEVICE; } } while (0)

namespace fxh {

const uint32_t kNumMips = 5;            // Fluid.cpp:229
const uint32_t kDefaultAdvectHalo = 6;  // measured z back-trace reach at 256^3: <= 3.5 cells over 400 steps (tools/reach_probe.py)
const uint32_t kFreezeStatRing = 1024;   // per-step statistics words of the sparse faithful solver kept on the device
const uint32_t kDefaultJacobiHalo = 8;   // sweeps per pressure exchange: 5 messages per 40 sweeps, +11% halo sweeps at 64 planes/rank

inline hipStream_t pick_stream(fx_ctx* ctx, void* s) { return s ? (hipStream_t)s : ctx->stream; }
inline size_t elem_size(const fx_ctx* c) { return c->half ? 2 : 4; }

struct DeviceGuard {
	int prev = -1;
	bool ok = true, changed = false;
	explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) { ok = hipSetDevice(dev) == hipSuccess; changed = true; } }
	~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }     // (nothing to restore on the usual one-device path: a guard per launch costs one hipGetDevice)
};

// ---- timing ---------------------------------------------------------------------------------
// ---- timing ---------------------------------------------------------------------------------
enum MarkKind { MK_ADVECT, MK_DIV, MK_JACOBI, MK_PROJECT, MK_LIGHT, MK_VIEW, MK_EXCH, MK_RESOLVE, MK_JACOBI_TAIL, MK_CHAIN };

size_t ev_record(fx_ctx* c, hipStream_t s);          // fx_context.cpp

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

int drain_timing(fx_ctx* c);
int ensure_stage(fx_ctx* ctx, size_t bytes);
void free_all(fx_ctx* c);
void destroy_lanes(fx_comm_group* g);
void group_release(fx_comm_group* g);

// planes of the local array a stage may compute / read, as global z ranges
struct Range { int lo, hi; };   // [lo, hi)
inline Range owned(const fx_ctx* c) { return Range{ c->g.z0, c->g.z0 + c->g.nz }; }
inline Range grown(const fx_ctx* c, int by)
{
	return Range{ std::max(c->g.z0 - by, 0), std::min(c->g.z0 + c->g.nz + by, c->g.Zg) };
}

inline bool multi_rank(const fx_ctx* c) { return c->group && c->nranks > 1; }
inline bool has_lower(const fx_ctx* c) { return c->nranks > 1 && c->rank > 0; }
inline bool has_upper(const fx_ctx* c) { return c->nranks > 1 && c->rank + 1 < c->nranks; }
inline bool is_driver(const fx_ctx* c) { return !c->group || !c->group->transport->is_local() || c->group->members[0] == c; }
inline bool group_broken(const fx_ctx* c) { return c->group && c->group->transport->is_local() && c->group->broken; }

// ---- fx_schedule.cpp: the step, phase by phase ------------------------------------------------------
int for_members(fx_ctx* ctx, std::vector<fx_ctx*>& out);
int overlap_level(const fx_ctx* lead);
int options_digest(const fx_ctx* c);
int advect_range(fx_ctx* ctx, hipStream_t s, Range r, bool own_only);   // planes [r.lo, r.hi); own_only: back-traces must stay inside the owned planes
int advect_all(fx_ctx* ctx, std::vector<fx_ctx*>& M, hipStream_t s);
int divergence_phase(fx_ctx* ctx, hipStream_t s);
int jacobi_all(fx_ctx* lead, std::vector<fx_ctx*>& M, hipStream_t s, uint32_t iters);
int project_phase(fx_ctx* ctx, hipStream_t s);
int simulate_impl(fx_ctx* ctx, hipStream_t s);

}  // namespace fxh
