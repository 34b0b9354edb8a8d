// fx_comm.cpp -- halo-exchange transports of the z-slab decomposition (no reference counterpart:
// the reference is single-GPU).
//
//   RcclTransport   one process per GPU; neighbour planes travel as ncclSend/ncclRecv pairs inside one
//                   ncclGroupStart/End per exchange.  xGMI is point-to-point, a slab chain loads two of the
//                   seven links of a GPU, so there is no ring collective anywhere on the step path.
//                   librccl is dlopen()ed on first use so that a single-GPU box never needs it.
//   LocalTransport  several slab contexts in ONE process; same halo geometry and phase schedule.  Two kinds of group:
//                   shared stream (fx_comm_init_local) -- one device, every member on one compute stream, planes travel as copy
//                   kernels on it: the decomposition's arithmetic on a 1-GPU box;
//                   peer (fx_comm_init_peer) -- every member on its own streams, on its own device if it likes: a receiver PULLS
//                   its halo planes out of the neighbour's memory (hipMemcpyPeerAsync across devices; no IPC handles -- one
//                   process owns all devices), ordered by a ready / done event pair per lane and exchange.
#include "fx_context.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace fx {

int exchange_items(fx_ctx* c, int which_set, int k, int pidx, ExchItem out[4])
{
	const size_t plane = c->g.plane();
	const size_t es = c->half ? 2 : 4;
	switch (which_set) {
	case EX_ADVECT_IN:
		out[0] = ExchItem{ (char*)c->vel[0], plane * es, 3, k, nullptr, c->adv_w_lo, c->adv_w_hi };
		out[1] = ExchItem{ (char*)c->col[1 - c->frame_parity], plane * es * 4, 1, k, nullptr, c->adv_w_lo, c->adv_w_hi };
		return 2;
	case EX_UZ1:
		out[0] = ExchItem{ (char*)c->vel[1] + 2 * (size_t)c->g.nzl() * plane * es, plane * es, 1, k, nullptr };
		return 1;
	case EX_DIV:
		out[0] = ExchItem{ (char*)c->b, plane * 4, 1, k, nullptr };
		return 1;
	case EX_PRESSURE:
		// the member's OWN current buffer: the members of an in-process group may have taken different numbers of launches for the
		// same sweeps (a 12-plane slab runs fours where its 8-plane neighbour runs ones), so the lead's index is not theirs.
		// (A rank of an RCCL chain is its own lead: pidx == c->p_cur there.  Found by the wide-row slab fuzz of round 6.)
		(void)pidx;
		out[0] = ExchItem{ (char*)c->p[c->p_cur & 1], plane * 4, 1, k, nullptr };
		if (!c->frozen) return 1;
		out[1] = ExchItem{ (char*)c->frozen, plane, 1, k, nullptr };      // the neighbour's freeze state travels with its pressure
		return 2;
	case EX_ADVECT_VEL:
		out[0] = ExchItem{ (char*)c->vel[0], plane * es, 3, k, nullptr, c->adv_w_lo, c->adv_w_hi };
		return 1;
	case EX_COLOR_CUR:
		out[0] = ExchItem{ (char*)c->col[c->frame_parity], plane * es * 4, 1, k, nullptr };
		return 1;
	case EX_FREEZE:                                      // (mask: a byte per 4-cell quad)
		out[0] = ExchItem{ (char*)c->fz_x_p, plane * 4, 1, k, nullptr };
		out[1] = ExchItem{ (char*)c->fz_x_m, (size_t)((c->g.X + 3) / 4) * c->g.Y, 1, k, nullptr };
		return 2;
	case EX_PRESSURE_FACE:
		out[0] = ExchItem{ (char*)c->p_face[(pidx >> 1) & 1], plane * 4, 1, k, (char*)c->p[(c->p_cur ^ pidx) & 1] };   // bit 0: relative to the member's current buffer
		return 1;
	}
	return 0;
}

void halo_segments(const fx_ctx* c, const ExchItem* items, int n, std::vector<Seg>& out)
{
	const Geom& g = c->g;
	for (int i = 0; i < n; ++i) {
		const ExchItem& it = items[i];
		if (it.k <= 0) continue;
		const size_t pb = it.plane_bytes;
		const int klo = it.k_lo >= 0 ? it.k_lo : it.k, khi = it.k_hi >= 0 ? it.k_hi : it.k;   // the same number on both sides of a face
		for (int cpt = 0; cpt < it.ncomp; ++cpt) {
			char* base = it.base + (size_t)cpt * g.nzl() * pb;
			char* rbase = (it.recv_base ? it.recv_base : it.base) + (size_t)cpt * g.nzl() * pb;
			if (c->rank > 0 && klo > 0)          // bottom k owned planes go down, the lower halo fills from below
				out.push_back(Seg{ base + (size_t)g.H * pb, rbase + (size_t)(g.H - klo) * pb, (size_t)klo * pb, -1 });
			if (c->rank + 1 < c->nranks && khi > 0)   // top k owned planes go up, the upper halo fills from above
				out.push_back(Seg{ base + (size_t)(g.H + g.nz - khi) * pb, rbase + (size_t)(g.H + g.nz) * pb, (size_t)khi * pb, +1 });
		}
	}
}

// ------------------------------------------------------------------------------------------------
struct LocalTransport : Transport {
	bool is_local() const override { return true; }
	int min_over_ranks(int v, hipStream_t, int* out) override { *out = v; return FX_OK; }   // the caller sees every member
	int allgather(const int*, int, int*, hipStream_t) override { return FX_OK; }
	int exchange(fx_comm_group* grp, const std::vector<std::vector<Seg>>& segs, const std::vector<hipStream_t>& streams, int channel) override
	{
		(void)channel;                                   // copies on the given streams either way
		const int n = (int)grp->members.size();
		if ((int)segs.size() != n || streams.empty()) return FX_E_STATE;
		const bool peer = streams.size() > 1;
		if (peer && ((int)streams.size() != n || (int)grp->lanes.size() != n)) return FX_E_STATE;
#ifdef FX_LAB
		// lab builds only (-DFX_LAB; results become WRONG): keep the streams / events of the schedule, drop the copies -- what the copies
		// themselves cost in loop-back.  Not in the shipped library: every fx_set_knob name leaves results unchanged.
		const bool no_copy = [] { const char* e = FX_KNOB("DEBUG_NO_COPY"); return e && e[0] == '1'; }();
		if (no_copy) return FX_OK;
#endif
		int home = -1;
		(void)hipGetDevice(&home);
		int rc = FX_OK;
		// peer groups: lane r may read its neighbours' planes once they are final there, and write its own halo once its own stream
		// has got here
		if (peer) {
			for (int r = 0; r < n && rc == FX_OK; ++r)
				if (hipSetDevice(grp->lanes[r].device) != hipSuccess || hipEventRecord(grp->lanes[r].x_ready, streams[r]) != hipSuccess) rc = FX_E_DEVICE;
			for (int r = 0; r < n && rc == FX_OK; ++r)
				for (int d = -1; d <= 1; d += 2) {
					const int q = r + d;
					if (q < 0 || q >= n) continue;
					if (hipSetDevice(grp->lanes[r].device) != hipSuccess || hipStreamWaitEvent(streams[r], grp->lanes[q].x_ready, 0) != hipSuccess) rc = FX_E_DEVICE;
				}
		}
		// every member "receives": its j-th segment from direction d pairs with the peer's j-th segment towards -d
		for (int r = 0; r < n && rc == FX_OK; ++r) {
			hipStream_t s = peer ? streams[r] : streams[0];
			if (peer && hipSetDevice(grp->lanes[r].device) != hipSuccess) { rc = FX_E_DEVICE; break; }
			for (int d = -1; d <= 1 && rc == FX_OK; d += 2) {
				const int q = r + d;
				if (q < 0 || q >= n) continue;
				size_t jp = 0;
				for (const Seg& mine : segs[r]) {
					if (mine.dir != d) continue;
					while (jp < segs[q].size() && segs[q][jp].dir != -d) ++jp;
					if (jp == segs[q].size() || segs[q][jp].bytes != mine.bytes) { rc = FX_E_STATE; break; }   // the lists must mirror
					hipError_t e;
					if (peer && grp->lanes[r].device != grp->lanes[q].device)
						e = hipMemcpyPeerAsync(mine.recv, grp->lanes[r].device, segs[q][jp].send, grp->lanes[q].device, mine.bytes, s);
					else e = launch_copy_bytes(mine.recv, segs[q][jp].send, mine.bytes, s);
					if (e != hipSuccess) { rc = FX_E_DEVICE; break; }
					++jp;
				}
			}
		}
		// ... and nobody moves on (to overwrite what a neighbour is still pulling) before both neighbours have pulled
		if (peer) {
			for (int r = 0; r < n && rc == FX_OK; ++r)
				if (hipSetDevice(grp->lanes[r].device) != hipSuccess || hipEventRecord(grp->lanes[r].x_done, streams[r]) != hipSuccess) rc = FX_E_DEVICE;
			for (int r = 0; r < n && rc == FX_OK; ++r)
				for (int d = -1; d <= 1; d += 2) {
					const int q = r + d;
					if (q < 0 || q >= n) continue;
					if (hipSetDevice(grp->lanes[r].device) != hipSuccess || hipStreamWaitEvent(streams[r], grp->lanes[q].x_done, 0) != hipSuccess) rc = FX_E_DEVICE;
				}
		}
		if (home >= 0) (void)hipSetDevice(home);
		return rc;
	}
	int gather(fx_comm_group* grp, const std::vector<GatherPart>& parts, int root, const std::vector<hipStream_t>& streams) override
	{
		const int n = (int)grp->members.size();
		const bool peer = streams.size() > 1;
		if (streams.empty() || (peer && ((int)streams.size() != n || root < 0 || root >= n))) return FX_E_STATE;
		hipStream_t s = peer ? streams[root] : streams[0];
		int home = -1;
		(void)hipGetDevice(&home);
		int rc = FX_OK;
		if (peer) {                                      // the root's stream waits for what every member has enqueued
			for (int r = 0; r < n && rc == FX_OK; ++r)
				if (hipSetDevice(grp->lanes[r].device) != hipSuccess || hipEventRecord(grp->lanes[r].x_ready, streams[r]) != hipSuccess) rc = FX_E_DEVICE;
			for (int r = 0; r < n && rc == FX_OK; ++r)
				if (r != root && (hipSetDevice(grp->lanes[root].device) != hipSuccess || hipStreamWaitEvent(s, grp->lanes[r].x_ready, 0) != hipSuccess)) rc = FX_E_DEVICE;
			if (rc == FX_OK && hipSetDevice(grp->lanes[root].device) != hipSuccess) rc = FX_E_DEVICE;
		}
		for (const GatherPart& p : parts) {
			if (rc != FX_OK || !p.bytes) continue;
			hipError_t e;
			if (peer && grp->lanes[p.rank].device != grp->lanes[root].device)
				e = hipMemcpyPeerAsync(p.dst, grp->lanes[root].device, p.src, grp->lanes[p.rank].device, p.bytes, s);
			else e = launch_copy_bytes(p.dst, p.src, p.bytes, s);
			if (e != hipSuccess) rc = FX_E_DEVICE;
		}
		if (peer && rc == FX_OK) {                       // the members must not overwrite their planes before the root has them
			if (hipEventRecord(grp->lanes[root].x_done, s) != hipSuccess) rc = FX_E_DEVICE;
			for (int r = 0; r < n && rc == FX_OK; ++r)
				if (r != root && (hipSetDevice(grp->lanes[r].device) != hipSuccess || hipStreamWaitEvent(streams[r], grp->lanes[root].x_done, 0) != hipSuccess)) rc = FX_E_DEVICE;
		}
		if (home >= 0) (void)hipSetDevice(home);
		return rc;
	}
};

Transport* make_local_transport() { return new LocalTransport(); }

// ------------------------------------------------------------------------------------------------
struct RcclApi {
	void* handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;     // optional: a library without them is never polled / aborted
	decltype(&ncclCommAbort) CommAbort = nullptr;
};

static RcclApi* rccl(std::string* err)
{
	static RcclApi api;
	static bool tried = false;
	if (api.handle) return &api;
	if (tried) { if (err) *err = "librccl not available"; return nullptr; }
	tried = true;
	// a process that already imported torch has its bundled librccl loaded: the soname lookup reuses it
	// FLUIDX_RCCL_LIB names one library explicitly (a site build of RCCL, or tests/mock_rccl's single-GPU stand-in) and
	// is then the only candidate: a wrong path fails loudly instead of silently binding some other librccl.
	const char* forced = std::getenv("FLUIDX_RCCL_LIB");
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	if (forced && forced[0]) api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
	else for (const char* n : names) {
		api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (api.handle) break;
	}
	if (!api.handle) { if (err) *err = std::string("dlopen librccl failed: ") + dlerror(); return nullptr; }
#define FX_SYM(f) api.f = (decltype(api.f))dlsym(api.handle, "nccl" #f); \
	if (!api.f) { if (err) *err = "librccl lacks nccl" #f; api.handle = nullptr; return nullptr; }
	FX_SYM(GetUniqueId) FX_SYM(CommInitRank) FX_SYM(CommDestroy) FX_SYM(Send) FX_SYM(Recv)
	FX_SYM(GroupStart) FX_SYM(GroupEnd) FX_SYM(AllReduce) FX_SYM(AllGather) FX_SYM(GetErrorString)
#undef FX_SYM
	api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(api.handle, "ncclCommGetAsyncError");
	api.CommAbort = (decltype(api.CommAbort))dlsym(api.handle, "ncclCommAbort");
	return &api;
}

// two ids: the communicator of the step's exchanges and the one of the side channel (Transport::exchange, channel 1)
size_t rccl_id_bytes() { return 2 * sizeof(ncclUniqueId); }

int rccl_get_unique_id(void* out, size_t bytes, std::string* err)
{
	if (bytes < sizeof(ncclUniqueId)) return FX_E_INVALID;
	RcclApi* a = rccl(err);
	if (!a) return FX_E_COMM;
	const size_t n = bytes >= 2 * sizeof(ncclUniqueId) ? 2 : 1;         // a caller with room for one id gets one communicator
	for (size_t i = 0; i < n; ++i) {
		ncclUniqueId id;
		const ncclResult_t r = a->GetUniqueId(&id);
		if (r != ncclSuccess) { if (err) *err = a->GetErrorString(r); return FX_E_COMM; }
		std::memcpy(static_cast<char*>(out) + i * sizeof id, &id, sizeof id);
	}
	return FX_OK;
}

struct RcclTransport : Transport {
	RcclApi* api;
	ncclComm_t comm, comm2;                          // comm2: the side channel (== comm when it could not be split off)
	int rank, nranks;
	bool dead = false;                               // aborted after an asynchronous error: nothing may be enqueued on the communicators any more
	std::string dead_why;
	bool is_local() const override { return false; }
	~RcclTransport() override { if (comm2 && comm2 != comm) api->CommDestroy(comm2); if (comm) api->CommDestroy(comm); }
	bool can_poll() const override { return api->CommGetAsyncError != nullptr; }
	// ncclCommGetAsyncError of both communicators.  Anything but success / in-progress: abort both (ncclCommAbort ends the kernels RCCL
	// has on the device, so the streams drain instead of spinning on a peer that will never answer) and stay dead.
	int poll_error(std::string* err) override
	{
		if (!dead && api->CommGetAsyncError) {
			ncclComm_t cs[2] = { comm, comm2 != comm ? comm2 : nullptr };
			for (ncclComm_t c : cs) {
				if (!c || dead) continue;
				ncclResult_t st = ncclSuccess;
				const ncclResult_t r = api->CommGetAsyncError(c, &st);
				if (r != ncclSuccess) st = r;
				if (st != ncclSuccess && st != ncclInProgress) {
					dead = true;
					dead_why = std::string("rccl: asynchronous communicator error (") + api->GetErrorString(st) + "); the communicators were aborted -- destroy the context and rejoin from a fresh process";
				}
			}
			if (dead) {
				if (api->CommAbort) {
					if (comm2 && comm2 != comm) (void)api->CommAbort(comm2);
					if (comm) (void)api->CommAbort(comm);
				}
				// (aborted communicators are freed by ncclCommAbort: the destructor must not touch them.  A library without ncclCommAbort leaves
				// a broken communicator behind: ncclCommDestroy on it can hang in its release path, so it is LEAKED rather than destroyed)
				comm = comm2 = nullptr;
			}
		}
		if (dead) { if (err) *err = dead_why; return FX_E_COMM; }
		return FX_OK;
	}
	int fail(fx_ctx* c, const char* what, ncclResult_t r)
	{
		c->last_error = std::string(what) + api->GetErrorString(r);
		// a failed call leaves the communicator in an undefined state (RCCL's own rule): abort it, so that neither this rank's streams nor
		// -- through the closed connections -- its neighbours wait on it for ever
		if (!dead) {
			dead = true; dead_why = c->last_error + "; the communicators were aborted";
			if (api->CommAbort) { if (comm2 && comm2 != comm) (void)api->CommAbort(comm2); if (comm) (void)api->CommAbort(comm); }
			comm = comm2 = nullptr;                        // (with or without ncclCommAbort: a broken communicator is never handed to ncclCommDestroy -- leaked rather than hung)
		}
		return FX_E_COMM;
	}
	// smallest value of `v` over the ranks (one-time set-up traffic: the ranks must take the same schedule decisions)
	int min_over_ranks(int v, hipStream_t s, int* out) override
	{
		if (poll_error(nullptr)) return FX_E_COMM;
		int* d = nullptr;
		if (hipMalloc((void**)&d, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return FX_E_NOMEM; }   // (clears the sticky last error)
		int rc = FX_OK;
		if (hipMemcpyAsync(d, &v, sizeof v, hipMemcpyHostToDevice, s) != hipSuccess) rc = FX_E_DEVICE;
		if (rc == FX_OK && api->AllReduce(d, d, 1, ncclInt32, ncclMin, comm, s) != ncclSuccess) rc = FX_E_COMM;
		if (rc == FX_OK && (hipMemcpyAsync(out, d, sizeof v, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = FX_E_DEVICE;
		(void)hipFree(d);
		return rc;
	}
	int allgather(const int* send_dev, int count, int* recv_dev, hipStream_t s) override
	{
		if (poll_error(nullptr)) return FX_E_COMM;
		return api->AllGather(send_dev, recv_dev, (size_t)count, ncclInt32, comm, s) == ncclSuccess ? FX_OK : FX_E_COMM;
	}
	int exchange(fx_comm_group* grp, const std::vector<std::vector<Seg>>& segs, const std::vector<hipStream_t>& streams, int channel) override
	{
		fx_ctx* c = grp->members[0];
		if (segs.size() != 1 || streams.size() != 1) return FX_E_STATE;
		if (poll_error(&c->last_error)) return FX_E_COMM;                    // before anything new is enqueued behind a broken link
		hipStream_t s = streams[0];
		ncclComm_t comm = channel == 1 ? this->comm2 : this->comm;
		ncclResult_t r = api->GroupStart();
		for (const Seg& sg : segs[0]) {
			if (r != ncclSuccess) break;
			const int peer = rank + sg.dir;
			if (peer < 0 || peer >= nranks) { r = ncclInvalidArgument; break; }
			r = api->Send(sg.send, sg.bytes, ncclInt8, peer, comm, s);
			if (r == ncclSuccess) r = api->Recv(sg.recv, sg.bytes, ncclInt8, peer, comm, s);
		}
		const ncclResult_t e = api->GroupEnd();
		if (r == ncclSuccess) r = e;
		if (r != ncclSuccess) return fail(c, "rccl: ", r);
		return FX_OK;
	}
	int gather(fx_comm_group* grp, const std::vector<GatherPart>& parts, int root, const std::vector<hipStream_t>& streams) override
	{
		fx_ctx* c = grp->members[0];
		if ((int)parts.size() != nranks || root < 0 || root >= nranks || streams.size() != 1) return FX_E_INVALID;
		if (poll_error(&c->last_error)) return FX_E_COMM;
		hipStream_t s = streams[0];
		ncclResult_t r = api->GroupStart();
		if (rank == root) {
			for (const GatherPart& p : parts) {
				if (r != ncclSuccess || !p.bytes) continue;
				if (p.rank == root) { if (hipMemcpyAsync(p.dst, p.src, p.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) r = ncclUnhandledCudaError; }
				else r = api->Recv(p.dst, p.bytes, ncclInt8, p.rank, comm, s);
			}
		} else if (parts[rank].bytes) {
			r = api->Send(parts[rank].src, parts[rank].bytes, ncclInt8, root, comm, s);
		}
		const ncclResult_t e = api->GroupEnd();
		if (r == ncclSuccess) r = e;
		if (r != ncclSuccess) return fail(c, "rccl gather: ", r);
		return FX_OK;
	}
};

Transport* make_rccl_transport(const void* id, size_t bytes, int rank, int nranks, int device, std::string* err)
{
	if (bytes < sizeof(ncclUniqueId)) { if (err) *err = "unique id too short"; return nullptr; }
	RcclApi* a = rccl(err);
	if (!a) return nullptr;
	if (hipSetDevice(device) != hipSuccess) { if (err) *err = "hipSetDevice failed"; return nullptr; }
	ncclUniqueId uid;
	std::memcpy(&uid, id, sizeof uid);
	RcclTransport* t = new RcclTransport();
	t->api = a; t->comm = nullptr; t->comm2 = nullptr; t->rank = rank; t->nranks = nranks;
	const ncclResult_t r = a->CommInitRank(&t->comm, nranks, uid, rank);
	if (r != ncclSuccess) { if (err) *err = std::string("ncclCommInitRank: ") + a->GetErrorString(r); t->comm = nullptr; delete t; return nullptr; }
	// the side channel: a second communicator from the second id, when the caller passed two (fx_comm_id_bytes); the plain
	// ncclCommInitRank route, the one every framework exercises
	t->comm2 = t->comm;
	const bool one_comm = [] { const char* e = FX_KNOB("RCCL_ONE_COMM"); return e && e[0] == '1'; }();
	if (bytes >= 2 * sizeof(ncclUniqueId) && !one_comm) {
		ncclUniqueId uid2;
		std::memcpy(&uid2, static_cast<const char*>(id) + sizeof uid, sizeof uid2);
		ncclComm_t c2 = nullptr;
		const ncclResult_t r2 = a->CommInitRank(&c2, nranks, uid2, rank);
		if (r2 != ncclSuccess) { if (err) *err = std::string("ncclCommInitRank (side channel): ") + a->GetErrorString(r2); delete t; return nullptr; }
		t->comm2 = c2;
	}
	return t;
}

}  // namespace fx
