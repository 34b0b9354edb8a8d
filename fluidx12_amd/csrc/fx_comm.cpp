// fx_comm.cpp -- halo-exchange transports of the z-slab decomposition (no reference counterpart:
// the reference is single-GPU).
//
//   RcclTransport   one process per GPU; neighbour planes travel as ncclSend/ncclRecv pairs inside one
//                   ncclGroupStart/End per exchange.  xGMI is point-to-point, a slab chain loads two of the
//                   seven links of a GPU, so there is no ring collective anywhere on the step path.
//                   librccl is dlopen()ed on first use so that a single-GPU box never needs it.
//   LocalTransport  several slab contexts in ONE process on ONE device; planes travel as device-to-device
//                   hipMemcpyAsync.  Same halo geometry, used to verify the decomposition on a 1-GPU box.
#include "fx_context.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdio>
#include <cstring>

namespace fx {

int exchange_arrays(fx_ctx* c, int which_set, ExchArray out[4])
{
	const size_t plane = c->g.plane();
	const size_t es = c->half ? 2 : 4;
	switch (which_set) {
	case EX_ADVECT_IN:      // what advection gathers from: velocity[0] and colour[!parity]
		out[0] = ExchArray{ (char*)c->vel[0], plane * es, 3 };
		out[1] = ExchArray{ (char*)c->col[1 - c->frame_parity], plane * es * 4, 1 };
		return 2;
	case EX_VEL1:           // advected velocity, read by the divergence on the halo planes
		out[0] = ExchArray{ (char*)c->vel[1], plane * es, 3 };
		return 1;
	case EX_PRESSURE:
		out[0] = ExchArray{ (char*)c->p[c->p_cur], plane * 4, 1 };
		return 1;
	}
	return 0;
}

// ------------------------------------------------------------------------------------------------
struct LocalTransport : Transport {
	bool is_local() const override { return true; }
	int exchange(fx_comm_group* grp, int which_set, int k, hipStream_t s) override
	{
		const int n = (int)grp->members.size();
		for (int r = 0; r + 1 < n; ++r) {
			fx_ctx* lo = grp->members[r];
			fx_ctx* hi = grp->members[r + 1];
			ExchArray a[4], bb[4];
			const int na = exchange_arrays(lo, which_set, a);
			exchange_arrays(hi, which_set, bb);
			for (int i = 0; i < na; ++i)
				for (int cpt = 0; cpt < a[i].count; ++cpt) {
					const size_t pb = a[i].plane_bytes;
					char* lob = a[i].base + (size_t)cpt * lo->g.nzl() * pb;
					char* hib = bb[i].base + (size_t)cpt * hi->g.nzl() * pb;
					// lo's top k owned planes -> hi's lower halo
					if (hipMemcpyAsync(hib + (size_t)(hi->g.H - k) * pb, lob + (size_t)(lo->g.H + lo->g.nz - k) * pb,
							(size_t)k * pb, hipMemcpyDeviceToDevice, s) != hipSuccess) return FX_E_DEVICE;
					// hi's bottom k owned planes -> lo's upper halo
					if (hipMemcpyAsync(lob + (size_t)(lo->g.H + lo->g.nz) * pb, hib + (size_t)hi->g.H * pb,
							(size_t)k * pb, hipMemcpyDeviceToDevice, s) != hipSuccess) return FX_E_DEVICE;
				}
		}
		return FX_OK;
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
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static RcclApi* rccl(std::string* err)
{
	static RcclApi api;
	static bool tried = false;
	if (api.handle) return &api;
	if (tried) { if (err) *err = "librccl not available"; return nullptr; }
	tried = true;
	// a process that already imported torch has its bundled librccl loaded: the soname lookup reuses it
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	for (const char* n : names) {
		api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (api.handle) break;
	}
	if (!api.handle) { if (err) *err = std::string("dlopen librccl failed: ") + dlerror(); return nullptr; }
#define FX_SYM(f) api.f = (decltype(api.f))dlsym(api.handle, "nccl" #f); \
	if (!api.f) { if (err) *err = "librccl lacks nccl" #f; api.handle = nullptr; return nullptr; }
	FX_SYM(GetUniqueId) FX_SYM(CommInitRank) FX_SYM(CommDestroy) FX_SYM(Send) FX_SYM(Recv)
	FX_SYM(GroupStart) FX_SYM(GroupEnd) FX_SYM(GetErrorString)
#undef FX_SYM
	return &api;
}

size_t rccl_id_bytes() { return sizeof(ncclUniqueId); }

int rccl_get_unique_id(void* out, size_t bytes, std::string* err)
{
	if (bytes < sizeof(ncclUniqueId)) return FX_E_INVALID;
	RcclApi* a = rccl(err);
	if (!a) return FX_E_COMM;
	ncclUniqueId id;
	const ncclResult_t r = a->GetUniqueId(&id);
	if (r != ncclSuccess) { if (err) *err = a->GetErrorString(r); return FX_E_COMM; }
	std::memcpy(out, &id, sizeof id);
	return FX_OK;
}

struct RcclTransport : Transport {
	RcclApi* api;
	ncclComm_t comm;
	int rank, nranks;
	bool is_local() const override { return false; }
	~RcclTransport() override { if (comm) api->CommDestroy(comm); }
	int exchange(fx_comm_group* grp, int which_set, int k, hipStream_t s) override
	{
		fx_ctx* c = grp->members[0];
		ExchArray a[4];
		const int na = exchange_arrays(c, which_set, a);
		const Geom& g = c->g;
		ncclResult_t r = api->GroupStart();
		for (int i = 0; i < na && r == ncclSuccess; ++i)
			for (int cpt = 0; cpt < a[i].count && r == ncclSuccess; ++cpt) {
				const size_t pb = a[i].plane_bytes, n = (size_t)k * pb;
				char* base = a[i].base + (size_t)cpt * g.nzl() * pb;
				if (rank > 0) {          // lower z-neighbour
					r = api->Send(base + (size_t)g.H * pb, n, ncclInt8, rank - 1, comm, s);
					if (r == ncclSuccess) r = api->Recv(base + (size_t)(g.H - k) * pb, n, ncclInt8, rank - 1, comm, s);
				}
				if (rank + 1 < nranks && r == ncclSuccess) {   // upper z-neighbour
					r = api->Send(base + (size_t)(g.H + g.nz - k) * pb, n, ncclInt8, rank + 1, comm, s);
					if (r == ncclSuccess) r = api->Recv(base + (size_t)(g.H + g.nz) * pb, n, ncclInt8, rank + 1, comm, s);
				}
			}
		const ncclResult_t e = api->GroupEnd();
		if (r == ncclSuccess) r = e;
		if (r != ncclSuccess) { c->last_error = std::string("rccl: ") + api->GetErrorString(r); return FX_E_COMM; }
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
	t->api = a; t->comm = nullptr; t->rank = rank; t->nranks = nranks;
	const ncclResult_t r = a->CommInitRank(&t->comm, nranks, uid, rank);
	if (r != ncclSuccess) { if (err) *err = std::string("ncclCommInitRank: ") + a->GetErrorString(r); t->comm = nullptr; delete t; return nullptr; }
	return t;
}

}  // namespace fx
