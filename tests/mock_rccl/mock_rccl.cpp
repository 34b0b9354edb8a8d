// TEST INFRASTRUCTURE -- a stand-in for the handful of RCCL entry points fx_comm.cpp binds, for boxes with ONE GPU.
//
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the one-process-per-GPU slab path
// (RcclTransport, fx_comm_init_rank, fx_comm_gather_color, bench.py under torch.distributed.run) could otherwise
// only run for the first time on the driver's 8-GPU node.  This library gives the same calls rendezvous semantics
// across PROCESSES that share a GPU, through files in a directory named by the unique id:
//
//   ncclSend/ncclRecv inside ncclGroupStart/End   recorded; at the outermost GroupEnd the stream is drained, every
//       send is published as <dir>/m_<src>_<dst>_<seq> (tmp + rename), then every recv waits for its file
//       m_<peer>_<me>_<seq>, checks the BYTE COUNT against what the receiver posted (a mismatch is an error, where
//       real RCCL would hang or corrupt), copies it to the device and unlinks it.  Matching is FIFO per ordered
//       pair, which is RCCL's rule.
//   ncclAllReduce                                  int32 / float, min / max / sum, by the same file exchange.
//   ncclAllGather                                  any of the element types above, by the same file exchange.
//
//   ncclCommGetAsyncError / ncclCommAbort         the failure path: every rank leaves <dir>/pid_<rank> at init; a wait that sees its peer's
//       process gone (kill(pid, 0)), or the peer's <dir>/dead_<rank> marker (written by ncclCommAbort), ends at once with
//       ncclRemoteError -- the stand-in for RCCL noticing a closed connection -- instead of running into the time-out.
//       FXMOCK_ASYNC_ERROR="<rank>:<n>" makes ncclCommGetAsyncError on that rank report ncclSystemError from its n-th completed group on
//       (an injected link failure).  After ncclCommAbort the handle is gone.
//
// It is deliberately STRICTER than RCCL: host-synchronous, every wait times out (FXMOCK_TIMEOUT_S, default 60 s) and
// returns ncclSystemError, so a schedule that makes ranks disagree fails a test instead of hanging a node.  It says
// nothing about link time.  Selected with FLUIDX_RCCL_LIB=<this .so>; FXMOCK_DIR is where the rendezvous
// directories go (default /dev/shm).  Built by tests/test_gpu_rccl_mock.py; never loaded by the product on its own.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <signal.h>
#include <errno.h>

namespace {

struct Op { bool send; void* ptr; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
	std::string dir;
	int rank = 0, nranks = 0;
	std::vector<unsigned long> send_seq, recv_seq;
	unsigned long ar_seq = 0, ag_seq = 0, groups_done = 0;
	std::vector<long> peer_pid;                      // read lazily from <dir>/pid_<rank>
};

thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm*, Op>> g_ops;

double timeout_s()
{
	const char* e = std::getenv("FXMOCK_TIMEOUT_S");
	return e ? std::atof(e) : 60.0;
}

bool write_file(const std::string& path, const void* data, size_t bytes)
{
	const std::string tmp = path + ".tmp";
	FILE* f = std::fopen(tmp.c_str(), "wb");
	if (!f) return false;
	const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
	std::fclose(f);
	return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}

// is rank `peer` of this communicator gone?  (its abort marker, or its process no longer exists)
bool peer_dead(Comm* c, int peer)
{
	if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return false;
	struct stat st;
	char b[64];
	std::snprintf(b, sizeof b, "/dead_%d", peer);
	if (stat((c->dir + b).c_str(), &st) == 0) return true;
	if (c->peer_pid.empty()) c->peer_pid.assign((size_t)c->nranks, 0);
	if (!c->peer_pid[(size_t)peer]) {
		std::snprintf(b, sizeof b, "/pid_%d", peer);
		if (FILE* f = std::fopen((c->dir + b).c_str(), "r")) { long v = 0; if (std::fscanf(f, "%ld", &v) == 1) c->peer_pid[(size_t)peer] = v; std::fclose(f); }
	}
	const long pid = c->peer_pid[(size_t)peer];
	if (pid <= 0) return false;
	if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) return true;
	// a process that has ended but was not reaped yet (a zombie) still answers kill(pid, 0): look at its state
	std::snprintf(b, sizeof b, "/proc/%ld/stat", pid);
	if (FILE* f = std::fopen(b, "r")) {
		char line[512] = {};
		const size_t n = std::fread(line, 1, sizeof line - 1, f);
		std::fclose(f);
		line[n] = 0;
		const char* p = std::strrchr(line, ')');
		if (p && p[1] == ' ' && (p[2] == 'Z' || p[2] == 'X')) return true;
	}
	return false;
}

thread_local bool g_remote_dead = false;             // the last failed wait ended because the peer is gone

// waits until `path` exists, then reads it whole; false on time-out or when the rank that should write it is gone
bool read_file(const std::string& path, std::vector<char>& out, Comm* c = nullptr, int from = -1)
{
	const auto t0 = std::chrono::steady_clock::now();
	struct stat st;
	unsigned long spins = 0;
	g_remote_dead = false;
	while (stat(path.c_str(), &st) != 0) {
		if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) {
			std::fprintf(stderr, "mock_rccl: timed out waiting for %s\n", path.c_str());
			return false;
		}
		if ((++spins & 1023u) == 0 && peer_dead(c, from) && stat(path.c_str(), &st) != 0) {   // every ~50 ms
			std::fprintf(stderr, "mock_rccl: rank %d is gone; giving up on %s\n", from, path.c_str());
			g_remote_dead = true;
			return false;
		}
		std::this_thread::sleep_for(std::chrono::microseconds(50));
	}
	out.resize((size_t)st.st_size);
	FILE* f = std::fopen(path.c_str(), "rb");
	if (!f) return false;
	const bool ok = out.empty() || std::fread(out.data(), 1, out.size(), f) == out.size();
	std::fclose(f);
	return ok;
}

std::string msg_name(const Comm* c, int src, int dst, unsigned long seq)
{
	char b[96];
	std::snprintf(b, sizeof b, "/m_%d_%d_%lu", src, dst, seq);
	return c->dir + b;
}

ncclResult_t flush()
{
	std::vector<std::pair<Comm*, Op>> ops;
	ops.swap(g_ops);
	// drain every stream that carries one of the operations: the payloads must be final before they are read
	std::vector<hipStream_t> seen;
	for (auto& co : ops) {
		bool dup = false;
		for (hipStream_t s : seen) dup |= (s == co.second.stream);
		if (dup) continue;
		seen.push_back(co.second.stream);
		if (hipStreamSynchronize(co.second.stream) != hipSuccess) return ncclUnhandledCudaError;
	}
	std::vector<char> host;
	for (auto& co : ops) {
		Comm* c = co.first; const Op& o = co.second;
		if (!o.send) continue;
		host.resize(o.bytes);
		if (o.bytes && hipMemcpy(host.data(), o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
		if (!write_file(msg_name(c, c->rank, o.peer, c->send_seq[o.peer]++), host.data(), o.bytes)) return ncclSystemError;
	}
	for (auto& co : ops) {
		Comm* c = co.first; const Op& o = co.second;
		if (o.send) continue;
		const std::string name = msg_name(c, o.peer, c->rank, c->recv_seq[o.peer]++);
		if (!read_file(name, host, c, o.peer)) return g_remote_dead ? ncclRemoteError : ncclSystemError;
		if (host.size() != o.bytes) {
			std::fprintf(stderr, "mock_rccl: rank %d posted a %zu-byte recv from %d, the matching send has %zu bytes (%s)\n",
			             c->rank, o.bytes, o.peer, host.size(), name.c_str());
			return ncclInvalidUsage;
		}
		// fault injection (FXMOCK_CORRUPT="rank:n"): the n-th message of at least 4 KiB that rank receives arrives with 1e-3 added to
		// every 251st word -- a halo plane that is not what the neighbour sent; what bench.py's `multi_rank_parity` must notice
		if (const char* e = std::getenv("FXMOCK_CORRUPT")) {
			int r_ = -1; long n_ = -1;
			if (std::sscanf(e, "%d:%ld", &r_, &n_) == 2 && r_ == c->rank && o.bytes >= 4096) {
				static long seen_ = 0;
				if (seen_++ == n_) {                                       // the exponent byte of every 251st word: some of them lie in a plane next to the owned ones
					for (size_t w_ = 0; 4 * w_ + 3 < o.bytes; w_ += 251) {   // (+ 1e-3 as fp32: enough to change every field, too little to trace out of the halo)
						float v_; std::memcpy(&v_, &host[4 * w_], 4); v_ += 1e-3f; std::memcpy(&host[4 * w_], &v_, 4);
					}
					std::fprintf(stderr, "mock_rccl: rank %d: corrupted one byte of every 251st word of a %zu-byte message from %d\n", c->rank, o.bytes, o.peer);
				}
			}
		}
		if (o.bytes && hipMemcpy(o.ptr, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
		unlink(name.c_str());
	}
	for (auto& co : ops) co.first->groups_done += 1;     // (per operation; only its growth matters)
	return ncclSuccess;
}

size_t type_bytes(ncclDataType_t t)
{
	switch (t) {
	case ncclInt8: case ncclUint8: return 1;
	case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
	case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
	default: return 0;
	}
}

template <class T> void reduce(T* acc, const T* in, size_t n, ncclRedOp_t op)
{
	for (size_t i = 0; i < n; ++i)
		acc[i] = op == ncclMin ? (in[i] < acc[i] ? in[i] : acc[i]) : op == ncclMax ? (in[i] > acc[i] ? in[i] : acc[i]) : acc[i] + in[i];
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
	std::memset(id, 0, sizeof *id);
	unsigned char rnd[12] = {};
	FILE* f = std::fopen("/dev/urandom", "rb");
	if (f) { (void)!std::fread(rnd, 1, sizeof rnd, f); std::fclose(f); }
	char* s = id->internal;
	std::memcpy(s, "fxmock_", 7);
	for (size_t i = 0; i < sizeof rnd; ++i) std::snprintf(s + 7 + 2 * i, 3, "%02x", rnd[i]);
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
	if (!comm || nranks < 1 || rank < 0 || rank >= nranks || std::strncmp(id.internal, "fxmock_", 7) != 0) return ncclInvalidArgument;
	const char* base = std::getenv("FXMOCK_DIR");
	Comm* c = new Comm();
	c->dir = std::string(base && base[0] ? base : "/dev/shm") + "/" + std::string(id.internal, strnlen(id.internal, 64));
	c->rank = rank; c->nranks = nranks;
	c->send_seq.assign(nranks, 0); c->recv_seq.assign(nranks, 0);
	mkdir(c->dir.c_str(), 0700);                       // every rank tries; EEXIST is the normal case
	struct stat st;
	if (stat(c->dir.c_str(), &st) != 0) { delete c; return ncclSystemError; }
	{
		char b[64], pid[32];
		std::snprintf(b, sizeof b, "/pid_%d", rank);
		const int n = std::snprintf(pid, sizeof pid, "%ld", (long)getpid());
		if (!write_file(c->dir + b, pid, (size_t)n)) { delete c; return ncclSystemError; }
	}
	*comm = reinterpret_cast<ncclComm_t>(c);
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	if (!c) return ncclSuccess;
	// remove what this rank published and nobody consumed (all-reduce contributions; sends of an aborted run)
	char mine[32], ar[32];
	std::snprintf(mine, sizeof mine, "m_%d_", c->rank);
	std::snprintf(ar, sizeof ar, "_r%d", c->rank);
	if (DIR* d = opendir(c->dir.c_str())) {
		while (dirent* e = readdir(d)) {
			const std::string n = e->d_name;
			const bool own = n.compare(0, std::strlen(mine), mine) == 0 ||
			                 ((n.compare(0, 3, "ar_") == 0 || n.compare(0, 3, "ag_") == 0) && n.size() > std::strlen(ar) && n.compare(n.size() - std::strlen(ar), std::strlen(ar), ar) == 0);
			if (own) unlink((c->dir + "/" + n).c_str());
		}
		closedir(d);
	}
	char pf[64];
	std::snprintf(pf, sizeof pf, "/pid_%d", c->rank);
	unlink((c->dir + pf).c_str());
	rmdir(c->dir.c_str());                              // succeeds for the last rank out
	delete c;
	return ncclSuccess;
}

// the asynchronous state of a communicator: an injected failure (FXMOCK_ASYNC_ERROR = "<rank>:<operations completed before it shows>"),
// or a neighbour that is gone
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	if (!c || !asyncError) return ncclInvalidArgument;
	*asyncError = ncclSuccess;
	if (const char* e = std::getenv("FXMOCK_ASYNC_ERROR")) {
		int r = -1; unsigned long n = 0;
		if (std::sscanf(e, "%d:%lu", &r, &n) == 2 && r == c->rank && c->groups_done >= n) *asyncError = ncclSystemError;
	}
	for (int d = -1; d <= 1 && *asyncError == ncclSuccess; d += 2)
		if (peer_dead(c, c->rank + d)) *asyncError = ncclRemoteError;
	return ncclSuccess;
}

// leaves the marker the neighbours' waits look for, and frees the handle (what it published stays: nobody may rely on it any more)
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	if (!c) return ncclSuccess;
	char b[64];
	std::snprintf(b, sizeof b, "/dead_%d", c->rank);
	(void)write_file(c->dir + b, "x", 1);
	delete c;
	return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
	if (g_depth <= 0) return ncclInvalidUsage;
	if (--g_depth > 0) return ncclSuccess;
	return flush();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	if (!c || peer < 0 || peer >= c->nranks || !type_bytes(type)) return ncclInvalidArgument;
	g_ops.push_back({ c, Op{ true, const_cast<void*>(buf), count * type_bytes(type), peer, stream } });
	return g_depth ? ncclSuccess : flush();
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	if (!c || peer < 0 || peer >= c->nranks || !type_bytes(type)) return ncclInvalidArgument;
	g_ops.push_back({ c, Op{ false, buf, count * type_bytes(type), peer, stream } });
	return g_depth ? ncclSuccess : flush();
}

ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	const size_t tb = type_bytes(type);
	if (!c || !(type == ncclInt32 || type == ncclFloat32) || !(op == ncclMin || op == ncclMax || op == ncclSum)) return ncclInvalidArgument;
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
	std::vector<char> mine(count * tb), other;
	if (hipMemcpy(mine.data(), sendbuf, mine.size(), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
	const unsigned long seq = c->ar_seq++;
	char b[64];
	std::snprintf(b, sizeof b, "/ar_%lu_r%d", seq, c->rank);
	if (!write_file(c->dir + b, mine.data(), mine.size())) return ncclSystemError;
	std::vector<char> acc = mine;
	for (int r = 0; r < c->nranks; ++r) {
		if (r == c->rank) continue;
		std::snprintf(b, sizeof b, "/ar_%lu_r%d", seq, r);
		if (!read_file(c->dir + b, other, c, r)) return g_remote_dead ? ncclRemoteError : ncclSystemError;
		if (other.size() != acc.size()) return ncclInvalidUsage;
		if (type == ncclInt32) reduce(reinterpret_cast<int*>(acc.data()), reinterpret_cast<const int*>(other.data()), count, op);
		else reduce(reinterpret_cast<float*>(acc.data()), reinterpret_cast<const float*>(other.data()), count, op);
	}
	if (hipMemcpy(recvbuf, acc.data(), acc.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
	return ncclSuccess;
}

// every rank publishes its `count` elements as <dir>/ag_<seq>_r<rank> and reads the others' in rank order.  A rank that reads
// sequence number s knows every rank has written s, i.e. has finished reading s - 1: it may then remove its own s - 1 file.
ncclResult_t ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
	Comm* c = reinterpret_cast<Comm*>(comm);
	const size_t tb = type_bytes(type);
	if (!c || !tb) return ncclInvalidArgument;
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
	std::vector<char> mine(count * tb), other, all((size_t)c->nranks * count * tb);
	if (hipMemcpy(mine.data(), sendbuf, mine.size(), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
	const unsigned long seq = c->ag_seq++;
	char b[64];
	std::snprintf(b, sizeof b, "/ag_%lu_r%d", seq, c->rank);
	if (!write_file(c->dir + b, mine.data(), mine.size())) return ncclSystemError;
	for (int r = 0; r < c->nranks; ++r) {
		if (r == c->rank) { std::memcpy(all.data() + (size_t)r * mine.size(), mine.data(), mine.size()); continue; }
		std::snprintf(b, sizeof b, "/ag_%lu_r%d", seq, r);
		if (!read_file(c->dir + b, other, c, r)) return g_remote_dead ? ncclRemoteError : ncclSystemError;
		if (other.size() != mine.size()) return ncclInvalidUsage;
		std::memcpy(all.data() + (size_t)r * mine.size(), other.data(), other.size());
	}
	if (seq > 0) { std::snprintf(b, sizeof b, "/ag_%lu_r%d", seq - 1, c->rank); unlink((c->dir + b).c_str()); }
	if (hipMemcpy(recvbuf, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
	return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
	switch (r) {
	case ncclSuccess: return "mock_rccl: success";
	case ncclUnhandledCudaError: return "mock_rccl: HIP call failed";
	case ncclSystemError: return "mock_rccl: rendezvous timed out or file error (ranks disagree on the exchange sequence?)";
	case ncclInvalidArgument: return "mock_rccl: invalid argument";
	case ncclInvalidUsage: return "mock_rccl: send/recv byte counts do not match";
	case ncclRemoteError: return "mock_rccl: a peer rank is gone (its process ended or it aborted its communicator)";
	default: return "mock_rccl: error";
	}
}

}  // extern "C"
