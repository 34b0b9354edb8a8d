// fx_checkpoint.cpp -- whole-grid state files ("field dump format for checkpoints").
//
// The reference has no checkpoint (its state dies with the window); what a later Fluid::Simulate depends on is exactly
// m_velocities[0], m_colors[parity] and m_incompress (Fluid.cpp:360-384: advect reads velocity[0] and colour[!parity'] after
// the parity flip, the Poisson solve warm-starts from the pressure).  One file holds those three fields of the WHOLE grid as
// dense fp32 in the fx_download layouts, so any decomposition can write it (every slab context stores its own planes at
// their offsets: ranks call fx_checkpoint_save on the same path concurrently) and any decomposition can resume from it.
// fp16-storage contexts lose nothing: their stored halves convert to fp32 and back exactly.
//
//   offset 0   char[8]  "FXCKPT03"
//          8   u32 X, Y, Z, storage (fx_storage of the writer, informational)
//         24   u64 steps (simulated steps so far, from fx_get_step_count of the writer; informational)
//         32   u32[8] reserved = 0
//         64   float velocity[3][Z][Y][X] | float colour[Z][Y][X][4] | float pressure[Z][Y][X]
//        end   u64 mark[Z]: steps + 1 of the save that wrote this z plane completely, 0 = incomplete
// A writer first clears the marks of its planes (and syncs), then writes its planes (and syncs), then sets the marks to the save's
// identity: the step count every rank of a chain agrees on without talking (they step together), + 1.  The loader requires every
// plane it reads to carry the identity of the HEADER: a save that was interrupted on any rank, or that one rank of a chain never
// made, leaves planes unmarked -- or, when the path held an older complete save (the periodic-checkpoint case: no O_TRUNC, the
// other slabs write the same file), marked with the OLDER save's identity -- and fx_checkpoint_load refuses them instead of
// resuming from zeros or from a mix of two time steps.  The identity is the step count alone: it tells apart the saves of ONE run at
// different steps (the periodic-checkpoint case).  It does NOT tell apart two runs that reach the same step count with other fields -- a
// restart from an upload, another dt or option set -- writing to one path: a caller who reuses a path across runs removes the file
// first (or gives each run its own).  A context whose
// advection has left its halo (FX_E_HALO pending) writes nothing.  The loader reads all three fields into host memory before it
// touches the context; on a chain it is called by every rank (like the save) and may follow any step.
#include "fx_context.h"

#include <cerrno>
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Header { char magic[8]; uint32_t X, Y, Z, storage; uint64_t steps; uint32_t reserved[8]; };
static_assert(sizeof(Header) == 64, "checkpoint header is 64 bytes");
const char kMagic[8] = { 'F', 'X', 'C', 'K', 'P', 'T', '0', '3' };

bool io_all(int fd, void* buf, size_t n, off_t off, bool write)
{
	char* p = static_cast<char*>(buf);
	while (n) {
		const ssize_t r = write ? pwrite(fd, p, n, off) : pread(fd, p, n, off);
		if (r < 0 && errno == EINTR) continue;
		if (r <= 0) return false;
		p += r; n -= (size_t)r; off += r;
	}
	return true;
}

struct Layout { size_t plane, cells; off_t vel, col, prs, done, end; };
Layout layout(uint32_t X, uint32_t Y, uint32_t Z)
{
	Layout l;
	l.plane = (size_t)X * Y; l.cells = l.plane * Z;
	l.vel = sizeof(Header); l.col = l.vel + (off_t)(3 * l.cells * 4); l.prs = l.col + (off_t)(4 * l.cells * 4); l.done = l.prs + (off_t)(l.cells * 4); l.end = l.done + (off_t)(Z * sizeof(uint64_t));
	return l;
}

}  // namespace

extern "C" {

int fx_checkpoint_save(fx_ctx* ctx, const char* path)
{
	if (!ctx || !path || !path[0]) return FX_E_INVALID;
	if (ctx->desc.flags & FX_FLAG_RENDER_ONLY) return FX_E_STATE;
	const uint32_t X = ctx->desc.grid_x, Y = ctx->desc.grid_y, Z = ctx->desc.grid_z;
	const Layout l = layout(X, Y, Z);
	const size_t z0 = (size_t)ctx->g.z0, nz = (size_t)ctx->g.nz, n = l.plane * nz;
	std::vector<float> vel, col, prs;
	try { vel.resize(3 * n); col.resize(4 * n); prs.resize(n); } catch (const std::bad_alloc&) { return FX_E_NOMEM; }
	// the fields first: a context that cannot hand them out (FX_E_HALO pending, device error) must not touch the file
	int rc;
	if ((rc = fx_download(ctx, FX_FIELD_VELOCITY, vel.data(), 3 * n * 4)) || (rc = fx_download(ctx, FX_FIELD_COLOR, col.data(), 4 * n * 4)) ||
		(rc = fx_download(ctx, FX_FIELD_PRESSURE, prs.data(), n * 4)))
		return rc;
	const int fd = open(path, O_RDWR | O_CREAT, 0644);                      // no O_TRUNC: the other slabs write the same file
	if (fd < 0) { ctx->last_error = std::string("checkpoint: cannot open ") + path + ": " + std::strerror(errno); return FX_E_INVALID; }
	Header h{};
	std::memcpy(h.magic, kMagic, 8);
	h.X = X; h.Y = Y; h.Z = Z; h.storage = ctx->desc.storage; h.steps = ctx->steps_simulated;
	std::vector<uint64_t> marks(nz, 0);
	bool ok = ftruncate(fd, l.end) == 0 && io_all(fd, &h, sizeof h, 0, true) &&    // every writer stores the identical header
		io_all(fd, marks.data(), nz * 8, l.done + (off_t)(z0 * 8), true) && fsync(fd) == 0;  // my planes are incomplete from here on
	for (int a = 0; a < 3 && ok; ++a)
		ok = io_all(fd, vel.data() + a * n, n * 4, l.vel + (off_t)(((size_t)a * l.cells + z0 * l.plane) * 4), true);
	ok = ok && io_all(fd, col.data(), 4 * n * 4, l.col + (off_t)(z0 * l.plane * 16), true);
	ok = ok && io_all(fd, prs.data(), n * 4, l.prs + (off_t)(z0 * l.plane * 4), true);
	ok = ok && fsync(fd) == 0;
	std::fill(marks.begin(), marks.end(), h.steps + 1);                     // the identity of this save
	ok = ok && io_all(fd, marks.data(), nz * 8, l.done + (off_t)(z0 * 8), true) && fsync(fd) == 0;
	close(fd);
	if (!ok) { ctx->last_error = std::string("checkpoint: write failed: ") + std::strerror(errno); return FX_E_INVALID; }
	return FX_OK;
}

int fx_checkpoint_load(fx_ctx* ctx, const char* path)
{
	if (!ctx || !path || !path[0]) return FX_E_INVALID;
	if (ctx->desc.flags & FX_FLAG_RENDER_ONLY) return FX_E_STATE;
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { ctx->last_error = std::string("checkpoint: cannot open ") + path + ": " + std::strerror(errno); return FX_E_INVALID; }
	Header h{};
	struct stat st{};
	bool ok = io_all(fd, &h, sizeof h, 0, false) && fstat(fd, &st) == 0;
	const Layout l = layout(h.X, h.Y, h.Z);
	if (!ok || std::memcmp(h.magic, kMagic, 8) != 0 || h.X != ctx->desc.grid_x || h.Y != ctx->desc.grid_y || h.Z != ctx->desc.grid_z || st.st_size != l.end) {
		close(fd);
		ctx->last_error = "checkpoint: not a FXCKPT03 file of this grid (or truncated)";
		return FX_E_INVALID;
	}
	const size_t z0 = (size_t)ctx->g.z0, nz = (size_t)ctx->g.nz, n = l.plane * nz;
	std::vector<float> vel, col, prs;
	std::vector<uint64_t> marks;
	try { vel.resize(3 * n); col.resize(4 * n); prs.resize(n); marks.resize(nz); } catch (const std::bad_alloc&) { close(fd); return FX_E_NOMEM; }
	ok = io_all(fd, marks.data(), nz * 8, l.done + (off_t)(z0 * 8), false);
	for (size_t i = 0; ok && i < nz; ++i)
		if (marks[i] != h.steps + 1) {
			close(fd);
			ctx->last_error = marks[i] == 0 ? "checkpoint: the file holds incomplete planes (an interrupted save, or a rank of the chain never wrote)"
				: "checkpoint: the file mixes planes of two saves (a rank of the chain did not take part in the last one)";
			return FX_E_INVALID;
		}
	for (int a = 0; a < 3 && ok; ++a)
		ok = io_all(fd, vel.data() + a * n, n * 4, l.vel + (off_t)(((size_t)a * l.cells + z0 * l.plane) * 4), false);
	ok = ok && io_all(fd, col.data(), 4 * n * 4, l.col + (off_t)(z0 * l.plane * 16), false);
	ok = ok && io_all(fd, prs.data(), n * 4, l.prs + (off_t)(z0 * l.plane * 4), false);
	close(fd);
	if (!ok) { ctx->last_error = std::string("checkpoint: read failed: ") + std::strerror(errno); return FX_E_INVALID; }
	// everything is in host memory: only now the context changes.  A load is made by EVERY rank of a chain, so what a single rank's
	// fx_upload must refuse (its neighbours could not know: the measured advection need, the early colour halo) is simply
	// invalidated here -- on all ranks alike, which therefore keep walking the same schedule.
	int rc;
	ctx->collective_upload = true;
	rc = fx_upload(ctx, FX_FIELD_VELOCITY, vel.data(), 3 * n * 4);
	if (!rc) rc = fx_upload(ctx, FX_FIELD_COLOR, col.data(), 4 * n * 4);
	if (!rc) rc = fx_upload(ctx, FX_FIELD_PRESSURE, prs.data(), n * 4);
	ctx->collective_upload = false;
	if (rc) return rc;
	ctx->steps_simulated = h.steps;
	return FX_OK;
}

}  // extern "C"
