// fx_context.h -- the state behind `fx_ctx` (the HIP re-statement of class Fluid's members,
// /root/reference/FluidX12/Content/Fluid.h:79-127): fields in HBM, frame constants, streams,
// timing events and the slab-exchange transport.
#pragma once
#include "fx_internal.h"
#include <vector>
#include <string>

struct fx_comm_group;
struct fx_lane;

struct fx_ctx {
	fx_desc desc;
	fx::Geom g;
	int half;                       // 1 = fp16 velocity/colour storage
	int device;
	hipStream_t stream;             // context-owned stream (used when the caller passes NULL)
	bool owns_stream;

	// ---- fields (XUSG textures of Fluid.h:93-97 -> hipMalloc) --------------------------------
	void* vel[2];                   // 3 component planes each, local extent incl. halo
	void* col[2];                   // rgba texels
	float* p[2];                    // pressure ping-pong (m_incompress); p[p_cur] is current
	int p_cur;
	float* p_face[2];               // slab contexts: scratch levels of the Jacobi face chains (same geometry as p)
	float* b;                       // divergence
	uint8_t* frozen;                // faithful-mode freeze mask (null in fixed mode)
	uint8_t* frozen_alt = nullptr;  // 2-D grids: the tile kernel's output mask (fx_jacobi2d.hip reads one and writes the other; they swap per launch)
	// faithful mode, single-domain fast path (fx_jacobi_freeze.hip): third pressure buffer, two quad-nibble freeze masks, tile marks,
	// and a ring of per-step "last level that left a cell relaxing" words
	float* p_aux = nullptr;
	uint32_t* fz_active_host = nullptr;   // pinned, device-visible: tiles the last dense sweeps left relaxing (written by k_count_marks, read late and without waiting)
	uint32_t* fz_active_dev = nullptr;
	hipEvent_t fz_active_ev = nullptr;    // recorded behind k_count_marks; the count is taken over two solves later, behind this event: the same on every run
	bool fz_active_pending = false;
	int fz_dense_n = 0;                   // masked strip launches per solve in use (0..2; chosen with hysteresis from the count of relaxing tiles, fx_schedule.cpp)
	uint8_t* fz_mask[3] = { nullptr, nullptr, nullptr };   // [2]: the third one of the masked strip launches (three pressure buffers rotate, so do the masks)
	uint32_t* fz_tile_next = nullptr;  // per tile: tag of the solve that listed it last
	void* fz_list[2] = { nullptr, nullptr };   // work lists of alternate launches
	uint32_t* fz_counts = nullptr;  // list lengths of two solves [2][kFreezeSlots launches][8 sub-lists]
	uint32_t* fz_stat = nullptr;
	float* fz_x_p = nullptr;          // slab ranks: the buffers whose face planes the next EX_FREEZE exchange carries
	uint8_t* fz_x_m = nullptr;
	bool fz_fuse_div = false;          // this step's divergence is left to the sparse solver's dense sweep (set by simulate_impl, consumed by jacobi_freeze)
	uint32_t* adv_far = nullptr;       // scratch of the staged advection: the far-tracing voxels it defers (allocated at the first advection)
	bool adv_far_tried = false;
	size_t adv_far_words = 0;
	uint32_t adv_far_turn = 0;        // which of the scratch's two totals the next advection appends through
	uint32_t fz_gen = 0;            // solves so far (tags tile marks and stat words)
	uint32_t fz_gen_mark = 0;       // fz_gen when the timing window opened
	std::vector<uint32_t> fz_iters; // sweep cap of the solve with tag g, at [g % ring]
	uint32_t* lightmap;             // R11G11B10F packed (m_lightMap), owned planes only
	uint8_t* cube;                  // RGBA8 cube map, 5 mips back to back (m_cubeMap)
	size_t cube_mip_offset[5];
	float* env;                     // radiance cube of the sky pass, float [6][env_n][env_n][3] (fx_set_environment)
	uint32_t env_n;
	fx::RenderAccel accel = {};     // scratch of the accelerated ray marches (occupancy grid + masks, alpha side volume, light-voxel list), rebuilt per fx_render
	bool accel_ok = false;          // ... and whether all of it could be allocated
	const void* accel_alpha_of = nullptr;   // the colour buffer whose alpha the side volume holds already (written by the advection that made the field), or null
	bool rendered_since_step = false;       // a context that renders its frames has the NEXT advection write the side volume (one more store per voxel); one that only simulates does not pay for it
	bool lightmap_filled = false;           // the light map holds what a filling build pass (k_build_fill) and its ray kernels left, nothing else has written it since
	float lightmap_key[9] = {};             // ... with these constants behind the unlit value (light colour, ambient, light probe on / off)
	hipStream_t last_step_stream = nullptr;  // the stream the last fx_simulate was driven on (fx_synchronize of an RCCL rank polls it)
	hipStream_t rendered_on = nullptr;      // ... provided that advection runs on the stream the render ran on (the side volume is not double-buffered like the colour)
	int opt_render_accel = 1;       // FX_OPT_RENDER_ACCEL
	uint8_t* target;                // W x H RGBA8 render target of the cube resolve (lazily allocated)
	float* target_float;            // the resolve's SV_TARGET before the output merger (parity tests; lazily allocated)
	float* sh_dev;                  // 27 floats
	bool has_sh;
	unsigned* halo_overflow;        // device flag set when a back-trace leaves the halo
	float* stage;                   // fp32 staging for upload/download conversion
	size_t stage_bytes;
	float* sh_scratch[4];
	size_t sh_scratch_n;

	// ---- frame state (Fluid.h:108-127) --------------------------------------------------------
	uint32_t max_ray_samples, max_light_samples;
	uint32_t ray_samples, cube_lod, visibility_mask;
	uint8_t frame_parity;
	float time_step;
	float edge_pixels;
	bool frame_valid, view_valid;
	fx::FrameConsts fc;

	// ---- timing ---------------------------------------------------------------------------------
	bool timing_on;
	std::vector<hipEvent_t> ev;     // ring of events, 8 per recorded step/render
	size_t ev_used;
	struct Mark { int kind; size_t e0, e1; uint64_t launches, sweeps; };
	std::vector<Mark> marks;
	fx_timing acc;

	// ---- multi-GPU ------------------------------------------------------------------------------
	fx_comm_group* group;           // null = single context
	int rank, nranks;
	int opt_overlap;                // FX_OPT_OVERLAP
	int opt_round;                  // FX_OPT_JACOBI_ROUND
	int opt_count_samples = 0;      // FX_OPT_COUNT_SAMPLES
	unsigned long long* sample_counters = nullptr;   // device, kSampleShards x 3 (allocated when the option is first switched on)
	std::string last_error;
	int col_halo_buf = -1;          // FX_OPT_OVERLAP 3: index of the colour buffer whose halo planes the previous step already exchanged (-1: none)
	uint64_t steps_simulated = 0;   // fx_simulate calls with dt > 0 (recorded in checkpoints)
	// ---- per-step record of a slab rank (ABI 4): { planes the next advection needs below / above the slab, digest of the schedule
	// options, halo-overflow flag }.  Written on the device behind the projection, all-gathered over the chain (RCCL) and copied to
	// pinned host memory; the NEXT fx_simulate waits for it: it sizes the advection exchange per face (FX_OPT_ADAPTIVE_HALO), and
	// an overflow or an option mismatch on ANY rank stops EVERY rank there with the same status.
	int* step_rec = nullptr;        // device, 4 ints
	int* gath_dev = nullptr;        // device, 4 * nranks ints (RCCL groups)
	int* rec_host = nullptr;        // pinned host: 4 * nranks ints (RCCL) or 4 (loop-back member)
	hipEvent_t rec_ev = nullptr;
	bool rec_pending = false;       // the previous step's record is on its way
	bool need_valid = false;        // velocity[0] is what the record was measured on
	float rec_dt = 0.0f;            // time step the needs were measured with (they hold for any smaller one)
	int opt_adaptive = 1;           // FX_OPT_ADAPTIVE_HALO
	int adv_w_lo = 0, adv_w_hi = 0; // planes the current step's advection exchange carries across the lower / upper face
	bool halo_fault = false;        // an overflow was seen and not yet acknowledged by fx_synchronize
	bool collective_upload = false; // fx_checkpoint_load is uploading: every rank of the chain does the same, nothing needs refusing
	bool rec_in_project = false;    // this step's record was written by the projection launch itself (no k_face_need pass)
};

namespace fx {

// one array taking part in a halo exchange: `ncomp` back-to-back sub-arrays (velocity = 3 component
// planes) of nzl planes each, plane_bytes per plane, k boundary planes travelling to each z-neighbour
// recv_base (optional): the halo planes land in another array of the same geometry than the one the face planes leave from
// k_lo / k_hi: planes exchanged with the lower / upper neighbour when they differ from k (the advection halo follows the measured need per face)
struct ExchItem { char* base; size_t plane_bytes; int ncomp; int k; char* recv_base; int k_lo = -1, k_hi = -1; };

// a run of whole planes travelling between z-neighbours: `send` goes to rank + dir, `recv` comes from it.
// Both sides of a pair build their lists from the same items in the same order, so the j-th segment a rank
// sends upwards is the j-th segment its upper neighbour receives from below (RCCL matches send/recv of a
// pair in issue order; the loop-back transport pairs them by the same index).
struct Seg { char* send; char* recv; size_t bytes; int dir; };

// one rank's contribution to a gather onto `root`: `src` lives on rank `rank`, `dst` on the root
struct GatherPart { int rank; const char* src; char* dst; size_t bytes; };

// transport behind a group of slab contexts
// A group (fx_comm_group: its members, lanes, events) must be driven from ONE host thread: the x_ready / x_done event pair of a lane is shared
// by the exchanges issued on its compute, comm and face streams, and it is the order in which that one thread enqueues them that keeps
// a record from overtaking the wait it belongs to.  Nothing in the library locks a group.
struct Transport {
	virtual ~Transport() {}
	// segs[i] = segments of grp->members[i] (RCCL: one member = this rank; loop-back: every rank)
	// streams: the stream the exchange is ordered on, per lane of the group (one entry; a peer group: one per member)
	// channel 1 = a second, independent communicator (RCCL: from the second unique id) for traffic that must not queue behind, or
	// in front of, the step's own exchanges: the early colour halo of FX_OPT_OVERLAP 3.  Operations of ONE communicator are
	// serialised in issue order even across streams.
	virtual int exchange(fx_comm_group* grp, const std::vector<std::vector<Seg>>& segs, const std::vector<hipStream_t>& streams, int channel = 0) = 0;
	// parts[r] for every rank r of the chain (RCCL: only this rank's src and, on the root, every dst are meaningful); `streams` as above:
	// what the members wrote on them is what travels; the copies are ordered on streams[root lane]
	virtual int gather(fx_comm_group* grp, const std::vector<GatherPart>& parts, int root, const std::vector<hipStream_t>& streams) = 0;
	virtual bool is_local() const = 0;
	// min of `v` over the ranks of the chain (RCCL: an all-reduce; loop-back: the caller combines its members itself)
	virtual int min_over_ranks(int v, hipStream_t s, int* out) = 0;
	// every rank contributes `count` device ints, every rank receives all of them in rank order (RCCL: ncclAllGather; the
	// loop-back transport has no use for it: its driver sees every member)
	virtual int allgather(const int* send_dev, int count, int* recv_dev, hipStream_t s) = 0;
	// asynchronous failure of the link layer (RCCL: ncclCommGetAsyncError -- a peer that died, a network error -- polled; on a failure
	// the communicators are aborted with ncclCommAbort, which ends their device kernels, and every later call on this transport returns
	// FX_E_COMM at once).  FX_OK while healthy or where the transport has nothing to poll; *err gets the reason.
	virtual int poll_error(std::string* err) { (void)err; return FX_OK; }
	virtual bool can_poll() const { return false; }
};

}  // namespace fx

// The streams and ordering events of one rank of a slab group.  An RCCL group has one lane (its one member); so has a loop-back
// group of the shared-stream kind, for ALL its members (their phases are ordered by the one compute stream); a loop-back group of
// the peer kind (fx_comm_init_peer: contexts on their own streams, possibly on their own devices) has a lane per member.
struct fx_lane {
	int device;
	hipStream_t compute;            // peer groups: the member's own stream.  Otherwise null: the stream the caller hands to fx_simulate
	// side stream the overlapped exchanges run on, and the two events that order it against the compute stream:
	// ev_ready = "the planes to send are final" (compute -> comm), ev_done = "halos have arrived"
	hipStream_t comm;
	hipEvent_t ev_ready, ev_done;
	hipStream_t face;               // the face chains of the overlapped pressure rounds run on their own stream, beside the interior sweeps
	hipEvent_t ev_col_ready, ev_col_done;   // FX_OPT_OVERLAP 3: "colour of this step is final" (compute -> comm), "its halo planes have arrived" (comm -> compute)
	hipEvent_t ev_int, ev_face1;    // "interior + face copy of the round done" (compute -> face), "the chain has read its input" (face -> compute)
	// peer groups, the handshake of an exchange with the neighbour lanes: x_ready = "what I send is final and my halo may be written",
	// x_done = "I have pulled my halo planes" (a sender must not overwrite them before)
	hipEvent_t x_ready, x_done;
};

struct fx_comm_group {
	std::vector<fx_ctx*> members;   // local transports: every rank; RCCL: just this rank
	fx::Transport* transport;
	int refs;
	std::vector<fx_lane> lanes;     // one, or one per member (per_member)
	bool per_member;
	int min_nz;                     // thinnest slab of the chain: every rank takes the same schedule decisions from it
	hipStream_t shared_stream;      // shared-stream loop-back groups: the one compute stream of all members (owned by the group)
	bool broken;                    // a member was destroyed: the survivors can only be destroyed
	fx_lane& lane_of(const fx_ctx* m);
};

inline fx_lane& fx_comm_group::lane_of(const fx_ctx* m) { return per_member ? lanes[(size_t)m->rank] : lanes[0]; }

namespace fx {
enum ExchSet {
	EX_ADVECT_IN = 0,   // what advection gathers from: velocity[0] (3 components) and colour[!parity]
	EX_UZ1 = 1,         // z-component of the advected velocity: all the divergence reads across a slab face
	EX_DIV = 2,         // divergence b
	EX_PRESSURE = 3,    // pressure buffer `pidx` (+ the freeze mask in faithful mode)
	EX_PRESSURE_FACE = 4, // face planes leave from scratch buffer p_face[pidx >> 1], halos land in pressure buffer pidx & 1
	EX_ADVECT_VEL = 5,   // velocity[0] only: the colour the advection gathers from was exchanged behind the previous step's pressure phase
	EX_COLOR_CUR = 6,    // colour[parity]: what the NEXT step's advection gathers from
	EX_FREEZE = 7        // the sparse faithful solver's current pressure buffer + quad-nibble mask (fx_ctx::fz_x_p / fz_x_m)
};
// items of one member for an exchange set; returns their number (<= 4)
int exchange_items(fx_ctx* c, int which_set, int k, int pidx, ExchItem out[4]);
// append the segments of `items` for the neighbours this member has
void halo_segments(const fx_ctx* c, const ExchItem* items, int n, std::vector<Seg>& out);
Transport* make_local_transport();
Transport* make_rccl_transport(const void* id, size_t bytes, int rank, int nranks, int device, std::string* err);
size_t rccl_id_bytes();
int rccl_get_unique_id(void* out, size_t bytes, std::string* err);
}  // namespace fx
