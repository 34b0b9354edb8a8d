/* fluidx_hip.h -- C ABI of the MI355X-native smoke solver + cube-map-space ray marcher.
 *
 * Drop-in boundary for the reference's `class Fluid` operator (StarsX/FluidX12,
 * FluidX12/Content/Fluid.h:20-35) and the SH side of `LightProbe` (Content/LightProbe.h:16-26):
 * every entry point below names the reference interface it replaces.  Plain C types only
 * (pointers + sizes); `void* stream` is a `hipStream_t` (the reference's `XUSG::CommandList*`
 * becomes the HIP stream work is enqueued on; NULL = the context's own stream).
 *
 * All functions return 0 (FX_OK) or a negative FX_E_* code and never throw across the ABI.
 * A context is not thread-safe (single caller thread, like the reference's UI thread).
 * There is no CPU fallback: without a HIP device fx_create fails with FX_E_DEVICE.
 */
#ifndef FLUIDX_HIP_H
#define FLUIDX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_ABI_VERSION 7   /* 7: fx_field_digest, fx_last_error; 6: fx_comm_init_peer, fx_set_knob / fx_knob_name, FX_OPT_RENDER_ACCEL, fx_timing.freeze_strip_launches; the launcher switches no longer come from FLUIDX_* environment variables */

enum fx_status {
	FX_OK = 0,
	FX_E_INVALID = -1,      /* bad argument / unsupported combination          */
	FX_E_DEVICE = -2,       /* HIP runtime error (no device, launch failure)   */
	FX_E_NOMEM = -3,        /* device or host allocation failed                */
	FX_E_STATE = -4,        /* call order violation (e.g. render before update) */
	FX_E_COMM = -5,         /* RCCL error / library unavailable                */
	FX_E_HALO = -6          /* back-trace left the exchanged halo (multi-GPU)  */
};

/* Fluid::RenderFlags (Fluid.h:12-18) */
enum fx_render_flags {
	FX_RAY_MARCH_DIRECT = 0,
	FX_RAY_MARCH_CUBEMAP = 1,
	FX_SEPARATE_LIGHT_PASS = 2,
	FX_OPTIMIZED = 3
};
#define FX_FRAME_COUNT 3            /* Fluid::FrameCount (Fluid.h:35) */

enum fx_storage { FX_STORAGE_FP32 = 0, FX_STORAGE_FP16 = 1 };   /* velocity/colour texel storage; the
                                                                  reference is RGBA16F (Fluid.cpp:207,213) */
enum fx_jacobi_mode { FX_JACOBI_FIXED = 0,      /* N lock-step sweeps, no early-out (BASELINE configs)     */
                      FX_JACOBI_FAITHFUL = 1 }; /* cap N (reference: 64) + per-cell freeze at |dx| < 1e-3
                                                   (CSPoisson.hlsli:11,24)                                  */
enum fx_address { FX_ADDRESS_CLAMP = 0,         /* FluidEZ.cpp:406 (default path of the reference)          */
                  FX_ADDRESS_MIRROR = 1 };      /* Fluid.cpp:452                                            */

/* fields for fx_upload / fx_download; host layouts are dense fp32 regardless of device storage:
 *   VELOCITY  float[3][Z][Y][X]  (component planes; the texture advect reads = m_velocities[0])
 *   VELOCITY1 float[3][Z][Y][X]  (m_velocities[1], advect output / project input)
 *   COLOR     float[Z][Y][X][4]  (m_colors[parity], the one every renderer reads)
 *   COLOR_PREV float[Z][Y][X][4] (m_colors[!parity])
 *   PRESSURE  float[Z][Y][X]     (m_incompress)      DIVERGENCE float[Z][Y][X] (scratch b)
 *   LIGHTMAP  float[Z][Y][X][3]  (m_lightMap decoded from R11G11B10F)
 *   CUBEMAP   uint8[6][S][S][4]  (mip `lod` of m_cubeMap, S = X >> lod, R8G8B8A8_UNORM)
 * Z = the context's own slab (slab_nz planes), never the halo. */
enum fx_field {
	FX_FIELD_VELOCITY = 0, FX_FIELD_VELOCITY1 = 1, FX_FIELD_COLOR = 2, FX_FIELD_COLOR_PREV = 3,
	FX_FIELD_PRESSURE = 4, FX_FIELD_DIVERGENCE = 5, FX_FIELD_LIGHTMAP = 6, FX_FIELD_CUBEMAP = 7,
	FX_FIELD_TARGET = 8,        /* render target of fx_render_cube: uint8 [viewport_h][viewport_w][4] (download only) */
	FX_FIELD_TARGET_FLOAT = 9   /* the resolve's output before the blend: float [h][w][4], zeros where discarded     */
};

typedef struct fx_ctx fx_ctx;

/* replaces the arguments of Fluid::Init (Fluid.cpp:189-270) */
typedef struct fx_desc {
	uint32_t struct_size;       /* = sizeof(fx_desc)                                        */
	uint32_t grid_x, grid_y, grid_z;    /* global grid; grid_x == grid_y (Fluid.cpp:201); grid_z == 1 -> 2D */
	uint32_t viewport_w, viewport_h;    /* Init(width, height); 0 x 0 = a context that only simulates (fx_render: FX_E_INVALID) */
	uint32_t storage;           /* fx_storage                                               */
	uint32_t jacobi_iters;      /* N (BASELINE: 20/40/80; reference faithful: 64)           */
	uint32_t jacobi_mode;       /* fx_jacobi_mode                                           */
	uint32_t advect_address;    /* fx_address                                               */
	int32_t  device;            /* HIP device ordinal, -1 = current                         */
	uint32_t slab_z0, slab_nz;  /* z-slab owned by this context; {0, 0} = whole grid        */
	uint32_t halo_advect;       /* planes allocated (and at most exchanged) per face for the advection (0 = default 6) */
	uint32_t halo_jacobi;       /* sweeps per pressure halo exchange (0 = default)          */
	uint32_t flags;             /* FX_FLAG_*                                                */
} fx_desc;

/* fx_desc.flags: bits 0-3 = Jacobi sweeps fused per launch (temporal blocking), 0 = library default,
 * 1 = one launch per sweep; results are bit-identical for every setting */
#define FX_FLAG_JACOBI_FUSE_MASK 0xFu
/* multi-rank contexts: keep every halo exchange on the compute stream (no side comm stream, no face-first
 * ordering); results are bit-identical either way -- the switch exists to measure what the overlap buys */
#define FX_FLAG_NO_OVERLAP 0x10u
/* a context that only renders: one colour buffer + light map / cube map / target, no velocity, pressure or divergence.
 * It receives its colour from fx_upload or fx_comm_gather_color; fx_simulate and the stage calls return FX_E_STATE. */
#define FX_FLAG_RENDER_ONLY 0x20u

/* values Fluid::UpdateFrame derives (Fluid.cpp:324-333) */
typedef struct fx_frame_info {
	uint32_t cube_lod;          /* m_cubeMapLOD                         */
	uint32_t cube_size;         /* grid_x >> cube_lod                   */
	uint32_t ray_samples;       /* m_raySampleCount                     */
	uint32_t visibility_mask;   /* m_visibilityMask                     */
	uint32_t frame_parity;      /* m_frameParity                        */
	float    edge_pixels;       /* EstimateCubeEdgePixelSize            */
	float    time_step;
	float    world_view_proj_i[16];  /* CBPerObject.WorldViewProjI as its 4 constant-buffer rows (Fluid.cpp:318; ABI 2) */
	float    screen_to_world[16];    /* LightProbe CBPerFrame.ScreenToWorld rows (LightProbe.cpp:70-76; ABI 2)        */
} fx_frame_info;

/* HIP-event timings accumulated by fx_simulate / fx_render while enabled (milliseconds, launch counts) */
typedef struct fx_timing {
	double   advect_ms, divergence_ms, jacobi_ms, project_ms, light_ms, view_ms, exchange_ms;
	uint64_t steps, jacobi_launches, jacobi_sweeps, renders;
	double   resolve_ms;        /* fx_render_cube (ABI 2) */
	/* ABI 3: when a step mixes launch shapes (40 sweeps = 12 launches of three + 2 of two), the launches with the most
	 * sweeps each -- the dominant kernel -- are also booked on their own; otherwise these equal the jacobi_* totals */
	double   jacobi_main_ms;
	uint64_t jacobi_main_launches, jacobi_main_sweeps;
	/* ABI 4 (slab ranks): bytes this rank SENT in halo exchanges, and planes its advection exchange carried across its lower +
	 * upper face (summed over the timed steps; halo_advect per face without FX_OPT_ADAPTIVE_HALO) */
	uint64_t exchange_bytes, advect_halo_planes;
	double   chain_ms;          /* face chains of the overlapped pressure rounds (their own stream, beside the interior sweeps) */
	/* ABI 5 (FX_JACOBI_FAITHFUL, single-domain contexts): pressure solves since the last reset, and the sweeps the reference's
	 * loop (CSPoisson.hlsli:11-25) executes in them -- per solve 1 + the last sweep that left a cell relaxing, at most N.
	 * jacobi_sweeps above counts the levels ENQUEUED (always N: the sparse solver needs no read-back to stop early). */
	uint64_t freeze_solves, freeze_sweeps;
	uint64_t exchange_calls;    /* slab ranks: halo exchanges issued (one RCCL group call each) over the timed steps */
	/* FX_OPT_COUNT_SAMPLES: trilinear colour fetches of the view rays, density fetches of the light / AO rays (the light pass's one
	 * fetch per voxel included), light-map fetches -- summed over the renders since the last reset */
	uint64_t view_samples, light_samples, lightmap_fetches;
	/* ABI 6 (FX_JACOBI_FAITHFUL, single domain, X = 256): launches of the masked strip pipeline among jacobi_launches -- which of the two
	 * launch sequences the solves took (decided per solve from a tile count measured two solves earlier: a function of the step sequence) */
	uint64_t freeze_strip_launches;
} fx_timing;

int fx_abi_version(void);
const char* fx_error_string(int status);

/* Fluid::Fluid + Fluid::Init (Fluid.cpp:168-270): allocates all fields zero-initialised */
int fx_create(fx_ctx** out, const fx_desc* desc);
/* Fluid::~Fluid */
int fx_destroy(fx_ctx* ctx);
/* Fluid::SetMaxSamples (Fluid.cpp:272-276) */
int fx_set_max_samples(fx_ctx* ctx, uint32_t max_ray_samples, uint32_t max_light_samples);
/* Fluid::SetSH (Fluid.cpp:278-281): 9 x float3 host coefficients, NULL detaches the probe */
int fx_set_sh(fx_ctx* ctx, const float* coeffs27);
/* Fluid::UpdateFrame (Fluid.cpp:283-346); matrices row-major, row-vector convention (DirectXMath).
 * view/proj/eye may be NULL for a pure simulation context (no rendering state is updated). */
int fx_update_frame(fx_ctx* ctx, float time_step, uint8_t frame_index,
	const float view[16], const float proj[16], const float eye[3]);
/* Fluid::Simulate (Fluid.cpp:348-410): enqueues advect + divergence + N sweeps + project.  Asynchronous like the reference's call, with one
 * exception: with FX_JACOBI_FAITHFUL on a single 256-wide domain every fourth solve takes over a tile count that a launch two solves earlier
 * left in host-visible memory and waits for THAT launch's event first (hipEventSynchronize; long past in practice) -- so the host runs at
 * most about two steps ahead of the device there, and such a context must not be stepped inside a stream capture.  The wait is what makes
 * the launch sequence a function of (the last uploaded state, the steps since) instead of when the host happened to look; fx_upload of a
 * simulation field starts that adaptation over. */
int fx_simulate(fx_ctx* ctx, void* stream, uint8_t frame_index);
/* Fluid::Render (Fluid.cpp:412-446), all four flag combinations:
 *   flags & FX_RAY_MARCH_CUBEMAP   cube-map-space march (merged, or light volume + view pass with FX_SEPARATE_LIGHT_PASS):
 *                                  writes the cube map (and light map); fx_render_cube below puts it on the screen
 *   otherwise                      direct screen-space march, one ray per pixel of the viewport (PSRayCast / PSRayCastV,
 *                                  Fluid.cpp:932-972), blended straight into the render target (PREMULTIPLIED) */
/* (FX_E_INVALID also for a 3-D grid with grid_y * (grid_z + 1) >= 2^24 or 2^32 voxels: tap indices are 32 bits wide.) */
int fx_render(fx_ctx* ctx, void* stream, uint8_t frame_index, uint8_t flags);
/* The caller-side half of the cube path (row f-1 of SURVEY.md 8): the render target the reference's caller binds.
 * fx_clear_render_target = ClearRenderTargetView (FluidX12.cpp:471-472; the demo clears to (0.2, 0.2, 0.2, 0));
 * fx_render_cube = Fluid::renderCube (Fluid.cpp:910-931) in its raster-free per-pixel form (PSRayCastCube.hlsl):
 * resolves mip `cube_lod` of the cube map the last fx_render wrote onto the viewport_w x viewport_h RGBA8 target with
 * the PREMULTIPLIED blend (Fluid.cpp:653).  Read the result with fx_download(FX_FIELD_TARGET). */
int fx_clear_render_target(fx_ctx* ctx, void* stream, const float rgba[4]);
int fx_render_cube(fx_ctx* ctx, void* stream, uint8_t frame_index);

/* The light probe's sky pass (LightProbe::RenderEnvironment, LightProbe.cpp:85-97; PSEnvironment.hlsl), which the demo draws
 * before the volume (FluidX12.cpp:483): fx_set_environment keeps a copy of the radiance cube float[6][n][n][3] on the device
 * (NULL releases it), fx_render_environment writes it onto the render target as seen by the camera of the last
 * fx_update_frame (no blending: rgb, alpha 0). */
int fx_set_environment(fx_ctx* ctx, const float* cube, uint32_t n);
int fx_render_environment(fx_ctx* ctx, void* stream, uint8_t frame_index);

int fx_get_frame_info(fx_ctx* ctx, fx_frame_info* out);

/* blocks until everything enqueued by this context has finished; reports FX_E_HALO if the advection back-trace of a multi-GPU
 * step left the exchanged halo ON THIS RANK, and thereby acknowledges it.  Until then fx_download of a simulation field,
 * fx_checkpoint_save and fx_comm_gather_color of this rank return FX_E_HALO as well.  Independently the fault travels with the
 * per-step record: the next fx_simulate returns FX_E_HALO on EVERY rank of the chain, once, without stepping (its inputs are
 * untouched) -- the chain-wide notice, whichever of the two calls comes first; the fx_simulate after that starts clean on every
 * rank, with or without an fx_synchronize in between (the notice takes the faulting rank's device flag down and keeps the fault on
 * the host: that rank's read-back and checkpoints go on refusing until its fx_synchronize).  No rank runs on, or stores, fields that
 * differ from the single-domain run without having been told. */
int fx_synchronize(fx_ctx* ctx);
/* Also FX_E_DEVICE from fx_synchronize: a strip kernel's LDS hand-over wait ran out (a protocol error or a lost wave; the waits are bounded,
 * ~10 ms) -- the pressure field that launch wrote is not the solver's; the report clears the fault.  fx_last_error: what the last failed
 * call of this context had to say beyond its status (a HIP error string, which kernel family timed out, ...); "" if nothing; the pointer is
 * valid until the context's next call. */
const char* fx_last_error(fx_ctx* ctx);

/* checkpoint / parity access (no reference counterpart; the reference cannot read fields back) */
int fx_upload(fx_ctx* ctx, int field, const void* host, size_t bytes);
int fx_download(fx_ctx* ctx, int field, void* host, size_t bytes);
size_t fx_field_bytes(fx_ctx* ctx, int field);
/* A 128-bit digest of global planes [z_begin, z_begin + z_count) of a simulation field (VELOCITY .. DIVERGENCE), computed ON THE DEVICE
 * from the stored bits (fp16 storage: the fp16 bits) and each element's GLOBAL position: a sum over the elements of a 64-bit mix of
 * (bits, global element index), twice with different seeds.  The sum does not depend on how the planes are spread over contexts, so
 * a slab rank's digest of its owned planes equals the single-domain context's digest of the same planes if and only if (to 2^-128)
 * the fields agree bit for bit -- what `bench.py --gpus N` certifies its timed steps with, without reading 0.5 GB per rank back.
 * The planes must be owned by this context (FX_E_INVALID otherwise); z_count = 0 = all owned planes from z_begin = the first.
 * Blocks like fx_download; FX_E_HALO while a halo fault is pending. */
int fx_field_digest(fx_ctx* ctx, int field, uint32_t z_begin, uint32_t z_count, uint64_t out[2]);

/* Checkpoint / resume (the reference keeps no state across runs).  One file holds what a later
 * fx_simulate depends on -- velocity[0], colour[parity], pressure (Fluid.cpp:360-384) -- for the WHOLE grid, dense fp32 in the
 * layouts above behind a 64-byte header ("FXCKPT01", X, Y, Z, storage, step count).  Every slab context stores / loads its own
 * planes at their offsets, so the ranks of a chain call these on the same path and a run may resume under another
 * decomposition.  fp16-storage contexts round-trip exactly.  Resuming continues bit-identically to the uninterrupted run. */
int fx_checkpoint_save(fx_ctx* ctx, const char* path);
int fx_checkpoint_load(fx_ctx* ctx, const char* path);

/* individual stages of Simulate, exposed for per-kernel parity tests and micro-benchmarks */
int fx_advect(fx_ctx* ctx, void* stream);
int fx_divergence(fx_ctx* ctx, void* stream);
int fx_jacobi(fx_ctx* ctx, void* stream, uint32_t iters);
int fx_project(fx_ctx* ctx, void* stream);

/* LightProbe::TransformSH + GetSH (LightProbe.h:22,26; LightProbeEZ.cpp:117-123,183-278):
 * order-3 SH of a radiance cube float[6][N][N][3] (host), coefficients to out27 (host) */
int fx_sh_transform(fx_ctx* ctx, const float* cube, uint32_t n, float* out27);

/* LightProbe::Init's asset path (LightProbe.cpp:41-46 hands a DDS file to XUSG's loader, row f-4): a DDS cube map to linear float
 * radiance.  Accepted: DX10-header files in DXGI_FORMAT_BC6H_UF16 (the reference's Bin/Assets/rnl_cross.dds; decoded on the device),
 * R32G32B32A32_FLOAT, R32G32B32_FLOAT, R16G16B16A16_FLOAT, R8G8B8A8_UNORM, and legacy headers with FourCC 113 / 116
 * (D3DFMT_A16B16G16R16F / A32B32G32R32F); every face carries its whole mip chain.  fx_dds_cube_info: edge of mip 0 and mip count
 * (FX_E_INVALID for any other container / format); fx_dds_decode_cube: mip `mip` of all six faces (order +X -X +Y -Y +Z -Z) as
 * out_cube float[6][n][n][3] (n = max(size >> mip, 1); out_floats must be 18 n^2; alpha is dropped). */
int fx_dds_cube_info(const void* dds, size_t bytes, uint32_t* size, uint32_t* mips);
int fx_dds_decode_cube(fx_ctx* ctx, const void* dds, size_t bytes, uint32_t mip, float* out_cube, size_t out_floats);

/* timing */
int fx_timing_enable(fx_ctx* ctx, int enable);
int fx_timing_read(fx_ctx* ctx, fx_timing* out, int reset);

/* ---- multi-GPU z-slabs (no reference counterpart) --------------------------------
 * One context per rank.  RCCL transport: fx_comm_id_bytes/fx_comm_get_unique_id on rank 0, broadcast
 * the bytes out of band, fx_comm_init_rank on every rank (rank r owns slab r).  In-process groups (one process, several slab
 * contexts, rank 0 drives them all): fx_comm_init_local -- one device, every member on ONE compute stream (the decomposition's
 * arithmetic on a 1-GPU box) -- and fx_comm_init_peer -- every member keeps its own streams and may live on its own device
 * (fx_desc.device): a rank pulls its halo planes straight out of its neighbour's memory (hipDeviceEnablePeerAccess +
 * hipMemcpyPeerAsync; no IPC handles, no RCCL), the members run concurrently, ordered by events per exchange.  With a peer group
 * fx_simulate ignores its stream argument: each member's work goes to the member's own stream.
 * The id is TWO ncclUniqueIds (256 bytes): the communicator of the step's exchanges and a second one for traffic that must not
 * queue with them (FX_OPT_OVERLAP 3).  Passing only the first 128 bytes gives one communicator serving both. */
size_t fx_comm_id_bytes(void);
int fx_comm_get_unique_id(void* id_out, size_t bytes);
int fx_comm_init_rank(fx_ctx* ctx, const void* id, size_t bytes, int rank, int nranks);
int fx_comm_init_local(fx_ctx** ctxs, int nranks);
int fx_comm_init_peer(fx_ctx** ctxs, int nranks);

/* Multi-GPU rendering (row f-3 of SURVEY.md 8), the exact way: rays cross slabs, so the colour field is gathered.  Every
 * rank of the slab group sends its owned planes of colour[parity] to rank `root`, where they land in `full`, a whole-grid
 * context on the root's device (same grid and storage; FX_FLAG_RENDER_ONLY keeps it small) that then runs
 * fx_update_frame(dt = 0, camera) + fx_render like any single-GPU context -- the picture is bit-identical to the
 * single-domain one.  Call on every rank; `full` is read on the root only (NULL elsewhere).  slab_z0 / slab_nz list the
 * nranks slabs (the root places the planes by them; a loop-back group knows its members and accepts NULL). */
int fx_comm_gather_color(fx_ctx* ctx, void* stream, fx_ctx* full, int root, const uint32_t* slab_z0, const uint32_t* slab_nz);

/* How the slab schedule hides the exchanges; results are bit-identical for every setting, and every rank of a
 * group must use the same values (a loop-back group reads those of its first context).
 *   FX_OPT_OVERLAP       0 = every exchange on the compute stream;
 *                        1 = the advection halo travels on a side stream behind the interior advection;
 *                        2 = (default) additionally each pressure exchange travels behind the interior sweeps
 *                            of its round (the face planes are swept first)
 *                        3 = additionally the colour half of the NEXT step's advection halo (4 of its 7 plane-units)
 *                            leaves as soon as this step's advection has written it and travels behind the pressure
 *                            phase; the next step only exchanges the velocity.  While such a halo is out, fx_upload of a
 *                            colour field into an RCCL rank returns FX_E_STATE (its neighbours could not know; a
 *                            loop-back group simply exchanges the colour again)
 *   FX_OPT_JACOBI_ROUND  sweeps per pressure exchange, 1 .. fx_desc.halo_jacobi (default = halo_jacobi)
 *   FX_OPT_ADAPTIVE_HALO 1 = (default) the advection exchange carries, per slab face, exactly the planes the coming advection
 *                        will touch: behind every projection a kernel measures them on the velocity just written (the z taps
 *                        of every voxel's back-trace, computed with the advection's own arithmetic), the ranks all-gather the
 *                        two numbers, and the next step exchanges max(need of the two slabs sharing the face) planes instead of
 *                        fx_desc.halo_advect -- which remains the allocation and the limit: a need beyond it stops the step with
 *                        FX_E_HALO on EVERY rank before a field is touched.  The measurement holds for time steps up to the one
 *                        it was taken with; a larger one, the first step, or a velocity upload fall back to halo_advect planes.
 *                        0 = always halo_advect planes.
 *   FX_OPT_COUNT_SAMPLES 1 = fx_render counts the samples its marches take (fx_timing.view_samples / light_samples / lightmap_fetches): every
 *                        thread adds its counts with atomics, so a counted render is for statistics, not for timing.  Local to the context.
 *   FX_OPT_RENDER_ACCEL  1 = (default) fx_render runs the accelerated marches: occupancy masks of the colour field held in the LDS, an
 *                        alpha-only side volume for the density taps, the lit light-map voxels compacted into a list.  0 = the plain
 *                        kernels, where every sample gathers its taps like the reference's shaders.  Bit-identical pictures either
 *                        way (the accelerated path only skips fetches whose result is known); local to the context.  Grids of more
 *                        than 2^28 voxels always take the plain kernels (the accelerated ones address by 32-bit byte offsets).  While
 *                        it is on, a context that renders its frames has the advection of the NEXT fx_simulate -- when that runs on
 *                        the stream the render ran on -- store the side volume along with the colour field (one more 4-byte store
 *                        per voxel), so that the render does not read the field a second time to extract it.
 * On an RCCL chain fx_set_option (of the three schedule options) is COLLECTIVE: every rank calls it with the same arguments between two steps; the values are
 * compared across the chain and a disagreement returns FX_E_INVALID everywhere with nothing changed.  While FX_OPT_ADAPTIVE_HALO
 * is on and a step has run, fx_upload(FX_FIELD_VELOCITY) into an RCCL rank returns FX_E_STATE (its neighbours have sized the next
 * exchange from the old field); switch the option off first. */
enum fx_option { FX_OPT_OVERLAP = 1, FX_OPT_JACOBI_ROUND = 2, FX_OPT_ADAPTIVE_HALO = 3, FX_OPT_COUNT_SAMPLES = 4, FX_OPT_RENDER_ACCEL = 5 };
int fx_set_option(fx_ctx* ctx, uint32_t option, uint32_t value);

/* Measurement switches of the kernel launchers (no reference counterpart; docs/LAB.md lists them): which kernel serves a geometry,
 * chunk sizes, tile orders -- A/B runs and the parity tests that pit one kernel of the library against another.  Process-wide,
 * none changes a result.  name: e.g. "ADVECT_LDS", "JACOBI_T" (fx_knob_name enumerates them, NULL behind the last); value: the
 * text the switch parses, NULL = back to the default.  FX_E_INVALID for an unknown name.  The library reads no environment variable
 * for them (the one it reads: FLUIDX_RCCL_LIB, the path of librccl.so). */
int fx_set_knob(const char* name, const char* value);
const char* fx_knob_name(uint32_t index);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDX_HIP_H */
