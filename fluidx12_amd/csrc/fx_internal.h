// fx_internal.h -- context layout and kernel launcher declarations shared by the C-ABI
// implementation (fx_api.cpp) and the HIP kernels (fx_sim.hip, fx_render.hip, fx_sh.hip).
// Product code: never includes or links anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/fluidx_hip.h"
#include "fx_knobs.h"

namespace fx {

// Geometry of one z-slab as the kernels see it.  Local plane l <-> global z = z0 - H + l.
struct Geom {
	int X, Y;        // plane extent
	int Zg;          // global depth
	int z0;          // first owned global plane
	int nz;          // owned planes
	int H;           // allocated halo planes on each side (0 for a single-GPU context)
	int zlo, zhi;    // inclusive range of global planes whose data is present locally
	__host__ __device__ size_t plane() const { return (size_t)X * Y; }
	__host__ __device__ int nzl() const { return nz + 2 * H; }
	__host__ __device__ size_t cells_local() const { return plane() * (size_t)nzl(); }
	__host__ __device__ size_t cells_owned() const { return plane() * (size_t)nz; }
	// local plane index of global plane z (must be present)
	__host__ __device__ int lz(int z) const { return z - z0 + H; }
};

// frame constants of the ray-march kernels (Common.hlsli:15-30 cbPerObject/cbPerFrame)
struct FrameConsts {
	float world_i[12];     // XMFLOAT3X4: rows of the transposed inverse world
	float world[12];
	float eye_pt[3];
	float light_pt[3];
	float light_color[4];
	float ambient[4];
	float wvp_i[16];       // CBPerObject.WorldViewProjI as its four constant-buffer rows (Fluid.cpp:318)
	float s2w[16];         // LightProbe's ScreenToWorld = transpose(inverse(view * proj)) rows (LightProbe.cpp:70-76)
};

struct SimParams {
	float dt;
	int address;           // fx_address
	int is3d;
};

// ---- simulation launchers (fx_sim.hip); `half_store` selects __half storage of velocity/colour
// z_begin/z_end: global plane range to compute (within the locally present range)
// alpha: when the launch writes every voxel of an unsliced grid on the staged path it also writes the stored alpha, as fp32, to `out`
// (the render's side volume, fx_render_accel.hip) and says so in `written`
struct AdvectAlpha { float* out; bool written; };
hipError_t launch_advect(const Geom& g, const SimParams& sp, int half_store, const void* vel_in, const void* col_in,
	void* vel_out, void* col_out, int z_begin, int z_end, unsigned* halo_overflow, hipStream_t s, uint32_t* far_scratch = nullptr, size_t far_words = 0, int far_parity = 0, bool* far_used = nullptr,
	AdvectAlpha* alpha = nullptr);
// LDS-staged variant (fx_advect_lds.hip); hipErrorNotSupported when the geometry has no such path (force: also below the size where it pays)
hipError_t launch_advect_lds(const Geom& g, const SimParams& sp, int half_store, const void* vel_in, const void* col_in,
	void* vel_out, void* col_out, int z_begin, int z_end, unsigned* halo_overflow, uint32_t* far_scratch, size_t far_words, int far_parity, bool* far_used, hipStream_t s, bool force,
	AdvectAlpha* alpha = nullptr);
// far_scratch: advect_far_words(g, planes) words lent by the caller (its first two words ZEROED once; far_parity alternates over the launches that report *far_used) let the staged kernel
// defer far-tracing voxels to a second, small launch
size_t advect_far_words(const Geom& g, int nzp);
hipError_t launch_divergence(const Geom& g, int half_store, const void* vel, float* b, int z_begin, int z_end, hipStream_t s);
// one lock-step sweep p_in -> p_out on planes [z_begin, z_end); frozen may be null
hipError_t launch_jacobi_sweep(const Geom& g, const float* p_in, const float* b, float* p_out, uint8_t* frozen,
	int z_begin, int z_end, hipStream_t s);
// the same sweep over two disjoint plane ranges in one launch (the two face zones of a slab)
hipError_t launch_jacobi_sweep2(const Geom& g, const float* p_in, const float* b, float* p_out, uint8_t* frozen,
	int z_begin, int z_end, int z_begin2, int z_end2, hipStream_t s);
// `sweeps` lock-step sweeps fused in one launch (temporal blocking); result in p_out.  Planes [z_begin, z_end)
// of p_out are valid afterwards provided p_in/b are valid on [z_begin - sweeps, z_end + sweeps) (or the
// global boundary).  Returns hipErrorNotSupported when the geometry has no fused path.
hipError_t launch_jacobi_fused(const Geom& g, const float* p_in, const float* b, float* p_out, int sweeps,
	int z_begin, int z_end, hipStream_t s);
// two or three sweeps per launch, register-resident strips (fx_jacobi_strip.hip)
bool jacobi_strip_supported(const Geom& g);
bool jacobi_strip_wide(const Geom& g);       // X = 512: only the two-sweep wide kernel exists
hipError_t launch_jacobi_strip(const Geom& g, const float* p_in, const float* b, float* p_out, int sweeps, int z_begin, int z_end, hipStream_t s);
// three sweeps per launch, register strips + the LDS as a second register file (fx_jacobi_strip3.hip; X = 256)
bool jacobi_strip3_supported(const Geom& g);
hipError_t launch_jacobi_strip3(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s);
// four sweeps per launch, a workgroup's four waves as a quad over 16 rows (fx_jacobi_strip4.hip; X = 256, Y % 16 == 0)
bool jacobi_strip4_supported(const Geom& g);
hipError_t launch_jacobi_strip4(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s);
// two sweeps per launch, one 4 x 4-row block per wave (fx_jacobi_block.hip; X = 128)
bool jacobi_block2_supported(const Geom& g);
hipError_t launch_jacobi_block2(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s);
// the same scheme for any X = CPL x (<= 64 lanes), CPL <= 4: rows that are no multiple of four cells (150^3, the reference's GI preset)
bool jacobi_blockg_supported(const Geom& g);
hipError_t launch_jacobi_blockg(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s);
// the reference's own solve (cap N + per-cell early-out) as a sparse solver (fx_jacobi_freeze.hip): a dense first sweep that writes
// level 1 to TWO buffers and marks the 32 x 8 x 8 tiles that still relax, then launches of <= 4 levels over the marked tiles only.
// stat: device word raised to stat_hi | level for every level that leaves a cell relaxing.
bool jacobi_freeze_supported(const Geom& g);
int jacobi_freeze_tiles(const Geom& g);
size_t jacobi_freeze_mask_bytes(const Geom& g);
size_t jacobi_freeze_list_bytes(const Geom& g);     // one work list (eight per-XCD sub-lists of { tile, box } entries)
const int kFreezeSlots = 128;                       // launches a solve may enqueue
size_t jacobi_freeze_count_words();                 // list-length counters of one solve: kFreezeSlots launches x 8 sub-lists
int jacobi_freeze_levels_per_launch();
// the device-side work state of one solve: tile_mark[tile] == gen <=> the tile is on the first list already; the dense sweep fills
// list[0] and counts[0][8] (zero on entry); tile launch n reads list[n & 1] / counts[n] and appends its surviving tiles to
// list[(n + 1) & 1] / counts[n + 1]; counts_next = the next solve's counters, cleared by this solve's dense launch
struct FreezeWork { uint32_t* tile_mark; uint32_t gen; void* list[2]; int cap; uint32_t* counts; uint32_t* counts_next; };
// a slab rank runs the solver on a view of the planes it holds (jacobi_freeze_view): `g` = the view, the field pointers advanced to its plane 0,
// [z_begin, z_begin + nzp) = the owned planes in view coordinates (nzp = 0: all), vel_comp_cells = cells between velocity components (0: the view's)
Geom jacobi_freeze_view(const Geom& g, int* first_plane, int* own_begin);
hipError_t launch_freeze_dense(const Geom& g, const float* p_in, const float* b, float* pA, float* pB, uint8_t* mA, uint8_t* mB,
	const FreezeWork& w, hipStream_t s, const void* vel = nullptr, int vel_half = 0, int z_begin = 0, int nzp = 0, size_t vel_comp_cells = 0);   // vel: compute (and store) the divergence of this velocity instead of reading b
bool jacobi_freeze_can_fuse_divergence(const Geom& g);
// flag_tag (first tile launch only, n == 0): the value a tile's mark must hold to be taken -- w.gen as k_freeze_dense writes it, or the tag the
// last masked strip launch wrote (0: w.gen)
hipError_t launch_freeze_tiles(const Geom& g, const float* p_src, const float* b, float* p_dst, const uint8_t* m_src, uint8_t* m_dst,
	const FreezeWork& w, int n, int levels, int level_base, uint32_t* stat, uint32_t stat_hi, hipStream_t s, int z_begin = 0, int nzp = 0, uint32_t flag_tag = 0);
// three more levels for EVERY cell on the streaming strip pipeline (fx_jacobi_stripm.hip; X = 256, single domain): level_in in p_in / m_in ->
// level_in + 3 in p_outA = p_outB, nibbles in m_outA = m_outB; tiles that still relax get `tag` in tile_mark
bool jacobi_freeze_strip_supported(const Geom& g);
hipError_t launch_count_marks(const uint32_t* tile_mark, uint32_t gen, int ntiles, uint32_t* out, hipStream_t s);   // tiles with tile_mark == gen, into *out (a host-visible word)
hipError_t launch_freeze_strip3(const Geom& g, const float* p_in, const float* b, float* p_outA, float* p_outB, const uint8_t* m_in, uint8_t* m_outA, uint8_t* m_outB,
	uint32_t* tile_mark, uint32_t tag, uint32_t* stat, uint32_t stat_hi, int level_in, hipStream_t s);
bool jacobi_freeze_strip4_supported(const Geom& g);
hipError_t launch_freeze_strip4(const Geom& g, const float* p_in, const float* b, float* p_outA, float* p_outB, const uint8_t* m_in, uint8_t* m_outA, uint8_t* m_outB,
	uint32_t* tile_mark, uint32_t tag, uint32_t* stat, uint32_t stat_hi, int level_in, hipStream_t s);
// 2-D grids (fx_jacobi2d.hip): up to jacobi2d_max_sweeps (0: not a 2-D grid / switched off) lock-step sweeps per launch on LDS tiles, with or without the freeze bytes
int jacobi2d_max_sweeps(const Geom& g);
hipError_t launch_jacobi2d(const Geom& g, const float* p_in, const float* b, float* p_out, const uint8_t* frozen_in, uint8_t* frozen_out, int sweeps, hipStream_t s);
// sweeps fused per launch for this geometry (1 = no fused path); requested > 0 overrides the default
int jacobi_fused_max_sweeps(const Geom& g, int requested, int nzp);
bool jacobi_prefers_three(const Geom& g, int requested, int nzp);
bool jacobi_prefers_four(const Geom& g, int requested, int nzp);
// The LDS hand-overs of the strip kernels wait in bounded loops; a wait that runs out raises a device word (per translation unit) instead
// of hanging the device or continuing silently.  Read-and-clear on the current device; fx_synchronize turns a raised word into FX_E_DEVICE.
// fx_field_digest's kernel: two wrapping sums of 64-bit mixes of (stored bits, key0 + element index) over `count` elements, added to out[0..1]
hipError_t launch_digest(const void* v, size_t count, int elem_bytes, unsigned long long key0, unsigned long long* out, hipStream_t s);
hipError_t strip3_fault_take(unsigned* out);
hipError_t strip4_fault_take(unsigned* out);
// rec (optional, slab ranks): the step record of launch_face_need is produced by this launch when it can be (fp32 3-D kernel
// over exactly the owned planes); *rec_done tells whether it was
hipError_t launch_project(const Geom& g, const SimParams& sp, int half_store, const void* vel_in, const float* p,
	void* vel_out, int z_begin, int z_end, hipStream_t s, int* rec = nullptr, int digest = 0, const unsigned* halo_overflow = nullptr,
	bool* rec_done = nullptr);
// multi-GPU: what the next advection will need from the z-neighbours (rec[0] planes below, rec[1] above; exact for time steps
// <= dt), closed with the options digest (rec[2]) and this step's halo-overflow flag (rec[3]); rec = 4 device ints
hipError_t launch_face_need(const Geom& g, int half_store, const void* vel, float dt, int address, int digest, const unsigned* halo_overflow, int* rec, hipStream_t s);
hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s);   // device-to-device, as a kernel
hipError_t launch_copy_velocity(const Geom& g, int half_store, const void* vel_in, void* vel_out, hipStream_t s);

// ---- layout conversion between dense fp32 host layouts and device storage (fx_sim.hip)
// scalar planes: nplanes x cells fp32 <-> T ; colour: cells x 4
hipError_t launch_to_storage(const float* src, void* dst, size_t n, int half_store, hipStream_t s);
hipError_t launch_from_storage(const void* src, float* dst, size_t n, int half_store, hipStream_t s);

// ---- ray march launchers
// counters (FX_OPT_COUNT_SAMPLES, else null): 64 shards x { colour samples of view rays, density samples of light / AO rays, light-map
// fetches }, added to by every thread that took one
const int kSampleShards = 64;
// plain path (fx_render.hip): every sample gathers its taps
hipError_t launch_raymarch_light(const Geom& g, int half_store, const void* color, uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, uint32_t num_samples, hipStream_t s, unsigned long long* counters = nullptr);
hipError_t launch_raymarch_view(const Geom& g, int half_store, const void* color, const uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, int cube_size, uint32_t mask, uint32_t num_samples,
	uint32_t num_light_samples, int separate, uint8_t* cube, hipStream_t s, unsigned long long* counters = nullptr);
// direct screen-space march (row f-2): one ray per pixel, blended into the RGBA8 target (and/or kept as float4)
hipError_t launch_raycast_direct(const Geom& g, int half_store, const void* color, const uint32_t* lightmap,
	const FrameConsts& fc, const float* sh, int W, int H, uint32_t num_samples, uint32_t num_light_samples, int separate,
	uint8_t* target, float* out_float, hipStream_t s, unsigned long long* counters = nullptr);
// accelerated path (fx_render_accel.hip; the default, bit-identical to the plain one).  Device scratch, owned by the context:
struct RenderAccel {
	float* occ;          // per 4^3 block: max alpha over everything a sample based in the block can touch (n cells) + the maxima it is dilated from (n cells)
	float* alpha;        // alpha-only fp32 copy of the colour volume, X * Y * Zg
	uint32_t* bits;      // { pos, vis } masks of the fine level (fine_words each) [+ { pos, vis } of the LDS level (mask_words each) when msh > 0]
	uint32_t* list;      // ids of the light-map voxels that cast rays, a segment of X * Y entries per z plane
	float* gi;           // light probe only (allocated by fx_set_sh): direction of each listed voxel's occlusion ray, 3 floats per list entry
	uint32_t* cells;     // ids of the 4^3 cells that may hold a lit voxel, a segment of CX * CY entries per cell layer
	uint32_t* ctr;       // counters, a cache line apart: list lengths per z plane, work heads of the view march, cell-list lengths per layer
	int CX, CY, CZ;      // fine level: 4^3 blocks
	int msh, MX, MY, MZ; // level held in the LDS: (4 << msh)^3 blocks, at most 262144 of them
	uint32_t fine_words, mask_words;
	uint32_t frame;      // renders built so far: the counters are double-buffered (set frame & 1), every build pass clears the list lengths of the NEXT render's set
	uint32_t fill_frame; // filling build passes so far: each leaves a bit per cell "holds a lit voxel" (in `cells`, two sets) for the next one
};
void render_accel_layout(const Geom& g, RenderAccel* a);       // fills the dimensions
size_t render_accel_bits_words(const RenderAccel& a);
size_t render_accel_ctr_words(const Geom& g);
// alpha_current: a.alpha holds this colour field's alpha already (AdvectAlpha)
// fill (with alpha_current, on grids whose extents are powers of two -- a voxel's centre sample is then its own alpha): the pass also writes the
// unlit voxels' light-map constant and lists the lit voxels, i.e. it does k_light_cells' and k_light_classify's work in the same sweep over
// the side volume; *filled says whether it did (launch_accel_light is then told so)
// incremental: the light map still holds what the previous filling pass and its ray kernels left, and the unlit value is the same -- only
// the cells that held a lit voxel then need the constant again
struct LightFill { uint32_t* lightmap; const FrameConsts* fc; int has_sh; bool incremental; };
hipError_t launch_accel_build(const Geom& g, int half_store, const void* color, const RenderAccel& a, hipStream_t s, bool alpha_current = false,
	const LightFill* fill = nullptr, bool* filled = nullptr);
hipError_t launch_accel_light(const Geom& g, const RenderAccel& a, uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	uint32_t num_samples, hipStream_t s, unsigned long long* counters = nullptr, bool filled = false);
hipError_t launch_accel_view(const Geom& g, int half_store, const void* color, const uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	int cube_size, uint32_t mask, uint32_t num_samples, uint32_t num_light_samples, int separate, uint8_t* cube, const RenderAccel& a, hipStream_t s,
	unsigned long long* counters = nullptr);
hipError_t launch_accel_direct(const Geom& g, int half_store, const void* color, const uint32_t* lightmap, const FrameConsts& fc, const float* sh,
	int W, int H, uint32_t num_samples, uint32_t num_light_samples, int separate, uint8_t* target, float* out_float, const RenderAccel& a, hipStream_t s,
	unsigned long long* counters = nullptr);
// 2-D visualiser (PSVisualizeColor): colour[parity] of a Z = 1 grid onto the render target
hipError_t launch_visualize_color(const Geom& g, int half_store, const void* color, int W, int H, uint8_t* target, float* out_float, hipStream_t s);
hipError_t launch_lightmap_decode(const uint32_t* lightmap, float* out, size_t n, hipStream_t s);

// ---- cube map -> screen resolve (fx_resolve.hip; row f-1)
hipError_t launch_resolve_cube(const uint8_t* cube_mip, int N, const FrameConsts& fc, int W, int H, uint8_t* target,
	float* out_float, hipStream_t s);
hipError_t launch_clear_target(uint8_t* target, int W, int H, const float rgba[4], hipStream_t s);
// sky pass (PSEnvironment): float radiance cube [6][n][n][3] on the device -> target (opaque write) and/or float4
hipError_t launch_environment(const float* cube, int n, const FrameConsts& fc, int W, int H, uint8_t* target, float* out_float, hipStream_t s);

// ---- BC6H_UF16 / DDS cube (fx_bc6h.hip; row f-4)
hipError_t launch_bc6h_decode(const void* blocks_dev, int nbx, int nby, int n, float* out_dev, hipStream_t s);
enum { DDS_BC6H_UF16 = 0, DDS_RGBA32F = 1, DDS_RGB32F = 2, DDS_RGBA16F = 3, DDS_RGBA8 = 4 };
struct DdsCube { uint32_t size, mips; int kind; size_t face_offset[6], mip_offset[16]; };
bool dds_cube_layout(const void* dds, size_t bytes, DdsCube* out);
void dds_linear_face_to_rgb(const void* texels, int kind, size_t n, float* out);

// ---- SH light probe (fx_sh.hip): cube float[6][n][n][3] (device) -> out float[27] (device)
hipError_t launch_sh_transform(const float* cube, int n, float* scratch0, float* scratch1, float* w0, float* w1,
	float* out27, hipStream_t s);
size_t sh_scratch_floats(int n, int which);

}  // namespace fx
