// fx_advect_lds.hip -- CSAdvect.hlsl:41-79 (/root/reference/FluidX12/Content/Shaders/) with the trilinear taps taken from an
// LDS-staged tile instead of 35 per-lane gathers through the texture-address path.
//
// Why: k_advect_fast (fx_sim.hip) issues 3 + 24 + 8 vector loads per voxel whose addresses depend on the voxel's own
// velocity.  rocprofv3 (profiles/archive/r02a_sq_counters.json): its waves issue 10 % of their cycles, stand 36 % at a full vector-memory
// queue and 54 % in s_waitcnt -- 19 cycles per wave-load against 6.7 for the L1's data path; every (row, plane) of the
// fields is fetched by up to eight different gather instructions of neighbouring waves, out of a 32-KiB L1 that 24 resident
// waves overflow.  But the back-trace is short almost everywhere: |u| dt N < 1 cell for 93-99 % of the 64-voxel rows of the
// 256^3 bench state (steps 5-120, tools/reach_hist.py), because the plume is a small part of the volume.
//
// Design: a workgroup of 8 waves owns a 64 x 8 (x, y) tile and streams along z.  A ring of FOUR plane slots in the LDS holds
// planes z-1, z, z+1 of the tile plus a one-cell border (66 x 10 cells: colour as float4, velocity as three float planes,
// 18 KiB per slot) while plane z+2 is in flight: `global_load_lds` (the gfx950 LDS-DMA load: no staging VGPRs, no ds_write pass)
// fills the fourth slot straight from HBM, lane-linear -- a slot is laid out in the order the lanes enumerate its cells,
// addressing (CLAMP / MIRROR, slab range) is applied to the SOURCE address.  Per voxel: own velocity from the LDS, back-trace,
// and if every lane of the wave lands inside the +-1 window (a wave-uniform ballot) the 24 + 8 taps are LDS reads
// (ds_read2_b32 / ds_read_b128, lanes = consecutive cells, conflict-free).  A voxel with a longer trace is put on a list and advected
// by k_advect_far right behind this kernel, one thread per noted voxel, with the gathers of k_advect_fast, arithmetic unchanged
// (round 3; before, its wave gathered inside this kernel and held the workgroup's barrier meanwhile: FLUIDX_ADVECT_DEFER=0).  Fields
// are read from HBM 1.29 x (the tile border) instead of being gathered 8 x through L1/L2.  73.9 KiB of LDS per workgroup: two
// workgroups (16 waves) per CU.
//
// Arithmetic, association order and rounding are those of k_advect / k_advect_fast: bit-identical outputs
// (tests/test_gpu_sim.py::test_advect_lds_path_bit_identical, ::test_advect_deferred_voxels_over_changing_flows).  X >= 64, Y >= 8;
// fp32 or binary16 storage.  P2 = power-of-two extents: reciprocal multiplies and shifts; otherwise (150^3, the reference's GI preset,
// Bin/FluidGI.bat:1) divisions as in k_advect, rows and planes addressed by multiplies, and the last tile of a row / column of tiles
// carries lanes without a voxel (they stage and keep the barriers, nothing else).
#include "fx_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

namespace fx {

namespace {

constexpr int TX = 64;                            // voxels per workgroup and plane: TX x TY (TY = 8 rows, one per wave; 16 as an experiment)
constexpr int HX = TX + 2;                        // staged cells per row
constexpr int NSLOT = 4;
// staged cells per plane, bytes per velocity component (fp16 storage: the half sits zero-extended in a dword -- a 2-byte LDS-DMA load
// writes a dword per lane, tools/micro/ldslds2.cpp), colour bytes per slot: fp32 = one float4 per cell; fp16 = two dword planes
// (r|g, b|a) -- there is no 8-byte LDS-DMA
template <bool HALF, int TY> struct Lay {
	static constexpr int NCELL = HX * (TY + 2);                                  // 660 (TY = 8)
	static constexpr int VEL_BYTES = NCELL * 4;                                  // 2640
	static constexpr int COL_BYTES = HALF ? 2 * NCELL * 4 : NCELL * 16;          // 5280 | 10560
	static constexpr int SLOT_BYTES = COL_BYTES + 3 * VEL_BYTES;                 // 13200 | 18480 (multiples of 16)
};
typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// the fp32 result is rounded to binary16 in a SEPARATE step, as a typed store of an fp32 register does (fx_sim.hip, Store<true>)
__device__ __forceinline__ h16 to_h16(float v) { asm("" : "+v"(v)); return (h16)v; }
__device__ __forceinline__ float lo_half(uint32_t d) { return (float)__builtin_bit_cast(h16, (uint16_t)(d & 0xffffu)); }
__device__ __forceinline__ float hi_half(uint32_t d) { return (float)__builtin_bit_cast(h16, (uint16_t)(d >> 16)); }

__device__ __forceinline__ float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

__device__ __forceinline__ int addr_tap(int i, int n, int mode)
{
	if (mode == FX_ADDRESS_MIRROR) {
		const int period = 2 * n;
		int m = i % period;
		if (m < 0) m += period;
		return m < n ? m : period - 1 - m;
	}
	return min(max(i, 0), n - 1);
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// one LDS-DMA load per lane: 16 or 4 bytes from `src` (per lane) to lds_base + lane * size (wave-uniform base)
// The four LDS-DMA loads that stage 64 cells (one per lane) of a plane: colour (16 B per lane) to lds_col + 16 lane, the three
// velocity components (4 B per lane) to lds_vel + k * VEL_BYTES + 4 lane.  `global_load_lds_*` writes to M0 + lane * size; the
// LDS offsets are wave-uniform (SGPRs).  Inline assembly on purpose: through the builtin the compiler knows the LDS is being
// written and puts `s_waitcnt vmcnt(0)` in front of the next ds_read -- every tap read of plane z would then wait for plane
// z + 2, which has the whole iteration to land.  The workgroup barrier (after an explicit vmcnt(0)) is what orders the ring.
// Addresses: a wave-uniform base per field (SGPR pair: the field + the plane's offset, made by scalar adds per plane) + a 32-bit byte
// offset per lane that never changes (the lane's cell inside a plane) -- `global_load_lds_* v_offset, s[base:base+1]`.  (Round 6; as
// 64-bit addresses per lane every plane cost eight v_lshl_add_u64 and their moves, and the kernel is bound by what it issues.)
template <int VEL_BYTES>
__device__ __forceinline__ void stage64(const void* col, const void* v0, const void* v1, const void* v2, uint32_t off_col, uint32_t off_vel, uint32_t lds_col, uint32_t lds_vel)
{
	uint32_t keep;
	lds_col = __builtin_amdgcn_readfirstlane(lds_col);     // wave-uniform by construction; pins the operands to SGPRs for the "s" constraints
	lds_vel = __builtin_amdgcn_readfirstlane(lds_vel);
	asm volatile(
		"s_mov_b32 %0, m0\n\t"
		"s_mov_b32 m0, %7\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dwordx4 %5, %1\n\t"
		"s_mov_b32 m0, %8\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dword %6, %2\n\t"
		"s_add_u32 m0, m0, %9\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dword %6, %3\n\t"
		"s_add_u32 m0, m0, %9\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dword %6, %4\n\t"
		"s_mov_b32 m0, %0"
		: "=&s"(keep)
		: "s"(col), "s"(v0), "s"(v1), "s"(v2), "v"(off_col), "v"(off_vel), "s"(lds_col), "s"(lds_vel), "n"(VEL_BYTES)
		: "scc");                                              // s_add_u32 writes SCC: the compiler must not keep a compare live across the statement
}

// fp16 storage: colour texel = 8 bytes -> its two dwords to two LDS planes NCELL * 4 bytes apart; a velocity half -> a dword per lane
template <int VEL_BYTES>
__device__ __forceinline__ void stage64h(const void* col, const void* v0, const void* v1, const void* v2, uint32_t off_col, uint32_t off_vel, uint32_t lds_col, uint32_t lds_vel)
{
	uint32_t keep;
	lds_col = __builtin_amdgcn_readfirstlane(lds_col);
	lds_vel = __builtin_amdgcn_readfirstlane(lds_vel);
	const uint32_t off_col_hi = off_col + 4u;                     // (an instruction offset would shift the LDS address too)
	asm volatile(
		"s_mov_b32 %0, m0\n\t"
		"s_mov_b32 m0, %8\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dword %5, %1\n\t"
		"s_add_u32 m0, m0, %10\n\t"
		"s_nop 0\n\t"
		"global_load_lds_dword %6, %1\n\t"
		"s_mov_b32 m0, %9\n\t"
		"s_nop 0\n\t"
		"global_load_lds_ushort %7, %2\n\t"
		"s_add_u32 m0, m0, %10\n\t"
		"s_nop 0\n\t"
		"global_load_lds_ushort %7, %3\n\t"
		"s_add_u32 m0, m0, %10\n\t"
		"s_nop 0\n\t"
		"global_load_lds_ushort %7, %4\n\t"
		"s_mov_b32 m0, %0"
		: "=&s"(keep)
		: "s"(col), "s"(v0), "s"(v1), "s"(v2), "v"(off_col), "v"(off_col_hi), "v"(off_vel), "s"(lds_col), "s"(lds_vel), "n"(VEL_BYTES)
		: "scc");
}

template <typename T>
__device__ __forceinline__ T ldg32(const void* base, uint32_t byte_off)
{
	return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}


// the 24 + 8 taps of a voxel as gathers from global memory (k_advect_fast's): a trace that leaves the staged window
// row / plane offsets in cells: shifts on power-of-two grids
template <bool P2> __device__ __forceinline__ uint32_t rows(uint32_t y, int lgX, const Geom& g) { return P2 ? y << lgX : y * (uint32_t)g.X; }
template <bool P2> __device__ __forceinline__ uint32_t planes(uint32_t z, int lgP, const Geom& g) { return P2 ? z << lgP : z * ((uint32_t)g.X * (uint32_t)g.Y); }

template <bool HALF, bool P2>
__device__ __forceinline__ void advect_gather(const Geom& g, const SimParams& sp, int ix, int iy, int z0, int z1, bool z_present, float fx, float fy, float fz,
	const char* v0, const char* v1, const char* v2, const void* __restrict__ col_in, int lgX, int lgP, unsigned* halo_overflow, float (&u)[3], float (&c)[4])
{
	// ---- a longer trace somewhere in the wave: the gathers of k_advect_fast, from global memory ----------------------
	const int xa0 = addr_tap(ix, g.X, sp.address), xa1 = addr_tap(ix + 1, g.X, sp.address);
	const int ya0 = addr_tap(iy, g.Y, sp.address), ya1 = addr_tap(iy + 1, g.Y, sp.address);
	if (!z_present) {
		atomicOr(halo_overflow, 1u);
		z0 = min(max(z0, g.zlo), g.zhi);
		z1 = min(max(z1, g.zlo), g.zhi);
	}
	const uint32_t p0 = planes<P2>((uint32_t)g.lz(z0), lgP, g), p1 = planes<P2>((uint32_t)g.lz(z1), lgP, g);
	const uint32_t ry0 = rows<P2>((uint32_t)ya0, lgX, g), ry1 = rows<P2>((uint32_t)ya1, lgX, g);
	const uint32_t c000 = p0 + ry0 + (uint32_t)xa0, c100 = p0 + ry0 + (uint32_t)xa1;
	const uint32_t c010 = p0 + ry1 + (uint32_t)xa0, c110 = p0 + ry1 + (uint32_t)xa1;
	const uint32_t c001 = p1 + ry0 + (uint32_t)xa0, c101 = p1 + ry0 + (uint32_t)xa1;
	const uint32_t c011 = p1 + ry1 + (uint32_t)xa0, c111 = p1 + ry1 + (uint32_t)xa1;
	const char* vb[3] = { v0, v1, v2 };
	auto gv = [](const char* b, uint32_t cell) -> float { return HALF ? (float)ldg32<h16>(b, cell * 2u) : ldg32<float>(b, cell * 4u); };
	auto gc = [&](uint32_t cell) -> float4 {
		if (HALF) { const h16x4 h = ldg32<h16x4>(col_in, cell * 8u); return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w); }
		return ldg32<float4>(col_in, cell * 16u);
	};
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float c00 = lerpf(gv(vb[a], c000), gv(vb[a], c100), fx);
		const float c10 = lerpf(gv(vb[a], c010), gv(vb[a], c110), fx);
		const float c01 = lerpf(gv(vb[a], c001), gv(vb[a], c101), fx);
		const float c11 = lerpf(gv(vb[a], c011), gv(vb[a], c111), fx);
		u[a] = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
	}
	const float4 t000 = gc(c000), t100 = gc(c100), t010 = gc(c010), t110 = gc(c110);
	const float4 t001 = gc(c001), t101 = gc(c101), t011 = gc(c011), t111 = gc(c111);
#define FX_TRI(m) lerpf(lerpf(lerpf(t000.m, t100.m, fx), lerpf(t010.m, t110.m, fx), fy), \
	lerpf(lerpf(t001.m, t101.m, fx), lerpf(t011.m, t111.m, fx), fy), fz)
	c[0] = FX_TRI(x); c[1] = FX_TRI(y); c[2] = FX_TRI(z); c[3] = FX_TRI(w);
#undef FX_TRI
}

// impulse (CSAdvect.hlsl:59-68), attenuation and the stores of one voxel; u / c = the traced velocity / colour
// ALPHA: the stored alpha once more, as fp32, into the render's alpha-only side volume (fx_render_accel.hip: every density tap of the
// marches reads that volume; written here, the render's build pass no longer reads the whole colour field to extract it)
template <bool HALF, bool ALPHA>
__device__ __forceinline__ void advect_finish(const SimParams& sp, float (&u)[3], float (&c)[4], float ex, float dx, float dz, float dt, float atten,
	uint32_t id, uint32_t stride, void* __restrict__ vel_out, void* __restrict__ col_out, float* __restrict__ alpha_out)
{
	// ---- impulse (CSAdvect.hlsl:59-68).  exp2(ex) >= e^-4 needs ex >= -5.77: a wave whose lanes are all far below that
	// skips the transcendental; the decision itself still uses the computed basis, exactly as before
	if (__builtin_amdgcn_ballot_w64(ex > -6.5f) != 0) {
		const float basis = exp2f(ex);
		if (basis >= 0.0183156393f) {
			float Fx, Fy, Fz;
			if (sp.is3d) {
				Fx = fmaf(basis, 0.0f, dz * -200.0f);
				Fy = fmaf(basis, 192.0f, 0.0f);
				Fz = fmaf(basis, 0.0f, dx * 200.0f);
			} else {
				Fx = 0.0f; Fy = basis * 48.0f; Fz = 0.0f;
			}
			u[0] = fmaf(Fx, dt, u[0]); u[1] = fmaf(Fy, dt, u[1]); u[2] = fmaf(Fz, dt, u[2]);
			const float bdt = basis * dt;
			c[0] = saturatef(fmaf(bdt, 8.0f, c[0]));
			c[1] = saturatef(fmaf(bdt, 16.0f, c[1]));
			c[2] = saturatef(fmaf(bdt, 40.0f, c[2]));
			c[3] = saturatef(fmaf(bdt, 40.0f, c[3]));
		}
	}
	// the stores: uniform bases + 32-bit byte offsets (`global_store v_offset, v_data, s[base]`; the launcher keeps 16 x cells below 2^32)
	char* vo = static_cast<char*>(vel_out);
	char* co = static_cast<char*>(col_out);
	if (HALF) {
		*reinterpret_cast<h16*>(vo + id * 2u) = to_h16(u[0] * atten);
		*reinterpret_cast<h16*>(vo + (stride + id) * 2u) = to_h16(u[1] * atten);
		*reinterpret_cast<h16*>(vo + (2u * stride + id) * 2u) = to_h16(u[2] * atten);
		h16x4 hc;
		hc.x = to_h16(c[0] * atten); hc.y = to_h16(c[1] * atten); hc.z = to_h16(c[2] * atten); hc.w = to_h16(c[3] * atten);
		*reinterpret_cast<h16x4*>(co + id * 8u) = hc;
		if (ALPHA) *reinterpret_cast<float*>(reinterpret_cast<char*>(alpha_out) + id * 4u) = (float)hc.w;
	} else {
		*reinterpret_cast<float*>(vo + id * 4u) = u[0] * atten;
		*reinterpret_cast<float*>(vo + (stride + id) * 4u) = u[1] * atten;
		*reinterpret_cast<float*>(vo + (2u * stride + id) * 4u) = u[2] * atten;
		*reinterpret_cast<float4*>(co + id * 16u) = make_float4(c[0] * atten, c[1] * atten, c[2] * atten, c[3] * atten);
		if (ALPHA) *reinterpret_cast<float*>(reinterpret_cast<char*>(alpha_out) + id * 4u) = c[3] * atten;
	}
}

}  // namespace

// HALF: fp16 storage of velocity / colour (BASELINE configs[4], the reference's RGBA16F): arithmetic unchanged (fp32), half the HBM
// bytes, 13.2 instead of 18.5 KB per slot
// DEFER: a voxel whose trace leaves the staged window is not gathered here -- its wave would issue 35 scattered loads for a few lanes and
// hold the workgroup's barrier meanwhile -- but appended to the workgroup's segment of `far_list` (a placeholder is stored to its cell)
// and advected by k_advect_far afterwards.  7 % of the waves of a developed plume (frame 132) have such a lane, 1.5 % of the voxels.
template <bool HALF, int TY, bool DEFER, bool P2, bool ALPHA>
__global__ __launch_bounds__(64 * TY) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_advect_lds(const Geom g, const SimParams sp,
	const void* __restrict__ vel_in, const void* __restrict__ col_in, void* __restrict__ vel_out, void* __restrict__ col_out, float* __restrict__ alpha_out,
	int z_begin, int nzp, int zchunk, int nchunks, unsigned* halo_overflow, float rX, float rY, float rZ, float inv_rr,
	int lgX, int lgY, int lg_gx, int lg_gy, uint32_t* __restrict__ far_list, uint32_t* __restrict__ far_flat, uint32_t* __restrict__ far_total, uint32_t far_cap)
{
	__shared__ uint32_t far_n, far_base;
	extern __shared__ __attribute__((aligned(16))) char lds[];   // NSLOT slots: [colour float4 x NCELL][velocity float x 3 x NCELL]

	const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// tile order: XCD k (= workgroup index % 8) walks the k-th contiguous eighth of the (x, y, chunk)-ordered tile sequence, so
	// that the tiles sharing a border row are neighbours in ONE L2
	// (not P2: lg_gx / lg_gy carry the tile COUNTS per row / column of tiles)
	const int ntx = P2 ? 1 << lg_gx : lg_gx, nty = P2 ? 1 << lg_gy : lg_gy;
	const int ntiles = ntx * nty * nchunks;
	int tl = (int)blockIdx.x;
	{
		const int q = ntiles >> 3, r = ntiles & 7, xcd = tl & 7, j = tl >> 3;
		tl = xcd * q + min(xcd, r) + j;
	}
	const int tx = P2 ? tl & (ntx - 1) : tl % ntx, ty = P2 ? (tl >> lg_gx) & (nty - 1) : (tl / ntx) % nty, chunk = P2 ? tl >> (lg_gx + lg_gy) : tl / (ntx * nty);
	const int x0t = tx << 6, y0t = ty * TY;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_begin + nzp);
	if (zb >= ze) return;                                        // uniform for the workgroup

	const int lgP = lgX + lgY;
	constexpr int COL_BYTES = Lay<HALF, TY>::COL_BYTES, SLOT_BYTES = Lay<HALF, TY>::SLOT_BYTES;
	constexpr int NCELL = Lay<HALF, TY>::NCELL, VEL_BYTES = Lay<HALF, TY>::VEL_BYTES, NT = 64 * TY;
	constexpr uint32_t ES = HALF ? 2 : 4, CS = HALF ? 8 : 16;    // bytes per velocity element / colour texel in HBM
	const uint32_t stride = planes<P2>((uint32_t)g.nzl(), lgP, g);   // cells between velocity component planes
	const char* v0 = reinterpret_cast<const char*>(vel_in);
	const char* v1 = v0 + (size_t)stride * ES;
	const char* v2 = v1 + (size_t)stride * ES;

	// ---- which cells of a staged plane this lane fetches: cell c = tid (every wave) and 512 + tid (waves 0..2) -------------
	// source = the ADDRESSED cell (clamp / mirror of the unclamped coordinate), so a tap index needs no addressing afterwards
	uint32_t src_a, src_b = 0;
	{
		const int c = tid, r = c / HX, cc = c - r * HX;
		src_a = rows<P2>((uint32_t)addr_tap(y0t - 1 + r, g.Y, sp.address), lgX, g) + (uint32_t)addr_tap(x0t - 1 + cc, g.X, sp.address);
	}
	const bool has_b = NT + tid < NCELL;
	if (has_b) {
		const int c = NT + tid, r = c / HX, cc = c - r * HX;
		src_b = rows<P2>((uint32_t)addr_tap(y0t - 1 + r, g.Y, sp.address), lgX, g) + (uint32_t)addr_tap(x0t - 1 + cc, g.X, sp.address);
	}
	const bool wave_has_b = NT + wave * 64 < NCELL;              // waves 0, 1, 2

	// stage global plane zq (unclamped; may be -1 or Zg, or beyond what this slab holds) into its ring slot
	const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)lds;
	auto fill = [&](int zq) {
		const int za = min(max(addr_tap(zq, g.Zg, sp.address), g.zlo), g.zhi);   // voxels that would need a plane this slab lacks take the flagged path
		const size_t pz = planes<P2>((uint32_t)g.lz(za), lgP, g);                  // wave-uniform: the plane's offset goes into the bases
		const uint32_t slot = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(((zq + 4) & 3) * SLOT_BYTES));
		const char* bc = reinterpret_cast<const char*>(col_in) + pz * CS;
		const char *b0 = v0 + pz * ES, *b1 = v1 + pz * ES, *b2 = v2 + pz * ES;
		if (HALF) stage64h<VEL_BYTES>(bc, b0, b1, b2, src_a * CS, src_a * ES, slot + (uint32_t)wave * (64 * 4), slot + COL_BYTES + (uint32_t)wave * (64 * 4));
		else stage64<VEL_BYTES>(bc, b0, b1, b2, src_a * CS, src_a * ES, slot + (uint32_t)wave * (64 * 16), slot + COL_BYTES + (uint32_t)wave * (64 * 4));
		if (wave_has_b) {
			if (has_b) {
				if (HALF) stage64h<VEL_BYTES>(bc, b0, b1, b2, src_b * CS, src_b * ES, slot + (NT + (uint32_t)wave * 64) * 4, slot + COL_BYTES + (NT + (uint32_t)wave * 64) * 4);
				else stage64<VEL_BYTES>(bc, b0, b1, b2, src_b * CS, src_b * ES, slot + (NT + (uint32_t)wave * 64) * 16, slot + COL_BYTES + (NT + (uint32_t)wave * 64) * 4);
			}
		}
	};

	if (DEFER && tid == 0) far_n = 0u;
	fill(zb - 1);
	fill(zb);
	fill(zb + 1);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // LDS-DMA completion is counted by vmcnt
	__syncthreads();

	const int x = x0t + lane, y = y0t + wave;
	const bool has_voxel = P2 || (x < g.X && y < g.Y);           // the last tiles of a row / column on other grids
	const float dt = sp.dt;
	const float px = P2 ? ((float)x + 0.5f) * rX : ((float)x + 0.5f) / (float)g.X;     // (== / (float)g.X exactly on a power of two; Simulation.hlsli:10)
	const float py = P2 ? ((float)y + 0.5f) * rY : ((float)y + 0.5f) / (float)g.Y;
	const float dx = px + -0.5f, dy = py + -0.100000001f;
	const float dxy2 = fmaf(dy, dy, dx * dx);
	const uint32_t own = (uint32_t)((wave + 1) * HX + lane + 1);  // this voxel's cell inside a slot
	const float atten = fmaxf(fmaf(-dt, 0.200000003f, 1.0f), 0.0f);

	auto compute = [&](int z) {
		if (!has_voxel) return;
		const float pz = P2 ? ((float)z + 0.5f) * rZ : ((float)z + 0.5f) / (float)g.Zg;
		const float dz = pz + -0.5f;
		const float d2 = fmaf(dz, dz, dxy2);
		const float ex = ((d2 * -4.0f) * inv_rr) * 1.44269502f;

		const char* sc = lds + ((z + 4) & 3) * SLOT_BYTES;      // slot of plane z
		const uint32_t* vs = reinterpret_cast<const uint32_t*>(sc + COL_BYTES);
		auto vval = [](uint32_t d) -> float { return HALF ? lo_half(d) : __uint_as_float(d); };      // a staged velocity dword as fp32
		const float u0x = vval(vs[own]), u0y = vval(vs[NCELL + own]), u0z = vval(vs[2 * NCELL + own]);
		const float ax = fmaf(-u0x, dt, px), ay = fmaf(-u0y, dt, py), az = fmaf(-u0z, dt, pz);
		const float tx_ = ax * (float)g.X - 0.5f, ty_ = ay * (float)g.Y - 0.5f, tz_ = az * (float)g.Zg - 0.5f;
		const float flx = floorf(tx_), fly = floorf(ty_), flz = floorf(tz_);
		const float fx = tx_ - flx, fy = ty_ - fly, fz = tz_ - flz;
		const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
		const int jx = ix - x, jy = iy - y, jz = iz - z;        // in {-1, 0} when the trace stays inside the window
		int z0 = addr_tap(iz, g.Zg, sp.address), z1 = addr_tap(iz + 1, g.Zg, sp.address);
		const bool z_present = z0 >= g.zlo && z0 <= g.zhi && z1 >= g.zlo && z1 <= g.zhi;
		const bool inwin = (unsigned)(jx + 1) <= 1u && (unsigned)(jy + 1) <= 1u && (unsigned)(jz + 1) <= 1u && z_present;

		float u[3], c[4];
		if (DEFER && !inwin) {                                   // noted for k_advect_far; below it reads the window like a voxel at rest
			const uint32_t slot = atomicAdd(&far_n, 1u);
			far_list[(size_t)blockIdx.x * far_cap + slot] = P2 ? (uint32_t)x | (uint32_t)y << lgX | (uint32_t)z << lgP
				: ((uint32_t)z * (uint32_t)g.Y + (uint32_t)y) * (uint32_t)g.X + (uint32_t)x;
		}
		if (DEFER || inwin) {
			// ---- the lanes that trace into the staged window: all 32 taps are LDS reads.  Without DEFER a mixed wave runs both
			// branches under partial exec masks (its gathers then only carry the few far-tracing lanes)
			const int kx = DEFER && !inwin ? 0 : jx, ky = DEFER && !inwin ? 0 : jy, kz = DEFER && !inwin ? z : iz;
			const uint32_t c0 = (uint32_t)((int)own + ky * HX + kx);               // cell of tap (x0, y0) inside a slot
			const char* s0 = lds + ((kz + 4) & 3) * SLOT_BYTES;
			const char* s1 = lds + ((kz + 5) & 3) * SLOT_BYTES;
#pragma unroll
			for (int a = 0; a < 3; ++a) {
				const uint32_t* p0 = reinterpret_cast<const uint32_t*>(s0 + COL_BYTES + a * VEL_BYTES) + c0;
				const uint32_t* p1 = reinterpret_cast<const uint32_t*>(s1 + COL_BYTES + a * VEL_BYTES) + c0;
				const float c00 = lerpf(vval(p0[0]), vval(p0[1]), fx);
				const float c10 = lerpf(vval(p0[HX]), vval(p0[HX + 1]), fx);
				const float c01 = lerpf(vval(p1[0]), vval(p1[1]), fx);
				const float c11 = lerpf(vval(p1[HX]), vval(p1[HX + 1]), fx);
				u[a] = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
			}
			float4 t000, t100, t010, t110, t001, t101, t011, t111;
			if (HALF) {
				// a texel = dword c0 of the r|g plane and dword c0 of the b|a plane
				auto tex = [](const char* slot, uint32_t c) -> float4 {
					const uint32_t rg = reinterpret_cast<const uint32_t*>(slot)[c], ba = reinterpret_cast<const uint32_t*>(slot)[NCELL + c];
					return make_float4(lo_half(rg), hi_half(rg), lo_half(ba), hi_half(ba));
				};
				t000 = tex(s0, c0); t100 = tex(s0, c0 + 1); t010 = tex(s0, c0 + HX); t110 = tex(s0, c0 + HX + 1);
				t001 = tex(s1, c0); t101 = tex(s1, c0 + 1); t011 = tex(s1, c0 + HX); t111 = tex(s1, c0 + HX + 1);
			} else {
				const float4* q0 = reinterpret_cast<const float4*>(s0) + c0;
				const float4* q1 = reinterpret_cast<const float4*>(s1) + c0;
				t000 = q0[0]; t100 = q0[1]; t010 = q0[HX]; t110 = q0[HX + 1];
				t001 = q1[0]; t101 = q1[1]; t011 = q1[HX]; t111 = q1[HX + 1];
			}
#define FX_TRI(m) lerpf(lerpf(lerpf(t000.m, t100.m, fx), lerpf(t010.m, t110.m, fx), fy), \
	lerpf(lerpf(t001.m, t101.m, fx), lerpf(t011.m, t111.m, fx), fy), fz)
			c[0] = FX_TRI(x); c[1] = FX_TRI(y); c[2] = FX_TRI(z); c[3] = FX_TRI(w);
#undef FX_TRI
		} else {
			advect_gather<HALF, P2>(g, sp, ix, iy, z0, z1, z_present, fx, fy, fz, v0, v1, v2, col_in, lgX, lgP, halo_overflow, u, c);
		}

		const uint32_t id = planes<P2>((uint32_t)g.lz(z), lgP, g) + rows<P2>((uint32_t)y, lgX, g) + (uint32_t)x;
		advect_finish<HALF, ALPHA>(sp, u, c, ex, dx, dz, dt, atten, id, stride, vel_out, col_out, alpha_out);
	};

	for (int z = zb; z < ze; ++z) {
		if (z + 2 <= ze) fill(z + 2);                           // plane ze is the z+1 of the chunk's last plane
		compute(z);
		// the four stores of this step (five with ALPHA) were issued after the LDS-DMA loads and complete after them: vmcnt(4) = "plane
		// z+2 has landed" without waiting for the store acknowledgements.  That the compiler emits exactly that many store instructions
		// per step behind the fill is checked on the generated ISA by tests/test_isa_contract.py
		// (a wave without a single voxel -- rows beyond Y in the last tile of a column -- has issued no store: it waits for everything)
		if (!(P2 || y < g.Y)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		else if (ALPHA) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
		__syncthreads();                                         // plane z+2 is in the ring; plane z-1's slot may be overwritten
	}
	// (behind the loop's last barrier: every append is in.)  The workgroup's notes move from its segment to ONE list every workgroup
	// appends to -- a returning atomic per workgroup, not per note -- so that k_advect_far can share them out evenly: the plume's
	// workgroups note thousands of voxels, most workgroups none.
	if (DEFER) {
		const uint32_t n = far_n;
		if (n != 0u) {                                              // uniform
			if (tid == 0) far_base = atomicAdd(far_total, n);
			__syncthreads();
			const uint32_t base = far_base;
			const uint32_t* seg = far_list + (size_t)blockIdx.x * far_cap;
			for (uint32_t i = (uint32_t)tid; i < n; i += (uint32_t)NT) far_flat[base + i] = seg[i];
		}
	}
}

// The voxels k_advect_lds<.., DEFER = true> put aside, one per thread off the common list: CSAdvect.hlsl:41-79 for one voxel with every tap a
// gather (k_advect_fast's arithmetic, so the result is the one the staged path would have produced had the window been wide enough).
template <bool HALF, bool P2, bool ALPHA>
__global__ __launch_bounds__(256) void k_advect_far(const Geom g, const SimParams sp,
	const void* __restrict__ vel_in, const void* __restrict__ col_in, void* __restrict__ vel_out, void* __restrict__ col_out, float* __restrict__ alpha_out,
	const uint32_t* __restrict__ far_flat, const uint32_t* __restrict__ far_total, uint32_t* __restrict__ far_total_next, unsigned* halo_overflow,
	float rX, float rY, float rZ, float inv_rr, int lgX, int lgY)
{
	const uint32_t n = *far_total;
	if (blockIdx.x == 0 && threadIdx.x == 0) *far_total_next = 0u;   // the counter the NEXT step's advection appends through (the two alternate)
	const int lgP = lgX + lgY;
	constexpr uint32_t ES = HALF ? 2 : 4;
	const uint32_t stride = planes<P2>((uint32_t)g.nzl(), lgP, g);
	const char* v0 = reinterpret_cast<const char*>(vel_in);
	const char* v1 = v0 + (size_t)stride * ES;
	const char* v2 = v1 + (size_t)stride * ES;
	const float dt = sp.dt;
	const float atten = fmaxf(fmaf(-dt, 0.200000003f, 1.0f), 0.0f);
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const uint32_t code = far_flat[i];
		int x, y, z;
		if (P2) { x = (int)(code & (uint32_t)(g.X - 1)); y = (int)((code >> lgX) & (uint32_t)(g.Y - 1)); z = (int)(code >> lgP); }
		else { const uint32_t XY = (uint32_t)g.X * (uint32_t)g.Y, rem = code % XY; z = (int)(code / XY); y = (int)(rem / (uint32_t)g.X); x = (int)(rem % (uint32_t)g.X); }
		const float px = P2 ? ((float)x + 0.5f) * rX : ((float)x + 0.5f) / (float)g.X;
		const float py = P2 ? ((float)y + 0.5f) * rY : ((float)y + 0.5f) / (float)g.Y;
		const float pz = P2 ? ((float)z + 0.5f) * rZ : ((float)z + 0.5f) / (float)g.Zg;
		const float dx = px + -0.5f, dy = py + -0.100000001f, dz = pz + -0.5f;
		const float dxy2 = fmaf(dy, dy, dx * dx);
		const float d2 = fmaf(dz, dz, dxy2);
		const float ex = ((d2 * -4.0f) * inv_rr) * 1.44269502f;
		const uint32_t id = planes<P2>((uint32_t)g.lz(z), lgP, g) + rows<P2>((uint32_t)y, lgX, g) + (uint32_t)x;
		auto gv = [](const char* b, uint32_t cell) -> float { return HALF ? (float)ldg32<h16>(b, cell * 2u) : ldg32<float>(b, cell * 4u); };
		const float u0x = gv(v0, id), u0y = gv(v1, id), u0z = gv(v2, id);
		const float ax = fmaf(-u0x, dt, px), ay = fmaf(-u0y, dt, py), az = fmaf(-u0z, dt, pz);
		const float tx_ = ax * (float)g.X - 0.5f, ty_ = ay * (float)g.Y - 0.5f, tz_ = az * (float)g.Zg - 0.5f;
		const float flx = floorf(tx_), fly = floorf(ty_), flz = floorf(tz_);
		const float fx = tx_ - flx, fy = ty_ - fly, fz = tz_ - flz;
		const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
		const int z0 = addr_tap(iz, g.Zg, sp.address), z1 = addr_tap(iz + 1, g.Zg, sp.address);
		const bool z_present = z0 >= g.zlo && z0 <= g.zhi && z1 >= g.zlo && z1 <= g.zhi;
		float u[3], c[4];
		advect_gather<HALF, P2>(g, sp, ix, iy, z0, z1, z_present, fx, fy, fz, v0, v1, v2, col_in, lgX, lgP, halo_overflow, u, c);
		advect_finish<HALF, ALPHA>(sp, u, c, ex, dx, dz, dt, atten, id, stride, vel_out, col_out, alpha_out);
	}
}


// hipErrorNotSupported: the geometry has no LDS path (the caller falls back to k_advect_fast / k_advect)
hipError_t launch_advect_lds(const Geom& g, const SimParams& sp, int half_store, const void* vel_in, const void* col_in,
	void* vel_out, void* col_out, int z_begin, int z_end, unsigned* halo_overflow, uint32_t* far_scratch, size_t far_words, int far_parity, bool* far_used, hipStream_t s, bool force,
	AdvectAlpha* alpha)
{
	if (far_used) *far_used = false;
	if (alpha) alpha->written = false;
	auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
	const int nzp = z_end - z_begin;
	// rows per workgroup tile.  16 (one 1024-thread workgroup per CU, 1.16 x instead of 1.29 x border) measured 0.228 / 0.269 ms against
	// 0.223 / 0.263 for 8 (256^3, states of step 25 / 110): the bytes it saves it loses to the single workgroup's barrier stalls
	const bool p2 = pow2(g.X) && pow2(g.Y) && pow2(g.Zg);
	const int TY = p2 && FX_KNOB_INT("ADVECT_TILE_ROWS", 8) == 16 ? 16 : 8;
	// where it pays: from 4 M voxels per launch.  128^3 measured 0.032-0.040 ms against 0.030 for k_advect_fast, 150^3 0.049-0.053 against
	// 0.057 for k_advect -- and inside whole steps, with the second launch for the far-tracing voxels, 0.0586 against 0.0555: too few
	// workgroups either way
	const size_t min_voxels = (size_t)1 << 22;
	if (g.Zg <= 1 || g.X < TX || g.Y < TY || nzp < 12 || (!force && (size_t)g.X * g.Y * (size_t)nzp < min_voxels) ||
		g.cells_local() * 16 >= ((size_t)1 << 32))
		return hipErrorNotSupported;
	const int half_on = FX_KNOB_INT("ADVECT_LDS_HALF", 1);
	if (half_store && !half_on) return hipErrorNotSupported;
	auto lg = [](int v) { int k = 0; while ((1 << k) < v) ++k; return k; };
	const int ntx = (g.X + TX - 1) / TX, nty = (g.Y + TY - 1) / TY;
	const int lgX = lg(g.X), lgY = lg(g.Y), lg_gx = p2 ? lg(ntx) : ntx, lg_gy = p2 ? lg(nty) : nty;      // (other extents: the kernel gets the tile counts themselves)
	const int tiles_xy = ntx * nty;
	// planes per workgroup.  Every chunk re-reads two planes.  While the far-tracing voxels were gathered inside this kernel the
	// workgroups over the plume took several times longer than the rest and short chunks (2048 workgroups of 16 planes at 256^3)
	// balanced that: 0.216 ms with 16, 0.219 with 8, 0.241 with 32, 0.258 with 64.  With those voxels deferred every workgroup costs
	// the same: 16 / 32 / 64 planes measure 0.211 / 0.203 / 0.212 ms (fp32), 0.147 / 0.143 / 0.144 (fp16), within the noise -- 32.
	const bool want_defer = far_scratch && FX_KNOB_INT("ADVECT_DEFER", 1) != 0;
	int zchunk = FX_KNOB_INT("ADVECT_ZCHUNK", want_defer ? 32 : 16);
	// grids of a few million voxels (150^3: 57 tiles per plane) have too few workgroups for 32-plane chunks to fill the chip and too many
	// with short ones for one resident round (two workgroups per CU): as many chunks as keep them all resident at once.  150^3, us per
	// launch by planes per chunk: 6 49.8, 8 51.4, 12 51.2, 15 54.8, 19 49.0, 25 57.5, 32 67.5, 64 122 (k_advect: 57.3)
	if (!p2 && !FX_KNOB("ADVECT_ZCHUNK")) {
		const int chunks = std::max(1, 512 / tiles_xy);
		zchunk = std::max(8, (nzp + chunks - 1) / chunks);
	}
	if (zchunk < 4) zchunk = 4;
	if (zchunk > nzp) zchunk = nzp;
	const int nchunks = (nzp + zchunk - 1) / zchunk;
	const float rX = 1.0f / (float)g.X, rY = 1.0f / (float)g.Y, rZ = 1.0f / (float)g.Zg, inv_rr = sp.is3d ? 256.0f : 1024.0f;
	// far-tracing voxels deferred to k_advect_far when the caller lent scratch: [two alternating totals][the common list: an id per voxel]
	// [a segment of 64 * TY * zchunk ids per workgroup] (FLUIDX_ADVECT_DEFER=0: gathered inside the staged kernel, as before)
	const uint32_t nwg = (uint32_t)(tiles_xy * nchunks), far_cap = (uint32_t)(TX * TY * zchunk);
	const size_t flat_words = (size_t)g.X * g.Y * (size_t)nzp;
	const bool defer = want_defer && far_words >= 2 + flat_words + (size_t)nwg * far_cap && ((uint64_t)g.Zg << (lgX + lgY)) <= ((uint64_t)1 << 32);   // (a note = x | y << lgX | z << lgP, or the voxel's index in the whole grid)
	if (far_used) *far_used = defer;                              // (the caller alternates far_parity over the launches that did defer)
	uint32_t* far_total = far_scratch ? far_scratch + (far_parity & 1) : nullptr;
	uint32_t* far_total_next = far_scratch ? far_scratch + ((far_parity & 1) ^ 1) : nullptr;
	uint32_t* far_flat = far_scratch ? far_scratch + 2 : nullptr;
	uint32_t* far_list = far_scratch ? far_scratch + 2 + flat_words : nullptr;
#define FX_ADV(H_, TY_, D_, P_, A_) do { \
		static bool attr_set = false; \
		if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_advect_lds<H_, TY_, D_, P_, A_>), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * Lay<H_, TY_>::SLOT_BYTES); attr_set = true; } \
		hipLaunchKernelGGL((k_advect_lds<H_, TY_, D_, P_, A_>), dim3(tiles_xy * nchunks), dim3(64 * TY_), (NSLOT * Lay<H_, TY_>::SLOT_BYTES), s, g, sp, vel_in, col_in, vel_out, col_out, alpha_out, \
			z_begin, nzp, zchunk, nchunks, halo_overflow, rX, rY, rZ, inv_rr, lgX, lgY, lg_gx, lg_gy, far_list, far_flat, far_total, far_cap); } while (0)
#define FX_FAR(H_, P_, A_) hipLaunchKernelGGL((k_advect_far<H_, P_, A_>), dim3(far_wgs), dim3(256), 0, s, g, sp, vel_in, col_in, vel_out, col_out, alpha_out, far_flat, far_total, far_total_next, halo_overflow, rX, rY, rZ, inv_rr, lgX, lgY)
	const int far_wgs = 1024;                                       // 262144 threads: a developed 256^3 plume notes 250-400 thousand voxels
	// the render's alpha side volume rides along on the default path (8-row tiles, far voxels deferred) of a context whose local planes
	// are the whole grid (the side volume is indexed by global voxel)
	float* alpha_out = alpha && alpha->out && defer && TY == 8 && g.nzl() == g.Zg && z_begin == 0 && z_end == g.Zg ? alpha->out : nullptr;
	if (alpha_out) {
		if (half_store) { if (p2) { FX_ADV(true, 8, true, true, true); FX_FAR(true, true, true); } else { FX_ADV(true, 8, true, false, true); FX_FAR(true, false, true); } }
		else { if (p2) { FX_ADV(false, 8, true, true, true); FX_FAR(false, true, true); } else { FX_ADV(false, 8, true, false, true); FX_FAR(false, false, true); } }
		alpha->written = true;
	} else if (!p2) {
		if (defer) {
			if (half_store) { FX_ADV(true, 8, true, false, false); FX_FAR(true, false, false); } else { FX_ADV(false, 8, true, false, false); FX_FAR(false, false, false); }
		} else { if (half_store) FX_ADV(true, 8, false, false, false); else FX_ADV(false, 8, false, false, false); }
	} else if (defer) {
		if (TY == 16) { if (half_store) FX_ADV(true, 16, true, true, false); else FX_ADV(false, 16, true, true, false); }
		else { if (half_store) FX_ADV(true, 8, true, true, false); else FX_ADV(false, 8, true, true, false); }
		if (half_store) FX_FAR(true, true, false); else FX_FAR(false, true, false);
	} else {
		if (TY == 16) { if (half_store) FX_ADV(true, 16, false, true, false); else FX_ADV(false, 16, false, true, false); }
		else { if (half_store) FX_ADV(true, 8, false, true, false); else FX_ADV(false, 8, false, true, false); }
	}
#undef FX_FAR
#undef FX_ADV
	return hipGetLastError();
}

// scratch words launch_advect_lds needs to defer far-tracing voxels for planes [z_begin, z_end) of `g` (0: no staged path there)
size_t advect_far_words(const Geom& g, int nzp)
{
	if (g.Zg <= 1 || g.X < TX || g.Y < 16 || nzp < 12) return 0;
	const size_t tiled = (size_t)((g.X + TX - 1) / TX * TX) * (size_t)((g.Y + 7) / 8 * 8);             // a plane as the 64 x 8 tiles cover it
	return 2 + (size_t)g.X * g.Y * (size_t)nzp + tiled * (size_t)(nzp + 32);   // two totals, the common list, the workgroups' segments (whole chunks)
}

}  // namespace fx
