// fx_jacobi_strip4.hip -- FOUR lock-step Jacobi sweeps per launch (X = 256 and X = 512): k_jacobi_strip3c's streaming register / LDS
// windows with the waves of a workgroup sharing a band of rows, every inner boundary an LDS mailbox.  The kernels:
//   k_jacobi_strip4o      X = 256: EIGHT waves per workgroup, two per SIMD, over a band of 14 rows (1 + 2 + 2 + 2 + 2 + 2 + 2 + 1);
//   k_jacobi_strip4x<NT>  X = 512 (round 6): the same octet on HALF-row waves -- 2 x-halves x (1 + 2 + 2 + 1 rows), a band of six, the x cut inside
//                         the workgroup; a launch's band-planes cut into one run per CU; NT: non-temporal output stores (fields beyond the cache);
//   k_jacobi_strip4q      (lab builds, STRIP4_OCTET=0) four waves, one per SIMD, over a band of 16 rows (3 + 5 + 5 + 3): the octet's predecessor;
// and the octet once more as k_freeze_strip4o: four levels of the reference's OWN loop (a cell leaves it for good once a sweep changes it by
// less than 1e-3) for every cell, the freeze nibbles carried along the windows -- the masked strip launch of the sparse solver
// (fx_schedule.cpp jacobi_freeze; k_freeze_strip3 of fx_jacobi_stripm.hip is the three-level, one-wave-per-SIMD predecessor).
//
// Restates CSPoisson.hlsli:8-26 (/root/reference/FluidX12/Content/Shaders/) like every Jacobi kernel here: the per-cell arithmetic and its
// association order, ((((((L - b) + R) + U) + D) + F) + B) * (1/6), are unchanged (relax4_pairs, fx_pk.h), so four fused sweeps are
// bit-identical to four single ones.
//
// Why bands.  k_jacobi_strip3c (fx_jacobi_strip3.hip) is a PAIR design: two 4-row strips, each recomputing its outer y-halo.  A fourth
// level in that design needs 22 row updates per z step and wave for 16 useful ones, 36 LDS rows per wave + a 24-KiB mailbox (168 KiB: more
// than the CU has) and ~400 registers.  In a band only the two OUTER waves recompute a halo (their outer side), the inner waves
// recompute nothing -- every row a wave needs across an inner boundary is its neighbour's own edge row of the previous z step, handed over
// through a 1-KiB LDS mailbox per level, direction and step parity, ordered by per-wave step counters as in k_jacobi_strip3c (wait for
// the neighbour's step q - 1, read, only then publish step q: a wave may run a step ahead of its neighbours; the rows are fetched a
// sweep EARLY and only checked at the hand-over).
//
// Why eight waves.  The quad (round 5, first half) made a z step as long as its ~570 instructions: a lone wave on a SIMD pays ~5 cycles
// per instruction whatever it is (tools/micro/issue_rate.cpp) and leaves the VALU pipe half idle.  With two waves per SIMD the same
// arithmetic runs at the fabric's rate: 256 registers per wave are enough once the inner waves take two rows and keep EVERYTHING in
// registers (the input window is a level-0 register window of three planes rotated by name with the level windows, the b planes waiting
// for sweeps 2..4 a ring of three), and the LDS (123 KiB) holds only the 84 mailbox rows of the seven inner boundaries and the outer
// waves' parked planes.  What a z step costs is its slowest wave: the outer waves take ONE row (+ halo: 4 + 3 + 2 + 1 = 10 row updates
// against 8), run at raised priority, and a 3-row inner wave does not fit 256 registers -- hence bands of 14, and the last band of a
// plane shifted up over its neighbour (the shared rows are computed twice, from the same inputs by the same arithmetic: both
// workgroups store the same bits).  256^3: 45.4-46.1 us per launch = 11.3-11.5 us per sweep (the quad: 53.1; k_jacobi_strip3c: 13.5 per
// sweep), 265 MB of fabric traffic per launch = 5.8 TB/s: the kernel is bandwidth-bound again, on a third fewer bytes per sweep.
//
// Per z step q (input plane q in registers, prefetched during step q - 1); level-l plane q - l is produced by sweep l:
//   sweep 1   from input planes q-2, q-1, q and b[q-1]; then the prefetch of plane q + 1 / b[q] is issued
//   sweep l   (2..4) from the level l-1 window (registers) and b[q-l]; the neighbours' edge rows of level l-1 come from the mailbox
//   sweep 4   is the output (stored unconditionally: a step of the first chunk whose plane lies below it writes over plane zb, which
//             the same wave stores for good four steps later)
// Traffic: p and b are read once (+ 8 halo planes per chunk of 20) and the result written once per FOUR sweeps.
#include "fx_internal.h"
#include "fx_pk.h"
#include <algorithm>
#include <climits>
#include <cstdlib>

namespace fx {

namespace {

#ifndef FX_STRIP4_OUTER_ROWS
#define FX_STRIP4_OUTER_ROWS 3
#endif
constexpr int NRO = FX_STRIP4_OUTER_ROWS;   // rows of an outer wave
constexpr int NRI = 8 - NRO;                // rows of an inner wave (a workgroup = 2 x (NRO + NRI) = 16 rows)

// A: the wave recomputes a halo ABOVE its rows (the quad's top wave), W: below (the bottom wave).
// XS (X = 512, k_jacobi_strip4x): the wave holds HALF a row -- 1: the left half (its lane 63 looks across the cut at x = 256), 2: the right
// half (its lane 0 looks at x = 255); 0: the wave is the row.
// XT (any X >= 256, k_jacobi_strip4t): the wave holds 256 cells of a LONGER row, an x tile with its own recomputed halo -- one lane (four
// cells, four sweeps) at every side that is no wall; those lanes compute along and store nothing (Strip4::keep).
template <int NR_, bool A_, bool W_, int NW_ = 4, int XS_ = 0, bool NT_ = false, bool XT_ = false> struct Role4 {
	static constexpr int NR = NR_;
	static constexpr bool XT = XT_;
	static constexpr bool NT = NT_;                                           // the output rows as non-temporal stores (store_row4nt)
	static constexpr int NW = NW_;                                            // waves per workgroup: NW - 1 inner boundaries, NW counters per level
	static constexpr int XS = XS_;
	// half-row waves: where the cut cells of levels 1..3 sit among the six a wave publishes per step (rows of the NEXT level: N2 + N3 + NR
	// = 2 + 2 + 2 for an inner wave, 3 + 2 + 1 for an outer one; pairs stay register pairs: 0 1 2 | 3 | 4 5)
	static constexpr int XO1 = 0, XO2 = (A_ || W_) ? 4 : 2, XO3 = (A_ || W_) ? 3 : 4;
	static constexpr bool A = A_, W = W_;
	static constexpr int NI = NR + (A ? 4 : 1) + (W ? 4 : 1);                 // input rows per plane; row i <-> y0 - (A ? 4 : 1) + i
	static constexpr int N1 = NR + (A ? 3 : 0) + (W ? 3 : 0);                 // level-l rows; row j <-> y0 - (A ? 4 - l : 0) + j
	static constexpr int N2 = NR + (A ? 2 : 0) + (W ? 2 : 0);
	static constexpr int N3 = NR + (A ? 1 : 0) + (W ? 1 : 0);
	static constexpr int UP = A ? 1 : 0;                                      // index shift between consecutive levels
	static constexpr int LDS_ROWS = 2 * NI + (NW == 4 ? 3 : 2) * N2;           // two input planes; three parked b planes (rows of level 2) -- two for the octet's lean waves, which keep b[q-2] in registers
};
typedef Role4<NRO, true, false> RoleTop;
typedef Role4<NRI, false, false> RoleMid;
typedef Role4<NRO, false, true> RoleBot;
constexpr int Q_LDS_ROWS = RoleTop::LDS_ROWS + 2 * RoleMid::LDS_ROWS + RoleBot::LDS_ROWS;
constexpr int Q_XROWS = 2 * 3 * 3 * 2;          // [step parity][boundary][level 1..3][0: the upper wave's row, 1: the lower wave's]
static_assert((Q_LDS_ROWS + Q_XROWS) * 1024 + 64 <= 160 * 1024, "the quad's windows must fit the CU's LDS");

#ifdef FX_O_ONLYMID
#define FX_WU(w) (((w) + 7) & 7)
#define FX_WD(w) (((w) + 1) & 7)
template <int NW> __device__ __forceinline__ constexpr int xrow(int par, int bnd, int lv, int dir) { return ((((par * NW + bnd) * 3 + lv) * 2 + dir)) * 64; }
#else
#define FX_WU(w) ((w) - 1)
#define FX_WD(w) ((w) + 1)
template <int NW> __device__ __forceinline__ constexpr int xrow(int par, int bnd, int lv, int dir) { return ((((par * (NW - 1) + bnd) * 3 + lv) * 2 + dir)) * 64; }
#endif

// the windows are native vectors, not HIP's float4 struct: a struct copied under a condition becomes a select of POINTERS into the window
// array, which keeps the whole array out of registers
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 f4(v4f v) { return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ v4f relax4q(v4f c, v4f U, v4f D, v4f F, v4f Bk, v4f bb)
{
	const float4 r = relax4_pairs(f4(c), f4(U), f4(D), f4(F), f4(Bk), f4(bb), 0.0f, true, true);
	return v4f{ r.x, r.y, r.z, r.w };
}

// ... of a HALF-row wave (XS = 1 / 2, see Role4): the lane at the cut takes the partner's cell across it, E[HI], as the `old` operand of its
// DPP shift -- built into the pair by the v_pk_mov_b32 that builds it anyway (no instruction more than relax4q)
template <int XS, int HI>
__device__ __forceinline__ v4f relax4qx(v4f c, v4f U, v4f D, v4f F, v4f Bk, v4f bb, fx_f2 E)
{
	const fx_f2 c01 = { c.x, c.y }, c23 = { c.z, c.w };
	fx_f2 lx = XS == 2 ? pk_mov_sel<HI, 0>(E, c01) : pk_mov(c01, c01, 0);            // (edge | c.x, c.x)
	const fx_f2 mid = pk_mov(c01, c23, 1);                                           // (c.y, c.z)
	fx_f2 rx = XS == 1 ? pk_mov_sel<1, HI>(c23, E) : pk_mov(c23, c23, 2);            // (c.w, edge | c.w)
	const float cx_ = c.x, cw_ = c.w, lx0_ = lx.x, rx1_ = rx.y;
	lx.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lx0_), __builtin_bit_cast(int, cw_), 0x138, 0xf, 0xf, false));
	rx.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, rx1_), __builtin_bit_cast(int, cx_), 0x130, 0xf, 0xf, false));
	const fx_f2 b01 = { bb.x, bb.y }, b23 = { bb.z, bb.w }, U01 = { U.x, U.y }, U23 = { U.z, U.w }, D01 = { D.x, D.y }, D23 = { D.z, D.w };
	const fx_f2 F01 = { F.x, F.y }, F23 = { F.z, F.w }, B01 = { Bk.x, Bk.y }, B23 = { Bk.z, Bk.w };
	fx_f2 s01 = (((((lx - b01) + mid) + U01) + D01) + F01) + B01;
	fx_f2 s23 = (((((mid - b23) + rx) + U23) + D23) + F23) + B23;
	const float inv = __uint_as_float(0x3e2aaaabu);
	s01 *= inv; s23 *= inv;
	return v4f{ s01.x, s01.y, s23.x, s23.y };
}
// ... with the cell taken from the six (EA, EB) that came in from the partner for this step: component C
template <int XS, int C>
__device__ __forceinline__ v4f relax4qe(v4f c, v4f U, v4f D, v4f F, v4f Bk, v4f bb, v4f EA, fx_f2 EB)
{
	static_assert(C >= 0 && C < 6, "six cut cells per step");
	const fx_f2 E = C < 2 ? fx_f2{ EA.x, EA.y } : C < 4 ? fx_f2{ EA.z, EA.w } : EB;
	return relax4qx<XS, C & 1>(c, U, D, F, Bk, bb, E);
}

__device__ __forceinline__ uint32_t opaque32q(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
// an output row.  FX_S4_SC1: as a write-through store (`sc0 sc1`: the line leaves the XCD's L2 instead of staying there dirty) --
// a volatile store through a global-address-space pointer is how the compiler is told (it keeps counting the store in its vmcnt waits)
typedef __attribute__((address_space(1))) v4f g_v4f;
__device__ __forceinline__ void store_row4(char* base, uint32_t off, v4f v)
{
#ifdef FX_S4_SC1
	*(volatile g_v4f*)(g_v4f*)(base + off) = v;
#elif defined(FX_S4_NT)
	__builtin_nontemporal_store(v, reinterpret_cast<v4f*>(base + off));   // (experiment: the output is not read again before the next launch)
#else
	*reinterpret_cast<v4f*>(base + off) = v;
#endif
}
// ... as a non-temporal store (Role4::NT, compile-time: behind a run-time flag the launch lost 4-7 %): fields beyond the Infinity Cache
// (p + b of 512^3 are 1 GiB) are not read again before the next launch, and lines that do not stay behind leave the L2 to the halo rows the
// bands share -- k_jacobi_strip4x at 512^3: 343.6 -> 335 us per launch in the bench, 96.2 -> 94.4 in the sweep micro-benchmark; on fields
// that fit (512 x 512 x 64: +4 %; 256^3: worse, round 5) the plain store
__device__ __forceinline__ void store_row4nt(char* base, uint32_t off, v4f v)
{
	__builtin_nontemporal_store(v, reinterpret_cast<v4f*>(base + off));
}

// ... of an x tile (Role4::XT): the lanes that are the tile's own x halo store nothing.  Not a branch around the stores (the compiler pulls
// the last sweep's arithmetic into it and splits the step into blocks: X = 256 through the tiled kernel 9 % behind the octet) but an offset
// the hardware refuses: the plane as a buffer resource of 2 GiB - 1 bytes, a halo lane's offset beyond it -- the lane's write is dropped.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store_row4xt(char* base, uint32_t off, bool keep, v4f v)
{
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
	const uint32_t o = keep ? off : 0xfffffff0u;
	__builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rs, (int)o, 0, NT ? 2 : 0);
}

// ---- the reference's own loop (k_freeze_strip4o; CSPoisson.hlsli:8-26: a cell leaves the loop for good once a sweep changes it by less
// than 1e-3): relax4m of fx_jacobi_stripm.hip on native vectors -- relax4_pairs' sum, kept for the freeze test; nib = the quad's frozen
// bits on entry, returned updated; a cell frozen on entry keeps its value
constexpr float kBelow4 = 0.00100000005f;                            // CSPoisson.hlsli:24 as compiled (0x3a83126f)
__device__ __forceinline__ v4f relax4qm(v4f c, v4f U, v4f D, v4f F, v4f Bk, v4f bb, uint32_t nib, uint32_t& nib_out)
{
	const fx_f2 c01 = { c.x, c.y }, c23 = { c.z, c.w };
	fx_f2 lx = pk_mov(c01, c01, 0);
	const fx_f2 mid = pk_mov(c01, c23, 1);
	fx_f2 rx = pk_mov(c23, c23, 2);
	// (scalars first: __builtin_bit_cast of an ELEMENT of a native vector, `c.w`, reads the vector's first element with this compiler)
	const float cx_ = c.x, cw_ = c.w, lx0_ = lx.x, rx1_ = rx.y;
	lx.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lx0_), __builtin_bit_cast(int, cw_), 0x138, 0xf, 0xf, false));
	rx.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, rx1_), __builtin_bit_cast(int, cx_), 0x130, 0xf, 0xf, false));
	const fx_f2 b01 = { bb.x, bb.y }, b23 = { bb.z, bb.w }, U01 = { U.x, U.y }, U23 = { U.z, U.w }, D01 = { D.x, D.y }, D23 = { D.z, D.w };
	const fx_f2 F01 = { F.x, F.y }, F23 = { F.z, F.w }, B01 = { Bk.x, Bk.y }, B23 = { Bk.z, Bk.w };
	const fx_f2 s01 = (((((lx - b01) + mid) + U01) + D01) + F01) + B01;
	const fx_f2 s23 = (((((mid - b23) + rx) + U23) + D23) + F23) + B23;
	const float inv = __uint_as_float(0x3e2aaaabu);
#ifdef FX_M_NOTEST
	const bool f0 = false, f1 = false, f2 = false, f3 = false;
#else
	const bool f0 = fabsf(fmaf(s01.x, inv, -c.x)) < kBelow4, f1 = fabsf(fmaf(s01.y, inv, -c.y)) < kBelow4;
	const bool f2 = fabsf(fmaf(s23.x, inv, -c.z)) < kBelow4, f3 = fabsf(fmaf(s23.y, inv, -c.w)) < kBelow4;
#endif
	fx_f2 x01 = s01, x23 = s23;
	x01 *= inv; x23 *= inv;
	nib_out = nib | (f0 ? 1u : 0u) | (f1 ? 2u : 0u) | (f2 ? 4u : 0u) | (f3 ? 8u : 0u);
	return v4f{ (nib & 1u) ? c.x : x01.x, (nib & 2u) ? c.y : x01.y, (nib & 4u) ? c.z : x23.x, (nib & 8u) ? c.w : x23.y };
}
#define FXQ_NIB(m, r) (((m) >> (4 * (r))) & 15u)

// The same update with the bookkeeping in fewer instructions (the masked kernel is bound by what it issues: 51 VALU per quad above, 39 here;
// FX_M_PLAIN builds the plain form).  mword / pos: a nibble word and the bit of the quad's first cell in it; a frozen cell is kept by a
// bit-field insert under the sign-extended flag (v_bfe_i32 + v_bfi_b32 instead of and + compare + select).  facc collects the sweep's
// freeze decisions: the compare's carry is shifted in by an add-with-carry (facc = 2 facc + vcc), cell 3 first, so that a caller that
// walks its rows from the LAST to the first ends up with row r's new bits in nibble r; the bits a cell came in with are OR-ed in by the
// caller, once per plane.  The difference |s/6 - c| of two cells comes from one v_pk_fma_f32 (fused like fmaf).
__device__ __forceinline__ void frz_shift_in4(uint32_t& facc, float t3, float t2, float t1, float t0, float below)
{
	// (one statement: between two asm statements that pass a register on the compiler puts a wait state)
	asm("v_cmp_lt_f32_e64 vcc, |%1|, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
		"v_cmp_lt_f32_e64 vcc, |%2|, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
		"v_cmp_lt_f32_e64 vcc, |%3|, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n\t"
		"v_cmp_lt_f32_e64 vcc, |%4|, %5\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc"
		: "+v"(facc) : "v"(t3), "v"(t2), "v"(t1), "v"(t0), "s"(below) : "vcc");
}
__device__ __forceinline__ float frz_keep(uint32_t mword, int bit, float c, float x)
{
	uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)mword, bit, 1);             // 0 / ~0
	asm("" : "+v"(m));                                                              // (seen through, the compiler turns the insert back into and + compare + select)
	return __uint_as_float((m & __float_as_uint(c)) | (~m & __float_as_uint(x)));
}
__device__ __forceinline__ v4f relax4qf(v4f c, v4f U, v4f D, v4f F, v4f Bk, v4f bb, uint32_t mword, int pos, uint32_t& facc)
{
	const fx_f2 c01 = { c.x, c.y }, c23 = { c.z, c.w };
	fx_f2 lx = pk_mov(c01, c01, 0);
	const fx_f2 mid = pk_mov(c01, c23, 1);
	fx_f2 rx = pk_mov(c23, c23, 2);
	const float cx_ = c.x, cw_ = c.w, lx0_ = lx.x, rx1_ = rx.y;
	lx.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, lx0_), __builtin_bit_cast(int, cw_), 0x138, 0xf, 0xf, false));
	rx.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, rx1_), __builtin_bit_cast(int, cx_), 0x130, 0xf, 0xf, false));
	const fx_f2 b01 = { bb.x, bb.y }, b23 = { bb.z, bb.w }, U01 = { U.x, U.y }, U23 = { U.z, U.w }, D01 = { D.x, D.y }, D23 = { D.z, D.w };
	const fx_f2 F01 = { F.x, F.y }, F23 = { F.z, F.w }, B01 = { Bk.x, Bk.y }, B23 = { Bk.z, Bk.w };
	const fx_f2 s01 = (((((lx - b01) + mid) + U01) + D01) + F01) + B01;
	const fx_f2 s23 = (((((mid - b23) + rx) + U23) + D23) + F23) + B23;
	const float inv = __uint_as_float(0x3e2aaaabu);
	fx_f2 t01, t23;
	const fx_f2 inv2 = { inv, inv };
	asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t01) : "v"(s01), "s"(inv2), "v"(c01));
	asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t23) : "v"(s23), "s"(inv2), "v"(c23));
	fx_f2 x01 = s01, x23 = s23;
	x01 *= inv; x23 *= inv;
	frz_shift_in4(facc, t23.y, t23.x, t01.y, t01.x, kBelow4);
	return v4f{ frz_keep(mword, pos, c.x, x01.x), frz_keep(mword, pos + 1, c.y, x01.y), frz_keep(mword, pos + 2, c.z, x23.x), frz_keep(mword, pos + 3, c.w, x23.y) };
}

// what the masked kernel's launch carries besides the three fields (one kernel argument)
struct FrzArgs {
	float* p_outB; const uint8_t* m_in; uint8_t* m_outA; uint8_t* m_outB;
	uint32_t* tile_mark; uint32_t tag; int ntx, nty;
	uint32_t* stat; uint32_t stat_hi; int level_in;
};
// ... and what a wave of it carries along z besides the pressure windows (scalars only: the nibble windows are local arrays of the run
// functions, like the pressure windows).  A nibble word holds a plane's rows in the indexing of its level, row r at bits 4 r.
struct Frz4 {
	const uint8_t* pm;                  // nibble plane walking with Strip4::pp (input plane q + 1)
	char* outA; ptrdiff_t dB;           // p_outA; p_outB - p_outA in bytes
	uint8_t* mA; uint8_t* mB;
	uint32_t* tile_mark; uint32_t tag; int ntx, nty, y_own;
	uint32_t M0c;                       // nibbles of input plane q - 1 (the centre plane of sweep 1) in level-1 row indexing
	uint32_t rel;                       // bit l: one of the wave's own cells (planes of its chunk) still relaxed after level l (l = 0: on entry)
	uint32_t act;                       // bit m: own row m has a cell that still relaxes after the last level, in the current group of 8 planes
};

// A hand-over wait is BOUNDED (a neighbour is a z step away, ~2 us; 65 536 polls are ~10 ms: a protocol error must not hang the device)
// and a wait that runs out is LOUD: it raises this word, which fx_synchronize reads behind the device (strip4_fault_take) and returns as
// FX_E_DEVICE -- the pressure field of that launch is not to be trusted (ADVICE / VERDICT round 5: it used to continue silently).
__device__ unsigned g_strip4_fault;
constexpr int kWaitSpins4 = 1 << 16;
// (inline, and a plain atomic without a return value: a CALL in these kernels would cost the windows their registers)
__device__ __forceinline__ void strip4_raise_fault() { (void)__hip_atomic_fetch_or(&g_strip4_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// counter and row in ONE LDS round trip (see k_jacobi_strip3c): a counter that is high enough vouches for the row read behind it
__device__ __forceinline__ v4f lds_wait_read4(uint32_t flag_byte_addr, int need, uint32_t row_byte_addr)
{
	int f;
	v4f d;
	for (int spins = 0;; ++spins) {
		asm volatile("ds_read_b32 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f), "=&v"(d) : "v"(flag_byte_addr), "v"(row_byte_addr) : "memory");
		if (f >= need) break;
		if (__builtin_expect(spins > kWaitSpins4, 0)) { strip4_raise_fault(); break; }
		__builtin_amdgcn_s_sleep(1);
	}
	return d;
}
__device__ __forceinline__ void lds_post4(uint32_t lds_byte_addr, int v)
{
	asm volatile("ds_write_b32 %0, %1" :: "v"(lds_byte_addr), "v"(v) : "memory");
}

// what a wave carries along z
// (the register windows are separate local arrays of run4: one aggregate of all of them is not split into registers by the compiler)
template <class R> struct Strip4 {
	int s_ctr, s_old, s_b2, s_b3, s_b4;            // LDS slots (v4f offsets into the wave's slice): input planes q-1, q-2; b[q-2], b[q-3], b[q-4]
	const char* pp; const char* pbq; char* po; char* po_zb;     // plane bases walking along with q: input plane q+1, b plane q+1, output plane q-4
	size_t plane_bytes;
	v4f* lds; v4f* xbuf; int* xflag;
	uint32_t xf0, xb0;                             // LDS byte addresses of the step counters and of the mailbox
	int q, zb, ze, q_load_last, b_load_last, Zg, wave, lane;
	bool wall_top, wall_bot;                       // the strip's first own row is y = 0 / its last own row is y = Y - 1
	bool keep;                                     // x tiles (R::XT): this lane's four cells are the tile's to store (not its x halo)
	// half-row waves (R::XS): the cells across the cut.  Level 0 (the input) is fetched by the wave itself with the plane it prefetches --
	// ONE load, lane i takes the cell across the cut of input row i -- and handed to a row's update by v_readlane; levels 1..3 travel like the
	// edge ROWS: 16 bytes per wave, level and step parity in the LDS, written by the lane at the cut, read (a broadcast) by the partner
	uint32_t eoff;                                 // byte offset of "my" input cell across the cut inside a plane
	float E0;                                      // the input cells across the cut of the CENTRE plane q - 1 (fetched during step q - 1, a step behind the plane itself: one register)
	v4f* xe;                                       // [step parity][wave] x 8 cells (six used: Role4::XO1..3)
	uint32_t xe0;                                  // ... as an LDS byte address
};
template <int NW> __device__ __forceinline__ constexpr int xeslot(int par, int w) { return (par * NW + w) * 2; }   // in v4f units: 32 bytes per slot

// The cut cells of levels 1..3 travel ONCE per step (round 6, second half: per level -- a lane-masked write, a counter check and a slot read
// each -- they cost an inner wave ~50 of its ~400 instructions per step and 7-11 % of the launch).  A wave publishes its six cells and its
// step number behind sweep 4, under a hand-made exec mask (the lane at the cut only; control flow here is wave-uniform, exec is all ones);
// the partner fetches slot and counter at the top of its next step and checks the counter in front of sweep 2.  The counter (xflag row 3)
// vouches for the slot written in front of it (a wave's LDS operations execute in order); two slots by step parity: my counter at q - 1
// says I have read what the partner wrote two steps ago.
template <int XS>
__device__ __forceinline__ void xe_publish(uint32_t slot_addr, uint32_t cnt_addr, int q, float v0, float v1, float v2, float v3, float v4, float v5)
{
	if (XS == 1)
		asm volatile("s_bfm_b64 exec, 1, 63\n\tds_write2_b32 %0, %2, %3 offset1:1\n\tds_write2_b32 %0, %4, %5 offset0:2 offset1:3\n\tds_write2_b32 %0, %6, %7 offset0:4 offset1:5\n\t"
			"ds_write_b32 %1, %8\n\ts_mov_b64 exec, -1" :: "v"(slot_addr), "v"(cnt_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(q) : "memory");
	else
		asm volatile("s_mov_b64 exec, 1\n\tds_write2_b32 %0, %2, %3 offset1:1\n\tds_write2_b32 %0, %4, %5 offset0:2 offset1:3\n\tds_write2_b32 %0, %6, %7 offset0:4 offset1:5\n\t"
			"ds_write_b32 %1, %8\n\ts_mov_b64 exec, -1" :: "v"(slot_addr), "v"(cnt_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(q) : "memory");
}
struct XMail { int f; v4f a; fx_f2 b; };
template <class R>
__device__ __forceinline__ void xe_fetch(const Strip4<R>& st, XMail& x)
{
#ifdef FX_S4_NOXE
	x.f = INT_MAX; return;                                              // (timing experiment: the cut's cells never travel; results are wrong)
#endif
	const int wp = st.wave ^ 4, pr = (st.q - 1) & 1;
	x.f = __hip_atomic_load(reinterpret_cast<const int*>(st.xflag) + 3 * R::NW + wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
	__builtin_amdgcn_sched_barrier(0);
	x.a = st.xe[xeslot<R::NW>(pr, wp)];
	x.b = *reinterpret_cast<const fx_f2*>(st.xe + xeslot<R::NW>(pr, wp) + 1);
	asm volatile("" ::: "memory");
	__builtin_amdgcn_sched_barrier(0);
}
template <class R>
__device__ __forceinline__ void xe_check(const Strip4<R>& st, XMail& x)
{
	if (__builtin_expect(x.f < st.q - 1, 0)) {                          // the partner has not published its step q - 1 yet
		const int wp = st.wave ^ 4, pr = (st.q - 1) & 1;
		x.a = lds_wait_read4(st.xf0 + 4u * (uint32_t)(3 * R::NW + wp), st.q - 1, st.xe0 + 16u * (uint32_t)xeslot<R::NW>(pr, wp));
		x.b = *reinterpret_cast<const fx_f2*>(st.xe + xeslot<R::NW>(pr, wp) + 1);
	}
}
__device__ __forceinline__ float lane_cell(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

#define FXQ_LDS(st, slot, r) (st).lds[(slot) + (r) * 64]
// The plane behind the last one of the field is a copy of the last (clamped neighbour): once per field and level.  Written as
// `if (top) copy; else relax;` the compiler turns it into relax + four v_cndmask per row on EVERY step (24 of an inner wave's 239 VALU
// per z step); the relaxation now runs unconditionally (register and LDS operands only) and the copy sits behind a branch that an asm
// statement keeps from being turned into selects again.
#define FXQ_RARE_BRANCH asm volatile("; rare path")

// Hand-over of level L (1..3).  The neighbours' edge rows of THEIR step q - 1 (plane q - 1 - L: the centre plane of this step's sweep
// L + 1) come in, mine of this step (plane q - L) go out.  Order per level: read the neighbours' rows, make sure they were there
// (their counter), only then publish mine and my counter -- a wave that sees my counter at q knows I have read what it wrote two steps
// ago into the slot it is about to reuse, so two slots (step parity) suffice and a wave may run ahead of its neighbours.
// The read is ISSUED one sweep early (mail_fetch4: level 1 at the top of the step, level L behind hand-over L - 1) as compiler-visible
// LDS loads and only checked here: issued where it is needed, each hand-over is an exposed LDS round trip per neighbour -- six per step
// for an inner wave, 12.5 us of a 58-us launch (measured by leaving the hand-overs out).  A row fetched before its owner had published
// it (its counter says so) is fetched again by the waiting loop.
template <class R> struct Mail4 { int fu, fd; v4f hu, hd; };

template <class R, int L>
__device__ __forceinline__ void mail_fetch4(const Strip4<R>& st, Mail4<R>& m)
{
#ifdef FX_S4_NOHAND
	return;
#endif
	const int q = st.q, w = st.wave, pr = (q - 1) & 1;
#ifdef FX_S4_BARRIER
	if (!R::A) m.hu = st.xbuf[xrow<R::NW>(pr, FX_WU(w), L - 1, 0) + st.lane];     // (behind the step's barrier every neighbour's rows of step q - 1 are there)
	if (!R::W) m.hd = st.xbuf[xrow<R::NW>(pr, w, L - 1, 1) + st.lane];
	m.fu = m.fd = INT_MAX;
	return;
#endif
	const int* flags = reinterpret_cast<const int*>(st.xflag);
	// counter first, row behind it: a wave's LDS operations execute in order, so a counter that is high enough vouches for the row
	if (!R::A) m.fu = __hip_atomic_load(flags + (L - 1) * R::NW + FX_WU(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	if (!R::W) m.fd = __hip_atomic_load(flags + (L - 1) * R::NW + FX_WD(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	asm volatile("" ::: "memory");
	__builtin_amdgcn_sched_barrier(0);
	if (!R::A) m.hu = st.xbuf[xrow<R::NW>(pr, FX_WU(w), L - 1, 0) + st.lane];
	if (!R::W) m.hd = st.xbuf[xrow<R::NW>(pr, w, L - 1, 1) + st.lane];
	asm volatile("" ::: "memory");
	__builtin_amdgcn_sched_barrier(0);                                  // (the machine scheduler would otherwise sink the row loads to their use)
}

// (PN: the level's new plane, NL rows; its first and last row are the edge rows)
template <class R, int L, int NL>
__device__ __forceinline__ void hand_over4(const Strip4<R>& st, Mail4<R>& m, const v4f (&PN)[NL], v4f& HU, v4f& HD)
{
#ifdef FX_S4_NOHAND
	return;
#endif
	const int q = st.q, w = st.wave;
	const int pr = (q - 1) & 1, pw = q & 1;
	const v4f mine_top = PN[0], mine_bot = PN[NL - 1];
	if (!R::A) {
		if (__builtin_expect(m.fu < q - 1, 0)) m.hu = lds_wait_read4(st.xf0 + 4u * (uint32_t)((L - 1) * R::NW + FX_WU(w)), q - 1, st.xb0 + 16u * (uint32_t)(xrow<R::NW>(pr, FX_WU(w), L - 1, 0) + st.lane));
		HU = m.hu;
	}
	if (!R::W) {
		if (__builtin_expect(m.fd < q - 1, 0)) m.hd = lds_wait_read4(st.xf0 + 4u * (uint32_t)((L - 1) * R::NW + FX_WD(w)), q - 1, st.xb0 + 16u * (uint32_t)(xrow<R::NW>(pr, w, L - 1, 1) + st.lane));
		HD = m.hd;
	}
	asm volatile("" ::: "memory");
	if (!R::A) st.xbuf[xrow<R::NW>(pw, FX_WU(w), L - 1, 1) + st.lane] = mine_top;
	if (!R::W) st.xbuf[xrow<R::NW>(pw, w, L - 1, 0) + st.lane] = mine_bot;
#ifndef FX_S4_BARRIER
#ifdef FX_LAB_DROP_PUBLISH
	if (L == 2 && w == 3) return;                                       // (fault injection, tests/test_gpu_faults.py: wave 3 never posts level 2 -- its neighbours' waits must run out LOUDLY)
#endif
	if (st.lane == 0) lds_post4(st.xf0 + 4u * (uint32_t)((L - 1) * R::NW + w), q);        // LDS operations of a wave execute in order
#endif
}

// sweep L + 1 from the window of level L (NL rows) into NN rows; b rows from `Brow`
// (MK: the masked loop -- mctr = the nibbles of the centre plane in level-L indexing, mout = those of the new plane in level-(L+1) indexing)
template <class R, int L, int NL, int NN, bool MK = false>
__device__ __forceinline__ void relax_level4(const Strip4<R>& st, const v4f (&Pold)[NL], const v4f (&Pctr)[NL], const v4f (&Pnew)[NL],
	const v4f (&Bq)[NN], v4f HU, v4f HD, v4f (&out)[NN], uint32_t mctr, uint32_t& mout, const XMail* xm = nullptr)
{
	constexpr int EOFF = L == 1 ? R::XO1 : L == 2 ? R::XO2 : R::XO3;     // where this level's cut cells sit among the partner's six
	static_assert(!(MK && R::XS), "the masked loop runs on whole-row waves");
	uint32_t m_ = 0u;
#pragma unroll
	for (int kk = 0; kk < NN; ++kk) {
		constexpr int UP = R::UP;
#ifdef FX_M_PLAIN
		const int k = kk;
#else
		const int k = MK ? NN - 1 - kk : kk;                             // (the masked loop: last row first, see relax4qf)
#endif
		const int jc = k + UP;                                           // level-L index of this row
		const v4f c = Pctr[jc];
		v4f u = jc >= 1 ? Pctr[jc >= 1 ? jc - 1 : 0] : HU;
		v4f d = jc + 1 < NL ? Pctr[jc + 1 < NL ? jc + 1 : 0] : HD;
		if (R::A && k == 3 - L && st.wall_top) u = c;                    // rows outside the domain hold no data
		if (R::W && k == R::NR - 1 && st.wall_bot) d = c;
		if (MK) {
#ifdef FX_M_PLAIN
			uint32_t n_;
			out[k] = relax4qm(c, u, d, Pold[jc], Pnew[jc], Bq[k], FXQ_NIB(mctr, jc), n_);
			m_ |= n_ << (4 * k);
#else
			out[k] = relax4qf(c, u, d, Pold[jc], Pnew[jc], Bq[k], mctr, 4 * jc, m_);
#endif
		} else if (R::XS) {
			// (k is a constant of the unrolled loop; the switch only names it for the template)
			switch (k) {
			case 0: out[k] = relax4qe<R::XS ? R::XS : 1, (EOFF + 0) % 6>(c, u, d, Pold[jc], Pnew[jc], Bq[k], xm->a, xm->b); break;
			case 1: out[k] = relax4qe<R::XS ? R::XS : 1, (EOFF + 1) % 6>(c, u, d, Pold[jc], Pnew[jc], Bq[k], xm->a, xm->b); break;
			default: out[k] = relax4qe<R::XS ? R::XS : 1, (EOFF + 2) % 6>(c, u, d, Pold[jc], Pnew[jc], Bq[k], xm->a, xm->b); break;
			}
		} else
			out[k] = relax4q(c, u, d, Pold[jc], Pnew[jc], Bq[k]);
	}
#ifdef FX_M_PLAIN
	if (MK) mout = m_;
#else
	if (MK) mout = ((mctr >> (4 * R::UP)) & ((1u << (4 * NN)) - 1u)) | m_;   // the bits the rows came in with + this sweep's
#endif
}

// The masked loop's output plane q - 4 besides its pressure rows (stored by the caller to p_outA): the same rows to p_outB, the nibbles to
// both mask buffers (the tile launches alternate between two buffers and expect the tiles they do not list to agree in them), and the
// tile marks: a 32 x 8 x 8 tile with a cell that still relaxes gets `tag`, one plain store per tile behind the last plane of its group of
// eight (or of the chunk).  A wave's two rows may lie in two tile rows, so the activity is kept per row.
template <class R>
__device__ __forceinline__ void frz_out4(const Strip4<R>& st, Frz4& fz, char* dst_, const v4f (&X_)[R::NR], uint32_t m4_, const uint32_t (&roff)[R::NI])
{
	constexpr int NR = R::NR, RB = R::A ? 4 : 1;
	const int z = st.q - 4;
	const ptrdiff_t mo_ = (dst_ - fz.outA) >> 4;                         // a plane of nibble bytes is a sixteenth of a pressure plane
#ifndef FX_M_ONEP
#pragma unroll
	for (int m = 0; m < NR; ++m) store_row4(dst_ + fz.dB, opaque32q(roff[m + RB]), X_[m]);
#endif
#ifdef FX_M_NOBYTES
	if (st.q < -1000)
#endif
#pragma unroll
	for (int m = 0; m < NR; ++m) {
		const uint32_t o_ = opaque32q(roff[m + RB] >> 4);
		fz.mA[mo_ + o_] = (uint8_t)FXQ_NIB(m4_, m);
		fz.mB[mo_ + o_] = (uint8_t)FXQ_NIB(m4_, m);
	}
	if (z >= st.zb) {                                                    // (a step below the chunk has written over plane zb, see step4: nothing of it counts)
#pragma unroll
		for (int m = 0; m < NR; ++m) if (FXQ_NIB(m4_, m) != 15u) { fz.rel |= 16u; fz.act |= 1u << m; }
		if ((z & 7) == 7 || z == st.ze - 1) {
#pragma unroll
			for (int m = 0; m < NR; ++m) {
				const unsigned long long bal_ = __ballot((fz.act >> m) & 1u);
				if ((st.lane & 7) == 0 && ((bal_ >> st.lane) & 0xFFull) != 0ull)
					fz.tile_mark[((z >> 3) * fz.nty + ((fz.y_own + m) >> 3)) * fz.ntx + (st.lane >> 3)] = fz.tag;
			}
			fz.act = 0u;
		}
	}
}

// MK: the masked loop (k_freeze_strip4o) -- the nibble windows MW[level 1..3][slot] rotate with the pressure windows, NM = the nibble
// bytes of the plane in flight (rows of level 1), fz the rest; without MK none of them is touched
template <class R, int PH, bool S1, bool S2, bool S3, bool S4, bool MK = false>
__device__ __forceinline__ void step4(Strip4<R>& st, v4f (&P1)[3][R::N1], v4f (&P2)[3][R::N2], v4f (&P3)[3][R::N3], v4f (&NP)[R::NI], v4f (&NB)[R::N1], v4f (&NBn)[R::N1], v4f (&Bk)[R::N2], const uint32_t (&roff)[R::NI],
	Frz4& fz, uint32_t (&MW)[3][3], uint32_t (&NM)[R::N1])
{
	constexpr int NEW = PH % 3, CTR = (PH + 2) % 3, OLD = (PH + 1) % 3;
	constexpr int NI = R::NI, N1 = R::N1, N2 = R::N2, N3 = R::N3, NR = R::NR, UP = R::UP;
	static_assert(!MK || R::NW != 4, "the masked loop runs in the octet");
	constexpr int O1 = R::A ? 3 : 0, O2 = R::A ? 2 : 0, O3 = R::A ? 1 : 0;      // the first own row in the indexing of levels 1..3
	constexpr uint32_t OWN = (1u << (4 * NR)) - 1u;
	const int q = st.q;
	uint32_t M0n = 0u;
	if (MK) {                                                           // the input plane's nibbles have arrived with it
#pragma unroll
		for (int j = 0; j < N1; ++j) M0n |= (NM[j] & 15u) << (4 * j);
		if (q >= st.zb && q < st.ze && (~(M0n >> (4 * O1)) & OWN) != 0u) fz.rel |= 1u;
		if (q == 0) fz.M0c = M0n;
	}
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	Mail4<R> M1, M2, M3;
	M1.fu = M1.fd = M2.fu = M2.fd = M3.fu = M3.fd = INT_MIN; M1.hu = M1.hd = M2.hu = M2.hd = M3.hu = M3.hd = zero;
#ifdef FX_S4_BARRIER
	// one workgroup barrier per z step instead of the per-level counters: behind it every wave has published its rows of step q - 1
	// (its LDS writes are complete: lgkmcnt(0); the prefetch and the output stores in flight are NOT waited for)
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
	if (S1) mail_fetch4<R, 1>(st, M1);
	XMail xm; xm.f = INT_MIN; xm.a = zero; xm.b = fx_f2{ 0.0f, 0.0f };
	if (R::XS && S2) xe_fetch<R>(st, xm);
	// EARLY (the quad): the plane and the b rows that arrived go into their own registers and the NEXT prefetch is issued at once, a whole
	// step ahead of its use; b is fetched two steps ahead into NBn (L - b opens every cell's sum).  LEAN (the octet's outer waves, two waves
	// per SIMD: every register counts, the other wave of the SIMD covers the waits): no copies, no second b buffer, the prefetch goes out
	// behind sweep 1 into the registers that sweep has just read.
#ifdef FX_S4_LATE_PREFETCH
	constexpr bool EARLY = false;
#else
	constexpr bool EARLY = R::NW == 4;
#endif
	constexpr bool LEAN = R::NW != 4;
	v4f NPc[EARLY ? NI : 1], NBc[EARLY ? N1 : 1];
	auto issue_prefetch = [&]() __attribute__((always_inline)) {
		if (!LEAN) {
#pragma unroll
			for (int j = 0; j < N1; ++j) NB[j] = NBn[j];
		}
#ifdef FX_S4_NOLOAD
		if (q + 1 <= st.b_load_last && q < -1000) {
#else
		if (LEAN ? q <= st.b_load_last : q + 1 <= st.b_load_last) {
#endif
#pragma unroll
			for (int j = 0; j < N1; ++j) (LEAN ? NB[j] : NBn[j]) = *reinterpret_cast<const v4f*>(st.pbq + opaque32q(roff[j + 1]));
		}
#ifdef FX_S4_NOLOAD
		if (q + 1 <= st.q_load_last && q < -1000) {
#else
		if (q + 1 <= st.q_load_last) {
#endif
#pragma unroll
			for (int i = 0; i < NI; ++i) NP[i] = *reinterpret_cast<const v4f*>(st.pp + opaque32q(roff[i]));
			if (MK) {
#pragma unroll
				for (int j = 0; j < N1; ++j) NM[j] = fz.pm[opaque32q(roff[j + 1] >> 4)];
			}
		}
		// the cells across the cut of plane q, the next step's centre (past the last present plane E0 keeps plane zhi's)
		if (R::XS && q <= st.q_load_last) st.E0 = *reinterpret_cast<const float*>(st.pp - st.plane_bytes + st.eoff);
	};
	if (EARLY) {
#pragma unroll
		for (int i = 0; i < NI; ++i) NPc[i] = NP[i];
#pragma unroll
		for (int j = 0; j < N1; ++j) NBc[j] = NB[j];
		issue_prefetch();
	}
	auto np = [&](int i) __attribute__((always_inline)) -> v4f { return EARLY ? NPc[EARLY ? i : 0] : NP[i]; };
	auto nb = [&](int j) __attribute__((always_inline)) -> v4f { return EARLY ? NBc[EARLY ? j : 0] : NB[j]; };
	// ---- sweep 1: level-1 plane q-1 -----------------------------------------------------------------------------------
	if (q == 0) {                                                       // input plane -1 := plane 0, once (clamped front neighbour)
#pragma unroll
		for (int i = 0; i < NI; ++i) FXQ_LDS(st, st.s_ctr, i) = np(i);
	}
	// the centre plane's cells across the cut, rows of level 1 (row j's centre is input row j + 1): pairs for relax4qx
	fx_f2 E1_[R::XS ? (N1 + 1) / 2 : 1];
	if (R::XS) {
#pragma unroll
		for (int j = 0; j < N1; ++j) { const float e_ = lane_cell(st.E0, j + 1); if (j & 1) E1_[j >> 1].y = e_; else E1_[j >> 1].x = e_; }
	}
	if (S1) {
		// (the new plane goes through a local first and into the window by unconditional stores: stores to different window slots in the two
		// arms of a branch are merged by the compiler into one store through a selected POINTER, which keeps those slots in scratch)
		v4f T_[N1];
		uint32_t m1_ = 0u;
		{
			if (!LEAN) {
				v4f C_[NI], F_[N1];                                       // all LDS rows first: one wave per SIMD cannot hide a ds_read next to its use
#pragma unroll
				for (int i = 0; i < NI; ++i) C_[i] = FXQ_LDS(st, st.s_ctr, i);
#pragma unroll
				for (int j = 0; j < N1; ++j) F_[j] = FXQ_LDS(st, st.s_old, j + 1);
#pragma unroll
				for (int j = 0; j < N1; ++j) T_[j] = relax4q(C_[j + 1], C_[j], C_[j + 2], F_[j], np(j + 1), nb(j));
			} else {                                                      // lean: a sliding window of three rows (the SIMD's other wave covers the LDS latency; registers are what is short)
#ifndef FX_M_PLAIN
				if (MK) {                                                  // the masked loop walks the window upwards (last row first, see relax4qf)
					v4f d_ = FXQ_LDS(st, st.s_ctr, N1 + 1), c_ = FXQ_LDS(st, st.s_ctr, N1);
#pragma unroll
					for (int j = N1 - 1; j >= 0; --j) {
						const v4f u_ = FXQ_LDS(st, st.s_ctr, j), f_ = FXQ_LDS(st, st.s_old, j + 1);
						T_[j] = relax4qf(c_, u_, d_, f_, np(j + 1), nb(j), fz.M0c, 4 * j, m1_);
						d_ = c_; c_ = u_;
					}
					m1_ |= fz.M0c & ((1u << (4 * N1)) - 1u);
				} else
#endif
				{
				v4f u_ = FXQ_LDS(st, st.s_ctr, 0), c_ = FXQ_LDS(st, st.s_ctr, 1);
#pragma unroll
				for (int j = 0; j < N1; ++j) {
					const v4f d_ = FXQ_LDS(st, st.s_ctr, j + 2), f_ = FXQ_LDS(st, st.s_old, j + 1);
					if (MK) {
						uint32_t n_;
						T_[j] = relax4qm(c_, u_, d_, f_, np(j + 1), nb(j), FXQ_NIB(fz.M0c, j), n_);
						m1_ |= n_ << (4 * j);
					} else if (R::XS) {
						T_[j] = (j & 1) ? relax4qx<R::XS ? R::XS : 1, 1>(c_, u_, d_, f_, np(j + 1), nb(j), E1_[R::XS ? j >> 1 : 0])
						                : relax4qx<R::XS ? R::XS : 1, 0>(c_, u_, d_, f_, np(j + 1), nb(j), E1_[R::XS ? j >> 1 : 0]);
					} else
						T_[j] = relax4q(c_, u_, d_, f_, np(j + 1), nb(j));
					u_ = c_; c_ = d_;
#ifdef FX_O_SCHEDBAR
					__builtin_amdgcn_sched_barrier(0);
#endif
				}
				}
				if (MK && q - 1 >= st.zb && q - 1 < st.ze && (~(m1_ >> (4 * O1)) & OWN) != 0u) fz.rel |= 2u;
			}
		}
		if (__builtin_expect(q - 1 == st.Zg, 0)) {                         // level-1 plane Zg := plane Zg-1
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int j = 0; j < N1; ++j) T_[j] = P1[CTR][j];
			if (MK) m1_ = MW[0][CTR];
		}
#pragma unroll
		for (int j = 0; j < N1; ++j) P1[NEW][j] = T_[j];
		if (MK) MW[0][NEW] = m1_;
		if (__builtin_expect(q - 1 == 0, 0)) {                                               // level-1 plane -1 := plane 0
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int j = 0; j < N1; ++j) P1[CTR][j] = T_[j];
			if (MK) MW[0][CTR] = m1_;
		}
	}
	if (MK) fz.M0c = M0n;                                               // plane q is the next step's centre
	// b[q-4] (the output rows) leaves its slot before b[q-1] moves in; b[q-2] (rows of level 2) is wanted next
	v4f B4_[NR], B2_[N2];
#pragma unroll
	for (int m = 0; m < NR; ++m) B4_[m] = FXQ_LDS(st, st.s_b4, m + 2 * UP);
	if (!LEAN) {
#pragma unroll
		for (int k = 0; k < N2; ++k) B2_[k] = FXQ_LDS(st, st.s_b2, k);
	} else {                                                            // lean: b[q-2] waited in registers; it goes to the LDS now (over b[q-4]), b[q-1] takes its place
#pragma unroll
		for (int k = 0; k < N2; ++k) B2_[k] = Bk[k];
	}
	// the plane in flight moves to the LDS (over input plane q-2, dead now); the rows of b[q-1] later levels need over b[q-4]
#pragma unroll
	for (int i = 0; i < NI; ++i) FXQ_LDS(st, st.s_old, i) = np(i);
	if (!LEAN) {
#pragma unroll
		for (int k = 0; k < N2; ++k) FXQ_LDS(st, st.s_b4, k) = nb(k + UP);
		{ const int t_ = st.s_b4; st.s_b4 = st.s_b3; st.s_b3 = st.s_b2; st.s_b2 = t_; }     // after this: s_b2 = b[q-1], s_b3 = b[q-2], s_b4 = b[q-3]
	} else {
#pragma unroll
		for (int k = 0; k < N2; ++k) FXQ_LDS(st, st.s_b4, k) = B2_[k];
#pragma unroll
		for (int k = 0; k < N2; ++k) Bk[k] = nb(k + UP);
		{ const int t_ = st.s_b4; st.s_b4 = st.s_b3; st.s_b3 = t_; }                       // two slots: s_b3 = b[q-2] (just written), s_b4 = b[q-3]
	}
	{ const int t_ = st.s_old; st.s_old = st.s_ctr; st.s_ctr = t_; }
	if (!EARLY) issue_prefetch();
	st.pp += st.plane_bytes; st.pbq += st.plane_bytes;
	if (MK) fz.pm += st.plane_bytes >> 4;
	// hand-over 1, BEHIND the prefetch issue: a wait here must not delay the loads
	v4f HU1 = zero, HD1 = zero;
	if (S1) hand_over4<R, 1, N1>(st, M1, P1[NEW], HU1, HD1);
	if (R::XS && S2) xe_check<R>(st, xm);
	if (S2) mail_fetch4<R, 2>(st, M2);
	// ---- sweep 2: level-2 plane q-2 -----------------------------------------------------------------------------------
	if (S2) {
		v4f T_[N2];
		uint32_t m2_ = 0u;
		relax_level4<R, 1, N1, N2, MK>(st, P1[OLD], P1[CTR], P1[NEW], B2_, HU1, HD1, T_, MK ? MW[0][CTR] : 0u, m2_, &xm);
		if (MK && q - 2 >= st.zb && q - 2 < st.ze && (~(m2_ >> (4 * O2)) & OWN) != 0u) fz.rel |= 4u;
		if (__builtin_expect(q - 2 == st.Zg, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int k = 0; k < N2; ++k) T_[k] = P2[CTR][k];
			if (MK) m2_ = MW[1][CTR];
		}
#pragma unroll
		for (int k = 0; k < N2; ++k) P2[NEW][k] = T_[k];
		if (MK) MW[1][NEW] = m2_;
		if (__builtin_expect(q - 2 == 0, 0)) {                                               // level-2 plane -1 := plane 0
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int k = 0; k < N2; ++k) P2[CTR][k] = T_[k];
			if (MK) MW[1][CTR] = m2_;
		}
	}
	v4f B3_[N3];                                                     // b[q-3] (after the rotation: s_b4), rows of level 3
#pragma unroll
	for (int m = 0; m < N3; ++m) B3_[m] = FXQ_LDS(st, st.s_b4, m + UP);
	v4f HU2 = zero, HD2 = zero;
	if (S2) hand_over4<R, 2, N2>(st, M2, P2[NEW], HU2, HD2);
	if (S3) mail_fetch4<R, 3>(st, M3);
	// ---- sweep 3: level-3 plane q-3 -----------------------------------------------------------------------------------
	if (S3) {
		v4f T_[N3];
		uint32_t m3_ = 0u;
		relax_level4<R, 2, N2, N3, MK>(st, P2[OLD], P2[CTR], P2[NEW], B3_, HU2, HD2, T_, MK ? MW[1][CTR] : 0u, m3_, &xm);
		if (MK && q - 3 >= st.zb && q - 3 < st.ze && (~(m3_ >> (4 * O3)) & OWN) != 0u) fz.rel |= 8u;
		if (__builtin_expect(q - 3 == st.Zg, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int m = 0; m < N3; ++m) T_[m] = P3[CTR][m];
			if (MK) m3_ = MW[2][CTR];
		}
#pragma unroll
		for (int m = 0; m < N3; ++m) P3[NEW][m] = T_[m];
		if (MK) MW[2][NEW] = m3_;
		if (__builtin_expect(q - 3 == 0, 0)) {                                               // level-3 plane -1 := plane 0
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int m = 0; m < N3; ++m) P3[CTR][m] = T_[m];
			if (MK) MW[2][CTR] = m3_;
		}
	}
	v4f HU3 = zero, HD3 = zero;
	if (S3) hand_over4<R, 3, N3>(st, M3, P3[NEW], HU3, HD3);
	// ---- sweep 4: output plane q-4 ------------------------------------------------------------------------------------
	if (S4) {
		// UNCONDITIONAL stores: behind a branch the compiler counts no store when it waits for the prefetched rows of the next step, and each
		// of those waits then drains a store as well.  A step whose output plane lies below the chunk (the first steps of a chunk that starts at
		// the first present plane) writes its rows over plane zb instead, which this wave stores for good a few steps later (the stores of a
		// wave to one address keep their order); no step of the loop lies above the chunk (q <= ze + 3).
		v4f X_[NR];
		uint32_t m4_ = 0u;
		relax_level4<R, 3, N3, NR, MK>(st, P3[OLD], P3[CTR], P3[NEW], B4_, HU3, HD3, X_, MK ? MW[2][CTR] : 0u, m4_, &xm);
		char* dst_ = q - 4 >= st.zb ? st.po : st.po_zb;
#ifdef FX_S4_NOSTORE
		if (q < -1000)                                                   // (experiment: the arithmetic stays, the stores never execute)
#endif
#pragma unroll
		for (int m = 0; m < NR; ++m) {
			if (R::XT) store_row4xt<R::NT>(dst_, opaque32q(roff[m + (R::A ? 4 : 1)]), st.keep, X_[m]);
			else if (R::NT) store_row4nt(dst_, opaque32q(roff[m + (R::A ? 4 : 1)]), X_[m]);
			else store_row4(dst_, opaque32q(roff[m + (R::A ? 4 : 1)]), X_[m]);
		}
		if (MK) frz_out4<R>(st, fz, dst_, X_, m4_, roff);
	}
#ifndef FX_S4_NOXE
	if (R::XS) {                                                        // my cut cells of this step's new planes (levels 1..3), for the partner's next step
		float v_[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
		for (int k = 0; k < N2; ++k) v_[R::XO1 + k] = R::XS == 1 ? P1[NEW][k + UP].w : P1[NEW][k + UP].x;
#pragma unroll
		for (int k = 0; k < N3; ++k) v_[R::XO2 + k] = R::XS == 1 ? P2[NEW][k + UP].w : P2[NEW][k + UP].x;
#pragma unroll
		for (int k = 0; k < NR; ++k) v_[R::XO3 + k] = R::XS == 1 ? P3[NEW][k + UP].w : P3[NEW][k + UP].x;
		xe_publish<R::XS ? R::XS : 1>(st.xe0 + 16u * (uint32_t)xeslot<R::NW>(q & 1, st.wave), st.xf0 + 4u * (uint32_t)(3 * R::NW + st.wave), q, v_[0], v_[1], v_[2], v_[3], v_[4], v_[5]);
	}
#endif
	st.po += st.plane_bytes;
	++st.q;
}

// In front of the loop: as many stores as a step of the loop issues, to plane zb (which the wave stores for good four steps later; stores of
// a wave to one address keep their order).  Why: the compiler's `s_waitcnt vmcnt(N)` in front of the first use of a prefetched row counts
// the YOUNGER operations that may stay in flight -- a step's output stores -- and at the loop's head it takes the smaller count of the two
// ways in.  Coming from the prologue there were none, so every third step (the head of the three-phase loop) drained the previous step's
// stores before it touched its rows: vmcnt(1), vmcnt(0) where the other two phases say vmcnt(3), vmcnt(2).  (41.3 -> 40.9 us per launch:
// the stores' cost is the fabric's, not this wait.)
template <class R, bool MK>
__device__ __forceinline__ void head_stores4(const Strip4<R>& st, const Frz4& fz, const uint32_t (&roff)[R::NI])
{
#ifndef FX_S4_NOHEADSTORES
	constexpr int RB = R::A ? 4 : 1;
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
	for (int m = 0; m < R::NR; ++m) {
		if (R::XT) store_row4xt<false>(st.po_zb, opaque32q(roff[m + RB]), st.keep, zero);
		else store_row4(st.po_zb, opaque32q(roff[m + RB]), zero);
	}
	if (MK) {
		const ptrdiff_t mo_ = (st.po_zb - fz.outA) >> 4;
#pragma unroll
		for (int m = 0; m < R::NR; ++m) store_row4(st.po_zb + fz.dB, opaque32q(roff[m + RB]), zero);
#pragma unroll
		for (int m = 0; m < R::NR; ++m) {
			const uint32_t o_ = opaque32q(roff[m + RB] >> 4);
			fz.mA[mo_ + o_] = (uint8_t)0;
			fz.mB[mo_ + o_] = (uint8_t)0;
		}
	}
#endif
}

// the masked loop's set-up shared by both walks: the nibble bytes of the first plane in flight (q0), and of the plane before it (the
// centre of the first sweep) where the walk starts inside the field
template <class R, int NM_>
__device__ __forceinline__ void frz_begin4(const Geom& g, const FrzArgs& fa, const Strip4<R>& st, Frz4& fz, uint32_t (&MW)[3][3], uint32_t (&NM)[NM_],
	const uint32_t (&roff)[R::NI], float* p_out, int y0, int q0, bool fill)
{
	const size_t plane4 = st.plane_bytes >> 4;
	fz.outA = reinterpret_cast<char*>(p_out); fz.dB = reinterpret_cast<char*>(fa.p_outB) - reinterpret_cast<char*>(p_out);
	fz.mA = fa.m_outA; fz.mB = fa.m_outB; fz.tile_mark = fa.tile_mark; fz.tag = fa.tag; fz.ntx = fa.ntx; fz.nty = fa.nty; fz.y_own = y0;
	fz.M0c = 0u; fz.rel = 0u; fz.act = 0u;
#pragma unroll
	for (int l = 0; l < 3; ++l) MW[l][0] = MW[l][1] = MW[l][2] = 0u;
	const uint8_t* mb = fa.m_in + (size_t)g.lz(min(q0, st.q_load_last)) * plane4;
#pragma unroll
	for (int j = 0; j < NM_; ++j) NM[j] = mb[roff[j + 1] >> 4];
	if (fill) {
		const uint8_t* mc = fa.m_in + (size_t)g.lz(q0 - 1) * plane4;
#pragma unroll
		for (int j = 0; j < NM_; ++j) fz.M0c |= ((uint32_t)mc[roff[j + 1] >> 4] & 15u) << (4 * j);
	}
	fz.pm = fa.m_in + ((ptrdiff_t)g.lz(q0) + 1) * (ptrdiff_t)plane4;
}
// ... and its end: the last level that left one of the wave's own cells relaxing (level_in itself: a cell that came in relaxing), into
// fx_jacobi_freeze.hip's statistics word
__device__ __forceinline__ void frz_end4(const FrzArgs& fa, const Frz4& fz, int lane)
{
	const int lvl = __any(fz.rel & 16u) ? 4 : __any(fz.rel & 8u) ? 3 : __any(fz.rel & 4u) ? 2 : __any(fz.rel & 2u) ? 1 : __any(fz.rel & 1u) ? 0 : -1;
	if (lane == 0 && lvl >= 0) atomicMax(fa.stat, fa.stat_hi + (uint32_t)(fa.level_in + lvl));
}

template <class R, bool MK = false>
__device__ __forceinline__ void run4(const Geom& g, const float* __restrict__ p_in, const float* __restrict__ b, float* __restrict__ p_out,
	int zb, int ze, int y0, int wave, int lane, v4f* lds_slice, v4f* xbuf, int* xflag, const FrzArgs& fa, v4f* xe = nullptr, int x0 = 0, bool keep = true)
{
	Strip4<R> st;
	Frz4 fz;
	uint32_t MW[3][3], NM[R::N1];
#ifndef FX_O_NOPRIO
#ifndef FX_O_PRIO
#define FX_O_PRIO 2                                                  // (1, 2, 3: 40.3-41.4 us per launch, alike; none, -DFX_O_NOPRIO: 43.3-44.2)
#endif
	if (R::NW != 4) __builtin_amdgcn_s_setprio(FX_O_PRIO);              // the octet's outer waves have the longest z step: they go first on their SIMD
#endif
	v4f P1[3][R::N1], P2[3][R::N2], P3[3][R::N3], NP[R::NI], NB[R::N1], NBn[R::N1], Bk[R::N2];
	uint32_t roff[R::NI];
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	const int qs = max(zb - 4, g.zlo), q_last = ze - 1 + 4;
	st.q_load_last = min(q_last, g.zhi);
	st.b_load_last = min(q_last - 1, g.zhi);
	st.zb = zb; st.ze = ze; st.Zg = g.Zg; st.wave = wave; st.lane = lane;
	st.wall_top = y0 == 0; st.wall_bot = y0 + R::NR >= g.Y;
	st.lds = lds_slice + lane; st.xbuf = xbuf; st.xflag = xflag;
	st.xf0 = (uint32_t)(size_t)(__attribute__((address_space(3))) int*)xflag;
	st.xb0 = (uint32_t)(size_t)(__attribute__((address_space(3))) v4f*)xbuf;
	const int yb = y0 - (R::A ? 4 : 1);
	const uint32_t XCOL = R::XT ? (uint32_t)x0 : R::XS == 2 ? 256u : 0u;   // the first column of the wave's half of a row / of its x tile
	st.keep = keep;
#pragma unroll
	for (int i = 0; i < R::NI; ++i) roff[i] = ((uint32_t)min(max(yb + i, 0), g.Y - 1) * (uint32_t)g.X + XCOL + 4u * (uint32_t)lane) * 4u;
	st.xe = xe; st.xe0 = (uint32_t)(size_t)(__attribute__((address_space(3))) v4f*)xe;
	st.eoff = ((uint32_t)min(max(yb + min(lane, R::NI - 1), 0), g.Y - 1) * (uint32_t)g.X + (R::XS == 2 ? 255u : 256u)) * 4u;
	st.E0 = 0.0f;
	st.s_ctr = 0; st.s_old = R::NI * 64;
	st.s_b2 = 2 * R::NI * 64; st.s_b3 = st.s_b2 + (R::NW == 4 ? R::N2 * 64 : 0); st.s_b4 = st.s_b3 + R::N2 * 64;      // (lean waves: two b slots, s_b3 and s_b4)
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R::N1; ++i) P1[k][i] = zero;
#pragma unroll
		for (int i = 0; i < R::N2; ++i) P2[k][i] = zero;
#pragma unroll
		for (int i = 0; i < R::N3; ++i) P3[k][i] = zero;
	}
#pragma unroll
	for (int i = 0; i < R::LDS_ROWS; ++i) st.lds[i * 64] = zero;
#pragma unroll
	for (int k = 0; k < R::N2; ++k) Bk[k] = zero;
	const size_t plane = g.plane();
	st.plane_bytes = plane * 4;
	const bool fill = qs == zb - 4;
	// q0: the step the walk starts with.  A chunk that starts four planes below its first output plane has nothing to compute in its first
	// two steps (level-1 planes below zb - 3 feed nothing that is stored): their planes are fetched in ONE burst with the third and
	// parked, instead of two steps that each wait out a memory round trip with nothing to do meanwhile.
	const int q0 = fill ? qs + 2 : qs;
	{
		const char* pb = reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(q0, st.q_load_last)) * plane);
#pragma unroll
		for (int i = 0; i < R::NI; ++i) NP[i] = *reinterpret_cast<const v4f*>(pb + roff[i]);
		const char* bbase = reinterpret_cast<const char*>(b + (size_t)g.lz(min(max(q0 - 1, g.zlo), g.zhi)) * plane);
#pragma unroll
		for (int j = 0; j < R::N1; ++j) NB[j] = *reinterpret_cast<const v4f*>(bbase + roff[j + 1]);
		const char* bnext = reinterpret_cast<const char*>(b + (size_t)g.lz(min(max(q0, g.zlo), g.zhi)) * plane);
#pragma unroll
		for (int j = 0; j < R::N1; ++j) NBn[j] = *reinterpret_cast<const v4f*>(bnext + roff[j + 1]);
		// the first step's centre plane q0 - 1, clamped into the planes present (q0 = 0: input plane -1 := plane 0)
		if (R::XS) st.E0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(max(q0 - 1, g.zlo), g.zhi)) * plane) + st.eoff);
		if (fill) {                                                       // input planes q0 - 2 and q0 - 1 straight into their LDS slots
			const char* pa = reinterpret_cast<const char*>(p_in + (size_t)g.lz(qs) * plane);
			v4f A_[R::NI], B_[R::NI];
#pragma unroll
			for (int i = 0; i < R::NI; ++i) A_[i] = *reinterpret_cast<const v4f*>(pa + roff[i]);
#pragma unroll
			for (int i = 0; i < R::NI; ++i) B_[i] = *reinterpret_cast<const v4f*>(pa + st.plane_bytes + roff[i]);
#pragma unroll
			for (int i = 0; i < R::NI; ++i) FXQ_LDS(st, st.s_old, i) = A_[i];
#pragma unroll
			for (int i = 0; i < R::NI; ++i) FXQ_LDS(st, st.s_ctr, i) = B_[i];
		}
	}
	st.q = q0;
	st.pp = reinterpret_cast<const char*>(p_in) + ((ptrdiff_t)g.lz(q0) + 1) * (ptrdiff_t)st.plane_bytes;
	st.pbq = reinterpret_cast<const char*>(b) + ((ptrdiff_t)g.lz(q0) + (R::NW == 4 ? 1 : 0)) * (ptrdiff_t)st.plane_bytes;   // (the quad fetches b[q + 1] in step q, a lean wave b[q])
	st.po_zb = reinterpret_cast<char*>(p_out) + (ptrdiff_t)g.lz(zb) * (ptrdiff_t)st.plane_bytes;
	st.po = reinterpret_cast<char*>(p_out) + ((ptrdiff_t)g.lz(q0) - 4) * (ptrdiff_t)st.plane_bytes;      // (only dereferenced for planes inside the chunk)
	if (MK) frz_begin4<R, R::N1>(g, fa, st, fz, MW, NM, roff, p_out, y0, q0, fill);
	// the rest of the pipeline's fill, peeled as in k_jacobi_strip3c: level-l planes below zb - 4 + l feed nothing that is stored
	if (fill) {
		step4<R, 0, true, false, false, false, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		step4<R, 1, true, false, false, false, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		step4<R, 2, true, true, false, false, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		step4<R, 0, true, true, false, false, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		step4<R, 1, true, true, true, false, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		step4<R, 2, true, true, true, false, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
	}
	head_stores4<R, MK>(st, fz, roff);
	for (;;) {
		step4<R, 0, true, true, true, true, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		if (st.q > q_last) break;
		step4<R, 1, true, true, true, true, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		if (st.q > q_last) break;
		step4<R, 2, true, true, true, true, MK>(st, P1, P2, P3, NP, NB, NBn, Bk, roff, fz, MW, NM);
		if (st.q > q_last) break;
	}
	if (MK) frz_end4(fa, fz, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The OCTET (k_jacobi_strip4o): the same pipeline with EIGHT waves of two rows per workgroup, two per SIMD.  A lone wave pays ~5 cycles
// per instruction whatever it is (tools/micro/issue_rate.cpp) and leaves the VALU pipe half idle; a second wave per SIMD takes the other
// half -- if every wave fits 256 registers and the workgroup the LDS.  So the six inner waves keep EVERYTHING in registers (the input
// window is a level-0 register window of three planes rotated by name, the b planes waiting for sweeps 2..4 a ring of three); only the two
// outer waves (two rows + the recomputed halo) park their input and b planes in the LDS as the quad's waves do.  LDS: 2 x 26 rows + 84
// mailbox rows (seven inner boundaries) = 136 KiB.
// ---------------------------------------------------------------------------------------------------------------------------
template <class R, int PH, bool S1, bool S2, bool S3, bool S4, bool MK = false>
__device__ __forceinline__ void step4r(Strip4<R>& st, v4f (&I)[3][R::NI], v4f (&P1)[3][R::NR], v4f (&P2)[3][R::NR], v4f (&P3)[3][R::NR],
	v4f (&NB)[R::NR], v4f (&Bp)[3][R::NR], const uint32_t (&roff)[R::NI], Frz4& fz, uint32_t (&MW)[3][3], uint32_t (&NM)[R::NR])
{
	static_assert(!R::A && !R::W, "the register-window step serves inner waves");
	constexpr int NEW = PH % 3, CTR = (PH + 2) % 3, OLD = (PH + 1) % 3;
	constexpr int NI = R::NI, NR = R::NR;
	constexpr uint32_t OWN = (1u << (4 * NR)) - 1u;
	const int q = st.q;
	uint32_t M0n = 0u;
	if (MK) {                                                           // the input plane's nibbles have arrived with it
#pragma unroll
		for (int j = 0; j < NR; ++j) M0n |= (NM[j] & 15u) << (4 * j);
		if (q >= st.zb && q < st.ze && (~M0n & OWN) != 0u) fz.rel |= 1u;
		if (q == 0) fz.M0c = M0n;
	}
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	Mail4<R> M1, M2, M3;
	M1.fu = M1.fd = M2.fu = M2.fd = M3.fu = M3.fd = INT_MIN; M1.hu = M1.hd = M2.hu = M2.hd = M3.hu = M3.hd = zero;
	if (S1) mail_fetch4<R, 1>(st, M1);
	XMail xm; xm.f = INT_MIN; xm.a = zero; xm.b = fx_f2{ 0.0f, 0.0f };
	if (R::XS && S2) xe_fetch<R>(st, xm);
	// ---- sweep 1: level-1 plane q-1 from input planes q-2 (OLD), q-1 (CTR), q (NEW: arrived) and b[q-1] (NB) ----
	if (__builtin_expect(q == 0, 0)) {                                  // input plane -1 := plane 0, once
		FXQ_RARE_BRANCH;
#pragma unroll
		for (int i = 0; i < NI; ++i) I[CTR][i] = I[NEW][i];
	}
	fx_f2 E1_[R::XS ? (NR + 1) / 2 : 1];                                 // the centre plane's cells across the cut (see step4)
	if (R::XS) {
#pragma unroll
		for (int j = 0; j < NR; ++j) { const float e_ = lane_cell(st.E0, j + 1); if (j & 1) E1_[j >> 1].y = e_; else E1_[j >> 1].x = e_; }
	}
	if (S1) {
		v4f T_[NR];
		uint32_t m1_ = 0u;
		{
#pragma unroll
			for (int jj = 0; jj < NR; ++jj) {
#ifdef FX_M_PLAIN
				const int j = jj;
#else
				const int j = MK ? NR - 1 - jj : jj;
#endif
				if (MK) {
#ifdef FX_M_PLAIN
					uint32_t n_;
					T_[j] = relax4qm(I[CTR][j + 1], I[CTR][j], I[CTR][j + 2], I[OLD][j + 1], I[NEW][j + 1], NB[j], FXQ_NIB(fz.M0c, j), n_);
					m1_ |= n_ << (4 * j);
#else
					T_[j] = relax4qf(I[CTR][j + 1], I[CTR][j], I[CTR][j + 2], I[OLD][j + 1], I[NEW][j + 1], NB[j], fz.M0c, 4 * j, m1_);
#endif
				} else if (R::XS) {
					T_[j] = (j & 1) ? relax4qx<R::XS ? R::XS : 1, 1>(I[CTR][j + 1], I[CTR][j], I[CTR][j + 2], I[OLD][j + 1], I[NEW][j + 1], NB[j], E1_[R::XS ? j >> 1 : 0])
					                : relax4qx<R::XS ? R::XS : 1, 0>(I[CTR][j + 1], I[CTR][j], I[CTR][j + 2], I[OLD][j + 1], I[NEW][j + 1], NB[j], E1_[R::XS ? j >> 1 : 0]);
				} else
					T_[j] = relax4q(I[CTR][j + 1], I[CTR][j], I[CTR][j + 2], I[OLD][j + 1], I[NEW][j + 1], NB[j]);
			}
#ifndef FX_M_PLAIN
			if (MK) m1_ |= fz.M0c & OWN;
#endif
			if (MK && q - 1 >= st.zb && q - 1 < st.ze && (~m1_ & OWN) != 0u) fz.rel |= 2u;
		}
		if (__builtin_expect(q - 1 == st.Zg, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int j = 0; j < NR; ++j) T_[j] = P1[CTR][j];
			if (MK) m1_ = MW[0][CTR];
		}
#pragma unroll
		for (int j = 0; j < NR; ++j) P1[NEW][j] = T_[j];
		if (MK) MW[0][NEW] = m1_;
		if (__builtin_expect(q - 1 == 0, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int j = 0; j < NR; ++j) P1[CTR][j] = T_[j];
			if (MK) MW[0][CTR] = m1_;
		}
	}
	if (MK) fz.M0c = M0n;                                               // plane q is the next step's centre
	// the b ring: [NEW] = b[q-2], [CTR] = b[q-3], [OLD] = b[q-4] on entry (named like the windows: they rotate with them)
	v4f B2_[NR], B3_[NR], B4_[NR];
#pragma unroll
	for (int j = 0; j < NR; ++j) { B2_[j] = Bp[NEW][j]; B3_[j] = Bp[CTR][j]; B4_[j] = Bp[OLD][j]; }
#pragma unroll
	for (int j = 0; j < NR; ++j) Bp[OLD][j] = NB[j];                     // b[q-1] takes the place of b[q-4]: [OLD] is the next step's [NEW]
	// the prefetch: input plane q+1 into the registers of plane q-2 (dead: [OLD] is the next step's [NEW]), b[q] into NB
#ifdef FX_S4_NOLOAD
	if (q + 1 <= st.q_load_last && q < -1000) {                        // (timing experiment: the loads never execute)
#else
	if (q + 1 <= st.q_load_last) {
#endif
#pragma unroll
		for (int i = 0; i < NI; ++i) I[OLD][i] = *reinterpret_cast<const v4f*>(st.pp + opaque32q(roff[i]));
		if (MK) {
#pragma unroll
			for (int j = 0; j < NR; ++j) NM[j] = fz.pm[opaque32q(roff[j + 1] >> 4)];
		}
	} else {                                                            // past the last present plane the window keeps plane zhi
#pragma unroll
		for (int i = 0; i < NI; ++i) I[OLD][i] = I[NEW][i];
	}
#ifdef FX_S4_NOLOAD
	if (q <= st.b_load_last && q < -1000) {
#else
	if (q <= st.b_load_last) {
#endif
#pragma unroll
		for (int j = 0; j < NR; ++j) NB[j] = *reinterpret_cast<const v4f*>(st.pbq + opaque32q(roff[j + 1]));
	}
	if (R::XS && q <= st.q_load_last) st.E0 = *reinterpret_cast<const float*>(st.pp - st.plane_bytes + st.eoff);   // plane q's cells across the cut: the next step's centre
	st.pp += st.plane_bytes; st.pbq += st.plane_bytes;
	if (MK) fz.pm += st.plane_bytes >> 4;
	v4f HU1 = zero, HD1 = zero;
	if (S1) hand_over4<R, 1, NR>(st, M1, P1[NEW], HU1, HD1);
	if (R::XS && S2) xe_check<R>(st, xm);
	if (S2) mail_fetch4<R, 2>(st, M2);
	// ---- sweep 2 ----
	if (S2) {
		v4f T_[NR];
		uint32_t m2_ = 0u;
		relax_level4<R, 1, NR, NR, MK>(st, P1[OLD], P1[CTR], P1[NEW], B2_, HU1, HD1, T_, MK ? MW[0][CTR] : 0u, m2_, &xm);
		if (MK && q - 2 >= st.zb && q - 2 < st.ze && (~m2_ & OWN) != 0u) fz.rel |= 4u;
		if (__builtin_expect(q - 2 == st.Zg, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int k = 0; k < NR; ++k) T_[k] = P2[CTR][k];
			if (MK) m2_ = MW[1][CTR];
		}
#pragma unroll
		for (int k = 0; k < NR; ++k) P2[NEW][k] = T_[k];
		if (MK) MW[1][NEW] = m2_;
		if (__builtin_expect(q - 2 == 0, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int k = 0; k < NR; ++k) P2[CTR][k] = T_[k];
			if (MK) MW[1][CTR] = m2_;
		}
	}
	v4f HU2 = zero, HD2 = zero;
	if (S2) hand_over4<R, 2, NR>(st, M2, P2[NEW], HU2, HD2);
	if (S3) mail_fetch4<R, 3>(st, M3);
	// ---- sweep 3 ----
	if (S3) {
		v4f T_[NR];
		uint32_t m3_ = 0u;
		relax_level4<R, 2, NR, NR, MK>(st, P2[OLD], P2[CTR], P2[NEW], B3_, HU2, HD2, T_, MK ? MW[1][CTR] : 0u, m3_, &xm);
		if (MK && q - 3 >= st.zb && q - 3 < st.ze && (~m3_ & OWN) != 0u) fz.rel |= 8u;
		if (__builtin_expect(q - 3 == st.Zg, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int m = 0; m < NR; ++m) T_[m] = P3[CTR][m];
			if (MK) m3_ = MW[2][CTR];
		}
#pragma unroll
		for (int m = 0; m < NR; ++m) P3[NEW][m] = T_[m];
		if (MK) MW[2][NEW] = m3_;
		if (__builtin_expect(q - 3 == 0, 0)) {
			FXQ_RARE_BRANCH;
#pragma unroll
			for (int m = 0; m < NR; ++m) P3[CTR][m] = T_[m];
			if (MK) MW[2][CTR] = m3_;
		}
	}
	v4f HU3 = zero, HD3 = zero;
	if (S3) hand_over4<R, 3, NR>(st, M3, P3[NEW], HU3, HD3);
	// ---- sweep 4: the output ----
	if (S4) {
		v4f X_[NR];
		uint32_t m4_ = 0u;
		relax_level4<R, 3, NR, NR, MK>(st, P3[OLD], P3[CTR], P3[NEW], B4_, HU3, HD3, X_, MK ? MW[2][CTR] : 0u, m4_, &xm);
		char* dst_ = q - 4 >= st.zb ? st.po : st.po_zb;
#ifdef FX_S4_NOSTORE
		if (q < -1000)
#endif
#pragma unroll
		for (int m = 0; m < NR; ++m) {
			if (R::XT) store_row4xt<R::NT>(dst_, opaque32q(roff[m + 1]), st.keep, X_[m]);
			else if (R::NT) store_row4nt(dst_, opaque32q(roff[m + 1]), X_[m]);
			else store_row4(dst_, opaque32q(roff[m + 1]), X_[m]);
		}
		if (MK) frz_out4<R>(st, fz, dst_, X_, m4_, roff);
	}
#ifndef FX_S4_NOXE
	if (R::XS) {                                                        // my cut cells of this step's new planes (levels 1..3), for the partner's next step
		float v_[6] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
		for (int k = 0; k < NR; ++k) {
			v_[R::XO1 + k] = R::XS == 1 ? P1[NEW][k].w : P1[NEW][k].x;
			v_[R::XO2 + k] = R::XS == 1 ? P2[NEW][k].w : P2[NEW][k].x;
			v_[R::XO3 + k] = R::XS == 1 ? P3[NEW][k].w : P3[NEW][k].x;
		}
		xe_publish<R::XS ? R::XS : 1>(st.xe0 + 16u * (uint32_t)xeslot<R::NW>(q & 1, st.wave), st.xf0 + 4u * (uint32_t)(3 * R::NW + st.wave), q, v_[0], v_[1], v_[2], v_[3], v_[4], v_[5]);
	}
#endif
	st.po += st.plane_bytes;
	++st.q;
}

template <class R, bool MK = false>
__device__ __forceinline__ void run4r(const Geom& g, const float* __restrict__ p_in, const float* __restrict__ b, float* __restrict__ p_out,
	int zb, int ze, int y0, int wave, int lane, v4f* xbuf, int* xflag, const FrzArgs& fa, v4f* xe = nullptr, int x0 = 0, bool keep = true)
{
	Strip4<R> st;
	Frz4 fz;
	uint32_t MW[3][3], NM[R::NR];
	v4f I[3][R::NI], P1[3][R::NR], P2[3][R::NR], P3[3][R::NR], NB[R::NR], Bp[3][R::NR];
	uint32_t roff[R::NI];
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	const int qs = max(zb - 4, g.zlo), q_last = ze - 1 + 4;
	st.q_load_last = min(q_last, g.zhi);
	st.b_load_last = min(q_last - 1, g.zhi);
	st.zb = zb; st.ze = ze; st.Zg = g.Zg; st.wave = wave; st.lane = lane;
	st.wall_top = false; st.wall_bot = false;
	st.lds = nullptr; st.xbuf = xbuf; st.xflag = xflag;
	st.xf0 = (uint32_t)(size_t)(__attribute__((address_space(3))) int*)xflag;
	st.xb0 = (uint32_t)(size_t)(__attribute__((address_space(3))) v4f*)xbuf;
	st.s_ctr = st.s_old = st.s_b2 = st.s_b3 = st.s_b4 = 0;
	const int yb = y0 - 1;
	const uint32_t XCOL = R::XT ? (uint32_t)x0 : R::XS == 2 ? 256u : 0u;
	st.keep = keep;
#pragma unroll
	for (int i = 0; i < R::NI; ++i) roff[i] = ((uint32_t)min(max(yb + i, 0), g.Y - 1) * (uint32_t)g.X + XCOL + 4u * (uint32_t)lane) * 4u;
	st.xe = xe; st.xe0 = (uint32_t)(size_t)(__attribute__((address_space(3))) v4f*)xe;
	st.eoff = ((uint32_t)min(max(yb + min(lane, R::NI - 1), 0), g.Y - 1) * (uint32_t)g.X + (R::XS == 2 ? 255u : 256u)) * 4u;
	st.E0 = 0.0f;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int i = 0; i < R::NR; ++i) { P1[k][i] = zero; P2[k][i] = zero; P3[k][i] = zero; Bp[k][i] = zero; }
#pragma unroll
		for (int i = 0; i < R::NI; ++i) I[k][i] = zero;
	}
	const size_t plane = g.plane();
	st.plane_bytes = plane * 4;
	const bool fill = qs == zb - 4;
	const int q0 = fill ? qs + 2 : qs;
	// the walk starts with phase 0: [NEW] = slot 0 takes plane q0, [CTR] = slot 2 plane q0 - 1, [OLD] = slot 1 plane q0 - 2 (a chunk that starts
	// four planes below its first output plane fetches all three at once: its first two steps have nothing to compute)
	{
		const char* pb = reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(q0, st.q_load_last)) * plane);
#pragma unroll
		for (int i = 0; i < R::NI; ++i) I[0][i] = *reinterpret_cast<const v4f*>(pb + roff[i]);
		const char* bbase = reinterpret_cast<const char*>(b + (size_t)g.lz(min(max(q0 - 1, g.zlo), g.zhi)) * plane);
#pragma unroll
		for (int j = 0; j < R::NR; ++j) NB[j] = *reinterpret_cast<const v4f*>(bbase + roff[j + 1]);
		// the first step's centre plane q0 - 1, clamped into the planes present (q0 = 0: input plane -1 := plane 0)
		if (R::XS) st.E0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p_in + (size_t)g.lz(min(max(q0 - 1, g.zlo), g.zhi)) * plane) + st.eoff);
		if (fill) {
			const char* pa = reinterpret_cast<const char*>(p_in + (size_t)g.lz(qs) * plane);
#pragma unroll
			for (int i = 0; i < R::NI; ++i) I[1][i] = *reinterpret_cast<const v4f*>(pa + roff[i]);
#pragma unroll
			for (int i = 0; i < R::NI; ++i) I[2][i] = *reinterpret_cast<const v4f*>(pa + st.plane_bytes + roff[i]);
		}
	}
	st.q = q0;
	st.pp = reinterpret_cast<const char*>(p_in) + ((ptrdiff_t)g.lz(q0) + 1) * (ptrdiff_t)st.plane_bytes;
	st.pbq = reinterpret_cast<const char*>(b) + (ptrdiff_t)g.lz(q0) * (ptrdiff_t)st.plane_bytes;          // (this walk fetches b[q] in step q)
	st.po_zb = reinterpret_cast<char*>(p_out) + (ptrdiff_t)g.lz(zb) * (ptrdiff_t)st.plane_bytes;
	st.po = reinterpret_cast<char*>(p_out) + ((ptrdiff_t)g.lz(q0) - 4) * (ptrdiff_t)st.plane_bytes;
	if (MK) frz_begin4<R, R::NR>(g, fa, st, fz, MW, NM, roff, p_out, y0, q0, fill);
	if (fill) {
		step4r<R, 0, true, false, false, false, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		step4r<R, 1, true, false, false, false, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		step4r<R, 2, true, true, false, false, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		step4r<R, 0, true, true, false, false, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		step4r<R, 1, true, true, true, false, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		step4r<R, 2, true, true, true, false, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
	}
	head_stores4<R, MK>(st, fz, roff);
	for (;;) {
		step4r<R, 0, true, true, true, true, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		if (st.q > q_last) break;
		step4r<R, 1, true, true, true, true, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		if (st.q > q_last) break;
		step4r<R, 2, true, true, true, true, MK>(st, I, P1, P2, P3, NB, Bp, roff, fz, MW, NM);
		if (st.q > q_last) break;
	}
	if (MK) frz_end4(fa, fz, lane);
}

// rows per wave, top to bottom: 1 2 2 2 2 2 2 1 -- a band of FOURTEEN rows.  The outer waves (one row + the recomputed halo: 4 + 3 + 2 + 1 = 10
// row updates per step) park their input planes and two b planes in the LDS, the six 2-row waves (8 updates) live in registers.  What a
// step costs is its slowest wave: with two rows in the outer waves (14 updates + the LDS traffic) the octet ran at the quad's speed, and a
// 3-row inner wave does not fit 256 registers.  Bands of 14 do not tile 256 rows: the last band is shifted up to end at the last row and
// recomputes the rows it shares with its neighbour -- the same inputs through the same arithmetic, so both store the same bits.
typedef Role4<1, true, false, 8> OctTop;
typedef Role4<2, false, false, 8> OctMid;
typedef Role4<1, false, true, 8> OctBot;
constexpr int O_BAND = 14;
constexpr int O_LDS_ROWS = OctTop::LDS_ROWS + OctBot::LDS_ROWS;
#ifdef FX_O_ONLYMID
constexpr int O_XROWS = 2 * 8 * 3 * 2;
#else
constexpr int O_XROWS = 2 * 7 * 3 * 2;
#endif
static_assert((O_LDS_ROWS + O_XROWS) * 1024 + 128 <= 160 * 1024, "the octet's windows must fit the CU's LDS");

// Where band `grp` of `ngroups` starts.  Bands lie top-down; the last one is shifted up to end at the last row (the rows it shares with its
// neighbour are computed twice from the same inputs by the same arithmetic: both workgroups store the same bits).  An outer wave knows
// a wall only as ITS OWN row (wall_top / wall_bot), and the three level-1 halo rows it recomputes beyond that row must be rows of the
// field (the input row behind the last of them may be the clamped copy: that IS the wall's neighbour) -- a level-l row computed from
// clamped loads behind the wall is not what the wall row's missing neighbour stands for.  So a band whose lower halo would reach beyond the
// last row (the second-to-last band when Y % 14 is 1 or 2: Y = 128, 240, 30 ...) is shifted up too, until its halo ends at the last row;
// the upper halo of a shifted band lies inside from Y = 17 on (octet_rows_supported).
__device__ __host__ __forceinline__ int octet_band_y(int grp, int ngroups, int Y)
{
	if (grp == ngroups - 1) return Y - O_BAND;
	const int yg = grp * O_BAND;
	return yg + O_BAND + 3 > Y ? Y - O_BAND - 3 : yg;
}
// Y = 14: one band between both walls; from 17 on every shifted band keeps its three level-1 halo rows inside (15, 16: the last band's
// upper halo would cross the first row)
__host__ inline bool octet_rows_supported(int Y) { return Y == O_BAND || Y >= O_BAND + 3; }

__global__ __launch_bounds__(512, 2) void k_jacobi_strip4o(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	__shared__ v4f lds_all[O_LDS_ROWS * 64];
	__shared__ v4f xbuf[O_XROWS * 64];
	__shared__ int xflag[24];                                          // [level 1..3][wave]
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	int tile = (int)blockIdx.x;
	if (remap) {
		const int n = ngroups * nchunks, qn = n >> 3, r = n & 7;
		const int xcd = tile & 7, j = tile >> 3;
		tile = xcd * qn + min(xcd, r) + j;
	}
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 4, g.zlo);
	const bool fill = qs == zb - 4;
	if (threadIdx.x < 24) xflag[threadIdx.x] = fill ? qs + 2 * ((int)threadIdx.x / 8 + 1) - 1 : qs - 1;
	for (int i = (int)threadIdx.x; i < O_XROWS * 64; i += 512) xbuf[i] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	__syncthreads();
	const int yg = octet_band_y(grp, ngroups, g.Y);                     // (the last band is shifted up to end at the last row)
	const FrzArgs none{};
#ifdef FX_O_ONLYMID
	run4r<OctMid>(g, p_in, b, p_out, zb, ze, yg + 2 * wave, wave, lane, xbuf, xflag, none); return;   // (timing experiment: eight register-window waves in a ring; results are wrong)
#endif
	if (wave == 0) run4<OctTop>(g, p_in, b, p_out, zb, ze, yg, wave, lane, lds_all, xbuf, xflag, none);
	else if (wave == 7) run4<OctBot>(g, p_in, b, p_out, zb, ze, yg + O_BAND - 1, wave, lane, lds_all + OctTop::LDS_ROWS * 64, xbuf, xflag, none);
	else run4r<OctMid>(g, p_in, b, p_out, zb, ze, yg + 2 * wave - 1, wave, lane, xbuf, xflag, none);
}

// FOUR levels of the reference's own pressure loop for every cell, per launch: the octet with the freeze nibbles of fx_jacobi_freeze.hip
// carried along (k_freeze_strip3 of fx_jacobi_stripm.hip is the same idea on the three-sweep pipeline, one wave per SIMD).  Input: level L in
// p_in with its nibbles in fa.m_in; output: level L + 4 to BOTH p_outA and fa.p_outB and the nibbles to both mask buffers, the tile marks
// and the statistics word as k_freeze_strip3 leaves them.  The mailbox rows are pressures only: a neighbour's edge row arrives with its
// freeze decisions applied, and a sweep needs the nibbles of its CENTRE cells alone.  Single domain (Zg = nz, no halo planes).
__global__ __launch_bounds__(512, 2) void k_freeze_strip4o(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_outA, const FrzArgs fa, int zchunk, int ngroups, int nchunks)
{
	__shared__ v4f lds_all[O_LDS_ROWS * 64];
	__shared__ v4f xbuf[O_XROWS * 64];
	__shared__ int xflag[24];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	int tile = (int)blockIdx.x;
	{
		const int n = ngroups * nchunks, qn = n >> 3, r = n & 7;
		const int xcd = tile & 7, j = tile >> 3;
		tile = xcd * qn + min(xcd, r) + j;
	}
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int zb = chunk * zchunk, ze = min(zb + zchunk, g.Zg);
	const int qs = max(zb - 4, 0);
	const bool fill = qs == zb - 4;
	if (threadIdx.x < 24) xflag[threadIdx.x] = fill ? qs + 2 * ((int)threadIdx.x / 8 + 1) - 1 : qs - 1;
	for (int i = (int)threadIdx.x; i < O_XROWS * 64; i += 512) xbuf[i] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	__syncthreads();
	const int yg = octet_band_y(grp, ngroups, g.Y);
	if (wave == 0) run4<OctTop, true>(g, p_in, b, p_outA, zb, ze, yg, wave, lane, lds_all, xbuf, xflag, fa);
	else if (wave == 7) run4<OctBot, true>(g, p_in, b, p_outA, zb, ze, yg + O_BAND - 1, wave, lane, lds_all + OctTop::LDS_ROWS * 64, xbuf, xflag, fa);
	else run4r<OctMid, true>(g, p_in, b, p_outA, zb, ze, yg + 2 * wave - 1, wave, lane, xbuf, xflag, fa);
}

// ---------------------------------------------------------------------------------------------------------------------------
// X = 512 (k_jacobi_strip4x): the octet's pipeline on HALF rows.  A wave is 64 lanes x 4 cells = 256 cells; a 512-cell row in one wave
// would double every window (an outer wave's ten row updates per step -- the slowest wave of the step -- would become twenty), so the
// workgroup's eight waves are 2 x-halves x 4 waves top-down (1 + 2 + 2 + 1 rows: a band of SIX rows), the x cut INSIDE the workgroup:
//   * the input cells across the cut come with the plane a wave prefetches (one more load per step: lane i fetches row i's cell) and reach
//     a row's update through v_readlane;
//   * the cells of levels 1..3 travel like the edge rows -- the lane at the cut writes them into a 16-byte LDS slot per wave, level and step
//     parity, the partner (the wave with the same rows in the other half) reads the slot as a broadcast, a sweep early, and checks the
//     partner's step counter at the hand-over: the counter a wave posts per level now vouches for its edge rows AND its cut cells;
//   * the partner's cell enters the update as the `old` operand of the DPP shift that brings every other lane its x neighbour, placed into
//     the operand pair by the v_pk_mov_b32 that builds the pair anyway (relax4qx): no instruction more per update than at X = 256.
// Waves w and w + 4 share a SIMD: the chain positions (left half top-down 0..3, right half 4..7) are dealt so that each SIMD gets one
// outer wave (ten row updates per step) and one inner (eight).  LDS: 4 x 18 parked rows + 84 mailbox rows = 156 KiB.
// Work: the launch's band-planes, ordered (z chunk, band, plane), are cut into ONE contiguous run per workgroup, so any band count fills the
// chip's 256 CUs evenly -- 86 bands of six rows do not divide into 256 -- and a run that crosses into the next band becomes two pieces
// (each pays its own fill).  The chunks are as long as a run (the last one shorter): the runs of a chunk then ARE its bands, neighbouring
// bands walk the same planes at the same time on neighbouring CUs of one XCD, and the eight halo rows a six-row band reads beyond its
// own come out of that XCD's L2 instead of the fabric (band-major runs: every band alone at its depth, 1.8 x the compulsory traffic).
// ---------------------------------------------------------------------------------------------------------------------------
typedef Role4<1, true, false, 8, 1> XTopL;
typedef Role4<1, false, true, 8, 1> XBotL;
constexpr int X_BAND = 6;
constexpr int X_OUTER_ROWS = XTopL::LDS_ROWS;                          // = XBotL's
static_assert(XTopL::LDS_ROWS == XBotL::LDS_ROWS, "outer waves park alike");
constexpr int X_LDS_ROWS = 4 * X_OUTER_ROWS;
constexpr int X_XROWS = 2 * 7 * 3 * 2;                                 // (boundary 3, between the two halves' chains, is never used)
static_assert((X_LDS_ROWS + X_XROWS) * 1024 + 2 * 8 * 2 * 16 + 256 <= 160 * 1024, "the half-row octet's windows must fit the CU's LDS");

// where band `grp` starts: octet_band_y for bands of BAND rows
__device__ __host__ __forceinline__ int band_y(int grp, int ngroups, int Y, int BAND)
{
	if (grp == ngroups - 1) return Y - BAND;
	const int yg = grp * BAND;
	return yg + BAND + 3 > Y ? Y - BAND - 3 : yg;
}
__host__ inline bool band_rows_supported(int Y, int BAND) { return Y == BAND || Y >= BAND + 3; }

// The launch's band-planes in the order (z chunk, band, plane): `nch` chunks, chunk i = planes [zc[i], zc[i + 1]) of the launch's range.
// exact: workgroup k takes piece k (band k % bands of chunk k / bands) and nothing else -- nwg = bands x nch, any number of them.
constexpr int RUNS4_MAXCH = 32;
struct Runs4 { int bands, nzp, nch, nwg, minp, exact; int zc[RUNS4_MAXCH + 1]; };
// position s of that order -> its band, its plane (relative to the range) and the planes left in its piece (a band's planes of one chunk)
__device__ __forceinline__ void run_locate(const Runs4& r, int s, int& band, int& z, int& left)
{
	int i = 0;
	while (i + 1 < r.nch && s >= r.bands * r.zc[i + 1]) ++i;
	const int ci = r.zc[i + 1] - r.zc[i], q = s - r.bands * r.zc[i];
	band = q / ci;
	const int zo = q - band * ci;
	z = r.zc[i] + zo; left = ci - zo;
}
// the first position of workgroup k's run: k T / nwg, moved to the piece boundary when it would leave fewer than `minp` planes of a piece
// the first position of workgroup k's run: k T / nwg, moved to the piece boundary when it would leave fewer than `minp` planes of a piece.
// (Cuts placed under a step budget per run -- a piece costs its planes + its fill -- were built and measured level with these at every
// depth: the fill steps of a piece are cheaper than its full steps by about what the model charges for them.)
__device__ __forceinline__ int run_cut(const Runs4& r, int k)
{
	const int T = r.bands * r.nzp;
	if (k >= r.nwg) return T;
	if (r.exact) { const int i = k / r.bands; return r.bands * r.zc[i] + (k - i * r.bands) * (r.zc[i + 1] - r.zc[i]); }
	int s = (int)((long long)k * T / r.nwg), band, z, left;
	run_locate(r, s, band, z, left);
	int i = 0;
	while (i + 1 < r.nch && z >= r.zc[i + 1]) ++i;
	const int zo = z - r.zc[i];
	if (zo < r.minp) s -= zo;
	else if (left < r.minp) s += left;
	return s;
}

// NT: the output as non-temporal stores (fields beyond the Infinity Cache: store_row4nt)
template <bool NT>
__global__ __launch_bounds__(512, 2) void k_jacobi_strip4x(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, int z_begin, const Runs4 runs, int remap)
{
	typedef Role4<1, true, false, 8, 1, NT> XTopL;
	typedef Role4<2, false, false, 8, 1, NT> XMidL;
	typedef Role4<1, false, true, 8, 1, NT> XBotL;
	typedef Role4<1, true, false, 8, 2, NT> XTopR;
	typedef Role4<2, false, false, 8, 2, NT> XMidR;
	typedef Role4<1, false, true, 8, 2, NT> XBotR;
	__shared__ v4f lds_all[X_LDS_ROWS * 64];
	__shared__ v4f xbuf[X_XROWS * 64];
	__shared__ v4f xe[2 * 8 * 2];                                       // the cut cells: [step parity][wave] x 32 bytes
	__shared__ int xflag[32];                                          // [level 1..3][wave] as in the octet + [3][wave]: the last step whose cut cells the wave has published
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int c = wave < 4 ? wave : 4 + ((wave ^ 1) & 3);              // chain position: 0 1 2 3 | 5 4 7 6 -- one outer wave per SIMD
	int k = (int)blockIdx.x;
	if (remap) {                                                        // XCD x walks the x-th contiguous eighth of the runs
		const int qn = runs.nwg >> 3, r = runs.nwg & 7;
		const int xcd = k & 7, j = k >> 3;
		k = xcd * qn + min(xcd, r) + j;
	}
	// The pieces of a run are walked LAST FIRST: a run that crosses into the next band is [the tail of band b | the head of band b + 1], and
	// with the head first every workgroup of a chunk walks UP through the planes from the chunk's first one at the same time -- neighbouring
	// bands stay at the same depth (the halo rows come out of the L2) although the runs are a few planes longer than the chunk's pieces.
	// (Tail first, each run starts 4 planes deeper into its band than its neighbour: 84 of the 256 workgroups of a 512^3 launch then shared nothing.)
	// The list is made once, into the LDS: the walk below keeps nothing of the run arithmetic alive across a piece, whose windows take every
	// register there is.
	constexpr int MAXP = 8;
	__shared__ int piece[MAXP][3];                                      // { first plane, planes, first row of the band }
	__shared__ int npiece;
	if (threadIdx.x == 0) {
		const int t0 = run_cut(runs, k);
		int t1 = run_cut(runs, k + 1), n = 0;
		while (t0 < t1 && n < MAXP) {
			int band, zoff, left;
			run_locate(runs, t1 - 1, band, zoff, left);                   // the piece that ends the run
			int pz0 = 0;
			{ int i = 0; while (i + 1 < runs.nch && zoff >= runs.zc[i + 1]) ++i; pz0 = runs.zc[i]; }
			const int zfirst = max(pz0, zoff - (t1 - 1 - t0));
			piece[n][0] = z_begin + zfirst; piece[n][1] = zoff - zfirst + 1; piece[n][2] = band_y(band, runs.bands, g.Y, X_BAND);
			t1 -= zoff - zfirst + 1;
			++n;
		}
		// (the launcher makes runs of at most seven pieces; planes left over here would be planes nobody sweeps: loud, like a hand-over that times out)
		if (t0 < t1) strip4_raise_fault();
		npiece = n;
	}
	__syncthreads();
	const FrzArgs none{};
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	for (int ip = 0; ip < __builtin_amdgcn_readfirstlane(npiece); ++ip) {
		const int zb = __builtin_amdgcn_readfirstlane(piece[ip][0]), ze = zb + __builtin_amdgcn_readfirstlane(piece[ip][1]);
		const int yg = __builtin_amdgcn_readfirstlane(piece[ip][2]);
		const int qs = max(zb - 4, g.zlo);
		const bool fill = qs == zb - 4;
		// the lane id is made afresh for every piece (v_mbcnt; volatile: not hoisted): nothing per lane lives across a piece, whose windows
		// take every register there is
		int lane;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(lane));
		const int tid = wave * 64 + lane;
		__syncthreads();                                                // (the previous piece's mailbox reads are over)
		if (tid < 24) xflag[tid] = fill ? qs + 2 * (tid / 8 + 1) - 1 : qs - 1;
		else if (tid < 32) xflag[tid] = (fill ? qs + 2 : qs) - 1;          // the step in front of the walk's first one: its (never written) cells feed only sweeps whose planes are not kept
		for (int i = tid; i < X_XROWS * 64; i += 512) xbuf[i] = zero;
		if (tid < 2 * 8 * 2) xe[tid] = zero;
		__syncthreads();
		switch (c) {
		case 0: run4<XTopL>(g, p_in, b, p_out, zb, ze, yg, c, lane, lds_all, xbuf, xflag, none, xe); break;
		case 1: case 2: run4r<XMidL>(g, p_in, b, p_out, zb, ze, yg + 2 * c - 1, c, lane, xbuf, xflag, none, xe); break;
		case 3: run4<XBotL>(g, p_in, b, p_out, zb, ze, yg + X_BAND - 1, c, lane, lds_all + X_OUTER_ROWS * 64, xbuf, xflag, none, xe); break;
		case 4: run4<XTopR>(g, p_in, b, p_out, zb, ze, yg, c, lane, lds_all + 2 * X_OUTER_ROWS * 64, xbuf, xflag, none, xe); break;
		case 5: case 6: run4r<XMidR>(g, p_in, b, p_out, zb, ze, yg + 2 * (c - 4) - 1, c, lane, xbuf, xflag, none, xe); break;
		default: run4<XBotR>(g, p_in, b, p_out, zb, ze, yg + X_BAND - 1, c, lane, lds_all + 3 * X_OUTER_ROWS * 64, xbuf, xflag, none, xe); break;
		}
	}
}

// Any other row length from 256 cells on (X % 4 == 0; k_jacobi_strip4t): the octet of X = 256 on X TILES.  A wave still holds 256 consecutive cells
// of a row; where a tile's side is no wall its outermost lane is the tile's own x halo -- four cells for four sweeps, recomputed like the rows
// above and below a band: after sweep s the s cells next to the side are not the field's, after four exactly that lane's, which computes
// along and stores nothing (the wall clamp it applies at its outer cell is part of what is thrown away).  Tiles lie 248 cells apart, the last
// one ends at the row's end and keeps what its neighbour does not: 1 + ceil((X - 256) / 248) tiles.  Nothing crosses between the tiles of a
// row inside a launch, so a (tile, band) pair is a "band" of k_jacobi_strip4x's run arithmetic -- the launch's (tile, band)-planes in
// (z chunk, band, tile, plane) order, one contiguous run per workgroup, pieces walked last first (neighbouring tiles and bands at the same
// depth on neighbouring CUs: the halo lanes and rows come out of the XCD's L2).
// Rows SHORTER than 256 cells (whole quads) are one tile whose upper lanes are switched off for the whole walk: with EXEC clear a lane
// loads, computes and stores nothing, and the DPP shift that brings a lane its right-hand neighbour finds no source at the row's last
// lane and leaves the `old` operand -- the lane's own cell -- in place, exactly as at lane 63 of a full row: the wall costs nothing.
__device__ __host__ __forceinline__ int xtiles(int X) { return X <= 256 ? 1 : 1 + (X - 256 + 247) / 248; }
template <bool NT>
__global__ __launch_bounds__(512, 2) void k_jacobi_strip4t(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, int z_begin, const Runs4 runs, int ntx, int remap)
{
	typedef Role4<1, true, false, 8, 0, NT, true> TTop;
	typedef Role4<2, false, false, 8, 0, NT, true> TMid;
	typedef Role4<1, false, true, 8, 0, NT, true> TBot;
	__shared__ v4f lds_all[O_LDS_ROWS * 64];
	__shared__ v4f xbuf[O_XROWS * 64];
	__shared__ int xflag[24];
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	int k = (int)blockIdx.x;
	if (remap) {
		const int qn = runs.nwg >> 3, r = runs.nwg & 7;
		const int xcd = k & 7, j = k >> 3;
		k = xcd * qn + min(xcd, r) + j;
	}
	constexpr int MAXP = 8;
	__shared__ int piece[MAXP][3];                                      // { first plane, planes, (tile, band) }
	__shared__ int npiece;
	if (threadIdx.x == 0) {
		const int t0 = run_cut(runs, k);
		int t1 = run_cut(runs, k + 1), n = 0;
		while (t0 < t1 && n < MAXP) {
			int band, zoff, left;
			run_locate(runs, t1 - 1, band, zoff, left);
			int pz0 = 0;
			{ int i = 0; while (i + 1 < runs.nch && zoff >= runs.zc[i + 1]) ++i; pz0 = runs.zc[i]; }
			const int zfirst = max(pz0, zoff - (t1 - 1 - t0));
			piece[n][0] = z_begin + zfirst; piece[n][1] = zoff - zfirst + 1; piece[n][2] = band;
			t1 -= zoff - zfirst + 1;
			++n;
		}
		if (t0 < t1) strip4_raise_fault();
		npiece = n;
	}
	__syncthreads();
	const FrzArgs none{};
	const v4f zero = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	const int nby = runs.bands / ntx;
	for (int ip = 0; ip < __builtin_amdgcn_readfirstlane(npiece); ++ip) {
		const int zb = __builtin_amdgcn_readfirstlane(piece[ip][0]), ze = zb + __builtin_amdgcn_readfirstlane(piece[ip][1]);
		const int tb = __builtin_amdgcn_readfirstlane(piece[ip][2]);
		const int tx = tb % ntx, yg = octet_band_y(tb / ntx, nby, g.Y);
		// the tile's first column, and the lanes whose cells it keeps: from where its left neighbour stops (a wall: from the first) to its
		// last lane but one (a wall: the last)
		const int x0 = tx == ntx - 1 ? max(g.X - 256, 0) : 248 * tx;
		const int nl = min(g.X >> 2, 64);                               // lanes that hold cells (all 64 from 256 cells on)
		const int keep_lo = tx == 0 ? 0 : (248 * (tx - 1) + 252 - x0) >> 2, keep_hi = tx == ntx - 1 ? 63 : 62;
		const int qs = max(zb - 4, g.zlo);
		const bool fill = qs == zb - 4;
		int lane;
		asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(lane));
		const int tid = wave * 64 + lane;
		const bool keep = lane >= keep_lo && lane <= keep_hi;
		__syncthreads();                                                // (the previous piece's mailbox reads are over)
		if (tid < 24) xflag[tid] = fill ? qs + 2 * (tid / 8 + 1) - 1 : qs - 1;
		for (int i = tid; i < O_XROWS * 64; i += 512) xbuf[i] = zero;
		__syncthreads();
		if (lane < nl) {                                                // (no barrier inside: the waves of a piece meet through the LDS counters)
			if (wave == 0) run4<TTop>(g, p_in, b, p_out, zb, ze, yg, wave, lane, lds_all, xbuf, xflag, none, nullptr, x0, keep);
			else if (wave == 7) run4<TBot>(g, p_in, b, p_out, zb, ze, yg + O_BAND - 1, wave, lane, lds_all + TTop::LDS_ROWS * 64, xbuf, xflag, none, nullptr, x0, keep);
			else run4r<TMid>(g, p_in, b, p_out, zb, ze, yg + 2 * wave - 1, wave, lane, xbuf, xflag, none, nullptr, x0, keep);
		}
	}
}

#ifdef FX_LAB      // the quad (STRIP4_OCTET=0): superseded by the octet within round 5, kept as its A/B baseline in lab builds only
__global__ __launch_bounds__(256, 1) void k_jacobi_strip4q(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, int z_begin, int z_end, int zchunk, int ngroups, int nchunks, int remap)
{
	__shared__ v4f lds_all[Q_LDS_ROWS * 64];
	__shared__ v4f xbuf[Q_XROWS * 64];
	__shared__ int xflag[16];                                          // [level 1..3][wave]: the last z step whose edge rows the wave has published
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	int tile = (int)blockIdx.x;
	if (remap) {                                                        // XCD k walks the k-th contiguous eighth of the tile sequence
		const int n = ngroups * nchunks, qn = n >> 3, r = n & 7;
		const int xcd = tile & 7, j = tile >> 3;
		tile = xcd * qn + min(xcd, r) + j;
	}
	const int grp = tile % ngroups, chunk = tile / ngroups;
	const int zb = z_begin + chunk * zchunk, ze = min(zb + zchunk, z_end);
	const int qs = max(zb - 4, g.zlo);
	const bool fill = qs == zb - 4;
	// the counters start where the first active hand-over of each level expects them (level l: step qs + 2 l)
	if (threadIdx.x < 12) xflag[threadIdx.x] = fill ? qs + 2 * ((int)threadIdx.x / 4 + 1) - 1 : qs - 1;
	for (int i = (int)threadIdx.x; i < Q_XROWS * 64; i += 256) xbuf[i] = v4f{ 0.0f, 0.0f, 0.0f, 0.0f };
	__syncthreads();
	const int yg = grp * 16;
	const FrzArgs none{};
#ifdef FX_S4_ALLMID
	run4<RoleMid>(g, p_in, b, p_out, zb, ze, yg + 4 * wave, wave, lane, lds_all + (wave * RoleTop::LDS_ROWS) * 64, xbuf, xflag, none);
	return;
#endif
	if (wave == 0) run4<RoleTop>(g, p_in, b, p_out, zb, ze, yg, wave, lane, lds_all, xbuf, xflag, none);
	else if (wave == 3) run4<RoleBot>(g, p_in, b, p_out, zb, ze, yg + NRO + 2 * NRI, wave, lane, lds_all + (RoleTop::LDS_ROWS + 2 * RoleMid::LDS_ROWS) * 64, xbuf, xflag, none);
	else run4<RoleMid>(g, p_in, b, p_out, zb, ze, yg + NRO + (wave - 1) * NRI, wave, lane, lds_all + (RoleTop::LDS_ROWS + (wave - 1) * RoleMid::LDS_ROWS) * 64, xbuf, xflag, none);
}
#endif

}  // namespace

// read-and-clear of the hand-over fault word on the current device (fx_synchronize, behind the device)
hipError_t strip4_fault_take(unsigned* out)
{
	unsigned v = 0;
	hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_strip4_fault), sizeof v);
	if (e == hipSuccess && v) { const unsigned zero = 0; e = hipMemcpyToSymbol(HIP_SYMBOL(g_strip4_fault), &zero, sizeof zero); }
	*out = v;
	return e;
}

bool jacobi_strip4_supported(const Geom& g)
{
	// the octet takes Y = 14 and any Y >= 17 (bands of 14 rows, shifted where they or their halo would cross the last row: octet_band_y);
	// the quad (STRIP4_OCTET=0) whole bands of 16; X = 512: the half-row octet, bands of six rows
	if (g.Zg > 1 && g.X == 512) return FX_KNOB_INT("STRIP4X", 1) && band_rows_supported(g.Y, X_BAND);
	// any other row of whole quads from 256 cells on: the octet on x tiles (k_jacobi_strip4t); a plane's byte offsets stay below the 2 GiB of the buffer resource its output rows are stored through
	if (g.Zg > 1 && (g.X > 256 || (g.X >= 68 && g.X < 256 && g.X != 128)) && (g.X & 3) == 0) return FX_KNOB_INT("STRIP4T", 1) && octet_rows_supported(g.Y) && (uint64_t)g.X * (uint64_t)g.Y < ((uint64_t)1 << 29);
	if (g.Zg <= 1 || g.X != 256) return false;
	return FX_KNOB_INT("STRIP4_OCTET", 1) ? octet_rows_supported(g.Y) : ((g.Y & 15) == 0 && g.Y >= 16);
}

hipError_t launch_jacobi_strip4(const Geom& g, const float* p_in, const float* b, float* p_out, int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (!jacobi_strip4_supported(g)) return hipErrorNotSupported;
	const int forced_chunk = FX_KNOB_INT("STRIP4_ZCHUNK", 0);
	const int remap = FX_KNOB_INT("STRIP_REMAP", 1);
	if (g.X != 256 || FX_KNOB_INT("STRIP4T_256", 0)) {              // (STRIP4T_256, lab builds: X = 256 as ONE tile of k_jacobi_strip4t -- the octet with runs instead of a grid of chunks)
		// one run of band-planes per workgroup, one workgroup (156 KiB of LDS) per CU: 256 runs wherever a run is at least eight planes long
		const bool tiled = g.X != 512 || (octet_rows_supported(g.Y) && (z_end - z_begin) <= FX_KNOB_INT("STRIP4T_512", 96));
		// k_jacobi_strip4t: a "band" is an x tile of a band of 14 rows.  X = 512 over a thin range of planes (a slab rank's face zones and
		// shrinking rounds) as THREE x tiles: 768 cells computed for 512, but whole pieces on the octet's grid instead of runs of 1.3 six-row
		// bands -- us per sweep k_jacobi_strip4x / tiles at 512 x 512 x D: 16 7.1 / 6.1, 32 11.0 / 9.6, 64 16.4 / 15.7, 96 22.9 / 22.4,
		// 128 29.1 / 29.4, 256 51.0 / 54.4
		const int ntx = tiled ? xtiles(g.X) : 1;
		Runs4 r;
		r.bands = tiled ? ntx * ((g.Y + O_BAND - 1) / O_BAND) : (g.Y + X_BAND - 1) / X_BAND; r.nzp = z_end - z_begin;
		const long long T = (long long)r.bands * r.nzp;
		if (T >= ((long long)1 << 30)) return hipErrorNotSupported;
		const int forced_wgs = FX_KNOB_INT("STRIP4X_WGS", 0);
		r.nwg = forced_wgs > 0 ? forced_wgs : (int)std::min<long long>(256, std::max<long long>(1, T / 8));
		// More bands than CUs (rows of 1024 cells: 5 tiles x 74 bands): runs of 1.45 bands start at any depth, and no two neighbouring bands
		// walk the same planes at the same time -- every halo row comes from memory.  There a workgroup takes ONE piece (a band's planes of
		// one z chunk), more workgroups than CUs: an XCD's 32 CUs start 32 neighbouring bands at the chunk's first plane and walk up
		// together, the next 32 follow as they finish.  1..4 chunks, pieces of STRIP4T_PIECES = 64 planes or more: the count that fills the
		// rounds of 256 best, a piece charged twelve planes of fill; kept to runs where no count fills 85 %.  us per sweep, runs / pieces:
		// 1024^3 1005-1071 / 758-822 (370 bands x 2 chunks = 740 workgroups, 2.9 rounds), 1024 x 1024 x 512 521 / 423, x 256 243 / 193,
		// x 128 120 / 106 (two chunks of 64; one of 128: 121), x 64 59.0 / 62.8 (runs stay); 2048 x 2048 x 128 497 / 388.
		int exact_nch = 0;                                              // > 0: one piece per workgroup, this many z chunks
		const int piece_min = FX_KNOB_INT("STRIP4T_PIECES", 64);
		if (forced_wgs <= 0 && r.bands > 256 && piece_min > 0) {
			double best_score = 0.0;
			for (int n = 1; n <= 4 && r.nzp / n >= piece_min; ++n) {
				const long long P = (long long)r.bands * n;
				const double occ = (double)P / (double)(((P + 255) / 256) * 256), C = (double)r.nzp / n, score = occ * C / (C + 12.0);
				if (occ >= 0.85 && score > best_score + 1e-9) { best_score = score; exact_nch = n; }
			}
		}
		// Short runs (below 64 planes: grids the Infinity Cache holds) on tiles: the octet's own grid -- 256 / bands z chunks of four planes
		// or more, workgroup k = (chunk k / bands, band k % bands): whole pieces, and neighbouring workgroups are neighbouring bands at one
		// depth.  (X = 256 through this kernel, STRIP4T_256: runs 15.0 us per sweep, this grid ..., k_jacobi_strip4o 11.3.)
		if (forced_wgs <= 0 && tiled && !exact_nch && r.bands <= 256 && T / 256 < 64 && FX_KNOB_INT("STRIP4T_GRID", 1)) {
			const int n = std::min(std::max(256 / r.bands, 1), RUNS4_MAXCH);
			const int zc = std::min(std::max((r.nzp + n - 1) / n, 4), r.nzp), nch = (r.nzp + zc - 1) / zc;
			if (r.bands * nch >= 205) exact_nch = nch;                     // (at least 80 % of the CUs: 640 x 640 x 64 -- 138 bands, one chunk -- runs 23.4, this grid 29.9)
		}
		r.exact = 0;
		if (exact_nch) {
			const int C = (r.nzp + exact_nch - 1) / exact_nch;
			r.nch = 0;
			for (int z = 0; z < r.nzp; z += C) r.zc[r.nch++] = z;
			for (int i = r.nch; i <= RUNS4_MAXCH; ++i) r.zc[i] = r.nzp;
			r.nwg = r.bands * r.nch; r.exact = 1;
			r.minp = 1;
		} else {
		r.nwg = std::max(r.nwg, (r.bands + 5) / 6);                     // a run spans at most six bands (+ a head): at most seven pieces (the kernel lists eight)
		r.minp = std::min(FX_KNOB_INT("STRIP4X_MINP", 8), std::max(r.nzp / 2, 1));
		// chunks as long as a run (so that a chunk's runs are its bands), the last one shorter; a stub of a last chunk joins its neighbour;
		// STRIP4X_ORDER=0: one chunk (band-major runs: every band alone at its depth)
		// (where a run is shorter than 64 planes the pieces' fill outweighs the traffic saved -- everything is Infinity-Cache resident there:
		// us per sweep band-major / chunked at 512 x 512 x D: D = 64 17.0 / 20.2, 128 31.1 / 32.7, 256 55.1 / 53.9, 512 100.2 / 96.4)
		const int run = (int)((T + r.nwg - 1) / r.nwg);
		const int order = FX_KNOB_INT("STRIP4X_ORDER", -1);
		int C = (order < 0 ? run >= 64 : order != 0) ? std::min(std::max(run, 1), r.nzp) : r.nzp;
		if ((r.nzp + C - 1) / C > 8) C = (r.nzp + 7) / 8;
		r.nch = 0;
		for (int z = 0; z < r.nzp; z += C) r.zc[r.nch++] = z;
		if (r.nch > 1 && r.nzp - r.zc[r.nch - 1] < std::max(C / 3, r.minp)) --r.nch;
		r.zc[r.nch] = r.nzp;
		for (int i = r.nch + 1; i <= RUNS4_MAXCH; ++i) r.zc[i] = r.nzp;
		}
		// p + b beyond the Infinity Cache (320 MiB at 40 M cells): the output as non-temporal stores
		if (tiled) {
			if (g.cells_local() >= ((size_t)40 << 20) && FX_KNOB_INT("STRIP4X_NT", 1))
				hipLaunchKernelGGL(k_jacobi_strip4t<true>, dim3(r.nwg), dim3(512), 0, s, g, p_in, b, p_out, z_begin, r, ntx, remap);
			else
				hipLaunchKernelGGL(k_jacobi_strip4t<false>, dim3(r.nwg), dim3(512), 0, s, g, p_in, b, p_out, z_begin, r, ntx, remap);
			return hipGetLastError();
		}
		if (g.cells_local() >= ((size_t)40 << 20) && FX_KNOB_INT("STRIP4X_NT", 1))
			hipLaunchKernelGGL(k_jacobi_strip4x<true>, dim3(r.nwg), dim3(512), 0, s, g, p_in, b, p_out, z_begin, r, remap);
		else
			hipLaunchKernelGGL(k_jacobi_strip4x<false>, dim3(r.nwg), dim3(512), 0, s, g, p_in, b, p_out, z_begin, r, remap);
		return hipGetLastError();
	}
	const int ngroups = std::max(g.Y / 16, 1);                          // (the quad's; Y = 14 runs the octet)
	const int nzp = z_end - z_begin;
	int nchunks = (256 + ngroups - 1) / ngroups;                        // 256 workgroups of four waves: one wave per SIMD
	int zchunk = forced_chunk > 0 ? forced_chunk : (nzp + nchunks - 1) / nchunks;
	if (zchunk < 8) zchunk = 8;
	if (zchunk > nzp) zchunk = nzp;
	nchunks = (nzp + zchunk - 1) / zchunk;
	if (FX_KNOB_INT("STRIP4_OCTET", 1) && octet_rows_supported(g.Y)) {     // the octet (two waves per SIMD) is the default; 0 = the quad
		const int bands = (g.Y + O_BAND - 1) / O_BAND;                  // 19 bands of 14 rows at Y = 256 (the last one shifted)
		int nch = 256 / bands;                                          // one workgroup of eight waves per CU
		if (nch < 1) nch = 1;
		int zc = forced_chunk > 0 ? forced_chunk : (nzp + nch - 1) / nch;
		// (chunks of four planes or more -- a forced chunk is taken as given.  The floor was eight until round 6: thin ranges left CUs idle --
		// us per sweep at 256 x 256 x D, floor 8 / 4: D = 16 4.72 / 3.97 (38 / 76 workgroups), 32 4.93 / 3.96, 64 5.24 / 5.02 (chunks of five); from 96 alike.
		// Chunks of two or three planes: level or worse, and more workgroups than CUs much worse -- 64 planes in chunks of four: 7.2)
		const int zfloor = FX_KNOB_INT("STRIP4_ZFLOOR", 4);
		if (zc < zfloor && forced_chunk <= 0) zc = zfloor;
		if (zc > nzp) zc = nzp;
		nch = (nzp + zc - 1) / zc;
		hipLaunchKernelGGL(k_jacobi_strip4o, dim3(bands * nch), dim3(512), 0, s, g, p_in, b, p_out, z_begin, z_end, zc, bands, nch, remap);
	} else {
#ifdef FX_LAB
		hipLaunchKernelGGL(k_jacobi_strip4q, dim3(ngroups * nchunks), dim3(256), 0, s, g, p_in, b, p_out, z_begin, z_end, zchunk, ngroups, nchunks, remap);
#else
		(void)ngroups; (void)nchunks; (void)zchunk;
		return hipErrorNotSupported;
#endif
	}
	return hipGetLastError();
}

bool jacobi_freeze_strip4_supported(const Geom& g)
{
	return g.nz == g.Zg && g.H == 0 && g.X == 256 && octet_rows_supported(g.Y) && g.Zg >= 8 && (uint64_t)g.X * g.Y * (uint64_t)g.Zg < (1u << 30);
}

// levels level_in + 1 .. level_in + 4 for every cell: p_in / m_in -> p_outA = p_outB, m_outA = m_outB; tiles that still relax get `tag`
hipError_t launch_freeze_strip4(const Geom& g, const float* p_in, const float* b, float* p_outA, float* p_outB, const uint8_t* m_in, uint8_t* m_outA, uint8_t* m_outB,
	uint32_t* tile_mark, uint32_t tag, uint32_t* stat, uint32_t stat_hi, int level_in, hipStream_t s)
{
	if (!jacobi_freeze_strip4_supported(g)) return hipErrorNotSupported;
	const int bands = (g.Y + O_BAND - 1) / O_BAND;
	int nch = 256 / bands;
	if (nch < 1) nch = 1;
	int zc = (g.Zg + nch - 1) / nch;
	if (zc < 8) zc = 8;
	if (zc > g.Zg) zc = g.Zg;
	nch = (g.Zg + zc - 1) / zc;
	FrzArgs fa;
	fa.p_outB = p_outB; fa.m_in = m_in; fa.m_outA = m_outA; fa.m_outB = m_outB; fa.tile_mark = tile_mark; fa.tag = tag;
	fa.ntx = (g.X + 31) / 32; fa.nty = (g.Y + 7) / 8; fa.stat = stat; fa.stat_hi = stat_hi; fa.level_in = level_in;
	hipLaunchKernelGGL(k_freeze_strip4o, dim3(bands * nch), dim3(512), 0, s, g, p_in, b, p_outA, fa, zc, bands, nch);
	return hipGetLastError();
}

}  // namespace fx
