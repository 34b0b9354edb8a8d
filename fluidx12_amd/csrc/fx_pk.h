// fx_pk.h -- register PAIRS for the packed FP32 adds of the Jacobi kernels (device code only).
// The x-shifted operands of a row update, (L, c.x | c.y, c.z) and (c.y, c.z | c.w, R), need their cells in even-aligned register
// pairs.  v_pk_mov_b32 builds such a pair from halves of two others in one instruction; the compiler only finds it now and then
// and otherwise spends two v_mov_b32 per pair (six per float4 update), hence inline assembly.
#pragma once
#include <hip/hip_runtime.h>

namespace fx {

typedef float fx_f2 __attribute__((ext_vector_type(2)));

// sel 0: (a.lo, b.lo)   1: (a.hi, b.lo)   2: (a.hi, b.hi)
__device__ __forceinline__ fx_f2 pk_mov(fx_f2 a, fx_f2 b, int sel)
{
	fx_f2 d;
	if (sel == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
	else if (sel == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
	else asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
	return d;
}

// the general form: (a[SA], b[SB])
template <int SA, int SB>
__device__ __forceinline__ fx_f2 pk_mov_sel(fx_f2 a, fx_f2 b)
{
	fx_f2 d;
	if (SA == 0 && SB == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(d) : "v"(a), "v"(b));
	else if (SA == 1 && SB == 0) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
	else if (SA == 1 && SB == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
	else asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
	return d;
}

// ((((((L - b) + R) + U) + D) + F) + B) * (1/6) on a float4 column of a row that IS the wave (X = 256): x neighbours by DPP
// wave_shr:1 / wave_shl:1; the lanes without a source (0 / 63) are the clamped wall cells and keep the DPP's `old` operand.
// left_own / right_own: that end of the wave's row is a wall (the lane keeps its own cell); otherwise it is the cut of a half-row
// wave and the lane takes `edge`, the partner's cell across the cut.
__device__ __forceinline__ float4 relax4_pairs(float4 c, float4 U, float4 D, float4 F, float4 Bk, float4 bb, float edge, bool left_own, bool right_own)
{
	const fx_f2 c01 = { c.x, c.y }, c23 = { c.z, c.w };
	fx_f2 lx = pk_mov(c01, c01, 0);                                  // (c.x, c.x)
	const fx_f2 mid = pk_mov(c01, c23, 1);                           // (c.y, c.z)
	fx_f2 rx = pk_mov(c23, c23, 2);                                  // (c.w, c.w)
	const float ol = left_own ? lx.x : edge, orr = right_own ? rx.y : edge;
	lx.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ol), __builtin_bit_cast(int, c.w), 0x138, 0xf, 0xf, false));
	rx.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, orr), __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
	const fx_f2 b01 = { bb.x, bb.y }, b23 = { bb.z, bb.w }, U01 = { U.x, U.y }, U23 = { U.z, U.w }, D01 = { D.x, D.y }, D23 = { D.z, D.w };
	const fx_f2 F01 = { F.x, F.y }, F23 = { F.z, F.w }, B01 = { Bk.x, Bk.y }, B23 = { Bk.z, Bk.w };
	fx_f2 s01 = (((((lx - b01) + mid) + U01) + D01) + F01) + B01;
	fx_f2 s23 = (((((mid - b23) + rx) + U23) + D23) + F23) + B23;
	const float inv = __uint_as_float(0x3e2aaaabu);
	s01 *= inv; s23 *= inv;
	return make_float4(s01.x, s01.y, s23.x, s23.y);
}

}  // namespace fx
