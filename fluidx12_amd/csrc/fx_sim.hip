// fx_sim.hip -- gfx950 kernels of the simulation step.
//
//   k_advect      <- CSAdvect.hlsl:41-79          (semi-Lagrangian velocity + colour advection, impulse)
//   k_divergence  <- CSProject3D.hlsl:39-50,75-83 (CSProject2D.hlsl:37-46)
//   k_jacobi_*    <- CSPoisson.hlsli:8-26         (lock-step schedule: synchronous ping-pong Jacobi)
//   k_project     <- CSProject3D.hlsl:55-63,105-112 (CSProject2D.hlsl:51-59,99-105)
// (paths relative to /root/reference/FluidX12/Content/Shaders/).  The reference fuses divergence,
// relaxation and projection in one dispatch whose relaxation races by design; here they are separate
// launches on one HIP stream and the relaxation is a deterministic sweep.
//
// Numerics contract: fp32 arithmetic in the association order of the reference's
// shipped DXBC; a DXBC `mad` is fmaf(); nothing else is contracted (-ffp-contract=off).
// Layout: velocity = 3 component planes (SoA), colour = interleaved rgba texels, pressure and
// divergence = fp32 planes; x fastest, one wave64 = 64 consecutive x.  All memory-bound: no MFMA.
#include "fx_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

namespace fx {

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// two x-adjacent texels of one plane in ONE load (4-byte / 2-byte aligned: global memory allows it on gfx950)
struct __attribute__((packed, aligned(4))) F32Pair { float a, b; };
struct __attribute__((packed, aligned(2))) F16Pair { _Float16 a, b; };

template <bool HALF> struct Store;
template <> struct Store<false> {
	typedef float S;
	typedef float4 S4;
	static __device__ __forceinline__ void ld2(const S* p, size_t i, float& a, float& b)
	{
		const F32Pair v = *reinterpret_cast<const F32Pair*>(p + i);
		a = v.a; b = v.b;
	}
	static __device__ __forceinline__ float ld(const S* p, size_t i) { return p[i]; }
	static __device__ __forceinline__ float stored(float v) { return v; }               // the value a later load of this store returns
	static __device__ __forceinline__ void st(S* p, size_t i, float v) { p[i] = v; }
	static __device__ __forceinline__ float4 ld4(const S4* p, size_t i) { return p[i]; }
	static __device__ __forceinline__ void st4(S4* p, size_t i, float4 v) { p[i] = v; }
};
template <> struct Store<true> {
	typedef h16 S;
	typedef h16x4 S4;
	static __device__ __forceinline__ void ld2(const S* p, size_t i, float& a, float& b)
	{
		const F16Pair v = *reinterpret_cast<const F16Pair*>(p + i);
		a = (float)v.a; b = (float)v.b;
	}
	static __device__ __forceinline__ float ld(const S* p, size_t i) { return (float)p[i]; }
	// The fp32 result is rounded to binary16 in a SEPARATE step (RNE), as a typed store of an fp32 register does.  Left to
	// itself the compiler folds the producing multiply/FMA and the conversion into v_fma_mixlo_f16, which rounds the exact
	// product ONCE: different whenever the fp32 rounding lands on a binary16 tie (u * 0.95 does, for dt = 1/4).  The empty
	// asm makes the fp32 value opaque, so the conversion stays a plain v_cvt_f16_f32.
	static __device__ __forceinline__ float rounded_f32(float v) { asm("" : "+v"(v)); return v; }
	static __device__ __forceinline__ float stored(float v) { return (float)(h16)rounded_f32(v); }
	static __device__ __forceinline__ void st(S* p, size_t i, float v) { p[i] = (h16)rounded_f32(v); }   // RNE
	static __device__ __forceinline__ float4 ld4(const S4* p, size_t i)
	{
		const h16x4 h = p[i];
		return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
	}
	static __device__ __forceinline__ void st4(S4* p, size_t i, float4 v)
	{
		h16x4 h;
		h.x = (h16)rounded_f32(v.x); h.y = (h16)rounded_f32(v.y); h.z = (h16)rounded_f32(v.z); h.w = (h16)rounded_f32(v.w);
		p[i] = h;
	}
};

__device__ __forceinline__ float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }
__device__ __forceinline__ float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// ---- XCD-aware workgroup -> tile mapping ----------------------------------------------------------
// MI355X dispatches workgroup L to XCD L % 8 and every XCD has a private 4 MiB L2.  With the natural order
// (tile = L) the eight XCDs interleave at tile granularity, so every stencil halo row/plane is fetched by two
// L2s.  Here XCD k instead walks the k-th contiguous eighth of the tile sequence (tiles ordered x, then y,
// then z): its halo traffic shrinks to the two ends of its z range.  Speed only, never correctness.
struct Tile3 { int x, y, z; };
// remap: 0 = natural order; 1 = XCD k walks the k-th contiguous eighth of the (x, y, z)-ordered tile sequence (a z range);
// 2 = XCD k owns the k-th y-band of EVERY plane and walks it plane by plane (needs gy % 8 == 0, else natural order):
// the per-XCD working set per plane is 1/8 plane, so z-neighbour planes stay in its L2 even when a plane of all fields
// exceeds 4 MiB (advection: 1.75 MiB of velocity + colour per 256^2 plane).
__device__ __forceinline__ Tile3 xcd_tile(int gx, int gy, int gz, int remap)
{
	const int n = gx * gy * gz;
	int t = (int)blockIdx.x;
	Tile3 o;
	if (remap == 2 && (gy & 7) == 0) {
		const int band = gy >> 3, xcd = t & 7, j = t >> 3;
		o.x = j % gx;
		const int u = j / gx;
		o.y = xcd * band + u % band;
		o.z = u / band;
		return o;
	}
	if (remap == 1) {
		const int q = n >> 3, r = n & 7;
		const int xcd = t & 7, j = t >> 3;
		t = xcd * q + min(xcd, r) + j;
	}
	o.x = t % gx;
	const int u = t / gx;
	o.y = u % gy;
	o.z = u / gy;
	return o;
}

// D3D addressing of an integer tap (CLAMP / MIRROR)
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

// ---------------------------------------------------------------------------------------------
// advection
// ---------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void k_advect(const Geom g, const SimParams sp,
	const typename Store<HALF>::S* __restrict__ vel_in, const typename Store<HALF>::S4* __restrict__ col_in,
	typename Store<HALF>::S* __restrict__ vel_out, typename Store<HALF>::S4* __restrict__ col_out,
	int z_begin, int nzp, int remap, unsigned* halo_overflow, float* __restrict__ alpha_out)
{
	typedef Store<HALF> St;
	const int BX = blockDim.x, BY = blockDim.y, BZ = blockDim.z;
	const Tile3 tile = xcd_tile((g.X + BX - 1) / BX, (g.Y + BY - 1) / BY, (nzp + BZ - 1) / BZ, remap);
	const int x = tile.x * BX + threadIdx.x;
	const int y = tile.y * BY + threadIdx.y;
	const int z = z_begin + tile.z * BZ + threadIdx.z;
	if (x >= g.X || y >= g.Y || z >= z_begin + nzp) return;

	const size_t plane = g.plane();
	const size_t stride = g.cells_local();                  // distance between velocity component planes
	const size_t id = (size_t)g.lz(z) * plane + (size_t)y * g.X + x;
	const float dt = sp.dt;

	const float px = ((float)x + 0.5f) / (float)g.X;        // Simulation.hlsli:10
	const float py = ((float)y + 0.5f) / (float)g.Y;
	const float pz = ((float)z + 0.5f) / (float)g.Zg;
	const float dx = px + -0.5f, dy = py + -0.100000001f, dz = pz + -0.5f;   // Impulse.hlsli:14
	const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
	const float rr = sp.is3d ? 0.00390625f : 0.0009765625f;
	const float basis = exp2f(((d2 * -4.0f) / rr) * 1.44269502f);           // CSAdvect.hlsl:33-36,59
	float Fx, Fy, Fz;
	if (sp.is3d) {                                                           // :63-65
		Fx = fmaf(basis, 0.0f, dz * -200.0f);
		Fy = fmaf(basis, 192.0f, 0.0f);
		Fz = fmaf(basis, 0.0f, dx * 200.0f);
	} else {
		Fx = 0.0f; Fy = basis * 48.0f; Fz = 0.0f;
	}

	const float u0x = St::ld(vel_in, id), u0y = St::ld(vel_in, stride + id), u0z = St::ld(vel_in, 2 * stride + id);   // :47
	const float ax = fmaf(-u0x, dt, px), ay = fmaf(-u0y, dt, py), az = fmaf(-u0z, dt, pz);                             // :52

	// trilinear taps: t = u*N - 0.5, i0 = floor(t), f = t - i0
	const float tx = ax * (float)g.X - 0.5f, ty = ay * (float)g.Y - 0.5f, tz = az * (float)g.Zg - 0.5f;
	const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
	const float fx = tx - flx, fy = ty - fly, fz = tz - flz;
	const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
	const int x0 = addr_tap(ix, g.X, sp.address), x1 = addr_tap(ix + 1, g.X, sp.address);
	const int y0 = addr_tap(iy, g.Y, sp.address), y1 = addr_tap(iy + 1, g.Y, sp.address);
	int z0 = addr_tap(iz, g.Zg, sp.address), z1 = addr_tap(iz + 1, g.Zg, sp.address);
	if (z0 < g.zlo || z0 > g.zhi || z1 < g.zlo || z1 > g.zhi) {             // back-trace left the exchanged halo
		atomicOr(halo_overflow, 1u);
		z0 = min(max(z0, g.zlo), g.zhi);
		z1 = min(max(z1, g.zlo), g.zhi);
	}
	const size_t r00 = (size_t)g.lz(z0) * plane + (size_t)y0 * g.X, r10 = (size_t)g.lz(z0) * plane + (size_t)y1 * g.X;
	const size_t r01 = (size_t)g.lz(z1) * plane + (size_t)y0 * g.X, r11 = (size_t)g.lz(z1) * plane + (size_t)y1 * g.X;

	float u[3];
	if (g.X >= 2) {
		// the two x-taps of a row are adjacent (or equal, at a clamped/mirrored border): fetch them as one 8-byte
		// (4-byte for fp16) load at xa = min(x0, x1) and select -- 12 gathers instead of 24 (the kernel is TA-issue bound)
		const int xa = min(min(x0, x1), g.X - 2);
		const bool lo0 = x0 == xa, lo1 = x1 == xa;
#pragma unroll
		for (int a = 0; a < 3; ++a) {                                        // :53
			const typename St::S* f = vel_in + a * stride;
			float l00, h00, l10, h10, l01, h01, l11, h11;
			St::ld2(f, r00 + xa, l00, h00); St::ld2(f, r10 + xa, l10, h10);
			St::ld2(f, r01 + xa, l01, h01); St::ld2(f, r11 + xa, l11, h11);
			const float c00 = lerpf(lo0 ? l00 : h00, lo1 ? l00 : h00, fx);
			const float c10 = lerpf(lo0 ? l10 : h10, lo1 ? l10 : h10, fx);
			const float c01 = lerpf(lo0 ? l01 : h01, lo1 ? l01 : h01, fx);
			const float c11 = lerpf(lo0 ? l11 : h11, lo1 ? l11 : h11, fx);
			u[a] = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
		}
	} else {
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			const typename St::S* f = vel_in + a * stride;
			const float c00 = lerpf(St::ld(f, r00 + x0), St::ld(f, r00 + x1), fx);
			const float c10 = lerpf(St::ld(f, r10 + x0), St::ld(f, r10 + x1), fx);
			const float c01 = lerpf(St::ld(f, r01 + x0), St::ld(f, r01 + x1), fx);
			const float c11 = lerpf(St::ld(f, r11 + x0), St::ld(f, r11 + x1), fx);
			u[a] = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
		}
	}
	float c[4];
	{                                                                        // :54
		const float4 t000 = St::ld4(col_in, r00 + x0), t100 = St::ld4(col_in, r00 + x1);
		const float4 t010 = St::ld4(col_in, r10 + x0), t110 = St::ld4(col_in, r10 + x1);
		const float4 t001 = St::ld4(col_in, r01 + x0), t101 = St::ld4(col_in, r01 + x1);
		const float4 t011 = St::ld4(col_in, r11 + x0), t111 = St::ld4(col_in, r11 + x1);
#define FX_TRI(m) lerpf(lerpf(lerpf(t000.m, t100.m, fx), lerpf(t010.m, t110.m, fx), fy), \
	lerpf(lerpf(t001.m, t101.m, fx), lerpf(t011.m, t111.m, fx), fy), fz)
		c[0] = FX_TRI(x); c[1] = FX_TRI(y); c[2] = FX_TRI(z); c[3] = FX_TRI(w);
#undef FX_TRI
	}

	if (basis >= 0.0183156393f) {                                            // :60
		u[0] = fmaf(Fx, dt, u[0]); u[1] = fmaf(Fy, dt, u[1]); u[2] = fmaf(Fz, dt, u[2]);   // :66
		const float bdt = basis * dt;
		c[0] = saturatef(fmaf(bdt, 8.0f, c[0]));                             // :67, g_impulse = (.2,.4,1,1)*40
		c[1] = saturatef(fmaf(bdt, 16.0f, c[1]));
		c[2] = saturatef(fmaf(bdt, 40.0f, c[2]));
		c[3] = saturatef(fmaf(bdt, 40.0f, c[3]));
	}
	const float atten = fmaxf(fmaf(-dt, 0.200000003f, 1.0f), 0.0f);          // :74
	St::st(vel_out, id, u[0] * atten);                                       // :77
	St::st(vel_out, stride + id, u[1] * atten);
	St::st(vel_out, 2 * stride + id, u[2] * atten);
	St::st4(col_out, id, make_float4(c[0] * atten, c[1] * atten, c[2] * atten, c[3] * atten));   // :78
	if (alpha_out) alpha_out[id] = St::stored(c[3] * atten);                  // the render's alpha side volume (fx_render_accel.hip), unsliced grids only
}

// ---------------------------------------------------------------------------------------------
// advection, lean path: the same arithmetic for grids whose extents are powers of two and whose fields are < 4 GiB --
// every BASELINE config -- in 560 instead of 983 instructions per wave64: the three coordinate divisions become
// multiplications by the exact reciprocal (x / 2^k == x * 2^-k bit for bit), so does (d2 * -4) / r^2 (r^2 = 2^-8 or
// 2^-10); taps are addressed with 32-bit byte offsets from uniform bases (saddr + voffset loads instead of a 64-bit add
// per tap, one offset serves the three velocity planes); index products are shifts; the x-pair select logic is gone (the
// compiler split the unaligned pair load into two dword loads anyway).  Bit-identical to k_advect
// (tests/test_gpu_sim.py::test_advect_fast_path_bit_identical).  Measured: 0.3075 -> 0.2955 ms at 256^3 -- the kernel is
// bound by its gathers' latency/L1 path, not by instruction issue (fp16 storage: 0.21 ms); sharing the x+1 taps between
// neighbouring lanes by wave shuffles (12 + 4 loads instead of 24 + 8) made it SLOWER (0.311 ms) and was dropped.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T ldg32(const void* base, uint32_t byte_off)
{
	return *reinterpret_cast<const T*>(static_cast<const char*>(base) + byte_off);
}
template <bool HALF> struct Fetch;
template <> struct Fetch<false> {
	static __device__ __forceinline__ float s(const void* b, uint32_t cell) { return ldg32<float>(b, cell * 4u); }
	static __device__ __forceinline__ float4 v(const void* b, uint32_t cell) { return ldg32<float4>(b, cell * 16u); }
};
template <> struct Fetch<true> {
	static __device__ __forceinline__ float s(const void* b, uint32_t cell) { return (float)ldg32<h16>(b, cell * 2u); }
	static __device__ __forceinline__ float4 v(const void* b, uint32_t cell)
	{
		const h16x4 h = ldg32<h16x4>(b, cell * 8u);
		return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
	}
};

template <bool HALF>
__global__ __launch_bounds__(256) void k_advect_fast(const Geom g, const SimParams sp,
	const typename Store<HALF>::S* __restrict__ vel_in, const typename Store<HALF>::S4* __restrict__ col_in,
	typename Store<HALF>::S* __restrict__ vel_out, typename Store<HALF>::S4* __restrict__ col_out,
	int z_begin, int nzp, unsigned* halo_overflow, float rX, float rY, float rZ, float inv_rr, int lgX, int lgY, int lg_gx, int lg_gy, float* __restrict__ alpha_out)
{
	typedef Store<HALF> St;
	typedef Fetch<HALF> Ft;
	// 64 x 4 x 1 voxels per workgroup, tiles ordered x, y, z; tile counts and extents are powers of two: shifts, no multiplies
	const int tl = (int)blockIdx.x;
	const int x = ((tl & ((1 << lg_gx) - 1)) << 6) + (int)threadIdx.x;
	const int y = (((tl >> lg_gx) & ((1 << lg_gy) - 1)) << 2) + (int)threadIdx.y;
	const int z = z_begin + (tl >> (lg_gx + lg_gy));
	if (x >= g.X || y >= g.Y) return;

	const int lgP = lgX + lgY;
	const uint32_t stride = (uint32_t)g.nzl() << lgP;               // cells between velocity component planes
	const uint32_t id = ((uint32_t)g.lz(z) << lgP) + ((uint32_t)y << lgX) + (uint32_t)x;
	const float dt = sp.dt;
	const size_t esz = HALF ? 2 : 4;
	const char* v0 = reinterpret_cast<const char*>(vel_in);
	const char* v1 = v0 + (size_t)stride * esz;
	const char* v2 = v1 + (size_t)stride * esz;

	const float px = ((float)x + 0.5f) * rX;                        // == / (float)g.X, exactly (power of two)
	const float py = ((float)y + 0.5f) * rY;
	const float pz = ((float)z + 0.5f) * rZ;
	const float dx = px + -0.5f, dy = py + -0.100000001f, dz = pz + -0.5f;
	const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
	const float basis = exp2f(((d2 * -4.0f) * inv_rr) * 1.44269502f);
	float Fx, Fy, Fz;
	if (sp.is3d) {
		Fx = fmaf(basis, 0.0f, dz * -200.0f);
		Fy = fmaf(basis, 192.0f, 0.0f);
		Fz = fmaf(basis, 0.0f, dx * 200.0f);
	} else {
		Fx = 0.0f; Fy = basis * 48.0f; Fz = 0.0f;
	}

	const float u0x = Ft::s(v0, id), u0y = Ft::s(v1, id), u0z = Ft::s(v2, id);
	const float ax = fmaf(-u0x, dt, px), ay = fmaf(-u0y, dt, py), az = fmaf(-u0z, dt, pz);
	const float tx = ax * (float)g.X - 0.5f, ty = ay * (float)g.Y - 0.5f, tz = az * (float)g.Zg - 0.5f;
	const float flx = floorf(tx), fly = floorf(ty), flz = floorf(tz);
	const float fx = tx - flx, fy = ty - fly, fz = tz - flz;
	const int ix = (int)flx, iy = (int)fly, iz = (int)flz;
	const int x0 = addr_tap(ix, g.X, sp.address), x1 = addr_tap(ix + 1, g.X, sp.address);
	const int y0 = addr_tap(iy, g.Y, sp.address), y1 = addr_tap(iy + 1, g.Y, sp.address);
	int z0 = addr_tap(iz, g.Zg, sp.address), z1 = addr_tap(iz + 1, g.Zg, sp.address);
	if (z0 < g.zlo || z0 > g.zhi || z1 < g.zlo || z1 > g.zhi) {
		atomicOr(halo_overflow, 1u);
		z0 = min(max(z0, g.zlo), g.zhi);
		z1 = min(max(z1, g.zlo), g.zhi);
	}
	const uint32_t p0 = (uint32_t)g.lz(z0) << lgP, p1 = (uint32_t)g.lz(z1) << lgP;
	const uint32_t ry0 = (uint32_t)y0 << lgX, ry1 = (uint32_t)y1 << lgX;
	const uint32_t c000 = p0 + ry0 + (uint32_t)x0, c100 = p0 + ry0 + (uint32_t)x1;
	const uint32_t c010 = p0 + ry1 + (uint32_t)x0, c110 = p0 + ry1 + (uint32_t)x1;
	const uint32_t c001 = p1 + ry0 + (uint32_t)x0, c101 = p1 + ry0 + (uint32_t)x1;
	const uint32_t c011 = p1 + ry1 + (uint32_t)x0, c111 = p1 + ry1 + (uint32_t)x1;

	float u[3];
	const char* vb[3] = { v0, v1, v2 };
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		const float c00 = lerpf(Ft::s(vb[a], c000), Ft::s(vb[a], c100), fx);
		const float c10 = lerpf(Ft::s(vb[a], c010), Ft::s(vb[a], c110), fx);
		const float c01 = lerpf(Ft::s(vb[a], c001), Ft::s(vb[a], c101), fx);
		const float c11 = lerpf(Ft::s(vb[a], c011), Ft::s(vb[a], c111), fx);
		u[a] = lerpf(lerpf(c00, c10, fy), lerpf(c01, c11, fy), fz);
	}
	float c[4];
	{
		const float4 t000 = Ft::v(col_in, c000), t100 = Ft::v(col_in, c100);
		const float4 t010 = Ft::v(col_in, c010), t110 = Ft::v(col_in, c110);
		const float4 t001 = Ft::v(col_in, c001), t101 = Ft::v(col_in, c101);
		const float4 t011 = Ft::v(col_in, c011), t111 = Ft::v(col_in, c111);
#define FX_TRI(m) lerpf(lerpf(lerpf(t000.m, t100.m, fx), lerpf(t010.m, t110.m, fx), fy), \
	lerpf(lerpf(t001.m, t101.m, fx), lerpf(t011.m, t111.m, fx), fy), fz)
		c[0] = FX_TRI(x); c[1] = FX_TRI(y); c[2] = FX_TRI(z); c[3] = FX_TRI(w);
#undef FX_TRI
	}
	if (basis >= 0.0183156393f) {
		u[0] = fmaf(Fx, dt, u[0]); u[1] = fmaf(Fy, dt, u[1]); u[2] = fmaf(Fz, dt, u[2]);
		const float bdt = basis * dt;
		c[0] = saturatef(fmaf(bdt, 8.0f, c[0]));
		c[1] = saturatef(fmaf(bdt, 16.0f, c[1]));
		c[2] = saturatef(fmaf(bdt, 40.0f, c[2]));
		c[3] = saturatef(fmaf(bdt, 40.0f, c[3]));
	}
	const float atten = fmaxf(fmaf(-dt, 0.200000003f, 1.0f), 0.0f);
	St::st(vel_out, id, u[0] * atten);
	St::st(vel_out, (size_t)stride + id, u[1] * atten);
	St::st(vel_out, 2 * (size_t)stride + id, u[2] * atten);
	St::st4(col_out, id, make_float4(c[0] * atten, c[1] * atten, c[2] * atten, c[3] * atten));
	if (alpha_out) alpha_out[id] = St::stored(c[3] * atten);
}

// ---------------------------------------------------------------------------------------------
// divergence  b = 0.5 * ((fB - fF) + ((fD - fU) + (fR - fL)))   (2D: (fR - fL) + (fD - fU))
// ---------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void k_divergence(const Geom g, const typename Store<HALF>::S* __restrict__ vel,
	float* __restrict__ b, int z_begin, int nzp, int remap)
{
	typedef Store<HALF> St;
	const Tile3 tile = xcd_tile((g.X + 63) >> 6, (g.Y + 3) >> 2, nzp, remap);
	const int x = tile.x * 64 + threadIdx.x;
	const int y = tile.y * 4 + threadIdx.y;
	const int z = z_begin + tile.z;
	if (x >= g.X || y >= g.Y) return;
	const size_t plane = g.plane(), stride = g.cells_local();
	const size_t row = (size_t)g.lz(z) * plane + (size_t)y * g.X;
	const int xl = max(x, 1) - 1, xr = min(x + 1, g.X - 1);
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	const float ddx = -St::ld(vel, row + xl) + St::ld(vel, row + xr);
	const float ddy = -St::ld(vel, stride + (size_t)g.lz(z) * plane + (size_t)yu * g.X + x)
		+ St::ld(vel, stride + (size_t)g.lz(z) * plane + (size_t)yd * g.X + x);
	float S;
	if (g.Zg > 1) {
		const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
		const float ddz = -St::ld(vel, 2 * stride + (size_t)g.lz(zf) * plane + (size_t)y * g.X + x)
			+ St::ld(vel, 2 * stride + (size_t)g.lz(zb) * plane + (size_t)y * g.X + x);
		S = ddz + (ddy + ddx);
	} else {
		S = ddx + ddy;
	}
	b[row + x] = 0.5f * S;
}

// 3-D fast path: one thread = 4 consecutive x (16-B / 8-B loads of the y and z neighbour rows, 16-B store of b)
template <bool HALF>
__global__ __launch_bounds__(256) void k_divergence_v4(const Geom g, const typename Store<HALF>::S* __restrict__ vel, float* __restrict__ b,
	int z_begin, int nzp, int remap, int rows_per_block)
{
	typedef Store<HALF> St;
	typedef typename St::S4 S4;
	const int X4 = g.X >> 2;
	const Tile3 tile = xcd_tile((X4 + (int)blockDim.x - 1) / (int)blockDim.x, (g.Y + rows_per_block - 1) / rows_per_block, nzp, remap);
	const int x4 = tile.x * blockDim.x + threadIdx.x;
	const int y = tile.y * rows_per_block + threadIdx.y;
	const int z = z_begin + tile.z;
	if (x4 >= X4 || y >= g.Y) return;
	const size_t plane = g.plane(), stride = g.cells_local();
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
	const size_t zrow = (size_t)g.lz(z) * plane, off = zrow + (size_t)y * g.X + 4 * x4;
	const float4 cx = St::ld4(reinterpret_cast<const S4*>(vel + off), 0);
	const float L = x4 > 0 ? St::ld(vel, off - 1) : cx.x;
	const float R = x4 < X4 - 1 ? St::ld(vel, off + 4) : cx.w;
	const float4 U = St::ld4(reinterpret_cast<const S4*>(vel + stride + zrow + (size_t)yu * g.X + 4 * x4), 0);
	const float4 D = St::ld4(reinterpret_cast<const S4*>(vel + stride + zrow + (size_t)yd * g.X + 4 * x4), 0);
	const float4 F = St::ld4(reinterpret_cast<const S4*>(vel + 2 * stride + (size_t)g.lz(zf) * plane + (size_t)y * g.X + 4 * x4), 0);
	const float4 B = St::ld4(reinterpret_cast<const S4*>(vel + 2 * stride + (size_t)g.lz(zb) * plane + (size_t)y * g.X + 4 * x4), 0);
	float4 o;
	o.x = 0.5f * ((-F.x + B.x) + ((-U.x + D.x) + (-L + cx.y)));
	o.y = 0.5f * ((-F.y + B.y) + ((-U.y + D.y) + (-cx.x + cx.z)));
	o.z = 0.5f * ((-F.z + B.z) + ((-U.z + D.z) + (-cx.y + cx.w)));
	o.w = 0.5f * ((-F.w + B.w) + ((-U.w + D.w) + (-cx.z + R)));
	*reinterpret_cast<float4*>(b + off) = o;
}

// ---------------------------------------------------------------------------------------------
// Jacobi sweep, generic: any extent, 2D/3D, optional freeze mask (faithful early-out)
//   x = ((((((qL - b) + qR) + qU) + qD) + qF) + qB) * (1/6)     2D: (((qL - b) + qR) + qU) + qD) * 1/4
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_jacobi_generic(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, uint8_t* __restrict__ frozen, int z_begin, int nzp, int remap,
	int split, int z_begin2)
{
	const Tile3 tile = xcd_tile((g.X + 63) >> 6, (g.Y + 3) >> 2, nzp, remap);
	const int x = tile.x * 64 + threadIdx.x;
	const int y = tile.y * 4 + threadIdx.y;
	const int z = tile.z < split ? z_begin + tile.z : z_begin2 + (tile.z - split);    // two plane ranges in one launch
	if (x >= g.X || y >= g.Y) return;
	const size_t plane = g.plane();
	const size_t zrow = (size_t)g.lz(z) * plane;
	const size_t id = zrow + (size_t)y * g.X + x;
	const float x0 = p_in[id];
	if (frozen && frozen[id]) { p_out[id] = x0; return; }
	const int xl = max(x, 1) - 1, xr = min(x + 1, g.X - 1);
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	float s = p_in[zrow + (size_t)y * g.X + xl] - b[id];
	s = p_in[zrow + (size_t)y * g.X + xr] + s;
	s = p_in[zrow + (size_t)yu * g.X + x] + s;
	s = p_in[zrow + (size_t)yd * g.X + x] + s;
	float inv = 0.25f;
	if (g.Zg > 1) {
		const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
		s = p_in[(size_t)g.lz(zf) * plane + (size_t)y * g.X + x] + s;
		s = p_in[(size_t)g.lz(zb) * plane + (size_t)y * g.X + x] + s;
		inv = __uint_as_float(0x3e2aaaabu);
	}
	p_out[id] = s * inv;
	if (frozen && fabsf(fmaf(s, inv, -x0)) < 0.00100000005f) frozen[id] = 1;     // CSPoisson.hlsli:24
}

// ---------------------------------------------------------------------------------------------
// Jacobi sweep, 3D fast path: one thread = 4 consecutive x (16 B loads/stores), X % 4 == 0.
// Neighbour rows/planes come through L1/L2 (each line is re-read by the 5 stencil partners).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_jacobi_v4(const Geom g, const float* __restrict__ p_in,
	const float* __restrict__ b, float* __restrict__ p_out, int z_begin, int nzp, int remap, int rows_per_block,
	int split, int z_begin2)
{
	const int X4 = g.X >> 2;
	const int lane = threadIdx.x;                       // float4 column
	const Tile3 tile = xcd_tile((X4 + (int)blockDim.x - 1) / (int)blockDim.x, (g.Y + rows_per_block - 1) / rows_per_block, nzp, remap);
	const int x4 = tile.x * blockDim.x + lane;
	const int y = tile.y * rows_per_block + threadIdx.y;
	const int z = tile.z < split ? z_begin + tile.z : z_begin2 + (tile.z - split);    // two plane ranges in one launch
	if (x4 >= X4 || y >= g.Y) return;
	const size_t plane = g.plane();
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
	const size_t zrow = (size_t)g.lz(z) * plane;
	const size_t c_off = zrow + (size_t)y * g.X + 4 * x4;
	const float4 c = *reinterpret_cast<const float4*>(p_in + c_off);
	const float4 U = *reinterpret_cast<const float4*>(p_in + zrow + (size_t)yu * g.X + 4 * x4);
	const float4 D = *reinterpret_cast<const float4*>(p_in + zrow + (size_t)yd * g.X + 4 * x4);
	const float4 F = *reinterpret_cast<const float4*>(p_in + (size_t)g.lz(zf) * plane + (size_t)y * g.X + 4 * x4);
	const float4 B = *reinterpret_cast<const float4*>(p_in + (size_t)g.lz(zb) * plane + (size_t)y * g.X + 4 * x4);
	const float4 bb = *reinterpret_cast<const float4*>(b + c_off);
	// x neighbours: the adjacent float4 column sits in the adjacent lane (DPP wave_shr:1 / wave_shl:1, see fx_jacobi_strip.hip);
	// only a wave's first / last lane inside a row (X > 256, or the tail of a partial wave) still loads them
	const int wl = (int)((threadIdx.y * blockDim.x + threadIdx.x) & 63);
	float L = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.w), 0x138, 0xf, 0xf, false));
	float R = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
	if (x4 == 0) L = c.x; else if (wl == 0 || lane == 0) L = p_in[c_off - 1];
	if (x4 == X4 - 1) R = c.w; else if (wl == 63 || lane == (int)blockDim.x - 1) R = p_in[c_off + 4];
	const float inv = __uint_as_float(0x3e2aaaabu);
	float4 o;
	o.x = ((((((L - bb.x) + c.y) + U.x) + D.x) + F.x) + B.x) * inv;
	o.y = ((((((c.x - bb.y) + c.z) + U.y) + D.y) + F.y) + B.y) * inv;
	o.z = ((((((c.y - bb.z) + c.w) + U.z) + D.z) + F.z) + B.z) * inv;
	o.w = ((((((c.z - bb.w) + R) + U.w) + D.w) + F.w) + B.w) * inv;
	*reinterpret_cast<float4*>(p_out + c_off) = o;
}

// ---------------------------------------------------------------------------------------------
// projection + wall damping
// ---------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void k_project(const Geom g, const SimParams sp,
	const typename Store<HALF>::S* __restrict__ vel_in, const float* __restrict__ p,
	typename Store<HALF>::S* __restrict__ vel_out, int z_begin, int nzp, int remap)
{
	typedef Store<HALF> St;
	const Tile3 tile = xcd_tile((g.X + 63) >> 6, (g.Y + 3) >> 2, nzp, remap);
	const int x = tile.x * 64 + threadIdx.x;
	const int y = tile.y * 4 + threadIdx.y;
	const int z = z_begin + tile.z;
	if (x >= g.X || y >= g.Y) return;
	const size_t plane = g.plane(), stride = g.cells_local();
	const size_t zrow = (size_t)g.lz(z) * plane;
	const size_t id = zrow + (size_t)y * g.X + x;
	const int xl = max(x, 1) - 1, xr = min(x + 1, g.X - 1);
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	float u[3] = { St::ld(vel_in, id), St::ld(vel_in, stride + id), St::ld(vel_in, 2 * stride + id) };
	float grad[3];
	grad[0] = -p[zrow + (size_t)y * g.X + xl] + p[zrow + (size_t)y * g.X + xr];
	grad[1] = -p[zrow + (size_t)yu * g.X + x] + p[zrow + (size_t)yd * g.X + x];
	grad[2] = 0.0f;
	float k = 0.5f;
	if (sp.is3d) {
		const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
		grad[2] = -p[(size_t)g.lz(zf) * plane + (size_t)y * g.X + x] + p[(size_t)g.lz(zb) * plane + (size_t)y * g.X + x];
		k = __uint_as_float(0x3f855556u);                                  // 0.5f / 0.48f (g_density, CSProject3D.hlsl:26)
		u[2] = fmaf(-grad[2], k, u[2]);
	}
	u[0] = fmaf(-grad[0], k, u[0]);                                        // CSProject3D.hlsl:62
	u[1] = fmaf(-grad[1], k, u[1]);
	const int cell[3] = { x, y, z };
	const float dims[3] = { (float)g.X, (float)g.Y, (float)g.Zg };
#pragma unroll
	for (int a = 0; a < 3; ++a) {                                          // :106-108
		float pos = ((float)cell[a] + 0.5f) / dims[a];
		if (sp.is3d || a < 2) pos = fmaf(pos, 2.0f, -1.0f);
		float f = (-fabsf(pos) + 0.970000029f) * 33.3333359f;
		f = fminf(fmaxf(f, -1.0f), 1.0f);
		const float w = (0.0f < u[a] * pos) ? f : 1.0f;
		St::st(vel_out, a * stride + id, u[a] * w);
	}
}

// fp32 3-D fast path of the projection: one thread = 4 consecutive x (16-B loads of the three velocity components and of the
// pressure rows above / below / in front / behind, 16-B stores), the x neighbours of the pressure row through DPP lane shifts
// like k_jacobi_v4, 32-bit offsets.  Per-cell arithmetic is k_project's; RCP: extents are powers of two and the three
// coordinate divisions become multiplications by the exact reciprocal (bit-identical).
template <bool RCP, bool HALF>
__global__ __launch_bounds__(256) void k_project_v4(const Geom g, const typename Store<HALF>::S* __restrict__ vel_in, const float* __restrict__ p,
	typename Store<HALF>::S* __restrict__ vel_out, int z_begin, int nzp, int remap, int rows_per_block, float rX, float rY, float rZ,
	int* __restrict__ rec, float dt, int address, int digest, const unsigned* __restrict__ halo_overflow)
{
	const int X4 = g.X >> 2;
	const int lane = threadIdx.x;
	// slab ranks: this launch also measures what the next advection will need from the z-neighbours (see k_face_need) and
	// closes the step's record -- the projected u_z is in registers here anyway
	if (rec && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) { rec[2] = digest; rec[3] = (int)*halo_overflow; }
	const Tile3 tile = xcd_tile((X4 + (int)blockDim.x - 1) / (int)blockDim.x, (g.Y + rows_per_block - 1) / rows_per_block, nzp, remap);
	const int x4 = tile.x * blockDim.x + lane;
	const int y = tile.y * rows_per_block + threadIdx.y;
	const int z = z_begin + tile.z;
	if (x4 >= X4 || y >= g.Y) return;
	const uint32_t plane = (uint32_t)g.plane(), stride = (uint32_t)g.cells_local();
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
	const uint32_t zrow = (uint32_t)g.lz(z) * plane;
	const uint32_t off = zrow + (uint32_t)y * g.X + 4u * x4;
	const float4 c = *reinterpret_cast<const float4*>(p + off);
	const float4 U = *reinterpret_cast<const float4*>(p + zrow + (uint32_t)yu * g.X + 4u * x4);
	const float4 D = *reinterpret_cast<const float4*>(p + zrow + (uint32_t)yd * g.X + 4u * x4);
	const float4 F = *reinterpret_cast<const float4*>(p + (uint32_t)g.lz(zf) * plane + (uint32_t)y * g.X + 4u * x4);
	const float4 B = *reinterpret_cast<const float4*>(p + (uint32_t)g.lz(zb) * plane + (uint32_t)y * g.X + 4u * x4);
	typedef Store<HALF> St;
	typedef typename St::S4 S4;
	const float4 ux = St::ld4(reinterpret_cast<const S4*>(vel_in + off), 0);
	const float4 uy = St::ld4(reinterpret_cast<const S4*>(vel_in + stride + off), 0);
	const float4 uz = St::ld4(reinterpret_cast<const S4*>(vel_in + 2u * stride + off), 0);
	const int wl = (int)((threadIdx.y * blockDim.x + threadIdx.x) & 63);
	float L = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.w), 0x138, 0xf, 0xf, false));
	float R = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c.x), 0x130, 0xf, 0xf, false));
	if (x4 == 0) L = c.x; else if (wl == 0 || lane == 0) L = p[off - 1];
	if (x4 == X4 - 1) R = c.w; else if (wl == 63 || lane == (int)blockDim.x - 1) R = p[off + 4];
	const float k = __uint_as_float(0x3f855556u);                          // 0.5f / 0.48f (g_density, CSProject3D.hlsl:26)
	const float pl[4] = { L, c.x, c.y, c.z }, pr[4] = { c.y, c.z, c.w, R };
	const float pu[4] = { U.x, U.y, U.z, U.w }, pd[4] = { D.x, D.y, D.z, D.w };
	const float pf[4] = { F.x, F.y, F.z, F.w }, pb[4] = { B.x, B.y, B.z, B.w };
	const float vx[4] = { ux.x, ux.y, ux.z, ux.w }, vy[4] = { uy.x, uy.y, uy.z, uy.w }, vz[4] = { uz.x, uz.y, uz.z, uz.w };
	float py = RCP ? ((float)y + 0.5f) * rY : ((float)y + 0.5f) / (float)g.Y;
	float pz = RCP ? ((float)z + 0.5f) * rZ : ((float)z + 0.5f) / (float)g.Zg;
	py = fmaf(py, 2.0f, -1.0f); pz = fmaf(pz, 2.0f, -1.0f);
	float fy = (-fabsf(py) + 0.970000029f) * 33.3333359f, fz = (-fabsf(pz) + 0.970000029f) * 33.3333359f;
	fy = fminf(fmaxf(fy, -1.0f), 1.0f); fz = fminf(fmaxf(fz, -1.0f), 1.0f);
	float ox[4], oy[4], oz[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const float gx = -pl[i] + pr[i], gy = -pu[i] + pd[i], gz = -pf[i] + pb[i];
		const float wz_ = fmaf(-gz, k, vz[i]);                              // CSProject3D.hlsl:62 (the z component first, as k_project does)
		const float wx_ = fmaf(-gx, k, vx[i]), wy_ = fmaf(-gy, k, vy[i]);
		float px = RCP ? ((float)(4 * x4 + i) + 0.5f) * rX : ((float)(4 * x4 + i) + 0.5f) / (float)g.X;
		px = fmaf(px, 2.0f, -1.0f);
		float fxw = (-fabsf(px) + 0.970000029f) * 33.3333359f;
		fxw = fminf(fmaxf(fxw, -1.0f), 1.0f);
		ox[i] = wx_ * ((0.0f < wx_ * px) ? fxw : 1.0f);                    // :106-108
		oy[i] = wy_ * ((0.0f < wy_ * py) ? fy : 1.0f);
		oz[i] = wz_ * ((0.0f < wz_ * pz) ? fz : 1.0f);
	}
	St::st4(reinterpret_cast<S4*>(vel_out + off), 0, make_float4(ox[0], ox[1], ox[2], ox[3]));
	St::st4(reinterpret_cast<S4*>(vel_out + stride + off), 0, make_float4(oy[0], oy[1], oy[2], oy[3]));
	St::st4(reinterpret_cast<S4*>(vel_out + 2u * stride + off), 0, make_float4(oz[0], oz[1], oz[2], oz[3]));
	if (rec) {
#pragma unroll
		for (int i = 0; i < 4; ++i) oz[i] = St::stored(oz[i]);           // the next advection reads what was STORED (fp16 storage: the binary16 value)
		const float pzn = ((float)z + 0.5f) / (float)g.Zg;              // k_advect's pz (== * rZ for the power-of-two grids of the fast paths)
		int lo = 0, hi = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const float az = fmaf(-oz[i], dt, pzn);
			const int iz = (int)floorf(az * (float)g.Zg - 0.5f);
			const int a = addr_tap(iz, g.Zg, address), b_ = addr_tap(iz + 1, g.Zg, address);
			lo = max(lo, g.z0 - min(a, b_));
			hi = max(hi, max(a, b_) - (g.z0 + g.nz - 1));
		}
		if (__builtin_amdgcn_ballot_w64(lo > 0 || hi > 0) != 0) {       // rare: only the planes next to a face, and not even those when the flow leaves them alone
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) { lo = max(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
			if (((threadIdx.y * blockDim.x + threadIdx.x) & 63) == 0) { if (lo > 0) atomicMax(rec, lo); if (hi > 0) atomicMax(rec + 1, hi); }
		}
	}
}

// ---------------------------------------------------------------------------------------------
// Rows that are no multiple of four cells (150^3, the reference's GI preset: Bin/FluidGI.bat:1), fp32 fields: the v4 kernels'
// scheme with W = 3 or 2 cells per thread and 4-byte-aligned vector accesses (rows of 600 bytes are not 16-byte aligned).
// Per-cell arithmetic is k_divergence's / k_project's.
// ---------------------------------------------------------------------------------------------
template <int W> struct CellsW { float v[W]; };

template <int W>
__device__ __forceinline__ CellsW<W> ldw(const float* base, uint32_t cell)
{
	typedef float __attribute__((ext_vector_type(W))) vt;
	typedef vt __attribute__((aligned(4))) vu;
	const vu t = *reinterpret_cast<const vu*>(base + cell);
	CellsW<W> c;
#pragma unroll
	for (int i = 0; i < W; ++i) c.v[i] = t[i];
	return c;
}

template <int W>
__device__ __forceinline__ void stw(float* base, uint32_t cell, const CellsW<W>& c)
{
	typedef float __attribute__((ext_vector_type(W))) vt;
	typedef vt __attribute__((aligned(4))) vu;
	vt t;
#pragma unroll
	for (int i = 0; i < W; ++i) t[i] = c.v[i];
	*reinterpret_cast<vu*>(base + cell) = t;
}

// W cells of a velocity component in its storage type (binary16: pairs only -- an even row keeps them 4-byte aligned)
template <int W, bool HALF>
__device__ __forceinline__ CellsW<W> ldwv(const typename Store<HALF>::S* base, uint32_t cell)
{
	if constexpr (HALF) {
		static_assert(W == 2, "binary16 rows are read in pairs");
		CellsW<W> c;
		Store<true>::ld2(base, cell, c.v[0], c.v[1]);
		return c;
	} else return ldw<W>(base, cell);
}
template <int W, bool HALF>
__device__ __forceinline__ void stwv(typename Store<HALF>::S* base, uint32_t cell, const CellsW<W>& c)
{
	if constexpr (HALF) {
		F16Pair h;
		h.a = (h16)Store<true>::rounded_f32(c.v[0]); h.b = (h16)Store<true>::rounded_f32(c.v[1]);   // RNE in a separate step, as Store<true>::st
		*reinterpret_cast<F16Pair*>(base + cell) = h;
	} else stw<W>(base, cell, c);
}

template <int W, bool HALF>
__global__ __launch_bounds__(256) void k_divergence_vw(const Geom g, const typename Store<HALF>::S* __restrict__ vel, float* __restrict__ b,
	int z_begin, int nzp, int remap, int rows_per_block)
{
	typedef Store<HALF> St;
	const int XW = g.X / W;
	const Tile3 tile = xcd_tile((XW + (int)blockDim.x - 1) / (int)blockDim.x, (g.Y + rows_per_block - 1) / rows_per_block, nzp, remap);
	const int xw = tile.x * blockDim.x + threadIdx.x;
	const int y = tile.y * rows_per_block + threadIdx.y;
	const int z = z_begin + tile.z;
	if (xw >= XW || y >= g.Y) return;
	const uint32_t plane = (uint32_t)g.plane(), stride = (uint32_t)g.cells_local();
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
	const uint32_t zrow = (uint32_t)g.lz(z) * plane, col = (uint32_t)(W * xw), off = zrow + (uint32_t)y * g.X + col;
	const CellsW<W> cx = ldwv<W, HALF>(vel, off);
	const float L = xw > 0 ? St::ld(vel, off - 1) : cx.v[0];
	const float R = xw < XW - 1 ? St::ld(vel, off + W) : cx.v[W - 1];
	const CellsW<W> U = ldwv<W, HALF>(vel, stride + zrow + (uint32_t)yu * g.X + col), D = ldwv<W, HALF>(vel, stride + zrow + (uint32_t)yd * g.X + col);
	const CellsW<W> F = ldwv<W, HALF>(vel, 2u * stride + (uint32_t)g.lz(zf) * plane + (uint32_t)y * g.X + col);
	const CellsW<W> B = ldwv<W, HALF>(vel, 2u * stride + (uint32_t)g.lz(zb) * plane + (uint32_t)y * g.X + col);
	CellsW<W> o;
#pragma unroll
	for (int i = 0; i < W; ++i) {
		const float l = i == 0 ? L : cx.v[i > 0 ? i - 1 : 0], r = i == W - 1 ? R : cx.v[i < W - 1 ? i + 1 : 0];
		o.v[i] = 0.5f * ((-F.v[i] + B.v[i]) + ((-U.v[i] + D.v[i]) + (-l + r)));
	}
	stw<W>(b, off, o);
}

template <int W, bool HALF>
__global__ __launch_bounds__(256) void k_project_vw(const Geom g, const typename Store<HALF>::S* __restrict__ vel_in, const float* __restrict__ p,
	typename Store<HALF>::S* __restrict__ vel_out, int z_begin, int nzp, int remap, int rows_per_block)
{
	const int XW = g.X / W;
	const Tile3 tile = xcd_tile((XW + (int)blockDim.x - 1) / (int)blockDim.x, (g.Y + rows_per_block - 1) / rows_per_block, nzp, remap);
	const int xw = tile.x * blockDim.x + threadIdx.x;
	const int y = tile.y * rows_per_block + threadIdx.y;
	const int z = z_begin + tile.z;
	if (xw >= XW || y >= g.Y) return;
	const uint32_t plane = (uint32_t)g.plane(), stride = (uint32_t)g.cells_local();
	const int yu = max(y, 1) - 1, yd = min(y + 1, g.Y - 1);
	const int zf = max(z, 1) - 1, zb = min(z + 1, g.Zg - 1);
	const uint32_t zrow = (uint32_t)g.lz(z) * plane, col = (uint32_t)(W * xw), off = zrow + (uint32_t)y * g.X + col;
	const CellsW<W> c = ldw<W>(p, off);
	const float L = xw > 0 ? p[off - 1] : c.v[0];
	const float R = xw < XW - 1 ? p[off + W] : c.v[W - 1];
	const CellsW<W> U = ldw<W>(p, zrow + (uint32_t)yu * g.X + col), D = ldw<W>(p, zrow + (uint32_t)yd * g.X + col);
	const CellsW<W> F = ldw<W>(p, (uint32_t)g.lz(zf) * plane + (uint32_t)y * g.X + col);
	const CellsW<W> B = ldw<W>(p, (uint32_t)g.lz(zb) * plane + (uint32_t)y * g.X + col);
	const CellsW<W> ux = ldwv<W, HALF>(vel_in, off), uy = ldwv<W, HALF>(vel_in, stride + off), uz = ldwv<W, HALF>(vel_in, 2u * stride + off);
	const float k = __uint_as_float(0x3f855556u);                          // 0.5f / 0.48f (g_density, CSProject3D.hlsl:26)
	float py = ((float)y + 0.5f) / (float)g.Y, pz = ((float)z + 0.5f) / (float)g.Zg;
	py = fmaf(py, 2.0f, -1.0f); pz = fmaf(pz, 2.0f, -1.0f);
	float fy = (-fabsf(py) + 0.970000029f) * 33.3333359f, fz = (-fabsf(pz) + 0.970000029f) * 33.3333359f;
	fy = fminf(fmaxf(fy, -1.0f), 1.0f); fz = fminf(fmaxf(fz, -1.0f), 1.0f);
	CellsW<W> ox, oy, oz;
#pragma unroll
	for (int i = 0; i < W; ++i) {
		const float pl = i == 0 ? L : c.v[i > 0 ? i - 1 : 0], pr = i == W - 1 ? R : c.v[i < W - 1 ? i + 1 : 0];
		const float gx = -pl + pr, gy = -U.v[i] + D.v[i], gz = -F.v[i] + B.v[i];
		const float wz_ = fmaf(-gz, k, uz.v[i]);                            // CSProject3D.hlsl:62 (the z component first, as k_project does)
		const float wx_ = fmaf(-gx, k, ux.v[i]), wy_ = fmaf(-gy, k, uy.v[i]);
		float px = ((float)(W * xw + i) + 0.5f) / (float)g.X;
		px = fmaf(px, 2.0f, -1.0f);
		float fxw = (-fabsf(px) + 0.970000029f) * 33.3333359f;
		fxw = fminf(fmaxf(fxw, -1.0f), 1.0f);
		ox.v[i] = wx_ * ((0.0f < wx_ * px) ? fxw : 1.0f);                  // :106-108
		oy.v[i] = wy_ * ((0.0f < wy_ * py) ? fy : 1.0f);
		oz.v[i] = wz_ * ((0.0f < wz_ * pz) ? fz : 1.0f);
	}
	stwv<W, HALF>(vel_out, off, ox);
	stwv<W, HALF>(vel_out, stride + off, oy);
	stwv<W, HALF>(vel_out, 2u * stride + off, oz);
}

// cells per thread of the vW kernels: 3 or 2 where that divides the row, 0 = none (the scalar kernels)
static int vw_width(const Geom& g, int half_store)
{
	const int on = FX_KNOB_INT("ROW_VW", 1);
	if (!on || g.Zg <= 1 || g.cells_local() * 3 >= ((size_t)1 << 30)) return 0;
	if (half_store) return g.X % 2 == 0 ? 2 : 0;               // binary16: pairs (4-byte aligned in an even row)
	return g.X % 3 == 0 ? 3 : (g.X % 2 == 0 ? 2 : 0);
}

// ---------------------------------------------------------------------------------------------
// What the NEXT advection will need from the z-neighbours, measured on the velocity the projection has just written
// (multi-GPU slabs; no reference counterpart).  For every owned voxel the z taps of its back-trace are computed with the
// arithmetic of k_advect (pz, az = fma(-uz, dt, pz), tz = az * Zg - 0.5, floor; CLAMP / MIRROR); need[0] = planes wanted below
// the slab's first plane, need[1] = above its last.  Exact, not a bound: the exchange then carries exactly the planes the
// kernel will touch (for any dt' <= dt the need can only shrink).  One read of uz, one atomicMax per wave that needs anything.
// ---------------------------------------------------------------------------------------------
template <bool HALF>
__global__ __launch_bounds__(256) void k_face_need(const Geom g, const typename Store<HALF>::S* __restrict__ uz, float dt, int address, int* __restrict__ rec,
	int digest, const unsigned* __restrict__ halo_overflow)
{
	typedef Store<HALF> St;
	const size_t plane = g.plane();
	const int z = g.z0 + (int)blockIdx.y;                            // one owned plane per blockIdx.y
	const float pz = ((float)z + 0.5f) / (float)g.Zg;
	const typename St::S* row = uz + (size_t)g.lz(z) * plane;
	int lo = 0, hi = 0;
	auto one = [&](float u) {
		const float az = fmaf(-u, dt, pz);
		const int iz = (int)floorf(az * (float)g.Zg - 0.5f);
		const int a = addr_tap(iz, g.Zg, address), b_ = addr_tap(iz + 1, g.Zg, address);
		lo = max(lo, g.z0 - min(a, b_));
		hi = max(hi, max(a, b_) - (g.z0 + g.nz - 1));
	};
	if (!HALF && (plane & 3) == 0) {                                 // 16-byte loads
		const float4* row4 = reinterpret_cast<const float4*>(row);
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane / 4; i += (size_t)gridDim.x * blockDim.x) {
			const float4 v = row4[i];
			one(v.x); one(v.y); one(v.z); one(v.w);
		}
	} else {
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (size_t)gridDim.x * blockDim.x) one(St::ld(row, i));
	}
	// the record is closed by the first thread of the launch: digest of the schedule options, this step's halo-overflow flag
	// (final: the advection finished long before the projection whose output this kernel reads)
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { rec[2] = digest; rec[3] = (int)*halo_overflow; }
	if (__builtin_amdgcn_ballot_w64(lo > 0 || hi > 0) == 0) return;   // the usual case: nothing leaves the slab
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { lo = max(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
	if ((threadIdx.x & 63) == 0) { if (lo > 0) atomicMax(rec, lo); if (hi > 0) atomicMax(rec + 1, hi); }
}

template <bool HALF>
__global__ __launch_bounds__(256) void k_copy_owned(const Geom g, const typename Store<HALF>::S* __restrict__ src,
	typename Store<HALF>::S* __restrict__ dst)
{
	// copies the owned planes of the three component planes (dt <= 0 path, CSProject3D.hlsl:88,112)
	const size_t n = g.cells_owned();
	const size_t off = (size_t)g.H * g.plane();
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * n; i += (size_t)gridDim.x * blockDim.x) {
		const size_t a = i / n, r = i - a * n;
		dst[a * g.cells_local() + off + r] = src[a * g.cells_local() + off + r];
	}
}

template <bool HALF>
__global__ __launch_bounds__(256) void k_to_storage(const float* __restrict__ src, typename Store<HALF>::S* __restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		Store<HALF>::st(dst, i, src[i]);
}
template <bool HALF>
__global__ __launch_bounds__(256) void k_from_storage(const typename Store<HALF>::S* __restrict__ src, float* __restrict__ dst, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
		dst[i] = Store<HALF>::ld(src, i);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline dim3 grid_xyz(const Geom& g, int nzp) { return dim3(((g.X + 63) / 64) * ((g.Y + 3) / 4) * nzp, 1, 1); }
static inline unsigned grid_1d(size_t n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }
enum { REMAP_JACOBI = 0, REMAP_ADVECT = 1, REMAP_DIV = 2, REMAP_PROJECT = 3 };
// per-kernel mapping mode (see xcd_tile).  FLUIDX_XCD_REMAP="j,a,d,p" overrides (measurement knob).
static int xcd_remap_on(int which)
{
	int mode[4] = { 1, 0, 0, 0 };                              // measured: only the Jacobi sweeps gain (docs/LAB.md)
	const char* e = FX_KNOB("XCD_REMAP");
	int a, b_, c, d;
	if (e && sscanf(e, "%d,%d,%d,%d", &a, &b_, &c, &d) == 4) { mode[0] = a; mode[1] = b_; mode[2] = c; mode[3] = d; }
	return mode[which];
}

// Small grids (planes up to 181^2 cells, fields below 6 M cells: they live in L2 / Infinity Cache): divergence, projection and the general advection kernel
// also gain from the contiguous-eighth order -- with tiles dealt round robin over the XCDs every L2 fetches its own copy of the
// rows its neighbours' tiles share (150^3: k_advect 402 MB of fabric traffic for 189 MB of fields).  Measured, ms per launch,
// round robin / contiguous: 150^3 advection 0.061 / 0.055, divergence 0.0141 / 0.0112, projection 0.0222 / 0.0173; 128^3
// divergence 0.0097 / 0.0087, projection 0.0130 / 0.0119; 256^3 LOSES (0.053 / 0.059, 0.079 / 0.085: a plane of all fields
// no longer fits an L2 there).  FLUIDX_XCD_REMAP decides when set.
static int xcd_remap_for(int which, const Geom& g)
{
	const char* fe_ = FX_KNOB("XCD_REMAP");
	const bool forced = fe_ && *fe_;
	if (forced || which == REMAP_JACOBI) return xcd_remap_on(which);
	return g.Zg > 1 && g.plane() <= 32768 && g.cells_local() < (size_t)6 << 20 ? 1 : 0;      // (measured at 128^3 and 150^3 only: planes up to 181^2)
}

hipError_t launch_advect(const Geom& g, const SimParams& sp, int half_store, const void* vel_in, const void* col_in,
	void* vel_out, void* col_out, int z_begin, int z_end, unsigned* halo_overflow, hipStream_t s, uint32_t* far_scratch, size_t far_words, int far_parity, bool* far_used,
	AdvectAlpha* alpha)
{
	if (far_used) *far_used = false;
	if (alpha) alpha->written = false;
	if (z_end <= z_begin) return hipSuccess;
	// workgroup shape: FLUIDX_ADVECT_BLOCK="bx,by,bz" overrides (measurement knob)
	int bx = 64, by = 4, bz = 1;
	if (const char* e = FX_KNOB("ADVECT_BLOCK")) {
		int a, b_, c;
		if (sscanf(e, "%d,%d,%d", &a, &b_, &c) == 3 && a > 0 && b_ > 0 && c > 0 && a * b_ * c <= 256) { bx = a; by = b_; bz = c; }
	}
	const int nzp = z_end - z_begin;
	// taps from an LDS-staged tile (fx_advect_lds.hip) where the geometry allows it and the launch is large enough to pay;
	// FLUIDX_ADVECT_LDS=0 switches it off, =2 takes it for small launches too (A/B and parity tests; read per launch)
	{
		const char* le = FX_KNOB("ADVECT_LDS");
		if (!(le && le[0] == '0') && bx == 64 && by == 4 && bz == 1 && !xcd_remap_on(REMAP_ADVECT)) {
			const hipError_t e = launch_advect_lds(g, sp, half_store, vel_in, col_in, vel_out, col_out, z_begin, z_end, halo_overflow, far_scratch, far_words, far_parity, far_used, s, le && le[0] == '2', alpha);
			if (e != hipErrorNotSupported) return e;
		}
	}
	// (the gather kernels write the render's side volume on the same terms as the staged ones: one launch over a whole unsliced grid)
	float* alpha_out = alpha && alpha->out && g.nzl() == g.Zg && z_begin == 0 && z_end == g.Zg ? alpha->out : nullptr;
	if (alpha_out) alpha->written = true;
	// lean path (see k_advect_fast): power-of-two extents, fields below 4 GiB, default workgroup shape and tile order
	const char* fe = FX_KNOB("ADVECT_FAST");                  // "0" = always the general kernel (A/B tests; read per launch)
	const bool fast_off = fe && fe[0] == '0';
	auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
	if (!fast_off && pow2(g.X) && pow2(g.Y) && pow2(g.Zg) && g.X >= 2 && g.cells_local() * 16 < ((size_t)1 << 32) &&
		bx == 64 && by == 4 && bz == 1 && !xcd_remap_on(REMAP_ADVECT)) {
		const dim3 block(64, 4, 1), grid(((g.X + 63) / 64) * ((g.Y + 3) / 4) * nzp, 1, 1);
		const float rX = 1.0f / (float)g.X, rY = 1.0f / (float)g.Y, rZ = 1.0f / (float)g.Zg, inv_rr = sp.is3d ? 256.0f : 1024.0f;
		auto lg = [](int v) { int k = 0; while ((1 << k) < v) ++k; return k; };
		const int lgX = lg(g.X), lgY = lg(g.Y), lg_gx = lg((g.X + 63) / 64), lg_gy = lg((g.Y + 3) / 4);
		if (half_store)
			hipLaunchKernelGGL(k_advect_fast<true>, grid, block, 0, s, g, sp, (const h16*)vel_in, (const h16x4*)col_in,
				(h16*)vel_out, (h16x4*)col_out, z_begin, nzp, halo_overflow, rX, rY, rZ, inv_rr, lgX, lgY, lg_gx, lg_gy, alpha_out);
		else
			hipLaunchKernelGGL(k_advect_fast<false>, grid, block, 0, s, g, sp, (const float*)vel_in, (const float4*)col_in,
				(float*)vel_out, (float4*)col_out, z_begin, nzp, halo_overflow, rX, rY, rZ, inv_rr, lgX, lgY, lg_gx, lg_gy, alpha_out);
		return hipGetLastError();
	}
	const int cbz = g.Zg > 1 ? bz : 1;
	const dim3 block(bx, by, cbz), grid(((g.X + bx - 1) / bx) * ((g.Y + by - 1) / by) * ((nzp + cbz - 1) / cbz), 1, 1);
	if (half_store)
		hipLaunchKernelGGL(k_advect<true>, grid, block, 0, s, g, sp, (const h16*)vel_in, (const h16x4*)col_in,
			(h16*)vel_out, (h16x4*)col_out, z_begin, nzp, xcd_remap_for(REMAP_ADVECT, g), halo_overflow, alpha_out);
	else
		hipLaunchKernelGGL(k_advect<false>, grid, block, 0, s, g, sp, (const float*)vel_in, (const float4*)col_in,
			(float*)vel_out, (float4*)col_out, z_begin, nzp, xcd_remap_for(REMAP_ADVECT, g), halo_overflow, alpha_out);
	return hipGetLastError();
}

hipError_t launch_divergence(const Geom& g, int half_store, const void* vel, float* b, int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (g.Zg > 1 && (g.X & 3) == 0 && (g.cells_local() & 3) == 0) {
		const int nzp = z_end - z_begin, X4 = g.X >> 2;
		const int bx = X4 < 64 ? X4 : 64;
		int by = 256 / bx; if (by > g.Y) by = g.Y;
		const dim3 block(bx, by, 1), grid(((X4 + bx - 1) / bx) * ((g.Y + by - 1) / by) * nzp, 1, 1);
		if (half_store) hipLaunchKernelGGL(k_divergence_v4<true>, grid, block, 0, s, g, (const h16*)vel, b, z_begin, nzp, xcd_remap_for(REMAP_DIV, g), by);
		else hipLaunchKernelGGL(k_divergence_v4<false>, grid, block, 0, s, g, (const float*)vel, b, z_begin, nzp, xcd_remap_for(REMAP_DIV, g), by);
		return hipGetLastError();
	}
	if (const int w = vw_width(g, half_store)) {
		const int nzp = z_end - z_begin, XW = g.X / w;
		const int bx = XW < 64 ? XW : 64;
		int by = 256 / bx; if (by > g.Y) by = g.Y;
		const dim3 block(bx, by, 1), grid(((XW + bx - 1) / bx) * ((g.Y + by - 1) / by) * nzp, 1, 1);
		if (half_store) hipLaunchKernelGGL((k_divergence_vw<2, true>), grid, block, 0, s, g, (const h16*)vel, b, z_begin, nzp, xcd_remap_for(REMAP_DIV, g), by);
		else if (w == 3) hipLaunchKernelGGL((k_divergence_vw<3, false>), grid, block, 0, s, g, (const float*)vel, b, z_begin, nzp, xcd_remap_for(REMAP_DIV, g), by);
		else hipLaunchKernelGGL((k_divergence_vw<2, false>), grid, block, 0, s, g, (const float*)vel, b, z_begin, nzp, xcd_remap_for(REMAP_DIV, g), by);
		return hipGetLastError();
	}
	const dim3 grid = grid_xyz(g, z_end - z_begin), block(64, 4, 1);
	if (half_store) hipLaunchKernelGGL(k_divergence<true>, grid, block, 0, s, g, (const h16*)vel, b, z_begin, z_end - z_begin, xcd_remap_for(REMAP_DIV, g));
	else hipLaunchKernelGGL(k_divergence<false>, grid, block, 0, s, g, (const float*)vel, b, z_begin, z_end - z_begin, xcd_remap_for(REMAP_DIV, g));
	return hipGetLastError();
}

hipError_t launch_jacobi_sweep(const Geom& g, const float* p_in, const float* b, float* p_out, uint8_t* frozen,
	int z_begin, int z_end, hipStream_t s)
{
	return launch_jacobi_sweep2(g, p_in, b, p_out, frozen, z_begin, z_end, 0, 0, s);
}

// one sweep over two disjoint plane ranges (the two face zones of a slab) in a single launch
hipError_t launch_jacobi_sweep2(const Geom& g, const float* p_in, const float* b, float* p_out, uint8_t* frozen,
	int z_begin, int z_end, int z_begin2, int z_end2, hipStream_t s)
{
	if (z_end < z_begin) z_end = z_begin;
	if (z_end2 < z_begin2) z_end2 = z_begin2;
	const int split = z_end - z_begin;
	const int nzp = split + (z_end2 - z_begin2);
	if (nzp <= 0) return hipSuccess;
	if (!frozen && g.Zg > 1 && (g.X & 3) == 0) {
		const int X4 = g.X >> 2;
		const int bx = X4 < 64 ? X4 : 64;               // float4 columns per block row
		int by = 256 / bx; if (by < 1) by = 1; if (by > g.Y) by = g.Y;
		const dim3 block(bx, by, 1), grid(((X4 + bx - 1) / bx) * ((g.Y + by - 1) / by) * nzp, 1, 1);
		hipLaunchKernelGGL(k_jacobi_v4, grid, block, 0, s, g, p_in, b, p_out, z_begin, nzp, xcd_remap_on(REMAP_JACOBI), by, split, z_begin2);
	} else {
		hipLaunchKernelGGL(k_jacobi_generic, grid_xyz(g, nzp), dim3(64, 4, 1), 0, s, g, p_in, b, p_out, frozen, z_begin, nzp, xcd_remap_on(REMAP_JACOBI),
			split, z_begin2);
	}
	return hipGetLastError();
}

// ---- temporal blocking: which geometry fuses how many sweeps ----------------------------------------
// FLUIDX_JACOBI_T (1..3) overrides the sweeps fused per launch (measurement knob, DESIGN.md / profiles/).
// (The LDS tile kernel k_jacobi_tb<T> of round 1 -- z-streaming register windows + one LDS plane per level, two barriers per plane --
// lost to the register strips in every shape measured, profiles/archive/r01_jacobi_tile_sweep.txt, and was removed in round 3; with it went
// four sweeps per launch: jacobi_fuse = 4 now runs threes.)

static bool tb_supported(const Geom& g)
{
	const int LX = g.X >> 2;
	return g.Zg > 1 && (g.X & 3) == 0 && (LX == 16 || LX == 32 || LX == 64) && g.Y >= 16;
}

// where two sweeps per launch (register strips) beat one: measured on MI355X with the DPP lane shifts in place
// (us per sweep, one / two sweeps per launch): 256^3 30 / 17.8, 512x512x64 41 / 18.9, 512x512x32 21.8 / 12.0, 512x512x16
// 12.5 / 9.9, 256x256x64 10.0 / 9.0 -- but 256x256x32 6.2 / 8.2, 128^3 5.7 / 7.6, 128x128x32 3.3 / 7.8, 64^3 2.8 / 7.3: below
// ~4 M cells a launch is too short for 8-plane z chunks to fill the chip
static bool strip_profitable(const Geom& g, int nzp)
{
	if (!jacobi_strip_supported(g)) return false;
	return (size_t)g.X * g.Y * (size_t)nzp >= ((size_t)7 << 19);           // 3.5 M cells
}

int jacobi_fused_max_sweeps(const Geom& g, int requested, int nzp)
{
	const int forced = FX_KNOB_INT("JACOBI_T", 0);
	if (jacobi_strip_supported(g) && jacobi_strip_wide(g)) {            // X = 512: one fused shape (two sweeps, wide strips)
		const int want = requested > 0 ? requested : (forced > 0 ? forced : (strip_profitable(g, nzp) ? 2 : 1));
		if (want >= 4 && jacobi_strip4_supported(g) && nzp >= 2) return 4;     // four sweeps: the half-row octet (k_jacobi_strip4x, fx_jacobi_strip4.hip)
		return want >= 3 && jacobi_strip3_supported(g) ? 3 : (want >= 2 ? 2 : 1);
	}
	// any other row of whole quads longer than 256 cells: FOUR sweeps on x tiles of the octet (k_jacobi_strip4t, fx_jacobi_strip4.hip) where
	// asked for or preferred (jacobi_prefers_four); remainders through the paths below
	if (!tb_supported(g) && (requested >= 4 || (!requested && forced >= 4)) && jacobi_strip4_supported(g) && nzp >= 2) return 4;
	// rows that fit no strip / tile kernel (X no multiple of 4, or not 64 / 128 / 256 wide): the general block-per-wave kernel, two
	// sweeps per launch (fx_jacobi_block.hip; 150^3, the reference's GI preset: 19.3 us per single-sweep launch before)
	if (!tb_supported(g)) return jacobi_blockg_supported(g) && (!requested || requested == 2) && !forced && nzp >= 2 ? 2 : 1;   // (requested == 2: the slab rounds)
	// X = 128: a 4 x 4-row block per wave, two sweeps (fx_jacobi_block.hip) -- the strips have too few waves there.  (Round 6 built FOUR
	// sweeps per launch on 8 x 8-row tiles, a workgroup each with the 16 x 16-row cone in its waves' registers and the planes handed over
	// through the LDS: bit-exact, 24.9 us per launch against 2 x 6.8 -- eight barrier phases on one workgroup per CU; docs/LAB.md section 12.)
	if (jacobi_block2_supported(g) && !requested && !forced) return nzp >= 2 ? 2 : 1;
	// default: two sweeps per launch in the register-strip kernel (fx_jacobi_strip.hip) where the geometry allows it,
	// else one sweep per launch
	const int t = requested > 0 ? requested : (forced > 0 ? forced : (strip_profitable(g, nzp) ? 2 : 1));
	if (t >= 4 && jacobi_strip4_supported(g) && nzp >= 2) return 4;     // four sweeps: the quad kernel (fx_jacobi_strip4.hip)
	return t < 1 ? 1 : (t > 3 ? 3 : t);
}

// FOUR sweeps per launch (fx_jacobi_strip4.hip) where the kernels exist: 40 sweeps = 10 launches.  With the octet kernel (k_jacobi_strip4o,
// the default) from 24 planes of 256 x 256 -- round 5, us per sweep at 256 x 256 x D in ones / twos / threes / fours: D = 24 5.7 / 7.4 / 6.8 / 5.3,
// 32 6.2 / 7.7 / 6.9 / 5.2, 64 9.6 / 8.1 / 7.3 / 5.6, 96 - / - / 7.7 / 6.1, 128 8.8 (threes) / 7.1, 192 11.3 / 9.6, 256 14.0 / 12.1, 400 23.3 / 19.0;
// with the quad kernel (STRIP4_OCTET=0) from 144 planes.  JACOBI_PREFER4=0 keeps the threes; an explicit jacobi_fuse / JACOBI_T request is
// always honoured as given.
static size_t jacobi_tiled_four_from() { return (size_t)FX_KNOB_INT("STRIP4T_FROM", 1 << 20); }
bool jacobi_prefers_four(const Geom& g, int requested, int nzp)
{
	const int forced = FX_KNOB_INT("JACOBI_T", 0);
	const int prefer = FX_KNOB_INT("JACOBI_PREFER4", 1);
	const bool octet = FX_KNOB_INT("STRIP4_OCTET", 1) != 0;
	if (g.X == 512)                                                     // k_jacobi_strip4x from 96 planes; below, three x tiles of the octet (k_jacobi_strip4t) from SIX planes --
		// us per sweep at 512 x 512 x D, the round-5 schedule (ones below 16 planes, twos) / fours: 4 6.9 / 9.6, 6 7.0 / 4.9, 8 7.7 / 4.9, 12 10.1 / 5.5,
		// 16 10.2 / 6.3, 32 12.2 / 9.3, 48 15.8 / 12.3 (rows the octet's bands cannot be placed on, Y = 15, 16: from 64 planes, k_jacobi_strip4x)
		return prefer && !requested && !forced && jacobi_strip4_supported(g) && (size_t)g.X * g.Y * (size_t)nzp >= (g.Y >= 17 ? (size_t)3 << 19 : (size_t)1 << 24);
	if (g.X != 256)                                                     // k_jacobi_strip4t (x tiles of the octet): see the table at jacobi_tiled_four_from
		// (rows below 256 cells: a tile with its upper lanes switched off -- from 160 cells a row and 3.1 M cells; us per sweep, the block kernel's twos /
		// fours: 132^3 4.9 / 5.5, 160^3 7.7 / 6.2, 192^3 9.8 / 7.7, 224^3 15.2 / 9.6, 252^3 20.3 / 11.8; 192 x 192 x 48 4.0 / 5.5,
		// x 80 5.2 / 5.3, x 100 6.5 / 5.8; 224 x 224 x 48 4.1 / 5.6, x 64 5.8 / 5.2; 240 x 240 x 48 4.6 / 5.0, x 64 6.2 / 5.4; 160 x 160 x 100 5.1 / 5.3, x 128 6.1 / 5.9)
		return prefer && !requested && !forced && jacobi_strip4_supported(g) &&
			(g.X > 256 ? (size_t)g.X * g.Y * (size_t)nzp >= jacobi_tiled_four_from()
			           : g.X >= FX_KNOB_INT("STRIP4T_NARROW", 160) && (size_t)g.X * g.Y * (size_t)nzp >= (size_t)FX_KNOB_INT("STRIP4T_NARROW_FROM", 3 << 20));
	return prefer && !requested && !forced && jacobi_strip4_supported(g) && (size_t)g.X * g.Y * (size_t)nzp >= (octet ? (size_t)3 << 17 : (size_t)9 << 20);
	// (the octet from SIX planes of 256 x 256 since its z chunks may be four planes short -- round 6, us per sweep in ones / fours: D = 6 3.88 / 3.33,
	// 8 4.30 / 3.48, 12 4.79 / 3.90, 16 5.33 / 3.96, 20 5.58 / 3.87; with chunks of eight or more the fours started at 24 planes)
}

// Default schedule of the serial rounds (single domain, and slab ranks thick enough): THREE sweeps per launch (k_jacobi_strip3) where that kernel exists and the grid is large
// enough, the remainder as two-sweep launches (40 = 12 x 3 + 2 x 2).  Measured 256^3: Jacobi stage of the bench 0.664 ms
// against 0.714 ms in twos (15.0 against 14.3 G voxel-updates/s).  FLUIDX_JACOBI_PREFER3=0 keeps two sweeps per launch throughout; an explicit jacobi_fuse / FLUIDX_JACOBI_T
// request is always honoured as given.
bool jacobi_prefers_three(const Geom& g, int requested, int nzp)
{
	const int forced = FX_KNOB_INT("JACOBI_T", 0);
	const int prefer = FX_KNOB_INT("JACOBI_PREFER3", 1);
	const int no_lds3 = FX_KNOB_INT("STRIP3_OFF", 0);
	return prefer && !requested && !forced && !no_lds3 && jacobi_strip3_supported(g) &&
		(size_t)g.X * g.Y * (size_t)nzp >= (g.X == 512 ? (size_t)1 << 24 : (size_t)7 << 19);
	// X = 256 (k_jacobi_strip3c): wherever the strips pay at all -- round 5, us per sweep in twos / threes / fours: 256 x 256 x 64 8.1 / 7.2 / 7.5,
	// x 96 9.0 / 7.8 / 7.9, x 128 10.8 / 8.9 / 8.9, x 192 14.2 / 11.3 / 11.1 (the 12.6 M-cell threshold dated from the kernel before the
	// cooperative pairs).  X = 512 (k_jacobi_strip3h): from 16.8 M cells since the round-2 hand-over order -- 512x512x64 (a rank of
	// BASELINE configs[3]) 18.7 against 19.5 us per sweep, 512x512x128 36.2 against 43.7, 512^3 117.6 against 152 (before: 20.3 / 39.6 / 129)
}

hipError_t launch_jacobi_fused(const Geom& g, const float* p_in, const float* b, float* p_out, int sweeps,
	int z_begin, int z_end, hipStream_t s)
{
	if (z_end <= z_begin) return hipSuccess;
	if (jacobi_strip_supported(g) && jacobi_strip_wide(g)) {
		if (sweeps == 4) return launch_jacobi_strip4(g, p_in, b, p_out, z_begin, z_end, s);
		if (sweeps == 3 && jacobi_strip3_supported(g)) return launch_jacobi_strip3(g, p_in, b, p_out, z_begin, z_end, s);
		return sweeps == 2 ? launch_jacobi_strip(g, p_in, b, p_out, 2, z_begin, z_end, s) : hipErrorNotSupported;
	}
	if (!tb_supported(g)) {
		if (sweeps == 4) return launch_jacobi_strip4(g, p_in, b, p_out, z_begin, z_end, s);       // k_jacobi_strip4t
		return sweeps == 2 && jacobi_blockg_supported(g) ? launch_jacobi_blockg(g, p_in, b, p_out, z_begin, z_end, s) : hipErrorNotSupported;
	}
	switch (sweeps) {
	case 2:
		if (jacobi_block2_supported(g)) return launch_jacobi_block2(g, p_in, b, p_out, z_begin, z_end, s);
		return launch_jacobi_strip(g, p_in, b, p_out, 2, z_begin, z_end, s);
	case 3: {
		const int no_lds3 = FX_KNOB_INT("STRIP3_OFF", 0);    // 1 = the all-register three-sweep strips
		if (!no_lds3 && jacobi_strip3_supported(g)) return launch_jacobi_strip3(g, p_in, b, p_out, z_begin, z_end, s);
		return launch_jacobi_strip(g, p_in, b, p_out, 3, z_begin, z_end, s);
	}
	case 4: return launch_jacobi_strip4(g, p_in, b, p_out, z_begin, z_end, s);     // (hipErrorNotSupported where the quad kernel does not exist)
	default: return hipErrorNotSupported;            // (jacobi_fused_max_sweeps never offers more than four)
	}
}

// device-to-device copy of whole planes as a kernel (16-byte words): the loop-back transport's stand-in for a link.  A
// hipMemcpyAsync on a side stream took the SDMA path here (~50 GB/s inside one device) and made the overlapped schedules
// look 2-3x slower than the serial one on a 1-GPU box.
__global__ __launch_bounds__(256) void k_copy16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// fx_field_digest: out[0] += sum mix(bits, key0 + i), out[1] += sum mix'(...) over `count` elements of ES bytes each (splitmix64's
// finaliser; the sums wrap: order-free, so atomics and any decomposition of the range give the same two words)
__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}
template <typename T>
__global__ __launch_bounds__(256) void k_digest(const T* __restrict__ v, size_t count, unsigned long long key0, unsigned long long* __restrict__ out)
{
	unsigned long long a = 0, c = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
		const unsigned long long h = mix64(((key0 + i) << 32) ^ ((key0 + i) >> 32) ^ ((unsigned long long)v[i] * 0xD6E8FEB86659FD93ull));
		a += h;
		c += mix64(h ^ 0xA5A5A5A55A5A5A5Aull);
	}
	for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); c += __shfl_down(c, o); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(out, a); atomicAdd(out + 1, c); }
}

hipError_t launch_digest(const void* v, size_t count, int elem_bytes, unsigned long long key0, unsigned long long* out, hipStream_t s)
{
	if (!count) return hipSuccess;
	const unsigned grid = (unsigned)std::min<size_t>((count + 255) / 256, 4096);
	if (elem_bytes == 4) hipLaunchKernelGGL(k_digest<uint32_t>, dim3(grid), dim3(256), 0, s, static_cast<const uint32_t*>(v), count, key0, out);
	else hipLaunchKernelGGL(k_digest<uint16_t>, dim3(grid), dim3(256), 0, s, static_cast<const uint16_t*>(v), count, key0, out);
	return hipGetLastError();
}

hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s)
{
	if (!bytes) return hipSuccess;
	if ((bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
	const size_t n = bytes / 16;
	const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
	hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s, static_cast<uint4*>(dst), static_cast<const uint4*>(src), n);
	return hipGetLastError();
}

hipError_t launch_project(const Geom& g, const SimParams& sp, int half_store, const void* vel_in, const float* p,
	void* vel_out, int z_begin, int z_end, hipStream_t s, int* rec, int digest, const unsigned* halo_overflow, bool* rec_done)
{
	if (rec_done) *rec_done = false;
	if (z_end <= z_begin) return hipSuccess;
	const int v4_on = FX_KNOB_INT("PROJECT_V4", 1);
	if (v4_on && sp.is3d && (g.X & 3) == 0 && (g.cells_local() & 3) == 0 && g.cells_local() * 3 < ((size_t)1 << 30)) {
		auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
		const int nzp = z_end - z_begin, X4 = g.X >> 2;
		const int bx = X4 < 64 ? X4 : 64;
		int by = 256 / bx; if (by > g.Y) by = g.Y;
		const dim3 block(bx, by, 1), grid(((X4 + bx - 1) / bx) * ((g.Y + by - 1) / by) * nzp, 1, 1);
		const float rX = 1.0f / (float)g.X, rY = 1.0f / (float)g.Y, rZ = 1.0f / (float)g.Zg;
		// the step record rides along when this launch covers exactly the owned planes (what k_face_need would scan)
		int* r = (rec && z_begin == g.z0 && z_end == g.z0 + g.nz) ? rec : nullptr;
		if (r) {
			const hipError_t e = hipMemsetAsync(r, 0, 2 * sizeof(int), s);
			if (e != hipSuccess) return e;
			if (rec_done) *rec_done = true;
		}
		const bool rcp = pow2(g.X) && pow2(g.Y) && pow2(g.Zg);
#define FX_PV4(RCP_, H_, T_) hipLaunchKernelGGL((k_project_v4<RCP_, H_>), grid, block, 0, s, g, (const T_*)vel_in, p, (T_*)vel_out, z_begin, nzp, \
			xcd_remap_for(REMAP_PROJECT, g), by, rX, rY, rZ, r, sp.dt, sp.address, digest, halo_overflow)
		if (half_store) { if (rcp) FX_PV4(true, true, h16); else FX_PV4(false, true, h16); }
		else { if (rcp) FX_PV4(true, false, float); else FX_PV4(false, false, float); }
#undef FX_PV4
		return hipGetLastError();
	}
	if (const int w = sp.is3d ? vw_width(g, half_store) : 0) {
		const int nzp = z_end - z_begin, XW = g.X / w;
		const int bx = XW < 64 ? XW : 64;
		int by = 256 / bx; if (by > g.Y) by = g.Y;
		const dim3 block(bx, by, 1), grid(((XW + bx - 1) / bx) * ((g.Y + by - 1) / by) * nzp, 1, 1);
		if (half_store) hipLaunchKernelGGL((k_project_vw<2, true>), grid, block, 0, s, g, (const h16*)vel_in, p, (h16*)vel_out, z_begin, nzp, xcd_remap_for(REMAP_PROJECT, g), by);
		else if (w == 3) hipLaunchKernelGGL((k_project_vw<3, false>), grid, block, 0, s, g, (const float*)vel_in, p, (float*)vel_out, z_begin, nzp, xcd_remap_for(REMAP_PROJECT, g), by);
		else hipLaunchKernelGGL((k_project_vw<2, false>), grid, block, 0, s, g, (const float*)vel_in, p, (float*)vel_out, z_begin, nzp, xcd_remap_for(REMAP_PROJECT, g), by);
		return hipGetLastError();
	}
	const dim3 grid = grid_xyz(g, z_end - z_begin), block(64, 4, 1);
	if (half_store) hipLaunchKernelGGL(k_project<true>, grid, block, 0, s, g, sp, (const h16*)vel_in, p, (h16*)vel_out, z_begin, z_end - z_begin, xcd_remap_for(REMAP_PROJECT, g));
	else hipLaunchKernelGGL(k_project<false>, grid, block, 0, s, g, sp, (const float*)vel_in, p, (float*)vel_out, z_begin, z_end - z_begin, xcd_remap_for(REMAP_PROJECT, g));
	return hipGetLastError();
}

hipError_t launch_face_need(const Geom& g, int half_store, const void* vel, float dt, int address, int digest, const unsigned* halo_overflow, int* rec, hipStream_t s)
{
	hipError_t e = hipMemsetAsync(rec, 0, 2 * sizeof(int), s);
	if (e != hipSuccess) return e;
	const size_t es = half_store ? 2 : 4;
	const char* uz = static_cast<const char*>(vel) + 2 * g.cells_local() * es;
	const size_t per_thread = (!half_store && (g.plane() & 3) == 0) ? 16 : 4;      // cells a thread takes per plane (4 trips of 16 B, or 4 scalars)
	size_t gx = (g.plane() / per_thread + 255) / 256;
	if (gx < 1) gx = 1;
	if (gx > 64) gx = 64;
	const dim3 grid((unsigned)gx, (unsigned)g.nz, 1);
	if (half_store) hipLaunchKernelGGL(k_face_need<true>, grid, dim3(256), 0, s, g, (const h16*)uz, dt, address, rec, digest, halo_overflow);
	else hipLaunchKernelGGL(k_face_need<false>, grid, dim3(256), 0, s, g, (const float*)uz, dt, address, rec, digest, halo_overflow);
	return hipGetLastError();
}

hipError_t launch_copy_velocity(const Geom& g, int half_store, const void* vel_in, void* vel_out, hipStream_t s)
{
	const unsigned grid = grid_1d(3 * g.cells_owned());
	if (half_store) hipLaunchKernelGGL(k_copy_owned<true>, dim3(grid), dim3(256), 0, s, g, (const h16*)vel_in, (h16*)vel_out);
	else hipLaunchKernelGGL(k_copy_owned<false>, dim3(grid), dim3(256), 0, s, g, (const float*)vel_in, (float*)vel_out);
	return hipGetLastError();
}

hipError_t launch_to_storage(const float* src, void* dst, size_t n, int half_store, hipStream_t s)
{
	if (!n) return hipSuccess;
	if (half_store) hipLaunchKernelGGL(k_to_storage<true>, dim3(grid_1d(n)), dim3(256), 0, s, src, (h16*)dst, n);
	else hipLaunchKernelGGL(k_to_storage<false>, dim3(grid_1d(n)), dim3(256), 0, s, src, (float*)dst, n);
	return hipGetLastError();
}

hipError_t launch_from_storage(const void* src, float* dst, size_t n, int half_store, hipStream_t s)
{
	if (!n) return hipSuccess;
	if (half_store) hipLaunchKernelGGL(k_from_storage<true>, dim3(grid_1d(n)), dim3(256), 0, s, (const h16*)src, dst, n);
	else hipLaunchKernelGGL(k_from_storage<false>, dim3(grid_1d(n)), dim3(256), 0, s, (const float*)src, dst, n);
	return hipGetLastError();
}

}  // namespace fx
