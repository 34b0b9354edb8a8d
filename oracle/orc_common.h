// oracle/orc_common.h -- shared helpers of the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by or
// called from the product path (fluidx12_amd/, include/).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// The oracle is a scalar C++ restatement of the reference's HLSL compute shaders
// (reference = StarsX/FluidX12, mounted read-only at /root/reference).  Operation
// order, folded constants and `mad` placement follow the reference's SHIPPED DXBC
// binaries (Bin/*.cso, decoded with tools/dxbc.py), which is the only executable
// statement of its arithmetic; every function cites the HLSL file:line it restates.
//
// Conventions (stated once, used everywhere; see DESIGN.md "numerics contract"):
//   * all arithmetic is IEEE binary32, no contraction (-ffp-contract=off);
//   * a DXBC `mad` is restated as fmaf() (fused), the lowering current GPUs use;
//   * DXBC `rsq`/`exp`/`div` are restated with correctly rounded 1/sqrtf, exp2f, '/';
//   * texture sampling is D3D trilinear with fp32 weights: t = u*N - 0.5,
//     i0 = floor(t), f = t - i0, taps i0 and i0+1 per axis run through the address
//     mode, blend = lerp_x then lerp_y then lerp_z with lerp(a,b,f) = fmaf(f, b-a, a).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

namespace orc {

enum Address { ADDR_CLAMP = 0, ADDR_MIRROR = 1 };

static inline float bits2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// ---- IEEE binary16 storage emulation (typed UAV store to R16G16B16A16_FLOAT:
//      round-to-nearest-even, subnormals preserved; Fluid.cpp:207,213) ------------
static inline uint16_t f32_to_f16_bits(float f)
{
	const uint32_t x = f2bits(f);
	const uint32_t sign = (x >> 16) & 0x8000u;
	const uint32_t ax = x & 0x7FFFFFFFu;
	if (ax >= 0x7F800000u) return (uint16_t)(sign | (ax > 0x7F800000u ? 0x7E00u : 0x7C00u)); // nan / inf
	if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                 // rounds to inf
	if (ax < 0x33000001u) return (uint16_t)sign;                                              // rounds to 0
	int32_t e = (int32_t)(ax >> 23) - 127;
	uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
	int shift;
	uint32_t base;
	if (e < -14) { shift = 13 + (-14 - e); base = 0; }          // subnormal half
	else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7FFFFFu; }
	uint32_t q = m >> shift;
	const uint32_t rem = m & ((1u << shift) - 1u);
	const uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (q & 1u))) ++q;           // RNE; carry propagates into exponent
	return (uint16_t)(sign | (base + q));
}

static inline float f16_bits_to_f32(uint16_t h)
{
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	const uint32_t e = (h >> 10) & 0x1Fu;
	const uint32_t m = h & 0x3FFu;
	if (e == 0) {
		if (m == 0) return bits2f(sign);
		const float v = std::ldexp((float)m, -24);
		return sign ? -v : v;
	}
	if (e == 31) return bits2f(sign | 0x7F800000u | (m << 13));
	return bits2f(sign | ((e + 112u) << 23) | (m << 13));
}

static inline float quant_half(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }

// ---- R11G11B10_FLOAT emulation (light map, Fluid.cpp:226): unsigned, 5-bit exponent,
//      6/6/5-bit mantissa, negative -> 0, round-to-nearest-even -----------------------
static inline float quant_ufloat(float f, int mbits)
{
	if (!(f > 0.0f)) return 0.0f;                     // negatives and NaN -> 0
	const int drop = 23 - mbits;
	uint32_t x = f2bits(f);
	if (x >= 0x7F800000u) return f;                   // +inf stays
	int e = (int)(x >> 23) - 127;
	if (e < -14) {                                    // subnormal in the small format
		const float scale = std::ldexp(1.0f, 14 + mbits);
		const float q = std::nearbyintf(f * scale);   // default rounding mode = RNE
		return q / scale;
	}
	const uint32_t rem = x & ((1u << drop) - 1u);
	const uint32_t half = 1u << (drop - 1);
	x >>= drop;
	if (rem > half || (rem == half && (x & 1u))) ++x;
	x <<= drop;
	const float r = bits2f(x);
	const float maxv = std::ldexp(2.0f - std::ldexp(1.0f, -mbits), 15);
	return r > maxv ? INFINITY : r;
}

// ---- D3D texture addressing of one integer tap ----------------------------------
static inline int addr_tap(int i, int n, int mode)
{
	if (mode == ADDR_MIRROR) {
		const int period = 2 * n;
		int m = i % period;
		if (m < 0) m += period;
		return m < n ? m : period - 1 - m;
	}
	return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

static inline float lerpf(float a, float b, float f) { return std::fmaf(f, b - a, a); }

struct Taps { int i0[3], i1[3]; float f[3]; };

// uvw in normalised texture space; offset = integer texel offset (SampleLevel's
// `offset` argument, RayMarch.hlsli:92)
static inline Taps make_taps(const float uvw[3], const int dims[3], int mode, const int offset[3] = nullptr)
{
	Taps t;
	for (int a = 0; a < 3; ++a) {
		const float s = uvw[a] * (float)dims[a] - 0.5f;
		const float fl = std::floor(s);
		int i = (int)fl;
		t.f[a] = s - fl;
		if (offset) i += offset[a];
		t.i0[a] = addr_tap(i, dims[a], mode);
		t.i1[a] = addr_tap(i + 1, dims[a], mode);
	}
	return t;
}

// scalar field, layout [z][y][x]
static inline float sample_scalar(const float* f, const int dims[3], const Taps& t)
{
	const size_t X = dims[0], XY = (size_t)dims[0] * dims[1];
	auto at = [&](int x, int y, int z) { return f[(size_t)z * XY + (size_t)y * X + x]; };
	const float c00 = lerpf(at(t.i0[0], t.i0[1], t.i0[2]), at(t.i1[0], t.i0[1], t.i0[2]), t.f[0]);
	const float c10 = lerpf(at(t.i0[0], t.i1[1], t.i0[2]), at(t.i1[0], t.i1[1], t.i0[2]), t.f[0]);
	const float c01 = lerpf(at(t.i0[0], t.i0[1], t.i1[2]), at(t.i1[0], t.i0[1], t.i1[2]), t.f[0]);
	const float c11 = lerpf(at(t.i0[0], t.i1[1], t.i1[2]), at(t.i1[0], t.i1[1], t.i1[2]), t.f[0]);
	return lerpf(lerpf(c00, c10, t.f[1]), lerpf(c01, c11, t.f[1]), t.f[2]);
}

// interleaved field with `nc` channels per texel, layout [z][y][x][nc]; samples channel c
static inline float sample_chan(const float* f, int nc, int c, const int dims[3], const Taps& t)
{
	const size_t X = dims[0], XY = (size_t)dims[0] * dims[1];
	auto at = [&](int x, int y, int z) { return f[((size_t)z * XY + (size_t)y * X + x) * nc + c]; };
	const float c00 = lerpf(at(t.i0[0], t.i0[1], t.i0[2]), at(t.i1[0], t.i0[1], t.i0[2]), t.f[0]);
	const float c10 = lerpf(at(t.i0[0], t.i1[1], t.i0[2]), at(t.i1[0], t.i1[1], t.i0[2]), t.f[0]);
	const float c01 = lerpf(at(t.i0[0], t.i0[1], t.i1[2]), at(t.i1[0], t.i0[1], t.i1[2]), t.f[0]);
	const float c11 = lerpf(at(t.i0[0], t.i1[1], t.i1[2]), at(t.i1[0], t.i1[1], t.i1[2]), t.f[0]);
	return lerpf(lerpf(c00, c10, t.f[1]), lerpf(c01, c11, t.f[1]), t.f[2]);
}

static inline float saturate(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); }  // NaN -> 0 like D3D
static inline float dp3(const float a[3], const float b[3])
{
	return std::fmaf(a[2], b[2], std::fmaf(a[1], b[1], a[0] * b[0]));
}

}  // namespace orc
