// oracle/orc_sim.cpp -- CPU oracle for the simulation hot path (TEST INFRASTRUCTURE ONLY,
// see orc_common.h).  Restates, in scalar C++:
//   CSAdvect::main      /root/reference/FluidX12/Content/Shaders/CSAdvect.hlsl:41-79
//                       (+ Simulation.hlsli:8-19, Impulse.hlsli:14-18)
//   GetDivergence       CSProject3D.hlsl:39-50, CSProject2D.hlsl:37-46
//   Poisson             CSPoisson.hlsli:8-26 (ITER 64: CSProject3D.hlsl:13)
//   Project + wall      CSProject3D.hlsl:55-63,105-112; CSProject2D.hlsl:51-59,99-105
//   Fluid::Simulate     Content/Fluid.cpp:348-410 (host sequencing / ping-pong rule)
// following the operation order of the shipped Bin/CSAdvect.cso, CSProject3D.cso,
// CSProject2D.cso (tools/dxbc.py).  Layouts: velocity float[3][Z][Y][X] (SoA planes),
// colour float[Z][Y][X][4], scalars float[Z][Y][X].
//
// The reference relaxes the pressure in place with no cross-group synchronisation
// (chaotic relaxation, CSPoisson.hlsli:14-22), so its result is schedule dependent.
// The oracle (and the build) define the lock-step schedule: every sweep reads the
// previous sweep's values (synchronous Jacobi).  mode 0 = fixed sweep count, no early
// out; mode 1 = "faithful": 64-sweep cap with the per-cell early-out of :24 modelled
// as a freeze mask (a cell that breaks keeps its last stored value).
#include "orc_common.h"
#include "fx_oracle.h"
#include <vector>

using namespace orc;

namespace {

struct Grid {
	int X, Y, Z;
	size_t n() const { return (size_t)X * Y * Z; }
	size_t idx(int x, int y, int z) const { return ((size_t)z * Y + y) * X + x; }
};

inline float store_q(float v, int half) { return half ? quant_half(v) : v; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// CSAdvect.hlsl:41-79.  vel_in/vel_out: [3][Z][Y][X]; col_in/col_out: [Z][Y][X][4].
// ---------------------------------------------------------------------------------------------
void orc_advect(const float* vel_in, const float* col_in, float* vel_out, float* col_out,
	int X, int Y, int Z, float dt, int address_mode, int half_storage)
{
	const Grid g{ X, Y, Z };
	const size_t N = g.n();
	const int dims[3] = { X, Y, Z };
	const float fdims[3] = { (float)X, (float)Y, (float)Z };
	const bool is3D = 1.0f < fdims[2];                       // lt l(1.0), gridSize.z  (CSAdvect.hlsl:58,65)
	const float rr = is3D ? 0.00390625f : 0.0009765625f;     // r*r folded: (1/16)^2, (1/32)^2 (Impulse.hlsli:15)
	const float atten = std::fmax(std::fmaf(-dt, 0.200000003f, 1.0f), 0.0f);   // CSAdvect.hlsl:74

#pragma omp parallel for schedule(static) collapse(2)
	for (int z = 0; z < Z; ++z)
		for (int y = 0; y < Y; ++y)
			for (int x = 0; x < X; ++x) {
				const size_t id = g.idx(x, y, z);
				const int cell[3] = { x, y, z };
				float pos[3], disp[3];
				for (int a = 0; a < 3; ++a) pos[a] = ((float)cell[a] + 0.5f) / fdims[a];   // Simulation.hlsli:10
				disp[0] = pos[0] + -0.5f;                                                   // Impulse.hlsli:14
				disp[1] = pos[1] + -0.100000001f;
				disp[2] = pos[2] + -0.5f;
				const float d2 = dp3(disp, disp);
				// Gaussian (CSAdvect.hlsl:33-36): exp(-4 d2 / r^2) compiled as exp2((d2*-4)/rr * log2(e))
				const float basis = std::exp2f(((d2 * -4.0f) / rr) * 1.44269502f);
				float F[3];
				if (is3D) {                                       // :63-65, g_extForce*4 folded to 192
					F[0] = std::fmaf(basis, 0.0f, disp[2] * -200.0f);
					F[1] = std::fmaf(basis, 192.0f, 0.0f);
					F[2] = std::fmaf(basis, 0.0f, disp[0] * 200.0f);
				} else {
					F[0] = 0.0f; F[1] = basis * 48.0f; F[2] = 0.0f;
				}
				const float u0[3] = { vel_in[id], vel_in[N + id], vel_in[2 * N + id] };     // :47
				float adv[3];
				for (int a = 0; a < 3; ++a) adv[a] = std::fmaf(-u0[a], dt, pos[a]);         // :52
				const Taps t = make_taps(adv, dims, address_mode);
				float u[3], c[4];
				for (int a = 0; a < 3; ++a) u[a] = sample_scalar(vel_in + a * N, dims, t);  // :53
				for (int a = 0; a < 4; ++a) c[a] = sample_chan(col_in, 4, a, dims, t);      // :54
				if (basis >= 0.0183156393f) {                     // :60  exp(-4.0) folded
					for (int a = 0; a < 3; ++a) u[a] = std::fmaf(F[a], dt, u[a]);           // :66
					const float bdt = basis * dt;
					static const float imp[4] = { 8.0f, 16.0f, 40.0f, 40.0f };              // Impulse.hlsli:16-18
					for (int a = 0; a < 4; ++a) c[a] = saturate(std::fmaf(bdt, imp[a], c[a]));   // :67
				}
				// _PRE_MULTIPLIED_ is defined for CSAdvect (FluidX12.vcxproj:181) -> :70-72 compiled out
				for (int a = 0; a < 3; ++a) vel_out[a * N + id] = store_q(u[a] * atten, half_storage);   // :77
				for (int a = 0; a < 4; ++a) col_out[id * 4 + a] = store_q(c[a] * atten, half_storage);   // :78
			}
}

// ---------------------------------------------------------------------------------------------
// GetDivergence (CSProject3D.hlsl:39-50 / CSProject2D.hlsl:37-46) with the clamped neighbour
// table of CSProject3D.hlsl:75-83.  Association order as compiled:
//   3D: S = (fB - fF) + ((fD - fU) + (fR - fL));  2D: S = (fR - fL) + (fD - fU);  b = 0.5 S.
// ---------------------------------------------------------------------------------------------
void orc_divergence(const float* vel, float* b, int X, int Y, int Z)
{
	const Grid g{ X, Y, Z };
	const size_t N = g.n();
	const float* ux = vel; const float* uy = vel + N; const float* uz = vel + 2 * N;
#pragma omp parallel for schedule(static) collapse(2)
	for (int z = 0; z < Z; ++z)
		for (int y = 0; y < Y; ++y)
			for (int x = 0; x < X; ++x) {
				const int xl = std::max(x, 1) - 1, xr = std::min(x + 1, X - 1);
				const int yu = std::max(y, 1) - 1, yd = std::min(y + 1, Y - 1);
				const float dx = -ux[g.idx(xl, y, z)] + ux[g.idx(xr, y, z)];
				const float dy = -uy[g.idx(x, yu, z)] + uy[g.idx(x, yd, z)];
				float S;
				if (Z > 1) {
					const int zf = std::max(z, 1) - 1, zb = std::min(z + 1, Z - 1);
					const float dz = -uz[g.idx(x, y, zf)] + uz[g.idx(x, y, zb)];
					S = dz + (dy + dx);
				} else {
					S = dx + dy;
				}
				b[g.idx(x, y, z)] = 0.5f * S;
			}
}

// ---------------------------------------------------------------------------------------------
// One synchronous sweep of Poisson() (CSPoisson.hlsli:11-25): p_out = relax(p_in, b).
// As compiled:  x = ((((((qL - b) + qR) + qU) + qD) + qF) + qB) * (1/6 = 0x3e2aaaab)
//               2D: x = ((((qL - b) + qR) + qU) + qD) * 0.25
// frozen (may be null): per-cell early-out mask of :24 -- a frozen cell copies its value;
// a cell freezes after storing x when |fma(sum, 1/N, -x0)| < 0.001.
// Returns the number of cells still active after the sweep.
// ---------------------------------------------------------------------------------------------
long long orc_jacobi_sweep(const float* p_in, const float* b, float* p_out, uint8_t* frozen, int X, int Y, int Z)
{
	const Grid g{ X, Y, Z };
	const float inv = Z > 1 ? bits2f(0x3e2aaaabu) : 0.25f;
	long long active = 0;
#pragma omp parallel for schedule(static) collapse(2) reduction(+ : active)
	for (int z = 0; z < Z; ++z)
		for (int y = 0; y < Y; ++y)
			for (int x = 0; x < X; ++x) {
				const size_t id = g.idx(x, y, z);
				if (frozen && frozen[id]) { p_out[id] = p_in[id]; continue; }
				const int xl = std::max(x, 1) - 1, xr = std::min(x + 1, X - 1);
				const int yu = std::max(y, 1) - 1, yd = std::min(y + 1, Y - 1);
				float s = p_in[g.idx(xl, y, z)] - b[id];
				s = p_in[g.idx(xr, y, z)] + s;
				s = p_in[g.idx(x, yu, z)] + s;
				s = p_in[g.idx(x, yd, z)] + s;
				if (Z > 1) {
					const int zf = std::max(z, 1) - 1, zb = std::min(z + 1, Z - 1);
					s = p_in[g.idx(x, y, zf)] + s;
					s = p_in[g.idx(x, y, zb)] + s;
				}
				const float xnew = s * inv;
				p_out[id] = xnew;
				if (frozen) {
					if (std::fabs(std::fmaf(s, inv, -p_in[id])) < 0.00100000005f) frozen[id] = 1;
					else ++active;
				}
			}
	return frozen ? active : (long long)g.n();
}

// iters sweeps starting from p (in/out); tmp is scratch of the same size.  mode 0: fixed count;
// mode 1: faithful (cap = iters, normally 64, with freeze mask).  Returns sweeps executed.
int orc_jacobi(float* p, const float* b, float* tmp, int X, int Y, int Z, int iters, int mode)
{
	const size_t n = (size_t)X * Y * Z;
	std::vector<uint8_t> frozen;
	if (mode == 1) frozen.assign(n, 0);
	float* src = p; float* dst = tmp;
	int k = 0;
	for (; k < iters; ++k) {
		const long long active = orc_jacobi_sweep(src, b, dst, mode == 1 ? frozen.data() : nullptr, X, Y, Z);
		std::swap(src, dst);
		if (mode == 1 && active == 0) { ++k; break; }
	}
	if (src != p) std::memcpy(p, src, n * sizeof(float));
	return k;
}

// ---------------------------------------------------------------------------------------------
// Project + boundary (CSProject3D.hlsl:55-63,105-112; CSProject2D.hlsl:51-59,99-105).
// As compiled: u = fma(-(q+ - q-), 0.5f/0.48f = 0x3f855556, u)   (2D: factor 0.5, xy only)
//              pos = fma((id+0.5)/dims, 2, -1) (2D: z keeps (id+0.5)/dims)
//              f = min(max((0.97 - |pos|) * 33.3333359, -1), 1);  u *= (0 < u*pos) ? f : 1
// ---------------------------------------------------------------------------------------------
void orc_project(const float* vel_in, const float* p, float* vel_out, int X, int Y, int Z, int half_storage)
{
	const Grid g{ X, Y, Z };
	const size_t N = g.n();
	const float fdims[3] = { (float)X, (float)Y, (float)Z };
	const bool is3D = Z > 1;
	const float k = is3D ? bits2f(0x3f855556u) : 0.5f;
#pragma omp parallel for schedule(static) collapse(2)
	for (int z = 0; z < Z; ++z)
		for (int y = 0; y < Y; ++y)
			for (int x = 0; x < X; ++x) {
				const size_t id = g.idx(x, y, z);
				const int cell[3] = { x, y, z };
				const int xl = std::max(x, 1) - 1, xr = std::min(x + 1, X - 1);
				const int yu = std::max(y, 1) - 1, yd = std::min(y + 1, Y - 1);
				float u[3] = { vel_in[id], vel_in[N + id], vel_in[2 * N + id] };
				float grad[3];
				grad[0] = -p[g.idx(xl, y, z)] + p[g.idx(xr, y, z)];
				grad[1] = -p[g.idx(x, yu, z)] + p[g.idx(x, yd, z)];
				grad[2] = 0.0f;
				if (is3D) {
					const int zf = std::max(z, 1) - 1, zb = std::min(z + 1, Z - 1);
					grad[2] = -p[g.idx(x, y, zf)] + p[g.idx(x, y, zb)];
				}
				const int ncomp = is3D ? 3 : 2;
				for (int a = 0; a < ncomp; ++a) u[a] = std::fmaf(-grad[a], k, u[a]);
				for (int a = 0; a < 3; ++a) {
					float pos = ((float)cell[a] + 0.5f) / fdims[a];
					if (is3D || a < 2) pos = std::fmaf(pos, 2.0f, -1.0f);
					float f = (-std::fabs(pos) + 0.970000029f) * 33.3333359f;
					f = std::fmin(std::fmax(f, -1.0f), 1.0f);
					const float w = (0.0f < u[a] * pos) ? f : 1.0f;
					vel_out[a * N + id] = store_q(u[a] * w, half_storage);
				}
			}
}

// ---------------------------------------------------------------------------------------------
// One simulation step = Fluid::Simulate (Fluid.cpp:348-410) after UpdateFrame flipped the colour
// parity (Fluid.cpp:345).  vel0/vel1: the two velocity textures (advect 0->1, project 1->0);
// col_src/col_dst: colour[!parity] / colour[parity];  p: pressure (persistent, warm start);
// b, tmp: scratch scalars.  dt <= 0: advect still runs (identity sampling), project copies
// (CSProject3D.hlsl:88,112).
// ---------------------------------------------------------------------------------------------
void orc_simulate(float* vel0, float* vel1, const float* col_src, float* col_dst, float* p, float* b, float* tmp,
	int X, int Y, int Z, float dt, int iters, int mode, int address_mode, int half_storage)
{
	orc_advect(vel0, col_src, vel1, col_dst, X, Y, Z, dt, address_mode, half_storage);
	if (dt > 0.0f) {
		orc_divergence(vel1, b, X, Y, Z);
		orc_jacobi(p, b, tmp, X, Y, Z, iters, mode);
		orc_project(vel1, p, vel0, X, Y, Z, half_storage);
	} else {
		std::memcpy(vel0, vel1, 3 * (size_t)X * Y * Z * sizeof(float));
	}
}

void orc_quantize_half(const float* in, float* out, long long n)
{
	for (long long i = 0; i < n; ++i) out[i] = quant_half(in[i]);
}

uint16_t orc_f32_to_f16(float f) { return f32_to_f16_bits(f); }
float orc_f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }

}  // extern "C"
