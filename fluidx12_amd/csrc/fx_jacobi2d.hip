// fx_jacobi2d.hip -- the 2-D pressure relaxation (CSProject2D.hlsl:64-106 with CSPoisson.hlsli:8-26, /root/reference/FluidX12/Content/
// Shaders/): up to eight lock-step sweeps per launch on an LDS tile.
//
// A 2-D grid is small (Bin/Fluid2D.bat: 512 x 512 x 1 = 1 MiB of pressure): one sweep per launch (k_jacobi_generic) is all launch
// boundary -- 64 launches of ~3 us for the reference's loop.  Here a workgroup owns a 64 x 16-cell tile, stages it with a halo of T
// cells (p, b and -- reference configuration -- the freeze bytes) in the LDS and relaxes T times there: level k on the tile grown by
// T - k cells, halo cells RECOMPUTED with their freeze decisions (the same deterministic arithmetic their owner tile does), then
// stores its core.  The freeze bytes are DOUBLE-BUFFERED like the pressure (frozen_in -> frozen_out): a workgroup stages the bytes of its
// halo, which its neighbours own and rewrite -- in place, a workgroup that starts after a neighbour has finished would see halo cells
// frozen from level 0 instead of from the level at which they froze (every tile of a 512 x 512 grid is co-resident and stages before any
// stores, so only grids of more tiles than the chip holds showed it; tests/test_gpu_sim.py::test_2d_tile_kernel_on_more_tiles_than_fit).
// Per cell: s = (((L - b) + R) + U) + D, x = s * 1/4, frozen for good once |fma(s, 1/4, -x0)| < 1e-3 -- the
// arithmetic and order of k_jacobi_generic, so T sweeps here equal T launches of it bit for bit (tests/test_gpu_sim.py, the
// CSProject2D goldens, the 2-D fuzz family).
#include "fx_internal.h"

namespace fx {

namespace {
constexpr int kT = 8;                           // most sweeps per launch
constexpr int TX2 = 64, TY2 = 16;               // core
constexpr int EX2 = TX2 + 2 * kT, EY2 = TY2 + 2 * kT;
constexpr int NC2 = EX2 * EY2;                  // staged cells (80 x 32 = 2560)

template <bool MASK>
__global__ __launch_bounds__(256) void k_jacobi2d_tile(const Geom g, const float* __restrict__ p_in, const float* __restrict__ b,
	float* __restrict__ p_out, const uint8_t* __restrict__ frozen_in, uint8_t* __restrict__ frozen_out, int T)
{
	__shared__ float P[2][NC2];
	__shared__ float Bs[NC2];
	__shared__ uint8_t M[NC2];
	const int tid = (int)threadIdx.x;
	const int x0 = (int)blockIdx.x * TX2 - kT, y0 = (int)blockIdx.y * TY2 - kT;    // grid cell of staged cell (0, 0)
	// stage: cells outside the grid are never read (a border cell's missing neighbour is the cell itself), so they load a clamped copy
	for (int i = tid; i < NC2; i += 256) {
		const int ey = i / EX2, ex = i - ey * EX2;
		const int gx = min(max(x0 + ex, 0), g.X - 1), gy = min(max(y0 + ey, 0), g.Y - 1);
		const size_t id = (size_t)gy * g.X + gx;
		P[0][i] = p_in[id];
		Bs[i] = b[id];
		M[i] = MASK ? frozen_in[id] : (uint8_t)0;
	}
	__syncthreads();
	int cur = 0;
	for (int k = 1; k <= T; ++k) {
		const int m = kT - T + k;                                   // level k is needed on the core grown by T - k cells
		const int nx = EX2 - 2 * m, ny = EY2 - 2 * m;
		for (int i = tid; i < nx * ny; i += 256) {
			const int ry = i / nx, ex = m + i - ry * nx, ey = m + ry;
			const int gx = x0 + ex, gy = y0 + ey;
			if (gx < 0 || gx >= g.X || gy < 0 || gy >= g.Y) continue;
			const int c = ey * EX2 + ex;
			const float xc = P[cur][c];
			if (MASK && M[c]) { P[cur ^ 1][c] = xc; continue; }
			const float L = gx == 0 ? xc : P[cur][c - 1], R = gx == g.X - 1 ? xc : P[cur][c + 1];
			const float U = gy == 0 ? xc : P[cur][c - EX2], D = gy == g.Y - 1 ? xc : P[cur][c + EX2];
			float s = L - Bs[c];
			s = R + s; s = U + s; s = D + s;
			P[cur ^ 1][c] = s * 0.25f;
			if (MASK && fabsf(fmaf(s, 0.25f, -xc)) < 0.00100000005f) M[c] = 1;     // CSPoisson.hlsli:24
		}
		__syncthreads();
		cur ^= 1;
	}
	for (int i = tid; i < TX2 * TY2; i += 256) {
		const int ry = i / TX2, ex = kT + i - ry * TX2, ey = kT + ry;
		const int gx = x0 + ex, gy = y0 + ey;
		if (gx >= g.X || gy >= g.Y) continue;
		const size_t id = (size_t)gy * g.X + gx;
		p_out[id] = P[cur][ey * EX2 + ex];
		if (MASK) frozen_out[id] = M[ey * EX2 + ex];
	}
}
}  // namespace

int jacobi2d_max_sweeps(const Geom& g) { return g.Zg == 1 && g.nz == 1 && FX_KNOB_INT("JACOBI2D_TILE", 1) ? kT : 0; }

// `sweeps` (1 .. jacobi2d_max_sweeps) lock-step sweeps p_in -> p_out of a 2-D grid; frozen_in / frozen_out (both null, or two DIFFERENT
// buffers): the freeze bytes before and after
hipError_t launch_jacobi2d(const Geom& g, const float* p_in, const float* b, float* p_out, const uint8_t* frozen_in, uint8_t* frozen_out, int sweeps, hipStream_t s)
{
	if (g.Zg != 1 || sweeps < 1 || sweeps > kT) return hipErrorNotSupported;
	if ((frozen_in == nullptr) != (frozen_out == nullptr) || (frozen_in && frozen_in == frozen_out)) return hipErrorInvalidValue;
	const dim3 grid((g.X + TX2 - 1) / TX2, (g.Y + TY2 - 1) / TY2, 1), block(256);
	if (frozen_in) hipLaunchKernelGGL(k_jacobi2d_tile<true>, grid, block, 0, s, g, p_in, b, p_out, frozen_in, frozen_out, sweeps);
	else hipLaunchKernelGGL(k_jacobi2d_tile<false>, grid, block, 0, s, g, p_in, b, p_out, frozen_in, frozen_out, sweeps);
	return hipGetLastError();
}

}  // namespace fx
