// fx_bc6h.hip -- the radiance-asset path of LightProbe::Init (/root/reference/FluidX12/Content/LightProbe.cpp:41-46): a DDS
// cube map in DXGI_FORMAT_BC6H_UF16, the format of the reference's Bin/Assets/rnl_cross.dds.  In the
// reference the DDS container is read by XUSG and the blocks are decoded by the texture unit; here a host parser finds the
// requested mip of each face and k_bc6h_decode expands it: one thread per 4 x 4 block (16 B in, 16 texels x 3 floats out),
// header fields gathered through a per-mode bit map in constant memory, endpoints transformed/unquantised, texels
// interpolated with the 3- or 4-bit weights and finished as (x * 31) >> 6 = the half-float bit pattern (BC6H section of the
// D3D11 functional spec).  512 KiB of blocks -> 4.7 MiB of floats for the whole asset: a one-shot, launch-latency-sized pass.
#include "fx_internal.h"
#include <cstring>
#include <mutex>

namespace fx {

namespace {

enum { F_M = 0, F_D, F_RW, F_RX, F_RY, F_RZ, F_GW, F_GX, F_GY, F_GZ, F_BW, F_BX, F_BY, F_BZ, F_COUNT };

struct ModeInfo {
	uint8_t header_bits, transformed, regions, wbits, dbits[3], pad;
	uint8_t map[82][2];            // header bit i -> (field, bit of the field)
};

struct ModeSrc { int mode_bits, mode_value, transformed, regions, wbits, dr, dg, db; const char* layout; };

// D3D11 functional spec, BC6H mode table: fields LSB first (m mode, d partition, rw/gw/bw endpoint 0, r/g/b x, y, z = 1, 2, 3)
const ModeSrc kSrc[14] = {
	{ 2, 0x00, 1, 2, 10, 5, 5, 5, "m[1:0] gy[4] by[4] bz[4] rw[9:0] gw[9:0] bw[9:0] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 2, 0x01, 1, 2, 7, 6, 6, 6, "m[1:0] gy[5] gz[4] gz[5] rw[6:0] bz[0] bz[1] by[4] gw[6:0] by[5] bz[2] gy[4] bw[6:0] bz[3] bz[5] bz[4] rx[5:0] gy[3:0] gx[5:0] gz[3:0] bx[5:0] by[3:0] ry[5:0] rz[5:0] d[4:0]" },
	{ 5, 0x02, 1, 2, 11, 5, 4, 4, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[4:0] rw[10] gy[3:0] gx[3:0] gw[10] bz[0] gz[3:0] bx[3:0] bw[10] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x06, 1, 2, 11, 4, 5, 4, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10] gz[4] gy[3:0] gx[4:0] gw[10] gz[3:0] bx[3:0] bw[10] bz[1] by[3:0] ry[3:0] bz[0] bz[2] rz[3:0] gy[4] bz[3] d[4:0]" },
	{ 5, 0x0a, 1, 2, 11, 4, 4, 5, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10] by[4] gy[3:0] gx[3:0] gw[10] bz[0] gz[3:0] bx[4:0] bw[10] by[3:0] ry[3:0] bz[1] bz[2] rz[3:0] bz[4] bz[3] d[4:0]" },
	{ 5, 0x0e, 1, 2, 9, 5, 5, 5, "m[4:0] rw[8:0] by[4] gw[8:0] gy[4] bw[8:0] bz[4] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x12, 1, 2, 8, 6, 5, 5, "m[4:0] rw[7:0] gz[4] by[4] gw[7:0] bz[2] gy[4] bw[7:0] bz[3] bz[4] rx[5:0] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[5:0] rz[5:0] d[4:0]" },
	{ 5, 0x16, 1, 2, 8, 5, 6, 5, "m[4:0] rw[7:0] bz[0] by[4] gw[7:0] gy[5] gy[4] bw[7:0] gz[5] bz[4] rx[4:0] gz[4] gy[3:0] gx[5:0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x1a, 1, 2, 8, 5, 5, 6, "m[4:0] rw[7:0] bz[1] by[4] gw[7:0] by[5] gy[4] bw[7:0] bz[5] bz[4] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[5:0] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x1e, 0, 2, 6, 6, 6, 6, "m[4:0] rw[5:0] gz[4] bz[0] bz[1] by[4] gw[5:0] gy[5] by[5] bz[2] gy[4] bw[5:0] gz[5] bz[3] bz[5] bz[4] rx[5:0] gy[3:0] gx[5:0] gz[3:0] bx[5:0] by[3:0] ry[5:0] rz[5:0] d[4:0]" },
	{ 5, 0x03, 0, 1, 10, 10, 10, 10, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[9:0] gx[9:0] bx[9:0]" },
	{ 5, 0x07, 1, 1, 11, 9, 9, 9, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[8:0] rw[10] gx[8:0] gw[10] bx[8:0] bw[10]" },
	{ 5, 0x0b, 1, 1, 12, 8, 8, 8, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[7:0] rw[11] rw[10] gx[7:0] gw[11] gw[10] bx[7:0] bw[11] bw[10]" },
	{ 5, 0x0f, 1, 1, 16, 4, 4, 4, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[15] rw[14] rw[13] rw[12] rw[11] rw[10] gx[3:0] gw[15] gw[14] gw[13] gw[12] gw[11] gw[10] bx[3:0] bw[15] bw[14] bw[13] bw[12] bw[11] bw[10]" },
};

__constant__ ModeInfo c_modes[14];
__constant__ int8_t c_mode_of[32];                 // low 5 block bits -> mode index, -1 = reserved
__constant__ uint32_t c_partition[32];             // bit t = subset of texel t
__constant__ uint8_t c_anchor[32];
__constant__ uint8_t c_w3[8], c_w4[16];

const char* kPartitionRows[32] = {
	"0011001100110011", "0001000100010001", "0111011101110111", "0001001100110111", "0000000100010011", "0011011101111111", "0001001101111111", "0000000100110111",
	"0000000000010011", "0011011111111111", "0000000101111111", "0000000000010111", "0001011111111111", "0000000011111111", "0000111111111111", "0000000000001111",
	"0000100011101111", "0111000100000000", "0000000010001110", "0111001100010000", "0011000100000000", "0000100011001110", "0000000010001100", "0111001100110001",
	"0011000100010000", "0000100010001100", "0110011001100110", "0011011001101100", "0001011111101000", "0000111111110000", "0111000110001110", "0011100110011100",
};

hipError_t upload_tables()
{
	static std::once_flag once;
	static hipError_t status = hipSuccess;
	std::call_once(once, [] {
		static const char* names[F_COUNT] = { "m", "d", "rw", "rx", "ry", "rz", "gw", "gx", "gy", "gz", "bw", "bx", "by", "bz" };
		ModeInfo info[14];
		std::memset(info, 0, sizeof info);
		int8_t mode_of[32];
		for (int i = 0; i < 32; ++i) mode_of[i] = -1;
		for (int mi = 0; mi < 14; ++mi) {
			const ModeSrc& src = kSrc[mi];
			ModeInfo& o = info[mi];
			o.transformed = (uint8_t)src.transformed; o.regions = (uint8_t)src.regions; o.wbits = (uint8_t)src.wbits;
			o.dbits[0] = (uint8_t)src.dr; o.dbits[1] = (uint8_t)src.dg; o.dbits[2] = (uint8_t)src.db;
			int pos = 0;
			for (const char* p = src.layout; *p;) {
				while (*p == ' ') ++p;
				if (!*p) break;
				char name[4] = { 0, 0, 0, 0 };
				int nl = 0;
				while (*p && *p != '[') name[nl++] = *p++;
				++p;
				int hi = 0, lo;
				while (*p >= '0' && *p <= '9') hi = hi * 10 + (*p++ - '0');
				lo = hi;
				if (*p == ':') { ++p; lo = 0; while (*p >= '0' && *p <= '9') lo = lo * 10 + (*p++ - '0'); }
				++p;
				int id = 0;
				for (int k = 0; k < F_COUNT; ++k) if (!std::strcmp(name, names[k])) id = k;
				for (int b = lo; b <= hi; ++b) { o.map[pos][0] = (uint8_t)id; o.map[pos][1] = (uint8_t)b; ++pos; }
			}
			o.header_bits = (uint8_t)pos;
			if (src.mode_bits == 2) { for (int hi5 = 0; hi5 < 8; ++hi5) mode_of[(hi5 << 2) | src.mode_value] = (int8_t)mi; }
			else mode_of[src.mode_value] = (int8_t)mi;
		}
		uint32_t part[32];
		for (int i = 0; i < 32; ++i) { part[i] = 0; for (int t = 0; t < 16; ++t) if (kPartitionRows[i][t] == '1') part[i] |= 1u << t; }
		const uint8_t anchor[32] = { 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,2,8,2,2,8,8,15,2,8,2,2,8,8,2,2 };
		const uint8_t w3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 };
		const uint8_t w4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
		status = hipMemcpyToSymbol(HIP_SYMBOL(c_modes), info, sizeof info);
		if (status == hipSuccess) status = hipMemcpyToSymbol(HIP_SYMBOL(c_mode_of), mode_of, sizeof mode_of);
		if (status == hipSuccess) status = hipMemcpyToSymbol(HIP_SYMBOL(c_partition), part, sizeof part);
		if (status == hipSuccess) status = hipMemcpyToSymbol(HIP_SYMBOL(c_anchor), anchor, sizeof anchor);
		if (status == hipSuccess) status = hipMemcpyToSymbol(HIP_SYMBOL(c_w3), w3, sizeof w3);
		if (status == hipSuccess) status = hipMemcpyToSymbol(HIP_SYMBOL(c_w4), w4, sizeof w4);
	});
	return status;
}

__device__ __forceinline__ uint32_t blk_bit(const uint32_t w[4], int pos) { return (w[pos >> 5] >> (pos & 31)) & 1u; }
__device__ __forceinline__ uint32_t blk_bits(const uint32_t w[4], int pos, int n)
{
	const uint64_t lo = ((uint64_t)w[min((pos >> 5) + 1, 3)] << 32) | w[pos >> 5];
	return (uint32_t)(lo >> (pos & 31)) & ((1u << n) - 1u);
}

__device__ __forceinline__ float half_bits_to_float(uint32_t h)      // positive halves only (UF16)
{
	const uint32_t e = h >> 10, m = h & 0x3FFu;
	if (e == 0) return (float)m * 5.9604644775390625e-08f;            // subnormal: m * 2^-24
	if (e == 31) return __uint_as_float(0x7F800000u | (m << 13));
	return __uint_as_float(((e + 112u) << 23) | (m << 13));
}

}  // namespace

// blocks: nbx x nby blocks of one image, row-major; out: float[n][n][3] (n = image edge; partial edge blocks are clipped)
__global__ __launch_bounds__(64) void k_bc6h_decode(const uint4* __restrict__ blocks, int nbx, int nby, int n, float* __restrict__ out)
{
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nbx * nby) return;
	const uint4 q = blocks[b];
	const uint32_t w[4] = { q.x, q.y, q.z, q.w };
	const int mi = c_mode_of[w[0] & 31u];
	const int bx = b % nbx, by = b / nbx;
	if (mi < 0) {                                          // reserved modes decode to black
		for (int t = 0; t < 16; ++t) {
			const int x = bx * 4 + (t & 3), y = by * 4 + (t >> 2);
			if (x < n && y < n) { float* o = out + ((size_t)y * n + x) * 3; o[0] = o[1] = o[2] = 0.0f; }
		}
		return;
	}
	const ModeInfo& md = c_modes[mi];
	int f[F_COUNT];
#pragma unroll
	for (int i = 0; i < F_COUNT; ++i) f[i] = 0;
	for (int i = 0; i < md.header_bits; ++i) {
		const int id = md.map[i][0];
		const int v = (int)blk_bit(w, i) << md.map[i][1];
		// a switch keeps f[] in registers (a dynamically indexed array would go to scratch)
		switch (id) {
		case F_D: f[F_D] |= v; break;   case F_RW: f[F_RW] |= v; break; case F_RX: f[F_RX] |= v; break; case F_RY: f[F_RY] |= v; break;
		case F_RZ: f[F_RZ] |= v; break; case F_GW: f[F_GW] |= v; break; case F_GX: f[F_GX] |= v; break; case F_GY: f[F_GY] |= v; break;
		case F_GZ: f[F_GZ] |= v; break; case F_BW: f[F_BW] |= v; break; case F_BX: f[F_BX] |= v; break; case F_BY: f[F_BY] |= v; break;
		case F_BZ: f[F_BZ] |= v; break; default: break;
		}
	}
	int e[4][3] = { { f[F_RW], f[F_GW], f[F_BW] }, { f[F_RX], f[F_GX], f[F_BX] }, { f[F_RY], f[F_GY], f[F_BY] }, { f[F_RZ], f[F_GZ], f[F_BZ] } };
	const int wb = md.wbits, ne = md.regions * 2;
#pragma unroll
	for (int k = 1; k < 4; ++k)
#pragma unroll
		for (int c = 0; c < 3; ++c)
			if (md.transformed && k < ne) {
				const int db = md.dbits[c];
				const int d = (e[k][c] & (1 << (db - 1))) ? e[k][c] - (1 << db) : e[k][c];
				e[k][c] = (e[0][c] + d) & ((1 << wb) - 1);
			}
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const int v = e[k][c];
			e[k][c] = wb >= 15 ? v : v == 0 ? 0 : v == (1 << wb) - 1 ? 0xFFFF : ((v << 16) + 0x8000) >> wb;
		}
	const bool two = md.regions == 2;
	const int ib = two ? 3 : 4;
	const uint32_t part = two ? c_partition[f[F_D] & 31] : 0u;
	const int anchor2 = two ? c_anchor[f[F_D] & 31] : -1;
	int pos = md.header_bits;
	for (int t = 0; t < 16; ++t) {
		const int nb = (t == 0 || t == anchor2) ? ib - 1 : ib;
		const int idx = (int)blk_bits(w, pos, nb);
		pos += nb;
		const int wt = two ? c_w3[idx] : c_w4[idx];
		const int s = (part >> t) & 1u;
		const int x = bx * 4 + (t & 3), y = by * 4 + (t >> 2);
		if (x >= n || y >= n) continue;
		float* o = out + ((size_t)y * n + x) * 3;
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const int a = s ? e[2][c] : e[0][c], bb = s ? e[3][c] : e[1][c];
			const int v = (a * (64 - wt) + bb * wt + 32) >> 6;
			o[c] = half_bits_to_float((uint32_t)((v * 31) >> 6));
		}
	}
}

hipError_t launch_bc6h_decode(const void* blocks_dev, int nbx, int nby, int n, float* out_dev, hipStream_t s)
{
	hipError_t e = upload_tables();
	if (e != hipSuccess) return e;
	const int nb = nbx * nby;
	hipLaunchKernelGGL(k_bc6h_decode, dim3((nb + 63) / 64), dim3(64), 0, s, reinterpret_cast<const uint4*>(blocks_dev), nbx, nby, n, out_dev);
	return hipGetLastError();
}

// DDS container: magic, DDS_HEADER (124 B), DDS_HEADER_DXT10 (20 B); cube faces +X -X +Y -Y +Z -Z, each with its mip chain
// The DDS cube maps LightProbe::Init can be handed (LightProbe.cpp:41-46 goes through XUSG's DDS loader, which takes any DXGI format):
// DX10-header files in BC6H_UF16 (the reference's own asset) and in the uncompressed formats an HDR or LDR cube is commonly saved in --
// R32G32B32A32_FLOAT, R32G32B32_FLOAT, R16G16B16A16_FLOAT, R8G8B8A8_UNORM --, and legacy headers whose FourCC is D3DFMT_A16B16G16R16F (113)
// or D3DFMT_A32B32G32R32F (116).  Faces +X -X +Y -Y +Z -Z, each followed by its mip chain.
bool dds_cube_layout(const void* dds, size_t bytes, DdsCube* out)
{
	const uint8_t* p = static_cast<const uint8_t*>(dds);
	if (!p || !out || bytes < 128 || std::memcmp(p, "DDS ", 4) != 0) return false;
	uint32_t h[31];
	std::memcpy(h, p + 4, sizeof h);
	if (h[0] != 124 || h[18] != 32) return false;
	const uint32_t height = h[2], width = h[3], nm = h[6] ? h[6] : 1;
	size_t data = 128;
	int kind = -1;
	if (std::memcmp(&h[20], "DX10", 4) == 0) {
		if (bytes < 148) return false;
		uint32_t dx[5];
		std::memcpy(dx, p + 128, sizeof dx);
		if (!(dx[2] & 4u) /* TEXTURECUBE */ || dx[1] != 3 /* TEXTURE2D */) return false;
		switch (dx[0]) {
		case 95: kind = DDS_BC6H_UF16; break;
		case 2: kind = DDS_RGBA32F; break;
		case 6: kind = DDS_RGB32F; break;
		case 10: kind = DDS_RGBA16F; break;
		case 28: kind = DDS_RGBA8; break;
		default: return false;
		}
		data = 148;
	} else if ((h[19] & 4u) /* DDPF_FOURCC */ && (h[27] & 0x200u) /* DDSCAPS2_CUBEMAP */ && (h[27] & 0xFC00u) == 0xFC00u /* all six faces */) {
		if (h[20] == 113) kind = DDS_RGBA16F;
		else if (h[20] == 116) kind = DDS_RGBA32F;
		else return false;
	} else return false;
	if (width != height || !width || nm > 15) return false;
	// D3D11's largest texture extent; also keeps (width + 3) inside 32 bits (0xFFFFFFFF used to wrap to a zero-block face and pass)
	if (width > 16384u || (nm > 1 && (width >> (nm - 1)) == 0)) return false;
	static const size_t texel[] = { 0, 16, 12, 8, 4 };
	size_t per_face = 0;
	for (uint32_t m = 0; m < nm; ++m) {
		const size_t n = width >> m ? width >> m : 1, nb = (n + 3) / 4;
		out->mip_offset[m] = per_face;
		per_face += kind == DDS_BC6H_UF16 ? nb * nb * 16 : n * n * texel[kind];
	}
	if (data + 6 * per_face > bytes) return false;
	for (int f = 0; f < 6; ++f) out->face_offset[f] = data + (size_t)f * per_face;
	out->size = width; out->mips = nm; out->kind = kind;
	return true;
}

static float half_bits_to_float(uint16_t v)
{
	const uint32_t s = (uint32_t)(v & 0x8000u) << 16, e = (v >> 10) & 31u, m = v & 1023u;
	uint32_t u;
	if (e == 0) {
		if (!m) u = s;
		else { int k = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++k; } u = s | ((uint32_t)(113 - k) << 23) | ((mm & 1023u) << 13); }
	} else if (e == 31) u = s | 0x7F800000u | (m << 13);
	else u = s | ((e + 112u) << 23) | (m << 13);
	float f;
	std::memcpy(&f, &u, 4);
	return f;
}

// one face of an uncompressed mip -> float rgb (host; a light probe is a few hundred kilobytes)
void dds_linear_face_to_rgb(const void* texels, int kind, size_t n, float* out)
{
	const uint8_t* p = static_cast<const uint8_t*>(texels);
	for (size_t i = 0; i < n * n; ++i) {
		float c[3];
		if (kind == DDS_RGBA32F || kind == DDS_RGB32F) std::memcpy(c, p + i * (kind == DDS_RGBA32F ? 16 : 12), 12);
		else if (kind == DDS_RGBA16F) { uint16_t hh[3]; std::memcpy(hh, p + i * 8, 6); for (int k = 0; k < 3; ++k) c[k] = half_bits_to_float(hh[k]); }
		else for (int k = 0; k < 3; ++k) c[k] = (float)p[i * 4 + k] / 255.0f;
		out[3 * i] = c[0]; out[3 * i + 1] = c[1]; out[3 * i + 2] = c[2];
	}
}

}  // namespace fx
