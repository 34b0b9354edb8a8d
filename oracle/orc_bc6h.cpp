// orc_bc6h.cpp -- TEST INFRASTRUCTURE (CPU oracle), never linked into the product.
//
// BC6H_UF16 block decoding and the DDS cube-map container, i.e. what D3D12's sampler and XUSG's DDS loader do for the
// reference when LightProbe::Init loads Bin/Assets/rnl_cross.dds (/root/reference/FluidX12/Content/LightProbe.cpp:41-46;
// 256^2 x 6 faces, 9 mips, DXGI_FORMAT_BC6H_UF16 = 95 in a DX10 header).  Neither decoder is in the reference tree
// (hardware / un-vendored XUSG), so this restates the published format: the BC6H section of the D3D11
// functional spec (mode table, partition/anchor tables, unquantisation, interpolation weights, the x * 31 / 64 finish) and
// the DDS header layout of the DirectX documentation.  Pinned by known-answer blocks, by the decoder agreeing with itself
// across the asset's mip chain (mip n+1 is a box filter of mip n to within BC6H quantisation, encoded in other modes and
// partitions) and by the HIP decoder matching it bit for bit on random blocks (every 128-bit pattern is a defined block).
#include "orc_common.h"
#include "fx_oracle.h"
#include <string>
#include <vector>

namespace {

// field ids
enum { M = 0, D, RW, RX, RY, RZ, GW, GX, GY, GZ, BW, BX, BY, BZ, NF };

struct ModeDesc {
	int mode_bits, mode_value;      // low bits of the block
	bool transformed;
	int regions;                    // 1 or 2
	int wbits;                      // endpoint-0 precision
	int dbits[3];                   // delta (or endpoint 1..3) precision r, g, b
	const char* layout;             // header fields LSB first: "name[hi:lo]" or "name[bit]"
};

// D3D11 functional spec, BC6H mode table (m = mode, d = partition, rw/gw/bw = endpoint 0, rx.. = 1, ry.. = 2, rz.. = 3)
const ModeDesc kModes[14] = {
	{ 2, 0x00, true, 2, 10, { 5, 5, 5 }, "m[1:0] gy[4] by[4] bz[4] rw[9:0] gw[9:0] bw[9:0] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 2, 0x01, true, 2, 7, { 6, 6, 6 }, "m[1:0] gy[5] gz[4] gz[5] rw[6:0] bz[0] bz[1] by[4] gw[6:0] by[5] bz[2] gy[4] bw[6:0] bz[3] bz[5] bz[4] rx[5:0] gy[3:0] gx[5:0] gz[3:0] bx[5:0] by[3:0] ry[5:0] rz[5:0] d[4:0]" },
	{ 5, 0x02, true, 2, 11, { 5, 4, 4 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[4:0] rw[10] gy[3:0] gx[3:0] gw[10] bz[0] gz[3:0] bx[3:0] bw[10] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x06, true, 2, 11, { 4, 5, 4 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10] gz[4] gy[3:0] gx[4:0] gw[10] gz[3:0] bx[3:0] bw[10] bz[1] by[3:0] ry[3:0] bz[0] bz[2] rz[3:0] gy[4] bz[3] d[4:0]" },
	{ 5, 0x0a, true, 2, 11, { 4, 4, 5 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[10] by[4] gy[3:0] gx[3:0] gw[10] bz[0] gz[3:0] bx[4:0] bw[10] by[3:0] ry[3:0] bz[1] bz[2] rz[3:0] bz[4] bz[3] d[4:0]" },
	{ 5, 0x0e, true, 2, 9, { 5, 5, 5 }, "m[4:0] rw[8:0] by[4] gw[8:0] gy[4] bw[8:0] bz[4] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x12, true, 2, 8, { 6, 5, 5 }, "m[4:0] rw[7:0] gz[4] by[4] gw[7:0] bz[2] gy[4] bw[7:0] bz[3] bz[4] rx[5:0] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[5:0] rz[5:0] d[4:0]" },
	{ 5, 0x16, true, 2, 8, { 5, 6, 5 }, "m[4:0] rw[7:0] bz[0] by[4] gw[7:0] gy[5] gy[4] bw[7:0] gz[5] bz[4] rx[4:0] gz[4] gy[3:0] gx[5:0] gz[3:0] bx[4:0] bz[1] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x1a, true, 2, 8, { 5, 5, 6 }, "m[4:0] rw[7:0] bz[1] by[4] gw[7:0] by[5] gy[4] bw[7:0] bz[5] bz[4] rx[4:0] gz[4] gy[3:0] gx[4:0] bz[0] gz[3:0] bx[5:0] by[3:0] ry[4:0] bz[2] rz[4:0] bz[3] d[4:0]" },
	{ 5, 0x1e, false, 2, 6, { 6, 6, 6 }, "m[4:0] rw[5:0] gz[4] bz[0] bz[1] by[4] gw[5:0] gy[5] by[5] bz[2] gy[4] bw[5:0] gz[5] bz[3] bz[5] bz[4] rx[5:0] gy[3:0] gx[5:0] gz[3:0] bx[5:0] by[3:0] ry[5:0] rz[5:0] d[4:0]" },
	{ 5, 0x03, false, 1, 10, { 10, 10, 10 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[9:0] gx[9:0] bx[9:0]" },
	{ 5, 0x07, true, 1, 11, { 9, 9, 9 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[8:0] rw[10] gx[8:0] gw[10] bx[8:0] bw[10]" },
	{ 5, 0x0b, true, 1, 12, { 8, 8, 8 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[7:0] rw[11] rw[10] gx[7:0] gw[11] gw[10] bx[7:0] bw[11] bw[10]" },
	{ 5, 0x0f, true, 1, 16, { 4, 4, 4 }, "m[4:0] rw[9:0] gw[9:0] bw[9:0] rx[3:0] rw[15] rw[14] rw[13] rw[12] rw[11] rw[10] gx[3:0] gw[15] gw[14] gw[13] gw[12] gw[11] gw[10] bx[3:0] bw[15] bw[14] bw[13] bw[12] bw[11] bw[10]" },
};

// two-subset partitions 0..31 (shared with BC7) and the anchor (fix-up) index of the second subset
const uint8_t kPartition[32][16] = {
	{ 0,0,1,1,0,0,1,1,0,0,1,1,0,0,1,1 }, { 0,0,0,1,0,0,0,1,0,0,0,1,0,0,0,1 }, { 0,1,1,1,0,1,1,1,0,1,1,1,0,1,1,1 }, { 0,0,0,1,0,0,1,1,0,0,1,1,0,1,1,1 },
	{ 0,0,0,0,0,0,0,1,0,0,0,1,0,0,1,1 }, { 0,0,1,1,0,1,1,1,0,1,1,1,1,1,1,1 }, { 0,0,0,1,0,0,1,1,0,1,1,1,1,1,1,1 }, { 0,0,0,0,0,0,0,1,0,0,1,1,0,1,1,1 },
	{ 0,0,0,0,0,0,0,0,0,0,0,1,0,0,1,1 }, { 0,0,1,1,0,1,1,1,1,1,1,1,1,1,1,1 }, { 0,0,0,0,0,0,0,1,0,1,1,1,1,1,1,1 }, { 0,0,0,0,0,0,0,0,0,0,0,1,0,1,1,1 },
	{ 0,0,0,1,0,1,1,1,1,1,1,1,1,1,1,1 }, { 0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1 }, { 0,0,0,0,1,1,1,1,1,1,1,1,1,1,1,1 }, { 0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1 },
	{ 0,0,0,0,1,0,0,0,1,1,1,0,1,1,1,1 }, { 0,1,1,1,0,0,0,1,0,0,0,0,0,0,0,0 }, { 0,0,0,0,0,0,0,0,1,0,0,0,1,1,1,0 }, { 0,1,1,1,0,0,1,1,0,0,0,1,0,0,0,0 },
	{ 0,0,1,1,0,0,0,1,0,0,0,0,0,0,0,0 }, { 0,0,0,0,1,0,0,0,1,1,0,0,1,1,1,0 }, { 0,0,0,0,0,0,0,0,1,0,0,0,1,1,0,0 }, { 0,1,1,1,0,0,1,1,0,0,1,1,0,0,0,1 },
	{ 0,0,1,1,0,0,0,1,0,0,0,1,0,0,0,0 }, { 0,0,0,0,1,0,0,0,1,0,0,0,1,1,0,0 }, { 0,1,1,0,0,1,1,0,0,1,1,0,0,1,1,0 }, { 0,0,1,1,0,1,1,0,0,1,1,0,1,1,0,0 },
	{ 0,0,0,1,0,1,1,1,1,1,1,0,1,0,0,0 }, { 0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0 }, { 0,1,1,1,0,0,0,1,1,0,0,0,1,1,1,0 }, { 0,0,1,1,1,0,0,1,1,0,0,1,1,1,0,0 },
};
const uint8_t kAnchor[32] = { 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,2,8,2,2,8,8,15,2,8,2,2,8,8,2,2 };
const int kWeights3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 };
const int kWeights4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };

inline int get_bit(const uint8_t* blk, int pos) { return (blk[pos >> 3] >> (pos & 7)) & 1; }
inline int get_bits(const uint8_t* blk, int pos, int n)
{
	int v = 0;
	for (int i = 0; i < n; ++i) v |= get_bit(blk, pos + i) << i;
	return v;
}

int field_id(const std::string& n)
{
	static const char* names[NF] = { "m", "d", "rw", "rx", "ry", "rz", "gw", "gx", "gy", "gz", "bw", "bx", "by", "bz" };
	for (int i = 0; i < NF; ++i) if (n == names[i]) return i;
	return -1;
}

// walk the layout string, filling f[] and returning the number of header bits
int parse_header(const uint8_t* blk, const ModeDesc& md, int f[NF])
{
	for (int i = 0; i < NF; ++i) f[i] = 0;
	int pos = 0;
	const char* p = md.layout;
	while (*p) {
		while (*p == ' ') ++p;
		if (!*p) break;
		std::string name;
		while (*p && *p != '[') name += *p++;
		++p;
		int hi = 0, lo;
		while (*p >= '0' && *p <= '9') hi = hi * 10 + (*p++ - '0');
		lo = hi;
		if (*p == ':') { ++p; lo = 0; while (*p >= '0' && *p <= '9') lo = lo * 10 + (*p++ - '0'); }
		++p;                                                    // ']'
		const int id = field_id(name);
		for (int b = lo; b <= hi; ++b) f[id] |= get_bit(blk, pos++) << b;
	}
	return pos;
}

inline int sext(int v, int bits) { return (v & (1 << (bits - 1))) ? v - (1 << bits) : v; }

inline int unquantize(int comp, int bits)                       // unsigned variant
{
	if (bits >= 15) return comp;
	if (comp == 0) return 0;
	if (comp == (1 << bits) - 1) return 0xFFFF;
	return ((comp << 16) + 0x8000) >> bits;
}

}  // namespace

extern "C" {

// one 16-byte BC6H_UF16 block -> 16 texels (row-major 4 x 4) of 3 half-float bit patterns; returns the mode 1..14, 0 = reserved
int orc_bc6h_decode_block(const uint8_t* blk, uint16_t* out_half_rgb /* [16][3] */)
{
	const int m2 = get_bits(blk, 0, 2), m5 = get_bits(blk, 0, 5);
	int mi = -1;
	for (int i = 0; i < 14; ++i)
		if ((kModes[i].mode_bits == 2 && m2 == kModes[i].mode_value) || (kModes[i].mode_bits == 5 && m2 >= 2 && m5 == kModes[i].mode_value)) { mi = i; break; }
	if (mi < 0) { for (int i = 0; i < 48; ++i) out_half_rgb[i] = 0; return 0; }      // reserved modes decode to black
	const ModeDesc& md = kModes[mi];
	int f[NF];
	const int hdr = parse_header(blk, md, f);
	int e[4][3] = { { f[RW], f[GW], f[BW] }, { f[RX], f[GX], f[BX] }, { f[RY], f[GY], f[BY] }, { f[RZ], f[GZ], f[BZ] } };
	const int ne = md.regions * 2;
	if (md.transformed) {
		const int mask = (1 << md.wbits) - 1;
		for (int k = 1; k < ne; ++k)
			for (int c = 0; c < 3; ++c) e[k][c] = (e[0][c] + sext(e[k][c], md.dbits[c])) & mask;
	}
	for (int k = 0; k < ne; ++k)
		for (int c = 0; c < 3; ++c) e[k][c] = unquantize(e[k][c], md.wbits);
	const int ibits = md.regions == 1 ? 4 : 3;
	const int* weights = md.regions == 1 ? kWeights4 : kWeights3;
	const uint8_t* part = kPartition[f[D] & 31];
	int pos = hdr;
	for (int t = 0; t < 16; ++t) {
		const int subset = md.regions == 2 ? part[t] : 0;
		const bool anchor = t == 0 || (md.regions == 2 && t == kAnchor[f[D] & 31]);
		const int nb = anchor ? ibits - 1 : ibits;
		const int idx = get_bits(blk, pos, nb);
		pos += nb;
		const int w = weights[idx];
		for (int c = 0; c < 3; ++c) {
			const int a = e[2 * subset][c], b = e[2 * subset + 1][c];
			const int v = (a * (64 - w) + b * w + 32) >> 6;
			out_half_rgb[t * 3 + c] = (uint16_t)((v * 31) >> 6);
		}
	}
	return mi + 1;
}

// DDS cube map in BC6H_UF16 (DX10 header): face `face`, mip `mip` -> float[n][n][3]; returns n (0 on a malformed file)
int orc_dds_bc6h_cube_face(const uint8_t* dds, size_t bytes, int face, int mip, float* out_rgb, int* mode_hist /* [15] or null */)
{
	if (bytes < 148 || std::memcmp(dds, "DDS ", 4) != 0) return 0;
	uint32_t h[31];
	std::memcpy(h, dds + 4, 124);
	const uint32_t height = h[2], width = h[3], mips = h[6] ? h[6] : 1;
	uint32_t dx[5];
	std::memcpy(dx, dds + 128, 20);
	if (h[0] != 124 || h[18] != 32 || std::memcmp(&h[20], "DX10", 4) != 0) return 0;
	if (dx[0] != 95 || !(dx[2] & 4u) || width != height || face < 0 || face > 5 || mip < 0 || (uint32_t)mip >= mips) return 0;
	if (!width || width > 16384u || mips > 15) return 0;                 // same bounds as the product's parser: (width + 3) must not wrap
	size_t face_bytes = 0, mip_off = 0;
	for (uint32_t m = 0; m < mips; ++m) {
		const size_t bw = std::max<uint32_t>(1, ((width >> m) + 3) / 4);
		if ((int)m == mip) mip_off = face_bytes;
		face_bytes += bw * bw * 16;
	}
	const int n = (int)std::max<uint32_t>(1, width >> mip), nb = (n + 3) / 4;
	const size_t off = 148 + (size_t)face * face_bytes + mip_off;
	if (off + (size_t)nb * nb * 16 > bytes) return 0;
	for (int by = 0; by < nb; ++by)
		for (int bx = 0; bx < nb; ++bx) {
			uint16_t hv[48];
			const int mode = orc_bc6h_decode_block(dds + off + ((size_t)by * nb + bx) * 16, hv);
			if (mode_hist) mode_hist[mode] += 1;
			for (int t = 0; t < 16; ++t) {
				const int x = bx * 4 + (t & 3), y = by * 4 + (t >> 2);
				if (x >= n || y >= n) continue;
				for (int c = 0; c < 3; ++c) out_rgb[((size_t)y * n + x) * 3 + c] = orc::f16_bits_to_f32(hv[t * 3 + c]);
			}
		}
	return n;
}

}  // extern "C"
