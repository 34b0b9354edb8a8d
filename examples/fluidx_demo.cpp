// fluidx_demo.cpp -- the frame loop of the reference's demo driver (FluidX12/FluidX12.cpp) on the C++ shim:
// OnUpdate's time-step rule (:266) and default camera (:243-253), PopulateCommandList's Simulate + Render
// (:465,489-490), minus the window.  Build (after `python -m fluidx12_amd.build`):
//   hipcc -std=c++17 examples/fluidx_demo.cpp -o fluidx_demo -Lfluidx12_amd -lfluidx_hip -Wl,-rpath,$PWD/fluidx12_amd
// Usage: fluidx_demo [-gridSize X Y Z] [-maxRaySamples N] [-maxLightSamples N] [-radiance cube.dds] [-frames N] [-screenshot out.png|out.ppm] [-resume in.fxck] [-checkpoint out.fxck]
// (FluidX12.cpp:398-433; the screen shot is a PNG like the reference's (FluidX12.cpp:640-660), written without a compression library, or a binary PPM by extension)
#include "../fluidx12_amd/csrc/Fluid.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

using namespace fluidx;

static XMFLOAT4X4 LookAtLH(const float eye[3], const float at[3], const float up[3])
{
	float z[3] = { at[0] - eye[0], at[1] - eye[1], at[2] - eye[2] };
	float l = std::sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
	for (float& v : z) v /= l;
	float x[3] = { up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0] };
	l = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
	for (float& v : x) v /= l;
	const float y[3] = { z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0] };
	auto dot = [&](const float* a) { return a[0] * eye[0] + a[1] * eye[1] + a[2] * eye[2]; };
	XMFLOAT4X4 m = { { { x[0], y[0], z[0], 0 }, { x[1], y[1], z[1], 0 }, { x[2], y[2], z[2], 0 }, { -dot(x), -dot(y), -dot(z), 1 } } };
	return m;
}

static XMFLOAT4X4 PerspectiveFovLH(float fovy, float aspect, float zn, float zf)
{
	const float h = std::cos(0.5f * fovy) / std::sin(0.5f * fovy), w = h / aspect, q = zf / (zf - zn);
	XMFLOAT4X4 m = { { { w, 0, 0, 0 }, { 0, h, 0, 0 }, { 0, 0, q, 1 }, { 0, 0, -q * zn, 0 } } };
	return m;
}

// A PNG of the RGBA8 render target, as FluidX12.cpp:640-660 saves one (8-bit RGBA, no interlace).  The pixel rows go into the zlib
// stream as STORED deflate blocks: every PNG reader takes that, and the demo needs no compression library.
static uint32_t crc32_of(const uint8_t* d, size_t n, uint32_t c = 0)
{
	static uint32_t table[256];
	if (!table[1]) for (uint32_t i = 0; i < 256; ++i) { uint32_t v = i; for (int k = 0; k < 8; ++k) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1; table[i] = v; }
	c = ~c;
	for (size_t i = 0; i < n; ++i) c = table[(c ^ d[i]) & 255u] ^ (c >> 8);
	return ~c;
}
static void png_chunk(FILE* fp, const char type[5], const std::vector<uint8_t>& body)
{
	std::vector<uint8_t> b(4 + body.size());
	std::memcpy(b.data(), type, 4);
	if (!body.empty()) std::memcpy(b.data() + 4, body.data(), body.size());
	const uint32_t len = (uint32_t)body.size(), crc = crc32_of(b.data(), b.size());
	const uint8_t l[4] = { (uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len };
	const uint8_t c[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
	std::fwrite(l, 1, 4, fp); std::fwrite(b.data(), 1, b.size(), fp); std::fwrite(c, 1, 4, fp);
}
static void write_png(FILE* fp, const uint8_t* rgba, uint32_t w, uint32_t h)
{
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	std::fwrite(sig, 1, 8, fp);
	std::vector<uint8_t> ihdr = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0 };
	png_chunk(fp, "IHDR", ihdr);
	std::vector<uint8_t> raw;                                            // filter byte 0 + the row
	raw.reserve((size_t)h * (1 + 4 * (size_t)w));
	for (uint32_t y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4); }
	std::vector<uint8_t> z = { 0x78, 0x01 };
	uint32_t a = 1, b = 0;                                               // Adler-32 of the raw bytes
	for (size_t pos = 0; pos < raw.size() || pos == 0;) {
		const size_t n = std::min<size_t>(65535, raw.size() - pos);
		z.push_back(pos + n >= raw.size() ? 1 : 0);                       // BFINAL, BTYPE = 00 (stored)
		z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8)); z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
		z.insert(z.end(), raw.begin() + (ptrdiff_t)pos, raw.begin() + (ptrdiff_t)(pos + n));
		for (size_t i = pos; i < pos + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
		pos += n;
		if (!n) break;
	}
	const uint32_t ad = (b << 16) | a;
	z.push_back((uint8_t)(ad >> 24)); z.push_back((uint8_t)(ad >> 16)); z.push_back((uint8_t)(ad >> 8)); z.push_back((uint8_t)ad);
	png_chunk(fp, "IDAT", z);
	png_chunk(fp, "IEND", {});
}

int main(int argc, char** argv)
{
	XMUINT3 grid = { 128, 128, 128 };                  // FluidX12.cpp:44
	uint32_t maxRay = 192, maxLight = 64, frames = 100; // FluidX12.cpp:38-39
	const uint32_t width = 800, height = 800;           // Main.cpp:17
	const char* screenshot = nullptr;
	const char* radiance = nullptr;                     // FluidGI.bat: -radiance Assets/rnl_cross.dds
	const char* resume = nullptr;                       // not in the reference: continue from / leave behind a state file
	const char* checkpoint = nullptr;
	for (int i = 1; i < argc; ++i) {
		if (!std::strcmp(argv[i], "-gridSize") && i + 3 < argc) { grid.x = atoi(argv[++i]); grid.y = atoi(argv[++i]); grid.z = atoi(argv[++i]); }
		else if (!std::strcmp(argv[i], "-maxRaySamples") && i + 1 < argc) maxRay = atoi(argv[++i]);
		else if (!std::strcmp(argv[i], "-maxLightSamples") && i + 1 < argc) maxLight = atoi(argv[++i]);
		else if (!std::strcmp(argv[i], "-frames") && i + 1 < argc) frames = atoi(argv[++i]);
		else if (!std::strcmp(argv[i], "-screenshot") && i + 1 < argc) screenshot = argv[++i];
		else if (!std::strcmp(argv[i], "-radiance") && i + 1 < argc) radiance = argv[++i];
		else if (!std::strcmp(argv[i], "-resume") && i + 1 < argc) resume = argv[++i];
		else if (!std::strcmp(argv[i], "-checkpoint") && i + 1 < argc) checkpoint = argv[++i];
	}
	Fluid fluid;
	if (!fluid.Init(nullptr, width, height, grid)) {   // ThrowIfFailed(E_FAIL) in the reference (FluidX12.cpp:198-200)
		std::fprintf(stderr, "Fluid::Init failed: %s\n", fx_error_string(fluid.LastStatus()));
		return 1;
	}
	fluid.SetMaxSamples(maxRay, maxLight);
	if (resume && !fluid.LoadCheckpoint(resume)) { std::fprintf(stderr, "cannot resume from %s: %s\n", resume, fx_error_string(fluid.LastStatus())); return 1; }
	LightProbe probe;                                   // FluidX12.cpp:189-195, 205-210: load, TransformSH, SetSH
	if (radiance) {
		if (!probe.Init(fluid, radiance)) { std::fprintf(stderr, "cannot load %s as a BC6H_UF16 DDS cube map\n", radiance); return 1; }
		probe.TransformSH(fluid);
		fluid.SetSH(probe.GetSH());
		std::printf("light probe %s: SH L00 = (%.3f, %.3f, %.3f)\n", radiance, probe.GetSH()[0], probe.GetSH()[1], probe.GetSH()[2]);
	}
	const float eye[3] = { 4.0f, 16.0f, -40.0f }, at[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 };
	const XMFLOAT4X4 view = LookAtLH(eye, at, up);
	const XMFLOAT4X4 proj = PerspectiveFovLH(3.141592654f / 4.0f, width / (float)height, 1.0f, 1000.0f);
	const XMFLOAT3 eyePt = { eye[0], eye[1], eye[2] };
	const float timeStep = (grid.z > 1 ? 2.0f : 1.0f) / grid.y;    // FluidX12.cpp:266

	const auto t0 = std::chrono::steady_clock::now();
	for (uint32_t f = 0; f < frames; ++f) {
		const uint8_t frameIndex = f % Fluid::FrameCount;
		fluid.UpdateFrame(timeStep, frameIndex, view, proj, eyePt);
		fluid.Simulate(nullptr, frameIndex);
		const float clearColor[4] = { 0.2f, 0.2f, 0.2f, 0.0f };                    // FluidX12.cpp:471-472
		if (grid.z > 1) {
			fluid.ClearRenderTarget(nullptr, clearColor);
			if (radiance) probe.RenderEnvironment(fluid, nullptr, frameIndex);      // FluidX12.cpp:483: the sky first
			fluid.Render(nullptr, frameIndex, Fluid::OPTIMIZED);                    // marches + renderCube onto the target
		}
		if (fluid.LastStatus() != FX_OK) { std::fprintf(stderr, "frame %u: %s\n", f, fx_error_string(fluid.LastStatus())); return 1; }
	}
	if (fx_synchronize(fluid.Handle()) != FX_OK) return 1;
	if (checkpoint && !fluid.SaveCheckpoint(checkpoint)) { std::fprintf(stderr, "cannot write %s: %s\n", checkpoint, fx_error_string(fluid.LastStatus())); return 1; }
	if (screenshot && grid.z > 1) {
		std::vector<uint8_t> rgba;
		if (!fluid.ReadRenderTarget(rgba)) { std::fprintf(stderr, "read-back failed\n"); return 1; }
		FILE* fp = std::fopen(screenshot, "wb");
		if (!fp) { std::perror(screenshot); return 1; }
		const size_t nl = std::strlen(screenshot);
		if (nl > 4 && !std::strcmp(screenshot + nl - 4, ".png")) write_png(fp, rgba.data(), width, height);      // FluidX12.cpp:640-660 writes a PNG (stbi_write_png)
		else {
			std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
			for (size_t p = 0; p < (size_t)width * height; ++p) std::fwrite(&rgba[4 * p], 1, 3, fp);
		}
		std::fclose(fp);
	}
	const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	fx_frame_info fi;
	fx_get_frame_info(fluid.Handle(), &fi);
	std::printf("%u frames of %ux%ux%u in %.3f s (%.1f fps); cube LOD %u (%u^2), %u ray samples, mask 0x%x\n",
		frames, grid.x, grid.y, grid.z, s, frames / s, fi.cube_lod, fi.cube_size, fi.ray_samples, fi.visibility_mask);
	return 0;
}
