// Fluid.hpp -- C++ host-side mirror of the reference's operator classes over the C ABI
// (include/fluidx_hip.h).  Same method names, argument meaning and error behaviour as
// /root/reference/FluidX12/Content/Fluid.h:20-35 and Content/LightProbe.h:16-26, so a caller written
// against the reference (FluidX12/FluidX12.cpp:197-201, 277, 465, 484-500) ports by dropping the XUSG
// arguments: CommandList* -> hipStream_t (void*), descriptor-table lib / uploaders / formats -> gone.
// Header-only; links against libfluidx_hip.so.
#pragma once
#include "../../include/fluidx_hip.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace fluidx {

struct XMUINT3 { uint32_t x, y, z; };
struct XMFLOAT3 { float x, y, z; };
struct XMFLOAT4X4 { float m[4][4]; };      // row-major, row-vector convention (DirectXMath)

struct FluidOptions {                       // knobs the reference fixes at compile time / in FluidX12.cpp
	uint32_t storage = FX_STORAGE_FP16;            // the reference stores RGBA16F (Fluid.cpp:207,213)
	uint32_t jacobiIters = 64;                     // ITER (CSProject3D.hlsl:13)
	uint32_t jacobiMode = FX_JACOBI_FAITHFUL;      // per-cell early-out (CSPoisson.hlsli:24)
	uint32_t advectAddress = FX_ADDRESS_CLAMP;     // FluidEZ.cpp:406 (Fluid.cpp:452 uses MIRROR)
	int32_t device = -1;
};

class Fluid {
public:
	typedef FluidOptions Options;
	enum RenderFlags : uint8_t {           // Fluid.h:12-18
		RAY_MARCH_DIRECT = 0,
		RAY_MARCH_CUBEMAP = (1 << 0),
		SEPARATE_LIGHT_PASS = (1 << 1),
		OPTIMIZED = RAY_MARCH_CUBEMAP | SEPARATE_LIGHT_PASS
	};
	static const uint8_t FrameCount = FX_FRAME_COUNT;   // Fluid.h:35

	Fluid() : m_ctx(nullptr) {}
	virtual ~Fluid() { if (m_ctx) fx_destroy(m_ctx); }
	Fluid(const Fluid&) = delete;
	Fluid& operator=(const Fluid&) = delete;

	// Fluid::Init (Fluid.cpp:189-270): false on failure, like XUSG_N_RETURN(..., false)
	bool Init(void* /*pCommandList*/, uint32_t width, uint32_t height, const XMUINT3& gridSize, const Options& opt = Options())
	{
		if (m_ctx) { fx_destroy(m_ctx); m_ctx = nullptr; }
		fx_desc d;
		std::memset(&d, 0, sizeof d);
		d.struct_size = sizeof d;
		d.grid_x = gridSize.x; d.grid_y = gridSize.y; d.grid_z = gridSize.z;
		d.viewport_w = width; d.viewport_h = height;
		d.storage = opt.storage; d.jacobi_iters = opt.jacobiIters; d.jacobi_mode = opt.jacobiMode;
		d.advect_address = opt.advectAddress; d.device = opt.device;
		m_status = fx_create(&m_ctx, &d);
		return m_status == FX_OK;
	}
	void SetMaxSamples(uint32_t maxRaySamples, uint32_t maxLightSamples) { m_status = fx_set_max_samples(m_ctx, maxRaySamples, maxLightSamples); }
	void SetSH(const float* coeffSH /* 9 x float3, or nullptr */) { m_status = fx_set_sh(m_ctx, coeffSH); }
	void UpdateFrame(float timeStep, uint8_t frameIndex, const XMFLOAT4X4& view, const XMFLOAT4X4& proj, const XMFLOAT3& eyePt)
	{
		const float eye[3] = { eyePt.x, eyePt.y, eyePt.z };
		m_status = fx_update_frame(m_ctx, timeStep, frameIndex, &view.m[0][0], &proj.m[0][0], eye);
	}
	void Simulate(void* pCommandList /* hipStream_t */, uint8_t frameIndex) { m_status = fx_simulate(m_ctx, pCommandList, frameIndex); }
	// Fluid::Render (Fluid.cpp:412-446).  With a render target in use (ClearRenderTarget called at least once) the
	// cube path ends in renderCube like the reference's (Fluid.cpp:430); without one only the cube map is produced.
	void Render(void* pCommandList /* hipStream_t */, uint8_t frameIndex, uint8_t flags)
	{
		m_status = fx_render(m_ctx, pCommandList, frameIndex, flags);
		if (m_status == FX_OK && m_hasTarget && (flags & RAY_MARCH_CUBEMAP)) m_status = fx_render_cube(m_ctx, pCommandList, frameIndex);
	}
	// the caller's ClearRenderTargetView on the swap-chain target (FluidX12.cpp:471-472)
	void ClearRenderTarget(void* pCommandList, const float clearColor[4])
	{
		m_status = fx_clear_render_target(m_ctx, pCommandList, clearColor);
		m_hasTarget = m_status == FX_OK;
	}
	void UseRenderTarget() { m_hasTarget = true; }       // something (the sky pass) has drawn on the target: Render resolves onto it
	// read the RGBA8 target back (the reference's screen-shot path reads the back buffer, FluidX12.cpp:640-660)
	bool ReadRenderTarget(std::vector<uint8_t>& rgba)
	{
		rgba.resize(fx_field_bytes(m_ctx, FX_FIELD_TARGET));
		if (rgba.empty()) return false;
		m_status = fx_download(m_ctx, FX_FIELD_TARGET, rgba.data(), rgba.size());
		return m_status == FX_OK;
	}

	// not in the reference (its state dies with the window): whole-grid state files, see fx_checkpoint_save
	bool SaveCheckpoint(const char* path) { m_status = fx_checkpoint_save(m_ctx, path); return m_status == FX_OK; }
	bool LoadCheckpoint(const char* path) { m_status = fx_checkpoint_load(m_ctx, path); return m_status == FX_OK; }

	// not in the reference: the void methods above cannot report failure there either (debug layer only)
	// waits for everything this object enqueued; false (LastStatus() == FX_E_HALO) if a multi-GPU step's advection left its halo:
	// the reference's void methods cannot report that, so a caller that renders or stores fields checks here (or LastStatus())
	bool Synchronize() { m_status = fx_synchronize(m_ctx); return m_status == FX_OK; }
	int LastStatus() const { return m_status; }
	fx_ctx* Handle() const { return m_ctx; }

protected:
	fx_ctx* m_ctx;
	int m_status = FX_OK;
	bool m_hasTarget = false;
};

// SH side of class LightProbe (LightProbe.h:16-26); the DDS loader and the sky pass are out of scope
class LightProbe {
public:
	bool Init(const float* radianceCube /* float[6][n][n][3] */, uint32_t n)
	{
		if (!radianceCube || !n) return false;
		m_cube.assign(radianceCube, radianceCube + (size_t)6 * n * n * 3);
		m_n = n;
		return true;
	}
	// LightProbe::Init(..., fileName) (LightProbe.cpp:41-46): a DDS cube map in BC6H_UF16 such as Bin/Assets/rnl_cross.dds
	bool Init(Fluid& fluid, const char* fileName, uint32_t mip = 0)
	{
		FILE* fp = std::fopen(fileName, "rb");
		if (!fp) return false;
		std::vector<uint8_t> dds;
		uint8_t buf[65536];
		for (size_t got; (got = std::fread(buf, 1, sizeof buf, fp)) > 0;) dds.insert(dds.end(), buf, buf + got);
		std::fclose(fp);
		uint32_t size = 0, mips = 0;
		m_status = fx_dds_cube_info(dds.data(), dds.size(), &size, &mips);
		if (m_status != FX_OK || mip >= mips) return false;
		m_n = (size >> mip) ? size >> mip : 1;
		m_cube.resize((size_t)6 * m_n * m_n * 3);
		m_status = fx_dds_decode_cube(fluid.Handle(), dds.data(), dds.size(), mip, m_cube.data(), m_cube.size());
		return m_status == FX_OK;
	}
	// LightProbe::RenderEnvironment (LightProbe.cpp:85-97): the radiance cube as the sky behind the volume.  The first call
	// uploads the cube; draw it BEFORE Fluid::Render like the demo does (FluidX12.cpp:483)
	void RenderEnvironment(Fluid& fluid, void* pCommandList, uint8_t frameIndex)
	{
		if (!m_envSet) { m_status = fx_set_environment(fluid.Handle(), m_cube.data(), m_n); m_envSet = m_status == FX_OK; }
		if (m_envSet) m_status = fx_render_environment(fluid.Handle(), pCommandList, frameIndex);
		fluid.UseRenderTarget();
	}
	void TransformSH(Fluid& fluid) { m_status = fx_sh_transform(fluid.Handle(), m_cube.data(), m_n, m_sh); }   // LightProbeEZ.cpp:117-123
	const float* GetSH() const { return m_sh; }
	int LastStatus() const { return m_status; }
private:
	std::vector<float> m_cube;
	uint32_t m_n = 0;
	float m_sh[27] = {};
	int m_status = FX_OK;
	bool m_envSet = false;
};

}  // namespace fluidx
