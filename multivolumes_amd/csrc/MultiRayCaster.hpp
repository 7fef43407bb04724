// MultiRayCaster.hpp -- header-only C++ mirror of the reference's hot-path owner
// (MultiVolumes/Content/MultiRayCaster.h:28-50) on top of the C ABI in include/multivolumes.h.
//
// A maintainer of the reference swaps the XUSG-recording class for this one: same method
// names, argument meaning and bool-on-failure convention (XUSG_N_RETURN, XUSG.h:12-15).  D3D12 objects in the
// signatures (command lists, descriptor tables, formats) are dropped; DirectXMath types become float arrays in
// the same row-vector / row-major convention.
#pragma once

#include "../../include/multivolumes.h"
#include "../../include/multivolumes_host.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace mvamd
{
class MultiRayCaster
{
public:
	static const uint8_t FrameCount = 3;	// MultiRayCaster.h:52

	MultiRayCaster() = default;
	~MultiRayCaster() { if (m_ctx) mv_destroy(m_ctx); }
	MultiRayCaster(const MultiRayCaster&) = delete;
	MultiRayCaster& operator=(const MultiRayCaster&) = delete;

	// Init (MultiRayCaster.cpp:79-166) + SetViewport (:226-241)
	bool Init(uint32_t gridSize, uint32_t lightGridSize, uint32_t numVolumes, uint32_t numVolumeSrcs,
		uint32_t width, uint32_t height, uint32_t shadowMapSize = 1024, int device = 0, void* hipStream = nullptr,
		uint32_t shardIndex = 0, uint32_t shardCount = 1, uint32_t flags = 0, uint32_t framesInFlight = 0, uint32_t framesPerLaunch = 0)
	{
		mv_config cfg;
		std::memset(&cfg, 0, sizeof(cfg));
		cfg.struct_size = sizeof(cfg);
		cfg.grid_size = gridSize; cfg.light_grid_size = lightGridSize;
		cfg.num_volumes = numVolumes; cfg.num_volume_srcs = numVolumeSrcs;
		cfg.viewport_w = width; cfg.viewport_h = height; cfg.shadow_map_size = shadowMapSize;
		cfg.device = device; cfg.stream = hipStream; cfg.shard_index = shardIndex; cfg.shard_count = shardCount; cfg.flags = flags; cfg.frames_in_flight = framesInFlight; cfg.frames_per_launch = framesPerLaunch;
		if (mv_create(&cfg, &m_ctx) != MV_OK) return false;
		m_numVolumes = numVolumes; m_width = width; m_height = height;
		m_volumeWorlds.assign((size_t)numVolumes * 12, 0.0f);
		m_cb.assign(MV_CB_PER_FRAME_BYTES, 0);
		m_perObject.assign((size_t)numVolumes * MV_PER_OBJECT_BYTES, 0);
		const float center[3] = { 0.0f, 0.0f, 0.0f };
		SetVolumesWorld(20.0f, center);	// MultiRayCaster.cpp:141
		return true;
	}

	// LoadVolumeData (MultiRayCaster.cpp:168-209): the DDS loader is out of scope; the caller hands over the texels.
	bool LoadVolumeData(uint32_t i, const uint16_t* rgba16f) { return mv_volume_upload_rgba16f(m_ctx, i, rgba16f) == MV_OK; }
	bool LoadVolumeDataR32F(uint32_t i, const float* a, uint32_t w, uint32_t h, uint32_t d) { return mv_volume_load_r32f(m_ctx, i, a, w, h, d) == MV_OK; }
	// InitVolumeData (MultiRayCaster.cpp:243-264)
	bool InitVolumeData(uint32_t i) { return mv_volume_init_procedural(m_ctx, i) == MV_OK; }

	// SetViewport (MultiRayCaster.cpp:226-241): re-creates what is W x H (composite image, RayCast layers, own depth map), keeps the rest
	bool SetViewport(uint32_t width, uint32_t height)
	{
		if (mv_set_viewport(m_ctx, width, height) != MV_OK) return false;
		m_width = width; m_height = height;
		return true;
	}

	// SetRenderTargets (MultiRayCaster.cpp:211-224): DEPTH_MAP / SHADOW_MAP as borrowed device pointers
	bool SetRenderTargets(const float* d_depthMap, const uint16_t* d_shadowMap)
	{
		return mv_set_depth_map_device(m_ctx, d_depthMap) == MV_OK && mv_set_shadow_map_device(m_ctx, d_shadowMap) == MV_OK;
	}
	bool SetRenderTargetsHost(const float* depthMap, const uint16_t* shadowMap)
	{
		return mv_set_depth_map(m_ctx, depthMap) == MV_OK && mv_set_shadow_map(m_ctx, shadowMap) == MV_OK;
	}

	void SetSH(const float* coeffSH27) { mv_set_sh(m_ctx, coeffSH27); }													// :266-269
	void SetMaxSamples(uint32_t maxRaySamples, uint32_t maxLightSamples) { mv_set_max_samples(m_ctx, maxRaySamples, maxLightSamples); }	// :271-275
	void SetVolumesWorld(float size, const float center[3]) { mvh_set_volumes_world(m_numVolumes, size, center, m_volumeWorlds.data()); }	// :277-295
	void SetVolumeWorld(uint32_t i, float size, const float pos[3]) { mvh_set_volume_world(size, pos, &m_volumeWorlds[(size_t)i * 12]); }	// :297-303
	void SetLight(const float pos[3], const float color[3], float intensity)												// :305-309
	{
		std::memcpy(m_lightPt, pos, sizeof(m_lightPt));
		m_lightColor[0] = color[0]; m_lightColor[1] = color[1]; m_lightColor[2] = color[2]; m_lightColor[3] = intensity;
	}
	void SetAmbient(const float color[3], float intensity)																// :311-314
	{
		m_ambient[0] = color[0]; m_ambient[1] = color[1]; m_ambient[2] = color[2]; m_ambient[3] = intensity;
	}

	// UpdateFrame (MultiRayCaster.cpp:316-353). viewProj: row-major, row vectors; shadowVP: as ObjectRenderer::GetShadowVP() (transposed).
	bool UpdateFrame(const float viewProj[16], const float shadowVP[16], const float eyePt[3])
	{
		mvh_update_frame(m_numVolumes, m_volumeWorlds.data(), viewProj, shadowVP, eyePt, (float)m_width, (float)m_height,
			m_lightPt, m_lightColor, m_ambient, m_frameIdx, m_cb.data(), m_perObject.data());
		return mv_update_frame(m_ctx, m_cb.data(), m_perObject.data()) == MV_OK;
	}

	// Render (MultiRayCaster.cpp:355-385), K-buffer OIT, non-work-graph branch
	bool Render(bool composite = true)
	{
		const bool ok = mv_render(m_ctx, composite ? 1 : 0) == MV_OK;
		++m_frameIdx;	// :384
		return ok;
	}

	// One frame of the multi-GPU loop (include/multivolumes.h mv_dist_frame; after mv_dist_init_rccl / mv_dist_init_callbacks on Context()):
	// begins this frame (upload .. view march) and finishes the previous one (exchange, composite); DistFlush finishes the last.
	bool DistFrame(const float viewProj[16], const float shadowVP[16], const float eyePt[3], bool composite = true)
	{
		mvh_update_frame(m_numVolumes, m_volumeWorlds.data(), viewProj, shadowVP, eyePt, (float)m_width, (float)m_height,
			m_lightPt, m_lightColor, m_ambient, m_frameIdx, m_cb.data(), m_perObject.data());
		const bool ok = mv_dist_frame(m_ctx, m_cb.data(), m_perObject.data(), composite ? 1 : 0) == MV_OK;
		++m_frameIdx;
		return ok;
	}
	bool DistFlush() { return mv_dist_flush(m_ctx) == MV_OK; }

	mv_ctx* Context() const { return m_ctx; }
	const char* LastError() const { return mv_last_error(m_ctx); }

protected:
	mv_ctx* m_ctx = nullptr;
	uint32_t m_numVolumes = 0, m_width = 0, m_height = 0, m_frameIdx = 0;
	float m_lightPt[3] = { 75.0f, 75.0f, -75.0f };			// MultiRayCaster.cpp:67
	float m_lightColor[4] = { 1.0f, 0.7f, 0.3f, 1.0f };		// :68
	float m_ambient[4] = { 0.0f, 0.3f, 1.0f, 0.4f };		// :69
	std::vector<float> m_volumeWorlds;
	std::vector<uint8_t> m_cb, m_perObject;
};
} // namespace mvamd
