/*
 * multivolumes_host.h -- host-side helpers that reproduce what the reference's C++ does AROUND the hot path:
 * DirectXMath-convention matrices, MultiRayCaster::SetVolumesWorld / UpdateFrame, ObjectRenderer's shadow
 * view-projection, and the deterministic synthetic inputs of SURVEY.md 8(d).  Pure CPU code (no HIP calls);
 * it produces the byte blobs the C ABI in multivolumes.h consumes.
 *
 * Matrices are DirectXMath style: row vectors (v' = v * M), row-major float[16].
 */
#ifndef MULTIVOLUMES_HOST_H
#define MULTIVOLUMES_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* XMMatrixLookAtLH / XMMatrixPerspectiveFovLH / XMMatrixOrthographicLH (call sites MultiVolumes.cpp:264-279, ObjectRenderer.cpp:176-185) */
void mvh_look_at_lh(const float eye[3], const float focus[3], const float up[3], float out[16]);
void mvh_perspective_fov_lh(float fovY, float aspect, float zn, float zf, float out[16]);
void mvh_orthographic_lh(float w, float h, float zn, float zf, float out[16]);
void mvh_multiply(const float a[16], const float b[16], float out[16]);
int mvh_inverse(const float m[16], float out[16]); /* 0 on success, -1 if singular (computed in double) */

/* MultiRayCaster::SetVolumeWorld / SetVolumesWorld (MultiRayCaster.cpp:277-303): XMFLOAT3X4 as stored, 12 floats per volume.
 * Volumes the reference's integer-division grid does not place keep an all-zero matrix, like the reference's value-initialised vector. */
void mvh_set_volume_world(float size, const float pos[3], float world[12]);
void mvh_set_volumes_world(uint32_t V, float size, const float center[3], float* worlds);

/* ObjectRenderer::UpdateFrame's light view-projection (ObjectRenderer.cpp:176-185), stored transposed like m_shadowVP. */
void mvh_shadow_view_proj(const float lightPos[3], float sceneSize, float outT[16]);

/* MultiRayCaster::UpdateFrame (MultiRayCaster.cpp:316-353): fills the 212-byte CBPerFrame and V x 224-byte PerObject. */
void mvh_update_frame(uint32_t V, const float* worlds, const float viewProj[16], const float shadowVPT[16],
	const float eye[3], float viewportW, float viewportH, const float lightPos[3], const float lightColor[4],
	const float ambient[4], uint32_t frameIdx, void* cbPerFrame, void* perObject);

/* Synthetic inputs.  Source k = 0: the CSInitGridData field; k >= 1: that field with alpha scaled by a
 * 3-octave value noise in [0.25, 1] seeded by the k-th iterate of the LCG of Shaders/Random.hlsli:34-41 from 0x2545F491. */
void mvh_synth_volume(uint32_t N, uint32_t k, uint16_t* rgba);
/* D32F depth of an analytic sphere seen through viewProj (1.0 where the pixel misses) -- mesh stand-in. */
void mvh_synth_sphere_depth(uint32_t W, uint32_t H, const float viewProj[16], const float center[3], float radius, float* depth);
/* D16 depth of the same sphere through an orthographic light view-projection given TRANSPOSED (as mvh_shadow_view_proj returns it). */
void mvh_synth_sphere_shadow(uint32_t M, const float lightViewProjT[16], const float center[3], float radius, uint16_t* d16);

/* Wavefront OBJ positions + triangles (the subset of XUSG/Optional/XUSGObjLoader.cpp the depth passes need): `v x y z`,
 * `f a b c ...` with a, a/b, a//c or a/b/c references, negative (relative) indices, polygons fan-triangulated.
 * Arrays are malloc'ed; release with mvh_free.  Returns 0, or -1 if the file cannot be read / has no faces. */
int mvh_load_obj(const char* path, float** positions, uint32_t* numVerts, uint32_t** indices, uint32_t* numTris, float aabbMin[3], float aabbMax[3]);
void mvh_free(void* p);
/* Volume texture of a .dds file as float32 [depth][height][width] (mip 0 only): what DDS::Loader hands CSR32FToRGBA16F as g_txSrc
 * (MultiRayCaster.cpp:168-180; the shader samples .x, CSR32FToRGBA16F.hlsl:16-26).  Reads DDS_HEADER volumes (DDSD_DEPTH /
 * DDSCAPS2_VOLUME) and DX10-extended 3D textures; channel 0 of R32_FLOAT, R16_FLOAT, R8_UNORM, R16_UNORM, R32G32B32A32_FLOAT,
 * R16G16B16A16_FLOAT, R8G8B8A8_UNORM (DX10 DXGI_FORMAT or the legacy D3DFMT four-CC / bit-mask equivalents).  malloc'ed; release
 * with mvh_free.  Returns 0, -1 unreadable / not a DDS, -2 not a volume texture, -3 unsupported pixel format, -4 truncated. */
int mvh_load_dds_volume(const char* path, float** texels, uint32_t* width, uint32_t* height, uint32_t* depth);
/* ObjectRenderer::SetWorld (ObjectRenderer.cpp:147-154, no rotation): Scale(scale) * Translate(pos), row-major 4x4 */
void mvh_mesh_world(float scale, const float pos[3], float world[16]);
/* transposed store of a row-major 4x4 (XMStoreFloat4x4(XMMatrixTranspose(M))) */
void mvh_transpose(const float m[16], float outT[16]);

#ifdef __cplusplus
}
#endif
#endif
