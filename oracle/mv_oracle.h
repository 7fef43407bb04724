/*
 * mv_oracle.h -- CPU ORACLE for the cube-map-space ray-march path of StarsX/MultiVolumes.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or call it.  The product (multivolumes_amd/) never does.
 *
 * PARITY: the reference ships no tests or golden vectors and its sources cannot be built here (HLSL -> DXIL via a
 * Windows-only dxc.exe, D3D12 + closed XUSG.dll) -- but it ships its COMPILED shaders, and oracle/dxil runs those on the CPU
 * (oracle/_ref/libmv_dxref.so).  Cull, light march, view march and the composite's pixel shaders of this file are PINNED to them
 * (tests/test_oracle_vs_reference_dxil.py; vectors in tests/golden/dxil_small.npz); the fixed-function stages -- sampler, format
 * conversion, rasteriser -- are a stated contract.  This file is a line-for-line scalar fp32 restatement of the HLSL, also anchored by analytic
 * known-answer tests (tests/test_oracle_kat*.py) derived from the shader math.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/MultiVolumes/Content unless noted; Sh/ = Shaders/).
 *
 * Buffer formats (the reference's D3D12 resources, MultiRayCaster.cpp:99-126, 455-549):
 *   volume      RGBA16F  N^3      uint16[z][y][x][4]
 *   light map   R11G11B10F L^3    uint32[z][y][x]   (R bits 0-10, G 11-21, B 22-31)
 *   depth map   D32F/R32F W*H     float[y][x]       (1.0 = far)
 *   shadow map  D16_UNORM M*M     uint16[y][x]
 *   cube map    RGBA16F, 6 faces, 5 mips: uint16[mip-chain][face][y][x][4]   per volume
 *   cube depth  R32F   same shape: float[mip-chain][face][y][x]             per volume
 *   PerObject   224 B / volume, CBPerFrame 212 B (MultiRayCaster.cpp:23-41), VolumeDesc u32
 *   (MultiRayCaster.cpp:43-48), VolumeInfo 4 x u16 (MultiRayCaster.cpp:50-56).
 */
#ifndef MV_ORACLE_H
#define MV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVO_NUM_CUBE_MIP 5 /* SharedConsts.h:6 */
#define MVO_NUM_OIT_LAYERS 8 /* SharedConsts.h:7 */
#define MVO_PER_OBJECT_BYTES 224
#define MVO_CB_PER_FRAME_BYTES 212

/* counters returned by the march functions */
typedef struct mvo_stats
{
	uint64_t view_samples;   /* iterations of Sh/CSRayMarch.hlsl:117-153 that reach GetSample */
	uint64_t lit_samples;    /* of those, the ones with alpha > 0.01 (light-map fetch) */
	uint64_t rays;           /* texels of visible faces whose ray hit the box (they write) */
	uint64_t light_samples;  /* iterations of CastLightRay (Sh/RayMarch.hlsli:206-229) */
} mvo_stats;

/* ---- formats ---- */
uint16_t mvo_f32_to_f16(float f);            /* RNE */
float    mvo_f16_to_f32(uint16_t h);
uint32_t mvo_pack_r11g11b10f(float r, float g, float b); /* RNE, negative -> 0 */
void     mvo_unpack_r11g11b10f(uint32_t p, float rgb[3]);

/* texels per volume in a cube-map mip chain (6 faces x sum of (N>>m)^2) and offset of mip m */
uint64_t mvo_cube_chain_texels(uint32_t N);
uint64_t mvo_cube_mip_offset(uint32_t N, uint32_t mip);

/* ---- fixed-function sampler contract (a5) ---- */
void  mvo_sample_rgba16f(const uint16_t* tex, uint32_t N, const float uvw[3], const int32_t off[3], float out[4]);
void  mvo_sample_r11g11b10f(const uint32_t* tex, uint32_t L, const float uvw[3], float out[3]);
float mvo_sample_point_r32f(const float* tex, uint32_t W, uint32_t H, float u, float v);
float mvo_sample_r32f_3d(const float* src, uint32_t w, uint32_t h, uint32_t d, const float uvw[3]); /* trilinear, clamp */
float mvo_sample_cmp_d16(const uint16_t* tex, uint32_t M, float u, float v, float ref);

/* ---- device functions of Sh/RayMarch.hlsli exposed for known-answer tests ---- */
int   mvo_compute_ray_origin(float origin[3], const float dir[3]);                 /* :128-155 */
float mvo_compute_target_hit(const float origin[3], const float target[3], const float dir[3]); /* :160-165 */
float mvo_get_step(float dDensity, float transm, float density, float step);       /* :182-192 */
void  mvo_get_local_pos(float px, float py, uint32_t slice, uint32_t dims, float out[3]); /* Sh/CSRayMarch.hlsl:28-53 */
void  mvo_cast_light_ray(float* transm, const uint16_t* tex, uint32_t N, const float origin[3],
	const float dir[3], float stepScale, uint32_t numSamples, uint64_t* samples);      /* :197-230 */
void  mvo_eval_sh_irradiance(const float sh27[27], const float norm[3], float out[3]); /* XUSG/Shaders/SHIrradianceTypeless.hlsli:15-36 */

/* ---- volume producers (a6) ---- */
void mvo_init_grid_data(uint32_t N, uint16_t* rgba);                               /* Sh/CSInitGridData.hlsl:10-27 */
void mvo_r32f_to_rgba16f(const float* src, uint32_t w, uint32_t h, uint32_t d, uint32_t N, uint16_t* rgba); /* Sh/CSR32FToRGBA16F.hlsl:16-26 */

/* ---- a1: CSVolumeCull ---- */
/* volumeInfo: V x 4 u16 {MipLevel, SmpCount, MaskBits, VolTexId}; untouched for culled volumes.
 * Lists are written in ascending volume order (the reference's append order is scheduling dependent). */
void mvo_cull(uint32_t V, const void* perObject, const uint32_t* volumeDescs, const void* cbPerFrame,
	uint32_t maxRaySamples, uint16_t* volumeInfo, uint32_t* visibleVolumes, uint32_t* visibleCount,
	uint32_t* cubeMapVolumes, uint32_t* cubeMapCount);

/* ---- a2: CSRayMarchL.  Updates the light map of ONE volume, returns its id. ---- */
uint32_t mvo_march_light(uint32_t V, uint32_t N, uint32_t L, uint32_t M,
	const uint16_t* const* volumes /* S pointers */, const uint32_t* volumeDescs,
	const void* perObject, const void* cbPerFrame, const uint16_t* shadowMap,
	const float* sh27 /* NULL -> hasLightProbe = 0 */, const uint32_t* visibleVolumes, uint32_t visibleCount,
	uint32_t maxLightSamples, uint32_t* lightMaps /* V x L^3, only [volumeId] written */, mvo_stats* stats);

/* ---- a3: CSRayMarchV ---- */
void mvo_march_view(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H,
	const uint16_t* const* volumes, const uint32_t* lightMaps /* V x L^3 */,
	const float* depthMap, const void* perObject, const void* cbPerFrame,
	const uint16_t* volumeInfo, const uint32_t* cubeMapVolumes, uint32_t cubeMapCount,
	uint16_t* cubeMaps /* V x chain x 4 */, float* cubeDepths /* V x chain */, mvo_stats* stats);

/* ---- a7: K-buffer OIT composite (VSCube/PSDepthPeel/PSCube/PSResolveOIT) ---- */
void mvo_composite(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H,
	uint32_t y0, uint32_t y1 /* rows [y0,y1) */,
	const uint16_t* const* volumes, const uint32_t* lightMaps,
	const float* depthMap, const void* perObject, const void* cbPerFrame,
	const uint16_t* volumeInfo, const uint32_t* visibleVolumes, uint32_t visibleCount,
	const uint16_t* cubeMaps, const float* cubeDepths,
	uint16_t* outRGBA /* W x H x 4 halfs, rows outside [y0,y1) untouched */, mvo_stats* stats);
/* the same drawn over a W x H RGBA16F colour target (PREMULTIPLITED blend, MultiRayCaster.cpp:931): out = src + dst (1 - src.a); background may be NULL */
void mvo_composite_over(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H,
	uint32_t y0, uint32_t y1 /* rows [y0,y1) */,
	const uint16_t* const* volumes, const uint32_t* lightMaps,
	const float* depthMap, const void* perObject, const void* cbPerFrame,
	const uint16_t* volumeInfo, const uint32_t* visibleVolumes, uint32_t visibleCount,
	const uint16_t* cubeMaps, const float* cubeDepths,
	const uint16_t* background, uint16_t* outRGBA /* W x H x 4 halfs, rows outside [y0,y1) untouched */, mvo_stats* stats);

/* number of OpenMP threads the march functions will use (1 if built without OpenMP) */
/* composite building blocks, exported for the known-answer tests */
float mvo_unproject_z(float depth);                                                /* Sh/PSCube.hlsli:21-26 */
void  mvo_cube_cast(const uint16_t* cubeMap, const float* cubeDepth, uint32_t d, float depthPix,
	const float pos[3], const float rayDir[3], float out[4]);                       /* Sh/PSCube.hlsli:31-108 */
void  mvo_cube_face_coords(const float dir[3], uint32_t* face, float st[2]);       /* D3D TextureCube face selection */
uint64_t mvo_cube_texel_index(uint32_t face, int32_t i, int32_t j, uint32_t d);
void  mvo_cube_footprint(const float dir[3], uint32_t d, uint64_t idx[4], float weights[4]); /* Gather order + bilinear weights */
/* every fragment (one per visible cube the pixel's ray leaves) of pixel (px, py): the analytic stand-in for VSCube + rasteriser */
uint32_t mvo_pixel_fragments(uint32_t W, uint32_t H, uint32_t px, uint32_t py, const void* perObject, const void* cbPerFrame,
	const uint32_t* visibleVolumes, uint32_t visibleCount, uint32_t* volume, float* lpt3, float* z);    /* Gather tap -> texel, across face edges */
/* optional modes, off by default: filter weights snapped to `weightBits` fractional bits (0 = exact fp32); the view march as
 * Sh/LibRayMarch.hlsl:205-287 states it (no ComputeTargetHit clamp) */
void  mvo_set_options(int weightBits, int libRayMarch);

int mvo_num_threads(void);
void mvo_set_num_threads(int n);
void mvo_set_light_block_range(uint32_t lo, uint32_t hi);	/* mvo_march_light marches the 4x4x4 voxel blocks [lo, hi) only (x fastest); default all */
void mvo_parallel_copy(void* dst, const void* src, size_t bytes);	/* first-touch spread of a freshly allocated array over the sockets' memory */

#ifdef __cplusplus
}
#endif
#endif
