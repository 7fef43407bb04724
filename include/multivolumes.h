/*
 * multivolumes.h -- C ABI of the MI355X-native cube-map-space volume ray-march path.
 *
 * This is the drop-in boundary for the path StarsX/MultiVolumes records in
 *   MultiRayCaster::cullVolumes   (MultiVolumes/Content/MultiRayCaster.cpp:1249-1285)
 *   MultiRayCaster::rayMarchL     (MultiRayCaster.cpp:1299-1327)
 *   MultiRayCaster::rayMarchV     (MultiRayCaster.cpp:1329-1368)
 *   cubeDepthPeel/renderCube/resolveOIT (MultiRayCaster.cpp:1440-1633)   [K-buffer OIT composite]
 * i.e. "where the DX12 dispatch sits" (MultiRayCaster.cpp:1284, 1326, 1367).
 *
 * The reference has no FFI of its own: the path lives behind protected C++ methods that record
 * D3D12 work.  Each entry point below names the reference method (file:line) whose effect it
 * replaces.  Plain pointers and sizes only; 0 = success, negative = error; no exceptions cross
 * the ABI.  The caller owns every host pointer; the context owns device memory unless a
 * *_device entry point says the pointer is borrowed.
 *
 * Buffer layouts are the reference's (MultiRayCaster.cpp:23-56, 99-126, 455-549):
 *   volume      RGBA16F  N^3      uint16[z][y][x][4]   (x fastest)
 *   light map   R11G11B10F L^3    uint32[z][y][x]      (R bits 0-10, G 11-21, B 22-31)
 *   depth map   R32F W*H          float[y][x]          (1.0 = far plane)
 *   shadow map  D16_UNORM M*M     uint16[y][x]
 *   cube map    RGBA16F (N>>mip)^2 x 6 faces: uint16[face][y][x][4]; faces +X,-X,+Y,-Y,+Z,-Z
 *   cube depth  R32F, same shape: float[face][y][x]
 *   PerObject   224 B per volume  {float4x4 WorldViewProj(T), float4x4 WorldViewProjI(T), float3x4 WorldI, float3x4 World}
 *   CBPerFrame  212 B             {float4 EyePos, float4 Viewport, float4x4 ScreenToWorld(T), float4x4 ShadowViewProj(T),
 *                                  float4 LightPos, float4 LightColor, float4 Ambient, uint FrameIdx}
 *   VolumeDesc  u32               VolTexId:14 | NumMips:4 << 14 | CubeMapSize:14 << 18
 *   VolumeInfo  4 x u16           {MipLevel, SmpCount, MaskBits (bit f = face f, bit 15 = cube-map scheme), VolTexId}
 *
 * Threading: one context per GPU; all work of a context is ordered on one hipStream_t;
 * a context is not re-entrant; independent contexts may be driven from different threads.
 */
#ifndef MULTIVOLUMES_H
#define MULTIVOLUMES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_NUM_CUBE_MIP 5			/* SharedConsts.h:6  NUM_CUBE_MIP */
#define MV_NUM_OIT_LAYERS 8			/* SharedConsts.h:7  NUM_OIT_LAYERS */
#define MV_PER_OBJECT_BYTES 224		/* MultiRayCaster.cpp:35-41 */
#define MV_CB_PER_FRAME_BYTES 212	/* MultiRayCaster.cpp:23-33 */
#define MV_CUBEMAP_RAYMARCH_BIT 0x8000u	/* Common.hlsli:11 */
#define MV_MAX_SHARDS 1024u		/* mv_config.shard_count limit */
#define MV_MAX_VOLUMES 1024u		/* mv_config.num_volumes limit (the reference's scenes use 2..64, Bin/all64.bat); volume sources: 16383 (VolTexId:14) */

/* error codes */
#define MV_OK 0
#define MV_ERR_INVALID_ARG (-1)
#define MV_ERR_HIP (-2)			/* a HIP runtime call failed; see mv_last_error() */
#define MV_ERR_NO_DEVICE (-3)	/* no usable gfx950 device: the product has no CPU fallback */
#define MV_ERR_OUT_OF_MEMORY (-4)
#define MV_ERR_NOT_READY (-5)	/* an input the stage needs was never supplied */
#define MV_ERR_RANGE (-6)
#define MV_ERR_TIMEOUT (-7)		/* multi-GPU frame loop, peer-store transport: another rank did not reach a stage of a frame within the timeout */

/* mv_config.flags */
#define MV_FLAG_NONE 0u
#define MV_FLAG_FORCE_MIP0_ALL_FACES 1u	/* stress variant of SURVEY 8(d): cull forces mip 0, all 6 faces, cube-map scheme */
#define MV_FLAG_PIPELINE_FRAMES 2u			/* frames in flight (the reference keeps FrameCount = 3, MultiRayCaster.h:52): constant upload + cull run ahead on
										   an internal stream, and frame i's light march, view march and composite run on internal view stream i mod 3
										   and write output set i mod 3 (4 slots of cull outputs, 3 light buffers per volume).  Everything a frame writes
										   AND everything it leaves alone is identical to the one-stream schedule: cube faces a frame does not march are
										   first brought up to date from the previous frame's set, so every set reads like the reference's single,
										   never-cleared one.  Calls stay asynchronous; every readback / mv_get_stats / input setter / producer
										   synchronises first. */
#define MV_FLAG_SINGLE_VIEW_STREAM 4u		/* with MV_FLAG_PIPELINE_FRAMES: keep the view march, exchange and composite of EVERY frame on the context's
										   stream, with two output sets (only upload, cull and light march run ahead on internal streams) -- for callers that order their own
										   work, e.g. RCCL collectives, on that stream between mv_march_view and mv_composite */

#define MV_FLAG_LIST_ORDER 8u				/* march the tiles in list order instead of the cost-binned order (longest rays first).  The order only
										   decides which wavefront marches which ray and when: results are bit-identical either way (tests, A/B timing) */

#define MV_FLAG_RAY_QUEUE 16u				/* view march as the two-stage device-side pipeline (ray setup -> persistent march that refills finished lanes from
										   a ray queue: ballot compaction) instead of one wavefront per 8x8 tile.  Same results bit for bit; it pays when
										   a frame has many rays per lane of the GPU */

#define MV_FLAG_CULL_ON_VIEW_STREAM 32u	/* with MV_FLAG_PIPELINE_FRAMES (not MV_FLAG_SINGLE_VIEW_STREAM): no stream of its own for the constant upload and the cull --
											   frame i's go in front of it on view stream i mod C.  That frees one of the four hardware queues a HIP process has:
											   FOUR whole frames in flight (frames_in_flight = 4) without streams sharing a queue -- what a rank of the multi-GPU
											   loop wants, whose frame is a chain of short kernels (profiles/r04_peer_rank_cost.txt).  On one GPU it is worth 1 %
											   (C3: 0.313 against 0.317 ms per frame, bench.py's default there).  Results are identical. */

#define MV_FLAG_FINEGRAINED_PEER_BUFFERS 64u	/* the buffers other ranks store into (peer-store transport: light maps, cube maps / depths, frame buffers) are allocated
											   FINE-GRAINED (hipExtMallocWithFlags hipDeviceMallocFinegrained): memory the hardware keeps coherent between agents
											   inside running kernels, which is what the HSA memory model asks of memory another GPU writes while a kernel here
											   polls for it.  Plain (coarse-grained) allocations are the default: the folded waits make the acquire themselves and
											   the transport's self-test checks it (mv_dist_peer_coherence); this flag is the belt beside those braces, for a
											   node where that self-test fails.  What it costs on ONE GPU: profiles/r05_finegrained.txt.  mv_create fails with
											   MV_ERR_OUT_OF_MEMORY where the runtime refuses such allocations. */

typedef struct mv_ctx mv_ctx;

typedef struct mv_config
{
	uint32_t struct_size;		/* = sizeof(mv_config) */
	uint32_t grid_size;			/* N: volume edge and cube-map face edge (MultiRayCaster.cpp:103, 113, 478) */
	uint32_t light_grid_size;	/* L (MultiRayCaster.cpp:123) */
	uint32_t num_volumes;		/* V (MultiRayCaster.cpp:109-111) */
	uint32_t num_volume_srcs;	/* S (MultiRayCaster.cpp:98) */
	uint32_t viewport_w, viewport_h;	/* SetViewport (MultiRayCaster.cpp:226-241) */
	uint32_t shadow_map_size;	/* M (ObjectRenderer.cpp:80-83; 1024 in the reference) */
	int32_t device;				/* HIP device ordinal */
	uint32_t flags;				/* MV_FLAG_* */
	void* stream;				/* hipStream_t to order all work on; NULL = the context creates its own */
	uint32_t shard_index;		/* multi-GPU: this context marches volumes v with v % shard_count == shard_index */
	uint32_t shard_count;		/* 0 or 1 = unsharded; at most MV_MAX_SHARDS */
	uint32_t frames_in_flight;	/* with MV_FLAG_PIPELINE_FRAMES (and not MV_FLAG_SINGLE_VIEW_STREAM): whole frames in flight = view streams = output sets,
								   2..5; 0 = 3, the reference's FrameCount (MultiRayCaster.h:52).  More than 3 need more than the four hardware
								   queues a HIP process has by default: GPU_MAX_HW_QUEUES >= frames_in_flight + 1 in the environment before the HIP
								   runtime initialises (the library sets no environment variable: the application does, as bench.py does).
								   (ABI 4; a struct_size without this field is accepted and means 0.) */
	uint32_t frames_per_launch;	/* B = 1..4 (0 = 1), with MV_FLAG_PIPELINE_FRAMES (not MV_FLAG_SINGLE_VIEW_STREAM): B consecutive frames share a view
								   stream, so frames_in_flight x B (at most 16) frames are in flight, with as many output sets and light buffers per volume.
								   What it is for: mv_dist_frame with the peer-store transport then launches the B frames of a group TOGETHER -- one
								   light march, one view march, one composite, one row push for all of them (blockIdx.y = the frame) -- because a rank
								   of eight holds an eighth of a frame: a wavefront per SIMD or less, every long kernel as long as its longest ray
								   whatever else the GPU could do meanwhile, and four hardware queues are all the frames that can overlap.  The frames
								   of a group reach the GPU when the group is full (or at mv_dist_flush / a setter; mv_synchronize alone does not send them, mv_dist_shutdown drops them): throughput for latency.
								   Results are identical.  (ABI 5: the field occupies what was padding -- zero the struct before filling it in.) */
} mv_config;

/* counters accumulated since the last mv_reset_stats (all device-side, exact) */
typedef struct mv_stats
{
	uint64_t view_samples;		/* iterations of CSRayMarch.hlsl:117-153 that reach GetSample */
	uint64_t lit_samples;		/* of those, alpha > 0.01 (light-map fetch, CSRayMarch.hlsl:128-130) */
	uint64_t rays;				/* texels of visible faces whose ray hit the box */
	uint64_t light_samples;		/* iterations of CastLightRay (RayMarch.hlsli:206-229) */
	uint64_t composite_samples;	/* screen-space RayCast samples (RayCast.hlsli:62-102) */
	uint64_t reserved[3];		/* [0] view-march lane slots (64 x longest ray of every wavefront), [1] lit samples among composite_samples that were
								   marched inside the view-march launch, [2] view / RayCast samples counted above whose eight texel loads were
								   not issued because their brick's occupancy bit proved alpha <= ZERO_THRESHOLD (results are unaffected) */
	double ms_cull, ms_march_light, ms_march_view, ms_composite;	/* HIP-event time per stage when profiling is on */
	uint64_t n_cull, n_march_light, n_march_view, n_composite;		/* launches those sums cover */
} mv_stats;

const char* mv_error_string(int code);
const char* mv_last_error(const mv_ctx* ctx);	/* text of the last failure on this context ("" if none) */
const char* mv_get_warning(const mv_ctx* ctx);	/* what mv_create found worth saying although it succeeded ("" if nothing): today, a context that keeps more streams busy than
												   the process has hardware queues -- GPU_MAX_HW_QUEUES >= frames_in_flight + 1 must be in the environment BEFORE the HIP runtime
												   starts (the library sets no environment variable; bench.py sets 8); without it the frames share queues and wait for each other */
#define MV_ABI_VERSION 6u	/* (mv_config MUST be zeroed before it is filled in: frames_per_launch, ABI 5, took the place of padding, and a caller of ABI 4 that left garbage there gets MV_ERR_INVALID_ARG or B times the output sets) 6: the peer-store export blob carries a per-process nonce and a staging map per rank (blobs of ABI 5 are refused), MV_FLAG_FINEGRAINED_PEER_BUFFERS, MV_DIST_PEER_NO_ACQUIRE / _PUSH_CELLS, mv_dist_peer_coherence, mv_dist_peer_info, mv_set_latency_sampling / mv_get_latency, mv_dealt_tiles, mv_light_rebuild_range, mv_get_warning (round 6, additive: the queue warning of mv_create no longer sits in mv_last_error); 5: peer-store transport (mv_dist_peer_export, mv_dist_init_peer, MV_ERR_TIMEOUT, MV_DIST_PEER_BYTES 8192), mv_config.frames_per_launch, MV_FLAG_CULL_ON_VIEW_STREAM; 2: MV_DIST_ID_BYTES 384 (three communicators), mv_dist_shutdown, mv_dist_set_inplace_gather, MV_MAX_SHARDS; 3: mv_set_viewport, mv_dist_info, mv_dist_rccl_available, mv_host_wait_ms; 4: mv_config.frames_in_flight, MV_DIST_ID_BYTES 640 (up to five communicators), light march shared out over the ranks */
uint32_t mv_abi_version(void);	/* == MV_ABI_VERSION of the header the library was built from */

/* MultiRayCaster::Init (MultiRayCaster.cpp:79-166): creates volumes[S], lightMaps[V], cubeMaps/cubeDepths[V] (5 mips,
 * zero-initialised), VolumeDesc[V] = {i % S, 5, N} (:471-479), VolumeInfo[V], visible / cube-map lists + counters. */
int mv_create(const mv_config* cfg, mv_ctx** out);
void mv_destroy(mv_ctx* ctx);

/* Result of LoadVolumeData (MultiRayCaster.cpp:168-209): volume source `src` := N^3 RGBA16F texels from the host. */
int mv_volume_upload_rgba16f(mv_ctx* ctx, uint32_t src, const uint16_t* rgba);
/* Same, from a device pointer (copied, not borrowed). */
int mv_volume_upload_rgba16f_device(mv_ctx* ctx, uint32_t src, const void* d_rgba);
/* InitVolumeData (MultiRayCaster.cpp:243-264) = CSInitGridData.hlsl:10-27. */
int mv_volume_init_procedural(mv_ctx* ctx, uint32_t src);
/* LoadVolumeData's compute pass = CSR32FToRGBA16F.hlsl:16-26 on a w x h x d R32F host array. */
int mv_volume_load_r32f(mv_ctx* ctx, uint32_t src, const float* a, uint32_t w, uint32_t h, uint32_t d);
/* createVolumeInfoBuffers (MultiRayCaster.cpp:471-488): replace the packed VolumeDesc[V]. */
int mv_set_volume_descs(mv_ctx* ctx, const uint32_t* packed);

/* SetViewport (MultiRayCaster.cpp:226-241; from SetRenderTargets :211-224 and on a window resize, MultiVolumes.cpp:382): the reference
 * re-creates its K-buffers and keeps everything else.  Here: the composite image, the RayCast layers and the context's own depth
 * map are re-created for w x h (the depth map must be supplied again); volumes, light maps, cube maps and counters stay, and the next
 * frame (mv_update_frame) renders exactly what a context created with w x h and the same history of cube maps would.
 * Not inside the multi-GPU frame loop (mv_dist_shutdown first). */
int mv_set_viewport(mv_ctx* ctx, uint32_t w, uint32_t h);

/* SetRenderTargets (MultiRayCaster.cpp:211-224) hands the path DEPTH_MAP and SHADOW_MAP. Host copies.
 * These and every other input setter / producer below (volumes, VolumeDesc, SH, light-map upload, mesh rasteriser) wait for the frames
 * in flight BEFORE they write.  Inside the multi-GPU frame loop they first enqueue the pending tail of the last mv_dist_frame (its
 * exchange, composite and row gather: collectives), so there they are collective calls: every rank makes them between the same frames. */
int mv_set_depth_map(mv_ctx* ctx, const float* d32);		/* W x H */
int mv_set_shadow_map(mv_ctx* ctx, const uint16_t* d16);	/* M x M */
/* ...or borrowed device pointers, as the reference borrows ObjectRenderer's depth buffers (NULL = back to own copy).  A borrowed
 * buffer is read by every frame begun while it is set, until that frame's composite has run: with frames in flight, and inside
 * mv_dist_frame (whose composite of frame k is enqueued by call k + 1), keep it alive and unchanged for three more frames. */
int mv_set_depth_map_device(mv_ctx* ctx, const float* d_d32);
int mv_set_shadow_map_device(mv_ctx* ctx, const uint16_t* d_d16);
/* SetSH (MultiRayCaster.cpp:266-269): 9 x float3; NULL -> g_hasLightProbe = 0 (MultiRayCaster.cpp:1321). */
int mv_set_sh(mv_ctx* ctx, const float* sh27);
/* SetMaxSamples (MultiRayCaster.cpp:271-275). Defaults 256 / 96 (MultiRayCaster.cpp:65-66). */
int mv_set_max_samples(mv_ctx* ctx, uint32_t maxRaySamples, uint32_t maxLightSamples);
/* UpdateFrame (MultiRayCaster.cpp:316-353): the 212-byte CBPerFrame and V x 224-byte PerObject it writes. */
int mv_update_frame(mv_ctx* ctx, const void* cbPerFrame, const void* perObject);

/* ---- mesh depth / shadow producer: what ObjectRenderer hands the path through SetRenderTargets ----
 * ObjectRenderer::RenderShadow / Render (ObjectRenderer.cpp:209-262, VSDepth.hlsl:24-27, VSBasePass.hlsl:44-60) rasterise the
 * mesh into SHADOW_MAP (D16, M x M) and DEPTH_MAP (D32F, viewport).  Here a gfx950 kernel rasterises the indexed triangle
 * list straight into the context's own depth / shadow maps (LESS test, cleared to 1.0; 1/256-pixel snapping, top-left rule).
 * Triangles with a vertex at or behind the eye plane (w <= 0) are dropped: the mesh is assumed in front of the camera. */
int mv_mesh_upload(mv_ctx* ctx, const float* positions /* numVerts x 3 */, uint32_t numVerts, const uint32_t* indices /* numTris x 3 */, uint32_t numTris);
/* worldViewProjT: the transposed 4x4 the reference stores in cbPerObject (ObjectRenderer.cpp:195-199); projBias: the Halton
 * TAA jitter of VSBasePass.hlsl:57 (NULL = none).  Makes the context's own depth map current (like mv_set_depth_map). */
int mv_mesh_render_depth(mv_ctx* ctx, const float worldViewProjT[16], const float projBias[2]);
/* shadowWVPT: world * lightViewProj, transposed (ObjectRenderer.cpp:186).  Makes the own shadow map current. */
int mv_mesh_render_shadow(mv_ctx* ctx, const float shadowWVPT[16]);
int mv_readback_depth_map(mv_ctx* ctx, float* dst /* W x H */);
int mv_readback_shadow_map(mv_ctx* ctx, uint16_t* dst /* M x M */);

/* cullVolumes (MultiRayCaster.cpp:1249-1285) = CSVolumeCull.hlsl:13-78. Lists come out in ascending volume order.
 * The three marches below need THIS frame's cull (the frame that the last mv_update_frame began): MV_ERR_NOT_READY otherwise. */
int mv_cull(mv_ctx* ctx);
/* rayMarchL (MultiRayCaster.cpp:1299-1327) = CSRayMarchL.hlsl:20-121: relights ONE volume, visible[FrameIdx % count].
 * On sharded contexts only the shard that owns that volume marches (the others return at once): the light maps evolve frame
 * by frame exactly as on one GPU.  A frame may leave it out (the reference then simply keeps its light maps): the frame -- whatever else it
 * does, mv_cull alone included -- and the frames after it read the maps that were last written (round 6: a frame with no stage that reads
 * light maps used to leave the relit volume's entry on an unwritten buffer; profiles/r06_api_fuzz.txt).  It must come before the frame's
 * mv_march_view / mv_composite. */
int mv_march_light(mv_ctx* ctx);
/* rayMarchV (MultiRayCaster.cpp:1329-1368) = CSRayMarch.hlsl:77-158 with _LIGHT_PASS_ (CSRayMarchV.hlsl). */
int mv_march_view(mv_ctx* ctx);
/* cubeDepthPeel + renderCube + resolveOIT (MultiRayCaster.cpp:1440-1492, 1544-1580, 1615-1633) over screen rows
 * [row0, row1) of the W x H RGBA16F output (row1 = 0 -> H). Background = transparent black (see mv_composite_over).
 * A frame may composite without mv_march_view (or march without compositing, or do neither): the composite then shows the cube maps
 * as the last marches left them and ray-casts screen-space-scheme volumes itself, as the reference's passes would; with
 * MV_FLAG_PIPELINE_FRAMES such a frame first drains the frames in flight. */
int mv_composite(mv_ctx* ctx, uint32_t row0, uint32_t row1);
/* The same drawn OVER an existing colour target, as the reference's resolve pass is (PREMULTIPLITED blend preset, MultiRayCaster.cpp:931,
 * 1615-1633): out = src + dst * (1 - src.a) on all four channels.  d_background: W x H RGBA16F texels on the device (borrowed, only read),
 * e.g. the scene the mesh renderer produced; NULL = transparent black = mv_composite. */
int mv_composite_over(mv_ctx* ctx, uint32_t row0, uint32_t row1, const void* d_background);
/* Render (MultiRayCaster.cpp:355-385), non-work-graph branch: cull -> marchL -> marchV [-> composite]; FrameIdx += 1 is the caller's job. */
int mv_render(mv_ctx* ctx, int with_composite);

/* CPU readback of the buffers the path emits (the reference's only "ReadBack" notion: XUSG.h:1614, 1860). */
int mv_readback_volume_info(mv_ctx* ctx, uint16_t* dst /* V x 4 */);
/* Everything the cull produced in ONE host wait (what a multi-GPU driver needs per frame): VolumeInfo[V], both lists with their
 * counts, and the minimum selected mip over the visible volumes (the exchange slot size).  Waits for the cull only: with frames
 * in flight the marches of earlier frames keep running.  It also brings back this frame's light-map table, so that
 * mv_get_buffer(MV_BUF_LIGHT_MAP) afterwards needs no host wait (the context's stream waits for the frame's light march). */
int mv_readback_cull(mv_ctx* ctx, uint16_t* volume_info /* V x 4 */, uint32_t* visible /* V */, uint32_t* visibleCount,
	uint32_t* cubeMapVolumes /* V */, uint32_t* cubeMapCount, uint32_t* minMip);
int mv_readback_visible_lists(mv_ctx* ctx, uint32_t* visible /* V */, uint32_t* visibleCount,
	uint32_t* cubeMapVolumes /* V */, uint32_t* cubeMapCount);
int mv_readback_volume(mv_ctx* ctx, uint32_t src, uint16_t* dst /* N^3 x 4 */);
int mv_readback_light_map(mv_ctx* ctx, uint32_t volume, uint32_t* dst /* L^3 */);
int mv_upload_light_map(mv_ctx* ctx, uint32_t volume, const uint32_t* src /* L^3 */);	/* test/exchange hook */
int mv_readback_cube_map(mv_ctx* ctx, uint32_t volume, uint32_t mip, uint16_t* dst /* 6 x d x d x 4 */);
int mv_readback_cube_depth(mv_ctx* ctx, uint32_t volume, uint32_t mip, float* dst /* 6 x d x d */);
/* (Contexts with frames in flight rotate output sets.  mv_readback_cube_map / _cube_depth / _composite read the CURRENT frame's set: what they return is
 * defined behind a frame that marched / composited -- the set is then complete, every face the frame did not march brought forward -- and equals the
 * one-stream context's; behind a frame that did neither it is that set's older content.) */
int mv_readback_composite(mv_ctx* ctx, uint16_t* dst /* W x H x 4, rows in screen order (also inside the multi-GPU frame loop, where the device buffer MV_BUF_COMPOSITE keeps them rank-major by 8-row tile row: see mv_dist_frame) */);
int mv_clear_cube_maps(mv_ctx* ctx);	/* zero every cube map / cube depth (creation state) */

int mv_get_stats(mv_ctx* ctx, mv_stats* out);	/* synchronises the stream */
int mv_reset_stats(mv_ctx* ctx);
int mv_set_profiling(mv_ctx* ctx, int on);		/* record HIP events around every stage launch (adds host syncs in mv_get_stats only) */
int mv_synchronize(mv_ctx* ctx);
/* Time (ms) mv_update_frame has spent waiting for the GPU to free a frame slot (four frames of constants are in flight at most, as the
 * reference waits on its fence with FrameCount = 3, MultiVolumes.cpp MoveToNextFrame): 0 while the host is the slower side. */
int mv_host_wait_ms(mv_ctx* ctx, double* ms, int reset);
/* Frame latency, sampled: every n-th frame (0 = off, the default) the library notes the host's clock when the frame's constants are handed
 * over (mv_update_frame / mv_dist_frame) and again -- a host function behind the frame's last kernel on its stream -- when the GPU is done
 * with it.  With frames in flight (and several frames per launch) that is the latency a frame pays for the throughput: depth x period.
 * mv_get_latency drains the frames in flight first. */
int mv_set_latency_sampling(mv_ctx* ctx, uint32_t every_n_frames);
int mv_get_latency(mv_ctx* ctx, double* mean_ms, double* max_ms, uint64_t* samples, int reset);

/* Raw handles for callers that keep data resident (torch / RCCL plumbing): */
void* mv_get_stream(mv_ctx* ctx);			/* the hipStream_t */
/* device pointers + byte sizes of context-owned buffers; `which` = MV_BUF_* */
#define MV_BUF_VOLUME 0			/* index = src */
#define MV_BUF_LIGHT_MAP 1		/* index = volume: the buffer this frame reads (device layout: L^3 cells of 16 B) */
#define MV_BUF_CUBE_MAP 2		/* index = volume: whole 5-mip chain, uint16[chain][4] */
#define MV_BUF_CUBE_DEPTH 3		/* index = volume */
#define MV_BUF_VOLUME_INFO 4
#define MV_BUF_COMPOSITE 5
#define MV_BUF_DEPTH_MAP 6
#define MV_BUF_SHADOW_MAP 7
#define MV_BUF_TILE_COST 9		/* the view march's tile cost table (one for all frame slots; index < frame slots accepted), uint32 per key: longest ray's samples + 1, 0 = never marched (diagnostics) */
#define MV_BUF_VOLUME_FLAGS 10	/* the cull's per-volume word of the current frame slot, uint32: bit 0 visible, bit 1 cube-map scheme, bits 2-4 mip, bits 5-10 marched faces, bits 16-31 the rank that marches the volume's cube map (sharded contexts; v % shard_count, or the cost deal inside mv_dist_frame) */
#define MV_BUF_EXCHANGE_OFFSETS 11	/* uint32 per volume: texel offset of its marched faces inside its owner's shard of the frame's cube-face all-gather (sharded contexts) */
#define MV_BUF_MARCH_TIMELINE 8	/* 8 x uint64 per march wavefront; only diagnostic builds (-DMV_TIMELINE) fill it (scripts/march_timeline.py) */
int mv_get_buffer(mv_ctx* ctx, uint32_t which, uint32_t index, void** d_ptr, size_t* bytes);

/* texels (not bytes) of one volume's 5-mip cube chain and the offset of mip m inside it */
uint64_t mv_cube_chain_texels(uint32_t N);
uint64_t mv_cube_mip_offset(uint32_t N, uint32_t mip);

/* ---- multi-GPU exchange: volumes shard v % shard_count; cube faces are all-gathered ---- */
/* Bytes one rank contributes per frame: (V / shard_count) slots x slot_bytes, slot = 6 x d x d x (8 + 4) bytes with
 * d = N >> min selected mip over all visible volumes (identical on every rank because cull is replicated).
 * mv_exchange_layout synchronises the stream; pack / unpack take slot_dim from it (or from mv_readback_cull) and do not. */
int mv_exchange_layout(mv_ctx* ctx, uint32_t* slot_dim, size_t* slot_bytes, size_t* shard_bytes);
/* Screen rows [row0, row1) of this shard (whole 8-row tiles split evenly over shard_count; the whole screen when unsharded).
 * The shard composites these rows (mv_composite(ctx, row0, row1)), and its view march also marches, for these rows only, the
 * screen-space RayCast (Sh/RayCast.hlsli:43-108) of EVERY volume on that scheme -- so the light maps of such volumes owned by
 * other shards must be current (mv_upload_light_map / a device-to-device copy into MV_BUF_LIGHT_MAP) before mv_march_view. */
int mv_shard_rows(mv_ctx* ctx, uint32_t* row0, uint32_t* row1);
/* Pack this shard's selected-mip cube faces (colour then depth per slot) into d_dst (device, shard_bytes). */
int mv_exchange_pack(mv_ctx* ctx, void* d_dst, uint32_t slot_dim);
/* Scatter an all-gathered buffer (shard_count x shard_bytes, rank-major) back into the cube maps / depths of every volume. */
int mv_exchange_unpack(mv_ctx* ctx, const void* d_src, uint32_t slot_dim);

/* ---- multi-GPU frame loop inside the library: one process per GPU, context r of `shard_count` = rank r ----
 * mv_dist_frame runs the whole per-frame protocol without Python and without a host wait on work of the current frame:
 *   frame k:   constant upload, cull; the light march of the ONE volume relit this frame (CSRayMarchL.hlsl:29-33) SHARED OUT over the ranks
 *              by 4x4x4 voxel blocks -- every rank holds every volume and marches 1 / shard_count of the blocks of the march's dispatch
 *              order -- and the shares all-gathered as plain R11G11B10F words (4 L^3 bytes in all), so that every rank keeps every light
 *              map current (every rank marches its own screen rows of every volume on the screen-space scheme); the view march of the
 *              cube-map volumes the cull dealt to this rank and of its rows of the screen-space volumes;
 *   frame k-1: pack the marched faces of this rank's cube-map volumes -> all-gather -> unpack -> composite of this rank's rows ->
 *              all-gather of the row bands.
 * The size of frame k-1's cube-face all-gather comes from its cull, which left it in pinned host memory a call ago (no copy, no event).
 * The composite of frame k is therefore complete after the NEXT mv_dist_frame (or mv_dist_flush).  Results equal the single-GPU frame
 * bit for bit.
 * Inside the loop the replicated cull deals the frame's work: cube-map volumes to ranks by estimated march cost (not v % shard_count),
 * screen rows by 8-row tile row t % shard_count (not mv_shard_rows' contiguous band).  The composite image is stored rank-major --
 * tile row t at block t % shard_count, row t / shard_count, blocks of ceil(tile rows / shard_count) tile rows -- so that the row gather
 * is one in-place all-gather; mv_readback_composite returns screen order.
 * Two schedules, chosen by the context's flags:
 *   MV_FLAG_PIPELINE_FRAMES                               whole frames on C view streams ("chains"; C = mv_config.frames_in_flight, 3 by
 *       default): everything of frame k, its collectives included, goes on view stream k % C with communicator k % C -- frames overlap
 *       as wholes, which a rank needs once its share of a frame no longer fills the GPU (the default of bench.py);
 *   MV_FLAG_PIPELINE_FRAMES | MV_FLAG_SINGLE_VIEW_STREAM  the view marches on one stream, frame k-1's exchange + composite on a second
 *       one beside frame k's march; communicator 0 = cube faces / rows, 1 = light maps.
 * Input setters called between two mv_dist_frame calls first enqueue the pending tail of the last frame (collectives: every rank must
 * make the same calls at the same point), so that frame is composited against the inputs it was begun with.
 * Transport: RCCL (the library opens librccl.so itself; one communicator per chain, two without chains) or caller-supplied collectives
 * (tests, other fabrics). */
#define MV_DIST_ID_BYTES 640	/* five RCCL unique ids (one per communicator; the loop uses as many as the context has frames in flight) */
int mv_dist_rccl_available(void);	/* 1 if librccl.so and every entry point the loop needs can be opened in this process: a LOCAL check, so that the ranks can
									   agree on the transport before any of them enters mv_dist_init_rccl (ncclCommInitRank is collective: a rank that
									   cannot open the library would leave the others waiting inside it) */
int mv_dist_unique_id(void* out /* MV_DIST_ID_BYTES; call on rank 0, hand the bytes to every rank */);
int mv_dist_init_rccl(mv_ctx* ctx, const void* id /* MV_DIST_ID_BYTES, the same on every rank */);
/* all-gather: every rank contributes `bytes` from `send`; `recv` receives shard_count x bytes in rank order.  all-reduce: in-place
 * integer sum of `count` uint32 (the frame loop itself only gathers since ABI 4; the self-test still checks it).  `stream` is the
 * hipStream_t the collective must be ordered on; which = 0 .. frames_in_flight - 1 names the communicator: collectives with different
 * `which` run on different streams at the same time and need separate communicators (chains: `which` = frame % frames_in_flight;
 * single view stream: 0 = cube faces / rows, 1 = light maps).  Return 0 on success. */
typedef int (*mv_dist_allgather_fn)(void* user, const void* send, void* recv, size_t bytes, void* stream, int which);
typedef int (*mv_dist_allreduce_u32_fn)(void* user, void* buf, size_t count, void* stream, int which);
int mv_dist_init_callbacks(mv_ctx* ctx, mv_dist_allgather_fn allgather, mv_dist_allreduce_u32_fn allreduce, void* user);
/* ---- peer-store transport ("7 simultaneous direct peer writes" over xGMI) ----
 * No collectives at all: every rank maps every rank's light maps, cube maps / depths and frame buffers (hipIpc handles between the
 * processes of one node; plain pointers between contexts of one process) and the kernels that PRODUCE a frame's data store it into all
 * of them -- the light march its share of the relit map, the view march its cube-face texels, the composite its screen rows.  A
 * collective's ordering is replaced by one flag per (rank, stage, frame chain) in uncached device memory, raised behind the producing
 * kernel and waited for in front of the consuming one (csrc/mv_peer.hip): a rank's frame is light share -> view march -> composite with
 * no pack, all-gather or unpack between them.  Results equal the single-GPU frame bit for bit, like the other transports'.
 *   1. every rank: mv_dist_peer_export(ctx, blob)           -- its handles (MV_DIST_PEER_BYTES);
 *   2. the caller gathers the blobs of all ranks, rank order -- any bootstrap: torch.distributed, MPI, a file;
 *   3. every rank: mv_dist_init_peer(ctx, all_blobs, 0)      -- maps the others; collective in effect (a barrier between 2 and the first frame is implied by the gather).
 * Needs a context created with MV_FLAG_PIPELINE_FRAMES (whole-frame chains; not MV_FLAG_SINGLE_VIEW_STREAM), contexts of identical
 * configuration and history (frames begun so far) on every rank, at most 64 ranks.  Every wait has a timeout (default 30 s,
 * mv_dist_set_peer_timeout_ms): a rank that is left waiting returns MV_ERR_TIMEOUT from its next mv_dist_frame / mv_dist_flush /
 * mv_synchronize instead of hanging.
 * MV_DIST_PEER_LOOPBACK: a rank measured ALONE (examples/dist_rank_cost.cpp): every other rank is a set of stand-in buffers on the same GPU
 * (about 1 GB each at C3) and the rank raises the other ranks' flags itself -- the kernels, their eight-fold stores and the schedule are a
 * rank's real ones, the images are not. */
#define MV_DIST_PEER_BYTES 8192
#define MV_DIST_PEER_LOOPBACK 1u
/* The flags are raised and waited for INSIDE the light march, the view march and the composite ("folded": a rank's chain is those three
 * kernels and nothing else) when every rank owns its GPU -- the library decides from the ranks' PCI bus ids -- and by one-wavefront kernels
 * of their own when ranks share a GPU (tests), where spinning consumers could crowd the producers they wait for out of the GPU.  These
 * two force the choice (tests of the folded form on one GPU with two or three ranks; A/B timing): */
#define MV_DIST_PEER_FOLD 2u
#define MV_DIST_PEER_NO_FOLD 4u
/* The ranks push their screen rows to rank 0 only ("gather tiles to rank 0"): rank 0 holds the whole frame, the other ranks'
 * frame buffers their own rows -- 1/7 of the row traffic (C5 on 8 ranks: 4 MB per rank and frame instead of 28).  Default: every rank
 * gets every row, as with the all-gathers. */
#define MV_DIST_PEER_ROWS_TO_ROOT 8u
/* Who marches what.  With several frames per launch (mv_config.frames_per_launch > 1) the view march of EVERY cube-map volume is
 * shared out by tiles (rank r marches the groups g of 8x8 tiles of each marched face with (g + volume) mod shard_count == r; a group is
 * two whole rows of tiles, or one, where that gives every rank equally many groups of the face, else one tile: mv_dealt_tiles): a march that stores
 * its texels into every rank's cube maps itself does not care who "owns" a volume, and eight ranks then carry an eighth of the tiles each
 * -- ten visible volumes dealt to eight ranks leave two ranks with two volumes, which a launch of four frames then waits for (0.080 against
 * 0.060 ms per C3 frame).  With one frame per launch whole volumes go to ranks, dealt by estimated cost as with the collectives (the north
 * star's "volumes shard across the GPUs"): there the frame is a chain of latencies, and a rank that touches one or two volumes runs it
 * faster (0.095 against 0.103 ms) than one that touches a slice of all of them.  These two flags force either deal
 * (profiles/r04_frames_per_launch.txt).  Results are identical. */
#define MV_DIST_PEER_DEAL_VOLUMES 16u
#define MV_DIST_PEER_DEAL_TILES 32u
/* The deal's rule as a function (tests, tools): the tiles of a cube face of tiles_x x tiles_x tiles of volume `volume` that rank shard_index of
 * shard_count marches, in its marching order, into tiles[0 .. capacity); returns how many there are.  Over the ranks: a partition of the face. */
uint32_t mv_dealt_tiles(uint32_t tiles_x, uint32_t shard_index, uint32_t shard_count, uint32_t volume, uint32_t* tiles, uint32_t capacity);
/* The light share-out's rule as a function (tests, tools; the arithmetic peerRebuildLightRole runs on the device): the z slices [out[0], out[1])
 * of the relit light map that rank shard_index of shard_count marches and pushes -- a slab of ceil(B / shard_count) adjacent layers of four slices,
 * B = ceil(L / 4); EMPTY, at L, for a rank beyond the last layer -- and the cells it rebuilds from the other ranks' texels: out[4] of them, cell k
 * being cell k of the map for k < out[2], else cell k + out[3].  out[3] + out[4] == L^3.  MV_ERR_INVALID_ARG outside 2 <= L <= 640, 1 <= shard_count <= 64. */
int mv_light_rebuild_range(uint32_t light_grid_size, uint32_t shard_index, uint32_t shard_count, uint32_t out[5]);
/* A folded wait ends with a system-scope ACQUIRE by the wavefront that polled (`buffer_inv sc0 sc1`: invalidate, no write-back) in front of
 * the workgroup's plain loads of what the flags announce -- the default.  MV_DIST_PEER_NO_ACQUIRE leaves it out (round 4's form: it rested
 * on the rotation of the buffers alone), for A/B timing. */
#define MV_DIST_PEER_NO_ACQUIRE 64u
/* A frame's relit light map travels as PLAIN R11G11B10F texels (4 bytes; the device keeps light maps as quad-replicated cells of 16) into a
 * staging map per frame slot on every rank, and every rank rebuilds the cells of the other ranks' layers itself -- workgroups of the view-march
 * launch between the ones that push and the ones that march (waits as kernels: a kernel of its own): a rank of eight sends 3.1 MB of light per
 * C3 frame instead of 12.4.  The default.  MV_DIST_PEER_PUSH_CELLS pushes whole cells (round 4), for A/B timing. */
#define MV_DIST_PEER_PUSH_CELLS 128u
int mv_dist_peer_export(mv_ctx* ctx, void* out /* MV_DIST_PEER_BYTES */);
int mv_dist_init_peer(mv_ctx* ctx, const void* all /* shard_count x MV_DIST_PEER_BYTES in rank order; NULL with MV_DIST_PEER_LOOPBACK */, uint32_t flags);
int mv_dist_set_peer_timeout_ms(mv_ctx* ctx, uint32_t ms);
/* The in-kernel coherence test by itself (collective: every rank calls it, after mv_dist_init_peer): ONE launch per rank reads every
 * rank's region of a buffer, waits INSIDE the kernel until every rank has, overwrites its region of every rank's buffer, waits inside the
 * kernel for everybody's stores, ends that wait with an acquire (acquire != 0) or not, and reads again with plain loads.
 * stale_words: words of the second read that still showed the old pattern (0 = what the folded waits need); first_read_bad: words of
 * the first read that did not show the old pattern (0 unless the mappings themselves are broken).  mv_dist_selftest runs it, with the
 * context's acquire setting, whenever the waits are folded, and fails on a stale word. */
int mv_dist_peer_coherence(mv_ctx* ctx, uint32_t acquire, uint32_t* stale_words, uint32_t* first_read_bad);
/* Every frame the loop checks, through four words behind each rank's share of the light exchange, that all ranks' replicated culls
 * decided the same (a caller that passed different frame constants on different ranks gets MV_ERR_INVALID_ARG instead of mismatched
 * collectives).  On by default.  Off is for a rank driven ALONE with collectives that deliver nothing (examples/dist_rank_cost.cpp,
 * scripts/exp_rank_cost.py: what one rank of N costs without a fabric) -- the other ranks' words are then not there to compare. */
int mv_dist_set_cull_crosscheck(mv_ctx* ctx, int enabled);
/* Caller-supplied collectives only: 1 = the all-gather may be called in place (send == recv + rank * bytes, NCCL's convention); the
 * row bands of the composite are then gathered straight into the frame buffer instead of through staging copies.  Default 0. */
int mv_dist_set_inplace_gather(mv_ctx* ctx, int enabled);
int mv_dist_shutdown(mv_ctx* ctx);	/* detach the frame loop (drains it first); mv_dist_init_* may then be called again, e.g. with the other transport. Collective in effect: every rank calls it */
int mv_dist_selftest(mv_ctx* ctx);	/* one all-gather (plain and in place) and one all-reduce per communicator, verified: call once after init on every rank */
int mv_dist_frame(mv_ctx* ctx, const void* cbPerFrame, const void* perObject, int with_composite);
int mv_dist_flush(mv_ctx* ctx);	/* enqueue the exchange + composite of the last frame begun (then mv_synchronize / readbacks see it) */
int mv_dist_exchanged_bytes(mv_ctx* ctx, uint64_t* bytes, int reset);	/* bytes this rank received through the collectives (peer stores: what the other ranks stored into its buffers, from the frames' own figures) */
/* peer-store transport: out[8] = { waits folded, folded waits acquire, rows to rank 0 only, whole volumes per rank, bytes pushed per texel of a
 * rank's share of the relit light map, kind of memory the flag block lives in (2 uncached, 1 fine-grained, 0 plain), stale words and
 * wrong first reads of the last in-kernel coherence test }; all zero on the other transports */
int mv_dist_peer_info(mv_ctx* ctx, uint32_t* out);
/* transport: 0 none, 1 RCCL, 2 caller-supplied collectives, 3 peer stores; ranks: the size of the communicator as RCCL reports it (ncclCommCount), or
 * shard_count for caller-supplied collectives; communicators: how many the loop keeps busy (3 with whole-frame chains, else 2) */
int mv_dist_info(mv_ctx* ctx, uint32_t* transport, uint32_t* ranks, uint32_t* communicators);

#ifdef __cplusplus
}
#endif
#endif
