// mv_api.cpp -- the C-ABI of include/multivolumes.h: context, device memory, stream-ordered stage launches.
//
// Host-side counterpart of MultiRayCaster (MultiVolumes/Content/MultiRayCaster.{h,cpp}): where the reference records
// D3D12 dispatches into an XUSG command list, this layer launches the gfx950 kernels of mv_kernels.hip on one HIP stream.
// Kernel order on the stream = the reference's barrier graph (cull -> marchL -> marchV -> composite).
// There is no CPU fallback: without a gfx950 device every entry point fails.
#include "mv_ctx.hpp"
using namespace mvi;

extern "C" {

const char* mv_error_string(int code)
{
	switch (code)
	{
	case MV_OK: return "ok";
	case MV_ERR_INVALID_ARG: return "invalid argument";
	case MV_ERR_HIP: return "HIP runtime error";
	case MV_ERR_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
	case MV_ERR_OUT_OF_MEMORY: return "out of device memory";
	case MV_ERR_NOT_READY: return "a required input was never supplied";
	case MV_ERR_TIMEOUT: return "another rank did not reach a stage of a frame within the timeout";
	case MV_ERR_RANGE: return "index out of range";
	default: return "unknown error";
	}
}

const char* mv_last_error(const mv_ctx* ctx) { return ctx ? ctx->lastError.c_str() : ""; }
const char* mv_get_warning(const mv_ctx* ctx) { return ctx ? ctx->warning.c_str() : ""; }
uint32_t mv_abi_version(void) { return MV_ABI_VERSION; }
uint64_t mv_cube_chain_texels(uint32_t N) { return mv::cubeChainTexels(N); }
uint64_t mv_cube_mip_offset(uint32_t N, uint32_t mip) { return mv::cubeMipOffset(N, mip); }

int mv_create(const mv_config* cfg, mv_ctx** out)
{
	// (struct_size names the caller's version of mv_config: the one without frames_in_flight, ABI <= 3's, is accepted too)
	constexpr uint32_t kConfigV1Size = offsetof(mv_config, frames_in_flight);
	if (!cfg || !out || (cfg->struct_size != sizeof(mv_config) && cfg->struct_size != kConfigV1Size && cfg->struct_size != ((kConfigV1Size + 7u) & ~7u))) return MV_ERR_INVALID_ARG;
	*out = nullptr;
	const uint32_t wantChains = cfg->struct_size == sizeof(mv_config) && cfg->frames_in_flight ? cfg->frames_in_flight : 3u;
	if (wantChains < 2u || wantChains > kMaxChains) return MV_ERR_INVALID_ARG;
	const uint32_t perLaunch = cfg->struct_size == sizeof(mv_config) && cfg->frames_per_launch ? cfg->frames_per_launch : 1u;
	if (perLaunch > mv::kMaxBatch || wantChains * perLaunch > kMaxFramesInFlight) return MV_ERR_INVALID_ARG;
	if (perLaunch > 1u && (!(cfg->flags & MV_FLAG_PIPELINE_FRAMES) || (cfg->flags & MV_FLAG_SINGLE_VIEW_STREAM))) return MV_ERR_INVALID_ARG;	// groups of frames share a view stream: whole-frame chains only
	if (cfg->grid_size < 2 || cfg->grid_size > 640 /* bricked alpha cells (16 B/texel) must stay < 4 GiB for 32-bit byte offsets */ || cfg->light_grid_size < 2 || cfg->light_grid_size > 640 /* 32-bit offsets into the 16 B/texel light quads */ || cfg->num_volumes < 1 || cfg->num_volumes > MV_MAX_VOLUMES /* the cull and the tile order keep per-volume tables in LDS (12 V and 48 V + 132 bytes of the 64 KiB a launch gets by default) */ ||
		cfg->num_volume_srcs < 1 || cfg->num_volume_srcs > 16383 || cfg->viewport_w < 1 || cfg->viewport_h < 1 || cfg->shadow_map_size < 1)
		return MV_ERR_INVALID_ARG;
	if (cfg->shard_count > 1 && cfg->shard_index >= cfg->shard_count) return MV_ERR_INVALID_ARG;
	if (cfg->shard_count > MV_MAX_SHARDS) return MV_ERR_INVALID_ARG;	// owner ids travel in 16 bits of a word, rows are dealt through a 32-bit reciprocal

	int nDev = 0;
	if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0 || cfg->device < 0 || cfg->device >= nDev) return MV_ERR_NO_DEVICE;
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return MV_ERR_NO_DEVICE;
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MV_ERR_NO_DEVICE;	// kernels are built for gfx950 only
	if (hipSetDevice(cfg->device) != hipSuccess) return MV_ERR_NO_DEVICE;

	mv_ctx* ctx = new (std::nothrow) mv_ctx();
	if (!ctx) return MV_ERR_OUT_OF_MEMORY;
	memset(&ctx->cfg, 0, sizeof(ctx->cfg));
	memcpy(&ctx->cfg, cfg, cfg->struct_size < sizeof(mv_config) ? cfg->struct_size : sizeof(mv_config));
	memset(&ctx->st, 0, sizeof(ctx->st));

	int rc = MV_OK;
	auto bail = [&](int code) { *out = nullptr; mv_destroy(ctx); return code; };

	if (cfg->stream) ctx->stream = reinterpret_cast<hipStream_t>(cfg->stream);
	else
	{
		if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) return bail(MV_ERR_HIP);
		ctx->ownStream = true;
	}

	ctx->streamA = ctx->stream;
	ctx->streamL = ctx->stream;
	for (uint32_t i = 0; i < kMaxOutSets; ++i) ctx->streamB[i] = ctx->stream;
	if (cfg->flags & MV_FLAG_PIPELINE_FRAMES)
	{
		// highest priority: stream A's short kernels (cull = one workgroup, light march) must not queue behind the
		// thousands of view-march workgroups of the previous frame, or the overlap is lost
		int prLow = 0, prHigh = 0;
		(void)hipDeviceGetStreamPriorityRange(&prLow, &prHigh);
		// MV_FLAG_CULL_ON_VIEW_STREAM: no stream A -- every frame's upload and cull go in front of it on its own view stream (mv_update_frame),
		// which leaves a hardware queue for a fourth chain
		ctx->cullOnChain = (cfg->flags & MV_FLAG_CULL_ON_VIEW_STREAM) && !(cfg->flags & MV_FLAG_SINGLE_VIEW_STREAM);
		if (!ctx->cullOnChain)
		{
			if (hipStreamCreateWithPriority(&ctx->streamA, hipStreamNonBlocking, prHigh) != hipSuccess) return bail(MV_ERR_HIP);
			ctx->ownStreamA = true;
		}
		// One view stream (MV_FLAG_SINGLE_VIEW_STREAM, the multi-GPU loop): the light marches get a stream of their own and two
		// output sets rotate.  Otherwise C output sets on C view streams (mv_config.frames_in_flight, default 3), each frame's light
		// march on its view stream.  A HIP process has four hardware queues by default, and streams that share one wait for each
		// other: with three chains that is four streams; more chains need GPU_MAX_HW_QUEUES >= C + 1 in the environment before the HIP
		// runtime starts -- the CALLER's job (bench.py does it): the library sets no environment variable.  A context that asks for more streams
		// than the process has hardware queues still works, its frames share queues and wait for each other (profiles/r03_frames_in_flight.txt);
		// mv_get_warning says so after mv_create.
		if (cfg->flags & MV_FLAG_SINGLE_VIEW_STREAM)
		{
			if (hipStreamCreateWithPriority(&ctx->streamL, hipStreamNonBlocking, prHigh) != hipSuccess) return bail(MV_ERR_HIP);
			ctx->ownStreamL = true;
			ctx->numOutSets = 2u;
		}
		else
		{
			ctx->lightOnViewStream = true;
			ctx->numOutSets = wantChains * perLaunch;
			ctx->framesPerLaunch = perLaunch;
			ctx->batchPos = perLaunch - 1u;	// (the first frame opens group 1)
		}
#ifdef MV_DIAGNOSTICS	// A/B measurements only (csrc/build.sh -DMV_DIAGNOSTICS); the shipped library reads no environment variable
		{ const char* e = getenv("MV_OUT_SETS"); if (e) ctx->numOutSets = (uint32_t)atoi(e); if (ctx->numOutSets < 2u || ctx->numOutSets > kMaxOutSets) ctx->numOutSets = 2u; }
#endif
		if (!(cfg->flags & MV_FLAG_SINGLE_VIEW_STREAM))
		{
			ctx->numViewStreams = ctx->numOutSets / ctx->framesPerLaunch;
			for (uint32_t i = 1; i < ctx->numViewStreams; ++i)
				if (hipStreamCreateWithFlags(&ctx->streamB[i], hipStreamNonBlocking) != hipSuccess) return bail(MV_ERR_HIP);
			ctx->ownStreamB1 = true;
		}
	}
	ctx->st.lightBufs = ctx->numOutSets > 3u ? ctx->numOutSets : 3u;
#ifndef MV_EXTRA_FRAME_SLOTS
#define MV_EXTRA_FRAME_SLOTS 0	// (A/B: more slots of constants + cull outputs than light buffers + 1, so that upload and cull may run further ahead of the frames in flight)
#endif
	ctx->numSlots = ctx->st.lightBufs + 1u + MV_EXTRA_FRAME_SLOTS;
	if (ctx->numSlots > kMaxFrameSlots) ctx->numSlots = kMaxFrameSlots;
	for (uint32_t i = 0; i < ctx->numSlots; ++i)
		if (hipEventCreateWithFlags(&ctx->evStageA[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->evMainDone[i], hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->evLight[i], hipEventDisableTiming) != hipSuccess) return bail(MV_ERR_HIP);
	if (hipEventCreateWithFlags(&ctx->evProducers, hipEventDisableTiming) != hipSuccess) return bail(MV_ERR_HIP);
	for (uint32_t i = 0; i < ctx->numSlots; ++i)
		if (hipEventCreateWithFlags(&ctx->evCopy[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->evLanded[i], hipEventDisableTiming) != hipSuccess) return bail(MV_ERR_HIP);

	DeviceState& st = ctx->st;
	st.N = cfg->grid_size; st.bricks = (cfg->grid_size + 3u) >> 2; st.L = cfg->light_grid_size;
	st.fN = (float)st.N; st.fNm1 = (float)st.N - 1.0f; st.fL = (float)st.L; st.fLp1 = (float)st.L + 1.0f;
	st.V = cfg->num_volumes; st.S = cfg->num_volume_srcs; st.W = cfg->viewport_w; st.H = cfg->viewport_h; st.M = cfg->shadow_map_size;
	st.flags = cfg->flags; st.shardIndex = cfg->shard_count > 1 ? cfg->shard_index : 0u; st.shardCount = cfg->shard_count > 1 ? cfg->shard_count : 1u;
	st.shardMagic = st.shardCount > 1u ? (uint32_t)(((1ull << 32) + st.shardCount - 1u) / st.shardCount) : 0xffffffffu;
	st.maxRaySamples = 256; st.maxLightSamples = 96;	// MultiRayCaster.cpp:65-66
	st.hasLightProbe = 0;

	const size_t brickedBytes = mv::volBrickCount(st.bricks) * 64 * sizeof(uint2);	// padded brick pitches (mv_device.hpp)
	const size_t chain = mv::cubeChainTexels(st.N);
	const size_t L3 = (size_t)st.L * st.L * st.L;

	// volumes[S] (MultiRayCaster.cpp:98-105)
	ctx->volumes.resize(st.S, nullptr);
	st.volStride = (brickedBytes + 511u) & ~(size_t)511u;
	if ((rc = devAlloc(ctx, &ctx->volPool, (size_t)st.S * st.volStride)) != MV_OK) return bail(rc);
	for (uint32_t i = 0; i < st.S; ++i) ctx->volumes[i] = reinterpret_cast<uint2*>(ctx->volPool + (size_t)i * st.volStride);
	st.volPool = ctx->volPool;
	if ((rc = devAlloc(ctx, &ctx->dVolumePtrs, st.S * sizeof(uint2*), false)) != MV_OK) return bail(rc);
	if (hipMemcpyAsync(ctx->dVolumePtrs, ctx->volumes.data(), st.S * sizeof(uint2*), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	st.volumes = ctx->dVolumePtrs;
	// alpha-cell companions (mv_device.hpp, VolumeView): 16 bytes per texel, bricked like the texels
	ctx->cells.resize(st.S, nullptr);
	for (uint32_t i = 0; i < st.S; ++i)
		if ((rc = devAlloc(ctx, &ctx->cells[i], 2 * brickedBytes)) != MV_OK) return bail(rc);
	if ((rc = devAlloc(ctx, &ctx->dCellPtrs, st.S * sizeof(uint4*), false)) != MV_OK) return bail(rc);
	if (hipMemcpyAsync(ctx->dCellPtrs, ctx->cells.data(), st.S * sizeof(uint4*), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	st.alphaCells = ctx->dCellPtrs;
	{
		const size_t regions = mv::occupancyRegions(st.bricks);
		ctx->occupancy.resize(st.S, nullptr);
		st.occStride = (uint32_t)(regions * regions * regions);
		if ((rc = devAlloc(ctx, &ctx->occPool, (size_t)st.S * st.occStride * sizeof(uint64_t))) != MV_OK) return bail(rc);
		for (uint32_t i = 0; i < st.S; ++i) ctx->occupancy[i] = ctx->occPool + (size_t)i * st.occStride;
		st.occPool = ctx->occPool;
		if ((rc = devAlloc(ctx, &ctx->dOccPtrs, st.S * sizeof(uint64_t*), false)) != MV_OK) return bail(rc);
		if (hipMemcpyAsync(ctx->dOccPtrs, ctx->occupancy.data(), st.S * sizeof(uint64_t*), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
		if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
		st.occupancy = ctx->dOccPtrs;
	}

	ctx->stagingBytes = (size_t)st.N * st.N * st.N * sizeof(uint2);
	if ((rc = devAlloc(ctx, &ctx->staging, ctx->stagingBytes, false)) != MV_OK) return bail(rc);

	// lightMaps[V], cubeMaps[V] (5 mips), cubeDepths[V] (MultiRayCaster.cpp:107-126), all zero-initialised
	#ifdef MV_EMULATE_LIGHT_PAIRS	// (timing experiment only, csrc/experiments/mv_march_tiles_r05.hip: room for 32-byte light cells; the images of such a build are garbage)
	if ((rc = devAlloc(ctx, &st.lightBuffers, 2u * (size_t)st.V * st.lightBufs * L3 * sizeof(uint4))) != MV_OK) return bail(rc);
#else
	if ((rc = devAllocShared(ctx, &st.lightBuffers, (size_t)st.V * st.lightBufs * L3 * sizeof(uint4))) != MV_OK) return bail(rc);
#endif
	if ((rc = devAlloc(ctx, &ctx->lightStaging, L3 * sizeof(uint32_t), false)) != MV_OK) return bail(rc);
	if ((rc = devAlloc(ctx, &ctx->ownDepth, (size_t)st.W * st.H * sizeof(float))) != MV_OK) return bail(rc);
	if ((rc = devAlloc(ctx, &ctx->ownShadow, (size_t)st.M * st.M * sizeof(uint16_t))) != MV_OK) return bail(rc);
	st.depthMap = ctx->ownDepth; st.shadowMap = ctx->ownShadow;

	// createVolumeInfoBuffers (MultiRayCaster.cpp:455-549)
	if ((rc = devAlloc(ctx, &ctx->dDescs, (size_t)st.V * sizeof(uint32_t))) != MV_OK) return bail(rc);
	if ((rc = devAlloc(ctx, &ctx->dSh, 27 * sizeof(float))) != MV_OK) return bail(rc);
	// RayCast layers of screen-space-scheme volumes are marched inside the view-march launch into fp32 RGBA layers; volumes
	// beyond the slot capacity fall back to marching inside the composite.
	viewportDerived(st);
	// constants + cull outputs, once per frame slot (FrameBufs)
	{
		std::vector<uint4*> table(st.V);
		for (uint32_t v = 0; v < st.V; ++v) table[v] = st.lightBuffers + (size_t)(st.lightBufs * v) * L3;
		for (uint32_t p = 0; p < ctx->numSlots; ++p)
		{
			FrameBufs& f = ctx->fb[p];
			// one upload: CBPerFrame, then PerObject[V] -- the slots' copies side by side in ONE allocation, like their pinned slices (the
			// multi-GPU loop with several frames per launch uploads a group's constants with one copy: mv_dist.cpp batchClose)
			if (p == 0)
			{
				uint8_t* all = nullptr;
				if ((rc = devAlloc(ctx, &all, (256 + (size_t)st.V * MV_PER_OBJECT_BYTES) * ctx->numSlots)) != MV_OK) return bail(rc);
				ctx->dCbAll = all;
			}
			f.dCb = ctx->dCbAll + (256 + (size_t)st.V * MV_PER_OBJECT_BYTES) * p;
			f.dPerObject = static_cast<uint8_t*>(f.dCb) + 256;
			// what mv_readback_cull brings back sits in ONE block, in the order of its pinned copy: counters (64 B), VolumeInfo[V],
			// visible[V], cube-map list[V], light table[V]
			{
				uint8_t* block = nullptr;
				if ((rc = devAlloc(ctx, &block, 64 + (size_t)st.V * 24)) != MV_OK) return bail(rc);
				f.cullBlock = block;
				f.counters = reinterpret_cast<uint32_t*>(block);	// kNumCounters read back + the cull's ticket at [8]
				f.volumeInfo = reinterpret_cast<uint16_t*>(block + 64);
				f.visibleVolumes = reinterpret_cast<uint32_t*>(block + 64 + (size_t)st.V * 8);
				f.cubeMapVolumes = reinterpret_cast<uint32_t*>(block + 64 + (size_t)st.V * 12);
				f.lightTable = reinterpret_cast<uint4**>(block + 64 + (size_t)st.V * 16);
			}
			if ((rc = devAlloc(ctx, &f.ownedVisible, (size_t)st.V * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.batchInfo, 8 * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.screenSlot, (size_t)st.V * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.exchangeOff, (size_t)st.V * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.workEntries, ((size_t)st.V * 6 + st.screenSlots) * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.workTileStart, ((size_t)st.V * 6 + st.screenSlots + 1) * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.cubeOffset, (size_t)st.V * sizeof(uint64_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.volumeFlags, (size_t)st.V * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.volumeRects, (size_t)st.V * 4 * sizeof(uint16_t))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.lightDirs, (size_t)st.V * sizeof(float4))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.lightOrder, ((size_t)((st.L + 3u) >> 2) * ((st.L + 3u) >> 2) * ((st.L + 3u) >> 2) + 4u) * sizeof(uint32_t))) != MV_OK) return bail(rc);	// + the direction the order was made for
			if ((rc = devAlloc(ctx, &f.volConsts, (size_t)st.V * sizeof(mv::VolumeConsts))) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.queueHeads, (size_t)mv::kRayQueues * mv::kRayQueueStride * sizeof(uint32_t))) != MV_OK) return bail(rc);
			if (hipMemcpyAsync(f.lightTable, table.data(), (size_t)st.V * sizeof(uint4*), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
		}
		if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	}
	if ((rc = devAlloc(ctx, &st.stats, kNumStats * sizeof(uint64_t))) != MV_OK) return bail(rc);
	st.maxTiles = mv::marchViewMaxTiles(st.V, st.N, st.shardCount);
	// RayRecord.out is a 31-bit element index into the cube chains / the screen layers
	if ((uint64_t)st.V * mv::cubeChainTexels(st.N) >= (1ull << 31) || (uint64_t)st.screenSlots * st.W * st.H >= (1ull << 31)) return bail(MV_ERR_INVALID_ARG);
	// The march is a persistent launch: as many wavefronts as the device holds at once.  With frames in flight it leaves a share of the
	// wave slots to the kernels of the other streams (the next frame's cull and light march, the previous frame's composite), which
	// could not start before its tail otherwise.
	{
		const uint32_t resident = mv::marchViewResidentWaves(cfg->device, (uint64_t)st.S * st.volStride > 0xffffffffull);
		uint32_t share = (cfg->flags & MV_FLAG_PIPELINE_FRAMES) ? 80u : 100u;
#ifdef MV_DIAGNOSTICS
		{ const char* e = getenv("MV_MARCH_SHARE"); if (e) share = (uint32_t)atoi(e); }
#endif
		st.marchWaves = (uint32_t)((uint64_t)resident * share / 100u);
		if (st.marchWaves < 64u) st.marchWaves = 64u;
	}
	st.lightBlocks = ((st.L + 3u) >> 2) * ((st.L + 3u) >> 2) * ((st.L + 3u) >> 2);
	ctx->orderTiles = !(cfg->flags & MV_FLAG_LIST_ORDER);
	for (uint32_t p = 0; p < ctx->numSlots; ++p)
	{
		if ((rc = devAlloc(ctx, &ctx->fb[p].tileOrder, mv::kOrderBins * ((size_t)st.maxTiles + st.maxScreenTiles + 1) * sizeof(uint32_t))) != MV_OK) return bail(rc);	// a region per cost bin
		if ((rc = devAlloc(ctx, &ctx->fb[p].binCounts, mv::kOrderBins * sizeof(uint32_t))) != MV_OK) return bail(rc);
	}
	// Per-tile cost table: ONE for all frame slots -- every view march writes its tiles' costs into it, every tile order reads what is there,
	// i.e. the costs of the newest frame whose march has got that far (one or two frames back in steady state).  A table per slot (rounds
	// 1-3) made every order four frames old, which a moving camera paid for: the silhouettes it sorts by had moved on (bench.py --orbit).
	// The order only decides which wavefront marches which tile, so costs of mixed age -- the march writes while the order reads -- cost
	// nothing but, at worst, a tile's place in the launch.
	{
		uint32_t* costs = nullptr;
		if ((rc = devAlloc(ctx, &costs, (size_t)mv::tileCostKeyCount(st.V, st.N, st.maxScreenTiles) * sizeof(uint32_t))) != MV_OK) return bail(rc);
		for (uint32_t p = 0; p < ctx->numSlots; ++p) ctx->fb[p].tileCost = costs;
	}
	if ((rc = devAlloc(ctx, &st.lightStats, (size_t)st.lightBlocks * sizeof(uint64_t))) != MV_OK) return bail(rc);
	// (MV_FORCE_TWO_SETS: validation only -- a one-stream context that alternates output sets like a pipelined one, see scripts/soak.py)
#ifdef MV_DIAGNOSTICS
	if (!(cfg->flags & MV_FLAG_PIPELINE_FRAMES) && getenv("MV_FORCE_TWO_SETS")) ctx->numOutSets = 2;
#endif
	for (uint32_t p = 0; p < ctx->numOutSets; ++p)
	{
		OutBufs& f = ctx->ob[p];
		// cubeMaps[V] (5 mips), cubeDepths[V] (MultiRayCaster.cpp:107-121), zero-initialised
		if ((rc = devAllocShared(ctx, &f.cubeMaps, (size_t)st.V * chain * sizeof(uint2))) != MV_OK) return bail(rc);
		if ((rc = devAllocShared(ctx, &f.cubeDepths, (size_t)st.V * chain * sizeof(float))) != MV_OK) return bail(rc);
		// (sharded: room for the rank-major layout of the multi-GPU frame loop, shard_count blocks of ceil(tile rows / shard_count) tile rows)
		if ((rc = devAllocShared(ctx, &f.composite, (size_t)st.W * compositeRows(st) * sizeof(uint2))) != MV_OK) return bail(rc);
		if ((rc = devAlloc(ctx, &f.screenLayers, (size_t)st.screenSlots * st.W * st.H * sizeof(float4))) != MV_OK) return bail(rc);
		if ((rc = devAlloc(ctx, &f.faceStamps, (size_t)st.V * MV_NUM_CUBE_MIP * 6 * sizeof(uint32_t))) != MV_OK) return bail(rc);
		if ((rc = devAlloc(ctx, &f.waveStats, (size_t)st.marchWaves * 8 * sizeof(uint64_t))) != MV_OK) return bail(rc);
		if (p == 0 && (rc = devAlloc(ctx, &st.timeline, (size_t)st.marchWaves * 8 * sizeof(uint64_t))) != MV_OK) return bail(rc);
		// the ray queue: 64 records per tile of the largest frame (every owned volume at mip 0 with 6 faces + every screen tile); never cleared
		if (cfg->flags & MV_FLAG_RAY_QUEUE)
		{
			if ((rc = devAlloc(ctx, &f.rayRecords, ((size_t)st.maxTiles + st.maxScreenTiles) * 64 * sizeof(mv::RayRecord), false)) != MV_OK) return bail(rc);
			if ((rc = devAlloc(ctx, &f.tileInfo, ((size_t)st.maxTiles + st.maxScreenTiles) * sizeof(mv::TileInfo), false)) != MV_OK) return bail(rc);
		}
		if ((rc = devAlloc(ctx, &f.compositeStats, (size_t)st.compositeSlots * sizeof(uint64_t))) != MV_OK) return bail(rc);
		if ((rc = devAlloc(ctx, &f.compositeRetry, ((size_t)st.compositeSlots + 2u) * sizeof(uint32_t))) != MV_OK) return bail(rc);
	}
	st.volumeDescs = ctx->dDescs; st.sh = ctx->dSh;
	ctx->slot = 0; ctx->set = 0;
	applyParity(ctx);

	// pinned upload ring + readback scratch
	ctx->sliceBytes = 256 + (size_t)st.V * MV_PER_OBJECT_BYTES;
	if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinned), ctx->sliceBytes * kMaxFrameSlots + 256, hipHostMallocDefault) != hipSuccess) return bail(MV_ERR_OUT_OF_MEMORY);
	ctx->pinnedStats = reinterpret_cast<uint64_t*>(ctx->pinned + ctx->sliceBytes * kMaxFrameSlots);
	ctx->pinnedCounters = reinterpret_cast<uint32_t*>(ctx->pinned + ctx->sliceBytes * kMaxFrameSlots + 128);
	if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinnedTiles), (1u + kMaxFrameSlots) * sizeof(unsigned long long), hipHostMallocCoherent) != hipSuccess) return bail(MV_ERR_OUT_OF_MEMORY);
	for (uint32_t i = 0; i <= kMaxFrameSlots; ++i) ctx->pinnedTiles[i] = ~0ull;	// [0] the newest cull's report, [1 + slot] every frame slot's own
	st.tilesReport = ctx->pinnedTiles;
	ctx->rayQueue = (cfg->flags & MV_FLAG_RAY_QUEUE) != 0;
	if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinnedCull), 64 + (size_t)st.V * 24, hipHostMallocDefault) != hipSuccess) return bail(MV_ERR_OUT_OF_MEMORY);

	// default VolumeDesc[V] = { i % S, NUM_CUBE_MIP, N } (MultiRayCaster.cpp:471-479)
	{
		std::vector<uint32_t> descs(st.V);
		for (uint32_t i = 0; i < st.V; ++i) descs[i] = (i % st.S) | ((uint32_t)MV_NUM_CUBE_MIP << 14) | (st.N << 18);
		if (hipMemcpyAsync(ctx->dDescs, descs.data(), descs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
		if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	}

	if (hipStreamSynchronize(ctx->stream) != hipSuccess) return bail(MV_ERR_HIP);
	{
		// (a warning, not an error: more streams than hardware queues work, and wait for each other)
		const uint32_t streams = ctx->numViewStreams + (ctx->cullOnChain ? 0u : 1u) + (ctx->ownStreamL ? 1u : 0u);
		const char* q = getenv("GPU_MAX_HW_QUEUES");
		const uint32_t queues = q ? (uint32_t)atoi(q) : 4u;
		if ((cfg->flags & MV_FLAG_PIPELINE_FRAMES) && streams > queues)
		{
			char buf[256];
			snprintf(buf, sizeof(buf), "warning: this context keeps %u streams busy and the process has %u hardware queues (GPU_MAX_HW_QUEUES%s): streams that share a queue wait for each other",
				streams, queues, q ? "" : " is not set: the runtime's default of 4");
			ctx->warning = buf;	// (not lastError: that is "" unless a call failed -- ADVICE r05)
		}
	}
	*out = ctx;
	return MV_OK;
}

void mv_destroy(mv_ctx* ctx)
{
	if (!ctx) return;
#ifdef MV_HOST_PROFILE
	if (gSectionCalls)
	{
		static const char* names[12] = { "update_frame", "cull + orders", "tail: exchange", "copy-forward", "tail: composite + rows", "light + exchange", "view march", "bookkeeping", "(report)", "(pack)", "(all-gather)", "(unpack)" };
		fprintf(stderr, "\nmv_dist_frame host time per call (us), %llu calls:", (unsigned long long)gSectionCalls);
		for (int i = 0; i < 12; ++i) { fprintf(stderr, "  %s %.2f", names[i], gSection[i] / gSectionCalls); gSection[i] = 0.0; }
		fprintf(stderr, "\n");
		gSectionCalls = 0;
	}
#endif
	if (ctx->stream) (void)syncAll(ctx);
	foldEvents(ctx);
	distRelease(ctx);
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i) { if (ctx->evCopy[i]) (void)hipEventDestroy(ctx->evCopy[i]); if (ctx->evLanded[i]) (void)hipEventDestroy(ctx->evLanded[i]); }
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i) { if (ctx->evStageA[i]) (void)hipEventDestroy(ctx->evStageA[i]); if (ctx->evMainDone[i]) (void)hipEventDestroy(ctx->evMainDone[i]); if (ctx->evLight[i]) (void)hipEventDestroy(ctx->evLight[i]); }
	if (ctx->evProducers) (void)hipEventDestroy(ctx->evProducers);
	if (ctx->ownStreamA && ctx->streamA) (void)hipStreamDestroy(ctx->streamA);
	if (ctx->ownStreamL && ctx->streamL) (void)hipStreamDestroy(ctx->streamL);
	for (uint32_t i = 1; i < kMaxOutSets; ++i) if (ctx->ownStreamB1 && ctx->streamB[i] && ctx->streamB[i] != ctx->stream) (void)hipStreamDestroy(ctx->streamB[i]);
	for (EventPair& ev : ctx->freeEvents) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
	for (void* p : ctx->allocs) (void)hipFree(p);
	if (ctx->meshPositions) (void)hipFree(ctx->meshPositions);
	if (ctx->meshIndices) (void)hipFree(ctx->meshIndices);
	if (ctx->rasterBits) (void)hipFree(ctx->rasterBits);
	if (ctx->dMeshMatrix) (void)hipFree(ctx->dMeshMatrix);
	if (ctx->pinned) (void)hipHostFree(ctx->pinned);
	if (ctx->pinnedCull) (void)hipHostFree(ctx->pinnedCull);
	if (ctx->pinnedTiles) (void)hipHostFree(ctx->pinnedTiles);
	if (ctx->ownStream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

int mv_volume_upload_rgba16f(mv_ctx* ctx, uint32_t src, const uint16_t* rgba)
{
	if (!ctx || !rgba) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, src < ctx->st.S, MV_ERR_RANGE, "volume source index out of range");
	MV_DRAIN(ctx);	// frames in flight on the internal streams may still read this volume and the staging buffer
	MV_HIP(ctx, hipMemcpyAsync(ctx->staging, rgba, ctx->stagingBytes, hipMemcpyHostToDevice, ctx->stream));
	mv::launchRetile(ctx->stream, ctx->staging, ctx->volumes[src], ctx->st.N, ctx->st.bricks);
	mv::launchBuildAlphaCells(ctx->stream, ctx->volumes[src], ctx->cells[src], ctx->st.N, ctx->st.bricks);
	mv::launchBuildOccupancy(ctx->stream, ctx->volumes[src], ctx->occupancy[src], ctx->st.N, ctx->st.bricks);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, (hipError_t)syncAll(ctx));	// the caller may reuse `rgba` and the staging buffer is shared
	return MV_OK;
}

int mv_volume_upload_rgba16f_device(mv_ctx* ctx, uint32_t src, const void* d_rgba)
{
	if (!ctx || !d_rgba) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, src < ctx->st.S, MV_ERR_RANGE, "volume source index out of range");
	MV_DRAIN(ctx);	// frames in flight may still read this volume
	mv::launchRetile(ctx->stream, reinterpret_cast<const uint2*>(d_rgba), ctx->volumes[src], ctx->st.N, ctx->st.bricks);
	mv::launchBuildAlphaCells(ctx->stream, ctx->volumes[src], ctx->cells[src], ctx->st.N, ctx->st.bricks);
	mv::launchBuildOccupancy(ctx->stream, ctx->volumes[src], ctx->occupancy[src], ctx->st.N, ctx->st.bricks);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, evRecord(ctx, ctx->evProducers, ctx->stream));	// stream A must not read this volume before it is complete
	ctx->producersDirty = true;
	return MV_OK;
}

int mv_volume_init_procedural(mv_ctx* ctx, uint32_t src)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, src < ctx->st.S, MV_ERR_RANGE, "volume source index out of range");
	MV_DRAIN(ctx);	// frames in flight may still read this volume
	mv::launchInitGridData(ctx->stream, ctx->volumes[src], ctx->st.N, ctx->st.bricks);
	mv::launchBuildAlphaCells(ctx->stream, ctx->volumes[src], ctx->cells[src], ctx->st.N, ctx->st.bricks);
	mv::launchBuildOccupancy(ctx->stream, ctx->volumes[src], ctx->occupancy[src], ctx->st.N, ctx->st.bricks);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, evRecord(ctx, ctx->evProducers, ctx->stream));	// stream A must not read this volume before it is complete
	ctx->producersDirty = true;
	return MV_OK;
}

int mv_volume_load_r32f(mv_ctx* ctx, uint32_t src, const float* a, uint32_t w, uint32_t h, uint32_t d)
{
	if (!ctx || !a || !w || !h || !d) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, src < ctx->st.S, MV_ERR_RANGE, "volume source index out of range");
	float* dSrc = nullptr;
	const size_t bytes = (size_t)w * h * d * sizeof(float);
	MV_DRAIN(ctx);	// frames in flight may still read this volume
	MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&dSrc), bytes));
	hipError_t e = hipMemcpyAsync(dSrc, a, bytes, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess)
	{
		mv::launchR32FToRGBA16F(ctx->stream, dSrc, w, h, d, ctx->volumes[src], ctx->st.N, ctx->st.bricks);
		mv::launchBuildAlphaCells(ctx->stream, ctx->volumes[src], ctx->cells[src], ctx->st.N, ctx->st.bricks);
		mv::launchBuildOccupancy(ctx->stream, ctx->volumes[src], ctx->occupancy[src], ctx->st.N, ctx->st.bricks);
		e = hipGetLastError();
	}
	const hipError_t e2 = hipStreamSynchronize(ctx->stream);
	(void)hipFree(dSrc);
	MV_HIP(ctx, e);
	MV_HIP(ctx, e2);
	return MV_OK;
}

int mv_set_volume_descs(mv_ctx* ctx, const uint32_t* packed)
{
	if (!ctx || !packed) return MV_ERR_INVALID_ARG;
	for (uint32_t i = 0; i < ctx->st.V; ++i)
	{
		MV_REQUIRE(ctx, (packed[i] & 0x3fffu) < ctx->st.S, MV_ERR_RANGE, "VolumeDesc.VolTexId out of range");
		MV_REQUIRE(ctx, (packed[i] >> 18) == ctx->st.N, MV_ERR_INVALID_ARG, "VolumeDesc.CubeMapSize must equal grid_size");
		const uint32_t mips = (packed[i] >> 14) & 0xfu;
		MV_REQUIRE(ctx, mips >= 1 && mips <= MV_NUM_CUBE_MIP, MV_ERR_INVALID_ARG, "VolumeDesc.NumMips must be 1..5");
	}
	MV_DRAIN(ctx);	// the cull / marches of frames in flight read VolumeDesc[V]
	MV_HIP(ctx, hipMemcpyAsync(ctx->dDescs, packed, (size_t)ctx->st.V * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

// SetViewport (MultiRayCaster.cpp:226-241; called by SetRenderTargets :211-224 and on a resize, MultiVolumes.cpp:382): the reference
// re-creates its K-buffers and nothing else.  Here everything that is W x H goes -- the composite image, the RayCast layers, the
// context's own depth map, the composite's counters, the screen-tile part of the tile order -- and the volumes, light maps, cube maps
// (with their stamps) and the frame counters stay.
int mv_set_viewport(mv_ctx* ctx, uint32_t w, uint32_t h)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, w >= 1 && h >= 1 && w <= 65535u && h <= 65535u, MV_ERR_INVALID_ARG, "viewport out of range");
	MV_REQUIRE(ctx, !ctx->dist, MV_ERR_INVALID_ARG, "detach the multi-GPU frame loop (mv_dist_shutdown) before changing the viewport: its row buffers are sized for it");
	DeviceState& st = ctx->st;
	if (w == st.W && h == st.H) return MV_OK;
	MV_DRAIN(ctx);
	MV_REQUIRE(ctx, (uint64_t)st.screenSlots * w * h < (1ull << 31), MV_ERR_INVALID_ARG, "viewport too large for the RayCast layers' 31-bit element index");
	// Everything that depends on W x H is allocated for the new size FIRST and swapped in only when every allocation has succeeded: a
	// failure half way leaves the context exactly as it was (old dimensions, old buffers), not with new dimensions over old, smaller buffers.
	DeviceState next = st;
	next.W = w; next.H = h;
	viewportDerived(next);
	struct Swap { void** slot; void* fresh; bool tracked; };
	std::vector<Swap> swaps;
	auto want = [&](auto** slot, size_t bytes, bool zero, bool tracked = true) -> int
	{
		void* fresh = nullptr;
		const hipError_t e = hipMalloc(&fresh, bytes ? bytes : 16);
		if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? MV_ERR_OUT_OF_MEMORY : MV_ERR_HIP, "mv_set_viewport: allocating for the new size", e);
		if (zero && hipMemsetAsync(fresh, 0, bytes ? bytes : 16, ctx->stream) != hipSuccess) { (void)hipFree(fresh); return fail(ctx, MV_ERR_HIP, "mv_set_viewport: clearing a new buffer"); }
		swaps.push_back({ reinterpret_cast<void**>(slot), fresh, tracked });
		return MV_OK;
	};
	int rc = want(&ctx->ownDepth, (size_t)w * h * sizeof(float), true);
	for (uint32_t p = 0; p < ctx->numSlots && rc == MV_OK; ++p)
		rc = want(&ctx->fb[p].tileOrder, mv::kOrderBins * ((size_t)next.maxTiles + next.maxScreenTiles + 1) * sizeof(uint32_t), true);
	uint32_t* costs = ctx->fb[0].tileCost;	// (one table for all slots)
	if (rc == MV_OK) rc = want(&costs, (size_t)mv::tileCostKeyCount(next.V, next.N, next.maxScreenTiles) * sizeof(uint32_t), true);
	for (uint32_t p = 0; p < ctx->numOutSets && rc == MV_OK; ++p)
	{
		OutBufs& f = ctx->ob[p];
		rc = want(&f.composite, (size_t)next.W * compositeRows(next) * sizeof(uint2), true);
		if (rc == MV_OK) rc = want(&f.screenLayers, (size_t)next.screenSlots * next.W * next.H * sizeof(float4), true);
		if (rc == MV_OK) rc = want(&f.compositeStats, (size_t)next.compositeSlots * sizeof(uint64_t), true);
		if (rc == MV_OK) rc = want(&f.compositeRetry, ((size_t)next.compositeSlots + 2u) * sizeof(uint32_t), true);
		if (rc == MV_OK && ctx->rayQueue) rc = want(&f.rayRecords, ((size_t)next.maxTiles + next.maxScreenTiles) * 64 * sizeof(mv::RayRecord), false);
		if (rc == MV_OK && ctx->rayQueue) rc = want(&f.tileInfo, ((size_t)next.maxTiles + next.maxScreenTiles) * sizeof(mv::TileInfo), false);
	}
	if (rc == MV_OK && ctx->rasterBits) rc = want(&ctx->rasterBits, std::max((size_t)w * h, (size_t)next.M * next.M) * sizeof(uint32_t), false, false);
	if (rc != MV_OK)
	{
		for (Swap& s : swaps) (void)hipFree(s.fresh);
		return rc;
	}
	// commit: pointers, then dimensions
	uint32_t* oldCosts = ctx->fb[0].tileCost;
	for (Swap& s : swaps)
	{
		void* old = s.slot == reinterpret_cast<void**>(&costs) ? static_cast<void*>(oldCosts) : *s.slot;
		if (s.tracked) { for (void*& t : ctx->allocs) if (t == old) t = s.fresh; }
		(void)hipFree(old);
		*s.slot = s.fresh;
	}
	for (uint32_t p = 0; p < ctx->numSlots; ++p) ctx->fb[p].tileCost = costs;
	st.W = w; st.H = h;
	ctx->cfg.viewport_w = w; ctx->cfg.viewport_h = h;
	viewportDerived(st);
	st.depthMap = ctx->ownDepth;	// (a borrowed depth map was the old viewport's: the caller hands over the new one)
	ctx->haveDepth = false;
	applyParity(ctx);
	ctx->haveCull = false;	// the frame in progress (if any) was culled for the old viewport: mv_update_frame begins the next one
	ctx->haveFrame = false;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_set_depth_map(mv_ctx* ctx, const float* d32)
{
	if (!ctx || !d32) return MV_ERR_INVALID_ARG;
	// Frames in flight (view march / composite on the internal streams) still read the depth map: wait for them BEFORE the
	// copy overwrites it.  Callers that change the depth every frame and want to keep frames in flight hand over a device
	// pointer per frame instead (mv_set_depth_map_device: borrowed, never written by the context).
	MV_DRAIN(ctx);
	MV_HIP(ctx, hipMemcpyAsync(ctx->ownDepth, d32, (size_t)ctx->st.W * ctx->st.H * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	ctx->st.depthMap = ctx->ownDepth;
	ctx->haveDepth = true;
	return MV_OK;
}

int mv_set_shadow_map(mv_ctx* ctx, const uint16_t* d16)
{
	if (!ctx || !d16) return MV_ERR_INVALID_ARG;
	MV_DRAIN(ctx);	// the light march of a frame in flight may still read the shadow map
	MV_HIP(ctx, hipMemcpyAsync(ctx->ownShadow, d16, (size_t)ctx->st.M * ctx->st.M * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	ctx->st.shadowMap = ctx->ownShadow;
	ctx->haveShadow = true;
	return MV_OK;
}

int mv_set_depth_map_device(mv_ctx* ctx, const float* d_d32)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	ctx->st.depthMap = d_d32 ? d_d32 : ctx->ownDepth;
	if (d_d32) ctx->haveDepth = true;
	return MV_OK;
}

int mv_set_shadow_map_device(mv_ctx* ctx, const uint16_t* d_d16)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	ctx->st.shadowMap = d_d16 ? d_d16 : ctx->ownShadow;
	if (d_d16) ctx->haveShadow = true;
	return MV_OK;
}

int mv_set_sh(mv_ctx* ctx, const float* sh27)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	if (sh27)
	{
		MV_DRAIN(ctx);	// the light march of a frame in flight may still read the coefficients
		MV_HIP(ctx, hipMemcpyAsync(ctx->dSh, sh27, 27 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
		MV_HIP(ctx, (hipError_t)syncAll(ctx));
	}
	ctx->st.hasLightProbe = sh27 ? 1u : 0u;
	return MV_OK;
}

int mv_set_max_samples(mv_ctx* ctx, uint32_t maxRaySamples, uint32_t maxLightSamples)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, maxRaySamples >= 1 && maxRaySamples <= 65535 && maxLightSamples >= 1 && maxLightSamples <= 65535, MV_ERR_INVALID_ARG,
		"sample counts must fit the u16 VolumeInfo.SmpCount field");
	ctx->st.maxRaySamples = maxRaySamples;
	ctx->st.maxLightSamples = maxLightSamples;
	return MV_OK;
}

// The cull rotates the light-table entry of the volume this frame relights to that volume's next buffer, expecting the
// frame's light march to fill it.  A caller may leave mv_march_light out of a frame (the reference then simply keeps its light
// maps): the first stage of such a frame that reads light maps puts the previous frame's table back, so the frame -- and the
// frames after it, which derive their table from this one -- read the maps that were last written.
int mvi::keepLightTable(mv_ctx* ctx)
{
	if (ctx->lightMarched || ctx->lightTableKept) return MV_OK;
	const FrameBufs& f = ctx->fb[ctx->slot];
	const FrameBufs& q = ctx->fb[(ctx->slot + ctx->numSlots - 1u) % ctx->numSlots];
	MV_HIP(ctx, hipMemcpyAsync(f.lightTable, q.lightTable, (size_t)ctx->st.V * sizeof(uint4*), hipMemcpyDeviceToDevice, ctx->streamA));
	MV_HIP(ctx, stageADone(ctx));
	ctx->lightTableKept = true;
	return MV_OK;
}

#ifndef MV_SLOT_SPIN_US
#define MV_SLOT_SPIN_US 2000	// how long mv_update_frame polls for a free frame slot before it blocks (0: block at once)
#endif
int mv_update_frame(mv_ctx* ctx, const void* cbPerFrame, const void* perObject)
{
	if (!ctx || !cbPerFrame || !perObject) return MV_ERR_INVALID_ARG;
	// The frame that ends here may have had NO stage that reads light maps (mv_cull alone, or nothing at all): nobody has put the previous
	// frame's light table back behind its cull's pre-rotation (keepLightTable), and this frame's cull derives its table from that one --
	// the volume that frame would have relit then pointed at a buffer nobody wrote, until its next relight (round 6: tests/test_api_fuzz.py
	// found it on the four-chain context, where an unwritten buffer is likeliest: a whole light map of zeros for up to V frames).
	// (frameBegun, not haveFrame: a viewport change between the two frames forgets the frame in progress but not its slot's table)
	if (ctx->frameBegun && !ctx->lightMarched && !ctx->lightTableKept && !ctx->batchExplicit) { const int rc = keepLightTable(ctx); if (rc != MV_OK) return rc; }
	// a new frame: the next slot of constants / cull buffers (and the other parity's march outputs)
	ctx->slot = (ctx->slot + 1u) % ctx->numSlots;
	++ctx->frame;
	const uint32_t prevSlot = (ctx->slot + ctx->numSlots - 1u) % ctx->numSlots;
	if (ctx->framesPerLaunch > 1u && !ctx->batchExplicit)
	{
		// (several frames per launch outside the multi-GPU loop: B consecutive frames share a view stream, nothing else changes)
		ctx->batchPos = (ctx->batchPos + 1u) % ctx->framesPerLaunch;
		if (ctx->batchPos == 0u) ++ctx->batchIndex;
	}
	ctx->prevChain = ctx->curChain;
	ctx->curChain = chainOfFrame(ctx);
	if (ctx->cullOnChain)
	{
		// (MV_FLAG_CULL_ON_VIEW_STREAM) this frame's stage A runs on its own view stream; what stream A's order used to give must then
		// be said: the cull reads the previous frame's light table, VolumeInfo and view hash (lightTablePrev, volumeInfoPrev, countersPrev)
		ctx->streamA = ctx->streamB[chainOfFrame(ctx)];
		// (the multi-GPU loop with several frames per launch culls a group's frames together, later: the wait goes in front of that launch
		// -- batchClose -- and the uploads need not wait for the culls of the group before)
		if (!ctx->batchExplicit) MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evStageA[prevSlot]));
	}
	ctx->set = (uint32_t)(ctx->frame % ctx->numOutSets);
	ctx->st.frameTag = (uint32_t)ctx->frame;
	latencyBegin(ctx, ctx->slot, ctx->frame);
	applyParity(ctx);
	// the slot's previous user (ctx->numSlots frames ago) must be done with it; volume producers on stream B must have finished
	{
	MV_SECTION(8);
	MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evMainDone[ctx->slot]));
	MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evLight[ctx->slot]));
	if (ctx->producersDirty) { MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evProducers)); ctx->producersDirty = false; }
	}
	// The constants go into this slot's pinned slice (the reference's upload-heap slices, MultiRayCaster.cpp:128-130, 461-467) and from
	// there to the device in one copy on stream A.  The slice's previous user is the frame numSlots back: its copy must have run --
	// in steady state it has, long ago (the slot's stage-A event covers it: no event of its own); a host that runs more than that many
	// frames ahead of the GPU waits here, as the reference's does on its fence.
	MV_SECTION(9);
	if (hipEventQuery(ctx->evStageA[ctx->slot]) != hipSuccess)
	{
		// (polled for a while before the blocking wait: a host that is only just ahead of the GPU finds the event done within
		// microseconds, while waking from hipEventSynchronize took ~100 us here -- time in which the queues of a rank of the multi-GPU loop ran dry)
		const auto t0 = std::chrono::steady_clock::now();
		bool done = false;
		while (!done && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(MV_SLOT_SPIN_US)) { for (int i = 0; i < 64; ++i) __builtin_ia32_pause(); done = hipEventQuery(ctx->evStageA[ctx->slot]) == hipSuccess; }
		if (!done) MV_HIP(ctx, hipEventSynchronize(ctx->evStageA[ctx->slot]));
		ctx->hostWaitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
#ifdef MV_HOST_PROFILE
		{
			static double waitUs[4]; static uint64_t waits[4], calls;
			waitUs[ctx->batchPos & 3u] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); ++waits[ctx->batchPos & 3u];
			if (++calls % 100 == 0) { fprintf(stderr, " [slot waits by position in the group:"); for (int i = 0; i < 4; ++i) { fprintf(stderr, " %llu x %.0f us", (unsigned long long)waits[i], waits[i] ? waitUs[i] / waits[i] : 0.0); waitUs[i] = 0; waits[i] = 0; } fprintf(stderr, "]"); }
		}
#endif
	}
	MV_SECTION(10);
	uint8_t* s = ctx->pinned + (size_t)ctx->slot * ctx->sliceBytes;
	memcpy(s, cbPerFrame, MV_CB_PER_FRAME_BYTES);
	memcpy(s + 256, perObject, (size_t)ctx->st.V * MV_PER_OBJECT_BYTES);
	if (!ctx->deferUpload) MV_HIP(ctx, hipMemcpyAsync(ctx->fb[ctx->slot].dCb, s, ctx->sliceBytes, hipMemcpyHostToDevice, ctx->streamA));	// (deferred: the multi-GPU loop uploads a group's slices together)
	MV_HIP(ctx, stageADone(ctx, false));	// (a frame that never culls still marks its copy; inside the multi-GPU loop the cull's record follows at once)
	ctx->haveFrame = true; ctx->frameBegun = true;
	ctx->haveCull = false;	// this frame's slot holds the cull outputs of four frames ago until mv_cull runs
	ctx->lightMarched = false; ctx->lightTableKept = false; ctx->copiedForward = false; ctx->marchedThisFrame = false;
	return MV_OK;
}

int mv_cull(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveFrame, MV_ERR_NOT_READY, "mv_update_frame has not been called");
	EventPair ev = { nullptr, nullptr, 0 };
	beginStage(ctx, 0, ev, ctx->streamA);
	mv::launchCull(ctx->streamA, ctx->st);
	endStage(ctx, ev, ctx->streamA);
	// Behind the cull on stream A (which runs frames ahead), one launch: the dispatch order of this frame's view march, from the cost
	// table the view march of the slot's previous user wrote (mv_update_frame already made stream A wait for that frame), and the
	// order of the light march's voxel blocks.
	mv::launchOrders(ctx->streamA, ctx->st);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, stageADone(ctx));
	ctx->haveCull = true;
	return MV_OK;
}

// EXPERIMENT (-DMV_DIAGNOSTICS builds, MV_DIST_LIGHT_ON_A=1; profiles/r03_light_on_stream_a.txt: slower): on a rank of the multi-GPU loop with whole-frame chains, the light share-out (march, all-gather,
// unpack) goes on stream A behind the frame's cull instead of in front of the frame's view march on its chain
int mv_march_light(mv_ctx* ctx) { return ctx ? marchLight(ctx, nullptr) : MV_ERR_INVALID_ARG; }

// wordsOut: inside the multi-GPU frame loop, where this rank's share of the relit map goes for the all-gather
int mvi::marchLight(mv_ctx* ctx, uint32_t* wordsOut)
{
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	MV_REQUIRE(ctx, ctx->haveShadow, MV_ERR_NOT_READY, "no shadow map (mv_set_shadow_map)");
	MV_REQUIRE(ctx, !ctx->lightTableKept, MV_ERR_NOT_READY, "mv_march_light must come before the frame's mv_march_view / mv_composite");
	// The light buffer this launch writes was last read by a frame at least lightBufs (3 by default) frames back (the relit volume's
	// table entry has rotated since): that frame's view march / composite must be over.  Older ones are, by stream order.
	// (slot + 1) mod numSlots is the slot of frame i - (numSlots - 1) = i - lightBufs; with three chains and three buffers that frame sits
	// on this frame's own stream and the wait is left out (evWait)
	// With several view streams the light march goes in front of its frame's view march on that frame's stream: every stream then
	// carries a whole frame (light march, view march, composite) and the frames overlap as wholes.  (A stream of its own for the
	// light marches serialises them at the share of the GPU one kernel gets beside two view marches, which made them the chain
	// every frame waited for: 0.349 -> 0.331 ms per frame at C3 with two output sets, 0.311 ms with three.)
	hipStream_t L = ctx->lightOnViewStream ? streamB(ctx) : ctx->streamL;
	if (distLightOnStreamA(ctx)) L = ctx->stream;
	MV_HIP(ctx, evWait(ctx, L, ctx->evMainDone[(ctx->slot + ctx->numSlots - ctx->st.lightBufs) % ctx->numSlots]));
	MV_HIP(ctx, evWait(ctx, L, ctx->evStageA[ctx->slot]));	// this frame's upload + cull
	EventPair ev = { nullptr, nullptr, 0 };
	beginStage(ctx, 1, ev, L);
	mv::launchMarchLight(L, ctx->st, wordsOut);
	endStage(ctx, ev, L);
	MV_HIP(ctx, hipGetLastError());
	if (!(ctx->leanEvents && ctx->st.shardCount > 1u)) MV_HIP(ctx, evRecord(ctx, ctx->evLight[ctx->slot], L));	// (lean: the light-map exchange follows on this stream and records it)
	ctx->lightMarched = true;
	return MV_OK;
}

// Bring the cube faces this frame does NOT march up to date from the other output set (mv_composite.hip copyForwardKernel).
// It reads the other set -- complete once the previous frame's view march (and, on a multi-GPU rank, its unpack) has finished --
// and writes faces nobody else writes this frame, so it can run beside this frame's view march; the composite of this frame and
// the view march of the NEXT frame (which writes the other set again) wait for it.
int mvi::copyForward(mv_ctx* ctx, hipStream_t s)
{
	const OutBufs& other = ctx->ob[(ctx->set + ctx->numOutSets - 1u) % ctx->numOutSets];
	mv::launchCopyForward(s, ctx->st, other.cubeMaps, other.cubeDepths, other.faceStamps, (uint32_t)ctx->frame, true);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, evRecord(ctx, ctx->evCopy[ctx->slot], s));
	ctx->copiedForward = true;
	return MV_OK;
}

// The same for a frame that cannot rely on "the previous frame's set is complete": the previous frame neither marched nor composited
// (its set was never brought up to date), or this frame has no view march at all (mv_composite without mv_march_view: the reference
// would show the cube maps as the last march left them).  Rare call patterns: the frames in flight are drained, then every other
// set is a source -- the stamps pick the newest data of every face whatever the order.
static int copyForwardFromAllSets(mv_ctx* ctx, bool marchFollows)
{
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	hipStream_t s = streamB(ctx);
	for (uint32_t p = 0; p < ctx->numOutSets; ++p)
	{
		if (p == ctx->set) continue;
		const OutBufs& other = ctx->ob[p];
		mv::launchCopyForward(s, ctx->st, other.cubeMaps, other.cubeDepths, other.faceStamps, (uint32_t)ctx->frame, marchFollows);
		MV_HIP(ctx, hipGetLastError());
	}
	MV_HIP(ctx, evRecord(ctx, ctx->evCopy[ctx->slot], s));
	ctx->copiedForward = true;
	return MV_OK;
}

#ifndef MV_COPY_FORWARD
#define MV_COPY_FORWARD 1	// (0: A/B timing only -- the two output sets then drift apart in the faces a frame does not march)
#endif
// First half of mv_march_view: bring the faces this frame does not march up to date in its output set (see copyForward above).
int mvi::viewCopyForward(mv_ctx* ctx)
{
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	MV_REQUIRE(ctx, ctx->haveDepth, MV_ERR_NOT_READY, "no depth map (mv_set_depth_map)");
	const uint32_t prevSlot = (ctx->slot + ctx->numSlots - 1u) % ctx->numSlots;
	const uint32_t n = ctx->numOutSets;
	const bool ownCopy = !ctx->dist || ctx->dist->chains;	// (the single-view-stream multi-GPU loop copies on its stream X)
	if (MV_COPY_FORWARD && n >= 2u && ownCopy && ctx->lastCompleteFrame != 0u && ctx->lastCompleteFrame + 1u != ctx->frame && !ctx->copiedForward)
	{
		// the previous frame left its set untouched (no view march, no composite): it is not the complete history the fast path reads
		const int rc = copyForwardFromAllSets(ctx, true);
		if (rc != MV_OK) return rc;
	}
	else if (MV_COPY_FORWARD && n >= 2u && ownCopy)
	{
		// (also when a composite of this frame has already copied, without marking the faces this march is about to write)
		// On the stream of the PREVIOUS frame's view march (and composite): stream order alone puts the copy behind the march that
		// completed the set it reads (a stream of its own would do, but HIP streams beyond four share hardware queues and wait for
		// each other: measured).  It writes this frame's set, so the set's last reader (the composite of frame k - n) must be
		// over, and it reads what the previous frame's copy wrote (a face nobody has marched for two frames travels set to set).
		hipStream_t s = ctx->streamB[ctx->prevChain];
		MV_HIP(ctx, evWait(ctx, s, ctx->evMainDone[(ctx->slot + ctx->numSlots - n) % ctx->numSlots]));
		MV_HIP(ctx, evWait(ctx, s, ctx->evCopy[prevSlot]));
		if (ctx->copiedForward) MV_HIP(ctx, evWait(ctx, s, ctx->evCopy[ctx->slot]));
		MV_HIP(ctx, evWait(ctx, s, ctx->evStageA[ctx->slot]));	// this frame's cull
		const int rc = copyForward(ctx, s);
		if (rc != MV_OK) return rc;
	}
	ctx->lastCompleteFrame = ctx->frame;
	return MV_OK;
}

// Second half: the march itself.
int mvi::viewMarch(mv_ctx* ctx)
{
	const uint32_t n = ctx->numOutSets;
	ctx->marchedThisFrame = true;
	MV_HIP(ctx, stageBBegin(ctx));
	// the set this march writes was read by the copy-forward of frame k - n + 1 (as the set before its own) ...
	if (n >= 2u) MV_HIP(ctx, evWait(ctx, streamB(ctx), ctx->evCopy[(ctx->slot + ctx->numSlots - (n - 1u)) % ctx->numSlots]));
	// ... and by the composite of frame k - n, its previous user (with SINGLE_VIEW_STREAM / one stream that is on this stream anyway)
	MV_HIP(ctx, evWait(ctx, streamB(ctx), ctx->evMainDone[(ctx->slot + ctx->numSlots - (n % ctx->numSlots)) % ctx->numSlots]));
	EventPair ev = { nullptr, nullptr, 0 };
	beginStage(ctx, 2, ev, streamB(ctx));
	// Nothing between the cull and the march waits for the host (the shape of the work-graph variant, Sh/LibRayMarch.hlsl:39-134).
	if (ctx->rayQueue)
	{
		// ray setup -> persistent march: fixed grids that read the tile count on the device
		mv::launchRaySetup(streamB(ctx), ctx->st);
		mv::launchMarchView(streamB(ctx), ctx->st);
	}
	else
	{
		// One wavefront per tile.  Every cull leaves {frame : tile count} in pinned memory; the grid is sized from the most recent
		// report there is -- this frame's, or the one or two frames older one if this frame's cull has not run yet -- plus a margin,
		// with one plain read and no wait.  Tile counts move slowly from frame to frame; when one jumps past the margin (a volume
		// changing scheme) the overflow launch inside launchMarchTiles marches the rest, so the estimate only affects speed.
		uint32_t tiles = 0xffffffffu;
		bool exact = false;
		if (ctx->pinnedTiles)
		{
			const unsigned long long v = *reinterpret_cast<const volatile unsigned long long*>(ctx->pinnedTiles);
			if (v != ~0ull)
			{
				const uint32_t count = (uint32_t)v;
				exact = (uint32_t)(v >> 32) == (uint32_t)ctx->frame;	// this frame's cull has already run (it runs ahead on stream A)
				tiles = exact ? count : count + count / 4u + 256u;
			}
		}
		// (peer-store transport: the main launch waits, the LAST launch raises "cube faces landed" -- the overflow launch if there is one)
		const bool overflow = !exact && mv::marchTilesBeyond(ctx->st, tiles) != 0xffffffffu;
		DeviceState stMain = ctx->st;
		if (overflow) stMain.fold.nSignal = 0u;
		const uint32_t beyond = mv::launchMarchTiles(streamB(ctx), stMain, tiles);
		// (an exact count is this frame's own: nothing can lie beyond the grid.  Otherwise the overflow launch follows on the same
		// stream: on the high-priority stream A it would have to wait for this frame's light march and held up the next frame's cull --
		// measured slower.  It is kept small, 64 workgroups, because its wavefronts must find slots on a GPU the other frame's march fills.)
		if (!exact && beyond != 0xffffffffu)
		{
			DeviceState stOver = ctx->st;
			stOver.fold.nWait = 0u;
			mv::launchMarchTilesOverflow(streamB(ctx), stOver, beyond);
		}
	}
	endStage(ctx, ev, streamB(ctx));
	MV_HIP(ctx, hipGetLastError());
	if (!ctx->leanEvents) MV_HIP(ctx, stageBDone(ctx));	// (lean: the frame's tail records it behind the composite, on this stream, before anything waits for it)
	return MV_OK;
}

int mv_march_view(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	int rc = keepLightTable(ctx);	// (a frame without mv_march_light reads the previous frame's light maps)
	if (rc == MV_OK) rc = viewCopyForward(ctx);
	return rc != MV_OK ? rc : viewMarch(ctx);
}

int mv_composite(mv_ctx* ctx, uint32_t row0, uint32_t row1) { return mv_composite_over(ctx, row0, row1, nullptr); }

int mv_composite_over(mv_ctx* ctx, uint32_t row0, uint32_t row1, const void* d_background)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	MV_REQUIRE(ctx, ctx->haveDepth, MV_ERR_NOT_READY, "no depth map (mv_set_depth_map)");
	if (row1 == 0 || row1 > ctx->st.H) row1 = ctx->st.H;
	MV_REQUIRE(ctx, row0 < row1, MV_ERR_RANGE, "empty row range");
	{ const int rc = keepLightTable(ctx); if (rc != MV_OK) return rc; }
	MV_HIP(ctx, stageBBegin(ctx));
	if (MV_COPY_FORWARD && ctx->numOutSets >= 2u && !ctx->dist && !ctx->copiedForward)
	{
		// a composite without a view march this frame: it shows the cube maps as the last marches left them, wherever they are
		const int rc = copyForwardFromAllSets(ctx, false);
		if (rc != MV_OK) return rc;
		ctx->lastCompleteFrame = ctx->frame;
	}
	if (ctx->numOutSets >= 2u && ctx->copiedForward) MV_HIP(ctx, evWait(ctx, streamB(ctx), ctx->evCopy[ctx->slot]));
	EventPair ev = { nullptr, nullptr, 0 };
	beginStage(ctx, 3, ev, streamB(ctx));
	if (ctx->marchedThisFrame) mv::launchComposite(streamB(ctx), ctx->st, row0, row1, d_background);
	else
	{
		// no view march this frame: the RayCast layers of the output set belong to an older frame's camera
		DeviceState st = ctx->st;
		st.flags |= mv::kFlagNoScreenLayers;
		mv::launchComposite(streamB(ctx), st, row0, row1, d_background);
	}
	endStage(ctx, ev, streamB(ctx));
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, stageBDone(ctx));
	return MV_OK;
}

int mv_render(mv_ctx* ctx, int with_composite)
{
	int rc;
	if ((rc = mv_cull(ctx)) != MV_OK) return rc;
	if ((rc = mv_march_light(ctx)) != MV_OK) return rc;
	if ((rc = mv_march_view(ctx)) != MV_OK) return rc;
	if (with_composite && (rc = mv_composite(ctx, 0, 0)) != MV_OK) return rc;
	return MV_OK;
}

int mv_readback_volume_info(mv_ctx* ctx, uint16_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemcpyAsync(dst, ctx->st.volumeInfo, (size_t)ctx->st.V * 4 * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_readback_visible_lists(mv_ctx* ctx, uint32_t* visible, uint32_t* visibleCount, uint32_t* cubeMapVolumes, uint32_t* cubeMapCount)
{
	if (!ctx || !visible || !visibleCount || !cubeMapVolumes || !cubeMapCount) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	uint32_t counters[kNumCounters];
	MV_HIP(ctx, hipMemcpyAsync(counters, ctx->st.counters, sizeof(counters), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, hipMemcpyAsync(visible, ctx->st.visibleVolumes, (size_t)ctx->st.V * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, hipMemcpyAsync(cubeMapVolumes, ctx->st.cubeMapVolumes, (size_t)ctx->st.V * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	*visibleCount = counters[0];
	*cubeMapCount = counters[1];
	return MV_OK;
}

int mv_readback_volume(mv_ctx* ctx, uint32_t src, uint16_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));	// the producing frame may have run on another stream
	MV_REQUIRE(ctx, src < ctx->st.S, MV_ERR_RANGE, "volume source index out of range");
	mv::launchUntile(ctx->stream, ctx->volumes[src], ctx->staging, ctx->st.N, ctx->st.bricks);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, hipMemcpyAsync(dst, ctx->staging, ctx->stagingBytes, hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

static int currentLightBuffer(mv_ctx* ctx, uint32_t volume, uint4** out)
{
	// (unsettled: see below -- then this frame's table, also the copy mv_readback_cull brought back, is not the one to read)
	if (ctx->lightTableFrame == ctx->frame && (ctx->lightMarched || ctx->lightTableKept))
	{
		// this frame's table came back with mv_readback_cull: no host synchronisation, the context's stream waits for the
		// frame's light march instead (work the caller orders on that stream then sees the relit map)
		*out = reinterpret_cast<uint4* const*>(ctx->pinnedCull + 64 + (size_t)ctx->st.V * 16)[volume];
		MV_HIP(ctx, evWait(ctx, ctx->stream, ctx->evStageA[ctx->slot]));
		MV_HIP(ctx, evWait(ctx, ctx->stream, ctx->evLight[ctx->slot]));
		return MV_OK;
	}
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	// The frame in progress has neither relit a volume nor put the previous frame's table back (a cull alone so far, or not even that):
	// its slot's table names, for the volume it would relight, a buffer nobody has written -- or is the table of numSlots frames ago.  The
	// maps that were last written are the previous frame's (what keepLightTable would copy here).
	const bool unsettled = ctx->frameBegun && !ctx->lightMarched && !ctx->lightTableKept;
	uint4* const* table = unsettled ? ctx->fb[(ctx->slot + ctx->numSlots - 1u) % ctx->numSlots].lightTable : ctx->st.lightTable;
	MV_HIP(ctx, hipMemcpy(out, table + volume, sizeof(uint4*), hipMemcpyDeviceToHost));
	return MV_OK;
}

int mv_readback_light_map(mv_ctx* ctx, uint32_t volume, uint32_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, volume < ctx->st.V, MV_ERR_RANGE, "volume index out of range");
	const size_t L3 = (size_t)ctx->st.L * ctx->st.L * ctx->st.L;
	uint4* buf = nullptr;
	const int rc = currentLightBuffer(ctx, volume, &buf);
	if (rc != MV_OK) return rc;
	mv::launchLightFromQuads(ctx->stream, buf, ctx->lightStaging, ctx->st.L);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, hipMemcpyAsync(dst, ctx->lightStaging, L3 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_upload_light_map(mv_ctx* ctx, uint32_t volume, const uint32_t* src)
{
	if (!ctx || !src) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, volume < ctx->st.V, MV_ERR_RANGE, "volume index out of range");
	const size_t L3 = (size_t)ctx->st.L * ctx->st.L * ctx->st.L;
	uint4* buf = nullptr;
	MV_DRAIN(ctx);	// frames in flight may read this buffer (and the shared staging buffer)
	const int rc = currentLightBuffer(ctx, volume, &buf);
	if (rc != MV_OK) return rc;
	MV_HIP(ctx, hipMemcpyAsync(ctx->lightStaging, src, L3 * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
	mv::launchLightToQuads(ctx->stream, ctx->lightStaging, buf, ctx->st.L);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_readback_cube_map(mv_ctx* ctx, uint32_t volume, uint32_t mip, uint16_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));	// the producing frame may have run on another stream
	MV_REQUIRE(ctx, volume < ctx->st.V && mip < MV_NUM_CUBE_MIP, MV_ERR_RANGE, "volume / mip out of range");
	const size_t d = mv::cubeMipDim(ctx->st.N, mip);
	const size_t off = (size_t)volume * mv::cubeChainTexels(ctx->st.N) + mv::cubeMipOffset(ctx->st.N, mip);
	MV_HIP(ctx, hipMemcpyAsync(dst, ctx->st.cubeMaps + off, 6 * d * d * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_readback_cube_depth(mv_ctx* ctx, uint32_t volume, uint32_t mip, float* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));	// the producing frame may have run on another stream
	MV_REQUIRE(ctx, volume < ctx->st.V && mip < MV_NUM_CUBE_MIP, MV_ERR_RANGE, "volume / mip out of range");
	const size_t d = mv::cubeMipDim(ctx->st.N, mip);
	const size_t off = (size_t)volume * mv::cubeChainTexels(ctx->st.N) + mv::cubeMipOffset(ctx->st.N, mip);
	MV_HIP(ctx, hipMemcpyAsync(dst, ctx->st.cubeDepths + off, 6 * d * d * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_readback_composite(mv_ctx* ctx, uint16_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));	// the producing frame may have run on another stream
	const DeviceState& st = ctx->st;
	const size_t rowBytes = (size_t)st.W * sizeof(uint2);
	if (!(st.flags & mv::kFlagInterleaveRows))
	{
		MV_HIP(ctx, hipMemcpyAsync(dst, st.composite, rowBytes * st.H, hipMemcpyDeviceToHost, ctx->stream));
		MV_HIP(ctx, (hipError_t)syncAll(ctx));
		return MV_OK;
	}
	// multi-GPU frame loop: the image is stored rank-major by tile row (mv_kernels.hpp kFlagInterleaveRows); put the rows back in order
	const uint32_t perRank = mv::interleavedRowsPerRank(st.H, st.shardCount), T = (st.H + 7u) >> 3;
	const size_t rows = compositeRows(st);
	uint8_t* tmp = static_cast<uint8_t*>(malloc(rowBytes * rows));
	if (!tmp) return MV_ERR_OUT_OF_MEMORY;
	const hipError_t e = hipMemcpy(tmp, st.composite, rowBytes * rows, hipMemcpyDeviceToHost);
	if (e == hipSuccess)
		for (uint32_t t = 0; t < T; ++t)
		{
			const uint32_t y0 = t * 8u, n = y0 + 8u <= st.H ? 8u : st.H - y0;
			const size_t stored = ((size_t)(t % st.shardCount) * perRank + t / st.shardCount) * 8u;
			memcpy(reinterpret_cast<uint8_t*>(dst) + (size_t)y0 * rowBytes, tmp + stored * rowBytes, (size_t)n * rowBytes);
		}
	free(tmp);
	MV_HIP(ctx, e);
	return MV_OK;
}

int mv_clear_cube_maps(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	const size_t chain = mv::cubeChainTexels(ctx->st.N);
	MV_DRAIN(ctx);
	for (uint32_t p = 0; p < ctx->numOutSets; ++p)
	{
		MV_HIP(ctx, hipMemsetAsync(ctx->ob[p].cubeMaps, 0, (size_t)ctx->st.V * chain * sizeof(uint2), ctx->stream));
		MV_HIP(ctx, hipMemsetAsync(ctx->ob[p].cubeDepths, 0, (size_t)ctx->st.V * chain * sizeof(float), ctx->stream));
		MV_HIP(ctx, hipMemsetAsync(ctx->ob[p].faceStamps, 0, (size_t)ctx->st.V * MV_NUM_CUBE_MIP * 6 * sizeof(uint32_t), ctx->stream));
	}
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

int mv_get_stats(mv_ctx* ctx, mv_stats* out)
{
	if (!ctx || !out) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemsetAsync(ctx->st.stats, 0, kNumStats * sizeof(uint64_t), ctx->stream));
	mv::launchReduceStats(ctx->stream, ctx->st, 1);
	for (uint32_t p = 0; p < ctx->numOutSets; ++p)
	{
		if (p == ctx->set) continue;
		DeviceState other = ctx->st;
		const OutBufs& q = ctx->ob[p];
		other.waveStats = q.waveStats; other.compositeStats = q.compositeStats;
		mv::launchReduceStats(ctx->stream, other, 0);
	}
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, hipMemcpyAsync(ctx->pinnedStats, ctx->st.stats, kNumStats * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	foldEvents(ctx);
	memset(out, 0, sizeof(*out));
	out->view_samples = ctx->pinnedStats[0];
	out->lit_samples = ctx->pinnedStats[1];
	out->rays = ctx->pinnedStats[2];
	out->light_samples = ctx->pinnedStats[3];
	out->composite_samples = ctx->pinnedStats[4];
	out->reserved[0] = ctx->pinnedStats[5];	// marchV lane slots (64 x longest ray per wavefront)
	out->reserved[2] = ctx->pinnedStats[7];	// view / RayCast samples marched without a texel fetch (their brick's occupancy bit was clear)
	out->reserved[1] = ctx->pinnedStats[6];	// lit samples among composite_samples that were marched inside the view-march launch
	out->ms_cull = ctx->msStage[0]; out->ms_march_light = ctx->msStage[1]; out->ms_march_view = ctx->msStage[2]; out->ms_composite = ctx->msStage[3];
	out->n_cull = ctx->nStage[0]; out->n_march_light = ctx->nStage[1]; out->n_march_view = ctx->nStage[2]; out->n_composite = ctx->nStage[3];
	return MV_OK;
}

int mv_reset_stats(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	foldEvents(ctx);
	MV_HIP(ctx, hipMemsetAsync(ctx->st.stats, 0, kNumStats * sizeof(uint64_t), ctx->stream));
	MV_HIP(ctx, hipMemsetAsync(ctx->st.lightStats, 0, (size_t)ctx->st.lightBlocks * sizeof(uint64_t), ctx->stream));
	for (uint32_t p = 0; p < ctx->numOutSets; ++p)
	{
		const OutBufs& f = ctx->ob[p];
		MV_HIP(ctx, hipMemsetAsync(f.waveStats, 0, (size_t)ctx->st.marchWaves * 8 * sizeof(uint64_t), ctx->stream));
		MV_HIP(ctx, hipMemsetAsync(f.compositeStats, 0, (size_t)ctx->st.compositeSlots * sizeof(uint64_t), ctx->stream));
	}
	for (int i = 0; i < 4; ++i) { ctx->msStage[i] = 0.0; ctx->nStage[i] = 0; }
	return MV_OK;
}

int mv_set_profiling(mv_ctx* ctx, int on)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	ctx->profiling = on != 0;
	return MV_OK;
}

int mv_set_latency_sampling(mv_ctx* ctx, uint32_t every_n_frames)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	if (ctx->latEvery) latencyHarvest(ctx, true);
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i) ctx->latOfSlot[i] = -1;
	for (uint32_t i = 0; i < mv_ctx::kLatencyPool; ++i) { ctx->lat[i].busy = false; ctx->lat[i].open = false; }
	ctx->latEvery = every_n_frames;
	return MV_OK;
}

int mv_get_latency(mv_ctx* ctx, double* mean_ms, double* max_ms, uint64_t* samples, int reset)
{
	if (!ctx || !mean_ms || !max_ms || !samples) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));	// (the frames in flight finish; the host functions behind them have run)
	latencyHarvest(ctx, true);
	*samples = ctx->latCount; *mean_ms = ctx->latCount ? ctx->latSumMs / (double)ctx->latCount : 0.0; *max_ms = ctx->latMaxMs;
	if (reset) { ctx->latSumMs = 0.0; ctx->latMaxMs = 0.0; ctx->latCount = 0; }
	return MV_OK;
}

int mv_host_wait_ms(mv_ctx* ctx, double* ms, int reset)
{
	if (!ctx || !ms) return MV_ERR_INVALID_ARG;
	*ms = ctx->hostWaitMs;
	if (reset) ctx->hostWaitMs = 0.0;
	return MV_OK;
}

int mv_synchronize(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return distCheck(ctx);	// (peer-store transport: a wait for another rank that gave up)
}

void* mv_get_stream(mv_ctx* ctx) { return ctx ? reinterpret_cast<void*>(ctx->stream) : nullptr; }

int mv_get_buffer(mv_ctx* ctx, uint32_t which, uint32_t index, void** d_ptr, size_t* bytes)
{
	if (!ctx || !d_ptr || !bytes) return MV_ERR_INVALID_ARG;
	// pointers of the CURRENT frame parity (with MV_FLAG_PIPELINE_FRAMES the outputs alternate between two buffers)
	const DeviceState& st = ctx->st;
	const size_t chain = mv::cubeChainTexels(st.N);
	const size_t L3 = (size_t)st.L * st.L * st.L;
	switch (which)
	{
	case MV_BUF_VOLUME:
		MV_REQUIRE(ctx, index < st.S, MV_ERR_RANGE, "volume source index out of range");
		*d_ptr = ctx->volumes[index]; *bytes = mv::volBrickCount(st.bricks) * 64 * sizeof(uint2); return MV_OK;
	case MV_BUF_LIGHT_MAP:
		MV_REQUIRE(ctx, index < st.V, MV_ERR_RANGE, "volume index out of range");
		{
			uint4* buf = nullptr;
			const int rc = currentLightBuffer(ctx, index, &buf);	// the buffer the current frame reads (synchronises)
			if (rc != MV_OK) return rc;
			*d_ptr = buf; *bytes = L3 * sizeof(uint4); return MV_OK;
		}
	case MV_BUF_CUBE_MAP:
		MV_REQUIRE(ctx, index < st.V, MV_ERR_RANGE, "volume index out of range");
		*d_ptr = st.cubeMaps + index * chain; *bytes = chain * sizeof(uint2); return MV_OK;
	case MV_BUF_CUBE_DEPTH:
		MV_REQUIRE(ctx, index < st.V, MV_ERR_RANGE, "volume index out of range");
		*d_ptr = st.cubeDepths + index * chain; *bytes = chain * sizeof(float); return MV_OK;
	case MV_BUF_VOLUME_INFO: *d_ptr = st.volumeInfo; *bytes = (size_t)st.V * 8; return MV_OK;
	case MV_BUF_COMPOSITE:	// (inside the multi-GPU frame loop the rows are stored rank-major by tile row: use mv_readback_composite)
		*d_ptr = st.composite; *bytes = (size_t)st.W * ((st.flags & mv::kFlagInterleaveRows) ? compositeRows(st) : st.H) * sizeof(uint2); return MV_OK;
	case MV_BUF_DEPTH_MAP: *d_ptr = ctx->ownDepth; *bytes = (size_t)st.W * st.H * sizeof(float); return MV_OK;
	case MV_BUF_SHADOW_MAP: *d_ptr = ctx->ownShadow; *bytes = (size_t)st.M * st.M * sizeof(uint16_t); return MV_OK;
	case MV_BUF_TILE_COST:
		MV_REQUIRE(ctx, index < ctx->numSlots, MV_ERR_RANGE, "frame slot out of range");
		*d_ptr = ctx->fb[index].tileCost; *bytes = (size_t)mv::tileCostKeyCount(st.V, st.N, st.maxScreenTiles) * sizeof(uint32_t); return MV_OK;
	case MV_BUF_VOLUME_FLAGS: *d_ptr = st.volumeFlags; *bytes = (size_t)st.V * sizeof(uint32_t); return MV_OK;
	case MV_BUF_EXCHANGE_OFFSETS: *d_ptr = st.exchangeOff; *bytes = (size_t)st.V * sizeof(uint32_t); return MV_OK;
	case MV_BUF_MARCH_TIMELINE: *d_ptr = st.timeline; *bytes = (size_t)st.marchWaves * 8 * sizeof(uint64_t); return MV_OK;
	default: return fail(ctx, MV_ERR_INVALID_ARG, "unknown buffer id");
	}
}

int mv_readback_cull(mv_ctx* ctx, uint16_t* volume_info, uint32_t* visible, uint32_t* visibleCount,
	uint32_t* cubeMapVolumes, uint32_t* cubeMapCount, uint32_t* minMip)
{
	if (!ctx || !volume_info || !visible || !visibleCount || !cubeMapVolumes || !cubeMapCount || !minMip) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	// The copies run behind the cull on stream A and only that stream is waited for: with frames in flight the host then
	// waits for this frame's cull (which runs ahead), not for the marches of the frames before it.
	const size_t V = ctx->st.V;
	uint8_t* p = ctx->pinnedCull;
	MV_HIP(ctx, hipMemcpyAsync(p, ctx->fb[ctx->slot].cullBlock, 64 + V * 24, hipMemcpyDeviceToHost, ctx->streamA));	// one copy (FrameBufs::cullBlock)
	MV_HIP(ctx, hipStreamSynchronize(ctx->streamA));
	ctx->lightTableFrame = ctx->frame;
	const uint32_t* counters = reinterpret_cast<const uint32_t*>(p);
	*visibleCount = counters[0]; *cubeMapCount = counters[1]; *minMip = counters[5];
	memcpy(volume_info, p + 64, V * 8);
	memcpy(visible, p + 64 + V * 8, V * 4);
	memcpy(cubeMapVolumes, p + 64 + V * 12, V * 4);
	return MV_OK;
}

} // extern "C"
