// mv_api.cpp -- the C-ABI of include/multivolumes.h: context, device memory, stream-ordered stage launches.
//
// Host-side counterpart of MultiRayCaster (MultiVolumes/Content/MultiRayCaster.{h,cpp}): where the reference records
// D3D12 dispatches into an XUSG command list, this layer launches the gfx950 kernels of mv_kernels.hip on one HIP stream.
// Kernel order on the stream = the reference's barrier graph (cull -> marchL -> marchV -> composite).
// There is no CPU fallback: without a gfx950 device every entry point fails.
#include "../../include/multivolumes.h"
#include "mv_kernels.hpp"
#include "mv_device.hpp"
#include "mv_mesh.hpp"

#include <algorithm>
#include <chrono>

#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <new>
#include <dlfcn.h>
#include <rccl/rccl.h>	// types and prototypes only: the library is opened with dlopen when mv_dist_init_rccl is called

using mv::DeviceState;

#ifdef MV_HOST_PROFILE	// (diagnostic builds: where the host's time inside mv_dist_frame goes; printed by mv_dist_shutdown / mv_destroy)
static double gSection[12]; static uint64_t gSectionCalls;
struct SectionTimer { int i; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); ~SectionTimer() { gSection[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); } };
#define MV_SECTION(i) SectionTimer section_##i{ i }
#else
#define MV_SECTION(i)
#endif


namespace
{
constexpr uint32_t kMaxScreenSlots = 32;	// screen-space-scheme volumes whose RayCast joins the view-march launch (16 B x W x H of layer each, per output set)
// Frames in flight (MV_FLAG_PIPELINE_FRAMES; mv_config.frames_in_flight, default 3 like the reference's FrameCount, MultiRayCaster.h:52): C whole
// frames on C view streams with C output sets.  What follows from C: light buffers per volume = max(3, C) (the light march of frame i
// writes the buffer the relit volume held that many relights ago: frame i - lightBufs is the last that can still read it), frame slots
// (constants + cull outputs; also the pinned upload slices) = light buffers + 1 (the cull of frame i + 1 runs while frame i - lightBufs + 1
// is still marching).  With C = 3: 3 light buffers, 4 slots -- the numbers of rounds 1 and 2.
constexpr uint32_t kMaxChains = 5;
constexpr uint32_t kMaxFrameSlots = kMaxChains + 1;
constexpr uint32_t kNumCounters = 8;	// read back; counters[8] (the cull's ticket) is device-only
constexpr uint32_t kNumStats = 8;

struct EventPair { hipEvent_t a, b; int stage; };
}

// Buffers that exist once per frame slot (constants + everything the cull writes): frame i+1's upload / cull / light march
// (stream A) may then run while frame i's view march / composite (stream B) still read frame i's copies.
struct FrameBufs
{
	void* dCb = nullptr;
	void* dPerObject = nullptr;
	uint8_t* cullBlock = nullptr;		// counters | VolumeInfo | visible | cube-map list | light table, contiguous
	uint16_t* volumeInfo = nullptr;
	uint32_t *visibleVolumes = nullptr, *cubeMapVolumes = nullptr, *counters = nullptr, *ownedVisible = nullptr;
	uint32_t *workEntries = nullptr, *workTileStart = nullptr;
	uint64_t* cubeOffset = nullptr;
	uint32_t* volumeFlags = nullptr;
	uint16_t* volumeRects = nullptr;
	float4* lightDirs = nullptr;
	uint32_t* screenSlot = nullptr;
	uint32_t* exchangeOff = nullptr;
	uint4** lightTable = nullptr;
	uint32_t* tileCost = nullptr;
	uint32_t* lightOrder = nullptr;	// dispatch order of the light march's voxel blocks (launchLightOrder), per slot
	uint32_t *tileOrder = nullptr, *binCounts = nullptr;	// dispatch order of the view march (launchTileOrder), per slot
	mv::VolumeConsts* volConsts = nullptr;
	uint32_t* queueHeads = nullptr;
};

// Outputs of the view march / composite and their counters: one set, or kMaxOutSets-or-fewer rotating sets with
// MV_FLAG_PIPELINE_FRAMES (frame i uses set i % numOutSets and view stream i % numOutSets), so that the view marches of
// consecutive frames may overlap on the GPU.
constexpr uint32_t kMaxOutSets = kMaxChains;
struct OutBufs
{
	uint2* cubeMaps = nullptr;
	float* cubeDepths = nullptr;
	uint2* composite = nullptr;
	float4* screenLayers = nullptr;
	uint32_t* faceStamps = nullptr;
	uint64_t *waveStats = nullptr, *compositeStats = nullptr;
	uint32_t* compositeRetry = nullptr;
	mv::RayRecord* rayRecords = nullptr;	// the view march's ray queue (mv_march.hip)
	mv::TileInfo* tileInfo = nullptr;
};

// state of the in-library multi-GPU frame loop (mv_dist_*)
constexpr uint32_t kDistChains = kMaxChains;	// (array sizes; the loop uses as many chains as the context has view streams)
struct DistState
{
	int transport = 0;				// 1 = RCCL, 2 = caller-supplied collectives
	void* rcclLib = nullptr;
	decltype(&ncclCommInitRank) commInitRank = nullptr;
	decltype(&ncclAllGather) allGather = nullptr;
	decltype(&ncclAllReduce) allReduce = nullptr;
	decltype(&ncclCommDestroy) commDestroy = nullptr;
	decltype(&ncclGetErrorString) errorString = nullptr;
	// Two ways to run the loop (chosen by how the context was created):
	// * chains (MV_FLAG_PIPELINE_FRAMES alone: three view streams, three output sets): frame k's light march, light-map exchange, view
	//   march, cube-face exchange, composite and row gather all go on view stream k % 3, with communicator / buffers k % 3 -- whole
	//   frames overlap, which is what a rank needs once its share of a frame no longer fills the GPU;
	// * single view stream (| MV_FLAG_SINGLE_VIEW_STREAM: two output sets): the view marches on the context's stream, the exchange and
	//   composite of frame k - 1 beside frame k's march on stream X; comm [0] = cube faces + row bands, [1] = light maps.
	bool chains = false;
	bool cullCrosscheck = true;	// the loop compares the ranks' culls through the light exchange's trailers (mv_dist_set_cull_crosscheck)
	bool inPlaceGather = false;	// the transport's all-gather accepts send == recv + rank * bytes (RCCL does; a caller says so with mv_dist_set_inplace_gather)
	uint32_t numComm = 2;
	uint32_t numChains = 1;			// chains: one per view stream of the context (3 by default), each with its communicator and buffers; else 1
	ncclComm_t comm[kDistChains] = {};
	mv_dist_allgather_fn cbAllGather = nullptr;
	mv_dist_allreduce_u32_fn cbAllReduce = nullptr;
	void* user = nullptr;
	hipStream_t streamX = nullptr;	// (single view stream) exchange + composite of frame k - 1, beside frame k's view march
	uint8_t *sendBuf[kDistChains] = {}, *recvBuf[kDistChains] = {};	// one shard / shard_count shards of marched cube faces ([0] only without chains)
	size_t sendCap = 0;
	uint32_t* lightWords[kDistChains] = {};	// the relit light map as every rank's share of its voxel blocks: shard_count x lightShareBytes (plain R11G11B10F words, 64 per block)
	uint8_t* lightSend[kDistChains] = {};	// this rank's share when the transport cannot gather in place
	size_t lightShareBytes = 0;
	uint8_t *rowSend[kDistChains] = {}, *rowRecv[kDistChains] = {};	// staged row-band gather (unequal bands, or a transport without in-place all-gather)
	size_t maxBandBytes = 0;
	unsigned long long* reports = nullptr;	// pinned, one word per frame slot: {frame tag : texels of the largest exchange shard}, written by the cull
	bool pending = false, pendingComposite = false;
	DeviceState stPending;			// the device state of the frame whose exchange + composite are still to be enqueued
	uint32_t slotPending = 0, chainPending = 0;
	uint64_t bytes = 0;
	std::vector<void*> allocs;		// device buffers of the loop (freed by distRelease: a shutdown followed by another init allocates afresh)
};

struct mv_ctx
{
	mv_config cfg;
	hipStream_t stream = nullptr;		// stream B: view march, composite, exchange, producers, readbacks
	hipStream_t streamA = nullptr;		// stream A: constant upload, cull (== stream unless MV_FLAG_PIPELINE_FRAMES); may run frames ahead
	hipStream_t streamL = nullptr;		// stream L: light march (== stream unless MV_FLAG_PIPELINE_FRAMES)
	hipStream_t streamB[kMaxOutSets] = {};	// view march + composite of frame i on streamB[i % numOutSets] ([0] == stream; the others == stream unless pipelined)
	OutBufs ob[kMaxOutSets];
	uint32_t numOutSets = 1, numViewStreams = 1;
	uint32_t set = 0;
	bool ownStream = false, ownStreamA = false, ownStreamL = false, ownStreamB1 = false;
	bool lightOnViewStream = false;		// pipelined with several view streams: a frame's light march runs on that frame's view stream
	DeviceState st;						// always holds the CURRENT frame slot's pointers (applyParity)
	FrameBufs fb[kMaxFrameSlots];		// constants + cull outputs per slot (numSlots of them in use)
	uint32_t numSlots = 4;				// light buffers per volume + 1
	uint32_t slot = 0;					// frame counter mod numSlots
	hipEvent_t evStageA[kMaxFrameSlots] = {}, evLight[kMaxFrameSlots] = {}, evMainDone[kMaxFrameSlots] = {}, evProducers = nullptr;
	hipEvent_t evCopy[kMaxFrameSlots] = {};	// a frame's unmarched cube faces are up to date (copyForward)
	bool copiedForward = false;			// this frame's copy-forward has been enqueued
	bool marchedThisFrame = false;		// mv_march_view has run for the current frame (its RayCast layers are this frame's)
	uint64_t lastCompleteFrame = 0;		// the last frame whose output set was completed (view march, or a composite-only frame's copy from every set)
	bool producersDirty = false;

	// owned device allocations
	std::vector<uint2*> volumes;		// volumes[i] = volPool + i * volStride: ONE allocation (the march addresses it as base + 32-bit offset)
	uint8_t* volPool = nullptr;
	uint64_t* occPool = nullptr;
	std::vector<uint4*> cells;
	uint2** dVolumePtrs = nullptr;
	uint4** dCellPtrs = nullptr;
	std::vector<uint64_t*> occupancy;	// brick occupancy words per volume source
	uint64_t** dOccPtrs = nullptr;
	uint32_t* lightStaging = nullptr;	// L^3 plain R11G11B10F (readback / upload conversion)
	bool orderTiles = true;				// false with MV_FLAG_LIST_ORDER
	// mesh depth / shadow producer (mv_mesh_*)
	float* meshPositions = nullptr;
	uint32_t* meshIndices = nullptr;
	uint32_t meshVerts = 0, meshTris = 0;
	uint32_t* rasterBits = nullptr;		// max(W x H, M x M) depth bits
	float* dMeshMatrix = nullptr;		// 16 floats
	uint2* staging = nullptr;			// N^3 linear RGBA16F (upload / readback re-tiling)
	size_t stagingBytes = 0;
	float* ownDepth = nullptr;
	uint16_t* ownShadow = nullptr;
	uint32_t* dDescs = nullptr;
	float* dSh = nullptr;
	std::vector<void*> allocs;			// everything to hipFree

	// pinned upload ring (the reference's upload-heap slices, MultiRayCaster.cpp:128-130, 461-467)
	uint8_t* pinned = nullptr;			// numSlots slices: frame i's constants in slice i % numSlots, read by that frame's cull where they lie
	size_t sliceBytes = 0;
	uint64_t* pinnedStats = nullptr;
	uint32_t* pinnedCounters = nullptr;
	unsigned long long* pinnedTiles = nullptr;	// written by the cull kernel: {frame tag : tile count} of the most recent cull that has run (a hint, never waited for)
	bool rayQueue = false;				// MV_FLAG_RAY_QUEUE: the two-stage ray-queue march (mv_march.hip) instead of one wavefront per tile
	uint8_t* pinnedCull = nullptr;		// counters + VolumeInfo + both lists + light table, one readback
	uint64_t frame = 0;					// frames begun (mv_update_frame)
	uint64_t lightTableFrame = ~0ull;	// frame whose light table (V device pointers) sits at the end of pinnedCull

	bool haveDepth = false, haveShadow = false, haveFrame = false, haveCull = false;
	bool leanEvents = false;			// inside mv_dist_frame on a chains context: event records that a later record of the same call supersedes are left out (the host enqueues a rank's frame in ~100 us: every call counts)
	bool lightMarched = false, lightTableKept = false;	// this frame: mv_march_light ran / the frame keeps the previous frame's light table (see keepLightTable)
	bool profiling = false;
	std::vector<EventPair> pending;
	std::vector<EventPair> freeEvents;
	double msStage[4] = { 0, 0, 0, 0 };
	uint64_t nStage[4] = { 0, 0, 0, 0 };
	std::string lastError;
	DistState* dist = nullptr;
	// Cross-stream ordering goes through evRecord / evWait below, which leave out the waits that are redundant: an event recorded on the
	// very stream that is asked to wait for it (stream order already says so), and a wait the stream has already made for the same
	// record.  With whole frames on three view streams frame k - 3 sits on frame k's stream, so half of a frame's waits are of the
	// first kind (a stream wait is ~1 us of host time on a pending event; a rank of the multi-GPU loop issued sixteen per frame).
	struct EvTrack { hipEvent_t e = nullptr; hipStream_t on = nullptr; uint64_t seq = 0; uint64_t waited[8] = {}; };
	EvTrack evTrack[24];
	hipStream_t evStreams[8] = {};
	uint64_t evSeq = 0;
	double hostWaitMs = 0.0;			// time mv_update_frame spent waiting for the GPU to free a frame slot (mv_host_wait_ms)
};

namespace
{
int fail(mv_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
	if (ctx)
	{
		char buf[512];
		if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s (%s)", what, hipGetErrorName(e), hipGetErrorString(e));
		else snprintf(buf, sizeof(buf), "%s", what);
		ctx->lastError = buf;
	}
	return code;
}

#define MV_HIP(ctx, call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? MV_ERR_OUT_OF_MEMORY : MV_ERR_HIP, #call, e_); } while (0)
#define MV_REQUIRE(ctx, cond, code, msg) do { if (!(cond)) return fail(ctx, code, msg); } while (0)

template <typename T>
int devAlloc(mv_ctx* ctx, T** out, size_t bytes, bool zero = true)
{
	void* p = nullptr;
	MV_HIP(ctx, hipMalloc(&p, bytes ? bytes : 16));
	ctx->allocs.push_back(p);
	if (zero) MV_HIP(ctx, hipMemsetAsync(p, 0, bytes ? bytes : 16, ctx->stream));
	*out = reinterpret_cast<T*>(p);
	return MV_OK;
}

// replace a buffer devAlloc made by one of another size (mv_set_viewport)
template <typename T>
int devRealloc(mv_ctx* ctx, T** p, size_t bytes, bool zero = true)
{
	void* fresh = nullptr;
	MV_HIP(ctx, hipMalloc(&fresh, bytes ? bytes : 16));
	if (zero) MV_HIP(ctx, hipMemsetAsync(fresh, 0, bytes ? bytes : 16, ctx->stream));
	for (void*& a : ctx->allocs)
		if (a == static_cast<void*>(*p)) { (void)hipFree(a); a = fresh; *p = reinterpret_cast<T*>(fresh); return MV_OK; }
	ctx->allocs.push_back(fresh);
	*p = reinterpret_cast<T*>(fresh);
	return MV_OK;
}

// rows of the composite buffer: H, or the padded rank-major layout (kFlagInterleaveRows)
static size_t compositeRows(const DeviceState& st)
{
	const uint32_t sc = st.shardCount ? st.shardCount : 1u;
	const size_t dealt = (size_t)sc * mv::interleavedRowsPerRank(st.H, sc) * 8u;
	return sc > 1u && dealt > st.H ? dealt : st.H;
}

// everything of DeviceState that follows from W x H (mv_create, mv_set_viewport)
void viewportDerived(DeviceState& st)
{
	st.rowsPerRank = mv::interleavedRowsPerRank(st.H, st.shardCount);
	// Rows this shard composites (multivolumes_amd/distributed.py::row_range states the same rule): whole 8-row tiles, split evenly.
	{
		const uint32_t tiles = (st.H + 7u) >> 3;
		const uint32_t t0 = (uint32_t)(((uint64_t)tiles * st.shardIndex) / st.shardCount), t1 = (uint32_t)(((uint64_t)tiles * (st.shardIndex + 1u)) / st.shardCount);
		st.bandRow0 = t0 * 8u < st.H ? t0 * 8u : st.H;
		st.bandRow1 = t1 * 8u < st.H ? t1 * 8u : st.H;
	}
	// Screen-space-scheme volumes are marched inside the view-march launch (up to kMaxScreenSlots of them), by every shard for ITS rows
	{
		uint32_t cap = kMaxScreenSlots;
#ifdef MV_DIAGNOSTICS
		{ const char* e = getenv("MV_SCREEN_SLOTS"); if (e) cap = (uint32_t)atoi(e); }
#endif
		st.screenSlots = st.V < cap ? st.V : cap;
	}
	{
		// tile rows this context can own: its contiguous band, or (multi-GPU frame loop) every shard_count-th tile row
		const uint32_t bandTiles = (st.bandRow1 - st.bandRow0 + 7u) >> 3, dealt = mv::interleavedRowsPerRank(st.H, st.shardCount ? st.shardCount : 1u);
		st.maxScreenTiles = st.screenSlots * ((st.W + 7u) >> 3) * (bandTiles > dealt ? bandTiles : dealt);
	}
	st.compositeSlots = ((st.W + 7u) >> 3) * ((st.H + 7u) >> 3);
}

void applyParity(mv_ctx* ctx)
{
	const FrameBufs& f = ctx->fb[ctx->slot];
	const FrameBufs& q = ctx->fb[(ctx->slot + ctx->numSlots - 1u) % ctx->numSlots];	// the previous frame's
	DeviceState& st = ctx->st;
	st.cbPerFrame = f.dCb; st.perObject = f.dPerObject;
	st.exchangeReport = (ctx->dist && ctx->dist->reports) ? ctx->dist->reports + (size_t)ctx->slot * 8u : nullptr;
	st.volumeInfo = f.volumeInfo; st.visibleVolumes = f.visibleVolumes; st.cubeMapVolumes = f.cubeMapVolumes;
	st.counters = f.counters; st.ownedVisible = f.ownedVisible; st.workEntries = f.workEntries; st.workTileStart = f.workTileStart;
	st.cubeOffset = f.cubeOffset; st.volumeFlags = f.volumeFlags; st.volumeRects = f.volumeRects; st.screenSlot = f.screenSlot; st.exchangeOff = f.exchangeOff; st.lightDirs = f.lightDirs; st.lightOrder = f.lightOrder;
	st.lightTable = f.lightTable; st.lightTablePrev = q.lightTable; st.volumeInfoPrev = q.volumeInfo; st.volConsts = f.volConsts; st.queueHeads = f.queueHeads;
	const OutBufs& o = ctx->ob[ctx->set];
	st.cubeMaps = o.cubeMaps; st.cubeDepths = o.cubeDepths; st.composite = o.composite; st.screenLayers = o.screenLayers; st.faceStamps = o.faceStamps;
	st.waveStats = o.waveStats; st.compositeStats = o.compositeStats; st.compositeRetry = o.compositeRetry; st.tileCost = f.tileCost;
	st.rayRecords = o.rayRecords; st.tileInfo = o.tileInfo;
	st.tileOrder = ctx->orderTiles ? f.tileOrder : nullptr;
	st.binCursor = ctx->orderTiles ? f.binCounts : nullptr;
	st.orderCap = st.maxTiles + st.maxScreenTiles + 1u;
}

int syncAll(mv_ctx* ctx)
{
	if (ctx->streamA != ctx->stream)
	{
		const hipError_t e = hipStreamSynchronize(ctx->streamA);
		if (e != hipSuccess) return (int)e;
	}
	if (ctx->streamL && ctx->streamL != ctx->stream)
	{
		const hipError_t e = hipStreamSynchronize(ctx->streamL);
		if (e != hipSuccess) return (int)e;
	}
	for (uint32_t i = 1; i < kMaxOutSets; ++i)
		if (ctx->streamB[i] && ctx->streamB[i] != ctx->stream)
		{
			const hipError_t e = hipStreamSynchronize(ctx->streamB[i]);
			if (e != hipSuccess) return (int)e;
		}
	if (ctx->dist && ctx->dist->streamX)
	{
		const hipError_t e = hipStreamSynchronize(ctx->dist->streamX);
		if (e != hipSuccess) return (int)e;
	}
	return (int)hipStreamSynchronize(ctx->stream);
}

mv_ctx::EvTrack* evFind(mv_ctx* ctx, hipEvent_t e)
{
	for (auto& t : ctx->evTrack) { if (t.e == e) return &t; if (!t.e) { t.e = e; return &t; } }
	return nullptr;	// (more events than slots: untracked, every wait is issued)
}
int evStreamIndex(mv_ctx* ctx, hipStream_t s)
{
	for (int i = 0; i < 8; ++i) { if (ctx->evStreams[i] == s) return i; if (!ctx->evStreams[i]) { ctx->evStreams[i] = s; return i; } }
	return -1;
}
void evForget(mv_ctx* ctx)	// streams were destroyed or created: nothing recorded so far may be relied on by address
{
	for (auto& t : ctx->evTrack) { t.on = nullptr; for (auto& w : t.waited) w = 0; }
	for (auto& x : ctx->evStreams) x = nullptr;
}
hipError_t evRecord(mv_ctx* ctx, hipEvent_t e, hipStream_t s)
{
	const hipError_t r = hipEventRecord(e, s);
	if (mv_ctx::EvTrack* t = evFind(ctx, e)) { t->on = r == hipSuccess ? s : nullptr; t->seq = ++ctx->evSeq; }
	return r;
}
hipError_t evWait(mv_ctx* ctx, hipStream_t s, hipEvent_t e)
{
	mv_ctx::EvTrack* t = evFind(ctx, e);
	const int si = evStreamIndex(ctx, s);
	if (t && si >= 0)
	{
		if (t->seq == 0) return hipSuccess;				// never recorded: nothing to wait for
		if (t->on == s) return hipSuccess;				// recorded on this stream: stream order
		if (t->waited[si] == t->seq) return hipSuccess;	// this stream has already waited for this record
		t->waited[si] = t->seq;
	}
	return hipStreamWaitEvent(s, e, 0);
}

// stage of stream A (cull, light march) finished being enqueued: stream B stages of this frame wait for it
// (every one of these orders streams against each other: a failure must reach the caller, or a later stage reads unfinished data)
hipError_t stageADone(mv_ctx* ctx, bool last = true) { return ctx->leanEvents && !last ? hipSuccess : evRecord(ctx, ctx->evStageA[ctx->slot], ctx->streamA); }
hipStream_t streamB(mv_ctx* ctx) { return ctx->streamB[ctx->frame % ctx->numViewStreams]; }
hipError_t stageBBegin(mv_ctx* ctx)
{
	const hipError_t e = evWait(ctx, streamB(ctx), ctx->evStageA[ctx->slot]);
	if (e != hipSuccess) return e;
	// this frame's light march, if one has been enqueued (and, when light marches run on their frames' view streams, those of the
	// frames before it on the OTHER view streams: a view march reads every visible volume's light map)
	const uint32_t others = ctx->lightOnViewStream ? ctx->numViewStreams - 1u : 0u;
	for (uint32_t j = 0; j <= others; ++j)
	{
		const hipError_t e2 = evWait(ctx, streamB(ctx), ctx->evLight[(ctx->slot + ctx->numSlots - j) % ctx->numSlots]);
		if (e2 != hipSuccess) return e2;
	}
	return hipSuccess;
}
// stage of stream B finished being enqueued: the next user of this slot's buffers (ctx->numSlots frames on) and the light march two frames on wait for it
hipError_t stageBDone(mv_ctx* ctx) { return evRecord(ctx, ctx->evMainDone[ctx->slot], streamB(ctx)); }

void beginStage(mv_ctx* ctx, int stage, EventPair& ev, hipStream_t stream)
{
	if (!ctx->profiling) return;
	// timing events only (never used for ordering): a failure here loses one timing sample, nothing else
	ev.a = ev.b = nullptr;
	if (!ctx->freeEvents.empty()) { ev = ctx->freeEvents.back(); ctx->freeEvents.pop_back(); }
	else if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) { ev.a = ev.b = nullptr; return; }
	ev.stage = stage;
	if (hipEventRecord(ev.a, stream) != hipSuccess) { ctx->freeEvents.push_back(ev); ev.a = ev.b = nullptr; }
}

void endStage(mv_ctx* ctx, EventPair& ev, hipStream_t stream)
{
	if (!ctx->profiling || !ev.a) return;
	if (hipEventRecord(ev.b, stream) != hipSuccess) { ctx->freeEvents.push_back(ev); return; }
	ctx->pending.push_back(ev);
}

// Inside the multi-GPU frame loop the exchange, composite and row gather of the last frame begun are still to be enqueued (they go out
// with the next mv_dist_frame).  Anything that is about to overwrite an input that tail reads -- depth / shadow map, SH, volumes, light
// maps, VolumeDesc -- enqueues it first, then waits for every stream: the pending frame is composited against the inputs it was begun
// with.  (The tail contains collectives: inside the loop these setters are collective calls, every rank makes them at the same point.)
int distTail(mv_ctx* ctx);
int drainForWrite(mv_ctx* ctx)
{
	if (ctx->dist && ctx->dist->pending)
	{
		const int rc = distTail(ctx);
		if (rc != MV_OK) return rc;
	}
	const hipError_t e = (hipError_t)syncAll(ctx);
	return e == hipSuccess ? MV_OK : fail(ctx, MV_ERR_HIP, "synchronising the frames in flight", e);
}
#define MV_DRAIN(ctx) do { const int rc_ = drainForWrite(ctx); if (rc_ != MV_OK) return rc_; } while (0)

void foldEvents(mv_ctx* ctx)
{
	for (EventPair& ev : ctx->pending)
	{
		(void)hipEventSynchronize(ev.b);
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess)
		{
			ctx->msStage[ev.stage] += ms;
			ctx->nStage[ev.stage] += 1;
		}
		ctx->freeEvents.push_back(ev);
	}
	ctx->pending.clear();
}

} // namespace

// More than three whole frames in flight need more than the four hardware queues the HIP runtime gives a process by default (streams
// that share a queue wait for each other: scripts/micro/hw_queues.cpp): when this library is loaded before the runtime has started --
// a C++ host that links it, a Python process that imports it before touching the GPU -- it asks for eight, unless the caller has said
// otherwise.  A process whose runtime is already up keeps what it has; mv_config.frames_in_flight > 3 then shares queues (slower, correct).
__attribute__((constructor)) static void mvAskForHardwareQueues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

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
	case MV_ERR_RANGE: return "index out of range";
	default: return "unknown error";
	}
}

const char* mv_last_error(const mv_ctx* ctx) { return ctx ? ctx->lastError.c_str() : ""; }
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
		if (hipStreamCreateWithPriority(&ctx->streamA, hipStreamNonBlocking, prHigh) != hipSuccess) return bail(MV_ERR_HIP);
		ctx->ownStreamA = true;
		// One view stream (MV_FLAG_SINGLE_VIEW_STREAM, the multi-GPU loop): the light marches get a stream of their own and two
		// output sets rotate.  Otherwise C output sets on C view streams (mv_config.frames_in_flight, default 3), each frame's light
		// march on its view stream.  A HIP process has four hardware queues by default, and streams that share one wait for each
		// other: with three chains that is four streams; more chains need GPU_MAX_HW_QUEUES >= C + 1 in the environment before the HIP
		// runtime starts (this library sets it to 8 when it is loaded, if the variable is not set -- see the constructor below).
		if (cfg->flags & MV_FLAG_SINGLE_VIEW_STREAM)
		{
			if (hipStreamCreateWithPriority(&ctx->streamL, hipStreamNonBlocking, prHigh) != hipSuccess) return bail(MV_ERR_HIP);
			ctx->ownStreamL = true;
			ctx->numOutSets = 2u;
		}
		else
		{
			ctx->lightOnViewStream = true;
			ctx->numOutSets = wantChains;
		}
#ifdef MV_DIAGNOSTICS	// A/B measurements only (csrc/build.sh -DMV_DIAGNOSTICS); the shipped library reads no environment variable
		{ const char* e = getenv("MV_OUT_SETS"); if (e) ctx->numOutSets = (uint32_t)atoi(e); if (ctx->numOutSets < 2u || ctx->numOutSets > kMaxOutSets) ctx->numOutSets = 2u; }
#endif
		if (!(cfg->flags & MV_FLAG_SINGLE_VIEW_STREAM))
		{
			ctx->numViewStreams = ctx->numOutSets;
			for (uint32_t i = 1; i < ctx->numViewStreams; ++i)
				if (hipStreamCreateWithFlags(&ctx->streamB[i], hipStreamNonBlocking) != hipSuccess) return bail(MV_ERR_HIP);
			ctx->ownStreamB1 = true;
		}
	}
	ctx->st.lightBufs = ctx->numOutSets > 3u ? ctx->numOutSets : 3u;
	ctx->numSlots = ctx->st.lightBufs + 1u;
	for (uint32_t i = 0; i < ctx->numSlots; ++i)
		if (hipEventCreateWithFlags(&ctx->evStageA[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->evMainDone[i], hipEventDisableTiming) != hipSuccess
			|| hipEventCreateWithFlags(&ctx->evLight[i], hipEventDisableTiming) != hipSuccess) return bail(MV_ERR_HIP);
	if (hipEventCreateWithFlags(&ctx->evProducers, hipEventDisableTiming) != hipSuccess) return bail(MV_ERR_HIP);
	for (uint32_t i = 0; i < ctx->numSlots; ++i)
		if (hipEventCreateWithFlags(&ctx->evCopy[i], hipEventDisableTiming) != hipSuccess) return bail(MV_ERR_HIP);

	DeviceState& st = ctx->st;
	st.N = cfg->grid_size; st.bricks = (cfg->grid_size + 3u) >> 2; st.L = cfg->light_grid_size;
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
	if ((rc = devAlloc(ctx, &st.lightBuffers, (size_t)st.V * st.lightBufs * L3 * sizeof(uint4))) != MV_OK) return bail(rc);
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
			if ((rc = devAlloc(ctx, &f.dCb, 256 + (size_t)st.V * MV_PER_OBJECT_BYTES)) != MV_OK) return bail(rc);	// one upload: CBPerFrame, then PerObject[V]
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
		if ((rc = devAlloc(ctx, &ctx->fb[p].tileOrder, 8u * ((size_t)st.maxTiles + st.maxScreenTiles + 1) * sizeof(uint32_t))) != MV_OK) return bail(rc);	// a region per cost bin
		if ((rc = devAlloc(ctx, &ctx->fb[p].binCounts, 8 * sizeof(uint32_t))) != MV_OK) return bail(rc);
		// per-tile cost table, one per slot: the view march of frame i writes it, the tile order of frame i + 4 reads it
		if ((rc = devAlloc(ctx, &ctx->fb[p].tileCost, (size_t)mv::tileCostKeyCount(st.V, st.N, st.maxScreenTiles) * sizeof(uint32_t))) != MV_OK) return bail(rc);
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
		if ((rc = devAlloc(ctx, &f.cubeMaps, (size_t)st.V * chain * sizeof(uint2))) != MV_OK) return bail(rc);
		if ((rc = devAlloc(ctx, &f.cubeDepths, (size_t)st.V * chain * sizeof(float))) != MV_OK) return bail(rc);
		// (sharded: room for the rank-major layout of the multi-GPU frame loop, shard_count blocks of ceil(tile rows / shard_count) tile rows)
		if ((rc = devAlloc(ctx, &f.composite, (size_t)st.W * compositeRows(st) * sizeof(uint2))) != MV_OK) return bail(rc);
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
	if (hipHostMalloc(reinterpret_cast<void**>(&ctx->pinnedTiles), 64, hipHostMallocCoherent) != hipSuccess) return bail(MV_ERR_OUT_OF_MEMORY);
	*ctx->pinnedTiles = ~0ull;
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
	*out = ctx;
	return MV_OK;
}

// Detach the multi-GPU frame loop: communicators, its stream, the pinned cull copies (device buffers stay in ctx->allocs until
// mv_destroy; the RCCL library stays loaded).
static void distRelease(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (!d) return;
	for (uint32_t i = 0; i < kDistChains; ++i) if (d->comm[i] && d->commDestroy) (void)d->commDestroy(d->comm[i]);
	if (d->streamX) (void)hipStreamDestroy(d->streamX);
	evForget(ctx);
	if (d->reports) (void)hipHostFree(d->reports);
	for (void* p : d->allocs) (void)hipFree(p);
	delete d;
	ctx->dist = nullptr;
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
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i) if (ctx->evCopy[i]) (void)hipEventDestroy(ctx->evCopy[i]);
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
	st.W = w; st.H = h;
	ctx->cfg.viewport_w = w; ctx->cfg.viewport_h = h;
	viewportDerived(st);
	int rc;
	if ((rc = devRealloc(ctx, &ctx->ownDepth, (size_t)w * h * sizeof(float))) != MV_OK) return rc;
	st.depthMap = ctx->ownDepth;	// (a borrowed depth map was the old viewport's: the caller hands over the new one)
	ctx->haveDepth = false;
	for (uint32_t p = 0; p < ctx->numSlots; ++p)
	{
		if ((rc = devRealloc(ctx, &ctx->fb[p].tileOrder, 8u * ((size_t)st.maxTiles + st.maxScreenTiles + 1) * sizeof(uint32_t))) != MV_OK) return rc;
		if ((rc = devRealloc(ctx, &ctx->fb[p].tileCost, (size_t)mv::tileCostKeyCount(st.V, st.N, st.maxScreenTiles) * sizeof(uint32_t))) != MV_OK) return rc;
	}
	for (uint32_t p = 0; p < ctx->numOutSets; ++p)
	{
		OutBufs& f = ctx->ob[p];
		if ((rc = devRealloc(ctx, &f.composite, (size_t)st.W * compositeRows(st) * sizeof(uint2))) != MV_OK) return rc;
		if ((rc = devRealloc(ctx, &f.screenLayers, (size_t)st.screenSlots * st.W * st.H * sizeof(float4))) != MV_OK) return rc;
		if ((rc = devRealloc(ctx, &f.compositeStats, (size_t)st.compositeSlots * sizeof(uint64_t))) != MV_OK) return rc;
		if ((rc = devRealloc(ctx, &f.compositeRetry, ((size_t)st.compositeSlots + 2u) * sizeof(uint32_t))) != MV_OK) return rc;
		if (ctx->rayQueue)
		{
			if ((rc = devRealloc(ctx, &f.rayRecords, ((size_t)st.maxTiles + st.maxScreenTiles) * 64 * sizeof(mv::RayRecord), false)) != MV_OK) return rc;
			if ((rc = devRealloc(ctx, &f.tileInfo, ((size_t)st.maxTiles + st.maxScreenTiles) * sizeof(mv::TileInfo), false)) != MV_OK) return rc;
		}
	}
	if (ctx->rasterBits)
	{
		(void)hipFree(ctx->rasterBits); ctx->rasterBits = nullptr;
		MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rasterBits), std::max((size_t)w * h, (size_t)st.M * st.M) * sizeof(uint32_t)));
	}
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
static int keepLightTable(mv_ctx* ctx)
{
	if (ctx->lightMarched || ctx->lightTableKept) return MV_OK;
	const FrameBufs& f = ctx->fb[ctx->slot];
	const FrameBufs& q = ctx->fb[(ctx->slot + ctx->numSlots - 1u) % ctx->numSlots];
	MV_HIP(ctx, hipMemcpyAsync(f.lightTable, q.lightTable, (size_t)ctx->st.V * sizeof(uint4*), hipMemcpyDeviceToDevice, ctx->streamA));
	MV_HIP(ctx, stageADone(ctx));
	ctx->lightTableKept = true;
	return MV_OK;
}

int mv_update_frame(mv_ctx* ctx, const void* cbPerFrame, const void* perObject)
{
	if (!ctx || !cbPerFrame || !perObject) return MV_ERR_INVALID_ARG;
	// a new frame: the next slot of constants / cull buffers (and the other parity's march outputs)
	ctx->slot = (ctx->slot + 1u) % ctx->numSlots;
	++ctx->frame;
	ctx->set = (uint32_t)(ctx->frame % ctx->numOutSets);
	ctx->st.frameTag = (uint32_t)ctx->frame;
	applyParity(ctx);
	// the slot's previous user (ctx->numSlots frames ago) must be done with it; volume producers on stream B must have finished
	MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evMainDone[ctx->slot]));
	MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evLight[ctx->slot]));
	if (ctx->producersDirty) { MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evProducers)); ctx->producersDirty = false; }
	// The constants go into this slot's pinned slice (the reference's upload-heap slices, MultiRayCaster.cpp:128-130, 461-467) and from
	// there to the device in one copy on stream A.  The slice's previous user is the frame numSlots back: its copy must have run --
	// in steady state it has, long ago (the slot's stage-A event covers it: no event of its own); a host that runs more than that many
	// frames ahead of the GPU waits here, as the reference's does on its fence.
	if (hipEventQuery(ctx->evStageA[ctx->slot]) != hipSuccess)
	{
		const auto t0 = std::chrono::steady_clock::now();
		MV_HIP(ctx, hipEventSynchronize(ctx->evStageA[ctx->slot]));
		ctx->hostWaitMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	}
	uint8_t* s = ctx->pinned + (size_t)ctx->slot * ctx->sliceBytes;
	memcpy(s, cbPerFrame, MV_CB_PER_FRAME_BYTES);
	memcpy(s + 256, perObject, (size_t)ctx->st.V * MV_PER_OBJECT_BYTES);
	MV_HIP(ctx, hipMemcpyAsync(ctx->fb[ctx->slot].dCb, s, ctx->sliceBytes, hipMemcpyHostToDevice, ctx->streamA));
	MV_HIP(ctx, stageADone(ctx, false));	// (a frame that never culls still marks its copy; inside the multi-GPU loop the cull's record follows at once)
	ctx->haveFrame = true;
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

static int marchLight(mv_ctx* ctx, uint32_t* wordsOut);
// EXPERIMENT (-DMV_DIAGNOSTICS builds, MV_DIST_LIGHT_ON_A=1; profiles/r03_light_on_stream_a.txt: slower): on a rank of the multi-GPU loop with whole-frame chains, the light share-out (march, all-gather,
// unpack) goes on stream A behind the frame's cull instead of in front of the frame's view march on its chain
static bool distLightOnStreamA(const mv_ctx* ctx);
int mv_march_light(mv_ctx* ctx) { return ctx ? marchLight(ctx, nullptr) : MV_ERR_INVALID_ARG; }

// wordsOut: inside the multi-GPU frame loop, where this rank's share of the relit map goes for the all-gather
static int marchLight(mv_ctx* ctx, uint32_t* wordsOut)
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
static int copyForward(mv_ctx* ctx, hipStream_t s)
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
static int viewCopyForward(mv_ctx* ctx)
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
		hipStream_t s = ctx->streamB[(ctx->frame + ctx->numViewStreams - 1u) % ctx->numViewStreams];
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
static int viewMarch(mv_ctx* ctx)
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
		const uint32_t beyond = mv::launchMarchTiles(streamB(ctx), ctx->st, tiles);
		// (an exact count is this frame's own: nothing can lie beyond the grid.  Otherwise the overflow launch follows on the same
		// stream: on the high-priority stream A it would have to wait for this frame's light march and held up the next frame's cull --
		// measured slower.  It is kept small, 64 workgroups, because its wavefronts must find slots on a GPU the other frame's march fills.)
		if (!exact && beyond != 0xffffffffu) mv::launchMarchTilesOverflow(streamB(ctx), ctx->st, beyond);
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
	if (ctx->lightTableFrame == ctx->frame)
	{
		// this frame's table came back with mv_readback_cull: no host synchronisation, the context's stream waits for the
		// frame's light march instead (work the caller orders on that stream then sees the relit map)
		*out = reinterpret_cast<uint4* const*>(ctx->pinnedCull + 64 + (size_t)ctx->st.V * 16)[volume];
		MV_HIP(ctx, evWait(ctx, ctx->stream, ctx->evStageA[ctx->slot]));
		MV_HIP(ctx, evWait(ctx, ctx->stream, ctx->evLight[ctx->slot]));
		return MV_OK;
	}
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemcpy(out, ctx->st.lightTable + volume, sizeof(uint4*), hipMemcpyDeviceToHost));
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
	const size_t d = ctx->st.N >> mip;
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
	const size_t d = ctx->st.N >> mip;
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
	return MV_OK;
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

// ---- mesh depth / shadow producer (ObjectRenderer.cpp:209-262) ----
int mv_mesh_upload(mv_ctx* ctx, const float* positions, uint32_t numVerts, const uint32_t* indices, uint32_t numTris)
{
	if (!ctx || !positions || !indices || !numVerts || !numTris) return MV_ERR_INVALID_ARG;
	for (size_t i = 0; i < (size_t)numTris * 3; ++i) MV_REQUIRE(ctx, indices[i] < numVerts, MV_ERR_RANGE, "mesh index out of range");
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	if (ctx->meshPositions) { (void)hipFree(ctx->meshPositions); ctx->meshPositions = nullptr; }
	if (ctx->meshIndices) { (void)hipFree(ctx->meshIndices); ctx->meshIndices = nullptr; }
	MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->meshPositions), (size_t)numVerts * 3 * sizeof(float)));
	MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->meshIndices), (size_t)numTris * 3 * sizeof(uint32_t)));
	if (!ctx->rasterBits)
	{
		const size_t n = std::max((size_t)ctx->st.W * ctx->st.H, (size_t)ctx->st.M * ctx->st.M);
		MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rasterBits), n * sizeof(uint32_t)));
		MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dMeshMatrix), 16 * sizeof(float)));
	}
	MV_HIP(ctx, hipMemcpy(ctx->meshPositions, positions, (size_t)numVerts * 3 * sizeof(float), hipMemcpyHostToDevice));
	MV_HIP(ctx, hipMemcpy(ctx->meshIndices, indices, (size_t)numTris * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
	ctx->meshVerts = numVerts; ctx->meshTris = numTris;
	return MV_OK;
}

static int rasterInto(mv_ctx* ctx, const float matT[16], float bx, float by, uint32_t W, uint32_t H)
{
	MV_REQUIRE(ctx, ctx->meshTris > 0, MV_ERR_NOT_READY, "no mesh (mv_mesh_upload)");
	MV_DRAIN(ctx);	// depth / shadow maps may still be read by frames in flight
	MV_HIP(ctx, hipMemcpyAsync(ctx->dMeshMatrix, matT, 16 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	mv::launchFillU32(ctx->stream, ctx->rasterBits, 0x3f800000u, (size_t)W * H);	// cleared to 1.0
	mv::launchRasterDepthBits(ctx->stream, ctx->meshPositions, ctx->meshIndices, ctx->meshTris, ctx->dMeshMatrix, bx, by, W, H, ctx->rasterBits);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
}

int mv_mesh_render_depth(mv_ctx* ctx, const float worldViewProjT[16], const float projBias[2])
{
	if (!ctx || !worldViewProjT) return MV_ERR_INVALID_ARG;
	const int rc = rasterInto(ctx, worldViewProjT, projBias ? projBias[0] : 0.0f, projBias ? projBias[1] : 0.0f, ctx->st.W, ctx->st.H);
	if (rc != MV_OK) return rc;
	// the depth bits ARE the D32F values
	MV_HIP(ctx, hipMemcpyAsync(ctx->ownDepth, ctx->rasterBits, (size_t)ctx->st.W * ctx->st.H * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	ctx->st.depthMap = ctx->ownDepth;
	ctx->haveDepth = true;
	return MV_OK;
}

int mv_mesh_render_shadow(mv_ctx* ctx, const float shadowWVPT[16])
{
	if (!ctx || !shadowWVPT) return MV_ERR_INVALID_ARG;
	const int rc = rasterInto(ctx, shadowWVPT, 0.0f, 0.0f, ctx->st.M, ctx->st.M);
	if (rc != MV_OK) return rc;
	mv::launchDepthBitsToD16(ctx->stream, ctx->rasterBits, ctx->ownShadow, (size_t)ctx->st.M * ctx->st.M);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	ctx->st.shadowMap = ctx->ownShadow;
	ctx->haveShadow = true;
	return MV_OK;
}

int mv_readback_depth_map(mv_ctx* ctx, float* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemcpy(dst, ctx->st.depthMap, (size_t)ctx->st.W * ctx->st.H * sizeof(float), hipMemcpyDeviceToHost));
	return MV_OK;
}

int mv_readback_shadow_map(mv_ctx* ctx, uint16_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemcpy(dst, ctx->st.shadowMap, (size_t)ctx->st.M * ctx->st.M * sizeof(uint16_t), hipMemcpyDeviceToHost));
	return MV_OK;
}

int mv_shard_rows(mv_ctx* ctx, uint32_t* row0, uint32_t* row1)
{
	if (!ctx || !row0 || !row1) return MV_ERR_INVALID_ARG;
	*row0 = ctx->st.bandRow0; *row1 = ctx->st.bandRow1;
	return MV_OK;
}

int mv_exchange_layout(mv_ctx* ctx, uint32_t* slot_dim, size_t* slot_bytes, size_t* shard_bytes)
{
	if (!ctx || !slot_dim || !slot_bytes || !shard_bytes) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	MV_HIP(ctx, stageBBegin(ctx));
	MV_HIP(ctx, hipMemcpyAsync(ctx->pinnedCounters, ctx->st.counters, kNumCounters * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	const uint32_t minMip = ctx->pinnedCounters[5] < MV_NUM_CUBE_MIP ? ctx->pinnedCounters[5] : 0u;
	const uint32_t d = ctx->st.N >> minMip;
	*slot_dim = d;
	*slot_bytes = (size_t)6 * d * d * (sizeof(uint2) + sizeof(float));
	const uint32_t sc = ctx->st.shardCount ? ctx->st.shardCount : 1u;
	*shard_bytes = *slot_bytes * ((ctx->st.V + sc - 1u) / sc);
	return MV_OK;
}

int mv_exchange_pack(mv_ctx* ctx, void* d_dst, uint32_t slot_dim)
{
	if (!ctx || !d_dst) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	MV_REQUIRE(ctx, slot_dim >= 1 && slot_dim <= ctx->st.N, MV_ERR_RANGE, "slot_dim out of range");
	mv::launchExchangePack(streamB(ctx), ctx->st, slot_dim, d_dst);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
}

int mv_exchange_unpack(mv_ctx* ctx, const void* d_src, uint32_t slot_dim)
{
	if (!ctx || !d_src) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->haveCull, MV_ERR_NOT_READY, "mv_cull has not been called");
	MV_REQUIRE(ctx, slot_dim >= 1 && slot_dim <= ctx->st.N, MV_ERR_RANGE, "slot_dim out of range");
	mv::launchExchangeUnpack(streamB(ctx), ctx->st, slot_dim, d_src);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
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


// ---------------------------------------------------------------------------------------------
// multi-GPU frame loop inside the library (include/multivolumes.h "mv_dist_*")
// ---------------------------------------------------------------------------------------------
static int distAllGather(mv_ctx* ctx, const void* send, void* recv, size_t bytes, hipStream_t stream, int which)
{
	DistState* d = ctx->dist;
	if (d->transport == 1)
	{
		const ncclResult_t r = d->allGather(send, recv, bytes, ncclUint8, d->comm[which], stream);
		if (r != ncclSuccess) return fail(ctx, MV_ERR_HIP, d->errorString ? d->errorString(r) : "ncclAllGather failed");
		return MV_OK;
	}
	if (d->cbAllGather(d->user, send, recv, bytes, reinterpret_cast<void*>(stream), which) != 0) return fail(ctx, MV_ERR_HIP, "the caller's all-gather failed");
	return MV_OK;
}

static int distAllReduceU32(mv_ctx* ctx, uint32_t* buf, size_t count, hipStream_t stream, int which)
{
	DistState* d = ctx->dist;
	if (d->transport == 1)
	{
		const ncclResult_t r = d->allReduce(buf, buf, count, ncclUint32, ncclSum, d->comm[which], stream);
		if (r != ncclSuccess) return fail(ctx, MV_ERR_HIP, d->errorString ? d->errorString(r) : "ncclAllReduce failed");
		return MV_OK;
	}
	if (d->cbAllReduce(d->user, buf, count, reinterpret_cast<void*>(stream), which) != 0) return fail(ctx, MV_ERR_HIP, "the caller's all-reduce failed");
	return MV_OK;
}

// buffers, streams and events shared by both transports
static int distAlloc(mv_ctx* ctx, uint8_t** out, size_t bytes, hipStream_t zeroOn = nullptr)
{
	void* p = nullptr;
	MV_HIP(ctx, hipMalloc(&p, bytes ? bytes : 16));
	ctx->dist->allocs.push_back(p);
	if (zeroOn) MV_HIP(ctx, hipMemsetAsync(p, 0, bytes ? bytes : 16, zeroOn));
	*out = static_cast<uint8_t*>(p);
	return MV_OK;
}

static int distCommonInitBody(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	const DeviceState& st = ctx->st;
	MV_REQUIRE(ctx, ctx->cfg.flags & MV_FLAG_PIPELINE_FRAMES, MV_ERR_INVALID_ARG,
		"mv_dist_* needs a context created with MV_FLAG_PIPELINE_FRAMES (whole frames on three view streams), optionally | MV_FLAG_SINGLE_VIEW_STREAM (two output sets, cull and light march ahead)");
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	if (!d->chains) MV_HIP(ctx, hipStreamCreateWithFlags(&d->streamX, hipStreamNonBlocking));
	evForget(ctx);
	const uint32_t world = st.shardCount;
	// One rank's shard of a frame's cube-face all-gather: the marched faces of the volumes the cull dealt to it.  The deal balances
	// estimated march cost, not texels -- a rank may be handed every cheap (low SmpCount) volume of a frame -- so the buffers hold the
	// worst case: every volume at mip 0 with six faces (C3: 75 MB per shard; 288 GB of HBM make that affordable).
	d->sendCap = (size_t)st.V * 6u * st.N * st.N * 12u;
	d->maxBandBytes = (size_t)mv::interleavedRowsPerRank(st.H, world) * 8u * st.W * sizeof(uint2);	// one rank's block of the rank-major composite image
	{
		// the light march of a frame is shared out by voxel blocks (mv_kernels.hip lightShare): 64 words per block, the same number of blocks per rank
		DeviceState split = st;
		split.flags |= mv::kFlagRotateAllLightMaps;
		d->lightShareBytes = ((size_t)mv::lightSharePerRank(split) + 1u) * 64u * sizeof(uint32_t);	// its blocks + a trailer (what its cull decided: checked by the unpack)
	}
	int rc;
	for (uint32_t c = 0; c < d->numChains; ++c)
	{
		if ((rc = distAlloc(ctx, &d->sendBuf[c], d->sendCap)) != MV_OK) return rc;
		if ((rc = distAlloc(ctx, &d->recvBuf[c], d->sendCap * world)) != MV_OK) return rc;
		uint8_t* w = nullptr;
		if ((rc = distAlloc(ctx, &w, std::max<size_t>(d->lightShareBytes * world, 4096), ctx->stream)) != MV_OK) return rc;
		d->lightWords[c] = reinterpret_cast<uint32_t*>(w);
		if ((rc = distAlloc(ctx, &d->lightSend[c], std::max<size_t>(d->lightShareBytes, 4096), ctx->stream)) != MV_OK) return rc;
		if ((rc = distAlloc(ctx, &d->rowSend[c], d->maxBandBytes, ctx->stream)) != MV_OK) return rc;
		if ((rc = distAlloc(ctx, &d->rowRecv[c], d->maxBandBytes * world)) != MV_OK) return rc;
	}
	if (hipHostMalloc(reinterpret_cast<void**>(&d->reports), kMaxFrameSlots * 64, hipHostMallocCoherent) != hipSuccess) return fail(ctx, MV_ERR_OUT_OF_MEMORY, "pinned exchange reports");
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i) { d->reports[i * 8u] = ~0ull; d->reports[i * 8u + 1u] = 0ull; }
	if (world > 1u) ctx->st.flags |= mv::kFlagInterleaveRows;	// screen rows dealt by tile row instead of in contiguous bands
#ifndef MV_DIST_STATIC_OWNERS	// (A/B builds only: v % shard_count inside the loop too)
	ctx->st.flags |= mv::kFlagBalanceOwners;	// the cube-map volumes of a frame are dealt to the ranks by estimated cost (replicated cull: the same deal everywhere)
#endif
	ctx->st.flags |= mv::kFlagRotateAllLightMaps;	// every rank rotates the relit volume's light buffer: the exchange fills it on the non-owners
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	return MV_OK;
}

// (a failure leaves the context without a frame loop, never with half of one)
static int distCommonInit(mv_ctx* ctx)
{
	const int rc = distCommonInitBody(ctx);
	if (rc != MV_OK)
	{
		(void)syncAll(ctx);
		distRelease(ctx);
		ctx->st.flags &= ~(mv::kFlagRotateAllLightMaps | mv::kFlagBalanceOwners | mv::kFlagInterleaveRows);
	}
	return rc;
}

static void* openRccl()
{
	void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
	return lib ? lib : dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
}

int mv_dist_rccl_available(void)
{
	void* lib = openRccl();
	if (!lib) return 0;
	for (const char* name : { "ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclAllReduce", "ncclBroadcast", "ncclCommDestroy", "ncclCommCount", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv" })
		if (!dlsym(lib, name)) return 0;
	return 1;
}

int mv_dist_unique_id(void* out)
{
	if (!out) return MV_ERR_INVALID_ARG;
	void* lib = openRccl();
	if (!lib) return MV_ERR_NOT_READY;
	auto getId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
	if (!getId) return MV_ERR_NOT_READY;
	static_assert(kDistChains * sizeof(ncclUniqueId) == MV_DIST_ID_BYTES, "one RCCL unique id per communicator");
	ncclUniqueId* ids = reinterpret_cast<ncclUniqueId*>(out);
	for (uint32_t i = 0; i < kDistChains; ++i) if (getId(&ids[i]) != ncclSuccess) return MV_ERR_HIP;
	return MV_OK;
}

int mv_dist_init_rccl(mv_ctx* ctx, const void* id)
{
	if (!ctx || !id) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, !ctx->dist, MV_ERR_INVALID_ARG, "mv_dist_init_* was already called on this context");
	MV_REQUIRE(ctx, ctx->st.shardCount >= 1, MV_ERR_INVALID_ARG, "shard_count");
	DistState* d = new (std::nothrow) DistState();
	if (!d) return MV_ERR_OUT_OF_MEMORY;
	d->transport = 1; d->inPlaceGather = true;
	d->chains = ctx->lightOnViewStream && ctx->numViewStreams >= 2u;
	d->numChains = d->chains ? ctx->numViewStreams : 1u;
	d->numComm = d->chains ? d->numChains : 2u;
	d->rcclLib = openRccl();
	if (d->rcclLib)
	{
		d->commInitRank = reinterpret_cast<decltype(d->commInitRank)>(dlsym(d->rcclLib, "ncclCommInitRank"));
		d->allGather = reinterpret_cast<decltype(d->allGather)>(dlsym(d->rcclLib, "ncclAllGather"));
		d->allReduce = reinterpret_cast<decltype(d->allReduce)>(dlsym(d->rcclLib, "ncclAllReduce"));
		d->commDestroy = reinterpret_cast<decltype(d->commDestroy)>(dlsym(d->rcclLib, "ncclCommDestroy"));
		d->errorString = reinterpret_cast<decltype(d->errorString)>(dlsym(d->rcclLib, "ncclGetErrorString"));
	}
	if (!d->rcclLib || !d->commInitRank || !d->allGather || !d->allReduce || !d->commDestroy) { delete d; return fail(ctx, MV_ERR_NOT_READY, "librccl.so could not be opened"); }
	{ const hipError_t e = hipSetDevice(ctx->cfg.device); if (e != hipSuccess) { delete d; return fail(ctx, MV_ERR_HIP, "hipSetDevice", e); } }
	const ncclUniqueId* ids = reinterpret_cast<const ncclUniqueId*>(id);
	for (uint32_t i = 0; i < d->numComm; ++i)
	{
		const ncclResult_t r = d->commInitRank(&d->comm[i], (int)ctx->st.shardCount, ids[i], (int)ctx->st.shardIndex);
		if (r != ncclSuccess)
		{
			const int rc = fail(ctx, MV_ERR_HIP, d->errorString ? d->errorString(r) : "ncclCommInitRank failed");
			for (uint32_t j = 0; j < i; ++j) if (d->comm[j]) (void)d->commDestroy(d->comm[j]);
			delete d;
			return rc;
		}
	}
	ctx->dist = d;
	return distCommonInit(ctx);
}

int mv_dist_init_callbacks(mv_ctx* ctx, mv_dist_allgather_fn allgather, mv_dist_allreduce_u32_fn allreduce, void* user)
{
	if (!ctx || !allgather || !allreduce) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, !ctx->dist, MV_ERR_INVALID_ARG, "mv_dist_init_* was already called on this context");
	DistState* d = new (std::nothrow) DistState();
	if (!d) return MV_ERR_OUT_OF_MEMORY;
	d->transport = 2; d->cbAllGather = allgather; d->cbAllReduce = allreduce; d->user = user;
	d->chains = ctx->lightOnViewStream && ctx->numViewStreams >= 2u;
	d->numChains = d->chains ? ctx->numViewStreams : 1u;
	d->numComm = d->chains ? d->numChains : 2u;
	ctx->dist = d;
	return distCommonInit(ctx);
}

int mv_dist_set_cull_crosscheck(mv_ctx* ctx, int enabled)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	ctx->dist->cullCrosscheck = enabled != 0;
	return MV_OK;
}

int mv_dist_set_inplace_gather(mv_ctx* ctx, int enabled)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	MV_REQUIRE(ctx, ctx->dist->transport == 2, MV_ERR_INVALID_ARG, "only for caller-supplied collectives (RCCL gathers in place already)");
	ctx->dist->inPlaceGather = enabled != 0;
	return MV_OK;
}

int mv_dist_shutdown(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	if (!ctx->dist) return MV_OK;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	if (ctx->dist->streamX) MV_HIP(ctx, hipStreamSynchronize(ctx->dist->streamX));
	distRelease(ctx);
	ctx->st.flags &= ~(mv::kFlagRotateAllLightMaps | mv::kFlagBalanceOwners | mv::kFlagInterleaveRows);
	return MV_OK;
}

// Exchange and composite of the pending frame k - 1, enqueued while frame k is being begun: by then the host has k - 1's cull results
// (they size the all-gather) without waiting.  Single view stream: both on stream X, beside frame k's view march.  Chains: on frame
// k - 1's own view stream, behind its view march, with that chain's communicator and buffers.
static hipStream_t distTailStream(mv_ctx* ctx, uint32_t chain) { return ctx->dist->chains ? ctx->streamB[chain] : ctx->dist->streamX; }

static int distTailExchange(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (!d->pending) return MV_OK;
	const DeviceState& st = d->stPending;
	const uint32_t slot = d->slotPending, world = st.shardCount;
	const uint32_t chain = d->chains ? d->chainPending : 0u;

	hipStream_t X = distTailStream(ctx, chain);
	if (!d->chains) MV_HIP(ctx, evWait(ctx, X, ctx->evMainDone[slot]));	// the frame's view march (chains: the same stream)
	if (world > 1)
	{
		// The frame's cull left {frame tag : texels of the largest shard of its deal} in pinned memory at least one mv_dist_frame ago: a
		// plain read in steady state (the wait only happens if the host is that far ahead of the GPU).  It gives the size of the frame's
		// all-gather, the same on every rank: the cull is replicated.
		volatile unsigned long long* report = d->reports + (size_t)slot * 8u;
		unsigned long long rep;
		{
			MV_SECTION(8);
			if ((uint32_t)(*report >> 32) != st.frameTag) MV_HIP(ctx, hipEventSynchronize(ctx->evStageA[slot]));
			rep = *report;
		}
		MV_REQUIRE(ctx, (uint32_t)(rep >> 32) == st.frameTag, MV_ERR_HIP, "the cull's exchange report is missing");
		// (the light exchange of this frame compared what every rank's cull decided: mv_kernels.hip lightUnpackKernel)
		// (mv_dist_set_cull_crosscheck(ctx, 0): a rank measured alone, with callbacks that exchange nothing -- the other ranks'
		// trailers are then not there to compare)
		MV_REQUIRE(ctx, !d->cullCrosscheck || report[1] != (((unsigned long long)st.frameTag << 32) | 1ull), MV_ERR_INVALID_ARG,
			"the ranks' culls of this frame disagree (different frame constants on different ranks?): its collectives would not match");
		const size_t shardBytes = std::max<size_t>(16, ((size_t)(uint32_t)rep * 12u + 15u) & ~(size_t)15u);
		MV_REQUIRE(ctx, shardBytes <= d->sendCap, MV_ERR_RANGE, "exchange shard larger than its buffer");
		{ MV_SECTION(9); mv::launchExchangeCompact(X, st, true, d->sendBuf[chain], shardBytes); }
		MV_HIP(ctx, hipGetLastError());
		int rc;
		{ MV_SECTION(10); rc = distAllGather(ctx, d->sendBuf[chain], d->recvBuf[chain], shardBytes, X, (int)chain); }
		if (rc != MV_OK) return rc;
		{ MV_SECTION(11); mv::launchExchangeCompact(X, st, false, d->recvBuf[chain], shardBytes); }
		MV_HIP(ctx, hipGetLastError());
		d->bytes += shardBytes * world;
	}
	return MV_OK;
}

static int distTailComposite(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (!d->pending) return MV_OK;
	const DeviceState& st = d->stPending;
	const uint32_t slot = d->slotPending, world = st.shardCount, rank = st.shardIndex;
	const uint32_t chain = d->chains ? d->chainPending : 0u;
	hipStream_t X = distTailStream(ctx, chain);
	if (d->pendingComposite)
	{
		// chains: the copy-forward of this frame's unmarched faces ran on the stream of the frame before it
		if (d->chains) MV_HIP(ctx, evWait(ctx, X, ctx->evCopy[slot]));
		mv::launchComposite(X, st, st.bandRow0, st.bandRow1);
		MV_HIP(ctx, hipGetLastError());
		if (world > 1)
		{
			// The image is stored rank-major (kFlagInterleaveRows): every rank's tile rows are one contiguous block of the same size, so
			// the frame buffer is the all-gather's output and this rank's block its input.
			const size_t rowBytes = (size_t)st.W * sizeof(uint2);
			const size_t block = d->maxBandBytes;
			uint8_t* frame = reinterpret_cast<uint8_t*>(st.composite);
			if (d->inPlaceGather)
			{
				const int rc = distAllGather(ctx, frame + (size_t)rank * block, frame, block, X, (int)chain);	// sendbuff == recvbuff + rank * count
				if (rc != MV_OK) return rc;
			}
			else
			{
				MV_HIP(ctx, hipMemcpyAsync(d->rowSend[chain], frame + (size_t)rank * block, block, hipMemcpyDeviceToDevice, X));
				const int rc = distAllGather(ctx, d->rowSend[chain], frame, block, X, (int)chain);
				if (rc != MV_OK) return rc;
			}
			d->bytes += (size_t)st.H * rowBytes;
		}
	}
	// everything that reads this frame's buffers has been enqueued: the next users of its slot / output set / light buffers wait for this
	MV_HIP(ctx, evRecord(ctx, ctx->evMainDone[slot], X));
	d->pending = false;
	return MV_OK;
}

namespace { int distTail(mv_ctx* ctx)
{
	const int rc = distTailExchange(ctx);
	return rc != MV_OK ? rc : distTailComposite(ctx);
} }

// The light march of frame k, shared out over the ranks by voxel blocks (every rank holds every volume): this rank marches its
// 1 / shard_count of the blocks, into its own copy of the relit volume's fresh light buffer and, as plain words, into its share of the
// all-gather; the other ranks' shares are then unpacked into the local copy.  Every rank keeps every light map current, because
// every rank marches its own screen rows of every volume on the screen-space scheme.
static bool distLightOnStreamA(const mv_ctx* ctx)
{
#ifdef MV_DIAGNOSTICS	// (the shipped library reads no environment variable)
	static const bool on = [] { const char* e = getenv("MV_DIST_LIGHT_ON_A"); return e && e[0] == '1'; }();
	return on && ctx->dist && ctx->dist->chains && ctx->st.shardCount > 1u;
#else
	(void)ctx;
	return false;
#endif
}

static int distLight(mv_ctx* ctx, uint32_t chain)
{
	DistState* d = ctx->dist;
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	uint8_t* lightAll = reinterpret_cast<uint8_t*>(d->lightWords[chain]);
	uint8_t* lightMine = d->inPlaceGather ? lightAll + (size_t)rank * d->lightShareBytes : d->lightSend[chain];
	int rc;
	if ((rc = marchLight(ctx, world > 1u ? reinterpret_cast<uint32_t*>(lightMine) : nullptr)) != MV_OK) return rc;
	if (world > 1u)
	{
		// The exchange goes in front of the view march that needs it, on that march's stream (chains: behind the light march, which is
		// on that stream too).  Single view stream: not behind the light march on stream L, where it would hold up the NEXT frame's
		// light march.
		hipStream_t E = d->chains ? streamB(ctx) : ctx->stream;
		if (distLightOnStreamA(ctx)) E = ctx->stream;
		if (!d->chains) MV_HIP(ctx, evWait(ctx, E, ctx->evLight[ctx->slot]));	// this rank's light march (chains: the same stream)
		MV_HIP(ctx, evWait(ctx, E, ctx->evStageA[ctx->slot]));	// this frame's cull: the relit volume, its rotated buffer
		MV_HIP(ctx, evWait(ctx, E, ctx->evMainDone[(ctx->slot + ctx->numSlots - ctx->st.lightBufs) % ctx->numSlots]));	// frame k - 3, the last reader of the buffer the unpack overwrites
		if ((rc = distAllGather(ctx, lightMine, lightAll, d->lightShareBytes, E, d->chains ? (int)chain : 1)) != MV_OK) return rc;
		mv::launchLightUnpack(E, ctx->st, d->lightWords[chain]);
		MV_HIP(ctx, hipGetLastError());
		MV_HIP(ctx, evRecord(ctx, ctx->evLight[ctx->slot], E));	// "this frame's light map is current on this rank": the next user of the slot waits for it
		d->bytes += d->lightShareBytes * world;
	}
	return MV_OK;
}

int mv_dist_frame(mv_ctx* ctx, const void* cbPerFrame, const void* perObject, int with_composite)
{
	if (!ctx || !cbPerFrame || !perObject) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	DistState* d = ctx->dist;
	int rc;
	struct Lean { mv_ctx* c; ~Lean() { c->leanEvents = false; } } lean = { ctx };
	ctx->leanEvents = d->chains;
	// ---- frame k: constants + cull (stream A) ----
	{ MV_SECTION(0); if ((rc = mv_update_frame(ctx, cbPerFrame, perObject)) != MV_OK) return rc; }
	{ MV_SECTION(1); if ((rc = mv_cull(ctx)) != MV_OK) return rc; }
#ifdef MV_HOST_PROFILE
	++gSectionCalls;
	if (getenv("MV_PROFILE_PHASE"))	// at the start of a call (after this frame's cull was enqueued): which earlier frames' stages has the GPU finished?
	{
		static uint64_t cullDone[6], tailDone[6], n;
		for (uint32_t b = 1; b <= 5; ++b)
		{
			const uint32_t sl = (ctx->slot + ctx->numSlots * 4u - b) % ctx->numSlots;
			if (b < ctx->numSlots && hipEventQuery(ctx->evStageA[sl]) == hipSuccess) ++cullDone[b];
			if (b < ctx->numSlots && hipEventQuery(ctx->evMainDone[sl]) == hipSuccess) ++tailDone[b];
		}
		if (++n % 200 == 0)
		{
			fprintf(stderr, " [phase at call start, share of calls: cull done of frame k-1 %.2f k-2 %.2f k-3 %.2f; tail done of k-1 %.2f k-2 %.2f k-3 %.2f]",
				cullDone[1] / 200.0, cullDone[2] / 200.0, cullDone[3] / 200.0, tailDone[1] / 200.0, tailDone[2] / 200.0, tailDone[3] / 200.0);
			for (auto& x : cullDone) x = 0; for (auto& x : tailDone) x = 0;
		}
	}
	if (getenv("MV_PROFILE_CULL_LATENCY"))	// how long after its enqueue the cull's report reaches pinned memory (spins: perturbs the loop)
	{
		const auto t0 = std::chrono::steady_clock::now();
		while ((uint32_t)(*reinterpret_cast<const volatile unsigned long long*>(ctx->pinnedTiles) >> 32) != (uint32_t)ctx->frame) {}
		const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
		static double sum = 0, worst = 0; static int n = 0;
		sum += us; worst = us > worst ? us : worst;
		if (++n % 200 == 0) { fprintf(stderr, " [cull latency after enqueue: avg %.1f us, max %.1f us]", sum / n, worst); sum = 0; worst = 0; n = 0; }
	}
#endif
	const uint32_t chain = d->chains ? (uint32_t)(ctx->frame % d->numChains) : 0u;
	if (d->chains)
	{
		// Whole frames on three view streams.  The ORDER OF THE CALLS is chosen for the two things the host reads from pinned memory
		// without waiting: the size of frame k - 1's cube-face all-gather (its cull's report: the later the host asks, the surer it is
		// there) and this frame's tile count (an exact grid for the view march, no overflow launch behind it).  So: this frame's light
		// march and its exchange first (stream k % 3), then frame k - 1's tail (stream (k - 1) % 3), this frame's view march last.  The
		// streams are independent, so the order costs the GPU nothing.
		{ MV_SECTION(5); if ((rc = distLight(ctx, chain)) != MV_OK) return rc; }
		// ---- frame k - 1's cube-face exchange, on ITS stream behind its view march; then k - 1's composite and row gather; then the
		// copy-forward of the faces frame k does not march, which reads k - 1's set on that same stream, i.e. behind the unpack ----
		{ MV_SECTION(2); if ((rc = distTailExchange(ctx)) != MV_OK) return rc; }
		{ MV_SECTION(4); if ((rc = distTailComposite(ctx)) != MV_OK) return rc; }
		// (the copy-forward behind the composite: the composite has already made this stream wait for frame k - 1's copy, which this
		// one builds on, and nothing needs frame k's copy before frame k's own composite, a call from now)
		{ MV_SECTION(3); if ((rc = viewCopyForward(ctx)) != MV_OK) return rc; }
		{ MV_SECTION(6); if ((rc = viewMarch(ctx)) != MV_OK) return rc; }
	}
	else
	{
		if ((rc = distLight(ctx, chain)) != MV_OK) return rc;
		if ((rc = mv_march_view(ctx)) != MV_OK) return rc;
		// ---- frame k - 1: exchange + composite, on stream X beside this frame's march ----
		if ((rc = distTail(ctx)) != MV_OK) return rc;
		// ---- frame k's unmarched cube faces <- frame k - 1's set, now complete on this rank (stream X is in order); frame k + 1's march waits for it ----
		MV_HIP(ctx, evWait(ctx, d->streamX, ctx->evStageA[ctx->slot]));
		if ((rc = copyForward(ctx, d->streamX)) != MV_OK) return rc;
	}
	{ MV_SECTION(7); d->stPending = ctx->st; d->slotPending = ctx->slot; d->chainPending = chain; d->pendingComposite = with_composite != 0; d->pending = true; }
	return MV_OK;
}

int mv_dist_flush(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	return distTail(ctx);
}

// One all-gather (separate buffers), one in-place all-gather and one all-reduce on each communicator, checked on the host: run once
// after mv_dist_init_* so that a broken transport fails before any frame does.
int mv_dist_selftest(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	DistState* d = ctx->dist;
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	const size_t n = 4096;	// bytes per rank
	MV_REQUIRE(ctx, d->sendCap >= n && d->maxBandBytes >= n, MV_ERR_RANGE, "self-test buffers");
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	std::vector<uint8_t> h(n * world), g(n * world);
	for (int which = 0; which < (int)d->numComm; ++which)
	{
		// the stream the frame loop uses this communicator on, and the buffers it uses there
		hipStream_t s = d->chains ? ctx->streamB[which] : which == 0 ? d->streamX : ctx->stream;
		const uint32_t c = d->chains ? (uint32_t)which : 0u;
		// all-gather, separate buffers
		for (size_t i = 0; i < n; ++i) h[i] = (uint8_t)(rank * 31u + i * 7u + which);
		MV_HIP(ctx, hipMemcpyAsync(d->sendBuf[c], h.data(), n, hipMemcpyHostToDevice, s));
		int rc = distAllGather(ctx, d->sendBuf[c], d->recvBuf[c], n, s, which);
		if (rc != MV_OK) return rc;
		MV_HIP(ctx, hipMemcpyAsync(g.data(), d->recvBuf[c], n * world, hipMemcpyDeviceToHost, s));
		MV_HIP(ctx, hipStreamSynchronize(s));
		for (uint32_t r = 0; r < world; ++r)
			for (size_t i = 0; i < n; ++i)
				MV_REQUIRE(ctx, g[r * n + i] == (uint8_t)(r * 31u + i * 7u + which), MV_ERR_HIP, "all-gather self-test: wrong bytes");
		// in place (the row-band gather of equal bands): sendbuff == recvbuff + rank * count
		if (d->inPlaceGather)
		{
			MV_HIP(ctx, hipMemsetAsync(d->rowRecv[c], 0, n * world, s));
			MV_HIP(ctx, hipMemcpyAsync(d->rowRecv[c] + (size_t)rank * n, h.data(), n, hipMemcpyHostToDevice, s));
			rc = distAllGather(ctx, d->rowRecv[c] + (size_t)rank * n, d->rowRecv[c], n, s, which);
			if (rc != MV_OK) return rc;
			MV_HIP(ctx, hipMemcpyAsync(g.data(), d->rowRecv[c], n * world, hipMemcpyDeviceToHost, s));
			MV_HIP(ctx, hipStreamSynchronize(s));
			for (uint32_t r = 0; r < world; ++r)
				for (size_t i = 0; i < n; ++i)
					MV_REQUIRE(ctx, g[r * n + i] == (uint8_t)(r * 31u + i * 7u + which), MV_ERR_HIP, "in-place all-gather self-test: wrong bytes");
		}
		// all-reduce (integer sum; only one rank contributes non-zero words, like the light-map exchange)
		std::vector<uint32_t> w(256, 0u);
		if (rank == world - 1u) for (uint32_t i = 0; i < 256; ++i) w[i] = 0x9e3779b9u * (i + 1u) + which;
		MV_HIP(ctx, hipMemcpyAsync(d->lightWords[c], w.data(), 1024, hipMemcpyHostToDevice, s));
		rc = distAllReduceU32(ctx, d->lightWords[c], 256, s, which);
		if (rc != MV_OK) return rc;
		MV_HIP(ctx, hipMemcpyAsync(w.data(), d->lightWords[c], 1024, hipMemcpyDeviceToHost, s));
		MV_HIP(ctx, hipStreamSynchronize(s));
		for (uint32_t i = 0; i < 256; ++i) MV_REQUIRE(ctx, w[i] == 0x9e3779b9u * (i + 1u) + which, MV_ERR_HIP, "all-reduce self-test: wrong words");
	}
	return MV_OK;
}

int mv_dist_info(mv_ctx* ctx, uint32_t* transport, uint32_t* ranks, uint32_t* communicators)
{
	if (!ctx || !transport || !ranks || !communicators) return MV_ERR_INVALID_ARG;
	*transport = 0; *ranks = 0; *communicators = 0;
	const DistState* d = ctx->dist;
	if (!d) return MV_OK;
	*transport = (uint32_t)d->transport; *communicators = d->numComm;
	*ranks = ctx->st.shardCount;
	if (d->transport == 1)
	{
		// what RCCL itself says about the communicator (ncclCommCount), not what the context was told
		auto commCount = reinterpret_cast<decltype(&ncclCommCount)>(dlsym(d->rcclLib, "ncclCommCount"));
		int n = 0;
		MV_REQUIRE(ctx, commCount && commCount(d->comm[0], &n) == ncclSuccess, MV_ERR_HIP, "ncclCommCount failed");
		*ranks = (uint32_t)n;
	}
	return MV_OK;
}

int mv_dist_exchanged_bytes(mv_ctx* ctx, uint64_t* bytes, int reset)
{
	if (!ctx || !bytes) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	*bytes = ctx->dist->bytes;
	if (reset) ctx->dist->bytes = 0;
	return MV_OK;
}

} // extern "C"
