// mv_ctx.hpp -- the context behind the C-ABI (include/multivolumes.h) and the host-side helpers its translation units share:
// mv_api.cpp (context, stages, readbacks), mv_dist.cpp (multi-GPU frame loop and its transports), mv_mesh_api.cpp (mesh depth / shadow producer).
#pragma once
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
#include <atomic>
#include <chrono>
#include <new>
#include <dlfcn.h>
#include <rccl/rccl.h>	// types and prototypes only: the library is opened with dlopen when mv_dist_init_rccl is called

using mv::DeviceState;

#ifdef MV_HOST_PROFILE	// (diagnostic builds: where the host's time inside mv_dist_frame goes; printed by mv_dist_shutdown / mv_destroy)
inline double gSection[12]; inline uint64_t gSectionCalls;
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
// Frames per launch (mv_config.frames_per_launch = B, 1..4; the multi-GPU loop with peer stores): B consecutive frames share a view stream and
// go through each of its long kernels together (mv_kernels.hpp DeviceStateBatch), so C x B frames are in flight: C x B output sets, C x B
// light buffers, C x B + 1 frame slots.
constexpr uint32_t kMaxChains = 5;
constexpr uint32_t kMaxFramesInFlight = 16;	// C x B
constexpr uint32_t kMaxFrameSlots = kMaxFramesInFlight + 1;
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
	uint32_t* batchInfo = nullptr;		// 8 words: what the ranks of the multi-GPU loop compare of their culls, the volumes the frames of a launch relight (cullKernel's epilogue)
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
constexpr uint32_t kMaxOutSets = kMaxFramesInFlight;
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
	int transport = 0;				// 1 = RCCL, 2 = caller-supplied collectives, 3 = peer stores (no collectives)
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
	uint64_t framePending = 0;		// (peer stores) its number in the loop's own count
	uint32_t numLightWords = 1;		// copies of lightWords / lightSend in use (see distCommonInitBody)
	uint64_t bytes = 0;
	std::vector<void*> allocs;		// device buffers of the loop (freed by distRelease: a shutdown followed by another init allocates afresh)
	// ---- peer-store transport (transport == 3; mv_dist_peer_export / mv_dist_init_peer, mv_peer.hip) ----
	bool peerExported = false, peerReady = false, peerLoopback = false;
	unsigned long long* peerFlags = nullptr;	// this rank's flag block (uncached device memory, written by the other ranks)
	int peerFlagsKind = 0;						// how it was allocated: 2 uncached, 1 fine-grained, 0 plain hipMalloc
	std::vector<void*> peerMapped;				// IPC mappings of the other ranks' buffers (closed by distRelease)
	unsigned long long** dPeerFlags = nullptr;	// device tables, [world] each: every rank's flag block ...
	uint4** dPeerLight = nullptr;				// ... light buffers ...
	uint32_t* lightPlain = nullptr;				// (plain light push) this rank's staging maps: numSlots x L^3 plain R11G11B10F texels, the slot of frame k at (k's frame slot) x L^3
	uint32_t** dPeerLightPlain = nullptr;		// ... every rank's
	uint32_t peerRebuildBlocks = 32;			// (plain light push) workgroups of a view-march launch that rebuild the other ranks' cells (mv_dist_init_peer: 256 for ranks that own their GPU, 32 for ranks that share one)
	bool peerPlainLight = true;					// a frame's relit map travels as plain texels and the receivers rebuild the cells (false: MV_DIST_PEER_PUSH_CELLS, round 4's whole cells)
	uint2** dPeerCube[kMaxOutSets] = {};		// ... and, per output set, cube maps, cube depths, frame buffer
	float** dPeerDepth[kMaxOutSets] = {};
	uint2** dPeerComposite[kMaxOutSets] = {};
	uint64_t peerFrame = 0;						// frames begun since mv_dist_init_peer (the same on every rank: the flags count in it)
	uint64_t peerTimeoutTicks = 30ull * 100000000ull;	// 30 s of wall_clock64 (100 MHz)
	uint32_t rowsStream[kDistChains] = {};		// per chain: the view stream its frames run on
	uint64_t rowsWaitFrame[kDistChains] = {};	// per chain: the newest frame whose rows this rank has composited but not yet waited for from the others (0 = none): enqueued lazily, by whatever synchronises
	bool peerDealVolumes = false;				// whole cube-map volumes per rank (dealt by cost: the default with one frame per launch) instead of every volume's tiles shared out
	bool peerRowsRoot = false;					// MV_DIST_PEER_ROWS_TO_ROOT: the ranks push their rows to rank 0 only
	bool peerFold = false;						// the waits are the consumers' own prologues (mv_kernels.hpp peerFoldWait) instead of one-wavefront kernels in front of them
	bool peerAcquire = true;					// a folded wait ends with a system-scope acquire by the wavefront that polled (mv_kernels.hpp peerFoldAcquire); false: MV_DIST_PEER_NO_ACQUIRE, round 4's form
	uint32_t peerStaleWords = 0, peerFirstReadBad = 0;	// what the last in-kernel coherence test saw (mv_dist_peer_coherence)
	uint32_t* dPeerCounters = nullptr;			// finished-workgroup counters of the push kernels: [kind][chain]
	uint32_t* dPeerMismatch = nullptr;			// self-test counter
	// several frames per launch (mv_dist_batch: ctx->framesPerLaunch > 1): the frames of the group being collected (their upload, cull and orders
	// are enqueued; light march and view march go out together when the group is full or flushed) and of the group before it (whose composite
	// and row push are still to be enqueued)
	struct BatchFrame { DeviceState st; uint32_t slot = 0, set = 0, prevChain = 0; uint64_t frame = 0; bool composite = false, culled = false; };
	BatchFrame cur[4], pend[4];
	uint32_t nCur = 0, nPend = 0;
	uint64_t pendBatch = 0;					// the pending group's number (peerFrame counts groups in this mode)
	uint64_t peerSelfTests = 0;					// rounds of the self-test so far (its flag row counts in them; collective, so equal on every rank)
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
	uint32_t framesPerLaunch = 1;		// B (mv_config.frames_per_launch): frames (k - 1) / B share view stream ((k - 1) / B) % C; the multi-GPU loop launches them together
	uint64_t batchIndex = 0;			// B > 1: which group of B frames the current frame belongs to (what picks its view stream)
	bool batchExplicit = false;			// the multi-GPU loop counts the groups itself (a flush may close one early); otherwise mv_update_frame derives batchIndex from the frame number
	uint32_t batchPos = 0;				// the current frame's position in its group
	uint32_t curChain = 0, prevChain = 0;	// the view stream of the current frame and of the frame before it (mv_update_frame)
	uint32_t set = 0;
	bool ownStream = false, ownStreamA = false, ownStreamL = false, ownStreamB1 = false;
	bool cullOnChain = false;			// MV_FLAG_CULL_ON_VIEW_STREAM: upload and cull on the frame's own view stream instead of stream A
	bool lightOnViewStream = false;		// pipelined with several view streams: a frame's light march runs on that frame's view stream
	DeviceState st;						// always holds the CURRENT frame slot's pointers (applyParity)
	FrameBufs fb[kMaxFrameSlots];		// constants + cull outputs per slot (numSlots of them in use)
	uint32_t numSlots = 4;				// light buffers per volume + 1
	uint32_t slot = 0;					// frame counter mod numSlots
	hipEvent_t evStageA[kMaxFrameSlots] = {}, evLight[kMaxFrameSlots] = {}, evMainDone[kMaxFrameSlots] = {}, evProducers = nullptr;
	hipEvent_t evCopy[kMaxFrameSlots] = {};	// a frame's unmarched cube faces are up to date (copyForward)
	hipEvent_t evLanded[kMaxFrameSlots] = {};	// (multi-GPU loop, several frames per launch) every rank's marched faces of the frame's group are in this rank's output sets
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
	uint8_t* dCbAll = nullptr;			// the slots' device copies of the constants, sliceBytes apart (fb[slot].dCb points into it)
	bool deferUpload = false;			// mv_update_frame fills the pinned slice only: the caller copies it (several frames per launch: one copy per group)
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
	bool frameBegun = false;			// some mv_update_frame has run on this context (unlike haveFrame, a viewport change does not clear it: the slot's light table still needs settling)
	// Frame latency, sampled (mv_set_latency_sampling): the host's clock when a frame's constants are handed over (mv_update_frame /
	// mv_dist_frame) and when the last kernel of that frame has finished -- a host function behind the record of the frame's "done" event.
	// A frame is marked more than once (behind its march, behind its composite or tail): a sample counts when every mark the host enqueued for it
	// has fired AND no further mark can come (its slot was handed to a newer frame, or everything was synchronised) -- round 5 counted a sample at its
	// first mark and let the second one write into whichever frame had taken the sample over.
	struct LatencySample { std::atomic<long long> startNs{0}, doneNs{0}; std::atomic<uint32_t> fired{0}; uint32_t enqueued = 0; bool busy = false, open = false; };
	static constexpr uint32_t kLatencyPool = 64;	// samples in flight (a frame slot is reused by the HOST before the GPU is done with the frame it held: samples live apart from slots)
	LatencySample lat[kLatencyPool];
	int latOfSlot[kMaxFrameSlots];					// the sample of the frame a slot holds, or -1
	uint32_t latEvery = 0;
	double latSumMs = 0.0, latMaxMs = 0.0;
	uint64_t latCount = 0;
	bool leanEvents = false;			// inside mv_dist_frame on a chains context: event records that a later record of the same call supersedes are left out (the host enqueues a rank's frame in ~100 us: every call counts)
	bool lightMarched = false, lightTableKept = false;	// this frame: mv_march_light ran / the frame keeps the previous frame's light table (see keepLightTable)
	bool profiling = false;
	std::vector<EventPair> pending;
	std::vector<EventPair> freeEvents;
	double msStage[4] = { 0, 0, 0, 0 };
	uint64_t nStage[4] = { 0, 0, 0, 0 };
	std::string lastError;
	std::string warning;				// mv_get_warning: what a caller should know although the call succeeded (lastError stays the text of the last FAILURE)
	DistState* dist = nullptr;
	// Cross-stream ordering goes through evRecord / evWait below, which leave out the waits that are redundant: an event recorded on the
	// very stream that is asked to wait for it (stream order already says so), and a wait the stream has already made for the same
	// record.  With whole frames on three view streams frame k - 3 sits on frame k's stream, so half of a frame's waits are of the
	// first kind (a stream wait is ~1 us of host time on a pending event; a rank of the multi-GPU loop issued sixteen per frame).
	struct EvTrack { hipEvent_t e = nullptr; hipStream_t on = nullptr; uint64_t seq = 0; uint64_t waited[8] = {}; };
	EvTrack evTrack[5 * kMaxFrameSlots + 4];
	hipStream_t evStreams[8] = {};
	uint64_t evSeq = 0;
	double hostWaitMs = 0.0;			// time mv_update_frame spent waiting for the GPU to free a frame slot (mv_host_wait_ms)
};

// entry points the translation units of the library share (not part of the ABI)
#define MV_INTERNAL __attribute__((visibility("hidden")))
extern "C" { namespace mvi
{
MV_INTERNAL int marchLight(mv_ctx* ctx, uint32_t* wordsOut);	// mv_api.cpp
MV_INTERNAL int keepLightTable(mv_ctx* ctx);
MV_INTERNAL int viewCopyForward(mv_ctx* ctx);
MV_INTERNAL int viewMarch(mv_ctx* ctx);
MV_INTERNAL int copyForward(mv_ctx* ctx, hipStream_t s);
MV_INTERNAL int distTail(mv_ctx* ctx);							// mv_dist.cpp
MV_INTERNAL void distRelease(mv_ctx* ctx);
MV_INTERNAL bool distLightOnStreamA(const mv_ctx* ctx);
MV_INTERNAL int distCheck(mv_ctx* ctx);						// (peer stores) MV_ERR_TIMEOUT if a wait for another rank has given up
MV_INTERNAL int distBeforeSync(mv_ctx* ctx);					// (peer stores) enqueue the waits for the other ranks' rows of the frames composited so far
} }	// (C linkage only because their definitions sit inside the extern "C" blocks of the ABI)

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

// a buffer OTHER ranks may store into (peer-store transport): fine-grained memory under MV_FLAG_FINEGRAINED_PEER_BUFFERS, plain otherwise
template <typename T>
int devAllocShared(mv_ctx* ctx, T** out, size_t bytes)
{
	if (!(ctx->cfg.flags & MV_FLAG_FINEGRAINED_PEER_BUFFERS)) return devAlloc(ctx, out, bytes);
	void* p = nullptr;
	if (hipExtMallocWithFlags(&p, bytes ? bytes : 16, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, MV_ERR_OUT_OF_MEMORY, "hipExtMallocWithFlags(hipDeviceMallocFinegrained): MV_FLAG_FINEGRAINED_PEER_BUFFERS on a runtime that refuses it"); }
	ctx->allocs.push_back(p);
	MV_HIP(ctx, hipMemsetAsync(p, 0, bytes ? bytes : 16, ctx->stream));
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
	st.tilesReportSlot = (ctx->pinnedTiles && ctx->framesPerLaunch > 1u) ? ctx->pinnedTiles + 1u + ctx->slot : nullptr;
	st.volumeInfo = f.volumeInfo; st.visibleVolumes = f.visibleVolumes; st.cubeMapVolumes = f.cubeMapVolumes;
	st.batchInfo = (ctx->dist && ctx->dist->peerReady) ? f.batchInfo : nullptr; st.batchInfoPrev = q.batchInfo; st.batchPos = ctx->batchPos; st.batchPad = 0u;
	st.counters = f.counters; st.countersPrev = q.counters; st.ownedVisible = f.ownedVisible; st.workEntries = f.workEntries; st.workTileStart = f.workTileStart;
	st.cubeOffset = f.cubeOffset; st.volumeFlags = f.volumeFlags; st.volumeRects = f.volumeRects; st.screenSlot = f.screenSlot; st.exchangeOff = f.exchangeOff; st.lightDirs = f.lightDirs; st.lightOrder = f.lightOrder;
	st.lightTable = f.lightTable; st.lightTablePrev = q.lightTable; st.volumeInfoPrev = q.volumeInfo; st.volConsts = f.volConsts; st.queueHeads = f.queueHeads;
	const OutBufs& o = ctx->ob[ctx->set];
	st.cubeMaps = o.cubeMaps; st.cubeDepths = o.cubeDepths; st.composite = o.composite; st.screenLayers = o.screenLayers; st.faceStamps = o.faceStamps;
	st.waveStats = o.waveStats; st.compositeStats = o.compositeStats; st.compositeRetry = o.compositeRetry; st.tileCost = f.tileCost;
	st.rayRecords = o.rayRecords; st.tileInfo = o.tileInfo;
	if (ctx->dist && ctx->dist->peerReady && st.shardCount > 1u)	// (a lone rank has nobody to store to: the plain kernels)
	{
		const DistState* d = ctx->dist;
		st.peerCount = st.shardCount; st.peerLight = d->dPeerLight;
		st.peerCubeMaps = d->dPeerCube[ctx->set]; st.peerCubeDepths = d->dPeerDepth[ctx->set]; st.peerComposite = d->dPeerComposite[ctx->set];
		st.lightPlainOff = (uint64_t)ctx->slot * st.L * st.L * st.L;
		st.lightPlain = (d->peerPlainLight && d->lightPlain) ? d->lightPlain + st.lightPlainOff : nullptr; st.peerLightPlain = d->dPeerLightPlain;
	}
	else { st.peerCount = 0u; st.peerLight = nullptr; st.peerCubeMaps = nullptr; st.peerCubeDepths = nullptr; st.peerComposite = nullptr; st.lightPlain = nullptr; st.peerLightPlain = nullptr; st.lightPlainOff = 0; }
	st.tileOrder = ctx->orderTiles ? f.tileOrder : nullptr;
	st.binCursor = ctx->orderTiles ? f.binCounts : nullptr;
	st.orderCap = st.maxTiles + st.maxScreenTiles + 1u;
}

int syncAll(mv_ctx* ctx)
{
	if (ctx->dist && ctx->dist->peerReady) (void)mvi::distBeforeSync(ctx);	// (a launch failure surfaces in the synchronisation below)
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
// the view stream of the current frame, and of the frame before it
uint32_t chainOfFrame(const mv_ctx* ctx) { return (uint32_t)((ctx->framesPerLaunch > 1u ? ctx->batchIndex : ctx->frame) % ctx->numViewStreams); }
hipStream_t streamB(mv_ctx* ctx) { return ctx->streamB[chainOfFrame(ctx)]; }
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
// (latency sampling) behind a record of slot's "done" event on stream s: the host's clock at that point of the stream
inline long long hostNowNs() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline void latencyDoneFn(void* p)
{
	mv_ctx::LatencySample* l = static_cast<mv_ctx::LatencySample*>(p);
	const long long now = hostNowNs();
	long long seen = l->doneNs.load(std::memory_order_relaxed);
	while (seen < now && !l->doneNs.compare_exchange_weak(seen, now, std::memory_order_relaxed)) {}	// (marks of one frame may sit on two streams: the later clock stands)
	l->fired.fetch_add(1u, std::memory_order_release);
}
// all: everything enqueued has run (the caller synchronised), so the samples of frames still in their slots are complete too
inline void latencyHarvest(mv_ctx* ctx, bool all = false)
{
	for (uint32_t i = 0; i < mv_ctx::kLatencyPool; ++i)
	{
		mv_ctx::LatencySample& l = ctx->lat[i];
		if (!l.busy || (l.open && !all)) continue;
		if (l.fired.load(std::memory_order_acquire) != l.enqueued) continue;	// a mark is still on its stream
		if (l.enqueued)
		{
			const double ms = (double)(l.doneNs.load(std::memory_order_relaxed) - l.startNs.load(std::memory_order_relaxed)) * 1e-6;
			ctx->latSumMs += ms; ctx->latMaxMs = ms > ctx->latMaxMs ? ms : ctx->latMaxMs; ++ctx->latCount;
		}
		l.busy = false; l.open = false;
		for (uint32_t sl = 0; sl < kMaxFrameSlots; ++sl) if (ctx->latOfSlot[sl] == (int)i) ctx->latOfSlot[sl] = -1;	// (no later mark of that frame finds a sample that may be another frame's by then)
	}
}
// a new frame in `slot`: every latEvery-th one is sampled, if a sample is free
inline void latencyBegin(mv_ctx* ctx, uint32_t slot, uint64_t frame)
{
	if (!ctx->latEvery) { ctx->latOfSlot[slot] = -1; return; }	// (the table is only defined once mv_set_latency_sampling has run)
	if (ctx->latOfSlot[slot] >= 0) ctx->lat[ctx->latOfSlot[slot]].open = false;	// the frame this slot held has been enqueued to its end: no more marks for it
	ctx->latOfSlot[slot] = -1;
	if (frame % ctx->latEvery != 0u) return;
	latencyHarvest(ctx);
	for (uint32_t i = 0; i < mv_ctx::kLatencyPool; ++i)
		if (!ctx->lat[i].busy)
		{
			mv_ctx::LatencySample& l = ctx->lat[i];
			l.busy = true; l.open = true; l.enqueued = 0u; l.fired.store(0u, std::memory_order_relaxed);
			l.doneNs.store(0, std::memory_order_relaxed); l.startNs.store(hostNowNs(), std::memory_order_relaxed);
			ctx->latOfSlot[slot] = (int)i;
			return;
		}
}
inline hipError_t latencyMark(mv_ctx* ctx, uint32_t slot, hipStream_t s)
{
	if (!ctx->latEvery || ctx->latOfSlot[slot] < 0) return hipSuccess;
	mv_ctx::LatencySample& l = ctx->lat[ctx->latOfSlot[slot]];
	++l.enqueued;
	const hipError_t e = hipLaunchHostFunc(s, latencyDoneFn, &l);
	if (e != hipSuccess) --l.enqueued;
	return e;
}
hipError_t stageBDone(mv_ctx* ctx)
{
	const hipError_t e = evRecord(ctx, ctx->evMainDone[ctx->slot], streamB(ctx));
	return e != hipSuccess ? e : latencyMark(ctx, ctx->slot, streamB(ctx));
}

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
int drainForWrite(mv_ctx* ctx)
{
	if (ctx->dist && (ctx->dist->pending || ctx->dist->nCur || ctx->dist->nPend))
	{
		const int rc = mvi::distTail(ctx);
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
			// (several frames per launch: bits 8.. of the stage hold the frames the launch carried -- the averages stay per frame)
			ctx->msStage[ev.stage & 0xff] += ms;
			ctx->nStage[ev.stage & 0xff] += (ev.stage >> 8) > 0 ? (uint64_t)(ev.stage >> 8) : 1u;
		}
		ctx->freeEvents.push_back(ev);
	}
	ctx->pending.clear();
}

} // namespace
