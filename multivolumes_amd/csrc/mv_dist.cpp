// mv_dist.cpp -- multi-GPU: the exchange building blocks (mv_exchange_*) and the frame loop inside the library (mv_dist_*)
// with its transports.  The reference is single-device (MultiVolumes.cpp:159-187): there is no call to match, the data flow is SURVEY 8(e)'s.
#include "mv_ctx.hpp"
#include <unistd.h>
using namespace mvi;

// Detach the multi-GPU frame loop: communicators, its stream, the pinned cull copies (device buffers stay in ctx->allocs until
// mv_destroy; the RCCL library stays loaded).
void mvi::distRelease(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (!d) return;
	for (uint32_t i = 0; i < kDistChains; ++i) if (d->comm[i] && d->commDestroy) (void)d->commDestroy(d->comm[i]);
	if (d->streamX) (void)hipStreamDestroy(d->streamX);
	evForget(ctx);
	if (d->reports) (void)hipHostFree(d->reports);
	for (void* p : d->peerMapped) (void)hipIpcCloseMemHandle(p);
	if (d->peerFlags) (void)hipFree(d->peerFlags);
	for (void* p : d->allocs) (void)hipFree(p);
	delete d;
	ctx->dist = nullptr;
	// the current frame's device state named buffers of the loop (the pinned exchange report, the peer tables): a stage called on it
	// now would write freed memory -- the next mv_update_frame starts clean
	ctx->st.exchangeReport = nullptr;
	ctx->st.peerCount = 0u; ctx->st.peerLight = nullptr; ctx->st.peerCubeMaps = nullptr; ctx->st.peerCubeDepths = nullptr; ctx->st.peerComposite = nullptr;
	ctx->haveCull = false; ctx->haveFrame = false;
}

extern "C" {

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
	for (uint32_t c = 0; c < d->numChains && d->transport != 3; ++c)	// (peer stores: nothing is packed, gathered or staged)
	{
		if ((rc = distAlloc(ctx, &d->sendBuf[c], d->sendCap)) != MV_OK) return rc;
		if ((rc = distAlloc(ctx, &d->recvBuf[c], d->sendCap * world)) != MV_OK) return rc;
		if ((rc = distAlloc(ctx, &d->rowSend[c], d->maxBandBytes, ctx->stream)) != MV_OK) return rc;
		if ((rc = distAlloc(ctx, &d->rowRecv[c], d->maxBandBytes * world)) != MV_OK) return rc;
	}
	// The shares of a frame's light march (all ranks' words + this rank's send copy): one buffer per chain -- or, with a single view
	// stream, one per LIGHT BUFFER, indexed by frame: there the light march of frame k + 1 (stream L) runs ahead of the all-gather and
	// unpack of frame k (the context's stream, behind view march k - 1) and must not write the words that gather has yet to send.  The
	// march already waits for frame k - lightBufs, the last reader of the light buffer it overwrites: the same wait now covers the words.
	d->numLightWords = d->chains ? d->numChains : ctx->st.lightBufs;
	for (uint32_t c = 0; c < d->numLightWords && d->transport != 3; ++c)
	{
		uint8_t* w = nullptr;
		if ((rc = distAlloc(ctx, &w, std::max<size_t>(d->lightShareBytes * world, 4096), ctx->stream)) != MV_OK) return rc;
		d->lightWords[c] = reinterpret_cast<uint32_t*>(w);
		if ((rc = distAlloc(ctx, &d->lightSend[c], std::max<size_t>(d->lightShareBytes, 4096), ctx->stream)) != MV_OK) return rc;
	}
	if (hipHostMalloc(reinterpret_cast<void**>(&d->reports), kMaxFrameSlots * 64, hipHostMallocCoherent) != hipSuccess) return fail(ctx, MV_ERR_OUT_OF_MEMORY, "pinned exchange reports");
	memset(d->reports, 0, kMaxFrameSlots * 64);	// per slot: [0] the cull's exchange report, [1] "the ranks' culls disagree", [2] "a wait for another rank gave up" (peer stores)
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i) d->reports[i * 8u] = ~0ull;
	if (world > 1u) ctx->st.flags |= mv::kFlagInterleaveRows;	// screen rows dealt by tile row instead of in contiguous bands
#ifndef MV_DIST_STATIC_OWNERS	// (A/B builds only: v % shard_count inside the loop too)
	ctx->st.flags |= mv::kFlagBalanceOwners;	// the cube-map volumes of a frame are dealt to the ranks by estimated cost (replicated cull: the same deal everywhere)
#endif
	ctx->st.flags |= mv::kFlagRotateAllLightMaps;	// every rank rotates the relit volume's light buffer: the exchange fills it on the non-owners
	if (d->transport == 3 && !d->peerDealVolumes) ctx->st.flags |= mv::kFlagDealTiles;	// (peer stores: the view march of every cube-map volume shared out by tiles; MV_DIST_PEER_DEAL_VOLUMES keeps whole volumes per rank)
	if (d->transport == 3) ctx->st.flags |= mv::kFlagLightLayers;	// (peer stores: the light march shared out by layers of voxel blocks -- whole cells, contiguous, pushed as plain ranges)
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
		ctx->st.flags &= ~(mv::kFlagRotateAllLightMaps | mv::kFlagBalanceOwners | mv::kFlagInterleaveRows | mv::kFlagLightLayers | mv::kFlagDealTiles);
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
	MV_REQUIRE(ctx, ctx->framesPerLaunch == 1u, MV_ERR_INVALID_ARG, "mv_config.frames_per_launch > 1 is the peer-store transport's (mv_dist_init_peer): a collective per frame cannot be shared by the frames of a launch");
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
	MV_REQUIRE(ctx, ctx->framesPerLaunch == 1u, MV_ERR_INVALID_ARG, "mv_config.frames_per_launch > 1 is the peer-store transport's (mv_dist_init_peer): a collective per frame cannot be shared by the frames of a launch");
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
	ctx->st.flags &= ~(mv::kFlagRotateAllLightMaps | mv::kFlagBalanceOwners | mv::kFlagInterleaveRows | mv::kFlagLightLayers | mv::kFlagDealTiles);
	return MV_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// peer-store transport (include/multivolumes.h "peer-store transport"; csrc/mv_peer.hip)
// ---------------------------------------------------------------------------------------------
enum { kStageLight = 0, kStageCube = 1, kStageRows = 2, kStageDone = 3 };

// what a rank tells the others about itself (MV_DIST_PEER_BYTES)
constexpr uint32_t kPeerBuffers = 2u + 3u * kMaxOutSets;	// flag block, light buffers, then per output set: cube maps, cube depths, frame buffer
struct PeerBlob
{
	uint32_t magic, version, rank, world;
	uint64_t pid, frame;
	uint32_t device, numOutSets, lightBufs, flagsKind;
	uint32_t N, L, V, S, W, H, cfgFlags, pad;
	char gpu[32];								// PCI bus id of the rank's GPU (ranks that share one keep the flag kernels: see mv_dist_init_peer)
	uint64_t bytes[kPeerBuffers];
	uint64_t pointer[kPeerBuffers];				// for contexts of the SAME process (an IPC handle cannot be opened by the process that made it)
	hipIpcMemHandle_t handle[kPeerBuffers];
};
static_assert(sizeof(PeerBlob) <= MV_DIST_PEER_BYTES, "MV_DIST_PEER_BYTES");
constexpr uint32_t kPeerMagic = 0x7265704du;	// "Mper"

static void peerBufferList(mv_ctx* ctx, void** ptr, uint64_t* bytes)
{
	const DeviceState& st = ctx->st;
	const size_t chain = mv::cubeChainTexels(st.N), L3 = (size_t)st.L * st.L * st.L;
	for (uint32_t i = 0; i < kPeerBuffers; ++i) { ptr[i] = nullptr; bytes[i] = 0; }
	ptr[0] = ctx->dist->peerFlags; bytes[0] = mv::kPeerFlagBytes;
	ptr[1] = st.lightBuffers; bytes[1] = (uint64_t)st.V * st.lightBufs * L3 * sizeof(uint4);
	for (uint32_t s = 0; s < ctx->numOutSets; ++s)
	{
		ptr[2 + 3 * s] = ctx->ob[s].cubeMaps; bytes[2 + 3 * s] = (uint64_t)st.V * chain * sizeof(uint2);
		ptr[3 + 3 * s] = ctx->ob[s].cubeDepths; bytes[3 + 3 * s] = (uint64_t)st.V * chain * sizeof(float);
		ptr[4 + 3 * s] = ctx->ob[s].composite; bytes[4 + 3 * s] = (uint64_t)st.W * compositeRows(st) * sizeof(uint2);
	}
}

static mv::PeerSync peerSyncBase(mv_ctx* ctx, uint32_t frameTag, uint32_t slot)
{
	DistState* d = ctx->dist;
	mv::PeerSync a;
	memset(&a, 0, sizeof(a));
	a.peerFlags = d->dPeerFlags; a.myFlags = d->peerFlags;
	a.world = ctx->st.shardCount; a.self = ctx->st.shardIndex; a.loopback = d->peerLoopback ? 1u : 0u;
	a.checkChain = 0xffffffffu; a.counters = ctx->fb[slot].counters; a.frameTag = frameTag;
	a.timeoutTicks = d->peerTimeoutTicks;
	a.report = d->reports + (size_t)slot * 8u;
	return a;
}

// the folded form of the same: the fields a launch carries in DeviceState.fold (mv_kernels.hpp)
enum { kFoldLight = 0, kFoldView = 1, kFoldComposite = 2 };
static void foldBase(mv_ctx* ctx, DeviceState& st, uint32_t kind, uint64_t frame, uint32_t frameTag, uint32_t slot)
{
	DistState* d = ctx->dist;
	memset(&st.fold, 0, sizeof(st.fold));
	st.fold.flagBlocks = d->dPeerFlags; st.fold.myFlags = d->peerFlags;
	st.fold.counter = d->dPeerCounters + kind * mv::kPeerMaxChains + (uint32_t)(frame % d->numChains);
	st.fold.report = d->reports + (size_t)slot * 8u;
	st.fold.timeoutTicks = d->peerTimeoutTicks;
	st.fold.loopback = d->peerLoopback ? 1u : 0u; st.fold.frameTag = frameTag; st.fold.checkChain = 0xffffffffu;
}
static void foldWait(mv_ctx* ctx, DeviceState& st, int stage, uint64_t frame)
{
	st.fold.waitRow[st.fold.nWait] = (uint32_t)stage * mv::kPeerMaxChains + (uint32_t)(frame % ctx->dist->numChains);
	st.fold.waitValue[st.fold.nWait++] = frame;
}
static void foldSignal(mv_ctx* ctx, DeviceState& st, int stage, uint64_t frame)
{
	st.fold.signalRow[st.fold.nSignal] = (uint32_t)stage * mv::kPeerMaxChains + (uint32_t)(frame % ctx->dist->numChains);
	st.fold.signalValue[st.fold.nSignal++] = frame;
}
static void foldClear(DeviceState& st) { st.fold.nWait = 0u; st.fold.nSignal = 0u; st.fold.checkChain = 0xffffffffu; st.fold.checkCompare = 0u; st.fold.stores = 0u; st.fold.pushBlocks = 0u; }
constexpr uint32_t kPeerLightPushBlocks = 224;	// workgroups (of the tile march's 128 threads) in front of a folded view march that push the light layers: 16 bytes x 4 per thread cover 1.8 MB

// in front of a consumer on stream s: every rank's `stage` of frame `frame` has landed here
static int peerWaitStage(mv_ctx* ctx, hipStream_t s, int stage, uint64_t frame, uint32_t frameTag, uint32_t slot)
{
	DistState* d = ctx->dist;
	mv::PeerSync a = peerSyncBase(ctx, frameTag, slot);
	a.nWait = 1; a.waitRow[0] = (uint32_t)stage * mv::kPeerMaxChains + (uint32_t)(frame % d->numChains); a.waitValue[0] = frame;
	mv::launchPeerSync(s, a);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
}

// A wait that gave up left {frame : row, column} in the slot's report (pinned); a cull that differed, {frame : 1}.
static int peerCheckReports(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i)
	{
		const unsigned long long c = *reinterpret_cast<volatile unsigned long long*>(d->reports + (size_t)i * 8u + 1u);
		if (c && d->cullCrosscheck)
			return fail(ctx, MV_ERR_INVALID_ARG, "the ranks' culls of a frame disagree (different frame constants on different ranks?): the ranks are not rendering the same frame");
		const unsigned long long t = *reinterpret_cast<volatile unsigned long long*>(d->reports + (size_t)i * 8u + 2u);
		if (t)
		{
			char msg[256];
			snprintf(msg, sizeof(msg), "peer-store transport: frame %u gave up waiting for rank %u (stage %u: 0 light share, 1 cube faces, 2 rows, 3 done) -- that rank is stalled, gone, or was handed other frames",
				(unsigned)(t >> 32), (unsigned)(t & 0xffu), (unsigned)(((t >> 8) & 0xffffu) / mv::kPeerMaxChains));
			return fail(ctx, MV_ERR_TIMEOUT, msg);
		}
	}
	return MV_OK;
}

// a wait of this rank's chain: as the consumer's own prologue (folded: the fields go into `st`, the next launch with them carries it) or
// as a one-wavefront kernel in front of it
struct PeerWaits { uint32_t n = 0; int stage[8]; uint64_t frame[8]; };
static int peerWaitFor(mv_ctx* ctx, hipStream_t s, DeviceState& st, uint32_t kind, const PeerWaits& w, uint64_t frame, uint32_t frameTag, uint32_t slot, bool compareCulls)
{
	DistState* d = ctx->dist;
	if (d->peerFold)
	{
		foldBase(ctx, st, kind, frame, frameTag, slot);
		for (uint32_t i = 0; i < w.n && i < 6u; ++i) foldWait(ctx, st, w.stage[i], w.frame[i]);
		if (compareCulls && d->cullCrosscheck) { st.fold.checkChain = (uint32_t)(frame % d->numChains); st.fold.checkCompare = 1u; }
		return MV_OK;
	}
	if (!w.n) return MV_OK;
	mv::PeerSync a = peerSyncBase(ctx, frameTag, slot);
	for (uint32_t i = 0; i < w.n; ++i) { a.waitRow[a.nWait] = (uint32_t)w.stage[i] * mv::kPeerMaxChains + (uint32_t)(w.frame[i] % d->numChains); a.waitValue[a.nWait++] = w.frame[i]; }
	if (compareCulls && d->cullCrosscheck) a.checkChain = (uint32_t)(frame % d->numChains);
	mv::launchPeerSync(s, a);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
}
// the fields of a push kernel's launch: which rows its last workgroup raises
static DeviceState peerPushState(mv_ctx* ctx, const DeviceState& st, uint32_t kind, uint64_t frame, uint32_t frameTag, uint32_t slot)
{
	DeviceState sp = st;
	foldBase(ctx, sp, kind, frame, frameTag, slot);
	return sp;
}

// The light march of frame k with peer stores: wait until every rank is done with the frames that still read what this frame overwrites,
// march this rank's layers of the relit map, push them into every rank's copy and raise the flag.
static int peerLight(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	const uint64_t k = d->peerFrame;
	const uint32_t C = d->numChains, lb = ctx->st.lightBufs;
	hipStream_t S = streamB(ctx);
	if (ctx->st.shardCount <= 1u) return mvi::marchLight(ctx, nullptr);
	// Frame k's light push overwrites, on every rank, the light buffer frame k - lightBufs was the last to read; its cube push, the
	// output set frame k - C composited (and frame k - C + 1 copied forward from).  "Done" is raised per chain, a row per chain: all
	// frames up to k - lightBufs are complete everywhere once the newest C of them are.
	PeerWaits w;
	for (uint32_t i = 0; i < C; ++i) if (k > lb + i) { w.stage[w.n] = kStageDone; w.frame[w.n++] = k - lb - i; }
	if (C < lb && k > C) { w.stage[w.n] = kStageDone; w.frame[w.n++] = k - C; }
	int rc = peerWaitFor(ctx, S, ctx->st, kFoldLight, w, k, ctx->st.frameTag, ctx->slot, false);
	if (rc == MV_OK) rc = mvi::marchLight(ctx, nullptr);
	foldClear(ctx->st);
	if (rc != MV_OK) return rc;
	if (d->peerFold)
	{
		// folded: the push is the leading workgroups of this frame's view march, the next kernel on this stream (peerBeforeView)
		MV_HIP(ctx, evRecord(ctx, ctx->evLight[ctx->slot], S));	// (this rank's layers are in place: what its own frames on the other streams order themselves by)
		d->bytes += (size_t)ctx->st.L * ctx->st.L * ctx->st.L * sizeof(uint4) / ctx->st.shardCount * (ctx->st.shardCount - 1u);
		return MV_OK;
	}
	DeviceState sp = peerPushState(ctx, ctx->st, kFoldLight, k, ctx->st.frameTag, ctx->slot);
	foldSignal(ctx, sp, kStageLight, k);
	if (d->cullCrosscheck) sp.fold.checkChain = (uint32_t)(k % C);	// (with what this rank's cull decided for the frame: compared by every rank's view march)
	mv::launchPeerPushLight(S, sp);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, evRecord(ctx, ctx->evLight[ctx->slot], S));	// (this rank's layers are in place and on their way: what the frames behind it on the other streams order themselves by; the other ranks' layers are the flags')
	d->bytes += (size_t)ctx->st.L * ctx->st.L * ctx->st.L * sizeof(uint4) / ctx->st.shardCount * (ctx->st.shardCount - 1u);
	return MV_OK;
}

// in front of frame k's view march: every rank's layers of the light maps relit in this frame and in the C - 1 before it (the frames on
// the other chains; earlier ones were waited for by this chain's previous march) have landed
static int peerBeforeView(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (ctx->st.shardCount <= 1u) return MV_OK;
	PeerWaits w;
	for (uint32_t j = 0; j < d->numChains; ++j) if (d->peerFrame > j) { w.stage[w.n] = kStageLight; w.frame[w.n++] = d->peerFrame - j; }
	const int rc = peerWaitFor(ctx, streamB(ctx), ctx->st, kFoldView, w, d->peerFrame, ctx->st.frameTag, ctx->slot, true);
	if (rc != MV_OK) return rc;
	// ...and what the march's last workgroup raises, behind the texels it stored into every rank's cube maps
	if (!d->peerFold) foldBase(ctx, ctx->st, kFoldView, d->peerFrame, ctx->st.frameTag, ctx->slot);	// (folded: peerWaitFor has filled the launch's fields in)
	foldSignal(ctx, ctx->st, kStageCube, d->peerFrame);
	ctx->st.fold.stores = 1u;	// (every launch of this march -- the overflow launch too, whichever of them waits or raises)
	if (d->peerFold)
	{
		// the leading workgroups of the main launch push this rank's light layers and raise "light layers landed" with the cull words
		ctx->st.fold.pushBlocks = kPeerLightPushBlocks;
		ctx->st.fold.pushRow = kStageLight * mv::kPeerMaxChains + (uint32_t)(d->peerFrame % d->numChains);
		ctx->st.fold.pushValue = d->peerFrame;
		ctx->st.fold.pushCounter = d->dPeerCounters + kFoldLight * mv::kPeerMaxChains + (uint32_t)(d->peerFrame % d->numChains);
		ctx->st.fold.pushCheckChain = d->cullCrosscheck ? (uint32_t)(d->peerFrame % d->numChains) : 0xffffffffu;
	}
	return MV_OK;
}

// Whatever is about to wait for the GPU (mv_synchronize, a readback, a setter) must also see the OTHER ranks' rows of the frames this
// rank has composited: their waits are enqueued here, lazily, instead of at the end of every frame's chain (where they would make a rank
// wait for the slowest rank's composite before it may reuse the stream).
int mvi::distCheck(mv_ctx* ctx) { return ctx->dist && ctx->dist->peerReady ? peerCheckReports(ctx) : MV_OK; }

int mvi::distBeforeSync(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (!d || !d->peerReady || ctx->st.shardCount <= 1u) return MV_OK;
	for (uint32_t c = 0; c < d->numChains; ++c)
	{
		const uint64_t f = d->rowsWaitFrame[c];
		if (!f) continue;
		d->rowsWaitFrame[c] = 0;
		mv::PeerSync a = peerSyncBase(ctx, (uint32_t)f, 0u);
		a.nWait = 1; a.waitRow[0] = kStageRows * mv::kPeerMaxChains + c; a.waitValue[0] = f;
		mv::launchPeerSync(ctx->streamB[d->rowsStream[c]], a);
		MV_HIP(ctx, hipGetLastError());
	}
	return MV_OK;
}

// Self-test of the mappings and of the flags, twice with different patterns (a stale cached line would show the first pattern in the
// second round): every rank writes its pattern into its region of every rank's frame buffer (output set 0), raises a flag, waits for
// everybody's, verifies all regions of its own buffer, and only then -- a second flag -- lets anybody overwrite them.
static int peerSelfTest(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	MV_REQUIRE(ctx, d->peerReady, MV_ERR_NOT_READY, "mv_dist_init_peer has not been called");
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	hipStream_t s = ctx->streamB[0];
	const size_t words = std::min<size_t>(65536, (size_t)ctx->st.W * compositeRows(ctx->st) / world);
	MV_REQUIRE(ctx, words >= 1, MV_ERR_RANGE, "self-test region");
	MV_HIP(ctx, hipMemsetAsync(d->dPeerMismatch, 0, 4, s));
	const uint32_t row = 4u * mv::kPeerMaxChains;	// the self-test's own row
	for (uint32_t round = 0; round < 2u; ++round)
	{
		const unsigned long long base = 2ull * (d->peerSelfTests++);
		mv::launchPeerPattern(s, d->dPeerComposite[0], world, d->peerLoopback ? 0u : rank, d->peerLoopback ? words * world : words, 17u + round);
		mv::PeerSync a = peerSyncBase(ctx, 0u, 0u);
		a.nSignal = 1; a.signalRow[0] = row; a.signalValue[0] = base + 1u; a.nWait = 1; a.waitRow[0] = row; a.waitValue[0] = base + 1u;
		mv::launchPeerSync(s, a);
		if (d->peerLoopback) mv::launchPeerVerify(s, ctx->ob[0].composite, 1u, words * world, 17u + round, d->dPeerMismatch);
		else mv::launchPeerVerify(s, ctx->ob[0].composite, world, words, 17u + round, d->dPeerMismatch);
		a.signalValue[0] = a.waitValue[0] = base + 2u;
		mv::launchPeerSync(s, a);
		MV_HIP(ctx, hipGetLastError());
	}
	uint32_t bad = 0;
	MV_HIP(ctx, hipMemcpyAsync(&bad, d->dPeerMismatch, 4, hipMemcpyDeviceToHost, s));
	MV_HIP(ctx, hipMemsetAsync(ctx->ob[0].composite, 0, (size_t)ctx->st.W * compositeRows(ctx->st) * sizeof(uint2), s));
	MV_HIP(ctx, hipStreamSynchronize(s));
	{ const int rc = peerCheckReports(ctx); if (rc != MV_OK) return rc; }
	MV_REQUIRE(ctx, bad == 0, MV_ERR_HIP, "peer-store self-test: a rank's pattern did not arrive (or arrived stale) in this rank's buffer");
	return MV_OK;
}

extern "C" {

int mv_dist_peer_export(mv_ctx* ctx, void* out)
{
	if (!ctx || !out) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, !ctx->dist, MV_ERR_INVALID_ARG, "mv_dist_init_* was already called on this context (mv_dist_shutdown first)");
	MV_REQUIRE(ctx, (ctx->cfg.flags & MV_FLAG_PIPELINE_FRAMES) && !(ctx->cfg.flags & MV_FLAG_SINGLE_VIEW_STREAM), MV_ERR_INVALID_ARG,
		"the peer-store transport needs whole-frame chains: a context created with MV_FLAG_PIPELINE_FRAMES and without MV_FLAG_SINGLE_VIEW_STREAM");
	MV_REQUIRE(ctx, ctx->st.shardCount <= mv::kPeerMaxRanks, MV_ERR_RANGE, "the peer-store transport serves at most 64 ranks (one node)");
	MV_REQUIRE(ctx, !ctx->rayQueue, MV_ERR_INVALID_ARG, "the peer-store transport stores from the tile march: not with MV_FLAG_RAY_QUEUE");
	MV_REQUIRE(ctx, ctx->framesPerLaunch == 1u || ctx->cullOnChain, MV_ERR_INVALID_ARG, "mv_config.frames_per_launch > 1 needs MV_FLAG_CULL_ON_VIEW_STREAM (a group's culls go in front of its marches on the group's view stream)");
	DistState* d = new (std::nothrow) DistState();
	if (!d) return MV_ERR_OUT_OF_MEMORY;
	d->transport = 3;
	d->chains = true; d->numChains = ctx->numViewStreams; d->numComm = 0;
	ctx->dist = d;
	auto bail = [&](int rc) { mvi::distRelease(ctx); return rc; };
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	// The flag block: device memory the OTHER ranks write and this rank's wait kernels poll -- uncached (every access goes to memory;
	// fine-grained or plain memory if the runtime refuses: between GPUs only the first is known to be coherent inside a running kernel)
	void* flags = nullptr;
	d->peerFlagsKind = 2;
	if (hipExtMallocWithFlags(&flags, mv::kPeerFlagBytes, hipDeviceMallocUncached) != hipSuccess)
	{
		(void)hipGetLastError();
		d->peerFlagsKind = 1;
		if (hipExtMallocWithFlags(&flags, mv::kPeerFlagBytes, hipDeviceMallocFinegrained) != hipSuccess)
		{
			(void)hipGetLastError();
			d->peerFlagsKind = 0;
			if (hipMalloc(&flags, mv::kPeerFlagBytes) != hipSuccess) return bail(fail(ctx, MV_ERR_OUT_OF_MEMORY, "the peer-store flag block"));
		}
	}
	d->peerFlags = static_cast<unsigned long long*>(flags);
	if (hipMemset(flags, 0, mv::kPeerFlagBytes) != hipSuccess) return bail(fail(ctx, MV_ERR_HIP, "clearing the peer-store flag block"));
	PeerBlob b;
	memset(&b, 0, sizeof(b));
	b.magic = kPeerMagic; b.version = MV_ABI_VERSION; b.rank = ctx->st.shardIndex; b.world = ctx->st.shardCount;
	b.pid = (uint64_t)getpid(); b.frame = ctx->frame; b.device = (uint32_t)ctx->cfg.device; b.numOutSets = ctx->numOutSets; b.lightBufs = ctx->st.lightBufs; b.flagsKind = (uint32_t)d->peerFlagsKind;
	b.N = ctx->st.N; b.L = ctx->st.L; b.V = ctx->st.V; b.S = ctx->st.S; b.W = ctx->st.W; b.H = ctx->st.H; b.cfgFlags = ctx->cfg.flags;
	if (hipDeviceGetPCIBusId(b.gpu, (int)sizeof(b.gpu), ctx->cfg.device) != hipSuccess) snprintf(b.gpu, sizeof(b.gpu), "device %d of pid %llu", ctx->cfg.device, (unsigned long long)b.pid);
	void* ptr[kPeerBuffers];
	peerBufferList(ctx, ptr, b.bytes);
	for (uint32_t i = 0; i < kPeerBuffers; ++i)
	{
		b.pointer[i] = reinterpret_cast<uint64_t>(ptr[i]);
		if (!ptr[i]) continue;
		const hipError_t e = hipIpcGetMemHandle(&b.handle[i], ptr[i]);
		if (e != hipSuccess && i == 0 && d->peerFlagsKind != 0)
		{
			// (a runtime that cannot export this kind of memory: a plain allocation instead)
			(void)hipGetLastError();
			(void)hipFree(d->peerFlags); d->peerFlags = nullptr;
			if (hipMalloc(&flags, mv::kPeerFlagBytes) != hipSuccess || hipMemset(flags, 0, mv::kPeerFlagBytes) != hipSuccess) return bail(fail(ctx, MV_ERR_OUT_OF_MEMORY, "the peer-store flag block"));
			d->peerFlags = static_cast<unsigned long long*>(flags); d->peerFlagsKind = 0; b.flagsKind = 0;
			ptr[0] = flags; b.pointer[0] = reinterpret_cast<uint64_t>(flags);
			const hipError_t e2 = hipIpcGetMemHandle(&b.handle[0], flags);
			if (e2 != hipSuccess) return bail(fail(ctx, MV_ERR_HIP, "hipIpcGetMemHandle (flag block)", e2));
			continue;
		}
		if (e != hipSuccess) return bail(fail(ctx, MV_ERR_HIP, "hipIpcGetMemHandle", e));
	}
	memset(out, 0, MV_DIST_PEER_BYTES);
	memcpy(out, &b, sizeof(b));
	d->peerExported = true;
	return MV_OK;
}

int mv_dist_init_peer(mv_ctx* ctx, const void* all, uint32_t flags)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	const bool loopback = (flags & MV_DIST_PEER_LOOPBACK) != 0;
	if (!ctx->dist && loopback)
	{
		uint8_t scratch[MV_DIST_PEER_BYTES];
		const int rc = mv_dist_peer_export(ctx, scratch);
		if (rc != MV_OK) return rc;
	}
	MV_REQUIRE(ctx, ctx->dist && ctx->dist->transport == 3 && ctx->dist->peerExported && !ctx->dist->peerReady, MV_ERR_NOT_READY, "mv_dist_peer_export has not been called on this context");
	MV_REQUIRE(ctx, all || loopback, MV_ERR_INVALID_ARG, "the ranks' export blobs");
	DistState* d = ctx->dist;
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	auto bail = [&](int rc) { mvi::distRelease(ctx); ctx->st.flags &= ~(mv::kFlagRotateAllLightMaps | mv::kFlagBalanceOwners | mv::kFlagInterleaveRows | mv::kFlagLightLayers | mv::kFlagDealTiles); return rc; };
	{ const hipError_t e = hipSetDevice(ctx->cfg.device); if (e != hipSuccess) return bail(fail(ctx, MV_ERR_HIP, "hipSetDevice", e)); }
	void* mine[kPeerBuffers]; uint64_t mineBytes[kPeerBuffers];
	peerBufferList(ctx, mine, mineBytes);
	std::vector<void*> table((size_t)kPeerBuffers * world, nullptr);	// [buffer][rank]
	bool distinctDevices = true;
	for (uint32_t r = 0; r < world && !loopback; ++r)
		for (uint32_t q = r + 1u; q < world; ++q)
			if (!memcmp(static_cast<const uint8_t*>(all) + (size_t)r * MV_DIST_PEER_BYTES + offsetof(PeerBlob, gpu), static_cast<const uint8_t*>(all) + (size_t)q * MV_DIST_PEER_BYTES + offsetof(PeerBlob, gpu), sizeof(PeerBlob::gpu))) distinctDevices = false;
	for (uint32_t r = 0; r < world; ++r)
	{
		if (r == rank) { for (uint32_t i = 0; i < kPeerBuffers; ++i) table[(size_t)i * world + r] = mine[i]; continue; }
		if (loopback)
		{
			// a rank measured alone: every other rank is a set of stand-in buffers of its own (never read: what the rank's kernels store
			// to seven peers must go to seven different places, as it would on a node -- seven stores to ONE address queue up behind each other)
			table[r] = mine[0];
			for (uint32_t i = 1; i < kPeerBuffers; ++i)
			{
				if (!mine[i]) continue;
				uint8_t* p = nullptr;
				const int rc = distAlloc(ctx, &p, mineBytes[i]);
				if (rc != MV_OK) return bail(rc);
				table[(size_t)i * world + r] = p;
			}
			continue;
		}
		PeerBlob b;
		memcpy(&b, static_cast<const uint8_t*>(all) + (size_t)r * MV_DIST_PEER_BYTES, sizeof(b));
		if (b.magic != kPeerMagic || b.version != MV_ABI_VERSION || b.rank != r || b.world != world) return bail(fail(ctx, MV_ERR_INVALID_ARG, "a peer's export blob is not that rank's (rank order? another library version?)"));
		if (b.N != ctx->st.N || b.L != ctx->st.L || b.V != ctx->st.V || b.S != ctx->st.S || b.W != ctx->st.W || b.H != ctx->st.H || b.numOutSets != ctx->numOutSets || b.lightBufs != ctx->st.lightBufs || b.cfgFlags != ctx->cfg.flags)
			return bail(fail(ctx, MV_ERR_INVALID_ARG, "the ranks' contexts differ in configuration: the peer-store transport addresses every rank's buffers by the same offsets"));
		if (b.frame != ctx->frame) return bail(fail(ctx, MV_ERR_INVALID_ARG, "the ranks' contexts differ in history (frames begun so far): their output sets and light buffers would rotate out of step"));
		for (uint32_t i = 0; i < kPeerBuffers; ++i)
		{
			if (!mine[i]) continue;
			if (b.bytes[i] != mineBytes[i]) return bail(fail(ctx, MV_ERR_INVALID_ARG, "a peer's buffer has another size"));
			if (b.pid == (uint64_t)getpid()) { table[(size_t)i * world + r] = reinterpret_cast<void*>(b.pointer[i]); continue; }	// a context of this process
			void* p = nullptr;
			const hipError_t e = hipIpcOpenMemHandle(&p, b.handle[i], hipIpcMemLazyEnablePeerAccess);
			if (e != hipSuccess) return bail(fail(ctx, MV_ERR_HIP, "hipIpcOpenMemHandle (is the peer on this node, HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", e));
			d->peerMapped.push_back(p);
			table[(size_t)i * world + r] = p;
		}
	}
	// device tables [world] per buffer
	{
		void** dev = nullptr;
		uint8_t* raw = nullptr;
		int rc = distAlloc(ctx, &raw, table.size() * sizeof(void*));
		if (rc != MV_OK) return bail(rc);
		dev = reinterpret_cast<void**>(raw);
		if (hipMemcpy(dev, table.data(), table.size() * sizeof(void*), hipMemcpyHostToDevice) != hipSuccess) return bail(fail(ctx, MV_ERR_HIP, "uploading the peer tables"));
		d->dPeerFlags = reinterpret_cast<unsigned long long**>(dev);
		d->dPeerLight = reinterpret_cast<uint4**>(dev + world);
		for (uint32_t s = 0; s < ctx->numOutSets; ++s)
		{
			d->dPeerCube[s] = reinterpret_cast<uint2**>(dev + (size_t)(2 + 3 * s) * world);
			d->dPeerDepth[s] = reinterpret_cast<float**>(dev + (size_t)(3 + 3 * s) * world);
			d->dPeerComposite[s] = reinterpret_cast<uint2**>(dev + (size_t)(4 + 3 * s) * world);
		}
		if ((rc = distAlloc(ctx, &raw, 64, ctx->stream)) != MV_OK) return bail(rc);
		d->dPeerMismatch = reinterpret_cast<uint32_t*>(raw);
		if ((rc = distAlloc(ctx, &raw, 3u * mv::kPeerMaxChains * sizeof(uint32_t), ctx->stream)) != MV_OK) return bail(rc);
		d->dPeerCounters = reinterpret_cast<uint32_t*>(raw);
	}
	d->peerLoopback = loopback;
	// Folded flags (raised / waited for inside the marches and the composite: three kernels of a rank's chain instead of eight) are for
	// ranks that own their GPU: a consumer's workgroups then spin on flags only OTHER GPUs raise.  Ranks that share a GPU (tests) keep the
	// flag kernels -- one wavefront each -- so that spinning consumers can never crowd the producers they wait for out of the GPU.
	d->peerFold = (flags & MV_DIST_PEER_FOLD) ? true : (flags & MV_DIST_PEER_NO_FOLD) ? false : (loopback || distinctDevices);
	if (ctx->framesPerLaunch > 1u)
	{
		// (several frames per launch exist in the folded form only: the launches that hold the frames carry their waits and flags)
		if (flags & MV_DIST_PEER_NO_FOLD) return bail(fail(ctx, MV_ERR_INVALID_ARG, "MV_DIST_PEER_NO_FOLD with mv_config.frames_per_launch > 1"));
		d->peerFold = true;
	}
	d->peerRowsRoot = (flags & MV_DIST_PEER_ROWS_TO_ROOT) != 0;
	d->peerDealVolumes = (flags & MV_DIST_PEER_DEAL_VOLUMES) != 0;
	d->peerFrame = 0;
	for (uint32_t c = 0; c < kDistChains; ++c) { d->rowsWaitFrame[c] = 0; d->rowsStream[c] = 0; }
	const int rc = distCommonInit(ctx);	// (releases everything on failure)
	if (rc != MV_OK) return rc;
	d->peerReady = true;
	return MV_OK;
}

int mv_dist_set_peer_timeout_ms(mv_ctx* ctx, uint32_t ms)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist && ctx->dist->transport == 3, MV_ERR_NOT_READY, "mv_dist_peer_export has not been called");
	ctx->dist->peerTimeoutTicks = (uint64_t)(ms ? ms : 1u) * 100000ull;	// wall_clock64 counts at 100 MHz
	return MV_OK;
}

} // extern "C"

extern "C" {

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
	if (d->transport == 3)
	{
		// Peer stores: the other ranks' view marches have written their faces into this rank's set -- wait (on the stream) until all
		// of them have.  Nothing here depends on a size, so the host waits for nothing: the cull's report only feeds the byte count.
		if (world <= 1u) return MV_OK;
		const unsigned long long rep = *reinterpret_cast<volatile unsigned long long*>(d->reports + (size_t)slot * 8u);
		if ((uint32_t)(rep >> 32) == st.frameTag) d->bytes += (size_t)(uint32_t)rep * 12u * (world - 1u);
		if (d->peerFold && d->pendingComposite && st.shardIndex < ((st.H + 7u) >> 3)) return MV_OK;	// (folded: the composite itself waits -- a rank with a row of the screen launches one)
		return peerWaitStage(ctx, X, kStageCube, d->framePending, st.frameTag, slot);
	}
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
		// The ranks' culls are compared by the kernel behind the frame's light exchange (lightUnpackKernel; peer stores: the light stage's
		// wait), which writes report[1].  Ordered, not best effort: that kernel has run once the frame's light event has -- a call ago in
		// steady state, so the query is all it costs.
		if (d->cullCrosscheck && hipEventQuery(ctx->evLight[slot]) != hipSuccess) MV_HIP(ctx, hipEventSynchronize(ctx->evLight[slot]));
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

// Peer stores: behind the composite of frame j on its stream: "this rank's rows of frame j have landed everywhere" and "this rank is done
// with frame j: its light buffers and output set may be overwritten".  (The copy-forward of frame j + 1 still reads frame j's set after
// this -- on this very stream.  The set is next written by the other ranks' view marches of frame j + C, which they start behind this
// rank's light share of frame j + C: a kernel of this same stream, behind that copy.)
static int distTailSignal(mv_ctx* ctx, hipStream_t X, uint32_t localChain, const DeviceState& st, uint64_t frame, uint32_t slot, bool composited)
{
	DistState* d = ctx->dist;
	if (d->transport != 3 || st.shardCount <= 1u) return MV_OK;
	const uint32_t c = (uint32_t)(frame % d->numChains);
	if (composited)
	{
		// push this rank's rows into every rank's frame buffer; its last workgroup raises both flags
		DeviceState sp = peerPushState(ctx, st, kFoldComposite, frame, st.frameTag, slot);
		foldSignal(ctx, sp, kStageRows, frame);
		foldSignal(ctx, sp, kStageDone, frame);
		sp.fold.rowsRoot = d->peerRowsRoot ? 1u : 0u;
		mv::launchPeerPushRows(X, sp);
		d->rowsWaitFrame[c] = frame; d->rowsStream[c] = localChain;
	}
	else
	{
		mv::PeerSync a = peerSyncBase(ctx, st.frameTag, slot);
		a.nSignal = 1; a.signalRow[0] = kStageDone * mv::kPeerMaxChains + c; a.signalValue[0] = frame;
		mv::launchPeerSync(X, a);
	}
	MV_HIP(ctx, hipGetLastError());
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
		if (d->transport == 3 && d->peerFold && world > 1u)
		{
			DeviceState stc = st;
			PeerWaits w; w.n = 1; w.stage[0] = kStageCube; w.frame[0] = d->framePending;
			const int rc = peerWaitFor(ctx, X, stc, kFoldComposite, w, d->framePending, st.frameTag, slot, false);
			if (rc != MV_OK) return rc;
			mv::launchComposite(X, stc, st.bandRow0, st.bandRow1);
		}
		else mv::launchComposite(X, st, st.bandRow0, st.bandRow1);
		MV_HIP(ctx, hipGetLastError());
		if (d->transport == 3) d->bytes += (size_t)st.H * st.W * sizeof(uint2) / world * (d->peerRowsRoot ? (st.shardIndex ? 0u : world - 1u) : world - 1u);	// (rows pushed into this rank's frame buffer)
		else if (world > 1)
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
	{ const int rc = distTailSignal(ctx, X, chain, st, d->framePending, slot, d->pendingComposite); if (rc != MV_OK) return rc; }
	// everything that reads this frame's buffers has been enqueued: the next users of its slot / output set / light buffers wait for this
	MV_HIP(ctx, evRecord(ctx, ctx->evMainDone[slot], X));
	d->pending = false;
	return MV_OK;
}

// ---------------------------------------------------------------------------------------------
// Several frames per launch (mv_config.frames_per_launch = B > 1; peer stores, folded flags, MV_FLAG_CULL_ON_VIEW_STREAM).
// A GROUP of up to B consecutive frames shares a view stream (group J on stream J % C) and one column of flag values (J): what was
// "frame k" to the flags is "group J" here.  Per mv_dist_frame call: the frame's upload, cull and orders go out at once (a cull needs the
// cull before it, they cannot be merged); the FIRST call of a group also enqueues the tail of the group before it -- the copy-forwards
// inside that group, ONE composite launch and ONE row push for all its frames -- and this frame's own copy-forward; the LAST call (or a
// flush) enqueues ONE light march and ONE view march for the group's frames (mv_kernels.hpp DeviceStateBatch: blockIdx.y = the frame).
// Ordering is the per-slot events' of the one-frame path (every frame records its own), plus stream order inside a group.
// ---------------------------------------------------------------------------------------------
#ifndef MV_BATCH_ONE_WAIT
#define MV_BATCH_ONE_WAIT 0	// 1: one wait kernel per group (in front of the light march: the done rows AND the older groups' light rows) instead of two -- measured slower (profiles/r04_frames_per_launch.txt)
#endif
static bool batchMode(const mv_ctx* ctx) { return ctx->dist && ctx->dist->transport == 3 && ctx->framesPerLaunch > 1u; }

// wait, on stream s, for the newest record on every OTHER stream among these events (an older record on the same stream is implied)
static int evWaitNewestPerStream(mv_ctx* ctx, hipStream_t s, hipEvent_t const* events, uint32_t n)
{
	hipStream_t seen[8]; uint64_t best[8]; hipEvent_t pick[8]; uint32_t m = 0;
	for (uint32_t i = 0; i < n; ++i)
	{
		mv_ctx::EvTrack* t = evFind(ctx, events[i]);
		if (!t || t->seq == 0 || t->on == s) continue;
		uint32_t j = 0;
		while (j < m && seen[j] != t->on) ++j;
		if (j == m) { if (m == 8u) { MV_HIP(ctx, evWait(ctx, s, events[i])); continue; } seen[m] = t->on; best[m] = 0; pick[m] = nullptr; ++m; }
		if (t->seq > best[j]) { best[j] = t->seq; pick[j] = events[i]; }
	}
	for (uint32_t j = 0; j < m; ++j) MV_HIP(ctx, evWait(ctx, s, pick[j]));
	return MV_OK;
}

// composite + row push of the pending group, on its stream
static int batchTail(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	const uint32_t n = d->nPend;
	if (!n) return MV_OK;
	const uint64_t P = d->pendBatch;
	const uint32_t C = d->numChains, chain = (uint32_t)(P % C);
	hipStream_t X = ctx->streamB[chain];
	const DistState::BatchFrame& last = d->pend[n - 1u];
	const uint32_t world = last.st.shardCount;
	const bool peers = world > 1u;
	uint32_t nComposite = 0;
	DeviceState sts[mv::kMaxBatch];
	for (uint32_t i = 0; i < n; ++i) if (d->pend[i].composite) sts[nComposite++] = d->pend[i].st;
	const uint32_t numSlots = ctx->numSlots, sets = ctx->numOutSets;
	// The copy-forwards first -- the one chain that runs through ALL frames (frame i's unmarched faces <- frame i - 1's set, complete only
	// once frame i - 1's own copy has run and every rank's marched faces of it have landed): they wait for the START of the group
	// before this one's tail (its faces landed, its copies), not for its composite and row push, so the tails of consecutive groups
	// overlap but for ~10 us per frame.
	{
		const DistState::BatchFrame& f0 = d->pend[0];
		const uint32_t first = f0.culled ? 1u : 0u;	// (a first frame begun the slow way has copied from every set already: distFrameBatch)
		const uint32_t prevSlot = (f0.slot + numSlots - 1u) % numSlots;
		if (!first)
		{
			MV_HIP(ctx, evWait(ctx, X, ctx->evMainDone[(f0.slot + numSlots - (sets % numSlots)) % numSlots]));	// the first frame's set: its last reader
			MV_HIP(ctx, evWait(ctx, X, ctx->evLanded[prevSlot]));
			MV_HIP(ctx, evWait(ctx, X, ctx->evCopy[prevSlot]));
		}
		// ONE kernel for the group's copies, one after the other inside it; in front of the first copy FROM a set of this group it waits
		// until every rank's marched faces of the group have landed (folded: one wavefront per workgroup polls)
		DeviceState cf[mv::kMaxBatch];
		uint32_t tags[mv::kMaxBatch];
		for (uint32_t i = 0; i < n; ++i) { cf[i] = d->pend[i].st; tags[i] = (uint32_t)d->pend[i].frame; }
		foldBase(ctx, cf[0], kFoldComposite, P, last.st.frameTag, last.slot);
		if (peers && n > 1u) foldWait(ctx, cf[0], kStageCube, P);
		if (n > first)
		{
			const OutBufs& other = ctx->ob[(f0.set + sets - 1u) % sets];
			mv::launchCopyForwardBatch(X, cf, first, n, other.cubeMaps, other.cubeDepths, other.faceStamps, tags);
			MV_HIP(ctx, hipGetLastError());
			for (uint32_t i = first; i < n; ++i) MV_HIP(ctx, evRecord(ctx, ctx->evCopy[d->pend[i].slot], X));
		}
	}
	// (a group of one frame: nothing above waited for the other ranks' faces -- a kernel of one wavefront does)
	if (peers && n == 1u) { const int rc = peerWaitStage(ctx, X, kStageCube, P, last.st.frameTag, last.slot); if (rc != MV_OK) return rc; }
	MV_HIP(ctx, evRecord(ctx, ctx->evLanded[last.slot], X));
	if (nComposite)
	{
		for (uint32_t i = 0; i < nComposite; ++i)
		{
			foldBase(ctx, sts[i], kFoldComposite, P, last.st.frameTag, last.slot);
		}
		EventPair evt = { nullptr, nullptr, 0 };
		beginStage(ctx, 3 | (int)(nComposite << 8), evt, X);
		mv::launchCompositeBatch(X, sts, nComposite);
		endStage(ctx, evt, X);
		MV_HIP(ctx, hipGetLastError());
		d->bytes += (size_t)nComposite * last.st.H * last.st.W * sizeof(uint2) / world * (d->peerRowsRoot ? (last.st.shardIndex ? 0u : world - 1u) : world - 1u);
	}
	if (peers)
	{
		if (nComposite)
		{
			// this rank's rows of every composited frame into every rank's frame buffers; the launch's last workgroup raises "rows landed" and "done"
			for (uint32_t i = 0; i < nComposite; ++i)
			{
				foldBase(ctx, sts[i], kFoldComposite, P, last.st.frameTag, last.slot);
				foldSignal(ctx, sts[i], kStageRows, P);
				foldSignal(ctx, sts[i], kStageDone, P);
				sts[i].fold.rowsRoot = d->peerRowsRoot ? 1u : 0u;
			}
			mv::launchPeerPushRowsBatch(X, sts, nComposite);
			d->rowsWaitFrame[chain] = P; d->rowsStream[chain] = chain;
		}
		else
		{
			mv::PeerSync a = peerSyncBase(ctx, last.st.frameTag, last.slot);
			a.nSignal = 1; a.signalRow[0] = kStageDone * mv::kPeerMaxChains + chain; a.signalValue[0] = P;
			mv::launchPeerSync(X, a);
		}
		MV_HIP(ctx, hipGetLastError());
	}
	// everything that reads these frames' buffers has been enqueued: the next users of their slots / output sets / light buffers wait for this
	for (uint32_t i = 0; i < n; ++i) MV_HIP(ctx, evRecord(ctx, ctx->evMainDone[d->pend[i].slot], X));
	d->nPend = 0;
	return MV_OK;
}

// light march + view march of the group being collected, on its stream
static int batchClose(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	const uint32_t n = d->nCur;
	if (!n) return MV_OK;
	const uint64_t J = d->peerFrame;
	const uint32_t C = d->numChains, chain = (uint32_t)(J % C);
	hipStream_t S = ctx->streamB[chain];
	const DistState::BatchFrame& last = d->cur[n - 1u];
	const bool peers = last.st.shardCount > 1u;
	const uint32_t numSlots = ctx->numSlots, sets = ctx->numOutSets, lb = ctx->st.lightBufs;
	DeviceState sts[mv::kMaxBatch];

	// ---- culls and orders: one launch each (the workgroups of frame i + 1 wait for frame i's inside it: mv_kernels.hip cullBatchKernel) ----
	{
		MV_SECTION(1);
		const uint32_t k0 = d->cur[0].culled ? 1u : 0u;
		if (n > k0)
		{
			for (uint32_t i = k0; i < n; ++i) sts[i] = d->cur[i].st;
			// the frames' constants: consecutive slots are consecutive slices, pinned and on the device -- one copy, two where the slots wrap around
			for (uint32_t i = k0; i < n; )
			{
				uint32_t run = 1;
				while (i + run < n && d->cur[i + run].slot == d->cur[i].slot + run) ++run;
				MV_HIP(ctx, hipMemcpyAsync(ctx->fb[d->cur[i].slot].dCb, ctx->pinned + (size_t)d->cur[i].slot * ctx->sliceBytes, (size_t)run * ctx->sliceBytes, hipMemcpyHostToDevice, S));
				i += run;
			}
			// (the cull of the frame before the group's first: the light table, VolumeInfo rows and view hash it left)
			MV_HIP(ctx, evWait(ctx, S, ctx->evStageA[(d->cur[k0].slot + numSlots - 1u) % numSlots]));
			EventPair evt = { nullptr, nullptr, 0 };
			beginStage(ctx, 0 | (int)((n - k0) << 8), evt, S);
			mv::launchCullBatch(S, sts + k0, n - k0);
			endStage(ctx, evt, S);
			mv::launchOrdersBatch(S, sts + k0, n - k0);
			MV_HIP(ctx, hipGetLastError());
			for (uint32_t i = k0; i < n; ++i) MV_HIP(ctx, evRecord(ctx, ctx->evStageA[d->cur[i].slot], S));
		}
		ctx->haveCull = true;
	}
	// ---- light marches: each frame relights its volume into that volume's freshly rotated buffer ----
	{
		MV_SECTION(5);
		hipEvent_t ev[2 * mv::kMaxBatch]; uint32_t m = 0;
		for (uint32_t i = 0; i < n; ++i) { ev[m++] = ctx->evMainDone[(d->cur[i].slot + numSlots - lb) % numSlots]; ev[m++] = ctx->evStageA[d->cur[i].slot]; }
		{ const int rc = evWaitNewestPerStream(ctx, S, ev, m); if (rc != MV_OK) return rc; }
		for (uint32_t i = 0; i < n; ++i)
		{
			sts[i] = d->cur[i].st;
			foldBase(ctx, sts[i], kFoldLight, J, last.st.frameTag, last.slot);
		}
		// Every rank is done with the groups whose frames still read what this group's light pushes and cube stores overwrite.  A
		// one-wavefront kernel in front of the launch, not the launch's own prologue: a launch that holds B frames would put B times the
		// workgroups on the GPU to spin -- enough, with the launches of the other streams doing the same, to keep the kernels they all
		// wait for (this rank's own tails on the other streams among them) from ever getting a slot.
		if (peers)
		{
			mv::PeerSync a = peerSyncBase(ctx, last.st.frameTag, last.slot);
			for (uint32_t c = 0; c < C; ++c) if (J > (uint64_t)C + c) { a.waitRow[a.nWait] = kStageDone * mv::kPeerMaxChains + (uint32_t)((J - C - c) % C); a.waitValue[a.nWait++] = J - C - c; }
#if MV_BATCH_ONE_WAIT	// (rejected: the light march then waits for the view march of the group before to start)
			// ... and, for the view march behind the light march: every rank's layers of the maps relit in the C - 1 groups before this one
			for (uint32_t c = 1; c < C; ++c)
				if (J > c)
				{
					if (a.nWait == 8u) { mv::launchPeerSync(S, a); a.nWait = 0; }	// (five chains: nine rows, a second kernel)
					a.waitRow[a.nWait] = kStageLight * mv::kPeerMaxChains + (uint32_t)((J - c) % C); a.waitValue[a.nWait++] = J - c;
				}
#endif
			if (a.nWait) { mv::launchPeerSync(S, a); MV_HIP(ctx, hipGetLastError()); }
		}
		EventPair evt = { nullptr, nullptr, 0 };
		beginStage(ctx, 1 | (int)(n << 8), evt, S);
		mv::launchMarchLightBatch(S, sts, n);
		endStage(ctx, evt, S);
		MV_HIP(ctx, hipGetLastError());
		for (uint32_t i = 0; i < n; ++i) MV_HIP(ctx, evRecord(ctx, ctx->evLight[d->cur[i].slot], S));
		if (peers) d->bytes += (size_t)n * last.st.L * last.st.L * last.st.L * sizeof(uint4) / last.st.shardCount * (last.st.shardCount - 1u);
	}
	// ---- view marches ----
	{
		MV_SECTION(6);
		hipEvent_t ev[3 * mv::kMaxBatch + kMaxFramesInFlight]; uint32_t m = 0;
		for (uint32_t i = 0; i < n; ++i)
		{
			const uint32_t slot = d->cur[i].slot;
			ev[m++] = ctx->evCopy[(slot + numSlots - (sets - 1u)) % numSlots];		// the copy-forward that read this set as its source
			ev[m++] = ctx->evMainDone[(slot + numSlots - (sets % numSlots)) % numSlots];	// the set's previous user
		}
		// this rank's light marches of the frames before these on the other streams (a view march reads every visible volume's light map)
		for (uint32_t j = 0; j < sets && j < numSlots; ++j) ev[m++] = ctx->evLight[(last.slot + numSlots - j) % numSlots];
		{ const int rc = evWaitNewestPerStream(ctx, S, ev, m); if (rc != MV_OK) return rc; }
		uint32_t tiles[mv::kMaxBatch], beyond[mv::kMaxBatch];
		bool overflow = false;
		for (uint32_t i = 0; i < n; ++i)
		{
			// the grid from this frame's own tile count if its cull has already reported it, else from the newest report there is, with a margin
			tiles[i] = 0xffffffffu;
			bool exact = false;
			const unsigned long long own = *reinterpret_cast<const volatile unsigned long long*>(ctx->pinnedTiles + 1u + d->cur[i].slot);
			const unsigned long long any = *reinterpret_cast<const volatile unsigned long long*>(ctx->pinnedTiles);
			if (own != ~0ull && (uint32_t)(own >> 32) == d->cur[i].st.frameTag) { tiles[i] = (uint32_t)own; exact = true; }
			else if (any != ~0ull) { const uint32_t count = (uint32_t)any; tiles[i] = count + count / 4u + 256u; }
			if (!exact && mv::marchTilesBeyond(d->cur[i].st, tiles[i]) != 0xffffffffu) overflow = true;
		}
		for (uint32_t i = 0; i < n; ++i)
		{
			sts[i] = d->cur[i].st;
			foldBase(ctx, sts[i], kFoldView, J, last.st.frameTag, last.slot);
			sts[i].fold.checkInfo = ctx->fb[last.slot].batchInfo;
			if (peers)
			{
				foldWait(ctx, sts[i], kStageLight, J);	// (row 0 = this group's own: waited for by the tiles that read a map relit in it; the older groups': the kernel below)
				if (d->cullCrosscheck) { sts[i].fold.checkChain = chain; sts[i].fold.checkCompare = 1u; }
				foldSignal(ctx, sts[i], kStageCube, J);
				sts[i].fold.stores = 1u;
				sts[i].fold.pushBlocks = kPeerLightPushBlocks;
				sts[i].fold.pushRow = kStageLight * mv::kPeerMaxChains + chain;
				sts[i].fold.pushValue = J;
				sts[i].fold.pushCounter = d->dPeerCounters + kFoldLight * mv::kPeerMaxChains + chain;
				sts[i].fold.pushCheckChain = d->cullCrosscheck ? chain : 0xffffffffu;
			}
		}
#if !MV_BATCH_ONE_WAIT
		// every rank's layers of the maps relit in the C - 1 groups before this one (on the other streams) have landed: again a kernel of
		// one wavefront in front of the launch instead of a wait in every workgroup of it
		if (peers && J > 1u)
		{
			mv::PeerSync a = peerSyncBase(ctx, last.st.frameTag, last.slot);
			for (uint32_t c = 1; c < C; ++c) if (J > c) { a.waitRow[a.nWait] = kStageLight * mv::kPeerMaxChains + (uint32_t)((J - c) % C); a.waitValue[a.nWait++] = J - c; }
			if (a.nWait) { mv::launchPeerSync(S, a); MV_HIP(ctx, hipGetLastError()); }
		}
#endif
		// (the main launch waits; the LAST launch raises "cube faces landed": the overflow launch if there is one)
		DeviceState main[mv::kMaxBatch];
		for (uint32_t i = 0; i < n; ++i) { main[i] = sts[i]; if (overflow) main[i].fold.nSignal = 0u; }
		EventPair evt = { nullptr, nullptr, 0 };
		beginStage(ctx, 2 | (int)(n << 8), evt, S);
		mv::launchMarchTilesBatch(S, main, n, tiles, beyond);
		if (overflow)
		{
			for (uint32_t i = 0; i < n; ++i) { sts[i].fold.nWait = 0u; sts[i].fold.pushBlocks = 0u; }
			mv::launchMarchTilesOverflowBatch(S, sts, n, beyond);
		}
		endStage(ctx, evt, S);
		MV_HIP(ctx, hipGetLastError());
	}
	for (uint32_t i = 0; i < n; ++i) d->pend[i] = d->cur[i];
	d->nPend = n; d->pendBatch = J; d->nCur = 0;
	return MV_OK;
}

static int distFrameBatch(mv_ctx* ctx, const void* cbPerFrame, const void* perObject, int with_composite)
{
	DistState* d = ctx->dist;
	MV_REQUIRE(ctx, d->peerReady, MV_ERR_NOT_READY, "mv_dist_init_peer has not been called (mv_dist_peer_export alone attaches nothing)");
	MV_REQUIRE(ctx, ctx->haveDepth, MV_ERR_NOT_READY, "no depth map (mv_set_depth_map)");
	MV_REQUIRE(ctx, ctx->haveShadow, MV_ERR_NOT_READY, "no shadow map (mv_set_shadow_map)");
	int rc;
	if ((rc = peerCheckReports(ctx)) != MV_OK) return rc;
	const bool first = d->nCur == 0u;
	if (first) ++d->peerFrame;	// (counts groups here)
	struct Lean { mv_ctx* c; ~Lean() { c->leanEvents = false; } } lean = { ctx };
	ctx->leanEvents = true;
#ifdef MV_HOST_PROFILE
	++gSectionCalls;
#endif
	ctx->batchExplicit = true; ctx->batchIndex = d->peerFrame; ctx->batchPos = d->nCur;
	ctx->deferUpload = true;
	{ MV_SECTION(0); rc = mv_update_frame(ctx, cbPerFrame, perObject); }	// (the constants into their pinned slice; upload and cull of the group's frames go out together, batchClose)
	ctx->batchExplicit = false; ctx->deferUpload = false;
	if (rc != MV_OK) return rc;
	bool culled = false;
	if (first)
	{
		// the group before this one: its composite and row push
		{ MV_SECTION(4); if ((rc = batchTail(ctx)) != MV_OK) return rc; }
		if (ctx->lastCompleteFrame != 0u && ctx->lastCompleteFrame + 1u != ctx->frame)
		{
			// (the frame before this one left its output set untouched -- frames outside the loop in between: this frame's copy-forward
			// takes the slow way, from every set, which needs its cull now)
			MV_SECTION(3);
			MV_HIP(ctx, evWait(ctx, ctx->streamA, ctx->evStageA[(ctx->slot + ctx->numSlots - 1u) % ctx->numSlots]));
			MV_HIP(ctx, hipMemcpyAsync(ctx->fb[ctx->slot].dCb, ctx->pinned + (size_t)ctx->slot * ctx->sliceBytes, ctx->sliceBytes, hipMemcpyHostToDevice, ctx->streamA));
			if ((rc = mv_cull(ctx)) != MV_OK) return rc;
			if ((rc = viewCopyForward(ctx)) != MV_OK) return rc;
			culled = true;
		}
	}
	ctx->lastCompleteFrame = ctx->frame;	// (its copy-forward goes out with the group: batchClose for the first frame, the group's tail for the others)
	DistState::BatchFrame& f = d->cur[d->nCur++];
	f.st = ctx->st; f.slot = ctx->slot; f.set = ctx->set; f.frame = ctx->frame; f.composite = with_composite != 0; f.culled = culled; f.prevChain = ctx->prevChain;
	ctx->marchedThisFrame = true; ctx->lightMarched = true;
	if (d->nCur == ctx->framesPerLaunch) return batchClose(ctx);
	return MV_OK;
}

int mvi::distTail(mv_ctx* ctx)
{
	if (batchMode(ctx))
	{
		const int rc = batchClose(ctx);
		return rc != MV_OK ? rc : batchTail(ctx);
	}
	const int rc = distTailExchange(ctx);
	return rc != MV_OK ? rc : distTailComposite(ctx);
}

// The light march of frame k, shared out over the ranks by voxel blocks (every rank holds every volume): this rank marches its
// 1 / shard_count of the blocks, into its own copy of the relit volume's fresh light buffer and, as plain words, into its share of the
// all-gather; the other ranks' shares are then unpacked into the local copy.  Every rank keeps every light map current, because
// every rank marches its own screen rows of every volume on the screen-space scheme.
bool mvi::distLightOnStreamA(const mv_ctx* ctx)
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
	if (d->transport == 3) return peerLight(ctx);
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	const uint32_t wordsIdx = d->chains ? chain : (uint32_t)(ctx->frame % d->numLightWords);	// (single view stream: see distCommonInitBody)
	uint8_t* lightAll = reinterpret_cast<uint8_t*>(d->lightWords[wordsIdx]);
	uint8_t* lightMine = d->inPlaceGather ? lightAll + (size_t)rank * d->lightShareBytes : d->lightSend[wordsIdx];
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
		mv::launchLightUnpack(E, ctx->st, d->lightWords[wordsIdx]);
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
	if (batchMode(ctx)) return distFrameBatch(ctx, cbPerFrame, perObject, with_composite);
	if (d->transport == 3)
	{
		MV_REQUIRE(ctx, d->peerReady, MV_ERR_NOT_READY, "mv_dist_init_peer has not been called (mv_dist_peer_export alone attaches nothing)");
		if ((rc = peerCheckReports(ctx)) != MV_OK) return rc;
		++d->peerFrame;
	}
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
		if (d->transport == 3 && (rc = peerBeforeView(ctx)) != MV_OK) return rc;	// (peer stores: every rank's light layers; folded into the march or a kernel in front of it)
		{ MV_SECTION(6); rc = viewMarch(ctx); }
		if (d->transport == 3) foldClear(ctx->st);
		if (rc != MV_OK) return rc;
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
	{ MV_SECTION(7); d->stPending = ctx->st; d->slotPending = ctx->slot; d->chainPending = chain; d->pendingComposite = with_composite != 0; d->pending = true; d->framePending = d->peerFrame; }
	return MV_OK;
}

int mv_dist_flush(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	const int rc = distTail(ctx);
	return rc != MV_OK ? rc : distCheck(ctx);
}

// One all-gather (separate buffers), one in-place all-gather and one all-reduce on each communicator, checked on the host: run once
// after mv_dist_init_* so that a broken transport fails before any frame does.
int mv_dist_selftest(mv_ctx* ctx)
{
	if (!ctx) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist, MV_ERR_NOT_READY, "mv_dist_init_rccl / mv_dist_init_callbacks has not been called");
	DistState* d = ctx->dist;
	if (d->transport == 3) return peerSelfTest(ctx);
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
