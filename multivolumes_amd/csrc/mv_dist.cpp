// mv_dist.cpp -- multi-GPU: the exchange building blocks (mv_exchange_*) and the frame loop inside the library (mv_dist_*)
// with its transports.  The reference is single-device (MultiVolumes.cpp:159-187): there is no call to match, the data flow is SURVEY 8(e)'s.
#include "mv_dist_int.hpp"
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
	ctx->st.peerCount = 0u; ctx->st.peerLight = nullptr; ctx->st.peerCubeMaps = nullptr; ctx->st.peerCubeDepths = nullptr; ctx->st.peerComposite = nullptr; ctx->st.lightPlain = nullptr; ctx->st.peerLightPlain = nullptr;
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
	const uint32_t d = mv::cubeMipDim(ctx->st.N, minMip);
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
int distAlloc(mv_ctx* ctx, uint8_t** out, size_t bytes, hipStream_t zeroOn)
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
int distCommonInit(mv_ctx* ctx)
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
static int distTailSignal(mv_ctx* ctx, hipStream_t X, uint32_t localChain, const DeviceState& st, uint64_t frame, uint32_t slot, bool composited, bool wanted)
{
	DistState* d = ctx->dist;
	if (d->transport != 3 || st.shardCount <= 1u) return MV_OK;
	const uint32_t c = (uint32_t)(frame % d->numChains);
	{
		// "done", and -- behind a composite, whose launches have stored this rank's rows into the other ranks' frame buffers themselves -- "rows
		// landed": a kernel of one wavefront; the kernel boundary in front of it is what says that those stores have been acknowledged
		mv::PeerSync a = peerSyncBase(ctx, st.frameTag, slot);
		a.nSignal = 1; a.signalRow[0] = kStageDone * mv::kPeerMaxChains + c; a.signalValue[0] = frame;
		if (wanted || composited)
		{
			// (the composite has stored this rank's rows into the other ranks' frame buffers itself -- or: a composite was asked for and this rank has no row of the screen: its -- empty -- share of the rows has landed all the same)
			a.nSignal = 2; a.signalRow[1] = kStageRows * mv::kPeerMaxChains + c; a.signalValue[1] = frame;
			d->rowsWaitFrame[c] = frame; d->rowsStream[c] = localChain;
		}
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
	bool compositeLaunched = false;	// (a rank without a row of the screen launches none: the flags are raised by a kernel of one wavefront then)
	if (d->pendingComposite)
	{
		// chains: the copy-forward of this frame's unmarched faces ran on the stream of the frame before it
		if (d->chains) MV_HIP(ctx, evWait(ctx, X, ctx->evCopy[slot]));
		uint32_t launched = 0;
		EventPair evt = { nullptr, nullptr, 0 };	// (profiling only: the composite's own time -- the bench line's kernel_ms.composite was 0 on this path until round 6)
		if (d->transport == 3 && world > 1u)
		{
			// peer stores: the composite waits for every rank's cube faces itself (folded; else a kernel in front of it has) and stores every
			// pixel it writes into the other ranks' frame buffers (rank 0's only: MV_DIST_PEER_ROWS_TO_ROOT); distTailSignal raises the flags
			DeviceState stc = st;
			PeerWaits w;
			if (d->peerFold) { w.n = 1; w.stage[0] = kStageCube; w.frame[0] = d->framePending; }
			const int rc = peerWaitFor(ctx, X, stc, kFoldComposite, w, d->framePending, st.frameTag, slot, false);
			if (rc != MV_OK) return rc;
			if (!d->peerFold) foldBase(ctx, stc, kFoldComposite, d->framePending, st.frameTag, slot);
			stc.fold.stores = 1u; stc.fold.rowsRoot = d->peerRowsRoot ? 1u : 0u;
			beginStage(ctx, 3, evt, X);
			launched = mv::launchComposite(X, stc, st.bandRow0, st.bandRow1);
		}
		else { beginStage(ctx, 3, evt, X); launched = mv::launchComposite(X, st, st.bandRow0, st.bandRow1); }
		endStage(ctx, evt, X);
		compositeLaunched = launched != 0u;
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
	{ const int rc = distTailSignal(ctx, X, chain, st, d->framePending, slot, d->pendingComposite && compositeLaunched, d->pendingComposite); if (rc != MV_OK) return rc; }
	// everything that reads this frame's buffers has been enqueued: the next users of its slot / output set / light buffers wait for this
	MV_HIP(ctx, evRecord(ctx, ctx->evMainDone[slot], X));
	MV_HIP(ctx, latencyMark(ctx, slot, X));
	d->pending = false;
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

int mv_dist_peer_info(mv_ctx* ctx, uint32_t* out)
{
	if (!ctx || !out) return MV_ERR_INVALID_ARG;
	for (uint32_t i = 0; i < 8u; ++i) out[i] = 0u;
	const DistState* d = ctx->dist;
	if (!d || d->transport != 3) return MV_OK;
	out[0] = d->peerFold ? 1u : 0u; out[1] = d->peerAcquire ? 1u : 0u; out[2] = d->peerRowsRoot ? 1u : 0u; out[3] = d->peerDealVolumes ? 1u : 0u;
	out[4] = (d->peerPlainLight && d->lightPlain) ? (uint32_t)sizeof(uint32_t) : (uint32_t)sizeof(uint4);	// bytes a rank pushes per texel of its share of the relit light map
	out[5] = (uint32_t)d->peerFlagsKind; out[6] = d->peerStaleWords; out[7] = d->peerFirstReadBad;
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
