// mv_dist_peer.cpp -- the peer-store transport of the multi-GPU frame loop (mv_dist_peer_export / mv_dist_init_peer: IPC-mapped buffers, flags,
// no collectives; csrc/mv_peer.hip) and its form with several frames per launch (mv_config.frames_per_launch).  The loop itself, the
// collective transports and the exchange building blocks: mv_dist.cpp.
#include "mv_dist_int.hpp"
#include <unistd.h>
using namespace mvi;

// ---------------------------------------------------------------------------------------------
// peer-store transport (include/multivolumes.h "peer-store transport"; csrc/mv_peer.hip)
// ---------------------------------------------------------------------------------------------

// what a rank tells the others about itself (MV_DIST_PEER_BYTES)
constexpr uint32_t kPeerBuffers = 3u + 3u * kMaxOutSets;	// flag block, light buffers, then per output set: cube maps, cube depths, frame buffer; last: the staging maps of the plain light push
struct PeerBlob
{
	uint32_t magic, version, rank, world;
	uint64_t pid, frame;
	uint64_t nonce;								// a random number of the exporting PROCESS (pids repeat across pid namespaces: containers)
	uint32_t device, numOutSets, lightBufs, flagsKind;
	uint32_t N, L, V, S, W, H, cfgFlags, pad;
	char gpu[32];								// PCI bus id of the rank's GPU (ranks that share one keep the flag kernels: see mv_dist_init_peer)
	uint64_t bytes[kPeerBuffers];
	uint64_t pointer[kPeerBuffers];				// for contexts of the SAME process (an IPC handle cannot be opened by the process that made it)
	hipIpcMemHandle_t handle[kPeerBuffers];
};
static_assert(sizeof(PeerBlob) <= MV_DIST_PEER_BYTES, "MV_DIST_PEER_BYTES");
constexpr uint32_t kPeerMagic = 0x7265704du;	// "Mper"

static uint64_t processNonce()
{
	static const uint64_t n = [] {
		uint64_t v = 0;
		FILE* f = fopen("/dev/urandom", "rb");
		if (f) { if (fread(&v, sizeof(v), 1, f) != 1) v = 0; fclose(f); }
		if (!v) v = ((uint64_t)getpid() << 32) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ reinterpret_cast<uint64_t>(&v);
		return v;
	}();
	return n;
}
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
	ptr[kPeerBuffers - 1u] = ctx->dist->lightPlain; bytes[kPeerBuffers - 1u] = ctx->dist->lightPlain ? (uint64_t)ctx->numSlots * L3 * sizeof(uint32_t) : 0u;
}

mv::PeerSync peerSyncBase(mv_ctx* ctx, uint32_t frameTag, uint32_t slot)
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
void foldBase(mv_ctx* ctx, DeviceState& st, uint32_t kind, uint64_t frame, uint32_t frameTag, uint32_t slot)
{
	DistState* d = ctx->dist;
	memset(&st.fold, 0, sizeof(st.fold));
	st.fold.flagBlocks = d->dPeerFlags; st.fold.myFlags = d->peerFlags;
	st.fold.counter = d->dPeerCounters + kind * mv::kPeerMaxChains + (uint32_t)(frame % d->numChains);
	st.fold.report = d->reports + (size_t)slot * 8u;
	st.fold.timeoutTicks = d->peerTimeoutTicks;
	st.fold.loopback = d->peerLoopback ? 1u : 0u; st.fold.frameTag = frameTag; st.fold.checkChain = 0xffffffffu;
	st.fold.acquire = d->peerAcquire ? 1u : 0u;
}
void foldWait(mv_ctx* ctx, DeviceState& st, int stage, uint64_t frame)
{
	st.fold.waitRow[st.fold.nWait] = (uint32_t)stage * mv::kPeerMaxChains + (uint32_t)(frame % ctx->dist->numChains);
	st.fold.waitValue[st.fold.nWait++] = frame;
}
void foldSignal(mv_ctx* ctx, DeviceState& st, int stage, uint64_t frame)
{
	st.fold.signalRow[st.fold.nSignal] = (uint32_t)stage * mv::kPeerMaxChains + (uint32_t)(frame % ctx->dist->numChains);
	st.fold.signalValue[st.fold.nSignal++] = frame;
}
void foldClear(DeviceState& st) { st.fold.nWait = 0u; st.fold.nSignal = 0u; st.fold.checkChain = 0xffffffffu; st.fold.checkCompare = 0u; st.fold.stores = 0u; st.fold.pushBlocks = 0u; st.fold.rebuildBlocks = 0u; st.fold.waitLocal = 0u; }
#ifndef MV_PEER_LIGHT_PUSH_BLOCKS
#define MV_PEER_LIGHT_PUSH_BLOCKS 224
#endif
#ifndef MV_PEER_REBUILD_BLOCKS
#define MV_PEER_REBUILD_BLOCKS 128
#endif
constexpr uint32_t kPeerLightPushBlocks = MV_PEER_LIGHT_PUSH_BLOCKS;	// workgroups (of the tile march's 128 threads) in front of a folded view march that push the light layers: 16 bytes x 4 per thread cover 1.8 MB

// in front of a consumer on stream s: every rank's `stage` of frame `frame` has landed here
int peerWaitStage(mv_ctx* ctx, hipStream_t s, int stage, uint64_t frame, uint32_t frameTag, uint32_t slot)
{
	DistState* d = ctx->dist;
	mv::PeerSync a = peerSyncBase(ctx, frameTag, slot);
	a.nWait = 1; a.waitRow[0] = (uint32_t)stage * mv::kPeerMaxChains + (uint32_t)(frame % d->numChains); a.waitValue[0] = frame;
	mv::launchPeerSync(s, a);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
}

// A wait that gave up left {frame : row, column} in the slot's report (pinned); a cull that differed, {frame : 1}.
int peerCheckReports(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	// (timeouts first, in every slot: a wait that gave up goes on to compare its cull words with whatever the stalled rank's last frame left there,
	// and "the culls disagree" would hide the stall -- round 6, tests/test_dist_frame_loop.py::test_a_stalled_rank_is_reported_soon_after_the_timeout)
	// (and a wait for ANOTHER rank before a wait for a stage of this rank's own: the kernel that would have raised that one is itself waiting for the stalled rank)
	unsigned long long t = 0ull;
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i)
	{
		const unsigned long long ti = *reinterpret_cast<volatile unsigned long long*>(d->reports + (size_t)i * 8u + 2u);
		if (ti && (!t || ((t & 0xffu) == ctx->st.shardIndex && (ti & 0xffu) != ctx->st.shardIndex))) t = ti;
	}
	if (t)
	{
		char msg[320];
		snprintf(msg, sizeof(msg), "peer-store transport: frame %u gave up waiting for rank %u (stage %u: 0 light share, 1 cube faces, 2 rows, 3 done, 4 this rank's own rebuild of the light cells) -- that rank is stalled, gone, or was handed other frames",
			(unsigned)(t >> 32), (unsigned)(t & 0xffu), (unsigned)(((t >> 8) & 0xffffu) / mv::kPeerMaxChains));
		return fail(ctx, MV_ERR_TIMEOUT, msg);
	}
	for (uint32_t i = 0; i < kMaxFrameSlots; ++i)
	{
		const unsigned long long c = *reinterpret_cast<volatile unsigned long long*>(d->reports + (size_t)i * 8u + 1u);
		if (c && d->cullCrosscheck)
			return fail(ctx, MV_ERR_INVALID_ARG, "the ranks' culls of a frame disagree (different frame constants on different ranks?): the ranks are not rendering the same frame");
	}
	return MV_OK;
}

// a wait of this rank's chain: as the consumer's own prologue (folded: the fields go into `st`, the next launch with them carries it) or
// as a one-wavefront kernel in front of it
int peerWaitFor(mv_ctx* ctx, hipStream_t s, DeviceState& st, uint32_t kind, const PeerWaits& w, uint64_t frame, uint32_t frameTag, uint32_t slot, bool compareCulls)
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
DeviceState peerPushState(mv_ctx* ctx, const DeviceState& st, uint32_t kind, uint64_t frame, uint32_t frameTag, uint32_t slot)
{
	DeviceState sp = st;
	foldBase(ctx, sp, kind, frame, frameTag, slot);
	return sp;
}

// The light march of frame k with peer stores: wait until every rank is done with the frames that still read what this frame overwrites,
// march this rank's layers of the relit map, push them into every rank's copy and raise the flag.
int peerLight(mv_ctx* ctx)
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
		d->bytes += (size_t)ctx->st.L * ctx->st.L * ctx->st.L * (ctx->st.lightPlain ? sizeof(uint32_t) : sizeof(uint4)) / ctx->st.shardCount * (ctx->st.shardCount - 1u);
		return MV_OK;
	}
	DeviceState sp = peerPushState(ctx, ctx->st, kFoldLight, k, ctx->st.frameTag, ctx->slot);
	foldSignal(ctx, sp, kStageLight, k);
	if (d->cullCrosscheck) sp.fold.checkChain = (uint32_t)(k % C);	// (with what this rank's cull decided for the frame: compared by every rank's view march)
	mv::launchPeerPushLight(S, sp);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, evRecord(ctx, ctx->evLight[ctx->slot], S));	// (this rank's layers are in place and on their way: what the frames behind it on the other streams order themselves by; the other ranks' layers are the flags')
	d->bytes += (size_t)ctx->st.L * ctx->st.L * ctx->st.L * (ctx->st.lightPlain ? sizeof(uint32_t) : sizeof(uint4)) / ctx->st.shardCount * (ctx->st.shardCount - 1u);
	return MV_OK;
}

// in front of frame k's view march: every rank's layers of the light maps relit in this frame and in the C - 1 before it (the frames on
// the other chains; earlier ones were waited for by this chain's previous march) have landed
int peerBeforeView(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	if (ctx->st.shardCount <= 1u) return MV_OK;
	const bool plain = ctx->st.lightPlain != nullptr;
	const uint32_t C = d->numChains, chain = (uint32_t)(d->peerFrame % C);
	PeerWaits w;
	if (d->peerFold)
	{
		// Folded: only THIS frame's row is waited for inside the march (by the wavefronts whose tile reads the map relit in this frame; its own
		// rank's column is raised by the push workgroups of this very launch, dispatched in front of them).  The rows of the C - 1 frames before
		// it -- on the other streams -- are waited for by a kernel of one wavefront in front of the launch, as batchClose does: their columns
		// include THIS rank's, raised by the push workgroups of launches on other streams, which a march whose every workgroup spins in its
		// prologue could keep from ever being dispatched (round 4's form) -- and every spinning workgroup would end its wait with
		// an invalidate of its CU's L1 (profiles/r05_peer_acquire.txt).  (Plain light push: what the older frames' readers need is this rank's
		// REBUILT cells -- the local row its rebuild workgroups raise, which implies that the texels had landed.)
		mv::PeerSync a = peerSyncBase(ctx, ctx->st.frameTag, ctx->slot);
		const uint32_t stage = plain ? kStageRebuilt : kStageLight;
		for (uint32_t j = 1; j < C; ++j) if (d->peerFrame > j) { a.waitRow[a.nWait] = stage * mv::kPeerMaxChains + (uint32_t)((d->peerFrame - j) % C); a.waitValue[a.nWait++] = d->peerFrame - j; }
		a.waitLocal = plain ? 1u : 0u;
		if (a.nWait) { mv::launchPeerSync(streamB(ctx), a); MV_HIP(ctx, hipGetLastError()); }
		w.stage[w.n] = kStageLight; w.frame[w.n++] = d->peerFrame;
		if (plain) { w.stage[w.n] = kStageRebuilt; w.frame[w.n++] = d->peerFrame; }
	}
	else
	{
		// waits as kernels: every rank's layers (texels) of this frame's map, then -- plain light push -- this rank's rebuild of them as a kernel
		// of its own, which raises the local row; the maps relit in the C - 1 frames before are waited for by their (local) rebuilt rows
		if (plain)
		{
			mv::PeerSync a = peerSyncBase(ctx, ctx->st.frameTag, ctx->slot);
			a.waitRow[a.nWait] = kStageLight * mv::kPeerMaxChains + chain; a.waitValue[a.nWait++] = d->peerFrame;
			if (d->cullCrosscheck) a.checkChain = chain;
			mv::launchPeerSync(streamB(ctx), a);
			MV_HIP(ctx, hipGetLastError());
			DeviceState sr = ctx->st;
			foldBase(ctx, sr, kFoldRebuild, d->peerFrame, ctx->st.frameTag, ctx->slot);
			sr.fold.rebuildRow = kStageRebuilt * mv::kPeerMaxChains + chain; sr.fold.rebuildValue = d->peerFrame;
			sr.fold.rebuildCounter = d->dPeerCounters + kFoldRebuild * mv::kPeerMaxChains + chain;
			mv::launchPeerRebuildLight(streamB(ctx), sr);
			MV_HIP(ctx, hipGetLastError());
			mv::PeerSync b = peerSyncBase(ctx, ctx->st.frameTag, ctx->slot);
			for (uint32_t j = 1; j < C; ++j) if (d->peerFrame > j) { b.waitRow[b.nWait] = kStageRebuilt * mv::kPeerMaxChains + (uint32_t)((d->peerFrame - j) % C); b.waitValue[b.nWait++] = d->peerFrame - j; }
			b.waitLocal = 1u;
			if (b.nWait) { mv::launchPeerSync(streamB(ctx), b); MV_HIP(ctx, hipGetLastError()); }
		}
		else
			for (uint32_t j = 0; j < C; ++j) if (d->peerFrame > j) { w.stage[w.n] = kStageLight; w.frame[w.n++] = d->peerFrame - j; }
	}
	const int rc = peerWaitFor(ctx, streamB(ctx), ctx->st, kFoldView, w, d->peerFrame, ctx->st.frameTag, ctx->slot, !(plain && !d->peerFold));
	if (rc != MV_OK) return rc;
	if (d->peerFold && plain) ctx->st.fold.waitLocal = 2u;	// (row 1 of the march's waits is this rank's own)
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
		if (plain)
		{
			ctx->st.fold.rebuildBlocks = d->peerRebuildBlocks;
			ctx->st.fold.rebuildRow = kStageRebuilt * mv::kPeerMaxChains + chain; ctx->st.fold.rebuildValue = d->peerFrame;
			ctx->st.fold.rebuildCounter = d->dPeerCounters + kFoldRebuild * mv::kPeerMaxChains + chain;
		}
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

int peerCoherenceTest(mv_ctx* ctx, bool acquire, uint32_t* stale, uint32_t* firstReadBad);

// Self-test of the mappings and of the flags, twice with different patterns (a stale cached line would show the first pattern in the
// second round): every rank writes its pattern into its region of every rank's frame buffer (output set 0), raises a flag, waits for
// everybody's, verifies all regions of its own buffer, and only then -- a second flag -- lets anybody overwrite them.
int peerSelfTest(mv_ctx* ctx)
{
	DistState* d = ctx->dist;
	MV_REQUIRE(ctx, d->peerReady, MV_ERR_NOT_READY, "mv_dist_init_peer has not been called");
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	hipStream_t s = ctx->streamB[0];
	const size_t words = std::min<size_t>(65536, (size_t)ctx->st.W * compositeRows(ctx->st) / world);
	MV_REQUIRE(ctx, words >= 1, MV_ERR_RANGE, "self-test region");
	MV_HIP(ctx, hipMemsetAsync(d->dPeerMismatch, 0, 4, s));
	const uint32_t row = mv::kPeerStages * mv::kPeerMaxChains;	// the self-test's own row
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
	// ... and what the FOLDED waits rest on: other ranks' stores seen by plain loads behind a wait INSIDE a running kernel, of lines that
	// kernel has read before (the self-test above only crosses kernel boundaries)
	if (d->peerFold)
	{
		uint32_t stale = 0, first = 0;
		const int rc = peerCoherenceTest(ctx, d->peerAcquire, &stale, &first);
		if (rc != MV_OK) return rc;
		MV_REQUIRE(ctx, first == 0, MV_ERR_HIP, "peer-store self-test: the in-kernel test's first read did not find the pattern the kernel-boundary test had verified");
		if (stale)
		{
			char msg[256];
			snprintf(msg, sizeof(msg), "peer-store self-test: %u words were STALE behind a wait inside a running kernel (folded waits%s): use MV_DIST_PEER_NO_FOLD (waits as kernels of their own)",
				stale, d->peerAcquire ? ", acquire on" : ", MV_DIST_PEER_NO_ACQUIRE");
			return fail(ctx, MV_ERR_HIP, msg);
		}
	}
	return MV_OK;
}

// The in-kernel coherence test (mv_peer.hip peerCoherenceKernel; collective: every rank calls it): behind the pattern rounds above, whose
// second pattern the frame buffers of output set 0 still hold.  acquire: end the in-kernel wait as the folded waits do.
int peerCoherenceTest(mv_ctx* ctx, bool acquire, uint32_t* stale, uint32_t* firstReadBad)
{
	DistState* d = ctx->dist;
	const uint32_t world = ctx->st.shardCount, rank = ctx->st.shardIndex;
	*stale = 0; *firstReadBad = 0;
	if (d->peerLoopback || world <= 1u) return MV_OK;	// (a rank alone has nobody to be incoherent with)
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	hipStream_t s = ctx->streamB[0];
	const size_t words = std::min<size_t>(65536, (size_t)ctx->st.W * compositeRows(ctx->st) / world);
	const uint32_t row = mv::kPeerStages * mv::kPeerMaxChains;
	// (everybody is past whatever it did to its buffer before -- the clearing at the end of the test in front of this one)
	{
		const unsigned long long b0 = 2ull * (d->peerSelfTests++);
		mv::PeerSync a = peerSyncBase(ctx, 0u, 0u);
		a.nSignal = 1; a.signalRow[0] = row; a.signalValue[0] = b0 + 1u; a.nWait = 1; a.waitRow[0] = row; a.waitValue[0] = b0 + 1u;
		mv::launchPeerSync(s, a);
	}
	// the old pattern, through the kernel-boundary path the rounds of peerSelfTest have verified
	{
		const unsigned long long base = 2ull * (d->peerSelfTests++);
		mv::launchPeerPattern(s, d->dPeerComposite[0], world, rank, words, 31u);
		mv::PeerSync a = peerSyncBase(ctx, 0u, 0u);
		a.nSignal = 1; a.signalRow[0] = row; a.signalValue[0] = base + 1u; a.nWait = 1; a.waitRow[0] = row; a.waitValue[0] = base + 1u;
		mv::launchPeerSync(s, a);
	}
	MV_HIP(ctx, hipMemsetAsync(d->dPeerMismatch, 0, 64, s));
	mv::PeerCoherence c;
	memset(&c, 0, sizeof(c));
	const unsigned long long base = 2ull * (d->peerSelfTests++);
	c.table = d->dPeerComposite[0]; c.mine = ctx->ob[0].composite; c.peerFlags = d->dPeerFlags; c.myFlags = d->peerFlags;
	c.world = world; c.self = rank; c.row = row; c.acquire = acquire ? 1u : 0u; c.wordsPerRank = words; c.saltOld = 31u; c.saltNew = 32u;
	c.valueRead = base + 1u; c.valueWritten = base + 2u;
	c.counters = d->dPeerMismatch + 4; c.mismatches = d->dPeerMismatch;
	c.timeoutTicks = d->peerTimeoutTicks; c.report = d->reports;
	mv::launchPeerCoherence(s, c);
	MV_HIP(ctx, hipGetLastError());
	uint32_t out[2] = { 0, 0 };
	MV_HIP(ctx, hipMemcpyAsync(out, d->dPeerMismatch, 8, hipMemcpyDeviceToHost, s));
	// nobody may overwrite the regions before everybody has verified them: one more round of the flag
	{
		const unsigned long long b2 = 2ull * (d->peerSelfTests++);
		mv::PeerSync a = peerSyncBase(ctx, 0u, 0u);
		a.nSignal = 1; a.signalRow[0] = row; a.signalValue[0] = b2 + 1u; a.nWait = 1; a.waitRow[0] = row; a.waitValue[0] = b2 + 1u;
		mv::launchPeerSync(s, a);
	}
	MV_HIP(ctx, hipMemsetAsync(ctx->ob[0].composite, 0, (size_t)ctx->st.W * compositeRows(ctx->st) * sizeof(uint2), s));
	MV_HIP(ctx, hipStreamSynchronize(s));
	{ const int rc = peerCheckReports(ctx); if (rc != MV_OK) return rc; }
	*stale = out[0]; *firstReadBad = out[1];
	d->peerStaleWords = out[0]; d->peerFirstReadBad = out[1];
	return MV_OK;
}

extern "C" {

uint32_t mv_dealt_tiles(uint32_t tiles_x, uint32_t shard_index, uint32_t shard_count, uint32_t volume, uint32_t* tiles, uint32_t capacity)
{
	if (!tiles_x || !shard_count || shard_index >= shard_count) return 0u;
	const uint32_t n = mv::dealtTilesOfFace(tiles_x * tiles_x, tiles_x, shard_index, shard_count, volume);
	for (uint32_t k = 0; k < n && k < capacity && tiles; ++k) tiles[k] = mv::dealtTile(k, tiles_x, shard_index, shard_count, volume);
	return n;
}

int mv_light_rebuild_range(uint32_t light_grid_size, uint32_t shard_index, uint32_t shard_count, uint32_t out[5])
{
	if (!out || light_grid_size < 2u || light_grid_size > 640u || !shard_count || shard_count > mv::kPeerMaxRanks || shard_index >= shard_count) return MV_ERR_INVALID_ARG;
	const mv::LightRebuildRange r = mv::lightRebuildRange(light_grid_size, shard_index, shard_count);
	out[0] = r.z0; out[1] = r.z1; out[2] = r.before; out[3] = r.skip; out[4] = r.total;
	return MV_OK;
}

int mv_dist_peer_coherence(mv_ctx* ctx, uint32_t acquire, uint32_t* stale_words, uint32_t* first_read_bad)
{
	if (!ctx || !stale_words || !first_read_bad) return MV_ERR_INVALID_ARG;
	MV_REQUIRE(ctx, ctx->dist && ctx->dist->transport == 3 && ctx->dist->peerReady, MV_ERR_NOT_READY, "mv_dist_init_peer has not been called");
	return peerCoherenceTest(ctx, acquire != 0u, stale_words, first_read_bad);
}

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
	{
		// the staging maps of the plain light push: one per frame slot (a slot is reused numSlots frames later: further apart than the light
		// buffers, whose "done" flags already hold a rank back from writing into what another still reads)
		uint8_t* raw = nullptr;
		const int rc = distAlloc(ctx, &raw, (size_t)ctx->numSlots * ctx->st.L * ctx->st.L * ctx->st.L * sizeof(uint32_t), ctx->stream);
		if (rc != MV_OK) return bail(rc);
		d->lightPlain = reinterpret_cast<uint32_t*>(raw);
	}
	PeerBlob b;
	memset(&b, 0, sizeof(b));
	b.magic = kPeerMagic; b.version = MV_ABI_VERSION; b.rank = ctx->st.shardIndex; b.world = ctx->st.shardCount;
	b.pid = (uint64_t)getpid(); b.nonce = processNonce(); b.frame = ctx->frame; b.device = (uint32_t)ctx->cfg.device; b.numOutSets = ctx->numOutSets; b.lightBufs = ctx->st.lightBufs; b.flagsKind = (uint32_t)d->peerFlagsKind;
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
			// (every rank's buffers are addressed with this rank's offsets, and the pushes store 16 bytes at a time: the same alignment everywhere)
			if ((b.pointer[i] & 15u) != (reinterpret_cast<uint64_t>(mine[i]) & 15u)) return bail(fail(ctx, MV_ERR_INVALID_ARG, "a peer's buffer is aligned differently from this rank's"));
			if (b.pid == (uint64_t)getpid() && b.nonce == processNonce()) { table[(size_t)i * world + r] = reinterpret_cast<void*>(b.pointer[i]); continue; }	// a context of this very process (pid AND nonce: pids repeat across containers)
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
		d->dPeerLightPlain = reinterpret_cast<uint32_t**>(dev + (size_t)(kPeerBuffers - 1u) * world);
		if ((rc = distAlloc(ctx, &raw, 64, ctx->stream)) != MV_OK) return bail(rc);
		d->dPeerMismatch = reinterpret_cast<uint32_t*>(raw);
		if ((rc = distAlloc(ctx, &raw, 4u * mv::kPeerMaxChains * sizeof(uint32_t), ctx->stream)) != MV_OK) return bail(rc);
		d->dPeerCounters = reinterpret_cast<uint32_t*>(raw);
	}
	d->peerLoopback = loopback;
	// Folded flags (raised / waited for inside the marches and the composite: three kernels of a rank's chain instead of eight) are for
	// ranks that own their GPU: a consumer's workgroups then spin on flags only OTHER GPUs raise -- or, the view march's own row, workgroups
	// of the same launch dispatched in front of them (peerBeforeView).  Ranks that share a GPU (tests) keep the
	// flag kernels -- one wavefront each -- so that spinning consumers can never crowd the producers they wait for out of the GPU.
	d->peerFold = (flags & MV_DIST_PEER_FOLD) ? true : (flags & MV_DIST_PEER_NO_FOLD) ? false : (loopback || distinctDevices);
	if (ctx->framesPerLaunch > 1u)
	{
		// (several frames per launch exist in the folded form only: the launches that hold the frames carry their waits and flags)
		if (flags & MV_DIST_PEER_NO_FOLD) return bail(fail(ctx, MV_ERR_INVALID_ARG, "MV_DIST_PEER_NO_FOLD with mv_config.frames_per_launch > 1"));
		d->peerFold = true;
	}
	d->peerAcquire = !(flags & MV_DIST_PEER_NO_ACQUIRE);
	d->peerPlainLight = !(flags & MV_DIST_PEER_PUSH_CELLS);
	// (plain light push) the rebuild workgroups of a launch spin until every rank's texels have landed: many on a GPU of their own (0.77 M
	// cells at L = 96: twelve iterations of four per thread; 64: 0.0685, 128: 0.0658, 256: 0.0664 ms per frame of a rank of eight), few where ranks SHARE a GPU (tests), which must leave room for the workgroups they wait for
	d->peerRebuildBlocks = (loopback || distinctDevices) ? MV_PEER_REBUILD_BLOCKS : 32u;
	d->peerRowsRoot = (flags & MV_DIST_PEER_ROWS_TO_ROOT) != 0;
	d->peerDealVolumes = (flags & MV_DIST_PEER_DEAL_VOLUMES) ? true : (flags & MV_DIST_PEER_DEAL_TILES) ? false : ctx->framesPerLaunch == 1u;	// (see the header: tiles pay with several frames per launch)
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

// ---------------------------------------------------------------------------------------------
// Several frames per launch (mv_config.frames_per_launch = B > 1; peer stores, folded flags, MV_FLAG_CULL_ON_VIEW_STREAM).
// A GROUP of up to B consecutive frames shares a view stream (group J on stream J % C) and one value of every flag row (J): what was
// "frame k" to the flags is "group J" here.  A call of mv_dist_frame puts the frame's constants into their pinned slice (the FIRST call
// of a group also enqueues the tail of the group before it: batchTail); the LAST call, a flush or a setter sends the group out
// (batchClose): one upload, one cull launch (frame i + 1's workgroups wait for frame i's inside it), one orders, light-march and
// view-march launch for all its frames (mv_kernels.hpp DeviceStateBatch).  The tail: one kernel for the group's copy-forwards (one after
// the other inside it), one composite launch, one row push.  Ordering is the per-slot events' of the one-frame path (every frame
// records its own), stream order inside a group, and the flags between ranks.
// ---------------------------------------------------------------------------------------------
static_assert(sizeof(mv::DeviceStateBatch) + 64 <= 4096, "the frames' states of a launch travel by value: a kernel's argument segment holds 4 KB");
#ifndef MV_BATCH_ONE_WAIT
#define MV_BATCH_ONE_WAIT 0	// 1: one wait kernel per group (in front of the light march: the done rows AND the older groups' light rows) instead of two -- measured slower (profiles/r04_frames_per_launch.txt)
#endif
bool batchMode(const mv_ctx* ctx) { return ctx->dist && ctx->dist->transport == 3 && ctx->framesPerLaunch > 1u; }

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
int batchTail(mv_ctx* ctx)
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
		// The composite of the group's frames: it stores every pixel it writes into the other ranks' frame buffers itself (rank 0's only:
		// MV_DIST_PEER_ROWS_TO_ROOT) -- round 4 had a push kernel behind it for that (15 us alone, 60-190 us beside the other streams'
		// marches: profiles/r05_rank_chain.txt); the flags follow from a kernel of one wavefront
		for (uint32_t i = 0; i < nComposite; ++i)
		{
			foldBase(ctx, sts[i], kFoldComposite, P, last.st.frameTag, last.slot);
			if (peers)
			{
				sts[i].fold.stores = 1u; sts[i].fold.rowsRoot = d->peerRowsRoot ? 1u : 0u;
			}
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
		{
			// (the flags behind the composite's launches: a kernel boundary is what says that every store of theirs has been acknowledged)
			// (no composite in this group, or this rank has no row of the screen: "done" by a kernel of one wavefront)
			mv::PeerSync a = peerSyncBase(ctx, last.st.frameTag, last.slot);
			a.nSignal = 1; a.signalRow[0] = kStageDone * mv::kPeerMaxChains + chain; a.signalValue[0] = P;
			if (nComposite) { a.nSignal = 2; a.signalRow[1] = kStageRows * mv::kPeerMaxChains + chain; a.signalValue[1] = P; d->rowsWaitFrame[chain] = P; d->rowsStream[chain] = chain; }
			mv::launchPeerSync(X, a);
		}
		MV_HIP(ctx, hipGetLastError());
	}
	// everything that reads these frames' buffers has been enqueued: the next users of their slots / output sets / light buffers wait for this
	for (uint32_t i = 0; i < n; ++i) { MV_HIP(ctx, evRecord(ctx, ctx->evMainDone[d->pend[i].slot], X)); MV_HIP(ctx, latencyMark(ctx, d->pend[i].slot, X)); }
	d->nPend = 0;
	return MV_OK;
}

// light march + view march of the group being collected, on its stream
int batchClose(mv_ctx* ctx)
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
		if (peers) d->bytes += (size_t)n * last.st.L * last.st.L * last.st.L * (last.st.lightPlain ? sizeof(uint32_t) : sizeof(uint4)) / last.st.shardCount * (last.st.shardCount - 1u);
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
				foldWait(ctx, sts[i], kStageLight, J);	// (row 0 = this group's own: waited for by the tiles that read a map relit in it -- plain light push: by the rebuild workgroups, and the tiles wait for THEIR row, 1; the older groups': the kernel below)
				if (sts[i].lightPlain)
				{
					foldWait(ctx, sts[i], kStageRebuilt, J); sts[i].fold.waitLocal = 2u;
					sts[i].fold.rebuildBlocks = d->peerRebuildBlocks;
					sts[i].fold.rebuildRow = kStageRebuilt * mv::kPeerMaxChains + chain; sts[i].fold.rebuildValue = J;
					sts[i].fold.rebuildCounter = d->dPeerCounters + kFoldRebuild * mv::kPeerMaxChains + chain;
				}
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
			const bool plain = last.st.lightPlain != nullptr;	// (then: this rank's rebuilt cells of those maps -- local rows)
			for (uint32_t c = 1; c < C; ++c) if (J > c) { a.waitRow[a.nWait] = (plain ? kStageRebuilt : kStageLight) * mv::kPeerMaxChains + (uint32_t)((J - c) % C); a.waitValue[a.nWait++] = J - c; }
			a.waitLocal = plain ? 1u : 0u;
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
			for (uint32_t i = 0; i < n; ++i) { sts[i].fold.nWait = 0u; sts[i].fold.pushBlocks = 0u; sts[i].fold.rebuildBlocks = 0u; }
			mv::launchMarchTilesOverflowBatch(S, sts, n, beyond);
		}
		endStage(ctx, evt, S);
		MV_HIP(ctx, hipGetLastError());
	}
	for (uint32_t i = 0; i < n; ++i) d->pend[i] = d->cur[i];
	d->nPend = n; d->pendBatch = J; d->nCur = 0;
	return MV_OK;
}

int distFrameBatch(mv_ctx* ctx, const void* cbPerFrame, const void* perObject, int with_composite)
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

