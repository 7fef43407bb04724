// mv_kernels.hpp -- launch interface between the C-ABI host layer (mv_api.cpp) and the gfx950 kernels (mv_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace mv
{
// Light-map buffers per volume: DeviceState.lightBufs (3, or as many as there are frames in flight).  The light march of frame i writes the
// buffer the relit volume's table entry held lightBufs relights ago, which no frame after i - lightBufs can still be reading.

// One ray of the view march, ready to be marched: the record the ray-setup stage emits and the march stage consumes -- the
// counterpart of the work-graph variant's RayMarchRecord (Sh/LibRayMarch.hlsl:39-134), at ray instead of tile granularity.
struct RayRecord
{
	float ox, oy, oz;		// origin in volume space, after ComputeRayOrigin (CSRayMarch.hlsl:96)
	float dx, dy, dz;		// unit direction
	float tMax;				// min(ComputeTargetHit, GetTMax) for a cube-face texel, GetTMax for a RayCast pixel
	uint32_t out;			// element index of the result: cubeMaps[out], or screenLayers[out & 0x7fffffff] when bit 31 is set; kNoRay = slot without a ray
};
static_assert(sizeof(RayRecord) == 32, "two 16-byte stores / loads per lane");
constexpr uint32_t kNoRay = 0xffffffffu, kScreenRayBit = 0x80000000u;
// The ray queue is dealt over kRayQueues heads (dispatch slot s belongs to queue s % kRayQueues): one head word serves about 88
// returning atomics per microsecond chip-wide, which 5000 wavefronts pulling 10^4 tiles would spend a third of the launch on.
constexpr uint32_t kRayQueues = 64, kRayQueueStride = 32;	// stride in uint32: every head on a 128-byte line of its own

// what the 64 records of one dispatch slot (one 8x8 tile) have in common: the tile's key in the cost table and a copy of its
// volume's constants, so that a lane taking a record needs ONE round trip (record + tile info are independent loads)
struct TileInfo
{
	uint64_t texelOff;		// byte offset of the volume's bricked texels inside volPool
	uint32_t lightOff;		// byte offset of the light map this frame reads inside lightBuffers
	uint32_t occOff;		// index of the volume's first occupancy word inside occPool
	uint32_t smpCount;		// VolumeInfo.SmpCount
	float stepScale;		// g_maxDist / SmpCount (CSRayMarch.hlsl:109)
	uint32_t costKey;
	uint32_t volume;
};
static_assert(sizeof(TileInfo) == 32, "two 16-byte loads");

// per-volume constants of the march, written by the cull every frame (everything a ray needs besides its record)
struct VolumeConsts
{
	uint64_t texelOff;		// byte offset of the volume's bricked texels inside volPool
	uint32_t lightOff;		// byte offset of the light map this frame reads inside lightBuffers
	uint32_t occOff;		// index of the volume's first occupancy word inside occPool
	uint32_t smpCount;		// VolumeInfo.SmpCount
	float stepScale;		// g_maxDist / SmpCount (CSRayMarch.hlsl:109)
	uint32_t pad[2];
};
static_assert(sizeof(VolumeConsts) == 32, "two 16-byte loads");

// Device-side state of one context. Every pointer is device memory; the struct is passed to kernels by value.
struct DeviceState
{
	// geometry of the problem
	uint32_t N;				// volume / cube-face edge
	uint32_t bricks;		// ceil(N / 4): volumes are stored as 4x4x4 bricks
	uint32_t L;				// light-map edge
	uint32_t V;				// number of volumes
	uint32_t S;				// number of volume sources
	uint32_t W, H;			// viewport
	uint32_t M;				// shadow-map edge
	uint32_t flags;
	uint32_t shardIndex, shardCount;
	uint32_t shardMagic, rowsPerRank;	// ceil(2^32 / shardCount): t / shardCount == umulhi(t, shardMagic) for the tile rows and ranks in question (t < 2^16, shardCount <= 2^10); ceil(tile rows / shardCount)
	uint32_t bandRow0, bandRow1;	// screen rows [bandRow0, bandRow1) this shard composites (8-row aligned; the whole screen when unsharded):
									// the screen tiles of the fused RayCast are marched for these rows only, for EVERY screen-space-scheme volume
	uint32_t maxRaySamples, maxLightSamples;
	uint32_t hasLightProbe;
	uint32_t lightBufs;				// light-map buffers per volume (see above)
	float fN, fNm1, fL, fLp1;		// (float)N, N - 1, (float)L, L + 1: the samplers' constants, as kernel arguments so that they sit in scalar registers
									// (a float the kernel converts from a scalar integer lives in a vector register through its loops)

	// per-frame constants (device copies of the reference's upload-heap slices)
	const void* cbPerFrame;			// CBPerFrame, 212 B
	const void* perObject;			// PerObject[V]
	const uint32_t* volumeDescs;	// VolumeDesc[V]
	const float* sh;				// 27 floats

	// resources
	const uint8_t* volPool;			// all S bricked volumes in ONE allocation, volStride bytes apart (volumes[i] = volPool + i volStride):
	uint64_t volStride;				//   the march addresses texels as pool base (SGPR) + 32-bit offset whatever the ray's volume
	const uint64_t* occPool;		// all S occupancy arrays, occStride words apart
	uint32_t occStride;
	const uint2* const* volumes;	// S pointers to bricked RGBA16F texels
	const uint4* const* alphaCells;	// S pointers to the bricked alpha cells (8 halfs per texel: its 2x2x2 footprint)
	const uint64_t* const* occupancy;	// S pointers to the brick occupancy words (mv_device.hpp), ceil(bricks / 4)^3 each
	// Light maps: L^3 quad-replicated R11G11B10F cells {t(x,y), t(x+1,y), t(x,y+1), t(x+1,y+1)} per volume.  Every volume
	// has lightBufs buffers; lightTable[v] is the one this frame reads (and, for the volume relit this frame, writes).
	// The cull derives it from the previous frame's table and rotates the relit volume's entry, so that the light march
	// of frames i+1, i+2 can run while the view march / composite of frame i still read frame i's light maps.
	uint4* lightBuffers;			// lightBufs V x L^3 cells: buffer k of volume v at index lightBufs v + k
	uint4** lightTable;				// V pointers (this frame)
	uint4* const* lightTablePrev;	// V pointers (previous frame)
	const uint16_t* volumeInfoPrev;	// previous frame's VolumeInfo (rows of culled volumes are carried over: the reference never clears them)
	const float* depthMap;			// W x H
	const uint16_t* shadowMap;		// M x M D16
	uint2* cubeMaps;				// V x chain RGBA16F texels
	float* cubeDepths;				// V x chain
	uint32_t* faceStamps;			// V x 5 x 6: frame that last wrote (volume, mip, face) into THIS output set (mv_composite.hip copyForwardKernel)
	uint2* composite;				// W x H RGBA16F
	float4* screenLayers;			// screenSlots x W x H fp32 RGBA: RayCast results of screen-space-scheme volumes (marched inside marchViewKernel)
	uint32_t* screenSlot;			// V: slot of volume v in screenLayers, or 0xffffffff (composite then marches it inline)
	uint32_t screenSlots;			// capacity

	// cull outputs
	uint16_t* volumeInfo;			// V x 4
	uint32_t* visibleVolumes;		// V
	uint32_t* cubeMapVolumes;		// V
	unsigned long long* tilesReport;	// pinned host memory (may be NULL): the cull stores {frameTag : tile count} here -- a HINT for the host's
									// next grid sizes, read without ever waiting for it
	unsigned long long* tilesReportSlot;	// the same word once more, in this frame slot's own place (several frames per launch: every frame's count, not just the newest)
	unsigned long long* exchangeReport;	// pinned host memory (NULL outside the multi-GPU frame loop): {frameTag : texels of the largest shard of this frame's cube-face exchange}
	uint32_t frameTag;				// low 32 bits of the context's frame counter
	const uint32_t* countersPrev;	// the previous frame's counters (its [10]: the hash the cull compares this frame's with)
	uint32_t* batchInfo;			// 8 words per frame slot (NULL outside the multi-GPU loop's peer stores): what the ranks compare of their culls, the volumes
	const uint32_t* batchInfoPrev;	//   relit by the frames of a launch so far (cullKernel's epilogue); the previous frame's
	uint32_t batchPos, batchPad;	// position of this frame among the frames of its launch (mv_config.frames_per_launch; 0 when there is one)
	uint32_t* counters;				// [0] visible count, [1] cube-map count, [2] work entries, [3] total tiles,
									// [4] owned visible count, [5] min selected mip over visible volumes, [6] cube tiles (screen tiles follow),
									// [8] the cull's ticket, [9] head of the march's ray queue (device-only), [10] hash of the frame's camera and volume
									// transforms, [11] that hash differs from the one of the slot's previous frame (whose march wrote the tile costs)
	VolumeConsts* volConsts;		// V (written by the cull)
	uint32_t* queueHeads;			// kRayQueues heads of the march's ray queue, one per 128-byte line (zeroed by the ray setup)
	uint32_t* ownedVisible;			// V: visible volumes of this shard (v % shardCount), ascending
	uint32_t* exchangeOff;			// V: texel offset of a volume's marched faces inside its owner's exchange shard (written by the cull)
	uint32_t* workEntries;			// 6 V + screenSlots: volume | face << 16 (face 7 = screen-space RayCast entry)
	uint32_t* workTileStart;		// entries + 1 prefix sums of 8x8 tiles
	uint64_t* cubeOffset;			// V: texel offset of the selected mip of volume v inside cubeMaps / cubeDepths
	uint16_t* volumeRects;			// V x 4: conservative screen rectangle {x0, y0, x1, y1} (pixels, inclusive) of a visible volume
	float4* lightDirs;				// V: normalize(mul(lightPos, (float3x3)WorldI_v)) -- CSRayMarchL.hlsl:88-89 evaluates it per voxel, it only depends on v
	uint32_t* volumeFlags;			// V: bit0 visible this frame, bit1 cube-map scheme, bits 2-4 mip, bits 5-10 face mask

	// Counters.  Every wavefront owns one slot and updates it with a plain read-modify-write (launches are stream
	// ordered), because tens of thousands of same-address atomics per launch cost more than the march itself.
	uint64_t* stats;				// totals, filled by launchReduceStats: [0] view samples, [1] lit samples, [2] rays,
									// [3] light samples, [4] composite samples, [5] marchV lane slots, [6] fused RayCast lit samples, [7] fetch-free samples
	uint64_t* timeline;				// marchWaves x 8 (filled by -DMV_TIMELINE builds only): per-wavefront stamps of the last march
	uint64_t* waveStats;			// marchWaves x 8: {view samples, lit, rays, lane slots, RayCast samples, RayCast lit, fetch-free samples, -} per march wavefront
	uint64_t* lightStats;			// ceil(L/4)^3: light samples per marchL block
	uint32_t* lightOrder;			// ceil(L/4)^3 (+ 4: the light direction it was made for): dispatch order of the light march's voxel blocks (lightOrderKernel), per frame slot
	uint64_t* compositeStats;		// compositeSlots: RayCast samples per composite wavefront
	uint32_t* compositeRetry;		// 2 + compositeSlots: {count, finished workgroups, tiles the composite's second launch has to march inline}
	uint32_t maxTiles, lightBlocks, compositeSlots, maxScreenTiles, marchWaves;
	const uint32_t* tileOrder;		// dispatch order of the tile list, longest bins first (launchOrders): bin b's tiles at [b * orderCap, b * orderCap + binCursor[b]); NULL = list order
	uint32_t* binCursor;			// kOrderBins: tiles in every bin (zeroed by the cull, counted up by the order kernel)
	uint32_t orderCap;				// entries between the bins' regions of tileOrder
	uint32_t* tileCost;				// V x 6 x ceil(N/8)^2 + maxScreenTiles + 1: samples of a tile's longest ray in the last launch that marched it, by stable tile key (one table for all frame slots)
	// ray queue of the view march (per output set: two frames may be between setup and march at the same time)
	RayRecord* rayRecords;			// (maxTiles + maxScreenTiles) x 64, in dispatch order: slot s holds the rays of tile tileOrder[s]
	TileInfo* tileInfo;				// maxTiles + maxScreenTiles
	// Peer-store transport of the multi-GPU frame loop (mv_dist_init_peer; mv_peer.hip): every rank maps every rank's light maps, cube
	// maps / depths and frame buffers (xGMI peer mappings: IPC handles between processes), and a push kernel behind each producer
	// copies what it produced into all of them -- no pack, no collective, no unpack.  peerCount = 0 outside that transport; else
	// shardCount, and entry [shardIndex] of every table is this context's own buffer.
	uint32_t peerCount;
	uint4* const* peerLight;		// [peerCount] every rank's lightBuffers (same layout everywhere: an address here + the offset of a map inside lightBuffers)
	uint2* const* peerCubeMaps;		// [peerCount] every rank's cube maps of THIS frame's output set
	float* const* peerCubeDepths;	// [peerCount]
	uint2* const* peerComposite;	// [peerCount] every rank's frame buffer of this output set
	// (plain light push, the default) a frame's relit map travels as PLAIN R11G11B10F texels -- 4 bytes each instead of the 16 of a
	// quad-replicated cell: a rank's share 0.44 MB instead of 1.77 MB per peer at L = 96 -- into a staging map per frame slot on every rank,
	// and the receiver rebuilds the cells of the other ranks' layers itself (peerRebuildLightRole).  NULL: whole cells are pushed (round 4).
	uint32_t* lightPlain;			// this rank's staging map of this frame's slot: L^3 words, [z][y][x]
	uint32_t* const* peerLightPlain;	// [peerCount] every rank's staging maps (all slots); this slot's starts lightPlainOff words in
	uint64_t lightPlainOff;
	// ...and the flags that order those stores (mv_peer.hip explains the protocol): a push kernel RAISES, from the last workgroup to
	// finish, its column of rows in every rank's block; a consumer (light march, view march, composite) may WAIT, in every workgroup's
	// prologue, until rows of this rank's flag block have reached a frame in every column ("folded" waits: ranks that own their GPU;
	// ranks that share one wait in a one-wavefront kernel of its own, peerSyncKernel).  nWait = nSignal = 0: neither.
	struct PeerFold
	{
		unsigned long long* const* flagBlocks;	// [peerCount] every rank's flag block
		unsigned long long* myFlags;
		uint32_t* counter;						// workgroups of this launch that have finished (zero between launches)
		unsigned long long* report;				// pinned: [2] = a wait that gave up
		unsigned long long timeoutTicks;
		uint32_t nWait, nSignal, loopback, frameTag;
		uint32_t checkChain, checkCompare;		// checkChain: 0xffffffff, or the chain whose slot of cull words this launch raises with (a producer) or, checkCompare != 0, compares every rank's behind its wait (a consumer) -- mv_peer.hip
		// (view march, folded) the leading pushBlocks workgroups of the launch do not march: they push this rank's light layers (peerPushLightRole)
		// and the last of them raises pushRow to pushValue in every rank's block, with this rank's cull words (chain pushCheckChain)
		uint32_t pushBlocks, pushRow, pushCheckChain;
		uint32_t acquire;						// a wait that saw its flags ends with a system-scope acquire (buffer_inv sc0 sc1) by the polling wavefront: peerFoldAcquire
		unsigned long long pushValue;
		uint32_t* pushCounter;
		const uint32_t* checkInfo;				// NULL, or the batchInfo of the launch's last frame: the cull words of all its frames (several frames per launch)
		uint32_t stores, rowsRoot;				// stores: (view march) this launch also stores its texels into the other ranks' cube maps; rowsRoot: (row push) to rank 0 only
		// (plain light push) behind the push workgroups, rebuildBlocks workgroups wait for every rank's texels of this frame (waitRow[0]),
		// rebuild the cells of the other ranks' layers and the last of them raises rebuildRow to rebuildValue in THIS rank's block, column
		// shardIndex: a local row (bit in waitLocal: only that column is waited for) -- what the tiles that read the relit map wait for
		uint32_t rebuildBlocks, rebuildRow, waitLocal, rebuildPad;
		unsigned long long rebuildValue;
		uint32_t* rebuildCounter;
		uint32_t waitRow[6], signalRow[2];
		unsigned long long waitValue[6], signalValue[2];
	} fold;
};
// Several frames in ONE launch (mv_config.frames_per_launch, the multi-GPU frame loop with peer stores): blockIdx.y picks the frame.  A
// rank of eight has an eighth of a frame's tiles, voxel blocks and rows -- a wavefront per SIMD or less, so that every long kernel of its
// chain lasts as long as its longest ray whatever else the GPU could do meanwhile, and four hardware queues are all the frames that can
// overlap.  B frames per launch put B wavefronts where one was: the chain gets little longer and carries B frames.
// The states travel by value in the kernel argument segment (784 bytes each: four fit under its 4 KB).
constexpr uint32_t kMaxBatch = 4;
struct DeviceStateBatch { DeviceState s[kMaxBatch]; };
// Stores into a peer's memory: written through at SYSTEM scope (global_store ... sc0 sc1: the acknowledgement comes from the memory that
// owns the address, nothing is left dirty in this GPU's L2), so that "complete" is the storing wavefront's s_waitcnt vmcnt(0) and no
// cache has to be written back in front of the flag that announces them.  16 bytes per store wherever a range allows it: narrower
// write-through stores are one fabric write each (dword ~6x, dwordx2 ~2.7x the time per byte of a dwordx4).
__device__ __forceinline__ void peerStore(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void peerStore8(uint2* p, uint2 v)
{
	__hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peerStore16(void* p, uint4 v)
{
	typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
	u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
	asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory");
}
// Every rank owns a block of flags that the OTHER ranks write and only it reads: row r, column p = "rank p has finished stage r.stage of
// frame <value>", rows = stage x chain (frames k with k % chains == chain follow each other on one stream, so a row only ever grows).
constexpr uint32_t kPeerMaxRanks = 64;		// ranks of one peer-store group (one node: 8); columns of a flag row
constexpr uint32_t kPeerStages = 5;			// 0 light share landed, 1 cube faces landed, 2 rows landed, 3 frame done (its buffers may be overwritten), 4 (local) light cells rebuilt from the landed texels
constexpr uint32_t kPeerMaxChains = 5;
constexpr uint32_t kPeerRows = kPeerStages * kPeerMaxChains + 1u;	// + the self-test's row
constexpr uint32_t kPeerCheckWords = 4;		// what a rank's cull decided for a frame (compared by the light stage's wait)
constexpr size_t kPeerFlagWords = (size_t)kPeerRows * kPeerMaxRanks;	// uint64 each; behind them kPeerMaxChains x kPeerMaxRanks x kPeerCheckWords uint32
constexpr size_t kPeerFlagBytes = kPeerFlagWords * 8u + (size_t)kPeerMaxChains * kPeerMaxRanks * kPeerCheckWords * 4u;

// Prologue of a consumer (every thread of the workgroup calls it, before any early exit): the first wavefront polls, the workgroup waits for it.
// The polls are relaxed system-scope loads (sc0 sc1: they bypass this GPU's caches).  What they announce was written through, by kernels of
// other GPUs, into THIS GPU's memory; the wavefront that saw the flags then makes the ACQUIRE the memory model asks for between another
// agent's release and this agent's plain loads -- peerFoldAcquire: `buffer_inv sc0 sc1`, invalidate only, no write-back -- and the
// workgroup's own loads of the data are issued behind the barrier below, i.e. behind that invalidate.  (Round 4 had no invalidate here and
// rested on "this launch has not read those addresses before, the buffers rotate"; that still holds and is why the invalidate finds
// little to drop -- but a stale line would have corrupted a frame silently, and the only evidence against one was a GPU with itself.
// What was measured three times a launch's length in round 4 were per-workgroup WRITE-BACKS (buffer_wbl2); the invalidate alone costs
// the launch of a rank profiles/r05_peer_acquire.txt's figure.  fold.acquire = 0, MV_DIST_PEER_NO_ACQUIRE, is round 4's form, for A/B.)
__device__ __forceinline__ void peerFoldAcquire(const DeviceState& st)
{
	if (st.fold.acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");	// system scope: s_waitcnt vmcnt(0); buffer_inv sc0 sc1
}
// (firstRow / endRow: the rows of fold.waitRow to wait for -- the view march waits for its newest row only in the workgroups that read
// the light map that row announces)
// A wait of this frame slot has given up (its report is set): the workgroups dispatched behind it, and the slot's later launches, do not each wait
// their own timeout for the same stalled rank; the host finds the report at its next call.  Looked at on the first unsuccessful poll and every
// 1024th after it (pinned host memory).
#ifndef MV_PEER_STICKY_GIVE_UP
#define MV_PEER_STICKY_GIVE_UP 1	// (0: every workgroup waits its own timeout -- tests/test_dist_frame_loop.py's stalled-rank test then fails: its teeth)
#endif
__device__ __forceinline__ bool peerGaveUp(const DeviceState& st)
{
	return MV_PEER_STICKY_GIVE_UP && __hip_atomic_load(st.fold.report + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
}
__device__ __forceinline__ void peerFoldWait(const DeviceState& st, uint32_t firstRow = 0u, uint32_t endRow = 0xffffffffu)
{
	if (endRow > st.fold.nWait) endRow = st.fold.nWait;
	if (firstRow >= endRow) return;
	if (threadIdx.x < 64u)
	{
		const uint32_t total = (endRow - firstRow) * st.peerCount;
		const unsigned long long t0 = wall_clock64();
		for (uint32_t i = threadIdx.x; i < total; i += 64u)
		{
			const uint32_t r = firstRow + i / st.peerCount, p = i - (r - firstRow) * st.peerCount;
			if (((st.fold.waitLocal >> r) & 1u) && p != st.shardIndex) continue;	// (a local row: this rank's own column is the only one ever raised)
			const unsigned long long* flag = st.fold.myFlags + (size_t)st.fold.waitRow[r] * kPeerMaxRanks + p;
			// (a local row is raised by a kernel of THIS rank, which may itself be waiting for a stalled rank: twice the patience, so that the
			// report names that rank and not this one)
			const unsigned long long patience = st.fold.timeoutTicks << ((st.fold.waitLocal >> r) & 1u);
			uint32_t polls = 0u;
			while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < st.fold.waitValue[r])
			{
				if (wall_clock64() - t0 > patience)
				{
					__hip_atomic_store(st.fold.report + 2, ((unsigned long long)st.fold.frameTag << 32) | ((unsigned long long)st.fold.waitRow[r] << 8) | p | 0x80000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					break;
				}
				if ((polls++ & 1023u) == 0u && peerGaveUp(st)) break;
				__builtin_amdgcn_s_sleep(8);
			}
		}
		peerFoldAcquire(st);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	// (the invalidate completes asynchronously: the barrier below must not open before it has)
	}
	__syncthreads();
	if (st.fold.checkCompare && st.fold.checkChain != 0xffffffffu && firstRow == 0u && blockIdx.x == 0u && threadIdx.x < 64u)	// (one workgroup compares, behind the wait for the row the words came with)
	{
		// every rank's cull must have decided the same for this frame (peerSyncKernel makes the same comparison)
		const uint32_t* all = reinterpret_cast<const uint32_t*>(st.fold.myFlags + kPeerFlagWords) + (size_t)st.fold.checkChain * kPeerMaxRanks * kPeerCheckWords;
		const uint32_t mine[4] = { st.fold.frameTag, st.counters[12], st.counters[0], st.counters[1] };
		bool differs = false;
		for (uint32_t p = threadIdx.x; p < st.peerCount; p += 64u)
			for (uint32_t w = 0; w < kPeerCheckWords; ++w)
				differs |= __hip_atomic_load(all + (size_t)p * kPeerCheckWords + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != mine[w];
		if (differs) __hip_atomic_store(st.fold.report + 1, ((unsigned long long)st.fold.frameTag << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// The same wait by ONE wavefront for itself (all its lanes call it; no barrier): the view march's wavefronts whose tile reads the light map
// relit in this frame wait for that frame's row -- and wavefront `compare` of the launch also compares the ranks' cull words behind it.
// What a rank's cull decided for the frame(s) of a launch, as the four words the ranks compare: the frame tag and the cull's words of
// the launch's last frame (checkInfo: cullKernel folds the earlier frames' in; one frame per launch: hash, visible, cube-map volumes).
// (Nothing here walks the other frames' states: they are a by-value kernel argument, which the compiler reads straight from the argument
// segment only while every access has one dynamic index at most -- otherwise each workgroup first copies 3 KB into scratch.)
__device__ __forceinline__ void peerCheckWordsOf(const DeviceState& st, uint32_t w[4])
{
	const uint32_t* info = st.fold.checkInfo;
	w[0] = st.fold.frameTag;
	if (info) { w[1] = info[0]; w[2] = info[1]; w[3] = info[2]; }
	else { w[1] = st.counters[12]; w[2] = st.counters[0]; w[3] = st.counters[1]; }
}
__device__ __forceinline__ void peerFoldWaitWave(const DeviceState& st, uint32_t firstRow, uint32_t endRow, bool compare)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t total = (endRow - firstRow) * st.peerCount;
	const unsigned long long t0 = wall_clock64();
	for (uint32_t i = lane; i < total; i += 64u)
	{
		const uint32_t r = firstRow + i / st.peerCount, p = i - (r - firstRow) * st.peerCount;
		if (((st.fold.waitLocal >> r) & 1u) && p != st.shardIndex) continue;
		const unsigned long long* flag = st.fold.myFlags + (size_t)st.fold.waitRow[r] * kPeerMaxRanks + p;
		const unsigned long long patience = st.fold.timeoutTicks << ((st.fold.waitLocal >> r) & 1u);	// (as in peerFoldWait)
		uint32_t polls = 0u;
		while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < st.fold.waitValue[r])
		{
			if (wall_clock64() - t0 > patience)
			{
				__hip_atomic_store(st.fold.report + 2, ((unsigned long long)st.fold.frameTag << 32) | ((unsigned long long)st.fold.waitRow[r] << 8) | p | 0x80000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				break;
			}
			if ((polls++ & 1023u) == 0u && peerGaveUp(st)) break;
			__builtin_amdgcn_s_sleep(8);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_wave_barrier();	// (the wavefront's loads of what the flags announce are issued behind every lane's poll)
	peerFoldAcquire(st);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (compare && st.fold.checkCompare && st.fold.checkChain != 0xffffffffu)
	{
		const uint32_t* all = reinterpret_cast<const uint32_t*>(st.fold.myFlags + kPeerFlagWords) + (size_t)st.fold.checkChain * kPeerMaxRanks * kPeerCheckWords;
		uint32_t mine[4];
		peerCheckWordsOf(st, mine);
		bool differs = false;
		for (uint32_t p = lane; p < st.peerCount; p += 64u)
			for (uint32_t w = 0; w < kPeerCheckWords; ++w)
				differs |= __hip_atomic_load(all + (size_t)p * kPeerCheckWords + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != mine[w];
		if (differs) __hip_atomic_store(st.fold.report + 1, ((unsigned long long)st.fold.frameTag << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

// Epilogue of a producer (every thread of the workgroup calls it, after a body that RETURNS instead of exiting): the last workgroup of
// the launch raises the flags, behind every workgroup's stores.  Again without cache maintenance: the stores into the other ranks'
// memory are written through at system scope (peerStore), so "complete" is each wavefront's s_waitcnt vmcnt(0) -- the acknowledgement
// of a write-through store comes from the memory that owns the address.  Every workgroup counts itself in AFTER that wait; the one
// that counts last knows every store of the launch has been acknowledged, and only then issues the flag stores.
// (localOnly: the rows are raised in THIS rank's block alone, column shardIndex -- what kernels of this rank wait for, e.g. "light cells rebuilt")
__device__ __forceinline__ void peerRaise(const DeviceState& st, uint32_t* counter, uint32_t expected, uint32_t nSignal, const uint32_t* rows, const unsigned long long* values, uint32_t checkChain, bool localOnly = false)
{
	__shared__ uint32_t sLast;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	// this wavefront's stores, into every rank's memory, have been acknowledged
	__syncthreads();
	if (threadIdx.x == 0u)
	{
		const uint32_t finished = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const bool last = finished == expected - 1u;
		if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);	// (the next launch that uses it is behind this one on its stream)
		sLast = last ? 1u : 0u;
	}
	__syncthreads();
	if (!sLast || threadIdx.x >= 64u) return;
	if (localOnly)
	{
		if (threadIdx.x == 0u)
			for (uint32_t i = 0; i < nSignal; ++i)
				__hip_atomic_store(st.fold.myFlags + (size_t)rows[i] * kPeerMaxRanks + st.shardIndex, values[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	// the last workgroup's first wavefront: lane p raises this rank's column in rank p's block (all ranks at once: a write-through
	// store and its acknowledgement are a trip over the fabric each)
	for (uint32_t p = threadIdx.x; p < st.peerCount; p += 64u)
	{
		unsigned long long* block = st.fold.loopback ? st.fold.myFlags : st.fold.flagBlocks[p];	// loopback: a rank measured alone raises every column of its own block
		const uint32_t column = st.fold.loopback ? p : st.shardIndex;
		if (checkChain != 0xffffffffu)
		{
			uint32_t* words = reinterpret_cast<uint32_t*>(block + kPeerFlagWords) + ((size_t)checkChain * kPeerMaxRanks + column) * kPeerCheckWords;
			uint32_t mine[4];
			peerCheckWordsOf(st, mine);
			peerStore(words + 0, mine[0]); peerStore(words + 1, mine[1]); peerStore(words + 2, mine[2]); peerStore(words + 3, mine[3]);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");	// the words are there before the flag that announces them
		}
		for (uint32_t i = 0; i < nSignal; ++i)
			__hip_atomic_store(block + (size_t)rows[i] * kPeerMaxRanks + column, values[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}
// (several frames per launch: the frames share the counter, the rows and the values -- the last workgroup of the WHOLE launch raises)
__device__ __forceinline__ void peerFoldSignal(const DeviceState& st)
{
	if (!st.fold.nSignal) return;
	peerRaise(st, st.fold.counter, gridDim.x * gridDim.y, st.fold.nSignal, st.fold.signalRow, st.fold.signalValue, st.fold.checkCompare ? 0xffffffffu : st.fold.checkChain);
}

constexpr uint32_t kFlagDealTiles = 0x04000000u;	// DeviceState.flags (internal, peer-store transport): the view march of EVERY cube-map volume is shared out by tiles -- rank r marches the groups g of
													// tiles of each marched face with (g + volume) % shardCount == r (dealtGroup below) -- instead of whole volumes dealt by estimated cost: ten visible volumes on eight ranks leave two ranks
													// with two volumes each (profiles/r04_frames_per_launch.txt); a march that stores its texels into every rank's cube maps itself does not care who owns what
// kFlagDealTiles: the k-th tile this rank marches of a face of volume v, and how many of a face's T tiles are its.  The unit of the deal is a
// GROUP of G consecutive tiles of the face, group g to rank (g + v) % shardCount: n <= MV_DEAL_ROWS whole rows of tiles, the largest n that
// leaves every rank the same number of groups -- a rank's wavefronts then march neighbouring tiles, whose rays share cache lines -- and
// single tiles (G = 1, round 4's deal: every shardCount-th tile) on faces that no such n divides.  A rank of eight at C3 (faces of 16 x 16
// tiles; n = 2 is a contiguous eighth): single tiles 0.0662 ms per frame, n = 1 0.0654, n = 2 0.0643; a rank of four: n = 2 0.0978, n = 4 (a
// contiguous quarter) 0.0988 -- two bands balance better than one; profiles/r05_rank_chain.txt.
#ifndef MV_DEAL_ROWS
#define MV_DEAL_ROWS 2
#endif
__host__ __device__ inline uint32_t dealtGroup(uint32_t tilesX, uint32_t shardCount)
{
	for (uint32_t n = MV_DEAL_ROWS; n >= 1u; --n)
		if (tilesX % n == 0u && (tilesX / n) % shardCount == 0u) return tilesX * n;
	return 1u;
}
__host__ __device__ inline uint32_t dealtTileFirst(uint32_t shardIndex, uint32_t shardCount, uint32_t v) { return (shardIndex + shardCount - v % shardCount) % shardCount; }	// (the rank's first group)
__host__ __device__ inline uint32_t dealtTilesOfFace(uint32_t T, uint32_t tilesX, uint32_t shardIndex, uint32_t shardCount, uint32_t v)
{
	const uint32_t G = dealtGroup(tilesX, shardCount), groups = (T + G - 1u) / G;
	const uint32_t first = dealtTileFirst(shardIndex, shardCount, v);
	if (first >= groups) return 0u;
	const uint32_t mine = (groups - first + shardCount - 1u) / shardCount;
	const bool ownsLast = (groups - 1u - first) % shardCount == 0u;
	return mine * G - (ownsLast ? groups * G - T : 0u);
}
__host__ __device__ inline uint32_t dealtTile(uint32_t k, uint32_t tilesX, uint32_t shardIndex, uint32_t shardCount, uint32_t v)	// the rank's k-th tile of a face
{
	const uint32_t G = dealtGroup(tilesX, shardCount);
	return (dealtTileFirst(shardIndex, shardCount, v) + (k / G) * shardCount) * G + k % G;
}
// Edge of mip `mip` of a cube map of edge N, as D3D defines a mip chain (and as the reference's shaders learn it: GetDimensions of the mip's view,
// CSRayMarch.hlsl:31, PSCube.hlsli:55): max(1, N >> mip).  Only differs from N >> mip for N < 16 (NumMips = 5), where round 1-5's `N >> mip` made the last
// mips EMPTY -- a composite that selected one divided by an edge of zero and read out of bounds (round 6: tests/test_random_scenes.py, a 9 x 2 viewport).
// (The cull's own `cubeMapSize >> mipLevel`, CSVolumeCull.hlsl:62, is the shader's arithmetic and stays.)
__host__ __device__ inline uint32_t cubeMipDim(uint32_t N, uint32_t mip) { const uint32_t d = N >> mip; return d ? d : 1u; }

__device__ __forceinline__ uint32_t cubeLocalTile(const DeviceState& st, uint32_t volumeId, uint32_t k)
{
	if (!(st.flags & kFlagDealTiles) || st.shardCount <= 1u) return k;
	const uint32_t dims = cubeMipDim(st.N, reinterpret_cast<const ushort4*>(st.volumeInfo)[volumeId].x);
	return dealtTile(k, (dims + 7u) >> 3, st.shardIndex, st.shardCount, volumeId);
}
// Stable key of a tile in the per-tile cost table (tileCost): cube tiles by (volume, face, tile of the face), screen tiles by
// their position among this frame's screen tiles.  A key that names another tile after a mip / visibility change only
// misorders the dispatch.
__host__ __device__ inline uint32_t tileCostKeyCount(uint32_t V, uint32_t N, uint32_t maxScreenTiles)
{
	const uint32_t t = (N + 7u) >> 3;
	return V * 6u * t * t + maxScreenTiles + 1u;
}
constexpr uint32_t kFlagInterleaveRows = 0x10000000u;	// DeviceState.flags (internal, mv_dist_frame contexts): this rank's screen rows are the 8-row tile rows t with t % shardCount == shardIndex (not the contiguous band), and the composite image is stored rank-major: tile row t at (t % shardCount) * ceil(T / shardCount) + t / shardCount
constexpr uint32_t kFlagBalanceOwners = 0x20000000u;	// DeviceState.flags (internal, mv_dist_frame contexts): the cull deals the cube-map volumes of a frame to the ranks by estimated march cost instead of v % shardCount (volumeFlags bits 16-31 name the owner either way)
constexpr uint32_t kFlagNoScreenLayers = 0x40000000u;	// DeviceState.flags (internal, per composite launch): this frame has had no view march, so the RayCast layers are not this frame's: the composite marches screen-space volumes itself
constexpr uint32_t kFlagLightLayers = 0x08000000u;	// DeviceState.flags (internal, peer-store transport): the light march is shared out by LAYERS of voxel blocks (block layer l of the z axis to rank l % shardCount) instead of ranges of the dispatch order: a light-map cell holds texels of one z only, so a rank's layers are whole cells, contiguous in memory -- what it pushes to the other ranks is plain ranges
// (kFlagLightLayers) which block layers of the z axis a rank marches: a slab of ceil(B / shardCount) adjacent ones (adjacent layers share
// the texels their light rays read: measured against every shardCount-th layer, profiles/r04_peer_rank_cost.txt)
#ifndef MV_LIGHT_LAYERS_INTERLEAVED
#define MV_LIGHT_LAYERS_INTERLEAVED 0
#endif
__host__ __device__ inline uint32_t lightLayersOfRank(uint32_t B, uint32_t rank, uint32_t shardCount)
{
	if (MV_LIGHT_LAYERS_INTERLEAVED) return B > rank ? (B - rank + shardCount - 1u) / shardCount : 0u;
	const uint32_t per = (B + shardCount - 1u) / shardCount, first = rank * per;
	return first >= B ? 0u : (B - first < per ? B - first : per);
}
__host__ __device__ inline uint32_t lightLayerOfRank(uint32_t B, uint32_t rank, uint32_t shardCount, uint32_t i)	// its i-th layer
{
	if (MV_LIGHT_LAYERS_INTERLEAVED) return rank + i * shardCount;
	return rank * ((B + shardCount - 1u) / shardCount) + i;
}
// (plain light push) the cells a rank REBUILDS from the texels the other ranks pushed: all L^3 of the map but its own slab [z0, z1) of slices --
// cell index k of `total` maps to cell k (k < before) or k + skip of the map.  A rank whose slab would start beyond the map (B layers over more
// ranks than ceil(B / per) of them get one: L = 48 on 8 ranks, rank 7) owns the EMPTY range at L: round 5 computed [first * 4, L) there, whose
// length wrapped -- 129 024 "cells" for a map of 110 592.  mv_light_rebuild_range states this through the C ABI for the CPU tests.
struct LightRebuildRange { uint32_t z0, z1, before, skip, total; };
__host__ __device__ inline LightRebuildRange lightRebuildRange(uint32_t L, uint32_t rank, uint32_t shardCount)
{
	LightRebuildRange r;
	const uint32_t B = (L + 3u) >> 2, plane = L * L;
	if (MV_LIGHT_LAYERS_INTERLEAVED) { r.z0 = r.z1 = 0u; r.before = 0u; r.skip = 0u; r.total = L * plane; return r; }	// (every cell is visited, a rank's own are left out one by one)
	const uint32_t per = (B + shardCount - 1u) / shardCount, mine = lightLayersOfRank(B, rank, shardCount);
	const uint64_t first = (uint64_t)rank * per;
	const uint32_t z0 = first * 4u < L ? (uint32_t)(first * 4u) : L;
	const uint32_t z1e = (first + mine) * 4u < L ? (uint32_t)((first + mine) * 4u) : L;
	r.z0 = z0; r.z1 = z1e > z0 ? z1e : z0;
	r.before = r.z0 * plane; r.skip = (r.z1 - r.z0) * plane; r.total = L * plane - r.skip;
	return r;
}
constexpr uint32_t kFlagRotateAllLightMaps = 0x80000000u;	// DeviceState.flags (internal): every rank rotates the relit volume's light buffer, not only its owner

// The 8-row tile rows of a screen rectangle [y0, y1] (pixel rows, y1 already clipped to the viewport) that this context marches and
// composites: those inside its contiguous band, or -- in the multi-GPU frame loop -- every shardCount-th one.  Work in the middle of
// the screen (the rectangles of the near volumes, most composite fragments) then spreads over all ranks instead of landing on the
// two that own the middle bands (C5 on 8 ranks: the slowest rank carried 2.2x the mean).
struct ScreenTileRows { uint32_t first, count, step; };
__host__ __device__ inline uint32_t interleavedRowsPerRank(uint32_t H, uint32_t shardCount) { return (((H + 7u) >> 3) + shardCount - 1u) / shardCount; }
// t / shardCount without a division (a scalar division is a long dependent chain, and the compiler evaluates both row partitions
// and selects: with `/` here the view march alone lost 5 % on ONE GPU, where none of this is ever taken)
__device__ __forceinline__ uint32_t divShards(const DeviceState& st, uint32_t t) { return __umulhi(t, st.shardMagic); }
// (IL = the interleaved partition, as a template parameter: the tile march is instantiated for both, because with the choice made at
// run time inside it the compiler evaluates both and selects, and the view march ALONE -- one GPU, nothing interleaved -- lost 5 %.)
template <bool IL>
__device__ __forceinline__ ScreenTileRows screenTileRowsT(const DeviceState& st, uint32_t y0, uint32_t y1)
{
	ScreenTileRows r;
	if (IL)
	{
		const uint32_t ty0 = y0 >> 3, ty1 = y1 >> 3, w = st.shardCount;
		const uint32_t m = ty0 - divShards(st, ty0) * w;	// ty0 % w
		r.first = ty0 + (st.shardIndex >= m ? st.shardIndex - m : st.shardIndex + w - m);
		r.count = (y0 <= y1 && r.first <= ty1) ? divShards(st, ty1 - r.first) + 1u : 0u;
		r.step = w;
		return r;
	}
	const uint32_t yb0 = y0 > st.bandRow0 ? y0 : st.bandRow0;
	const uint32_t yb1 = st.bandRow1 ? (y1 < st.bandRow1 - 1u ? y1 : st.bandRow1 - 1u) : 0u;
	r.first = yb0 >> 3;
	r.count = (yb0 <= yb1 && st.bandRow1 > st.bandRow0) ? (yb1 >> 3) - r.first + 1u : 0u;
	r.step = 1u;
	return r;
}
__device__ inline ScreenTileRows screenTileRows(const DeviceState& st, uint32_t y0, uint32_t y1)
{
	return (st.flags & kFlagInterleaveRows) ? screenTileRowsT<true>(st, y0, y1) : screenTileRowsT<false>(st, y0, y1);
}

// Key of a tile in the cost table (the sample count of its longest ray + 1 the last time it was marched; 0 = never): a cube tile by
// (volume, face, tile of the face), a screen tile by (screen slot, ABSOLUTE 8x8 pixel tile among this context's tile rows) -- not by its
// index in the volume's rectangle, which changes whenever the rectangle does (a moving camera).
template <bool IL>
__device__ __forceinline__ uint32_t tileCostKeyT(const DeviceState& st, uint32_t volumeId, uint32_t faceId, uint32_t localTile, uint32_t tile, uint32_t nCubeTiles)
{
	const uint32_t t = (st.N + 7u) >> 3;
	if (faceId == 7u)
	{
		const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[volumeId];
		const uint32_t x1 = rect.z < st.W - 1u ? rect.z : st.W - 1u, y1 = rect.w < st.H - 1u ? rect.w : st.H - 1u;
		const ScreenTileRows rows = screenTileRowsT<IL>(st, rect.y, y1);
		const uint32_t tx0 = rect.x >> 3, tw = (x1 >> 3) - tx0 + 1u;
		const uint32_t tilesX = (st.W + 7u) >> 3, bandTilesY = IL ? st.rowsPerRank : (st.bandRow1 - st.bandRow0 + 7u) >> 3;
		const uint32_t tyAbs = rows.first + (localTile / tw) * rows.step;
		const uint32_t tx = tx0 + localTile % tw, ty = IL ? divShards(st, tyAbs) : tyAbs - (st.bandRow0 >> 3);
		const uint32_t slot = st.screenSlot[volumeId];
		const uint32_t idx = ((slot < st.screenSlots ? slot : 0u) * bandTilesY + (ty < bandTilesY ? ty : bandTilesY - 1u)) * tilesX + (tx < tilesX ? tx : tilesX - 1u);
		return st.V * 6u * t * t + (idx < st.maxScreenTiles ? idx : st.maxScreenTiles);
	}
	return (volumeId * 6u + faceId) * t * t + localTile;
}
__device__ inline uint32_t tileCostKey(const DeviceState& st, uint32_t volumeId, uint32_t faceId, uint32_t localTile, uint32_t tile, uint32_t nCubeTiles)
{
	return (st.flags & kFlagInterleaveRows) ? tileCostKeyT<true>(st, volumeId, faceId, localTile, tile, nCubeTiles) : tileCostKeyT<false>(st, volumeId, faceId, localTile, tile, nCubeTiles);
}

// MV_TILE_BALANCE (experiment, profiles/r05_tile_timeline.txt): the tiles' cost is the TIME their wavefront took last time (10 ns ticks), binned
// four to the octave in sixteen bins, and the slots of every other group of 256 workgroups are dealt backwards: a launch that has the GPU to
// itself puts workgroups w, w + 256, w + 512 .. on the same CU (measured: period 512 slots at two wavefronts per workgroup), so dealing the
// sorted tiles there and back gives every CU the same sum of costs -- the launch ends when its slowest CU does.
#ifndef MV_TILE_BALANCE
#define MV_TILE_BALANCE 0
#endif
constexpr uint32_t kOrderBins = MV_TILE_BALANCE ? 16u : 8u;

// dispatch slot -> tile (wave-uniform: scalar loads and compares).  The cursors are the bin totals once the order kernel has run.
// `bin` (optional): the cost bin of the slot, 0 = the longest rays (or tiles never marched before) .. kOrderBins - 1 = the shortest; kOrderBins in list order.
__device__ __forceinline__ uint32_t orderedTile(const DeviceState& st, uint32_t slot, uint32_t* bin = nullptr)
{
	if (bin) *bin = kOrderBins;
	if (!st.tileOrder) return slot;
#if MV_TILE_BALANCE
	{
		uint32_t total = 0;
#pragma unroll
		for (uint32_t b = 0; b < kOrderBins; ++b) total += st.binCursor[b];
		const uint32_t w = slot >> 1, k = w >> 8;	// (whole groups of 512 slots only: a permutation of the slots there are)
		if ((k & 1u) && ((k + 1u) << 9) <= total) slot = (((k << 8) | (255u - (w & 255u))) << 1) | (slot & 1u);
	}
#endif
	uint32_t base = 0, at = slot;	// (a slot beyond the total -- never dispatched -- maps to itself)
	bool found = false;
#pragma unroll
	for (uint32_t b = 0; b < kOrderBins; ++b)
	{
		const uint32_t n = st.binCursor[b];
		if (!found && slot < base + n) { at = b * st.orderCap + (slot - base); found = true; if (bin) *bin = b; }
		base += n;
	}
	return found ? st.tileOrder[at] : slot;
}

__host__ __device__ inline uint64_t cubeChainTexels(uint32_t N)
{
	uint64_t s = 0;
	for (uint32_t m = 0; m < 5; ++m) { const uint64_t d = cubeMipDim(N, m); s += 6 * d * d; }
	return s;
}

__host__ __device__ inline uint64_t cubeMipOffset(uint32_t N, uint32_t mip)
{
	uint64_t s = 0;
	for (uint32_t m = 0; m < mip; ++m) { const uint64_t d = cubeMipDim(N, m); s += 6 * d * d; }
	return s;
}
// ---- push roles (mv_peer.hip explains the transport): copy what this rank produced into every OTHER rank's buffers ----
// the range [off, off + bytes) of `src` (multiples of 4 bytes) into table[p] for every p != self; thread t of n
template <typename T>
__device__ __forceinline__ void peerPushRange(const DeviceState& st, T* const* table, const T* src, size_t off, size_t bytes, size_t t, size_t n, uint32_t onlyTo = 0xffffffffu)
{
	const uint32_t p0 = onlyTo == 0xffffffffu ? 0u : onlyTo, p1 = onlyTo == 0xffffffffu ? st.peerCount : onlyTo + 1u;	// every other rank, or one
	const uint8_t* s = reinterpret_cast<const uint8_t*>(src);
	size_t head = (16u - ((reinterpret_cast<size_t>(s) + off) & 15u)) & 15u;	// (every rank's buffer has the same alignment: allocations of the same size from the same allocator, checked at init)
	if (head > bytes) head = bytes;
	const size_t body = (bytes - head) >> 4, tail = bytes - head - (body << 4);
	// Four 16-byte units per thread and batch, ALL loads in front of the stores: vmcnt counts loads and stores in order, so a load
	// behind a write-through store waits for that store's acknowledgement from the far memory -- one unit per iteration ran at the
	// latency of seven such acknowledgements per 16 bytes (measured: 12 MB in 31 us).
	for (size_t i = t; i < body; i += 4u * n)
	{
		uint4 v[4];
#pragma unroll
		for (uint32_t j = 0; j < 4u; ++j) if (i + j * n < body) v[j] = *reinterpret_cast<const uint4*>(s + off + head + ((i + j * n) << 4));
#pragma unroll
		for (uint32_t j = 0; j < 4u; ++j)
			if (i + j * n < body)
				for (uint32_t p = p0; p < p1; ++p)
					if (p != st.shardIndex) peerStore16(reinterpret_cast<uint8_t*>(table[p]) + off + head + ((i + j * n) << 4), v[j]);
	}
	// (the few words in front of and behind the aligned body)
	for (size_t i = t; i < (head + tail) >> 2; i += n)
	{
		const size_t o = off + (i < (head >> 2) ? (i << 2) : head + (body << 4) + ((i - (head >> 2)) << 2));
		const uint32_t v = *reinterpret_cast<const uint32_t*>(s + o);
		for (uint32_t p = p0; p < p1; ++p)
			if (p != st.shardIndex) peerStore(reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(table[p]) + o), v);
	}
}

// This rank's layers of the relit volume's light map (kFlagLightLayers, lightLayersOfRank: a layer is 4 z slices of L x L whole cells,
// contiguous) into every other rank's copy of the map; thread t of n.
__device__ __forceinline__ void peerPushLightRole(const DeviceState& st, size_t t, size_t n)
{
	const uint32_t* cbWords = reinterpret_cast<const uint32_t*>(st.cbPerFrame);
	const uint32_t frameIdx = cbWords[52];	// CBPerFrame.FrameIdx (byte 208)
	const uint32_t count = st.counters[0];
	// CSRayMarchL.hlsl:29-33 (mv_kernels.hip relitVolume)
	const uint32_t volumeId = count ? st.visibleVolumes[frameIdx % count] : frameIdx % st.V;
	const size_t mapOff = reinterpret_cast<const uint8_t*>(st.lightTable[volumeId]) - reinterpret_cast<const uint8_t*>(st.lightBuffers);
	const uint32_t B = (st.L + 3u) >> 2;
	const size_t slice = (size_t)st.L * st.L * sizeof(uint4);
	const uint32_t mine = lightLayersOfRank(B, st.shardIndex, st.shardCount);
	for (uint32_t i = 0; i < mine; ++i)
	{
		const uint32_t layer = lightLayerOfRank(B, st.shardIndex, st.shardCount, i);
		const uint32_t z0 = layer * 4u, z1 = z0 + 4u < st.L ? z0 + 4u : st.L;
		if (st.lightPlain)
		{
			// plain texels (the light march left them in this rank's staging map beside the cells it wrote): a quarter of the bytes
			const size_t plane = (size_t)st.L * st.L * sizeof(uint32_t);
			peerPushRange(st, st.peerLightPlain, st.lightPlain - st.lightPlainOff, st.lightPlainOff * sizeof(uint32_t) + (size_t)z0 * plane, (size_t)(z1 - z0) * plane, t, n);
		}
		else peerPushRange(st, st.peerLight, st.lightBuffers, mapOff + (size_t)z0 * slice, (size_t)(z1 - z0) * slice, t, n);
	}
}

// (plain light push) the cells of the OTHER ranks' layers of the relit map, rebuilt from the texels those ranks pushed into this rank's
// staging map: cell (x, y, z) = {t(x, y), t(x + 1, y), t(x, y + 1), t(x + 1, y + 1)} of slice z, +1 clamped -- a cell holds texels of one
// z only, so a rank's layers rebuild from that rank's texels alone.  Written through (sc0 sc1), like everything another CU's plain loads read
// behind a flag inside the same launch: plain stores + one agent-scope release per workgroup were measured slower, and slower the more
// workgroups there are -- every buffer_wbl2 walks its XCD's whole L2 (profiles/r05_plain_light_push.txt).  Thread t of n.
__device__ __forceinline__ void peerRebuildLightRole(const DeviceState& st, size_t t, size_t n)
{
	const uint32_t* cbWords = reinterpret_cast<const uint32_t*>(st.cbPerFrame);
	const uint32_t frameIdx = cbWords[52];
	const uint32_t count = st.counters[0];
	const uint32_t volumeId = count ? st.visibleVolumes[frameIdx % count] : frameIdx % st.V;	// CSRayMarchL.hlsl:29-33
	uint4* cells = st.lightTable[volumeId];
	const uint32_t L = st.L;
	const size_t plane = (size_t)L * L;
	// Consecutive lanes take consecutive cells (every load and the 16-byte store of a wavefront are contiguous), four cells per thread and
	// iteration with all sixteen loads in front of the four stores: vmcnt counts loads and stores in order, so a thread's next loads queue
	// behind its stores -- one cell per iteration with write-through stores ran at a store acknowledgement per cell, 95 cells per thread
	// (profiles/r05_plain_light_push.txt)
	// (32-bit indices: L <= 640 keeps L^3 below 2^28, and a 64-bit division per cell cost more than the cell)
	const LightRebuildRange rr = lightRebuildRange(L, st.shardIndex, st.shardCount);
	const uint32_t total32 = rr.total, before32 = rr.before, skip32 = rr.skip, plane32 = (uint32_t)plane, n32 = (uint32_t)n;
	for (uint32_t k = (uint32_t)t; k < total32; k += 4u * n32)
	{
		uint4 c[4];
		uint32_t at[4];
#pragma unroll
		for (uint32_t u = 0; u < 4u; ++u)
		{
			const uint32_t kk = k + u * n32;
			const uint32_t i = MV_LIGHT_LAYERS_INTERLEAVED ? kk : (kk < before32 ? kk : kk + skip32);	// skip this rank's own slab
			at[u] = kk < total32 ? i : 0xffffffffu;
			if (kk >= total32) continue;
			const uint32_t z = i / plane32, r = i - z * plane32, y = r / L, x = r - y * L;
			if (MV_LIGHT_LAYERS_INTERLEAVED && ((z >> 2) % st.shardCount) == st.shardIndex) { at[u] = 0xffffffffu; continue; }
			const uint32_t x1 = x + 1u < L ? x + 1u : L - 1u, y1 = y + 1u < L ? y + 1u : L - 1u;
			const uint32_t* s = st.lightPlain + (size_t)z * plane;
			c[u] = make_uint4(s[y * L + x], s[y * L + x1], s[y1 * L + x], s[y1 * L + x1]);
		}
#pragma unroll
		for (uint32_t u = 0; u < 4u; ++u) if (at[u] != 0xffffffffu) peerStore16(cells + at[u], c[u]);
	}
}

void launchRetile(hipStream_t s, const uint2* linear, uint2* bricked, uint32_t N, uint32_t bricks);
void launchUntile(hipStream_t s, const uint2* bricked, uint2* linear, uint32_t N, uint32_t bricks);
void launchInitGridData(hipStream_t s, uint2* bricked, uint32_t N, uint32_t bricks);
void launchR32FToRGBA16F(hipStream_t s, const float* src, uint32_t w, uint32_t h, uint32_t d, uint2* bricked, uint32_t N, uint32_t bricks);
void launchBuildAlphaCells(hipStream_t s, const uint2* bricked, uint4* cells, uint32_t N, uint32_t bricks);
void launchBuildOccupancy(hipStream_t s, const uint2* bricked, uint64_t* words, uint32_t N, uint32_t bricks);
void launchLightToQuads(hipStream_t s, const uint32_t* linear, uint4* quads, uint32_t L);
void launchLightFromQuads(hipStream_t s, const uint4* quads, uint32_t* linear, uint32_t L);
void launchCull(hipStream_t s, const DeviceState& st);
void launchOrders(hipStream_t s, const DeviceState& st);	// behind the cull, one launch: the cost-binned dispatch order of the view march's tiles + the XCD-aware order of the light march's voxel blocks
void launchMarchLight(hipStream_t s, const DeviceState& st, uint32_t* wordsOut = nullptr);	// wordsOut: this context's share of the relit map as plain words (multi-GPU frame loop), or NULL
uint32_t lightSharePerRank(const DeviceState& st);	// voxel blocks per rank of the light march's share-out (all of them outside the multi-GPU frame loop)
void launchLightUnpack(hipStream_t s, const DeviceState& st, const uint32_t* words);	// the all-gathered shares (64 words per block, lightSharePerRank blocks per rank) into the relit volume's light map
uint32_t launchMarchTiles(hipStream_t s, const DeviceState& st, uint32_t gridTiles);	// returns the first slot its grid does not cover (0xffffffff: none)
uint32_t marchTilesBeyond(const DeviceState& st, uint32_t gridTiles);	// what launchMarchTiles would return for that estimate, without launching
bool marchTilesNarrow(const DeviceState& st, uint64_t tiles);	// a launch of that many tiles takes the latency-oriented form (mv_march_narrow.hip: eight lanes per ray, speculative requests)
void launchMarchTilesNarrow(hipStream_t s, const DeviceState& st, uint32_t blocks);	// one workgroup per dispatch slot
void launchMarchTilesOverflow(hipStream_t s, const DeviceState& st, uint32_t firstSlot);	// one wavefront per tile (mv_march_tiles.hip)
// Several frames per launch (DeviceStateBatch; peer-store transport, folded flags: every state carries its launch's fold fields, the frames
// of a launch share counters, rows and values).  gridTiles[i]: the host's estimate of frame i's tile count; beyond[i]: the first dispatch slot
// of frame i the grid does not cover (0xffffffff: none) -- launchMarchTilesOverflowBatch marches those.
void launchCullBatch(hipStream_t s, const DeviceState* states, uint32_t n);	// consecutive frames: frame i + 1's workgroups wait for frame i's inside the launch
void launchOrdersBatch(hipStream_t s, const DeviceState* states, uint32_t n);
void launchMarchLightBatch(hipStream_t s, const DeviceState* states, uint32_t n);
void launchMarchTilesBatch(hipStream_t s, const DeviceState* states, uint32_t n, const uint32_t* gridTiles, uint32_t* beyond);
void launchMarchTilesOverflowBatch(hipStream_t s, const DeviceState* states, uint32_t n, const uint32_t* firstSlot);
uint32_t launchCompositeBatch(hipStream_t s, const DeviceState* states, uint32_t n);	// every state's own tile rows (kFlagInterleaveRows); returns the kernels launched
void launchCopyForwardBatch(hipStream_t s, const DeviceState* states, uint32_t first, uint32_t n, const uint2* prevMaps, const float* prevDepths, const uint32_t* prevStamps, const uint32_t* tags);	// frames first .. n - 1, one after the other in one kernel; states[0].fold: the wait before a set of the group is read
void launchRaySetup(hipStream_t s, const DeviceState& st);
void launchMarchView(hipStream_t s, const DeviceState& st);	// st.marchWaves wavefronts (see marchViewWaves)
uint32_t marchViewResidentWaves(int device, bool wideAddressing);	// wavefronts of the march kernel the device holds at once (occupancy query)
void launchReduceStats(hipStream_t s, const DeviceState& st, int includeLight);
uint32_t marchViewMaxTiles(uint32_t V, uint32_t N, uint32_t shardCount);
uint32_t launchComposite(hipStream_t s, const DeviceState& st, uint32_t row0, uint32_t row1, const void* background = nullptr);	// background: W x H RGBA16F (device) or NULL; returns the kernels launched (0: this context owns no rows)
void launchExchangePack(hipStream_t s, const DeviceState& st, uint32_t slotDim, void* dst);
void launchExchangeUnpack(hipStream_t s, const DeviceState& st, uint32_t slotDim, const void* src);
void launchExchangeCompact(hipStream_t s, const DeviceState& st, bool pack, void* buf, size_t shardBytes);	// marched faces only (mv_dist_frame)
void launchCopyForward(hipStream_t s, const DeviceState& st, const uint2* otherMaps, const float* otherDepths, const uint32_t* otherStamps, uint32_t tag, bool marchFollows = true);

// ---- peer-store transport (mv_peer.hip) ----
struct PeerSync
{
	unsigned long long* const* peerFlags;	// [world] every rank's flag block (device table)
	unsigned long long* myFlags;			// this rank's
	uint32_t world, self, loopback;			// loopback: a rank measured alone -- it signals every column of its own block
	uint32_t nSignal, signalRow[2];			// rows to raise to signalValue in every rank's block (after everything before this kernel on its stream)
	unsigned long long signalValue[2];
	uint32_t nWait, waitRow[8];				// rows of this rank's block that must reach waitValue in every column
	uint32_t waitLocal;						// != 0: the rows are local ones (only this rank's own column is ever raised: "light cells rebuilt")
	unsigned long long waitValue[8];
	uint32_t checkChain;					// 0xffffffff, or: signal with this rank's cull words of the frame, and compare every rank's after the wait
	const uint32_t* counters;				// the frame's cull counters (device)
	uint32_t frameTag;
	unsigned long long timeoutTicks;		// wall_clock64 ticks (100 MHz) a wait may last
	unsigned long long* report;				// pinned: [1] = {frameTag : 1} the ranks' culls disagree, [2] = {frameTag : row << 8 | column} a wait timed out
};
void launchPeerSync(hipStream_t s, const PeerSync& a);
// Behind a producer on its stream: copy what it produced into every other rank's buffers; the last workgroup raises st.fold's flags.
void launchPeerPushLight(hipStream_t s, const DeviceState& st);	// this rank's layers of the relit volume's light map
void launchPeerRebuildLight(hipStream_t s, const DeviceState& st);	// (plain light push, waits as kernels) the other ranks' layers: cells from the landed texels; raises fold.rebuildRow locally
void launchPeerPattern(hipStream_t s, uint2* const* table, uint32_t world, uint32_t self, size_t wordsPerRank, uint32_t salt);	// self-test: a pattern into every rank's buffer at this rank's region
void launchPeerVerify(hipStream_t s, const uint2* mine, uint32_t world, size_t wordsPerRank, uint32_t salt, uint32_t* mismatches);
// The self-test of what the folded waits rest on, in ONE launch per rank (mv_peer.hip peerCoherenceKernel): read every rank's region of
// this rank's buffer (the lines are in this GPU's caches now), tell everybody, wait INSIDE the kernel until everybody has read, overwrite
// this rank's region of every rank's buffer with a new pattern, raise, wait inside the kernel for everybody's, acquire as the folded
// waits do, read again with plain loads: a word of the old pattern is a stale line.
struct PeerCoherence
{
	uint2* const* table;					// [world] every rank's buffer
	uint2* mine;
	unsigned long long* const* peerFlags;	// [world]
	unsigned long long* myFlags;
	uint32_t world, self, row, acquire;
	size_t wordsPerRank;
	uint32_t saltOld, saltNew;
	unsigned long long valueRead, valueWritten;	// the flag row's values: "this rank has read", "this rank has written"
	uint32_t* counters;						// 2, zero
	uint32_t* mismatches;					// [0] += words of the second read that are not the new pattern, [1] += words of the FIRST read that are not the old one
	unsigned long long timeoutTicks;
	unsigned long long* report;
};
void launchPeerCoherence(hipStream_t s, const PeerCoherence& a);
} // namespace mv
