// mv_peer.hip -- synchronisation of the peer-store transport (mv_dist_init_peer; SURVEY 8e "7 simultaneous direct peer writes").
//
// In this transport there are no collectives: behind the kernels that produce a frame's light-map share, cube faces and screen rows a
// push kernel copies them straight into every rank's buffers, over xGMI peer mappings (below).  What is left of a collective is its
// ordering, and that is one flag per (rank, stage, chain): a rank RAISES its column of a row in every rank's flag block when a stage
// of a frame is complete on its side, and WAITS until every column of a row of its own block has reached the frame it needs.
// Memory model: the flag blocks are uncached device memory (hipDeviceMallocUncached).  Data: every store into another rank's memory is
// a write-through store at system scope (peerStore / peerStore16: acknowledged by the memory that owns the address); a push kernel's
// wavefronts drain them (s_waitcnt vmcnt(0)) before they count themselves in, and the workgroup that counts last raises the flags --
// so a flag is issued only after every store it announces has been acknowledged (mv_kernels.hpp peerFoldSignal).  A wait is either a
// one-wavefront kernel of its own in front of the consumer (peerSyncKernel: ranks that share a GPU) or the consumer's own prologue
// (peerFoldWait: ranks that own their GPU); either way the consumer's loads of the announced data are issued behind the poll that saw
// the flag, by a dispatch that has not read those addresses before (the buffers rotate: what a frame's peers write was last read C
// frames ago).  This is the pattern of a collective's kernels -- flag behind the data, reads behind the flag.
// A wait never hangs: it gives up after timeoutTicks and leaves {frame : row, column} in pinned memory, which the host turns into
// MV_ERR_TIMEOUT at its next call.
#include "mv_kernels.hpp"
#include "mv_device.hpp"

namespace mv
{
__global__ __launch_bounds__(64) void peerSyncKernel(PeerSync a)
{
	const uint32_t lane = threadIdx.x;
	// ---- raise ----
	if (a.nSignal)
	{
		for (uint32_t p = lane; p < a.world; p += 64u)
		{
			// loopback: this rank's own block, column p (it stands in for every peer)
			unsigned long long* block = a.loopback ? a.myFlags : a.peerFlags[p];
			const uint32_t column = a.loopback ? p : a.self;
			if (a.checkChain != 0xffffffffu)
			{
				uint32_t* words = reinterpret_cast<uint32_t*>(block + kPeerFlagWords) + ((size_t)a.checkChain * kPeerMaxRanks + column) * kPeerCheckWords;
				peerStore(words + 0, a.frameTag); peerStore(words + 1, a.counters[12]); peerStore(words + 2, a.counters[0]); peerStore(words + 3, a.counters[1]);
			}
			for (uint32_t i = 0; i < a.nSignal; ++i)
				__hip_atomic_store(block + (size_t)a.signalRow[i] * kPeerMaxRanks + column, a.signalValue[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	// ---- wait ----
	if (a.nWait)
	{
		const unsigned long long t0 = wall_clock64();
		const unsigned long long patience = a.timeoutTicks << (a.waitLocal ? 1u : 0u);	// (a local row: mv_kernels.hpp peerFoldWait)
		for (uint32_t i = 0; i < a.nWait; ++i)
			for (uint32_t p = lane; p < a.world; p += 64u)
			{
				if (a.waitLocal && p != a.self) continue;
				const unsigned long long* flag = a.myFlags + (size_t)a.waitRow[i] * kPeerMaxRanks + p;
				while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.waitValue[i])
				{
					if (wall_clock64() - t0 > patience)
					{
						__hip_atomic_store(a.report + 2, ((unsigned long long)a.frameTag << 32) | ((unsigned long long)a.waitRow[i] << 8) | p | 0x80000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
						break;
					}
					__builtin_amdgcn_s_sleep(8);
				}
			}
		if (a.checkChain != 0xffffffffu)
		{
			// every rank's cull must have decided the same for this frame (same inputs, same kernel): a caller that handed the ranks
			// different frame constants gets an error at its next call instead of a wrong image
			const uint32_t* all = reinterpret_cast<const uint32_t*>(a.myFlags + kPeerFlagWords) + (size_t)a.checkChain * kPeerMaxRanks * kPeerCheckWords;
			const uint32_t mine[4] = { a.frameTag, a.counters[12], a.counters[0], a.counters[1] };
			bool differs = false;
			for (uint32_t p = lane; p < a.world; p += 64u)
				for (uint32_t w = 0; w < kPeerCheckWords; ++w)
					differs |= __hip_atomic_load(all + (size_t)p * kPeerCheckWords + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != mine[w];
			if (differs) __hip_atomic_store(a.report + 1, ((unsigned long long)a.frameTag << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

void launchPeerSync(hipStream_t s, const PeerSync& a)
{
	hipLaunchKernelGGL(peerSyncKernel, dim3(1), dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------------------------
// Push kernels: behind the light march and the composite on their stream, copy what they produced into every OTHER rank's buffers -- the
// same byte ranges at the same offsets, every rank's buffers having one layout -- with 16-byte write-through stores, and raise the
// flags from the last workgroup (peerFoldSignal).  One kernel where a collective needs three (pack, all-gather, unpack).  (The view
// march stores its texels into the other ranks' cube maps itself, mv_march_tiles.hip: its stores are 8- and 4-byte texels that pair up
// into 16 bytes inside a wavefront and overlap the marching of the other tiles; the light march's and the composite's were measured
// both ways, profiles/r04_peer_store_forms.txt.)
// ---------------------------------------------------------------------------------------------
// this rank's layers of the relit volume's light map (mv_kernels.hpp peerPushLightRole), as a kernel of its own: ranks that share a GPU, whose
// view marches wait in front of their launch -- ranks that own their GPU run the same role as the leading workgroups of the view march
__global__ __launch_bounds__(256) void peerPushLightKernel(DeviceState st)
{
	peerPushLightRole(st, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
	peerFoldSignal(st);
}

// (plain light push, waits as kernels: behind the wait for every rank's texels on this stream) the other ranks' layers of the relit map
__global__ __launch_bounds__(256) void peerRebuildLightKernel(DeviceState st)
{
	peerRebuildLightRole(st, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
	peerRaise(st, st.fold.rebuildCounter, gridDim.x, 1u, &st.fold.rebuildRow, &st.fold.rebuildValue, 0xffffffffu, true);
}
void launchPeerRebuildLight(hipStream_t s, const DeviceState& st) { hipLaunchKernelGGL(peerRebuildLightKernel, dim3(128), dim3(256), 0, s, st); }

void launchPeerPushLight(hipStream_t s, const DeviceState& st) { hipLaunchKernelGGL(peerPushLightKernel, dim3(128), dim3(256), 0, s, st); }

// ---- self-test of the mappings: rank r writes a pattern into words [r * wordsPerRank, (r + 1) * wordsPerRank) of every rank's buffer ----
__device__ __forceinline__ uint2 peerPatternWord(uint32_t rank, size_t i, uint32_t salt)
{
	const uint32_t x = (uint32_t)i * 2654435761u + rank * 40503u + salt * 2246822519u;
	return make_uint2(x ^ (x >> 15), (x * 3266489917u) ^ rank);
}

__global__ __launch_bounds__(256) void peerPatternKernel(uint2* const* table, uint32_t world, uint32_t self, size_t wordsPerRank, uint32_t salt)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < wordsPerRank; i += (size_t)gridDim.x * blockDim.x)
	{
		const uint2 v = peerPatternWord(self, i, salt);
		for (uint32_t p = 0; p < world; ++p)
			if (p == self) table[p][(size_t)self * wordsPerRank + i] = v;
			else { peerStore(&table[p][(size_t)self * wordsPerRank + i].x, v.x); peerStore(&table[p][(size_t)self * wordsPerRank + i].y, v.y); }
	}
}

__global__ __launch_bounds__(256) void peerVerifyKernel(const uint2* mine, uint32_t world, size_t wordsPerRank, uint32_t salt, uint32_t* mismatches)
{
	uint32_t bad = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < wordsPerRank * world; i += (size_t)gridDim.x * blockDim.x)
	{
		const uint2 v = mine[i], e = peerPatternWord((uint32_t)(i / wordsPerRank), i % wordsPerRank, salt);
		bad += v.x != e.x || v.y != e.y;
	}
	if (bad) atomicAdd(mismatches, bad);
}

// ---- in-kernel coherence self-test (mv_kernels.hpp PeerCoherence) ----
// What a folded wait does -- poll inside the kernel, acquire, read with plain loads -- under the worst case for it: every line it then reads
// was read by this very launch before the other ranks overwrote it.  64 workgroups, all resident; a thread reads the same words both times.
__device__ __forceinline__ void coherenceRaise(const PeerCoherence& a, uint32_t* counter, unsigned long long value)
{
	__shared__ uint32_t sLast;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0u) sLast = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;
	__syncthreads();
	if (sLast && threadIdx.x < 64u)
		for (uint32_t p = threadIdx.x; p < a.world; p += 64u)
			__hip_atomic_store(a.peerFlags[p] + (size_t)a.row * kPeerMaxRanks + a.self, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void coherenceWait(const PeerCoherence& a, unsigned long long value, bool acquire)
{
	if (threadIdx.x < 64u)
	{
		const unsigned long long t0 = wall_clock64();
		for (uint32_t p = threadIdx.x; p < a.world; p += 64u)
		{
			const unsigned long long* flag = a.myFlags + (size_t)a.row * kPeerMaxRanks + p;
			while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value)
			{
				if (wall_clock64() - t0 > a.timeoutTicks)
				{
					__hip_atomic_store(a.report + 2, ((unsigned long long)a.row << 8) | p | 0x80000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					break;
				}
				__builtin_amdgcn_s_sleep(8);
			}
		}
		if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	__syncthreads();
}
__global__ __launch_bounds__(256) void peerCoherenceKernel(PeerCoherence a)
{
	const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n = (size_t)gridDim.x * blockDim.x, total = a.wordsPerRank * a.world;
	// 1. the first read: the old pattern, into this CU's L1 and this XCD's L2
	uint32_t badOld = 0;
	for (size_t i = t; i < total; i += n)
	{
		const uint2 v = a.mine[i], e = peerPatternWord((uint32_t)(i / a.wordsPerRank), i % a.wordsPerRank, a.saltOld);
		badOld += v.x != e.x || v.y != e.y;
	}
	if (badOld) atomicAdd(a.mismatches + 1, badOld);
	coherenceRaise(a, a.counters + 0, a.valueRead);
	coherenceWait(a, a.valueRead, false);	// everybody has read: nobody's first read can see the new pattern
	// 2. this rank's region of every rank's buffer: the new pattern, written through -- its own copy too, as the frame loop's kernels write
	// what a later role of the SAME launch reads (the rebuilt light cells): a plain store would stay dirty in the writer's XCD, and the
	// word's reader below is a workgroup of another XCD whenever wordsPerRank is not a multiple of 8 workgroups' words (odd viewports)
	for (size_t i = t; i < a.wordsPerRank; i += n)
	{
		const uint2 v = peerPatternWord(a.self, i, a.saltNew);
		for (uint32_t p = 0; p < a.world; ++p)
			peerStore8(&a.table[p][(size_t)a.self * a.wordsPerRank + i], v);
	}
	coherenceRaise(a, a.counters + 1, a.valueWritten);
	// 3. the folded wait, and the second read
	coherenceWait(a, a.valueWritten, a.acquire != 0u);
	uint32_t bad = 0;
	for (size_t i = t; i < total; i += n)
	{
		const uint2 v = a.mine[i], e = peerPatternWord((uint32_t)(i / a.wordsPerRank), i % a.wordsPerRank, a.saltNew);
		bad += v.x != e.x || v.y != e.y;
	}
	if (bad) atomicAdd(a.mismatches + 0, bad);
}
void launchPeerCoherence(hipStream_t s, const PeerCoherence& a) { hipLaunchKernelGGL(peerCoherenceKernel, dim3(64), dim3(256), 0, s, a); }

void launchPeerPattern(hipStream_t s, uint2* const* table, uint32_t world, uint32_t self, size_t wordsPerRank, uint32_t salt)
{
	hipLaunchKernelGGL(peerPatternKernel, dim3(64), dim3(256), 0, s, table, world, self, wordsPerRank, salt);
}

void launchPeerVerify(hipStream_t s, const uint2* mine, uint32_t world, size_t wordsPerRank, uint32_t salt, uint32_t* mismatches)
{
	hipLaunchKernelGGL(peerVerifyKernel, dim3(64), dim3(256), 0, s, mine, world, wordsPerRank, salt, mismatches);
}
} // namespace mv
