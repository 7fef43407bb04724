// mv_peer.hip -- synchronisation of the peer-store transport (mv_dist_init_peer; SURVEY 8e "7 simultaneous direct peer writes").
//
// In this transport there are no collectives: the kernels that produce a frame's light-map share, cube faces and screen rows store
// them straight into every rank's copy (mv_kernels.hpp storeToPeers), over xGMI peer mappings.  What is left of a collective is its
// ordering, and that is one flag per (rank, stage, chain): a rank RAISES its column of a row in every rank's flag block when a stage
// of a frame is complete on its side, and WAITS until every column of a row of its own block has reached the frame it needs.
// Memory model: the flag blocks are uncached device memory (hipDeviceMallocUncached); a signal is a system-scope release store made
// by a kernel that runs behind the producing kernel on the same stream (everything that kernel stored -- written through, see
// peerStore -- is complete and visible before the flag is); a wait polls with system-scope acquire loads in a kernel of its own, and
// the consumer is a LATER kernel on the same stream.  That is the pattern of a collective's kernels (flag written behind the data,
// the consumer launched behind the kernel that saw the flag), with the data path left out.
// A wait never hangs: it gives up after timeoutTicks and leaves {frame : row, column} in pinned memory, which the host turns into
// MV_ERR_TIMEOUT at its next call.
#include "mv_kernels.hpp"

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
		for (uint32_t i = 0; i < a.nWait; ++i)
			for (uint32_t p = lane; p < a.world; p += 64u)
			{
				const unsigned long long* flag = a.myFlags + (size_t)a.waitRow[i] * kPeerMaxRanks + p;
				while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.waitValue[i])
				{
					if (wall_clock64() - t0 > a.timeoutTicks)
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
			else peerStore(table[p] + (size_t)self * wordsPerRank + i, v);
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

void launchPeerPattern(hipStream_t s, uint2* const* table, uint32_t world, uint32_t self, size_t wordsPerRank, uint32_t salt)
{
	hipLaunchKernelGGL(peerPatternKernel, dim3(64), dim3(256), 0, s, table, world, self, wordsPerRank, salt);
}

void launchPeerVerify(hipStream_t s, const uint2* mine, uint32_t world, size_t wordsPerRank, uint32_t salt, uint32_t* mismatches)
{
	hipLaunchKernelGGL(peerVerifyKernel, dim3(64), dim3(256), 0, s, mine, world, wordsPerRank, salt, mismatches);
}
} // namespace mv
