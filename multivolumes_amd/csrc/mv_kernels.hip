// mv_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the cube-map-space ray-march path.
//
//   cullKernel        = Sh/CSVolumeCull.hlsl:13-78 + Sh/VolumeCull.hlsli           (a1)
//   marchLightKernel  = Sh/CSRayMarchL.hlsl:20-121                                 (a2)
//   marchViewKernel   = Sh/CSRayMarch.hlsl:77-158 compiled as CSRayMarchV          (a3)
//   initGridDataKernel / r32fToRGBA16FKernel = Sh/CSInitGridData.hlsl:10-27 / Sh/CSR32FToRGBA16F.hlsl:16-26 (a6)
//
// Wave64 throughout: one wavefront marches one 8x8 cube-face tile (the reference's thread-group tile,
// CSRayMarch.hlsl:77), eight 8-lane clusters per wavefront cull eight volumes (the reference packs four
// into a wave32, CSVolumeCull.hlsl:13-22).  No MFMA: the path is gather/stencil work bound by the memory system.
#include "mv_device.hpp"
#include "mv_kernels.hpp"

#include <stdlib.h>

namespace mv
{
// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t waveSumU32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// minimum over the wavefront, in every lane: a DPP reduction (row shifts, then the row broadcasts of gfx9), a dozen cycles per step
// where a __shfl goes through the LDS crossbar
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dppMinStepU64(unsigned long long v)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
	const unsigned long long o = ((unsigned long long)hi << 32) | lo;	// (a lane without a source, or a masked row, sees the identity)
	return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long waveMinU64(unsigned long long v)
{
	v = dppMinStepU64<0x111, 0xf>(v);	// row_shr:1
	v = dppMinStepU64<0x112, 0xf>(v);	// row_shr:2
	v = dppMinStepU64<0x114, 0xf>(v);	// row_shr:4
	v = dppMinStepU64<0x118, 0xf>(v);	// row_shr:8 -> lane 15 of every row holds the row's minimum
	v = dppMinStepU64<0x142, 0xa>(v);	// row_bcast:15 into rows 1 and 3
	v = dppMinStepU64<0x143, 0xc>(v);	// row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront's
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
	return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t waveExclusiveScanU32(uint32_t v, uint32_t lane, uint32_t& total)
{
	uint32_t incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1)
	{
		const uint32_t n = __shfl_up(incl, o);
		if (lane >= (uint32_t)o) incl += n;
	}
	total = __shfl(incl, 63);
	return incl - v;
}

// ---------------------------------------------------------------------------------------------
// volume producers / layout converters
// ---------------------------------------------------------------------------------------------
// One thread per texel of the BRICKED destination (coalesced 8-byte stores); reads are gathers from the linear source.
__global__ __launch_bounds__(256) void retileKernel(const uint2* __restrict__ linear, uint2* __restrict__ bricked, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
	{
		const size_t brick = i >> 6;
		const uint32_t in = (uint32_t)(i & 63);
		const uint32_t bx = (uint32_t)(brick % bricks), by = (uint32_t)((brick / bricks) % bricks), bz = (uint32_t)(brick / ((size_t)bricks * bricks));
		const uint32_t x = bx * 4 + (in & 3), y = by * 4 + ((in >> 2) & 3), z = bz * 4 + (in >> 4);
		uint2 t = make_uint2(0u, 0u);
		if (x < N && y < N && z < N) t = linear[((size_t)z * N + y) * N + x];
		bricked[volBrickIndex(bx, by, bz, bricks) * 64 + in] = t;
	}
}

__global__ __launch_bounds__(256) void untileKernel(const uint2* __restrict__ bricked, uint2* __restrict__ linear, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)N * N * N;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
	{
		const uint32_t x = (uint32_t)(i % N), y = (uint32_t)((i / N) % N), z = (uint32_t)(i / ((size_t)N * N));
		linear[i] = bricked[volTexelIndex(x, y, z, bricks)];
	}
}

// Sh/CSInitGridData.hlsl:10-27
__global__ __launch_bounds__(256) void initGridDataKernel(uint2* __restrict__ bricked, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	const float gridSize = (float)N;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
	{
		const size_t brick = i >> 6;
		const uint32_t in = (uint32_t)(i & 63);
		const uint32_t bx = (uint32_t)(brick % bricks), by = (uint32_t)((brick / bricks) % bricks), bz = (uint32_t)(brick / ((size_t)bricks * bricks));
		const uint32_t x = bx * 4 + (in & 3), y = by * 4 + ((in >> 2) & 3), z = bz * 4 + (in >> 4);
		uint2 t = make_uint2(0u, 0u);
		if (x < N && y < N && z < N)
		{
			const vec3 pos = v3(((float)x + 0.5f) / gridSize * 2.0f - 1.0f, ((float)y + 0.5f) / gridSize * 2.0f - 1.0f, ((float)z + 0.5f) / gridSize * 2.0f - 1.0f);
			const float r_sq = dot3(pos, pos);
			float a = 1.0f - r_sq;
			a *= a;
			a = saturatef(a * a * 2.0f);
			const float s = saturatef(pos.y * 0.5f + 0.2f);
			// lerp(colorD, colorU, s): colorU = (1, .6, 0), colorD = (.5, .8, 1)
			const float r = lerpf(0.5f, 1.0f, s), g = lerpf(0.8f, 0.6f, s), b = lerpf(1.0f, 0.0f, s);
			t.x = (uint32_t)float_to_half_bits(r) | ((uint32_t)float_to_half_bits(g) << 16);
			t.y = (uint32_t)float_to_half_bits(b) | ((uint32_t)float_to_half_bits(a) << 16);
		}
		bricked[volBrickIndex(bx, by, bz, bricks) * 64 + in] = t;
	}
}

// Sh/CSR32FToRGBA16F.hlsl:16-26
__global__ __launch_bounds__(256) void r32fToRGBA16FKernel(const float* __restrict__ src, uint32_t w, uint32_t h, uint32_t d,
	uint2* __restrict__ bricked, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	const float gridSize = (float)N;
	const uint32_t one = float_to_half_bits(1.0f);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
	{
		const size_t brick = i >> 6;
		const uint32_t in = (uint32_t)(i & 63);
		const uint32_t bx = (uint32_t)(brick % bricks), by = (uint32_t)((brick / bricks) % bricks), bz = (uint32_t)(brick / ((size_t)bricks * bricks));
		const uint32_t x = bx * 4 + (in & 3), y = by * 4 + ((in >> 2) & 3), z = bz * 4 + (in >> 4);
		uint2 t = make_uint2(0u, 0u);
		if (x < N && y < N && z < N)
		{
			const Taps tx = linearTaps(((float)x + 0.5f) / gridSize, w, 0), ty = linearTaps(((float)y + 0.5f) / gridSize, h, 0), tz = linearTaps(((float)z + 0.5f) / gridSize, d, 0);
#define MV_SRC(i_, j_, k_) src[((size_t)(k_) * h + (j_)) * w + (i_)]
			const float c00 = lerpf(MV_SRC(tx.i0, ty.i0, tz.i0), MV_SRC(tx.i1, ty.i0, tz.i0), tx.f);
			const float c10 = lerpf(MV_SRC(tx.i0, ty.i1, tz.i0), MV_SRC(tx.i1, ty.i1, tz.i0), tx.f);
			const float c01 = lerpf(MV_SRC(tx.i0, ty.i0, tz.i1), MV_SRC(tx.i1, ty.i0, tz.i1), tx.f);
			const float c11 = lerpf(MV_SRC(tx.i0, ty.i1, tz.i1), MV_SRC(tx.i1, ty.i1, tz.i1), tx.f);
#undef MV_SRC
			const float a = lerpf(lerpf(c00, c10, ty.f), lerpf(c01, c11, ty.f), tz.f);
			t.x = one | (one << 16);
			t.y = one | ((uint32_t)float_to_half_bits(a * 0.25f) << 16);
		}
		bricked[volBrickIndex(bx, by, bz, bricks) * 64 + in] = t;
	}
}

// Alpha-cell companion of a bricked volume (mv_device.hpp, VolumeView): cell i holds the eight alphas of the 2x2x2
// footprint that starts at texel i, +1 neighbours clamped to N-1.  Runs after every volume producer.
__global__ __launch_bounds__(256) void buildAlphaCellsKernel(const uint2* __restrict__ bricked, uint4* __restrict__ cells, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	const uint16_t* h = reinterpret_cast<const uint16_t*>(bricked);
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
	{
		const size_t brick = i >> 6;
		const uint32_t in = (uint32_t)(i & 63);
		const uint32_t bx = (uint32_t)(brick % bricks), by = (uint32_t)((brick / bricks) % bricks), bz = (uint32_t)(brick / ((size_t)bricks * bricks));
		const uint32_t x = bx * 4 + (in & 3), y = by * 4 + ((in >> 2) & 3), z = bz * 4 + (in >> 4);
		uint4 c = make_uint4(0u, 0u, 0u, 0u);
		if (x < N && y < N && z < N)
		{
			const uint32_t x1 = x + 1 < N ? x + 1 : N - 1, y1 = y + 1 < N ? y + 1 : N - 1, z1 = z + 1 < N ? z + 1 : N - 1;
#define MV_A(xx_, yy_, zz_) ((uint32_t)h[4 * volTexelIndex(xx_, yy_, zz_, bricks) + 3])
			c.x = MV_A(x, y, z) | (MV_A(x1, y, z) << 16);
			c.y = MV_A(x, y1, z) | (MV_A(x1, y1, z) << 16);
			c.z = MV_A(x, y, z1) | (MV_A(x1, y, z1) << 16);
			c.w = MV_A(x, y1, z1) | (MV_A(x1, y1, z1) << 16);
#undef MV_A
		}
#if MV_ALPHA_QUADS == 2
		reinterpret_cast<uint32_t*>(cells)[volBrickIndex(bx, by, bz, bricks) * 64 + in] = c.x;	// {a000, a100}
#elif MV_ALPHA_QUADS
		reinterpret_cast<uint2*>(cells)[volBrickIndex(bx, by, bz, bricks) * 64 + in] = make_uint2(c.x, c.z);	// {a000,a100 | a001,a101}: the y+1 pair is the next row's quad
#else
		cells[volBrickIndex(bx, by, bz, bricks) * 64 + in] = c;
#endif
	}
}

// brick occupancy (mv_device.hpp): one wavefront per 16^3-texel region, lane = brick of the region, the ballot is the word
__global__ __launch_bounds__(64) void buildOccupancyKernel(const uint2* __restrict__ bricked, uint64_t* __restrict__ words, uint32_t N, uint32_t bricks)
{
	const uint32_t regions = occupancyRegions(bricks);
	const uint32_t lane = threadIdx.x;
	for (uint32_t r = blockIdx.x; r < regions * regions * regions; r += gridDim.x)
	{
		const uint32_t rx = r % regions, ry = (r / regions) % regions, rz = r / (regions * regions);
		const uint32_t bx = rx * 4u + (lane & 3u), by = ry * 4u + ((lane >> 2) & 3u), bz = rz * 4u + (lane >> 4);
		bool occupied = false;
		if (bx < bricks && by < bricks && bz < bricks)
		{
			for (uint32_t dz = 0; dz < 5u; ++dz)
				for (uint32_t dy = 0; dy < 5u; ++dy)
					for (uint32_t dx = 0; dx < 5u; ++dx)
					{
						const uint32_t x = bx * 4u + dx, y = by * 4u + dy, z = bz * 4u + dz;
						if (x < N && y < N && z < N)
							occupied = occupied || half_bits_to_float(bricked[volTexelIndex(x, y, z, bricks)].y >> 16) > kOccupancyLimit;
					}
		}
		const uint64_t word = __ballot(occupied);
		if (lane == 0) words[r] = word;
	}
}

// light map: plain R11G11B10F [z][y][x] (the reference / ABI layout) <-> quad-replicated cells (device layout)
__global__ __launch_bounds__(256) void lightToQuadsKernel(const uint32_t* __restrict__ linear, uint4* __restrict__ quads, uint32_t L)
{
	const uint32_t total = L * L * L;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
	{
		const uint32_t x = i % L, y = (i / L) % L, z = i / (L * L);
		const uint32_t x1 = x + 1 < L ? x + 1 : L - 1, y1 = y + 1 < L ? y + 1 : L - 1;
		const uint32_t* s = linear + (size_t)z * L * L;
		quads[i] = make_uint4(s[y * L + x], s[y * L + x1], s[y1 * L + x], s[y1 * L + x1]);
	}
}

__global__ __launch_bounds__(256) void lightFromQuadsKernel(const uint4* __restrict__ quads, uint32_t* __restrict__ linear, uint32_t L)
{
	const uint32_t total = L * L * L;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) linear[i] = quads[i].x;
}

// The volume CSRayMarchL.hlsl:29-33 relights this frame: visible[frameIdx % visibleCount], else frameIdx % V -- ONE volume per
// frame whatever the number of shards (the lists are in ascending volume order on every shard: the cull is replicated).
// The shard that owns it (v % shardCount == shardIndex) marches its light map; the others do nothing that frame.
__device__ __forceinline__ uint32_t relitVolume(const DeviceState& st, uint32_t frameIdx, uint32_t visibleCount, const uint32_t* visibleVolumes)
{
	return visibleCount ? visibleVolumes[frameIdx % visibleCount] : frameIdx % st.V;
}
__device__ __forceinline__ bool ownsVolume(const DeviceState& st, uint32_t v)
{
	const uint32_t shardCount = st.shardCount ? st.shardCount : 1u;
	return v % shardCount == st.shardIndex;
}

// ---------------------------------------------------------------------------------------------
// a1: volume cull.  ceil(V / 8) single-wavefront workgroups; each culls 8 volumes (8 lanes = 8 cube vertices each).
// The wavefront that finishes last (a ticket in counters[8]) compacts the visible / cube-map lists in ascending volume
// order and builds the march work list (one entry per visible face of every owned cube-map volume, prefix sums of 8x8
// tiles), so that marchViewKernel is sized like the reference's work-graph variant (Sh/LibRayMarch.hlsl:120-121) without a
// host round trip.  Single-wave workgroups on purpose: with frames in flight the cull is dispatched while the previous
// frame's view march fills every CU, and a 512-thread workgroup (the first version) waited 200-600 us for one CU to free
// eight wave slots at once; a lone wave takes the first slot that retires.
// ---------------------------------------------------------------------------------------------
__constant__ uint8_t kEdgeLanes[12][2] = { {0,1},{3,2},{1,3},{2,0},{6,7},{5,4},{4,6},{7,5},{4,0},{2,6},{7,3},{1,5} };	// VolumeCull.hlsli:119-138
__constant__ uint8_t kFaceEdges[6][4] = { {8,3,9,6},{10,2,11,7},{0,8,5,11},{1,10,4,9},{0,2,1,3},{4,6,5,7} };			// VolumeCull.hlsli:213-223

constexpr uint32_t kCullThreads = 64;

// (chained: this frame's cull sits in one launch with the cull of the frame before it -- cullBatchKernel -- and reads what that one leaves)
__device__ __forceinline__ void cullBody(const DeviceState& st, bool chained = false);
__global__ __launch_bounds__(kCullThreads) void cullKernel(DeviceState st) { cullBody(st); }
// The culls of several consecutive frames in one launch (mv_kernels.hpp DeviceStateBatch; the multi-GPU loop with several frames per
// launch): blockIdx.y is the frame.  A frame's cull reads what the cull before it wrote -- the VolumeInfo rows of culled volumes, the
// light-map table it rotates, the view hash, the launch's cull words -- but only its LAST wavefront does, and only in the second half of
// its epilogue: the per-volume work of all frames (projection, masks, LOD), their compaction, the deal and the work lists run side by
// side, and the last wavefront of frame y waits for the last wavefront of frame y - 1 to say it is through (batchInfo[3] = its frame tag)
// right in front of the first thing it needs of it (round 4 made every workgroup of frame y wait at its start: the culls of a group were
// a chain of four whole culls, 108 us; profiles/r05_rank_chain.txt).  A handful of workgroups, dispatched in that order: all resident.
__global__ __launch_bounds__(kCullThreads) void cullBatchKernel(DeviceStateBatch p)
{
	cullBody(p.s[blockIdx.y], blockIdx.y != 0u);
}
// (the wait of a chained cull's last wavefront; all its lanes call it)
__device__ __forceinline__ void cullWaitForPreviousFrame(const DeviceState& st)
{
	const uint32_t want = st.frameTag - 1u;
	if ((threadIdx.x & 63u) == 0u) while (__hip_atomic_load(st.batchInfoPrev + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(2);
	__syncthreads();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ __forceinline__ void cullBody(const DeviceState& st, bool chained)
{
	extern __shared__ uint32_t sInfo[];	// per volume: bit0 visible, bit1 cube-map scheme, bits 2-4 mip, bits 5-10 face mask; then V x 2 words of screen rectangle
	uint32_t* sRect = sInfo + st.V;
	uint32_t* sCost = sRect + 2u * st.V;	// the deal below: estimated march cost of a cube-map volume, 0 once dealt

	// The frame's constants -- CBPerFrame and PerObject[V], what MultiRayCaster::UpdateFrame writes into its upload-heap slice
	// (MultiRayCaster.cpp:316-353), uploaded in front of this kernel -- go into LDS first: every wavefront fetches CBPerFrame and its own
	// eight PerObjects with one batch of coalesced loads and computes from the LDS copy.  (Round 3 also tried fetching them straight
	// from the pinned host slice, without the upload copy: the cull went from 15 to 30-45 us -- a wavefront's ~30 reads of 64 bytes over
	// PCIe are served one after the other -- and the cull heads every frame's chain of kernels; the copy engine moves the 4 KB in 4 us.)
	__shared__ uint32_t sConst[8u * (sizeof(PerObject) / 4u) + 56u];
	const uint32_t lane = threadIdx.x & 63u;
	constexpr uint32_t kPoWords = sizeof(PerObject) / 4u, kCbWords = sizeof(CBPerFrame) / 4u;
	const uint32_t firstVolume = blockIdx.x * (kCullThreads / 8);
	uint32_t partialHash = 0u;
	{
		const uint32_t n = firstVolume < st.V ? (st.V - firstVolume < 8u ? st.V - firstVolume : 8u) : 0u;
		const uint32_t* src = reinterpret_cast<const uint32_t*>(st.perObject) + (size_t)firstVolume * kPoWords;
		uint32_t w[7];
#pragma unroll
		for (uint32_t k = 0; k < 7u; ++k) { const uint32_t i = lane + 64u * k; w[k] = i < n * kPoWords ? src[i] : 0u; }
		const uint32_t c = lane < kCbWords ? reinterpret_cast<const uint32_t*>(st.cbPerFrame)[lane] : 0u;
#pragma unroll
		for (uint32_t k = 0; k < 7u; ++k)
		{
			const uint32_t i = lane + 64u * k;
			sConst[i] = w[k];
			if (i < n * kPoWords)
			{
				// this wavefront's share of the hash of the frame's transforms (see below): word j of PerObject[V] contributes (w + g (j + 25)) m
				partialHash ^= (w[k] + 0x9e3779b9u * ((uint32_t)firstVolume * kPoWords + i + 25u)) * 0x85ebca6bu;
			}
		}
		if (lane < 56u) sConst[8u * kPoWords + lane] = c;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) partialHash ^= __shfl_xor(partialHash, o);
		if (lane == 0 && partialHash) atomicXor(&st.counters[13], partialHash);	// (before this wavefront's ticket; the last one reads the total)
	}
	__syncthreads();
	const PerObject* perObjects = reinterpret_cast<const PerObject*>(sConst) - firstVolume;	// indexed by volume, valid for this wavefront's eight
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(sConst + 8u * kPoWords);
	const float vpW = cb->viewport[0], vpH = cb->viewport[1];
	const vec3 eyePt = v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]);

	const uint32_t li = lane & 7u;			// wTid.x: cube vertex / edge-pair / face index
	const uint32_t baseLane = lane & ~7u;	// 8 * wTid.y
	const uint32_t l6 = li < 6u ? li : 5u;	// lanes 6-7 shadow lane 5 so that every shuffle is executed wave-wide

	{
		const uint32_t volumeId = blockIdx.x * (kCullThreads / 8) + (threadIdx.x >> 3);
		const bool valid = volumeId < st.V;
		const PerObject& po = perObjects[valid ? volumeId : firstVolume];

		// ProjectToViewport (VolumeCull.hlsli:27-41)
		vec4 p;
		p.x = (li & 1u) ? 1.0f : -1.0f;
		p.y = ((li >> 1) & 1u) ? 1.0f : -1.0f;
		p.z = (li >> 2) ? 1.0f : -1.0f;
		p.w = 1.0f;
		p = mul44(p, po.wvp);
		p.x /= p.w; p.y /= p.w; p.z /= p.w;
		p.x = p.x * 0.5f + 0.5f;
		p.y = 0.5f - p.y * 0.5f;	// (as compiled, CSVolumeCull.cso: one rounding where the source's 1 - (y * 0.5 + 0.5) has two; oracle/mv_oracle.cpp mvo_cull)
		const float vx = p.x * vpW, vy = p.y * vpH, vz = p.z;

		// CSVolumeCull.hlsl:32-38
		const bool isInView = valid && (vx <= vpW && vy <= vpH && vx >= 0.0f && vy >= 0.0f) && vz > 0.0f && vz < 1.0f;
		const uint32_t volumeVis = (uint32_t)(__ballot(isInView) >> baseLane) & 0xffu;

		// GenVisibilityMask (VolumeCull.hlsli:46-66)
		const vec3 localSpaceEyePt = mul43(eyePt, po.worldI);
		const float viewComp = comp(localSpaceEyePt, (int)(l6 >> 1));
		const bool faceVis = li < 6u && ((li & 1u) ? viewComp > -1.0f : viewComp < 1.0f);
		uint32_t faceMask = (uint32_t)(__ballot(faceVis) >> baseLane) & 0xffu;

		// GetCubeEdgePairPerLane (VolumeCull.hlsli:164-181): lane li < 6 holds edges 2 li and 2 li + 1
		float ex[2], ey[2];
#pragma unroll
		for (uint32_t i = 0; i < 2; ++i)
		{
			const uint32_t e = 2u * l6 + i;
			const int a = (int)(baseLane + kEdgeLanes[e][0]), b = (int)(baseLane + kEdgeLanes[e][1]);
			const float v0x = __shfl(vx, a), v0y = __shfl(vy, a);
			const float v1x = __shfl(vx, b), v1y = __shfl(vy, b);
			ex[i] = v1x - v0x;
			ey[i] = v1y - v0y;
		}

		// EstimateCubeMaxEdgeLength (VolumeCull.hlsli:248-262)
		const float ms = __builtin_fmaxf(__builtin_sqrtf(ex[0] * ex[0] + ey[0] * ey[0]), __builtin_sqrtf(ex[1] * ex[1] + ey[1] * ey[1]));
		float maxEdge = __shfl(ms, (int)baseLane);
#pragma unroll
		for (uint32_t k = 1; k < 6; ++k) maxEdge = __builtin_fmaxf(maxEdge, __shfl(ms, (int)(baseLane + k)));

		// EstimateProjCoverage (VolumeCull.hlsli:299-322): face li gathers its four edges from the edge-pair lanes
		float fx[4], fy[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; ++k)
		{
			const uint32_t e = kFaceEdges[l6][k];
			const int srcLane = (int)(baseLane + (e >> 1));
			const float x0 = __shfl(ex[0], srcLane), y0 = __shfl(ey[0], srcLane);
			const float x1 = __shfl(ex[1], srcLane), y1 = __shfl(ey[1], srcLane);
			fx[k] = (e & 1u) ? x1 : x0;
			fy[k] = (e & 1u) ? y1 : y0;
		}
		float faceArea = 0.0f;
		if (li < 6u && (faceMask & (1u << li)))
			faceArea = 0.5f * __builtin_fabsf(fx[0] * fy[1] - fy[0] * fx[1]) + 0.5f * __builtin_fabsf(fx[2] * fy[3] - fy[2] * fx[3]);
		float projCov = __shfl(faceArea, (int)baseLane);	// WaveActiveSum, summed in lane order
#pragma unroll
		for (uint32_t k = 1; k < 6; ++k) projCov = projCov + __shfl(faceArea, (int)(baseLane + k));

		// Conservative screen rectangle of the cube for the composite's per-tile volume skip: bounding box of the eight
		// projected vertices, grown by one pixel; the whole viewport if any vertex is at or behind the eye plane.
		float bx0 = vx, bx1 = vx, by0 = vy, by1 = vy, wMin = p.w;
#pragma unroll
		for (int o = 1; o < 8; o <<= 1)
		{
			bx0 = __builtin_fminf(bx0, __shfl_xor(bx0, o)); bx1 = __builtin_fmaxf(bx1, __shfl_xor(bx1, o));
			by0 = __builtin_fminf(by0, __shfl_xor(by0, o)); by1 = __builtin_fmaxf(by1, __shfl_xor(by1, o));
			wMin = __builtin_fminf(wMin, __shfl_xor(wMin, o));
		}

		if (li == 0u && valid)
		{
			// the light direction in this volume's space: the light march needs it for every (voxel, volume) pair
			{
				const vec3 ld = normalize3(mul33(v3(cb->lightPos[0], cb->lightPos[1], cb->lightPos[2]), po.worldI));
				float4 o; o.x = ld.x; o.y = ld.y; o.z = ld.z; o.w = 0.0f;
				st.lightDirs[volumeId] = o;
			}
			ushort4 rect;
			if (wMin > 1e-3f && bx0 == bx0 && by0 == by0)
			{
				rect.x = (uint16_t)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(bx0) - 1.0f, 0.0f), 65535.0f);
				rect.y = (uint16_t)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(by0) - 1.0f, 0.0f), 65535.0f);
				rect.z = (uint16_t)__builtin_fminf(__builtin_fmaxf(__builtin_ceilf(bx1) + 1.0f, 0.0f), 65535.0f);
				rect.w = (uint16_t)__builtin_fminf(__builtin_fmaxf(__builtin_ceilf(by1) + 1.0f, 0.0f), 65535.0f);
			}
			else { rect.x = 0; rect.y = 0; rect.z = 65535; rect.w = 65535; }
			reinterpret_cast<ushort4*>(st.volumeRects)[volumeId] = rect;
			uint32_t info = 0u;
			if (volumeVis != 0u)
			{
				const uint32_t volumeIn = st.volumeDescs[volumeId];
				uint32_t raySampleCount = st.maxRaySamples;

				// EstimateCubeMapLOD (VolumeCull.hlsli:267-294)
				const uint32_t cubeMapSize = volumeIn >> 18;
				const uint32_t numMips = (volumeIn >> 14) & 0xfu;
				float s = maxEdge / 2.0f;
				float raySampleAmt = 2.0f * s * __uint_as_float(0x3f13cd3au);	// (as compiled: times the constant 1 / sqrt(3) folded to fp32, not a division)
				const uint32_t raySampleCnt = f2u_sat(__builtin_ceilf(raySampleAmt));
				raySampleCount = raySampleCnt < raySampleCount ? raySampleCnt : raySampleCount;
				raySampleAmt = __builtin_fminf(raySampleAmt, (float)raySampleCount);
				s = raySampleAmt / 2.0f * kSqrt3;
				const uint32_t level = f2u_sat(__builtin_fmaxf(log2f((float)cubeMapSize / s), 0.0f));
				uint32_t mipLevel = level < numMips - 1u ? level : numMips - 1u;

				// CSVolumeCull.hlsl:62-77
				const uint32_t edgeLength = cubeMapSize >> mipLevel;
				const float cubeMapPix = (float)(edgeLength * edgeLength) * (float)__popc(faceMask);
				bool useCubeMap = cubeMapPix <= projCov;
				if (st.flags & 1u)	// MV_FLAG_FORCE_MIP0_ALL_FACES (stress variant, not a reference mode)
				{
					mipLevel = 0u; faceMask = 0x3fu; useCubeMap = true; raySampleCount = st.maxRaySamples;
				}
				const uint32_t maskBits = useCubeMap ? (faceMask | kCubeMapRayMarchBit) : faceMask;
				const uint32_t volTexId = volumeIn & 0x3fffu;

				ushort4 out;
				out.x = (uint16_t)mipLevel; out.y = (uint16_t)raySampleCount; out.z = (uint16_t)maskBits; out.w = (uint16_t)volTexId;
				reinterpret_cast<ushort4*>(st.volumeInfo)[volumeId] = out;
				info = 1u | (useCubeMap ? 2u : 0u) | (mipLevel << 2) | ((faceMask & 0x3fu) << 5);
			}
			else if (!chained) reinterpret_cast<ushort4*>(st.volumeInfo)[volumeId] = reinterpret_cast<const ushort4*>(st.volumeInfoPrev)[volumeId];	// CSVolumeCull.hlsl:38 returns without writing
			// (chained: the row the frame before leaves there is not final yet -- the last wavefront copies it, below, behind its wait)
			st.volumeFlags[volumeId] = info;
		}
	}

	// last wavefront to get here builds the lists from everybody's flags and rectangles
	__threadfence();
	uint32_t ticket = 0;
	if (lane == 0) ticket = atomicAdd(&st.counters[8], 1u);
	ticket = __shfl(ticket, 0);
	if (ticket != gridDim.x - 1u) return;
	__threadfence();
	for (uint32_t v = lane; v < st.V; v += 64)
	{
		sInfo[v] = __hip_atomic_load(st.volumeFlags + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t* rects = reinterpret_cast<const uint32_t*>(st.volumeRects);
		sRect[2 * v] = __hip_atomic_load(rects + 2 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);		// x0 | y0 << 16
		sRect[2 * v + 1] = __hip_atomic_load(rects + 2 * v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);	// x1 | y1 << 16
	}
	__syncthreads();

	// ---- who marches which cube-map volume (sharded contexts): volumeFlags bits 16-31 = the owner ----
	// Default v % shardCount.  In the multi-GPU frame loop (kFlagBalanceOwners) the volumes of THIS frame are dealt by estimated
	// march cost -- marched faces x 8x8 tiles per face x SmpCount, all from the replicated VolumeInfo, so every rank makes the same
	// deal: longest first, each to the rank with the least so far (ties: lowest volume, lowest rank).  With v % shardCount the
	// slowest rank of C5 on 8 ranks carries 2.2x the mean (the volumes nearest the camera share a residue).  Lane r keeps rank r's
	// load; up to 64 ranks, else the default.  Also the layout of the frame's cube-face exchange: exchangeOff[v] = texels of the
	// owner's volumes below v, counters[12] = texels of the largest shard.
	const uint32_t shardCount = st.shardCount ? st.shardCount : 1u;
	if (shardCount > 1u)
	{
		const bool balance = (st.flags & kFlagBalanceOwners) && shardCount <= 64u && !(st.flags & kFlagDealTiles);	// (tiles dealt: every rank marches its tiles of every volume, nothing to balance)
		uint32_t nDeal = 0;
		for (uint32_t v = lane; v < st.V; v += 64)
		{
			const uint32_t info = sInfo[v] & 0xffffu;
			uint32_t cost = 0;
			if ((info & 3u) == 3u)
			{
				const uint32_t dims = cubeMipDim(st.N, (info >> 2) & 7u), tiles = ((dims + 7u) >> 3) * ((dims + 7u) >> 3);
				const unsigned long long infoBits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(st.volumeInfo) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				const uint32_t smp = (uint32_t)(infoBits >> 16) & 0xffffu;
				const unsigned long long c = (unsigned long long)__popc((info >> 5) & 0x3fu) * tiles * (smp ? smp : 1u);
				cost = c < 0xfffffffeull ? (uint32_t)c + 1u : 0xffffffffu;	// > 0 for every cube-map volume
			}
			sCost[v] = balance ? cost : 0u;
			sInfo[v] = info | ((v % shardCount) << 16);
			nDeal += cost ? 1u : 0u;
		}
		nDeal = waveSumU32(nDeal);
		__syncthreads();
		unsigned long long load = 0;
		if (balance && st.V <= 64u)
		{
			// Up to 64 volumes (every scene of the reference): volume v's cost lives in lane v, rank r's load in lane r.  The greedy deal
			// takes the volumes in the order (cost descending, v ascending) -- so every lane first finds its volume's POSITION in that order
			// by counting the volumes in front of it (one v_readlane per volume, no branches), a ds_permute puts the costs in that order,
			// and the loop that is sequential by nature only has left: the least loaded rank, by a DPP min-reduction of (load << 6 | rank).
			// ~0.1 us per volume dealt.  (Round 2's version searched the maximum and the minimum with scalar loops over v_readlane in
			// every step: 1.4 us per volume -- 14 of the cull's 30 us on a rank at C3, several times that at C5's 64 volumes.)
			const uint32_t myCost = lane < st.V ? sCost[lane] : 0u;
			uint32_t pos = 0;
			for (uint32_t j0 = 0; j0 < st.V; j0 += 8u)
#pragma unroll
				for (uint32_t jj = 0; jj < 8u; ++jj)
				{
					const uint32_t j = j0 + jj;	// (lanes beyond V hold cost 0: they are in front of nobody with a cost)
					const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)myCost, (int)j);
					pos += (cj > myCost || (cj == myCost && j < lane)) ? 1u : 0u;
				}
			if (lane >= st.V) pos = lane;	// (a permutation of all 64 lanes)
			const uint32_t sorted = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos << 2), (int)myCost);	// lane k: the k-th volume's cost
			unsigned long long load = 0;
			uint32_t ownerAt = 0;	// lane k: the rank the k-th volume went to
			for (uint32_t k = 0; k < nDeal; ++k)
			{
				const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)sorted, (int)k);
				const unsigned long long key = waveMinU64(lane < shardCount ? (load << 6) | lane : ~0ull);	// ties: the lowest rank
				const uint32_t minRank = (uint32_t)key & 63u;
				if (lane == minRank) load += c;
				if (lane == k) ownerAt = minRank;
			}
			const uint32_t myOwner = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos << 2), (int)ownerAt);
			if (lane < st.V && myCost) sInfo[lane] = (sInfo[lane] & 0xffffu) | (myOwner << 16);	// (the others keep v % shardCount)
			__syncthreads();
		}
		else
		for (uint32_t it = 0; balance && it < nDeal; ++it)
		{
			// the most expensive volume not dealt yet
			uint32_t bestCost = 0, bestV = 0xffffffffu;
			for (uint32_t v = lane; v < st.V; v += 64)
			{
				const uint32_t c = sCost[v];
				if (c > bestCost) { bestCost = c; bestV = v; }	// (ascending v: the first of equals stays)
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
			{
				const uint32_t oc = __shfl_xor(bestCost, o), ov = __shfl_xor(bestV, o);
				if (oc > bestCost || (oc == bestCost && ov < bestV)) { bestCost = oc; bestV = ov; }
			}
			// the rank with the least so far
			unsigned long long minLoad = lane < shardCount ? load : ~0ull;
			uint32_t minRank = lane;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
			{
				const unsigned long long ol = __shfl_xor(minLoad, o);
				const uint32_t orank = __shfl_xor(minRank, o);
				if (ol < minLoad || (ol == minLoad && orank < minRank)) { minLoad = ol; minRank = orank; }
			}
			if (lane == minRank) load += bestCost;
			if (lane == 0) { sCost[bestV] = 0u; sInfo[bestV] = (sInfo[bestV] & 0xffffu) | (minRank << 16); }
			__syncthreads();
		}
		// exchange layout: lane r walks rank r's volumes in ascending order (64 ranks at a time)
		uint32_t maxShard = 0;
		for (uint32_t r0 = 0; r0 < shardCount; r0 += 64)
		{
			uint32_t run = 0;
			for (uint32_t v = 0; v < st.V; ++v)
			{
				const uint32_t info = sInfo[v];
				if ((info & 3u) != 3u || (info >> 16) != r0 + lane) continue;
				const uint32_t d = cubeMipDim(st.N, (info >> 2) & 7u);
				st.exchangeOff[v] = run;
				run += (uint32_t)__popc((info >> 5) & 0x3fu) * d * d;
			}
			maxShard = run > maxShard ? run : maxShard;
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { const uint32_t m = __shfl_xor(maxShard, o); maxShard = m > maxShard ? m : maxShard; }
		if (lane == 0)
		{
			st.counters[12] = maxShard;
			// host-visible {frame tag : texels of the largest shard}: the size of this frame's cube-face all-gather, read by the host a call
			// later (mv_dist_frame) from pinned memory -- no copy behind the cull, no event
			if (st.exchangeReport) *st.exchangeReport = ((unsigned long long)st.frameTag << 32) | maxShard;
		}
		for (uint32_t v = lane; v < st.V; v += 64) st.volumeFlags[v] = sInfo[v];
	}

	// ---- list compaction + work list, first wavefront ----
	const uint64_t chain = cubeChainTexels(st.N);
	const uint64_t lower = (1ull << lane) - 1ull;
	uint32_t nVis = 0, nCube = 0, nOwned = 0, nEntries = 0, nTiles = 0, minMip = 0xffffffffu;
	for (uint32_t base = 0; base < st.V; base += 64)
	{
		const uint32_t v = base + lane;
		const uint32_t info = v < st.V ? sInfo[v] : 0u;
		const bool vis = info & 1u, cube = info & 2u;
		const bool owned = (v % shardCount) == st.shardIndex;	// (the light march's and the whole-cube exchange's notion: fixed)
		const bool dealTiles = (st.flags & kFlagDealTiles) && shardCount > 1u;
		const bool marches = (shardCount > 1u && !dealTiles) ? (info >> 16) == st.shardIndex : true;	// this frame's view march of a cube-map volume
		const uint32_t mip = (info >> 2) & 7u, mask = (info >> 5) & 0x3fu;
		const uint64_t bVis = __ballot(vis), bCube = __ballot(cube), bOwn = __ballot(vis && owned);
		if (vis)
		{
			st.visibleVolumes[nVis + __popcll(bVis & lower)] = v;
			st.cubeOffset[v] = (uint64_t)v * chain + cubeMipOffset(st.N, mip);
			minMip = mip < minMip ? mip : minMip;
		}
		if (cube) st.cubeMapVolumes[nCube + __popcll(bCube & lower)] = v;
		if (vis && owned) st.ownedVisible[nOwned + __popcll(bOwn & lower)] = v;
		nVis += __popcll(bVis); nCube += __popcll(bCube); nOwned += __popcll(bOwn);

		const uint32_t dims = cubeMipDim(st.N, mip);
		const uint32_t tilesOfFace = ((dims + 7u) >> 3) * ((dims + 7u) >> 3);
		const uint32_t tilesPerFace = dealTiles ? dealtTilesOfFace(tilesOfFace, (dims + 7u) >> 3, st.shardIndex, shardCount, v) : tilesOfFace;	// (this rank's)
		const uint32_t myFaces = (cube && marches && tilesPerFace) ? (uint32_t)__popc(mask) : 0u;
		uint32_t totFaces, totTiles;
		const uint32_t eBase = nEntries + waveExclusiveScanU32(myFaces, lane, totFaces);
		const uint32_t tBase = nTiles + waveExclusiveScanU32(myFaces * tilesPerFace, lane, totTiles);
		uint32_t k = 0;
		for (uint32_t f = 0; f < 6; ++f)
			if (myFaces && (mask & (1u << f)))
			{
				st.workEntries[eBase + k] = v | (f << 16);
				st.workTileStart[eBase + k] = tBase + k * tilesPerFace;
				++k;
			}
		nEntries += totFaces; nTiles += totTiles;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t m = __shfl_xor(minMip, o); minMip = m < minMip ? m : minMip; }

	if (chained)
	{
		// From here on the epilogue reads what the cull of the frame before left: its light table, its view hash, its cull words -- and the
		// VolumeInfo rows of the volumes THIS frame culls, which keep what they held (CSVolumeCull.hlsl:38 returns without writing)
		cullWaitForPreviousFrame(st);
		for (uint32_t v = lane; v < st.V; v += 64)
			if (!(sInfo[v] & 1u)) reinterpret_cast<ushort4*>(st.volumeInfo)[v] = reinterpret_cast<const ushort4*>(st.volumeInfoPrev)[v];
		__threadfence();	// (other lanes of this wavefront read these rows below: VolumeConsts)
	}
	// This frame's light-map table = last frame's, with the volume this shard is about to relight switched to its next buffer.
	uint32_t relitOfFrame = 0xffffffffu;	// (the volume CSRayMarchL relights this frame, whoever owns it)
	{
		uint32_t relit;
		if (nVis)
		{
			// the (frameIdx % nVis)-th visible volume, found in the flags (the list itself was written by other lanes)
			const uint32_t want = cb->frameIdx % nVis;
			uint32_t seen = 0; relit = 0xffffffffu;
			for (uint32_t base = 0; base < st.V; base += 64)
			{
				const uint32_t v = base + lane;
				const bool ov = v < st.V && (sInfo[v] & 1u);
				const uint64_t b = __ballot(ov);
				const uint32_t rank = seen + (uint32_t)__popcll(b & lower);
				const uint64_t hit = __ballot(ov && rank == want);
				if (hit) { relit = base + (uint32_t)__ffsll((unsigned long long)hit) - 1u; break; }
				seen += (uint32_t)__popcll(b);
			}
		}
		else relit = relitVolume(st, cb->frameIdx, 0u, nullptr);
		// another shard's: nothing rotates here -- unless the context keeps every light map current through mv_dist_frame's exchange,
		// which fills the fresh buffer on the ranks that do not own the volume
		relitOfFrame = relit;
		if (!ownsVolume(st, relit) && !(st.flags & kFlagRotateAllLightMaps)) relit = 0xffffffffu;
		const size_t L3 = (size_t)st.L * st.L * st.L;
		for (uint32_t v = lane; v < st.V; v += 64)
		{
			uint4* p = st.lightTablePrev[v];
			if (v == relit)
			{
				// next of the volume's lightBufs buffers, in rotation
				uint4* first = st.lightBuffers + (size_t)(st.lightBufs * v) * L3;
				const uint32_t k = (uint32_t)((size_t)(p - first) / L3);
				p = first + (size_t)((k + 1u) % st.lightBufs) * L3;
			}
			st.lightTable[v] = p;

			// what a ray of volume v needs besides its record (mv_march.hip): where its texels, its light map (this frame's
			// buffer) and its occupancy words are, SmpCount and the step (CSRayMarch.hlsl:109 g_maxDist / SmpCount)
			const unsigned long long infoBits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(st.volumeInfo) + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t smpCount = (uint32_t)(infoBits >> 16) & 0xffffu, volTexId = st.volumeDescs[v] & 0x3fffu;
			VolumeConsts vc;
			vc.texelOff = (uint64_t)volTexId * st.volStride;
			vc.lightOff = 0u;	// (the ray setup takes it from lightTable)
			vc.occOff = volTexId * st.occStride;
			vc.smpCount = smpCount;
			vc.stepScale = kMaxDist / (float)smpCount;
			vc.pad[0] = vc.pad[1] = 0u;
			st.volConsts[v] = vc;
		}
	}

	// Screen-space-scheme volumes (visible, cubeMapPix > projCov): their RayCast (RayCast.hlsli:43-108) is a march like
	// any other, so it joins the same launch as extra 8x8 PIXEL tiles over the volume's screen rectangle and leaves its
	// result in an fp32 layer the composite reads.  One slot per volume, up to st.screenSlots; the rest stay inline.
	const uint32_t nCubeTiles = nTiles;
	uint32_t nScreen = 0;
	for (uint32_t base = 0; base < st.V; base += 64)
	{
		const uint32_t v = base + lane;
		const uint32_t info = v < st.V ? sInfo[v] : 0u;
		const bool scr = (info & 1u) && !(info & 2u);
		const uint64_t bScr = __ballot(scr);
		const uint32_t slot = nScreen + (uint32_t)__popcll(bScr & lower);
		const bool take = scr && slot < st.screenSlots;
		if (v < st.V) st.screenSlot[v] = take ? slot : 0xffffffffu;
		uint32_t tiles = 0;
		if (take)
		{
			const uint32_t r0 = sRect[2 * v], r1 = sRect[2 * v + 1];
			const uint32_t x0 = r0 & 0xffffu, y0 = r0 >> 16;
			uint32_t x1 = r1 & 0xffffu, y1 = r1 >> 16;
			x1 = x1 < st.W - 1u ? x1 : st.W - 1u; y1 = y1 < st.H - 1u ? y1 : st.H - 1u;
			if (x0 <= x1 && y0 <= y1) tiles = ((x1 >> 3) - (x0 >> 3) + 1u) * screenTileRows(st, y0, y1).count;	// this shard's tile rows only
		}
		uint32_t totE, totT;
		const uint32_t eBase = nEntries + waveExclusiveScanU32(take ? 1u : 0u, lane, totE);
		const uint32_t tBase = nTiles + waveExclusiveScanU32(tiles, lane, totT);
		if (take)
		{
			st.workEntries[eBase] = v | (7u << 16);
			st.workTileStart[eBase] = tBase;
		}
		nEntries += totE; nTiles += totT; nScreen += (uint32_t)__popcll(bScr);
	}
	// what the tile costs of a frame depend on besides the volumes' contents: CBPerFrame's eye / viewport / screenToWorld and every PerObject
	// (the PerObject words were hashed by the wavefronts that fetched them; XOR of per-word terms: the order does not matter)
	uint32_t viewHash = 0u;
	{
		const uint32_t* cbWords = sConst + 8u * kPoWords;
		if (lane < 24u) viewHash = (cbWords[lane] + 0x9e3779b9u * (lane + 1u)) * 0x85ebca6bu;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) viewHash ^= __shfl_xor(viewHash, o);
		viewHash ^= __hip_atomic_load(&st.counters[13], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		viewHash |= 1u;	// never 0, the value of a slot no frame has used yet
	}
	if (lane == 0)
	{
		st.workTileStart[nEntries] = nTiles;
		st.counters[0] = nVis; st.counters[1] = nCube; st.counters[2] = nEntries; st.counters[3] = nTiles;
		st.counters[4] = nOwned; st.counters[5] = minMip == 0xffffffffu ? 0u : minMip; st.counters[6] = nCubeTiles;
		// stride of the dispatch permutation slot -> tile = slot * stride mod nTiles: about nTiles / golden ratio, coprime with nTiles
		uint32_t stride = (uint32_t)(((uint64_t)nTiles * 2654435769ull) >> 32) | 1u;
		for (;;)
		{
			uint32_t a = stride, b = nTiles;
			while (b) { const uint32_t r = a % b; a = b; b = r; }
			if (a == 1u || nTiles < 2u) break;
			stride += 2u;
		}
		st.counters[7] = nTiles < 2u ? 1u : stride;
		st.counters[8] = 0u;	// the ticket, for the next cull that uses this slot
		st.counters[13] = 0u;	// and the hash accumulator
		if (st.binCursor) { for (uint32_t b = 0; b < kOrderBins; ++b) st.binCursor[b] = 0u; }	// room taken in every bin of the tile order (ordersKernel follows)
		// [10] a hash of this frame's camera and volume transforms, [11] "it differs from the previous frame's": the tile costs this
		// frame's order reads were taken one or two frames ago (tileBinOf dilates them then)
		st.counters[11] = st.countersPrev[10] != viewHash ? 1u : 0u;
		st.counters[10] = viewHash;
		// host-visible {frame tag : tile count}: a hint for the grid sizes of later launches; the host never waits for it (mv_api.cpp)
		// (system-scope stores: written through to the host at once, whatever release the runtime gives this kernel's end)
		if (st.tilesReport) __hip_atomic_store(st.tilesReport, ((unsigned long long)st.frameTag << 32) | nTiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		if (st.tilesReportSlot) __hip_atomic_store(st.tilesReportSlot, ((unsigned long long)st.frameTag << 32) | nTiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		// What the ranks of the multi-GPU loop compare of their culls ([0..2]: hash, visible, cube-map volumes) and which volumes the
		// frames of this launch relight ([4 + position]) -- carried from frame to frame of a launch that holds several (batchPos > 0:
		// the previous frame's words are folded in, position-weighted), so that the launch's LAST frame holds the words for all of them
		// and the march needs no other frame's state (mv_kernels.hpp DeviceStateBatch).
		if (st.batchInfo)
		{
			const uint32_t pos = st.batchPos, h = st.counters[12];
			const bool carry = pos != 0u && st.batchInfoPrev;
			st.batchInfo[0] = (carry ? st.batchInfoPrev[0] : 0u) + (pos ? (h << (7u * pos)) | (h >> (32u - 7u * pos)) : h);
			st.batchInfo[1] = (carry ? st.batchInfoPrev[1] : 0u) + nVis * (2u * pos + 1u);
			st.batchInfo[2] = (carry ? st.batchInfoPrev[2] : 0u) + nCube * (2u * pos + 1u);
			for (uint32_t j = 0; j < kMaxBatch; ++j) st.batchInfo[4u + j] = j == pos ? relitOfFrame : (carry && j < pos ? st.batchInfoPrev[4u + j] : 0xffffffffu);
		}
	}
	// "this frame's cull is through" -- for the next frame's workgroups when they sit in the same launch (cullBatchKernel)
	if (st.batchInfo)
	{
		__threadfence();
		if (lane == 0) __hip_atomic_store(st.batchInfo + 3, st.frameTag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// ---------------------------------------------------------------------------------------------
// Sh/RayMarch.hlsli:197-230  CastLightRay (mipLevel 0)
// ---------------------------------------------------------------------------------------------
#ifndef MV_LIGHT_SPECULATE
#define MV_LIGHT_SPECULATE 1	// alpha footprints in flight per light ray (0 = the plain loop)
#endif
// the footprint of a density sample: through the per-axis address tables in LDS when the kernel has them (mv_device.hpp)
template <uint32_t TABLE_PITCH>
__device__ __forceinline__ FootprintAddr densityFootprint(const VolumeView& vol, const u32x4* axisTables, vec3 uvw)
{
	if (!axisTables) return footprintAddr(vol, uvw);
	FootprintAddr a;
	const u32x4 ex = axisTables[axisTableIndex(uvw.x, vol.N, a.fx)];
	const u32x4 ey = axisTables[TABLE_PITCH + axisTableIndex(uvw.y, vol.N, a.fy)];
	const u32x4 ez = axisTables[2u * TABLE_PITCH + axisTableIndex(uvw.z, vol.N, a.fz)];
	a.x0 = ex.x; a.x1 = ex.y;
	a.r00 = ey.x + ez.x; a.r10 = ey.y + ez.x; a.r01 = ey.x + ez.y; a.r11 = ey.y + ez.y;
	a.cx = __uint_as_float(__float_as_uint(a.fx) & ex.w);	// weight 0 where both taps are the same texel (the 16-byte and pair forms of the alpha companion)
	a.cy = __uint_as_float(__float_as_uint(a.fy) & ey.w);
	a.cz = __uint_as_float(__float_as_uint(a.fz) & ez.w);
	a.mx = ex.w; a.mz = ez.w;	// (the quad form -- the product's -- selects the taps instead and keeps the true weights: mv_device.hpp resolveAlpha)
	a.ix = a.iy = a.iz = 0u;	// (not needed by a cell fetch)
	return a;
}

#ifndef MV_LIGHT_WAVES
#define MV_LIGHT_WAVES 4	// voxel blocks (wavefronts) per light-march workgroup: they share the per-axis address tables in LDS
#endif
constexpr uint32_t kLightLutMaxN = 256, kLightWaves = MV_LIGHT_WAVES;

template <int K>
__device__ __forceinline__ void CastLightRay(float& transm, const VolumeView& vol, vec3 rayOrigin, vec3 rayDir,
	float stepScale, uint32_t numSamples, uint32_t& samples, const u32x4* axisTables = nullptr)
{
	float t = stepScale;
	float step = stepScale;
	float prevDensity = 0.0f;
	if constexpr (K > 0)
	{
	// The loop in batches of MV_LIGHT_SPECULATE samples whose alpha footprints are requested together, at t, t + stepScale,
	// t + 2 stepScale ... (the very additions the loop performs while the step stays stepScale; GetStep returns exactly
	// stepScale until the light is noticeably attenuated, 1.5 * factorEv * factorUi * (1 - transm) <= 1).  A wrong guess
	// discards the rest of the batch and starts a new one at the real t; coordinates beyond the box are clamped by the
	// sampler, so every address is valid.  Arithmetic and order are the reference's.
	// Measured at C3 (8.4 M light samples): batch 1 -> 0.128 ms, 2 -> 0.136, 3 -> 0.144, 4 -> 0.159, 8 -> 0.200: the launch
	// is bound by request throughput plus one ~0.055 ms longest-ray tail (scripts/exp_lightscale.py), not by the latency of
	// a ray's dependent samples, so requests that may be discarded only cost.  Hence 1 -- for a launch that fills the GPU.  A rank of the
	// multi-GPU loop marches 1 / shard_count of the voxel blocks, a wavefront or two per SIMD, where the launch lasts as long as its
	// longest ray's chain of dependent samples: a batch of 4 was tried there too (K is a template parameter of the kernel, chosen per
	// launch by MV_LIGHT_SPECULATE_BELOW) and is slower as well -- the chain is the ~190 dependent VALU instructions of an iteration.
	if (numSamples == 0u || outsideBox(rayAt(rayOrigin, rayDir, t))) return;
	for (uint32_t i = 0;;)
	{
		AlphaFootprint batch[K];
		{
			float tj = t;
#pragma unroll
			for (int j = 0; j < K; ++j)
			{
				batch[j] = fetchAlphaCell(vol, densityFootprint<kLightLutMaxN + 1u>(vol, axisTables, LocalToTex3DSpace(rayAt(rayOrigin, rayDir, tj))));
				tj += stepScale;
			}
		}
		bool restart = false;
#pragma unroll
		for (int j = 0; j < K; ++j)
		{
			if (restart) break;
			const float density = resolveAlpha(batch[j]);
			++samples;

			const float dDensity = density - prevDensity;
			const float opacity = saturatef(density * step);
			const float newStep = GetStep(dDensity, transm, opacity, stepScale);
			prevDensity = density;	// (:219 `step = (step + newStep) * 0.5` is dead: overwritten at :227)

			transm *= 1.0f - density * kAbsorption;
			if (transm < kZeroThreshold) return;

			step = newStep;
			t += step;
			if (++i >= numSamples) return;
			if (outsideBox(rayAt(rayOrigin, rayDir, t))) return;
			restart = newStep != stepScale;
		}
	}
	}
	else
	{
	for (uint32_t i = 0; i < numSamples; ++i)
	{
		const vec3 pos = rayAt(rayOrigin, rayDir, t);
		if (outsideBox(pos)) break;
		const vec3 uvw = LocalToTex3DSpace(pos);

		const float density = sampleAlpha16F(vol, uvw);
		++samples;

		const float dDensity = density - prevDensity;
		const float opacity = saturatef(density * step);
		const float newStep = GetStep(dDensity, transm, opacity, stepScale);
		prevDensity = density;	// (:219 `step = (step + newStep) * 0.5` is dead: overwritten at :227)

		transm *= 1.0f - density * kAbsorption;
		if (transm < kZeroThreshold) break;

		step = newStep;
		t += step;
	}
	}
}

// XUSG/Shaders/SHIrradianceTypeless.hlsli:15-36
__device__ __forceinline__ vec3 EvaluateSHIrradiance(const float* sh, vec3 norm)
{
	// XUSG/Shaders/SHIrradiance.hlsli as the reference's compiler emitted it inside CSRayMarchL.cso: constants folded to fp32
	// (3 c3, 2 c1, 2 c2 as single constants), the sum in this order
	const float c1 = 0.4290427565574646f, c3x3 = 0.7431238889694214f, c3 = 0.24770796298980713f, c4 = 0.8862269520759583f;
	const float c1x2 = 0.8580855131149292f, c2x2 = 1.0233267545700073f;
	const float x = -norm.x, y = -norm.y, z = norm.z;
	float r[3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch)
	{
#define MV_SHC(i) sh[3 * (i) + ch]
		const float t8 = (norm.x * norm.x - norm.y * norm.y) * c1 * MV_SHC(8);
		const float t6 = (z * z * c3x3 + -c3) * MV_SHC(6);
		const float t0 = MV_SHC(0) * c4;
		const float q = ((z * MV_SHC(5) + MV_SHC(4) * x) * y + MV_SHC(7) * x * z) * c1x2;
		const float l = (MV_SHC(3) * x + MV_SHC(1) * y + z * MV_SHC(2)) * c2x2;
		const float v = t6 + t0 + t8 + l + q;
#undef MV_SHC
		r[ch] = __builtin_fmaxf(0.0f, v);
	}
	return v3(r[0], r[1], r[2]);
}

// ---------------------------------------------------------------------------------------------
// a2: light-space march.  One wavefront = one 4x4x4 block of light-map voxels (the reference's thread group).
// ---------------------------------------------------------------------------------------------
#ifndef MV_LIGHT_MIN_WAVES
#define MV_LIGHT_MIN_WAVES 4	// measured: 8 -> 0.150 ms, 6/4 -> 0.144 ms
#endif
// kLightWaves voxel blocks per workgroup: they share the per-axis address tables of the volumes in LDS (LUT, N <= 256; every
// volume source has the same N and pitches).
// Which voxel blocks a context marches: all of them -- or, inside the multi-GPU frame loop (kFlagRotateAllLightMaps: every rank keeps
// every light map), its 1 / shardCount share of the dispatch order, entries [lo, hi).  Every rank holds every volume, so any rank can
// march any block; the shares are all-gathered as plain R11G11B10F words and unpacked by lightUnpackKernel below.  (Round 2 let the
// relit volume's owner march the whole map while the other ranks waited for it, and all-reduced it.)
struct LightShare { uint32_t lo, hi, per; };
__host__ __device__ inline LightShare lightShare(const DeviceState& st, uint32_t totalBlocks)
{
	LightShare r;
	if ((st.flags & kFlagLightLayers) && st.shardCount > 1u)
	{
		// peer-store transport: whole layers of voxel blocks, layer l of the z axis to rank l % shardCount (entry e of a rank's share is
		// block lightLayerBlock(e)); `per` is the largest share, the grid every rank launches
		uint32_t B = 1u;
		while (B * B * B < totalBlocks) ++B;
		const uint32_t mine = lightLayersOfRank(B, st.shardIndex, st.shardCount);
		r.per = ((B + st.shardCount - 1u) / st.shardCount) * B * B; r.lo = 0u; r.hi = mine * B * B;
		return r;
	}
	const bool split = (st.flags & kFlagRotateAllLightMaps) && st.shardCount > 1u;
	r.per = split ? (totalBlocks + st.shardCount - 1u) / st.shardCount : totalBlocks;
	r.lo = split ? st.shardIndex * r.per : 0u;
	r.lo = r.lo < totalBlocks ? r.lo : totalBlocks;
	r.hi = r.lo + r.per < totalBlocks ? r.lo + r.per : totalBlocks;
	return r;
}

// quad-replicated store of one light-map texel (see mv_device.hpp "light maps"): texel (x, y) is slot (sx, sy) of every cell (cx, cy)
// with min(cx + sx, L-1) == x and min(cy + sy, L-1) == y
__device__ __forceinline__ void storeLightTexel(uint32_t* quads, uint32_t L, uint32_t xx, uint32_t yy, uint32_t zz, uint32_t packed)
{
	const uint32_t cxs[3] = { xx, xx - 1u, xx }, sxs[3] = { 0u, 1u, 1u };
	const bool vx[3] = { true, xx >= 1u, xx == L - 1u };
	const uint32_t cys[3] = { yy, yy - 1u, yy };
	const bool vy[3] = { true, yy >= 1u, yy == L - 1u };
#pragma unroll
	for (int b = 0; b < 3; ++b)
#pragma unroll
		for (int a = 0; a < 3; ++a)
			if (vx[a] && vy[b])
				quads[((((size_t)zz * L + cys[b]) * L + cxs[a]) << 2) + sxs[a] + 2u * sxs[b]] = packed;
}

// wordsOut: NULL, or where this context's share of the map goes as plain words, 64 per block in dispatch order from entry `lo` on
template <bool LUT, bool ORDERED, int K, bool PEER>
__device__ __forceinline__ void marchLightBody(const DeviceState& st, uint32_t* __restrict__ wordsOut, u32x4* sAxis, unsigned long long (*sCand)[(1024u + 63u) / 64u]);

// PEER: the peer-store transport's instantiation with a folded wait (mv_kernels.hpp peerFoldWait); the others are untouched by it
template <bool LUT, bool ORDERED, int K, bool PEER = false>
__global__ __launch_bounds__(64 * kLightWaves, MV_LIGHT_MIN_WAVES) void marchLightKernel(DeviceState st, uint32_t* __restrict__ wordsOut)
{
	__shared__ u32x4 sAxis[LUT ? 3 * (kLightLutMaxN + 1) : 1];
	__shared__ unsigned long long sCand[kLightWaves][(1024u + 63u) / 64u];	// (MV_LIGHT_CANDIDATES: volumes beyond the first 64; MV_MAX_VOLUMES = 1024)
	// (peer-store transport, folded wait: until every rank is done with the frames that read what this launch -- and the push behind it -- overwrite)
	if constexpr (PEER) peerFoldWait(st);
	marchLightBody<LUT, ORDERED, K, PEER>(st, wordsOut, sAxis, sCand);
}

// several frames in one launch (mv_kernels.hpp DeviceStateBatch; peer-store transport, folded wait): blockIdx.y is the frame -- each
// relights its own volume into its own freshly rotated buffer
template <bool LUT, bool ORDERED, int K>
__global__ __launch_bounds__(64 * kLightWaves, MV_LIGHT_MIN_WAVES) void marchLightBatchKernel(DeviceStateBatch p)
{
	__shared__ u32x4 sAxis[LUT ? 3 * (kLightLutMaxN + 1) : 1];
	__shared__ unsigned long long sCand[kLightWaves][(1024u + 63u) / 64u];
	const DeviceState& st = p.s[blockIdx.y];
	peerFoldWait(st);
	marchLightBody<LUT, ORDERED, K, true>(st, nullptr, sAxis, sCand);
}

template <bool LUT, bool ORDERED, int K, bool PEER>
__device__ __forceinline__ void marchLightBody(const DeviceState& st, uint32_t* __restrict__ wordsOut, u32x4* sAxis, unsigned long long (*sCand)[(1024u + 63u) / 64u])
{
	const PerObject* perObjects = reinterpret_cast<const PerObject*>(st.perObject);
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);

	// CSRayMarchL.hlsl:29-33; on a sharded context only the owner of that volume marches -- or every rank its share of the blocks
	uint32_t volumeId = relitVolume(st, cb->frameIdx, st.counters[0], st.visibleVolumes);
	volumeId = __builtin_amdgcn_readfirstlane(volumeId);
	if (!(st.flags & kFlagRotateAllLightMaps) && !ownsVolume(st, volumeId)) return;
	// Behind a rank's share of the all-gather: what its replicated cull decided for this frame.  Every rank's cull must have decided the
	// same (same inputs, same kernel) -- the sizes of the frame's collectives follow from it -- so lightUnpackKernel compares the ranks'
	// words and raises a flag the host reads before it sizes the frame's cube-face all-gather (a divergence, i.e. callers that passed
	// different frame constants on different ranks, is then an error instead of a hang).
	if (wordsOut && blockIdx.x == 0 && threadIdx.x < 4u)
	{
		const uint32_t totalB = ((st.L + 3u) >> 2) * ((st.L + 3u) >> 2) * ((st.L + 3u) >> 2);
		const uint32_t word = threadIdx.x == 0 ? st.frameTag : threadIdx.x == 1 ? st.counters[12] : threadIdx.x == 2 ? st.counters[0] : st.counters[1];
		wordsOut[(size_t)lightShare(st, totalB).per * 64u + threadIdx.x] = word;
	}

	const u32x4* axisTables = nullptr;
	if (LUT)
	{
		const uint32_t P = st.N + 1u, regions = occupancyRegions(st.bricks);
		const uint32_t strideY = brickPitchY(st.bricks) * 512u, strideZ = brickPitchZ(st.bricks) * 512u;
		for (uint32_t k = threadIdx.x; k < 3u * P; k += 64u * kLightWaves)
			sAxis[(k / P) * (kLightLutMaxN + 1u) + k % P] = axisTableEntry(k / P, k % P, st.N, strideY, strideZ, regions);
		__syncthreads();
		axisTables = sAxis;
	}

	const uint32_t L = st.L;
	const uint32_t blocksPerAxis = (L + 3u) >> 2;
	const uint32_t totalBlocks = blocksPerAxis * blocksPerAxis * blocksPerAxis;
	const LightShare share = lightShare(st, totalBlocks);
	uint32_t e = share.lo + blockIdx.x * kLightWaves + (threadIdx.x >> 6);	// entry of the dispatch order
	if (ORDERED)
	{
		// workgroup i runs on XCD i % 8: it takes the next kLightWaves blocks of that XCD's eighth of (this context's share of) the tube order (lightOrderKernel)
		const uint32_t perXcd = (share.hi - share.lo + 7u) >> 3, xcd = blockIdx.x & 7u;
		const uint32_t k = xcd * perXcd + (blockIdx.x >> 3) * kLightWaves + (threadIdx.x >> 6);
		if (k >= (xcd + 1u) * perXcd) return;
		e = share.lo + k;
	}
	if (e >= share.hi) return;
	uint32_t b = ORDERED ? st.lightOrder[e] : e;	// (layers, ordered: the order holds this rank's blocks only, lightOrderRole)
	if (!ORDERED && (st.flags & kFlagLightLayers) && st.shardCount > 1u)
	{
		const uint32_t perLayer = blocksPerAxis * blocksPerAxis, i = e / perLayer;	// the i-th layer of this rank
		b = lightLayerOfRank(blocksPerAxis, st.shardIndex, st.shardCount, i) * perLayer + (e - i * perLayer);
	}
	const uint32_t bx = b % blocksPerAxis, by = (b / blocksPerAxis) % blocksPerAxis, bz = b / (blocksPerAxis * blocksPerAxis);
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t xx = bx * 4 + (lane & 3u), yy = by * 4 + ((lane >> 2) & 3u), zz = bz * 4 + (lane >> 4);
	const bool inRange = xx < L && yy < L && zz < L;

#ifndef MV_LIGHT_CANDIDATES
#define MV_LIGHT_CANDIDATES 0	// measured (round 3, profiles/r03_light_candidates.txt): no gain -- the loop's instructions are its samples, not its rejections
#endif
	// ---- candidate volumes of this voxel block (round 3 experiment, off) ----
	// CSRayMarchL.hlsl:77-110 walks ALL V volumes for every voxel and ComputeRayOrigin rejects nearly all of them (a voxel's light ray
	// touches two or three boxes of a 4 x 4 grid).  The light is directional: the rays of a block's 64 voxels towards the light are
	// parallel in every volume's space and start inside the image of the block, so one conservative test per (block, volume) -- lane n
	// tests volume n: the line through the block's centre against the box's circumscribed sphere grown by the block's radius there --
	// decides for the whole wavefront which volumes a light ray of the block can touch at all.  A clear bit means that every voxel's
	// ComputeRayOrigin (RayMarch.hlsli:128-155) would return false: the loop below then `continue`s exactly as the reference does
	// (which also skips that volume's AO ray, CSRayMarchL.hlsl:95).  Results are unchanged by construction; the per-voxel tests stay.
	unsigned long long cand0 = ~0ull;
	if (MV_LIGHT_CANDIDATES)
	{
		const PerObject& pb = perObjects[volumeId];
		const float h = 4.0f / (float)L;	// half extent of the block in the relit volume's [-1, 1] space
		const vec3 cw = mul43(v3(((float)(bx * 4u) + 2.0f) / (float)L * 2.0f - 1.0f, ((float)(by * 4u) + 2.0f) / (float)L * 2.0f - 1.0f, ((float)(bz * 4u) + 2.0f) / (float)L * 2.0f - 1.0f), pb.world);
		for (uint32_t n0 = 0; n0 < st.V; n0 += 64u)
		{
			const uint32_t n = n0 + lane;
			bool cand = false;
			if (n < st.V)
			{
				const float* wi = perObjects[n].worldI;
				const vec3 c = mul43(cw, wi);
				// images of the block's three half axes in volume n's space: columns of (WorldI_n) (World_0), times h
				float R = 0.0f;
#pragma unroll
				for (int i = 0; i < 3; ++i)
				{
					const vec3 a = v3(pb.world[i], pb.world[4 + i], pb.world[8 + i]);	// World_0 applied to unit axis i (linear part)
					const vec3 m = mul33(a, wi);
					R += __builtin_sqrtf(dot3(m, m));
				}
				R *= h;
				const float4 ld = st.lightDirs[n];
				const float oo = dot3(c, c), od = c.x * ld.x + c.y * ld.y + c.z * ld.z;
				const float d2 = __builtin_fmaxf(oo - od * od, 0.0f);
				const float reach = 1.7607f + R + 1e-3f * (1.0f + __builtin_sqrtf(oo));	// sqrt(3.1): the margin of missesUnitBox, + the block, + rounding
				cand = !(d2 > reach * reach + 1e-5f * oo);	// (NaN: candidate)
			}
			const unsigned long long m = __ballot(cand);
			if (n0 == 0u) cand0 = m;
			else *(volatile unsigned long long*)&sCand[threadIdx.x >> 6][n0 >> 6] = m;	// (every lane stores the same word)
		}
	}

	uint32_t nS = 0, packed = 0u;
	if (inRange)
	{
		const float gridSize = (float)L;
		const float gStep = kMaxDist / (float)st.maxLightSamples;	// RayMarch.hlsli:18
		const uint32_t volTexId0 = st.volumeDescs[volumeId] & 0x3fffu;
		const PerObject& po0 = perObjects[volumeId];
		const VolumeView vol0 = makeVolumeView(st.volumes[volTexId0], st.alphaCells[volTexId0], st.N, st.bricks);

		vec3 rayOrigin = v3(((float)xx + 0.5f) / gridSize * 2.0f - 1.0f, ((float)yy + 0.5f) / gridSize * 2.0f - 1.0f, ((float)zz + 0.5f) / gridSize * 2.0f - 1.0f);
		// LocalToTex3DSpace(rayOrigin) = ((x + 0.5) / L * 2 - 1) * 0.5 + 0.5, compiled (fast math) as (x + 0.5) / L
		const vec3 uvw = v3(((float)xx + 0.5f) / gridSize, ((float)yy + 0.5f) / gridSize, ((float)zz + 0.5f) / gridSize);
		const float density = sampleAlpha16F(vol0, uvw);
		const bool hasDensity = density >= kZeroThreshold;	// :46

		rayOrigin = mul43(rayOrigin, po0.world);	// :47 light-map space to world space

		// ShadowTest (RayMarch.hlsli:103-112)
		float shadow;
		{
			vec4 p; p.x = rayOrigin.x; p.y = rayOrigin.y; p.z = rayOrigin.z; p.w = 1.0f;
			// in the compiled order (CSRayMarchL.cso; oracle/mv_oracle.cpp mvo_march_light): v = 0.5 - y * 0.5, reference depth = (row.w - 0.0027) + dot3
			const vec4 ls = mul44(p, cb->shadowViewProj);
			const float* m = cb->shadowViewProj;
			const float su = ls.x * 0.5f + 0.5f;
			const float sv = 0.5f - ls.y * 0.5f;
			const float dz = p.x * m[8] + p.y * m[9] + p.z * m[10];
			shadow = sampleCmpD16(st.shadowMap, st.M, su, sv, (m[11] + -0.0027f) + dz);
		}

		float ao = 1.0f;
		vec3 irradiance = v3(0.0f, 0.0f, 0.0f);

		if (hasDensity)
		{
			vec3 aoRayDir = v3(0.0f, 0.0f, 0.0f);
			if (st.hasLightProbe)
			{
				// GetDensityGradient (RayMarch.hlsli:55-77)
				const float q0 = sampleAlpha16F(vol0, uvw, -1, 0, 0), q1 = sampleAlpha16F(vol0, uvw, 1, 0, 0);
				const float q2 = sampleAlpha16F(vol0, uvw, 0, -1, 0), q3 = sampleAlpha16F(vol0, uvw, 0, 1, 0);
				const float q4 = sampleAlpha16F(vol0, uvw, 0, 0, -1), q5 = sampleAlpha16F(vol0, uvw, 0, 0, 1);
				aoRayDir = v3(-(q1 - q0), -(q3 - q2), -(q5 - q4));
				const bool nz = __builtin_fabsf(aoRayDir.x) > 0.0f || __builtin_fabsf(aoRayDir.y) > 0.0f || __builtin_fabsf(aoRayDir.z) > 0.0f;
				aoRayDir = v3(nz ? aoRayDir.x : rayOrigin.x, nz ? aoRayDir.y : rayOrigin.y, nz ? aoRayDir.z : rayOrigin.z);	// :69 (rayOrigin is already world space here)
				aoRayDir = mul33(aoRayDir, po0.world);
				aoRayDir = normalize3(aoRayDir);
				irradiance = EvaluateSHIrradiance(st.sh, normalize3(aoRayDir));	// GetIrradiance normalises again (RayMarch.hlsli:121)
			}

			for (uint32_t n = 0; n < st.V; ++n)	// :77-110
			{
				// (wave-uniform) can a light ray of this block touch volume n at all?
				const bool cand = !MV_LIGHT_CANDIDATES || (((n < 64u ? cand0 : *(volatile unsigned long long*)&sCand[threadIdx.x >> 6][n >> 6]) >> (n & 63u)) & 1ull);
				if (!cand && !st.hasLightProbe) continue;	// no light ray (it misses, or the voxel is in shadow already) and no AO ray: nothing happens for n
				const uint32_t volTexId = st.volumeDescs[n] & 0x3fffu;
				const PerObject& po = perObjects[n];
				const VolumeView vol = makeVolumeView(st.volumes[volTexId], st.alphaCells[volTexId], st.N, st.bricks);
				vec3 localRayOrigin = mul43(rayOrigin, po.worldI);

				if (shadow >= kZeroThreshold)
				{
					if (!cand) continue;	// what missesUnitBox / ComputeRayOrigin below would conclude for every voxel of the block
					const float4 ld = st.lightDirs[n];	// = normalize(mul(lightPos, (float3x3)WorldI_n)), :88-89, evaluated once by the cull
					const vec3 rayDir = v3(ld.x, ld.y, ld.z);
					if (missesUnitBox(localRayOrigin, rayDir)) continue;	// what ComputeRayOrigin would conclude, at a fifth of the cost
					if (!ComputeRayOrigin(localRayOrigin, rayDir)) continue;
					CastLightRay<K>(shadow, vol, localRayOrigin, rayDir, gStep, st.maxLightSamples, nS, axisTables);
				}

				if (st.hasLightProbe)
				{
					const vec3 rayDir = normalize3(mul33(aoRayDir, po.worldI));
					if (missesUnitBox(localRayOrigin, rayDir)) continue;
					if (!ComputeRayOrigin(localRayOrigin, rayDir)) continue;
					float transm = 1.0f;
					CastLightRay<K>(transm, vol, localRayOrigin, rayDir, gStep, st.maxLightSamples, nS, axisTables);
					ao *= n == volumeId ? transm : __builtin_sqrtf(__builtin_sqrtf(saturatef(transm + 0.5f)));	// pow(x, 0.25)
				}
			}
		}

		const vec3 lightColor = v3(cb->lightColor[0] * cb->lightColor[3], cb->lightColor[1] * cb->lightColor[3], cb->lightColor[2] * cb->lightColor[3]);
		vec3 ambient = v3(cb->ambient[0] * cb->ambient[3], cb->ambient[1] * cb->ambient[3], cb->ambient[2] * cb->ambient[3]);
		if (st.hasLightProbe) ambient = v3(ao * irradiance.x, ao * irradiance.y, ao * irradiance.z);

		const vec3 result = v3(shadow * lightColor.x + ambient.x, shadow * lightColor.y + ambient.y, shadow * lightColor.z + ambient.z);
		packed = pack_r11g11b10f(result.x, result.y, result.z);
		storeLightTexel(reinterpret_cast<uint32_t*>(st.lightTable[volumeId]), L, xx, yy, zz, packed);
		if (st.lightPlain) st.lightPlain[((size_t)zz * L + yy) * L + xx] = packed;	// (peer stores, plain light push: what the push workgroups send)
	}
	if (wordsOut) wordsOut[(size_t)(e - share.lo) * 64u + lane] = packed;

	const uint32_t tot = waveSumU32(nS);
	// this voxel block's own slot; an atomic because light marches of consecutive frames may run side by side
	if (lane == 0 && tot) atomicAdd(reinterpret_cast<unsigned long long*>(st.lightStats) + b, (unsigned long long)tot);
}

#ifndef MV_ORDER_DILATE
#define MV_ORDER_DILATE 1	// bin a tile by the largest cost in its 3x3 neighbourhood when the camera or a volume has moved since the costs were taken (0: never, 2: always; A/B timing)
#endif
constexpr uint32_t kOrderWaves = 256;	// (kOrderBins: mv_kernels.hpp)

// ---------------------------------------------------------------------------------------------
// Dispatch order of the view march: tiles binned by the length of their longest ray in the last launch that used this
// frame slot (four frames back: the cost table lives with the cull outputs), longest bins first, list order inside a bin (a stable counting sort, so neighbouring tiles
// still follow each other and share L2 lines).  The tail of the launch is then made of short tiles instead of whichever
// long ones the list happened to end with: 0.362 -> 0.326 ms at C3.  A full sort by cost is SLOWER than list order
// (0.378 ms): it destroys the locality between neighbouring tiles; two bins give half the gain.  The order only decides
// which wavefront marches which tile -- every tile is marched exactly once whatever the costs say, so a stale or empty
// cost table (first frame, moved camera, changed mip) only costs speed.
// Two launches of kOrderWaves single-wave workgroups right behind the cull, each owning a contiguous chunk of the tile list:
// count (per-wave bin histogram) and scatter (every wave re-derives its 8 bases from the 256 x 8 histogram).
// ---------------------------------------------------------------------------------------------
__device__ inline uint32_t tileBinOf(const DeviceState& st, const uint32_t* sStart, const uint32_t* sEntry, uint32_t nEntries, uint32_t nCubeTiles, uint32_t tile)
{
	// last entry whose first tile is <= tile (entries with no tiles repeat a start value: take the last one)
	uint32_t lo = 0, hi = nEntries;
	while (hi - lo > 1u)
	{
		const uint32_t mid = (lo + hi) >> 1;
		if (sStart[mid] <= tile) lo = mid; else hi = mid;
	}
	const uint32_t we = sEntry[lo];
	const uint32_t volumeId = we & 0xffffu, faceId = we >> 16, localTile = faceId == 7u ? tile - sStart[lo] : cubeLocalTile(st, volumeId, tile - sStart[lo]);
	const uint32_t key = tileCostKey(st, volumeId, faceId, localTile, tile, nCubeTiles);
	const uint32_t width = st.maxRaySamples >= kOrderBins ? st.maxRaySamples / kOrderBins : 1u;
	uint32_t cost = st.tileCost[key];
#if MV_ORDER_DILATE
	if (MV_ORDER_DILATE == 2 || st.counters[11])
	// The table is a frame or two old.  With a moving camera (or moving volumes: the cull compares a hash of both with the one of the
	// frame that wrote the table) nearly every tile still costs what it did (5 % change bin, 1.2 samples of
	// mean drift at 0.004 rad per frame) -- but the few tiles a silhouette has moved INTO start late and run long, and the launch ends
	// with them (+10 %: scripts/exp_orbit_positions.py).  A tile is therefore binned by the largest cost among itself and its
	// neighbours in the face (or on the screen): a silhouette moves less than a tile in that time.
	{
		uint32_t tilesX, tilesY, stride = 1u;
		if (faceId == 7u)
		{
			const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[volumeId];
			const uint32_t x1 = rect.z < st.W - 1u ? rect.z : st.W - 1u, y1 = rect.w < st.H - 1u ? rect.w : st.H - 1u;
			const ScreenTileRows rows = screenTileRows(st, rect.y, y1);
			tilesX = (x1 >> 3) - (rect.x >> 3) + 1u; tilesY = rows.count ? rows.count : 1u;
			stride = (st.W + 7u) >> 3;	// screen keys are absolute: one row of tiles down = one row of the viewport
		}
		else
		{
			const uint32_t dims = cubeMipDim(st.N, reinterpret_cast<const ushort4*>(st.volumeInfo)[volumeId].x);
			tilesX = tilesY = (dims + 7u) >> 3;
			stride = tilesX;
		}
		const uint32_t lx = localTile % tilesX, ly = localTile / tilesX;
		for (int dy = -1; dy <= 1; ++dy)
			for (int dx = -1; dx <= 1; ++dx)
			{
				if ((dx | dy) == 0 || (int)lx + dx < 0 || (int)lx + dx >= (int)tilesX || (int)ly + dy < 0 || (int)ly + dy >= (int)tilesY) continue;
				const uint32_t c = st.tileCost[(uint32_t)((int)key + dy * (int)stride + dx)];
				cost = c > cost ? c : cost;
			}
	}
#endif
	if (cost == 0u) return 0u;	// never marched (a face or a screen tile that has just come into view): with the longest, where a wrong guess costs nothing
#if MV_TILE_BALANCE
	{
		// time costs: bin 0 from 2.56 us per step of the longest possible ray upwards, four bins to the octave below
		const float l = 4.0f * __log2f((float)st.maxRaySamples * 256.0f / (float)cost);
		return l <= 0.0f ? 0u : l >= (float)(kOrderBins - 1u) ? kOrderBins - 1u : (uint32_t)l;
	}
#endif
	const uint32_t b = (cost - 1u) / width;
	return kOrderBins - 1u - (b < kOrderBins - 1u ? b : kOrderBins - 1u);
}

// One pass (round 3; two launches with a 256 x 8 histogram between them before): every wavefront bins the tiles of its chunk (the bins
// of up to eight batches stay in one register, four bits each), reserves room for them in every bin with one returning atomic per
// bin (binCursor[b], zeroed by the cull), and scatters them.  Each bin has a region of its own in `order` (orderCap entries apart),
// so nobody needs the totals of the other wavefronts: the march derives dispatch slot -> (bin, index) from the eight cursors, which
// are the bin totals once this kernel has finished (orderedTile, mv_kernels.hpp).  Inside a bin the chunks follow each other in
// the order their wavefronts arrived; the tiles of a chunk stay in list order.
__device__ __forceinline__ void tileOrderRole(const DeviceState& st, uint32_t* sWork, uint32_t chunkIndex, uint32_t numChunks)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t nEntries = st.counters[2], nTiles = st.counters[3], nCubeTiles = st.counters[6];
	uint32_t* sStart = sWork;
	uint32_t* sEntry = sWork + nEntries + 1u;
	for (uint32_t i = threadIdx.x; i <= nEntries; i += blockDim.x) sStart[i] = st.workTileStart[i];
	for (uint32_t i = threadIdx.x; i < nEntries; i += blockDim.x) sEntry[i] = st.workEntries[i];
	__syncthreads();

	const uint32_t chunk = (nTiles + numChunks - 1u) / numChunks;
	const uint32_t begin = chunkIndex * chunk;
	const uint32_t end = begin + chunk < nTiles ? begin + chunk : nTiles;
	if (begin >= end) return;
	uint32_t* order = const_cast<uint32_t*>(st.tileOrder);

	uint32_t count[kOrderBins];
#pragma unroll
	for (uint32_t b = 0; b < kOrderBins; ++b) count[b] = 0;
	uint64_t cached = 0;	// bins of this lane's tile in batches 0..7, five bits each (kOrderBins = no tile)
	uint32_t batch = 0;
	for (uint32_t first = begin; first < end; first += 64u, ++batch)
	{
		const uint32_t tile = first + lane;
		const uint32_t bin = tile < end ? tileBinOf(st, sStart, sEntry, nEntries, nCubeTiles, tile) : kOrderBins;
		if (batch < 8u) cached |= (uint64_t)bin << (5u * batch);
#pragma unroll
		for (uint32_t b = 0; b < kOrderBins; ++b) count[b] += (uint32_t)__popcll(__ballot(bin == b));
	}
	// lane b reserves this chunk's room in bin b
	uint32_t mine = 0;
#pragma unroll
	for (uint32_t b = 0; b < kOrderBins; ++b) mine = lane == b ? count[b] : mine;
	uint32_t got = 0;
	if (lane < kOrderBins && mine) got = atomicAdd(&st.binCursor[lane], mine);
	uint32_t base[kOrderBins];
#pragma unroll
	for (uint32_t b = 0; b < kOrderBins; ++b) base[b] = __shfl(got, (int)b) + b * st.orderCap;

	batch = 0;
	for (uint32_t first = begin; first < end; first += 64u, ++batch)
	{
		const uint32_t tile = first + lane;
		const uint32_t bin = batch < 8u ? (uint32_t)(cached >> (5u * batch)) & 0x1fu : tile < end ? tileBinOf(st, sStart, sEntry, nEntries, nCubeTiles, tile) : kOrderBins;
#pragma unroll
		for (uint32_t b = 0; b < kOrderBins; ++b)
		{
			const uint64_t m = __ballot(bin == b);
			if (bin == b) order[base[b] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = tile;
			base[b] += (uint32_t)__popcll(m);
		}
	}
}

// Sums the per-wavefront counter slots into st.stats (run by mv_get_stats, not per frame).
__global__ __launch_bounds__(256) void reduceStatsKernel(DeviceState st, int includeLight)
{
	unsigned long long a[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };	// [6] = lit samples of the fused RayCast tiles, [7] = samples marched without a texel fetch
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nThreads = gridDim.x * blockDim.x;
	for (uint32_t i = tid; i < st.marchWaves; i += nThreads)
	{
		const uint64_t* w = st.waveStats + (size_t)i * 8u;
		a[0] += w[0]; a[1] += w[1]; a[2] += w[2]; a[5] += w[3]; a[4] += w[4]; a[6] += w[5]; a[7] += w[6];
	}
	if (includeLight) for (uint32_t i = tid; i < st.lightBlocks; i += nThreads) a[3] += st.lightStats[i];
	for (uint32_t i = tid; i < st.compositeSlots; i += nThreads) a[4] += st.compositeStats[i];
#pragma unroll
	for (int k = 0; k < 8; ++k)
	{
		unsigned long long v = a[k];
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if ((threadIdx.x & 63u) == 0u && v) atomicAdd(reinterpret_cast<unsigned long long*>(st.stats) + k, v);
	}
}

uint32_t marchViewMaxTiles(uint32_t V, uint32_t N, uint32_t shardCount)
{
	// every owned volume at mip 0 with all 6 faces
	const uint32_t sc = shardCount ? shardCount : 1u;
	const uint32_t owned = (V + sc - 1u) / sc;
	const uint32_t t = (N + 7u) >> 3;
	return owned * 6u * t * t;
}

void launchReduceStats(hipStream_t s, const DeviceState& st, int includeLight)
{
	hipLaunchKernelGGL(reduceStatsKernel, dim3(64), dim3(256), 0, s, st, includeLight);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline uint32_t gridFor(size_t total, uint32_t block)
{
	const size_t g = (total + block - 1) / block;
	return (uint32_t)(g < 8192 ? (g ? g : 1) : 8192);	// 256 CUs x 8 blocks x 4: grid-stride the rest
}

void launchRetile(hipStream_t s, const uint2* linear, uint2* bricked, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	hipLaunchKernelGGL(retileKernel, dim3(gridFor(total, 256)), dim3(256), 0, s, linear, bricked, N, bricks);
}

void launchUntile(hipStream_t s, const uint2* bricked, uint2* linear, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)N * N * N;
	hipLaunchKernelGGL(untileKernel, dim3(gridFor(total, 256)), dim3(256), 0, s, bricked, linear, N, bricks);
}

void launchInitGridData(hipStream_t s, uint2* bricked, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	hipLaunchKernelGGL(initGridDataKernel, dim3(gridFor(total, 256)), dim3(256), 0, s, bricked, N, bricks);
}

void launchR32FToRGBA16F(hipStream_t s, const float* src, uint32_t w, uint32_t h, uint32_t d, uint2* bricked, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	hipLaunchKernelGGL(r32fToRGBA16FKernel, dim3(gridFor(total, 256)), dim3(256), 0, s, src, w, h, d, bricked, N, bricks);
}

void launchBuildAlphaCells(hipStream_t s, const uint2* bricked, uint4* cells, uint32_t N, uint32_t bricks)
{
	const size_t total = (size_t)bricks * bricks * bricks * 64;
	hipLaunchKernelGGL(buildAlphaCellsKernel, dim3(gridFor(total, 256)), dim3(256), 0, s, bricked, cells, N, bricks);
}

void launchBuildOccupancy(hipStream_t s, const uint2* bricked, uint64_t* words, uint32_t N, uint32_t bricks)
{
	const uint32_t regions = occupancyRegions(bricks);
	hipLaunchKernelGGL(buildOccupancyKernel, dim3(gridFor((size_t)regions * regions * regions * 64, 64)), dim3(64), 0, s, bricked, words, N, bricks);
}

void launchLightToQuads(hipStream_t s, const uint32_t* linear, uint4* quads, uint32_t L)
{
	hipLaunchKernelGGL(lightToQuadsKernel, dim3(gridFor((size_t)L * L * L, 256)), dim3(256), 0, s, linear, quads, L);
}

void launchLightFromQuads(hipStream_t s, const uint4* quads, uint32_t* linear, uint32_t L)
{
	hipLaunchKernelGGL(lightFromQuadsKernel, dim3(gridFor((size_t)L * L * L, 256)), dim3(256), 0, s, quads, linear, L);
}

void launchCull(hipStream_t s, const DeviceState& st)
{
	hipLaunchKernelGGL(cullKernel, dim3((st.V + 7u) / 8u), dim3(kCullThreads), st.V * 4 * sizeof(uint32_t), s, st);
}
void launchCullBatch(hipStream_t s, const DeviceState* states, uint32_t n)
{
	DeviceStateBatch p;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i] = states[i < n ? i : n - 1u];
	hipLaunchKernelGGL(cullBatchKernel, dim3((states[0].V + 7u) / 8u, n), dim3(kCullThreads), states[0].V * 4 * sizeof(uint32_t), s, p);
}


// ---------------------------------------------------------------------------------------------
// Dispatch order of the light march, XCD-aware.  Every voxel of the relit volume casts a ray towards the light through its own
// volume, and the light is directional: the rays are parallel, so the 4x4x4 voxel blocks along the light direction walk through the
// same texels one after the other.  Workgroups are dealt round-robin over the 8 XCDs, each with its own L2; in plain x-y-z order every
// XCD touches the whole volume and every line is fetched by several L2s (3.2 L2 misses per line of the alpha structure at C3).
// This kernel sorts the blocks by the "tube" they belong to -- block coordinates sheared along the dominant axis of the light
// direction, so that the blocks of one tube follow the light -- and the march gives XCD x the x-th eighth of that order, one
// contiguous bundle of tubes, walked tube by tube.  The order only decides which wavefront marches which block.
// One small workgroup, a counting sort over at most (2 B)^2 tubes in LDS (B = blocks per axis <= kLightOrderMaxB), behind the cull.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kLightOrderMaxB = 32, kLightOrderThreads = 256;	// (four wavefronts: a 1024-thread workgroup waited 80-160 us for sixteen free wave slots on one CU of a GPU the view marches fill)
constexpr uint32_t kLightOrderLdsWords = 4u * kLightOrderMaxB * kLightOrderMaxB + kLightOrderThreads / 64u;
__device__ __forceinline__ void lightOrderRole(const DeviceState& st, uint32_t* sMem)
{
	uint32_t* sCursor = sMem;	// 4 kLightOrderMaxB^2 tube cursors
	uint32_t* sWave = sMem + 4u * kLightOrderMaxB * kLightOrderMaxB;
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
	const uint32_t volumeId = relitVolume(st, cb->frameIdx, st.counters[0], st.visibleVolumes);
	const uint32_t B = (st.L + 3u) >> 2, T = 2u * B, tubes = T * T;
	// (peer-store transport: the light march is shared out by layers of blocks -- layer l of the z axis to rank l % shardCount -- and the
	// order holds THIS rank's blocks only, in tube order: the i-th block of the loops below is ownBlock(i))
	const bool layers = (st.flags & kFlagLightLayers) && st.shardCount > 1u;
	const uint32_t myLayers = layers ? lightLayersOfRank(B, st.shardIndex, st.shardCount) : B;
	const uint32_t total = myLayers * B * B;
	auto ownBlock = [&](uint32_t i) -> uint32_t { if (!layers) return i; const uint32_t l = i / (B * B); return lightLayerOfRank(B, st.shardIndex, st.shardCount, l) * B * B + (i - l * B * B); };
	const float4 ld = st.lightDirs[volumeId];
	float d[3] = { ld.x, ld.y, ld.z };
	// The order depends on the direction only, and the slot's table still holds the order for the direction of four frames ago (kept
	// behind the table): with volumes that share an orientation and a light that stands still it is the same one, frame after frame.
	{
		uint32_t* tag = st.lightOrder + B * B * B;	// (behind the largest order there can be)
		const uint32_t what = total | (layers ? 0x80000000u : 0u);
		const bool same = tag[0] == __float_as_uint(d[0]) && tag[1] == __float_as_uint(d[1]) && tag[2] == __float_as_uint(d[2]) && tag[3] == what;
		__syncthreads();	// everybody has read the tag
		if (same) return;
		if (threadIdx.x == 0) { tag[0] = __float_as_uint(d[0]); tag[1] = __float_as_uint(d[1]); tag[2] = __float_as_uint(d[2]); tag[3] = what; }
	}
	// dominant axis a; p, q the other two; slopes of the shear (0 for a direction that is not a finite, non-zero vector)
	uint32_t a = 0;
	if (__builtin_fabsf(d[1]) > __builtin_fabsf(d[a])) a = 1;
	if (__builtin_fabsf(d[2]) > __builtin_fabsf(d[a])) a = 2;
	const uint32_t p = (a + 1u) % 3u, q = (a + 2u) % 3u;
	const bool usable = __builtin_fabsf(d[a]) > 0.0f && __builtin_fabsf(d[a]) < 3.0e38f;
	const float sp = usable ? d[p] / d[a] : 0.0f, sq = usable ? d[q] / d[a] : 0.0f;
	auto tubeOf = [&](uint32_t b) -> uint32_t
	{
		uint32_t c[3] = { b % B, (b / B) % B, b / (B * B) };
		const float ka = (float)c[a];
		// |slope| <= 1, so the sheared coordinate spans 2 B - 1 values: [-(B - 1), B - 1] for a positive slope, [0, 2 B - 2] for a negative one
		const int tp = (int)c[p] - (int)__builtin_rintf(sp * ka) + (sp >= 0.0f ? (int)B - 1 : 0), tq = (int)c[q] - (int)__builtin_rintf(sq * ka) + (sq >= 0.0f ? (int)B - 1 : 0);
		// (the clamp keeps any rounding surprise a valid, if poorer, bucket)
		const uint32_t up = (uint32_t)(tp < 0 ? 0 : tp) % T, uq = (uint32_t)(tq < 0 ? 0 : tq) % T;
		return up * T + uq;
	};
	for (uint32_t i = threadIdx.x; i < tubes; i += kLightOrderThreads) sCursor[i] = 0u;
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += kLightOrderThreads) atomicAdd(&sCursor[tubeOf(ownBlock(i))], 1u);
	__syncthreads();
	// exclusive prefix over the tubes: every thread owns a run of consecutive counts
	const uint32_t per = (tubes + kLightOrderThreads - 1u) / kLightOrderThreads;
	uint32_t sum = 0u;
	for (uint32_t k = 0; k < per; ++k) { const uint32_t i = threadIdx.x * per + k; sum += i < tubes ? sCursor[i] : 0u; }
	uint32_t incl = sum;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
	if (lane == 63u) sWave[wave] = incl;
	__syncthreads();
	uint32_t base = incl - sum;
	for (uint32_t w = 0; w < wave; ++w) base += sWave[w];
	for (uint32_t k = 0; k < per; ++k)
	{
		const uint32_t i = threadIdx.x * per + k;
		if (i < tubes) { const uint32_t c = sCursor[i]; sCursor[i] = base; base += c; }	// (a thread's own run: nobody else reads or writes it)
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += kLightOrderThreads) { const uint32_t b = ownBlock(i); st.lightOrder[atomicAdd(&sCursor[tubeOf(b)], 1u)] = b; }
	// Inside a tube the atomics above leave the blocks in arrival order.  The order must be the same on every rank of the multi-GPU loop
	// (ranks march disjoint ranges of it) and in every frame slot, so every tube is put into a defined order: along the light's dominant
	// axis, then by block index (a tube holds at most B blocks, one per step along that axis; an insertion sort in place).
	__threadfence_block();
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < tubes; i += kLightOrderThreads)
	{
		const uint32_t begin = i ? sCursor[i - 1u] : 0u, end = sCursor[i];
		for (uint32_t j = begin + 1u; j < end; ++j)
		{
			const uint32_t b = st.lightOrder[j];
			const uint32_t cb[3] = { b % B, (b / B) % B, b / (B * B) };
			const uint32_t key = (cb[a] << 20) | b;
			uint32_t k = j;
			while (k > begin)
			{
				const uint32_t o = st.lightOrder[k - 1u];
				const uint32_t co[3] = { o % B, (o / B) % B, o / (B * B) };
				if (((co[a] << 20) | o) <= key) break;
				st.lightOrder[k] = o;
				--k;
			}
			st.lightOrder[k] = b;
		}
	}
}

// The two orders a frame needs -- the dispatch order of the view march's tiles and (when the light direction has changed) of the light
// march's voxel blocks -- in ONE launch behind the cull: workgroups 0 .. kOrderGroups - 1 order the tiles, four chunks each (one
// per wavefront); the last workgroup orders the voxel blocks.  (Three launches before: every launch of a frame's prologue is ~4 us
// of host time and ~5 us of latency in front of the marches, which is what a rank of the multi-GPU loop is made of.)
constexpr uint32_t kOrderGroups = kOrderWaves * 64u / kLightOrderThreads;
__global__ __launch_bounds__(kLightOrderThreads) void ordersKernel(DeviceState st, uint32_t tileGroups)
{
	extern __shared__ uint32_t sOrders[];
	if (blockIdx.x < tileGroups) tileOrderRole(st, sOrders, blockIdx.x * (kLightOrderThreads / 64u) + (threadIdx.x >> 6), tileGroups * (kLightOrderThreads / 64u));
	else lightOrderRole(st, sOrders);
}

__global__ __launch_bounds__(kLightOrderThreads) void ordersBatchKernel(DeviceStateBatch p, uint32_t tileGroups)	// blockIdx.y = the frame
{
	extern __shared__ uint32_t sOrders[];
	const DeviceState& st = p.s[blockIdx.y];
	if (blockIdx.x < tileGroups) tileOrderRole(st, sOrders, blockIdx.x * (kLightOrderThreads / 64u) + (threadIdx.x >> 6), tileGroups * (kLightOrderThreads / 64u));
	else lightOrderRole(st, sOrders);
}
void launchOrdersBatch(hipStream_t s, const DeviceState* states, uint32_t n)
{
	const DeviceState& st = states[0];
	const bool tiles = st.tileOrder != nullptr;
	const bool light = st.lightOrder && ((st.L + 3u) >> 2) <= kLightOrderMaxB;
	if (!tiles && !light) return;
	DeviceStateBatch p;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i] = states[i < n ? i : n - 1u];
	const size_t tileLds = (2 * ((size_t)st.V * 6 + st.screenSlots) + 1) * sizeof(uint32_t), lightLds = (size_t)kLightOrderLdsWords * sizeof(uint32_t);
	const uint32_t tileGroups = tiles ? kOrderGroups : 0u;
	hipLaunchKernelGGL(ordersBatchKernel, dim3(tileGroups + (light ? 1u : 0u), n), dim3(kLightOrderThreads), tileLds > lightLds ? tileLds : lightLds, s, p, tileGroups);
}

void launchOrders(hipStream_t s, const DeviceState& st)
{
	const bool tiles = st.tileOrder != nullptr;
	const bool light = st.lightOrder && ((st.L + 3u) >> 2) <= kLightOrderMaxB;
	if (!tiles && !light) return;
	const size_t tileLds = (2 * ((size_t)st.V * 6 + st.screenSlots) + 1) * sizeof(uint32_t), lightLds = (size_t)kLightOrderLdsWords * sizeof(uint32_t);
	const uint32_t tileGroups = tiles ? kOrderGroups : 0u;
	hipLaunchKernelGGL(ordersKernel, dim3(tileGroups + (light ? 1u : 0u)), dim3(kLightOrderThreads), tileLds > lightLds ? tileLds : lightLds, s, st, tileGroups);
}

// The other ranks' shares of the relit light map, all-gathered as plain words (64 per voxel block, dispatch order, `per` blocks per
// rank), into this rank's copy of the map: one wavefront per block, the same store as the march's.
__global__ __launch_bounds__(256) void lightUnpackKernel(DeviceState st, const uint32_t* __restrict__ words, int ordered)
{
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
	const uint32_t volumeId = relitVolume(st, cb->frameIdx, st.counters[0], st.visibleVolumes);
	const uint32_t L = st.L, B = (L + 3u) >> 2, total = B * B * B;
	const LightShare share = lightShare(st, total);
	const size_t stride = ((size_t)share.per + 1u) * 64u;	// words per rank: `per` blocks + the trailer
	if (blockIdx.x == 0)
	{
		// every rank's cull decided the same?  (see marchLightKernel)
		bool differs = false;
		for (uint32_t r = threadIdx.x; r < st.shardCount; r += blockDim.x)
		{
			const uint32_t* t = words + (size_t)r * stride + (size_t)share.per * 64u;
			differs = differs || t[0] != st.frameTag || t[1] != st.counters[12] || t[2] != st.counters[0] || t[3] != st.counters[1];
		}
		if (differs && st.exchangeReport) st.exchangeReport[1] = ((unsigned long long)st.frameTag << 32) | 1ull;	// (pinned host memory; every writer stores the same word)
	}
	const uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (e >= total || (e >= share.lo && e < share.hi)) return;	// (its own share is in place already)
	const uint32_t b = ordered ? st.lightOrder[e] : e;
	const uint32_t r = e / share.per;
	const uint32_t xx = (b % B) * 4u + (lane & 3u), yy = ((b / B) % B) * 4u + ((lane >> 2) & 3u), zz = (b / (B * B)) * 4u + (lane >> 4);
	if (xx < L && yy < L && zz < L)
		storeLightTexel(reinterpret_cast<uint32_t*>(st.lightTable[volumeId]), L, xx, yy, zz, words[(size_t)r * stride + (size_t)(e - r * share.per) * 64u + lane]);
}

#ifndef MV_LIGHT_ORDER
#define MV_LIGHT_ORDER 1	// XCD-aware tube order of the light march's voxel blocks (lightOrderKernel); 0 = x-y-z order (A/B timing)
#endif
#ifndef MV_LIGHT_LUT
#define MV_LIGHT_LUT 1	// per-axis address tables in LDS for the light march (N <= 256); 0 = compute every tap's offsets (A/B timing)
#endif
static bool lightOrdered(const DeviceState& st) { return MV_LIGHT_ORDER && st.lightOrder && ((st.L + 3u) >> 2) <= kLightOrderMaxB; }

uint32_t lightSharePerRank(const DeviceState& st)
{
	const uint32_t b = (st.L + 3u) >> 2;
	return lightShare(st, b * b * b).per;
}

template <int K, bool PEER>
static void launchMarchLightKP(hipStream_t s, const DeviceState& st, uint32_t* wordsOut, uint32_t count)
{
	if (lightOrdered(st))
	{
		const uint32_t perXcd = (count + 7u) >> 3;
		const uint32_t groups = 8u * ((perXcd + kLightWaves - 1u) / kLightWaves);
		if (MV_LIGHT_LUT && st.N <= kLightLutMaxN) hipLaunchKernelGGL((marchLightKernel<true, true, K, PEER>), dim3(groups), dim3(64 * kLightWaves), 0, s, st, wordsOut);
		else hipLaunchKernelGGL((marchLightKernel<false, true, K, PEER>), dim3(groups), dim3(64 * kLightWaves), 0, s, st, wordsOut);
		return;
	}
	const uint32_t groups = (count + kLightWaves - 1u) / kLightWaves;
	if (MV_LIGHT_LUT && st.N <= kLightLutMaxN) hipLaunchKernelGGL((marchLightKernel<true, false, K, PEER>), dim3(groups), dim3(64 * kLightWaves), 0, s, st, wordsOut);
	else hipLaunchKernelGGL((marchLightKernel<false, false, K, PEER>), dim3(groups), dim3(64 * kLightWaves), 0, s, st, wordsOut);
}

template <int K>
static void launchMarchLightK(hipStream_t s, const DeviceState& st, uint32_t* wordsOut, uint32_t count)
{
	if (st.fold.nWait) launchMarchLightKP<K, true>(s, st, wordsOut, count);	// (peer-store transport, folded wait)
	else launchMarchLightKP<K, false>(s, st, wordsOut, count);
}

#ifndef MV_LIGHT_SPECULATE_BELOW
#define MV_LIGHT_SPECULATE_BELOW 0	// voxel blocks per launch under which alpha footprints are requested four samples ahead.  0 = never: measured (round 3,
									// profiles/r03_speculation.txt) on a rank's eighth of the C3 light march: 46 us -> 52 us with it
#endif
void launchMarchLight(hipStream_t s, const DeviceState& st, uint32_t* wordsOut)
{
	const uint32_t b = (st.L + 3u) >> 2, total = b * b * b;
	const uint32_t count = lightShare(st, total).per;	// (the same grid on every rank; the last rank's share may be shorter)
	if (MV_LIGHT_SPECULATE && count < MV_LIGHT_SPECULATE_BELOW) launchMarchLightK<4>(s, st, wordsOut, count);
	else launchMarchLightK<MV_LIGHT_SPECULATE ? 1 : 0>(s, st, wordsOut, count);
}

void launchMarchLightBatch(hipStream_t s, const DeviceState* states, uint32_t n)
{
	DeviceStateBatch p;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i] = states[i < n ? i : n - 1u];
	const DeviceState& st = states[0];
	const uint32_t b = (st.L + 3u) >> 2, total = b * b * b;
	const uint32_t count = lightShare(st, total).per;
	constexpr int K = MV_LIGHT_SPECULATE ? 1 : 0;
	const bool lut = MV_LIGHT_LUT && st.N <= kLightLutMaxN;
	if (lightOrdered(st))
	{
		const uint32_t perXcd = (count + 7u) >> 3;
		const dim3 grid(8u * ((perXcd + kLightWaves - 1u) / kLightWaves), n);
		if (lut) hipLaunchKernelGGL((marchLightBatchKernel<true, true, K>), grid, dim3(64 * kLightWaves), 0, s, p);
		else hipLaunchKernelGGL((marchLightBatchKernel<false, true, K>), grid, dim3(64 * kLightWaves), 0, s, p);
		return;
	}
	const dim3 grid((count + kLightWaves - 1u) / kLightWaves, n);
	if (lut) hipLaunchKernelGGL((marchLightBatchKernel<true, false, K>), grid, dim3(64 * kLightWaves), 0, s, p);
	else hipLaunchKernelGGL((marchLightBatchKernel<false, false, K>), grid, dim3(64 * kLightWaves), 0, s, p);
}

void launchLightUnpack(hipStream_t s, const DeviceState& st, const uint32_t* words)
{
	const uint32_t b = (st.L + 3u) >> 2, total = b * b * b;
	hipLaunchKernelGGL(lightUnpackKernel, dim3((total + 3u) / 4u), dim3(256), 0, s, st, words, lightOrdered(st) ? 1 : 0);
}

} // namespace mv
