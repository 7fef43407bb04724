// mv_march.hip -- a3, the view-space march into cube faces (Sh/CSRayMarch.hlsl:77-158 compiled as CSRayMarchV), plus the
// screen-space RayCast of volumes on that scheme (Sh/RayCast.hlsli:43-108), as a two-stage device-side pipeline:
//
//   raySetupKernel   one wavefront per 8x8 tile (the reference's thread-group tile) in dispatch order: everything of
//                    CSRayMarch.hlsl:84-107 that happens once per ray -- GetLocalPos, ComputeRayOrigin, ComputeTargetHit,
//                    GetClipPos / the depth fetch, the cube-depth store, GetTMax -- and emits one 32-byte RayRecord per texel.
//   marchViewKernel  persistent wavefronts that pull ray records from ONE device-side queue and run the sample loop of
//                    CSRayMarch.hlsl:111-155.  A lane whose ray ends (box exit, tMax, SmpCount, transmittance < 0.01) is
//                    refilled with the next record of the queue: `__ballot` of the idle lanes, a popcount prefix for each
//                    lane's rank, one atomic add per 64 records.  Lanes of a wavefront therefore march rays of different
//                    tiles, faces and volumes side by side; everything a ray needs travels in its record and in a 32-byte
//                    per-volume constant block, and volumes are addressed as pool base + 32-bit offset.
//
// This is the shape of the reference's work-graph variant (Sh/LibRayMarch.hlsl:39-134: the cull node emits RayMarchRecords that
// the RayMarch node consumes) taken to ray granularity, with no host round trip anywhere between cull and march: both grids
// have a fixed size and read the tile count from device memory.
// A ray's arithmetic does not depend on which lane or wavefront marches it: results are bit-identical to the one-tile-per-
// wavefront form (and to the oracle).
#include "mv_device.hpp"
#include "mv_kernels.hpp"

namespace mv
{
#ifndef MV_MARCH_WAVES
#define MV_MARCH_WAVES 4	// wavefronts per workgroup (they share the address tables in LDS; otherwise independent)
#endif
#ifndef MV_MARCH_MIN_WAVES
#define MV_MARCH_MIN_WAVES 4	// occupancy bound given to the register allocator (waves per SIMD)
#endif
#ifndef MV_REFILL_MIN
#define MV_REFILL_MIN 8		// idle lanes a wavefront waits for before it refills (a refill costs a record load: ~1 us of latency)
#endif
constexpr uint32_t kMarchWaves = MV_MARCH_WAVES;
constexpr uint32_t kCostRing = 32;	// tiles a wavefront tracks at a time (one per reservation, reused round robin)
constexpr uint32_t kSetupWaves = 4;

__device__ __forceinline__ uint32_t waveSumU32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// ---------------------------------------------------------------------------------------------
// stage 1: ray setup
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kSetupWaves) void raySetupKernel(DeviceState st)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t nEntries = st.counters[2], nTiles = st.counters[3], nCubeTiles = st.counters[6];
	const uint32_t nWaves = gridDim.x * kSetupWaves;
	if (blockIdx.x == 0 && threadIdx.x < kRayQueues) st.queueHeads[threadIdx.x * kRayQueueStride] = 0u;	// heads of the ray queue the march (next on this stream) pulls from
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);

	for (uint32_t slot = blockIdx.x * kSetupWaves + (threadIdx.x >> 6); slot < nTiles; slot += nWaves)
	{
		// Dispatch slot -> tile: cost-binned order (tileOrderKernel), or plain list order
		uint32_t tile = orderedTile(st, slot);
		tile = __builtin_amdgcn_readfirstlane(tile);

		// the work entry that owns this tile (wave-uniform)
		uint32_t entry = 0;
		for (uint32_t base = 0; base < nEntries; base += 64)
		{
			const uint32_t i = base + lane;
			const bool hit = i < nEntries && st.workTileStart[i] <= tile && tile < st.workTileStart[i + 1];
			const uint64_t b = __ballot(hit);
			if (b) { entry = base + (uint32_t)__ffsll((unsigned long long)b) - 1u; break; }
		}
		entry = __builtin_amdgcn_readfirstlane(entry);
		const uint32_t we = st.workEntries[entry];
		const uint32_t volumeId = we & 0xffffu, faceId = we >> 16;
		const uint32_t localTile = faceId == 7u ? tile - st.workTileStart[entry] : cubeLocalTile(st, volumeId, tile - st.workTileStart[entry]);

		const ushort4 vi = reinterpret_cast<const ushort4*>(st.volumeInfo)[volumeId];	// CSRayMarch.hlsl:81
		const uint32_t mipLevel = vi.x;
		const PerObject& po = reinterpret_cast<const PerObject*>(st.perObject)[volumeId];
		const bool screenTile = faceId == 7u;	// wave-uniform

		// a cube-face texel (CSRayMarch.hlsl:84-107) or a screen pixel of a RayCast volume (RayCast.hlsli:43-54)
		bool valid = false;
		vec3 rayOrigin = mul43(v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]), po.worldI);	// :89
		vec3 rayDir = v3(0.0f, 0.0f, 1.0f);
		float tMax = 0.0f;
		uint32_t out = kNoRay;
		if (!screenTile)
		{
			const uint32_t dims = cubeMipDim(st.N, mipLevel);
			const uint32_t tilesX = (dims + 7u) >> 3;
			const uint32_t px = (localTile % tilesX) * 8u + (lane & 7u), py = (localTile / tilesX) * 8u + (lane >> 3);
			if (px < dims && py < dims)
			{
				const vec3 target = GetLocalPos((float)px, (float)py, faceId, (float)dims);
				rayDir = normalize3(target - rayOrigin);
				if (ComputeRayOrigin(rayOrigin, rayDir))	// :96
				{
					tMax = ComputeTargetHit(rayOrigin, target, rayDir);
					const size_t outIndex = (size_t)st.cubeOffset[volumeId] + ((size_t)faceId * dims + py) * dims + px;

					// GetClipPos (:59-71)
					vec3 clip;
					{
						vec4 hPos;
						hPos.x = rayOrigin.x + rayDir.x * 0.01f;
						hPos.y = rayOrigin.y + rayDir.y * 0.01f;
						hPos.z = rayOrigin.z + rayDir.z * 0.01f;
						hPos.w = 1.0f;
						hPos = mul44(hPos, po.wvp);
						const float cx = hPos.x / hPos.w, cy = hPos.y / hPos.w;
						const float u = cx * 0.5f + 0.5f;
						const float v = 0.5f - cy * 0.5f;	// `1 - (y * 0.5 + 0.5)`, compiled (fast math) as 0.5 - y * 0.5
						clip = v3(cx, cy, samplePointR32F(st.depthMap, st.W, st.H, u, v));
					}
					st.cubeDepths[outIndex] = clip.z;	// :105
					tMax = __builtin_fminf(GetTMax(clip, rayOrigin, rayDir, po.wvpI), tMax);	// :106
					valid = true;
					out = (uint32_t)outIndex;	// < 2^31: mv_create bounds V x chain
				}
			}
		}
		else
		{
			// screen rectangle of the volume, clipped to the viewport, in 8x8 pixel tiles
			const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[volumeId];
			const uint32_t x1 = rect.z < st.W - 1u ? rect.z : st.W - 1u;
			const ScreenTileRows rows = screenTileRows(st, rect.y, rect.w < st.H - 1u ? rect.w : st.H - 1u);	// this context's tile rows of the rectangle (its band, or every shardCount-th)
			const uint32_t tx0 = rect.x >> 3, tw = (x1 >> 3) - tx0 + 1u;
			const uint32_t px = (tx0 + localTile % tw) * 8u + (lane & 7u), py = (rows.first + (localTile / tw) * rows.step) * 8u + (lane >> 3);
			if (px < st.W && py < st.H)
			{
				// PSCube.hlsl:37-39, then the fragment the rasteriser would have produced for this volume at this pixel
				float nx = ((float)px + 0.5f) / (float)st.W, ny = ((float)py + 0.5f) / (float)st.H;
				nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
				vec3 lpt;
				float z;
				const size_t outIndex = ((size_t)st.screenSlot[volumeId] * st.H + py) * st.W + px;
				if (findExitFragment(po, rayOrigin, pixelRayWorld(cb, nx, ny), lpt, z))
				{
					rayDir = normalize3(lpt - rayOrigin);	// PSCube.hlsl:33, 50: normalize(input.LPt - localSpaceEyePt)
					float4 zero; zero.x = zero.y = zero.z = zero.w = 0.0f;
					st.screenLayers[outIndex] = zero;	// RayCast returns 0 on a miss (RayCast.hlsli:46)
					if (ComputeRayOrigin(rayOrigin, rayDir))
					{
						const float depthPix = st.depthMap[(size_t)py * st.W + px];
						tMax = GetTMax(v3(nx, ny, depthPix), rayOrigin, rayDir, po.wvpI);	// RayCast.hlsli:50-51 (no ComputeTargetHit clamp)
						valid = true;
						out = (uint32_t)outIndex | kScreenRayBit;	// < 2^31: mv_create bounds slots x W x H
					}
				}
			}
		}

		// the tile's 64 records, one per lane (two 16-byte stores each, 2 KiB contiguous per wavefront)
		{
			u32x4 a, b;
			a.x = __float_as_uint(rayOrigin.x); a.y = __float_as_uint(rayOrigin.y); a.z = __float_as_uint(rayOrigin.z); a.w = __float_as_uint(rayDir.x);
			b.x = __float_as_uint(rayDir.y); b.y = __float_as_uint(rayDir.z); b.z = __float_as_uint(tMax); b.w = valid ? out : kNoRay;
			u32x4* dst = reinterpret_cast<u32x4*>(st.rayRecords + ((size_t)slot * 64u + lane));
			dst[0] = a; dst[1] = b;
		}
		if (lane == 0)
		{
			const VolumeConsts vc = st.volConsts[volumeId];
			TileInfo ti;
			ti.texelOff = vc.texelOff; ti.occOff = vc.occOff;
			// (from the frame's light table itself: the host may have put the previous frame's table back, mv_api.cpp keepLightTable)
			ti.lightOff = (uint32_t)(reinterpret_cast<const uint8_t*>(st.lightTable[volumeId]) - reinterpret_cast<const uint8_t*>(st.lightBuffers));
			ti.smpCount = vc.smpCount; ti.stepScale = vc.stepScale;
			ti.costKey = tileCostKey(st, volumeId, faceId, localTile, tile, nCubeTiles);
			ti.volume = volumeId;
			st.tileInfo[slot] = ti;
			st.tileCost[ti.costKey] = 0u;	// the march raises it to the sample count of the tile's longest ray
		}
	}
}

// ---------------------------------------------------------------------------------------------
// stage 2: the march
// ---------------------------------------------------------------------------------------------
// Every wavefront runs two things side by side:
//  * a LOADER (wave-uniform state machine, one step per loop iteration): it reserves the next chunk of 16 records from its queue
//    (one returning atomic, issued an iteration before its value is needed) and copies the chunk -- 16 records + the tile's
//    32-byte info -- straight into a ring in LDS with ONE LDS-DMA instruction (global_load_lds_dwordx4: lane l's 16 bytes land at
//    ring + 16 l; no VGPRs), so that records are in LDS before a lane needs them;
//  * the MARCH: one iteration of the reference loop per loop iteration for every lane that holds a ray; idle lanes take the
//    next records of the ring (ballot + popcount prefix: the active-ray compaction), which costs two LDS reads.
// A refill therefore never waits for memory; what a finished ray leaves behind is stored at once.
// WIDE = the volume pool is larger than 4 GiB (e.g. 64 x 256^3): a ray's texels are then addressed through a per-lane 64-bit
// pointer instead of pool base (SGPR) + 32-bit offset.
constexpr uint32_t kChunk = 16;				// records per reservation (a quarter of an 8x8 tile: two rows of texels)
constexpr uint32_t kChunkVec = 2 * kChunk + 2;	// uint4 per ring slot: 16 first halves, 16 second halves, tile info (2)
constexpr uint32_t kRing = 4;				// ring slots per wavefront
#ifndef MV_PREFETCH_CHUNKS
#define MV_PREFETCH_CHUNKS 3				// chunks a wavefront keeps reserved ahead of its lanes
#endif
constexpr uint32_t kStealTries = 3;			// other queues a wavefront tries once its own is empty (every queue has home wavefronts of its own)

typedef const __attribute__((address_space(1))) void* GlobalVoid;
typedef __attribute__((address_space(3))) void* LdsVoid;

template <bool WIDE>
__global__ __launch_bounds__(64 * kMarchWaves, MV_MARCH_MIN_WAVES) void marchViewKernel(DeviceState st)
{
	__shared__ u32x4 sRing[kMarchWaves][kRing][kChunkVec];	// the loader's record ring
	// What a lane's ray needs but the sample arithmetic does not hold in registers, per lane: {volume offset (lo, hi), light-map offset,
	// occupancy base | cached occupancy word (lo, hi), its index, result index, cost key | ring entry << 27}.  Register pressure decides
	// how many wavefronts a SIMD holds, and this kernel lives on latency hiding; LDS is idle otherwise.
	__shared__ u32x4 sRayA[kMarchWaves][64], sRayB[kMarchWaves][64];
	__shared__ uint2 sOutKey[kMarchWaves][64];	// per lane: where its ray's result goes; its tile's cost key | ring entry << 27
	__shared__ uint32_t sStats[kMarchWaves][8];	// per wavefront: {view samples, lit, -, -, RayCast samples, RayCast lit, -, -}
	__shared__ uint32_t sCost[kMarchWaves][kCostRing];	// running maximum of the sample counts of the rays of the tiles a wavefront has open

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	if (lane < 8u) sStats[wave][lane] = 0u;	// a row is only ever touched by its own wavefront: program order is enough
	if (lane < kCostRing) sCost[wave][lane] = 0u;

	const uint32_t nTiles = st.counters[3];	// dispatch slots of 64 records (kNoRay records included)

	const GlobalBytes volPool = (GlobalBytes)st.volPool;
	const GlobalBytes lightPool = (GlobalBytes)st.lightBuffers;
	const GlobalBytes occPool = (GlobalBytes)st.occPool;
	const uint32_t regions = occupancyRegions(st.bricks);
	const uint32_t strideY = brickPitchY(st.bricks) * 512u, strideZ = brickPitchZ(st.bricks) * 512u;
	const uint32_t N = st.N, L = st.L;

	// ---- the ray a lane holds ----
	bool active = false;
	vec3 o = v3(0.0f, 0.0f, 0.0f), d = v3(0.0f, 0.0f, 1.0f);
	float t = 0.0f, tMax = 0.0f, stepScale = 0.0f, prevDensity = 0.0f;
	vec4 scatter; scatter.x = scatter.y = scatter.z = scatter.w = 0.0f;
	uint32_t remaining = 0u;	// SmpCount - i of the loop at CSRayMarch.hlsl:117
	uint32_t nS = 0u, nL = 0u;
	bool prevLit = false;

	// ---- loader state (wave-uniform) ----
	uint32_t queue = (blockIdx.x * kMarchWaves + wave) % kRayQueues, steals = 0u;
	bool queuesDone = nTiles == 0u;	// nothing left to reserve anywhere this wavefront looks
	bool atomicPending = false;
	uint32_t atomicResult = 0u;		// lane 0: the value the pending atomic returned
	uint32_t ringW = 0u, ringR = 0u;	// chunks issued to / consumed from the ring (slot = index % kRing)
	uint32_t landed = 0u;			// chunks known to have arrived in LDS (ringR <= landed <= ringW)
	uint32_t consumed = 0u;			// records of chunk ringR already handed to lanes
	uint32_t iterations = 0u, nRays = 0u, chunksTaken = 0u;
	uint32_t nSkipped = 0u;	// per lane

	for (;;)
	{
		// ---- loader, step 1: the atomic issued one iteration ago names this wavefront's next chunk; start its copy into LDS ----
		if (atomicPending)
		{
			const uint32_t k = __builtin_amdgcn_readfirstlane(atomicResult);
			atomicPending = false;
			const uint32_t slot = (k >> 2) * kRayQueues + queue;	// the (k / 4)-th tile of this queue, quarter k % 4
			if (slot < nTiles)
			{
				const uint32_t first = slot * 64u + (k & 3u) * kChunk;
				const u32x4* src = lane < 2u * kChunk
					? reinterpret_cast<const u32x4*>(st.rayRecords + first + (lane & (kChunk - 1u))) + (lane >> 4)
					: reinterpret_cast<const u32x4*>(st.tileInfo + slot) + (lane & 1u);
				if (lane < kChunkVec) __builtin_amdgcn_global_load_lds((GlobalVoid)src, (LdsVoid)&sRing[wave][ringW % kRing][0], 16, 0, 0);
				++ringW;
			}
			else if (steals < kStealTries) { ++steals; queue = (queue + kRayQueues / (kStealTries + 1u)) % kRayQueues; }
			else queuesDone = true;
		}
		// ---- loader, step 2: keep MV_PREFETCH_CHUNKS chunks reserved ahead ----
		if (!queuesDone && ringW - ringR < (uint32_t)MV_PREFETCH_CHUNKS)
		{
			if (lane == 0u) atomicResult = atomicAdd(st.queueHeads + queue * kRayQueueStride, 1u);
			atomicPending = true;
		}

		// ---- refill: idle lanes take the next records of the ring, in lane order ----
		const uint64_t idle = ~__ballot(active);
		const uint32_t nIdle = (uint32_t)__popcll(idle);
		if (ringR != ringW && (nIdle >= (uint32_t)MV_REFILL_MIN || nIdle == 64u))
		{
			if (landed == ringR)
			{
				// chunk ringR was issued at least one iteration ago; nothing orders an LDS read behind an LDS-DMA but the issuing
				// wavefront's vmcnt (everything older -- the previous iteration's texels -- has been consumed already)
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				landed = atomicPending ? ringW : ringW;	// every copy issued so far is older than this wait
			}
			const uint32_t avail = kChunk - consumed;
			const uint32_t take = nIdle < avail ? nIdle : avail;
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));	// idle lanes below this one
			const volatile u32x4* ring = &sRing[wave][ringR % kRing][0];
			if (!active && rank < take)
			{
				const uint32_t j = consumed + rank;
				const u32x4 a = ring[j], b = ring[kChunk + j];
				const u32x4 c0 = ring[2u * kChunk], c1 = ring[2u * kChunk + 1u];
				if (b.w != kNoRay)
				{
					o = v3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
					d = v3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
					tMax = __uint_as_float(b.z);
					u32x4 ra, rb;
					ra.x = c0.x; ra.y = c0.y; ra.z = c0.z; ra.w = c0.w;	// texel offset (lo, hi), light offset, occupancy base
					rb.x = 0u; rb.y = 0u; rb.z = 0xffffffffu;			// cached occupancy word (none yet)
					rb.w = 0u;
					sRayA[wave][lane] = ra;
					sRayB[wave][lane] = rb;
					sOutKey[wave][lane] = make_uint2(b.w, c1.z | ((chunksTaken % kCostRing) << 27));
					remaining = c1.x; stepScale = __uint_as_float(c1.y);
					t = 0.0f; prevDensity = 0.0f;
					scatter.x = scatter.y = scatter.z = scatter.w = 0.0f;
					nS = 0u; nL = 0u;
					prevLit = false;
					active = true;
				}
			}
			nRays += (uint32_t)__popcll(__ballot(active && !(sOutKey[wave][lane].x & kScreenRayBit)) & idle);	// the counter covers cube-face rays, as CSRayMarch's
			consumed += take;
			if (consumed == kChunk)
			{
				consumed = 0u; ++ringR; ++chunksTaken;
				if (lane == 0u) sCost[wave][chunksTaken % kCostRing] = 0u;	// the next chunk's entry in the cost ring
			}
		}
		if (__ballot(active) == 0ull)
		{
			if (ringR == ringW && !atomicPending && queuesDone) break;
			continue;
		}
		++iterations;

		// ---- one iteration of the loop at CSRayMarch.hlsl:117-153 (RayCast.hlsli:62-102) for every lane that holds a ray ----
		if (active)
		{
			bool done = remaining == 0u;	// i < SmpCount (:117)
			if (!done)
			{
				const vec3 pos = rayAt(o, d, t);
				if (outsideBox(pos)) done = true;	// :120
				else
				{
					const vec3 uvw = LocalToTex3DSpace(pos);
					FootprintAddr fa = footprintAddr(N, strideY, strideZ, uvw);
					--remaining;

					// Empty space: a clear occupancy bit proves alpha <= ZERO_THRESHOLD, for which the loop only advances t
					// (mv_device.hpp "brick occupancy").  One 64-bit word covers a 16^3-texel region.
					const u32x4 ra = sRayA[wave][lane];
					u32x4 rb = sRayB[wave][lane];
					const uint32_t w = ra.w + occupancyWord(fa.ix, fa.iy, fa.iz, regions);
					if (w != rb.z)
					{
						const uint64_t ow = *(GlobalUint64)(occPool + ((size_t)w << 3));
						rb.x = (uint32_t)ow; rb.y = (uint32_t)(ow >> 32); rb.z = w;
						sRayB[wave][lane] = rb;
					}
					const uint64_t occWord = ((uint64_t)rb.y << 32) | rb.x;
					const uint32_t lightOff = ra.z;
					if (!((occWord >> occupancyBit(fa.ix, fa.iy, fa.iz)) & 1ull))
					{
						++nS;
						++nSkipped;
						prevLit = false;
						t += stepScale;
						if (t > tMax) done = true;	// :152
					}
					else
					{
						// the eight RGBA texels are fetched unconditionally and alpha comes from them (one round trip per sample);
						// the light quads are requested with them when the previous sample was lit (lit samples come in runs)
						ColourFootprint cf;
						if (WIDE) cf = fetchColourAt(volPool + (((uint64_t)ra.y << 32) | ra.x), fa);
						else { fa.x0 += ra.x; fa.x1 += ra.x; cf = fetchColourAt(volPool, fa); }
						LightFootprint lf;
						const bool early = prevLit;
						if (early) lf = fetchLightFootprintAt(lightPool, lightOff, L, uvw);
						const float alpha = resolveAlphaFromTexels(cf);
						++nS;
						float newStep = stepScale;
						const bool lit = alpha > kZeroThreshold;	// :128
						prevLit = lit;
						if (lit)
						{
							if (!early) lf = fetchLightFootprintAt(lightPool, lightOff, L, uvw);	// GetLight (RayMarch.hlsli:235-240)
							const vec3 light = resolveLight(lf);
							vec3 color = resolveRGB(cf);
							++nL;

							const float transm = 1.0f - scatter.w;
							const float dDensity = alpha - prevDensity;
							newStep = GetViewStep(dDensity, scatter.w, alpha, stepScale);
							prevDensity = alpha;	// (:136 `step = (step + newStep) * 0.5` is dead: overwritten at :150)

							// :140-146 as compiled: a = (transm * ABSORPTION) * alpha, scatter.rgb += (a * color) * light, scatter.a += a
							const float a = transm * kAbsorption * alpha;
							scatter.x += a * color.x * light.x;
							scatter.y += a * color.y * light.y;
							scatter.z += a * color.z * light.z;
							scatter.w += a;

							if (transm < kZeroThreshold) done = true;	// :146
						}
						if (!done)
						{
							t += newStep;
							if (t > tMax) done = true;	// :152
						}
					}
				}
			}

			if (done)
			{
				scatter.x *= kInvTwoPi; scatter.y *= kInvTwoPi; scatter.z *= kInvTwoPi;	// :155
				const uint2 ok = sOutKey[wave][lane];
				const uint32_t out = ok.x, key = ok.y;
				const bool screen = (out & kScreenRayBit) != 0u;
				if (!screen)
				{
					uint2 r;
					r.x = (uint32_t)float_to_half_bits(scatter.x) | ((uint32_t)float_to_half_bits(scatter.y) << 16);
					r.y = (uint32_t)float_to_half_bits(scatter.z) | ((uint32_t)float_to_half_bits(scatter.w) << 16);
					st.cubeMaps[out] = r;	// :157
				}
				else
				{
					float4 r; r.x = scatter.x; r.y = scatter.y; r.z = scatter.z; r.w = scatter.w;
					st.screenLayers[out & ~kScreenRayBit] = r;	// what RayCast returns (RayCast.hlsli:107); the composite applies PSCube.hlsl:56
				}
				// longest ray of the tile so far (tileOrderKernel bins the tile by it four frames on): only a new maximum is stored.
				// (A ray that outlives kCostRing chunks of its wavefront compares against another tile's maximum: that can only misorder
				// a later dispatch.)
				if (nS > atomicMax(&sCost[wave][key >> 27], nS)) st.tileCost[key & 0x7ffffffu] = nS + 1u;	// (0 = never marched)
				atomicAdd(&sStats[wave][screen ? 4 : 0], nS);
				atomicAdd(&sStats[wave][screen ? 5 : 1], nL);
				active = false;
			}
		}
	}

	// this wavefront's counter slot (the grid is fixed: a slot belongs to one wavefront; launches are stream ordered)
	const uint32_t totK = waveSumU32(nSkipped), totR = nRays;
	const uint32_t slot = blockIdx.x * kMarchWaves + wave;
	if (lane == 0u && slot < st.marchWaves)
	{
		uint64_t* s = st.waveStats + (size_t)slot * 8u;
		s[0] += sStats[wave][0]; s[1] += sStats[wave][1]; s[2] += totR; s[3] += (uint64_t)iterations * 64ull;
		s[4] += sStats[wave][4]; s[5] += sStats[wave][5]; s[6] += totK;
	}
}

void launchRaySetup(hipStream_t s, const DeviceState& st)
{
	// fixed grid, grid-stride over the device-side tile count: enough wavefronts to fill the GPU once
	hipLaunchKernelGGL(raySetupKernel, dim3(1024), dim3(64 * kSetupWaves), 0, s, st);
}

void launchMarchView(hipStream_t s, const DeviceState& st)
{
	const uint32_t blocks = (st.marchWaves + kMarchWaves - 1u) / kMarchWaves;
	if (st.volStride * st.S > 0xffffffffull) hipLaunchKernelGGL(marchViewKernel<true>, dim3(blocks), dim3(64 * kMarchWaves), 0, s, st);
	else hipLaunchKernelGGL(marchViewKernel<false>, dim3(blocks), dim3(64 * kMarchWaves), 0, s, st);
}

uint32_t marchViewResidentWaves(int device, bool wideAddressing)
{
	int perCu = 0, cus = 0;
	hipError_t e = wideAddressing
		? hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, marchViewKernel<true>, 64 * kMarchWaves, 0)
		: hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, marchViewKernel<false>, 64 * kMarchWaves, 0);
	if (e != hipSuccess || perCu <= 0) perCu = 4;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
	return (uint32_t)perCu * (uint32_t)cus * kMarchWaves;
}
} // namespace mv
