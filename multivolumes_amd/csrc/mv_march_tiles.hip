// mv_march_tiles.hip -- a3 (+ the fused screen-space RayCast) as one wavefront per 8x8 tile: the reference's thread-group tile
// (Sh/CSRayMarch.hlsl:77), ray setup and sample loop in one kernel, wavefronts dispatched by the hardware in the cost-binned
// tile order.  This is the form the context uses when a frame has only a few rays per lane of the GPU (C3: 0.66 M rays on
// 0.33 M resident lanes), where a wavefront's whole life is two or three tiles and the ray queue of mv_march.hip cannot pay for
// its record traffic and refills ("tiles or ray queue").  Same arithmetic, same results, bit for bit.
#include "mv_device.hpp"
#include "mv_kernels.hpp"

namespace mv
{
#ifndef MV_TILE_WAVES
#define MV_TILE_WAVES 2	// independent wavefronts per workgroup. Measured at C3: 1 -> 0.357 ms, 2 -> 0.352, 4 -> 0.364, 8 -> 0.416
#endif
constexpr uint32_t kTileWaves = MV_TILE_WAVES;
// (alpha comes from the eight RGBA texels themselves: testing a separate alpha structure first adds an HBM round trip to every lit
// sample -- measured 0.44 vs 0.39 ms at C3)
#ifndef MV_OCCUPANCY
#define MV_OCCUPANCY 1	// skip the texel fetch of samples whose brick occupancy bit is clear (results unchanged by construction)
#endif
#ifndef MV_PREDICT_LIT
#define MV_PREDICT_LIT 1
#endif
#ifndef MV_TILE_LUT
#define MV_TILE_LUT 1	// per-axis address tables in LDS (N <= 256); 0 = compute every tap's offsets (A/B timing)
#endif
#ifndef MV_BATCH_COLOCATE
#define MV_BATCH_COLOCATE 0	// several frames per launch: the same tile pair of every frame on the same XCD (round 6 experiment)
#endif
#ifndef MV_TILE_MIN_WAVES
#define MV_TILE_MIN_WAVES 3	// occupancy bound given to the register allocator / scheduler. Measured at C3: 3 -> 0.360 ms, 5 -> 0.395, 6 -> 0.414,
							// 7 -> 0.46, 8 -> 0.60 (spills); all of 1..5 allocate about 90 VGPRs, the looser bound schedules loads earlier
#endif

typedef const __attribute__((address_space(1))) void* GlobalVoidPtr;
typedef __attribute__((address_space(3))) void* LdsVoidPtr;


static __device__ __forceinline__ uint32_t waveSumU32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// OVERFLOW = false: the main launch, wavefront w marches dispatch slot w.  OVERFLOW = true: a small second launch that marches every
// slot from `firstSlot` on, a grid apart (the loop costs 20 VGPRs, i.e. one wavefront per SIMD, which is why the main launch has none).
// LUT = the volume's per-axis address tables live in LDS (mv_device.hpp "address tables"): used while they leave room for five
// wavefronts per SIMD (N <= kLutMaxN).
constexpr uint32_t kLutMaxN = 256;
// IL: the rows of a sharded context are dealt by tile row (kFlagInterleaveRows) -- compile-time, see screenTileRowsT
// (marchBlock: this workgroup's index among the workgroups that march)
template <bool OVERFLOW, bool LUT, bool IL, bool PEER>
__device__ __forceinline__ void marchTilesBody(const DeviceState& st, uint32_t firstSlot, u32x4* sAxis, uint32_t marchBlock, const DeviceState& orderSt);

// PEER: the peer-store transport's instantiation: a folded wait in front (mv_kernels.hpp peerFoldWait), every marched texel also stored
// into the other ranks' cube maps, the last workgroup raises "cube faces landed" (peerFoldSignal); the others are untouched by it
template <bool OVERFLOW, bool LUT, bool IL, bool PEER = false>
__global__ __launch_bounds__(64 * kTileWaves, MV_TILE_MIN_WAVES) void marchTilesKernel(DeviceState st, uint32_t firstSlot)
{
	__shared__ u32x4 sAxis[LUT ? 3 * (kLutMaxN + 1) : 1];
	// (peer-store transport, folded wait: for every rank's share of the light maps this frame reads)
	if constexpr (PEER)
	{
		// (folded) the leading workgroups of the main launch push this rank's layers of the relit light map into every rank's copy, and the
		// last of them raises "light layers landed" -- in front of the workgroups that march, which wait for every rank's (their own
		// rank's too: these very workgroups, dispatched ahead of them)
		if (!OVERFLOW && blockIdx.x < st.fold.pushBlocks)
		{
			peerPushLightRole(st, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)st.fold.pushBlocks * blockDim.x);
			peerRaise(st, st.fold.pushCounter, st.fold.pushBlocks, 1u, &st.fold.pushRow, &st.fold.pushValue, st.fold.pushCheckChain);
			peerFoldSignal(st);	// (counted among the launch's workgroups)
			return;
		}
		// (plain light push) the next rebuildBlocks workgroups wait for every rank's texels of the map relit in this frame, rebuild the
		// cells of the other ranks' layers from them and raise "cells rebuilt" locally
		if (!OVERFLOW && blockIdx.x < st.fold.pushBlocks + st.fold.rebuildBlocks)
		{
			peerFoldWait(st, 0u, 1u);
			peerRebuildLightRole(st, (size_t)(blockIdx.x - st.fold.pushBlocks) * blockDim.x + threadIdx.x, (size_t)st.fold.rebuildBlocks * blockDim.x);
			peerRaise(st, st.fold.rebuildCounter, st.fold.rebuildBlocks, 1u, &st.fold.rebuildRow, &st.fold.rebuildValue, 0xffffffffu, true);
			peerFoldSignal(st);
			return;
		}
		// (the light maps relit in the frames BEFORE this one: waited for by a kernel in front of this launch.  This frame's is waited for
		// inside the body, by the wavefronts whose tile reads the volume relit in this frame -- the others march at once, beside the push
		// above and the other ranks')
	}
	marchTilesBody<OVERFLOW, LUT, IL, PEER>(st, firstSlot, sAxis, (PEER && !OVERFLOW) ? blockIdx.x - st.fold.pushBlocks - st.fold.rebuildBlocks : blockIdx.x, st);
	if constexpr (PEER) peerFoldSignal(st);
}

// Several frames in one launch (mv_kernels.hpp DeviceStateBatch).  Always the peer-store transport's folded form: pushBlocks
// workgroups per frame push that frame's light layers, and ONE counter over all of them decides who raises "light layers landed" (to
// the launch's value: its group's) -- likewise "cube faces landed" behind the marches.
// The main launch's grid is one-dimensional, in the order workgroups are dispatched: first EVERY frame's push workgroups, then the
// marching workgroups, the frames interleaved -- a marching wavefront that waits for the light layers (the tiles that read a map relit in
// this launch) then waits for workgroups dispatched before it, whatever fills the GPU.  (The overflow launch: 64 workgroups per frame,
// blockIdx.y = the frame, firstSlots = its first dispatch slot per frame; 0xffffffff: the main grid covered that frame.)
template <bool OVERFLOW, bool LUT, bool IL>
__global__ __launch_bounds__(64 * kTileWaves, MV_TILE_MIN_WAVES) void marchTilesBatchKernel(DeviceStateBatch p, uint32_t n, uint4 firstSlots)
{
	__shared__ u32x4 sAxis[LUT ? 3 * (kLutMaxN + 1) : 1];
	if (OVERFLOW)
	{
		const DeviceState& st = p.s[blockIdx.y];
		const uint32_t first = blockIdx.y == 0u ? firstSlots.x : blockIdx.y == 1u ? firstSlots.y : blockIdx.y == 2u ? firstSlots.z : firstSlots.w;
		if (first != 0xffffffffu) marchTilesBody<true, LUT, IL, true>(st, first, sAxis, blockIdx.x, st);
		peerFoldSignal(st);
		return;
	}
	const uint32_t pushBlocks = p.s[0].fold.pushBlocks, pushAll = pushBlocks * n;
	if (blockIdx.x < pushAll)
	{
		const uint32_t frame = blockIdx.x / pushBlocks, block = blockIdx.x - frame * pushBlocks;
		const DeviceState& st = p.s[frame];
		peerPushLightRole(st, (size_t)block * blockDim.x + threadIdx.x, (size_t)pushBlocks * blockDim.x);
		peerRaise(st, st.fold.pushCounter, pushAll, 1u, &st.fold.pushRow, &st.fold.pushValue, st.fold.pushCheckChain);
		peerFoldSignal(st);	// (counted among the launch's workgroups)
		return;
	}
	const uint32_t rebuildBlocks = p.s[0].fold.rebuildBlocks, rebuildAll = rebuildBlocks * n;
	if (blockIdx.x < pushAll + rebuildAll)
	{
		// (plain light push) every frame's rebuild workgroups, behind ALL push workgroups in dispatch order; one counter over all of them
		// decides who raises "cells rebuilt" (to the launch's value: its group's)
		const uint32_t r = blockIdx.x - pushAll, frame = r / rebuildBlocks, block = r - frame * rebuildBlocks;
		const DeviceState& st = p.s[frame];
		peerFoldWait(st, 0u, 1u);
		peerRebuildLightRole(st, (size_t)block * blockDim.x + threadIdx.x, (size_t)rebuildBlocks * blockDim.x);
		peerRaise(st, st.fold.rebuildCounter, rebuildAll, 1u, &st.fold.rebuildRow, &st.fold.rebuildValue, 0xffffffffu, true);
		peerFoldSignal(st);
		return;
	}
	const uint32_t m = blockIdx.x - pushAll - rebuildAll;
#if MV_BATCH_COLOCATE != 0
	// Workgroups are dealt round-robin over the eight XCDs, each with an L2 of its own.  The frames of a launch march (nearly) the same tiles
	// through the same volumes: eight consecutive workgroups -- one per XCD -- take eight consecutive tile pairs of ONE frame, the next eight
	// the same pairs of the next frame, so the n marches of a tile pair run side by side on the SAME XCD and all but the first find the
	// texels in its L2 (the grid is a multiple of 8 n workgroups: marchTilesBlocks rounds a frame's blocks up to a multiple of eight).
	const uint32_t g = m / (8u * n), r = m - g * 8u * n, frame = r >> 3, marchBlock = g * 8u + (r & 7u);
#else
	const uint32_t marchBlock = m / n, frame = m - marchBlock * n;	// (frames interleaved workgroup by workgroup: a tile pair's n marches on n different XCDs)
#endif
	const DeviceState& st = p.s[frame];
	marchTilesBody<false, LUT, IL, true>(st, 0u, sAxis, marchBlock, MV_BATCH_COLOCATE == 2 ? p.s[0] : st);	// (2: EXPERIMENT, identical work lists only -- every frame in frame 0's tile order)
	peerFoldSignal(st);
}

template <bool OVERFLOW, bool LUT, bool IL, bool PEER>
__device__ __forceinline__ void marchTilesBody(const DeviceState& st, uint32_t firstSlot, u32x4* sAxis, uint32_t marchBlock, const DeviceState& orderSt)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t nEntries = st.counters[2], nTiles = st.counters[3];
	if (LUT)
	{
		if (!OVERFLOW && marchBlock * kTileWaves >= nTiles) return;	// a surplus workgroup: nothing to march, no table to build
		const uint32_t P = st.N + 1u, regions = occupancyRegions(st.bricks);
		const uint32_t strideY = brickPitchY(st.bricks) * 512u, strideZ = brickPitchZ(st.bricks) * 512u;
		for (uint32_t k = threadIdx.x; k < 3u * P; k += 64u * kTileWaves)
			sAxis[(k / P) * (kLutMaxN + 1u) + k % P] = axisTableEntry(k / P, k % P, st.N, strideY, strideZ, regions);
		__syncthreads();
	}

	// Dispatch slot -> tile: list order or the cost-binned order (tileOrderKernel).  Workgroups are dealt round-robin over the 8
	// XCDs, so every XCD sees an even mix of volumes and of cube / screen tiles; placement only affects speed.
	// The host sizes the main grid from the most recent tile count the cull has reported (plus a margin) without ever waiting for
	// this frame's: wavefronts beyond the real count exit at once, and if the count grew past the grid the overflow launch that
	// follows marches the rest.
	for (uint32_t slot = firstSlot + marchBlock * kTileWaves + wave; slot < nTiles; slot += OVERFLOW ? gridDim.x * kTileWaves : 0xffffffffu - slot)
	{
#ifdef MV_TILE_TIMELINE	// (diagnostic build, scripts/tile_timeline.py: per-wavefront stamps of the tiles dispatched first)
	const unsigned long long tlStart = wall_clock64();
	unsigned long long tlLoop = 0;
#endif
#ifdef MV_TILE_ROTATE	// (with it: the first 4096 dispatch slots take each other's tiles, rotated by the frame -- the same tile on eight different CUs over eight frames)
	const uint32_t orderSlot = slot < 4096u && nTiles >= 4096u ? (slot + (st.frameTag & 7u) * 74u) & 4095u : slot;
#else
	const uint32_t orderSlot = slot;
#endif
#if MV_TILE_BALANCE
	static_assert(kTileWaves == 2u, "orderedTile's dealing assumes two wavefronts per workgroup");
	const unsigned long long tileStart = wall_clock64();
#endif
	uint32_t costBin = kOrderBins;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(orderedTile(orderSt, __builtin_amdgcn_readfirstlane(orderSlot), &costBin));

	// find the work entry that owns this tile (wave-uniform)
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
	const uint32_t localTile = faceId == 7u ? tile - st.workTileStart[entry] : cubeLocalTile(st, volumeId, tile - st.workTileStart[entry]);	// (kFlagDealTiles: this rank's k-th tile of the face)

	const ushort4 vi = reinterpret_cast<const ushort4*>(st.volumeInfo)[volumeId];	// CSRayMarch.hlsl:81
	const uint32_t mipLevel = vi.x, smpCount = vi.y, volTexId = vi.w;
	const PerObject& po = reinterpret_cast<const PerObject*>(st.perObject)[volumeId];
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
	const bool screenTile = faceId == 7u;	// wave-uniform
	if constexpr (PEER)
	{
		// (folded wait, row 0: every rank's layers of the light map relit in THIS frame -- needed by the tiles of that volume only; the
		// launch's first marching wavefront waits for it whatever its tile, to compare the ranks' cull words that came with it)
		if (!OVERFLOW && st.fold.nWait)
		{
			const uint32_t count = st.counters[0];
			const uint32_t relit = count ? st.visibleVolumes[cb->frameIdx % count] : cb->frameIdx % st.V;	// CSRayMarchL.hlsl:29-33
			const bool first = marchBlock == 0u && wave == 0u;
			bool fresh = volumeId == relit;
			// (several frames per launch: the maps relit in the EARLIER frames of this launch arrive with the same row; the cull listed them)
			for (uint32_t j = 0; j < st.batchPos; ++j) fresh = fresh || volumeId == st.batchInfo[4u + j];
			// (plain light push: the tiles wait for the cells this rank's rebuild workgroups make of the texels -- row 1, local -- and the
			// launch's first marching wavefront for the ranks' texels too, row 0, whose cull words it compares)
			const uint32_t tileRow = st.fold.rebuildBlocks ? 1u : 0u;
			if (first) peerFoldWaitWave(st, 0u, tileRow + 1u, true);
			else if (fresh) peerFoldWaitWave(st, tileRow, tileRow + 1u, false);	// (wave-uniform)
		}
	}

	// ---- ray setup: a cube-face texel (CSRayMarch.hlsl:84-107) or a screen pixel of a RayCast volume (RayCast.hlsli:43-54)
	uint32_t nSamples = 0, nLit = 0, nRays = 0, nSkipped = 0;
	bool valid = false;
	vec3 rayOrigin = mul43(v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]), po.worldI);	// :89
	vec3 rayDir = v3(0.0f, 0.0f, 1.0f);
	float tMax = 0.0f;
	size_t outIndex = 0;
	float peerDepth = 0.0f;	// (PEER) what this lane stored into the cube depth
	uint2 peerColour; peerColour.x = peerColour.y = 0u;
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
				outIndex = (size_t)st.cubeOffset[volumeId] + ((size_t)faceId * dims + py) * dims + px;

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
				if constexpr (PEER) peerDepth = clip.z;
				tMax = __builtin_fminf(GetTMax(clip, rayOrigin, rayDir, po.wvpI), tMax);	// :106
				valid = true;
				nRays = 1;
			}
		}
	}
	else
	{
		// screen rectangle of the volume, clipped to the viewport, in 8x8 pixel tiles
		const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[volumeId];
		const uint32_t x1 = rect.z < st.W - 1u ? rect.z : st.W - 1u;
		const ScreenTileRows rows = screenTileRowsT<IL>(st, rect.y, rect.w < st.H - 1u ? rect.w : st.H - 1u);	// this context's tile rows of the rectangle (its band, or every shardCount-th)
		const uint32_t tx0 = rect.x >> 3, tw = (x1 >> 3) - tx0 + 1u;
		const uint32_t px = (tx0 + localTile % tw) * 8u + (lane & 7u), py = (rows.first + (localTile / tw) * rows.step) * 8u + (lane >> 3);
		if (px < st.W && py < st.H)
		{
			// PSCube.hlsl:37-39, then the fragment the rasteriser would have produced for this volume at this pixel
			float nx = ((float)px + 0.5f) / (float)st.W, ny = ((float)py + 0.5f) / (float)st.H;
			nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
			vec3 lpt;
			float z;
			outIndex = ((size_t)st.screenSlot[volumeId] * st.H + py) * st.W + px;
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
				}
			}
		}
	}

#ifdef MV_TILE_TIMELINE
	tlLoop = wall_clock64();
#endif
	if (valid)
	{
		{
			const float stepScale = kMaxDist / (float)smpCount;
			const VolumeView vol = makeVolumeView(st.volumes[volTexId], st.alphaCells[volTexId], st.N, st.bricks);
			const uint4* lightQuads = st.lightTable[volumeId];

			vec4 scatter; scatter.x = scatter.y = scatter.z = scatter.w = 0.0f;
			float t = 0.0f;
			float prevDensity = 0.0f;
			// The loop at :117-153.  One aligned 16-byte alpha-cell load decides the alpha test (:128); only lit samples
			// fetch the eight RGBA16F texels and the two light-map quads (issued together: one round trip).
			bool prevLit = false;
#if MV_OCCUPANCY
			const uint64_t* occupancy = st.occupancy[volTexId];
			const uint32_t regions = occupancyRegions(st.bricks);
			uint32_t occIndex = 0xffffffffu;
			uint64_t occWord = 0;
#endif
			for (uint32_t i = 0; i < smpCount; ++i)
			{
				const vec3 pos = rayAt(rayOrigin, rayDir, t);
				if (outsideBox(pos)) break;	// :120
				const vec3 uvw = LocalToTex3DSpace(pos);

				FootprintAddr fa;
				uint32_t occSum = 0u;
				if (LUT)
				{
					const u32x4 ex = sAxis[axisTableIndex(uvw.x, vol.N, fa.fx)];
					const u32x4 ey = sAxis[(kLutMaxN + 1u) + axisTableIndex(uvw.y, vol.N, fa.fy)];
					const u32x4 ez = sAxis[2u * (kLutMaxN + 1u) + axisTableIndex(uvw.z, vol.N, fa.fz)];
					fa.x0 = ex.x; fa.x1 = ex.y;
					fa.r00 = ey.x + ez.x; fa.r10 = ey.y + ez.x; fa.r01 = ey.x + ez.y; fa.r11 = ey.y + ez.y;
					occSum = ex.z + ey.z + ez.z;
				}
				else fa = footprintAddr(vol, uvw);
#if MV_OCCUPANCY
				// Empty space: a clear occupancy bit proves alpha <= ZERO_THRESHOLD, for which the loop below only advances t.
				// At C3 this skips the fetch of 44 % of the samples (87 % of the unlit ones): 0.325 -> 0.258 ms, HBM traffic
				// 1.66 -> 1.24 GB per launch.  One 64-bit word covers a 16^3-texel region and is reloaded (an L2 hit) when the
				// ray enters another one; requesting the next region's word ahead of time was measured slower (0.289 ms).
				{
					const uint32_t w = LUT ? occSum & 0xffffffu : occupancyWord(fa.ix, fa.iy, fa.iz, regions);
					if (w != occIndex) { occWord = occupancy[w]; occIndex = w; }
					if (!((occWord >> (LUT ? occSum >> 24 : occupancyBit(fa.ix, fa.iy, fa.iz))) & 1ull))
					{
						++nSamples;
						++nSkipped;
						prevLit = false;
						t += stepScale;
						if (t > tMax) break;	// :152
						continue;
					}
				}
#endif
				// latency-oriented form: the eight RGBA texels are fetched unconditionally and alpha comes from them
				ColourFootprint cf = fetchColour(vol, fa);
				LightFootprint lf;
				const bool early = MV_PREDICT_LIT && prevLit;
				if (early) lf = fetchLightFootprint(lightQuads, st.L, uvw);
				const float alpha = resolveAlphaFromTexels(cf);
				++nSamples;
				float newStep = stepScale;
				const bool lit = alpha > kZeroThreshold;	// :128
				prevLit = lit;
				if (lit)
				{
					if (!early) lf = fetchLightFootprint(lightQuads, st.L, uvw);
					const vec3 light = resolveLight(lf);
					vec3 color = resolveRGB(cf);
					++nLit;

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

					if (transm < kZeroThreshold) break;	// :146
				}

				t += newStep;
				if (t > tMax) break;	// :152
			}

			scatter.x *= kInvTwoPi; scatter.y *= kInvTwoPi; scatter.z *= kInvTwoPi;	// :155

			if (!screenTile)
			{
				uint2 o;
				o.x = (uint32_t)float_to_half_bits(scatter.x) | ((uint32_t)float_to_half_bits(scatter.y) << 16);
				o.y = (uint32_t)float_to_half_bits(scatter.z) | ((uint32_t)float_to_half_bits(scatter.w) << 16);
				st.cubeMaps[outIndex] = o;	// :157
				if constexpr (PEER) peerColour = o;
			}
			else
			{
				float4 o; o.x = scatter.x; o.y = scatter.y; o.z = scatter.z; o.w = scatter.w;
				st.screenLayers[outIndex] = o;	// what RayCast returns (RayCast.hlsli:107); the composite applies PSCube.hlsl:56
			}
		}
	}

	if constexpr (PEER)
	{
		// Peer-store transport: what this wavefront wrote into its own cube map and cube depth, into every other rank's -- the texels of
		// rays that hit, as the reference writes them (CSRayMarch.hlsl:105, 157), 16 bytes per store where neighbouring lanes allow it: a
		// tile row is 8 lanes = 8 texels in a row of the face, so an even lane takes its right neighbour's colour along (2 x 8 bytes) and
		// every fourth lane its three neighbours' depths (4 x 4 bytes); narrower write-through stores are a fabric write each.
		if (!screenTile && st.peerCount > 1u)	// (wave-uniform)
		{
			const bool have = valid;
			const uint2 right = make_uint2(__shfl_down(peerColour.x, 1), __shfl_down(peerColour.y, 1));
			const bool haveRight = __shfl_down(have ? 1 : 0, 1) != 0, haveLeft = __shfl_up(have ? 1 : 0, 1) != 0;
			const bool even = !(lane & 1u);
			const bool pair = even && have && haveRight && !(outIndex & 1u);	// (16-byte aligned: faces start at even texels whenever their edge is even)
			const bool single = have && !pair && !(!even && haveLeft && !((outIndex - 1u) & 1u));	// (an odd lane whose left neighbour took it along stores nothing)
			float d1 = __shfl_down(peerDepth, 1), d2 = __shfl_down(peerDepth, 2), d3 = __shfl_down(peerDepth, 3);
			const unsigned long long hits = __ballot(have);
			const bool quad = !(lane & 3u) && ((hits >> lane) & 0xfull) == 0xfull && !(outIndex & 3u);
			const bool inQuad = ((hits >> (lane & ~3u)) & 0xfull) == 0xfull && !((outIndex - (lane & 3u)) & 3u);
			for (uint32_t p = 0; p < st.peerCount; ++p)
			{
				if (p == st.shardIndex) continue;
				uint2* colours = st.peerCubeMaps[p];
				float* depths = st.peerCubeDepths[p];
				if (pair) peerStore16(colours + outIndex, make_uint4(peerColour.x, peerColour.y, right.x, right.y));
				else if (single) peerStore8(colours + outIndex, peerColour);
				if (quad) peerStore16(depths + outIndex, make_uint4(__float_as_uint(peerDepth), __float_as_uint(d1), __float_as_uint(d2), __float_as_uint(d3)));
				else if (have && !inQuad) peerStore(reinterpret_cast<uint32_t*>(depths + outIndex), __float_as_uint(peerDepth));
			}
		}
	}
	const uint32_t totS = waveSumU32(nSamples), totL = waveSumU32(nLit), totR = waveSumU32(nRays), totK = waveSumU32(nSkipped);
	uint32_t maxS = nSamples;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t m = __shfl_xor(maxS, o); maxS = m > maxS ? m : maxS; }
	const uint32_t nCubeTiles = st.counters[6];
#ifdef MV_TILE_TIMELINE
	if (lane == 0 && slot < st.marchWaves)
	{
		unsigned long long* tl = reinterpret_cast<unsigned long long*>(st.timeline) + (size_t)slot * 8u;
		uint32_t xcc, hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hw));
		tl[0] = tlStart; tl[1] = tlLoop; tl[2] = wall_clock64(); tl[3] = maxS | (costBin << 16) | (faceId << 20) | (volumeId << 24);
		tl[4] = hw | ((unsigned long long)(xcc & 0xfu) << 32); tl[5] = tile; tl[6] = totL | ((unsigned long long)totS << 32); tl[7] = totK;
	}
#endif
	if (lane == 0)
	{
#if MV_TILE_BALANCE
		{ const unsigned long long took = wall_clock64() - tileStart; st.tileCost[tileCostKeyT<IL>(st, volumeId, faceId, localTile, tile, nCubeTiles)] = took < 1ull ? 1u : took > 0xffffffull ? 0xffffffu : (uint32_t)took; }
#else
		st.tileCost[tileCostKeyT<IL>(st, volumeId, faceId, localTile, tile, nCubeTiles)] = maxS + 1u;	// (0 = never marched)
#endif
		// counters: a slot of the wave-stats table shared by the tiles that map to it (no-return atomics, once per tile)
		unsigned long long* s = reinterpret_cast<unsigned long long*>(st.waveStats) + (size_t)(slot % st.marchWaves) * 8u;
		if (!screenTile)
		{
			atomicAdd(s + 0, (unsigned long long)totS); atomicAdd(s + 1, (unsigned long long)totL); atomicAdd(s + 2, (unsigned long long)totR);
			atomicAdd(s + 3, (unsigned long long)maxS * 64ull);	// lane slots the wave occupied (utilisation = view samples / slots)
		}
		else { atomicAdd(s + 4, (unsigned long long)totS); atomicAdd(s + 5, (unsigned long long)totL); }
		if (totK) atomicAdd(s + 6, (unsigned long long)totK);
	}
	if (!OVERFLOW) break;
	}
}


// the main launch's grid for the host's estimate of the frame's tile count, and the tiles it covers: one wavefront per tile, kTileWaves
// per workgroup (-DMV_TILE_NARROW_BELOW builds: the narrow form's one workgroup per tile below that many tiles)
static uint64_t marchTilesBlocks(const DeviceState& st, uint32_t gridTiles, bool* narrow, uint64_t* covered)
{
	const uint64_t maxTiles = (uint64_t)st.maxTiles + st.maxScreenTiles;
	const uint64_t tiles = gridTiles < maxTiles ? (gridTiles ? gridTiles : 1u) : maxTiles;
#ifdef MV_TILE_NARROW_BELOW	// (an experiment: csrc/experiments/mv_march_narrow.hip, eight lanes per ray -- measured slower at every size)
	*narrow = marchTilesNarrow(st, tiles);
#else
	*narrow = false;
#endif
	const uint64_t perBlock = *narrow ? 1u : kTileWaves;
	uint64_t blocks = (tiles + perBlock - 1) / perBlock;
	blocks = ((blocks + 7) / 8) * 8;
	*covered = blocks * perBlock;
	return blocks;
}
uint32_t marchTilesBeyond(const DeviceState& st, uint32_t gridTiles)
{
	bool narrow; uint64_t covered;
	marchTilesBlocks(st, gridTiles, &narrow, &covered);
	return covered < (uint64_t)st.maxTiles + st.maxScreenTiles ? (uint32_t)covered : 0xffffffffu;
}

uint32_t launchMarchTiles(hipStream_t s, const DeviceState& st, uint32_t gridTiles)
{
	// gridTiles: the host's estimate of this frame's tile count (the cull's latest report + a margin; anything large when it has
	// none).  The grid covers min(estimate, largest possible frame); returns the first dispatch slot beyond it (what the overflow
	// launch has to cover), or 0xffffffff if the grid covers every frame there can be.
	bool narrow; uint64_t covered;
	const uint64_t blocks = marchTilesBlocks(st, gridTiles, &narrow, &covered);
#ifdef MV_TILE_NARROW_BELOW
	if (narrow) { launchMarchTilesNarrow(s, st, (uint32_t)blocks); return marchTilesBeyond(st, gridTiles); }
#endif
	const bool il = st.flags & kFlagInterleaveRows;
	const bool peer = st.peerCount > 1u && st.fold.stores;
	const dim3 grid((uint32_t)blocks + (peer ? st.fold.pushBlocks + st.fold.rebuildBlocks : 0u)), block(64 * kTileWaves);
	const bool lut = st.N <= kLutMaxN && MV_TILE_LUT != 0;
	if (st.peerCount > 1u && st.fold.stores)	// (peer-store transport)
	{
		if (lut) { if (il) hipLaunchKernelGGL((marchTilesKernel<false, true, true, true>), grid, block, 0, s, st, 0u); else hipLaunchKernelGGL((marchTilesKernel<false, true, false, true>), grid, block, 0, s, st, 0u); }
		else if (il) hipLaunchKernelGGL((marchTilesKernel<false, false, true, true>), grid, block, 0, s, st, 0u);
		else hipLaunchKernelGGL((marchTilesKernel<false, false, false, true>), grid, block, 0, s, st, 0u);
	}
	else if (lut)
	{
		if (il) hipLaunchKernelGGL((marchTilesKernel<false, true, true>), grid, block, 0, s, st, 0u);
		else hipLaunchKernelGGL((marchTilesKernel<false, true, false>), grid, block, 0, s, st, 0u);
	}
	else if (il) hipLaunchKernelGGL((marchTilesKernel<false, false, true>), grid, block, 0, s, st, 0u);
	else hipLaunchKernelGGL((marchTilesKernel<false, false, false>), grid, block, 0, s, st, 0u);
	return marchTilesBeyond(st, gridTiles);
}

void launchMarchTilesBatch(hipStream_t s, const DeviceState* states, uint32_t n, const uint32_t* gridTiles, uint32_t* beyond)
{
	DeviceStateBatch p;
	uint64_t blocks = 0;
	for (uint32_t i = 0; i < kMaxBatch; ++i)
	{
		p.s[i] = states[i < n ? i : n - 1u];
		if (i >= n) continue;
		bool narrow; uint64_t covered;
		const uint64_t b = marchTilesBlocks(states[i], gridTiles[i], &narrow, &covered);
		blocks = b > blocks ? b : blocks;
	}
	const uint64_t most = (uint64_t)states[0].maxTiles + states[0].maxScreenTiles;
	for (uint32_t i = 0; i < n; ++i) beyond[i] = blocks * kTileWaves < most ? (uint32_t)(blocks * kTileWaves) : 0xffffffffu;
	const bool il = states[0].flags & kFlagInterleaveRows;
	const bool lut = states[0].N <= kLutMaxN && MV_TILE_LUT != 0;
	const dim3 grid(((uint32_t)blocks + states[0].fold.pushBlocks + states[0].fold.rebuildBlocks) * n), block(64 * kTileWaves);
	const uint4 none = make_uint4(0u, 0u, 0u, 0u);
	if (lut) { if (il) hipLaunchKernelGGL((marchTilesBatchKernel<false, true, true>), grid, block, 0, s, p, n, none); else hipLaunchKernelGGL((marchTilesBatchKernel<false, true, false>), grid, block, 0, s, p, n, none); }
	else if (il) hipLaunchKernelGGL((marchTilesBatchKernel<false, false, true>), grid, block, 0, s, p, n, none);
	else hipLaunchKernelGGL((marchTilesBatchKernel<false, false, false>), grid, block, 0, s, p, n, none);
}

void launchMarchTilesOverflowBatch(hipStream_t s, const DeviceState* states, uint32_t n, const uint32_t* firstSlot)
{
	DeviceStateBatch p;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i] = states[i < n ? i : n - 1u];
	const uint4 first = make_uint4(firstSlot[0], n > 1u ? firstSlot[1] : 0xffffffffu, n > 2u ? firstSlot[2] : 0xffffffffu, n > 3u ? firstSlot[3] : 0xffffffffu);
	const bool il = states[0].flags & kFlagInterleaveRows;
	if (il) hipLaunchKernelGGL((marchTilesBatchKernel<true, false, true>), dim3(64, n), dim3(64 * kTileWaves), 0, s, p, n, first);
	else hipLaunchKernelGGL((marchTilesBatchKernel<true, false, false>), dim3(64, n), dim3(64 * kTileWaves), 0, s, p, n, first);
}

void launchMarchTilesOverflow(hipStream_t s, const DeviceState& st, uint32_t firstSlot)
{
	const bool il = st.flags & kFlagInterleaveRows;
	if (st.peerCount > 1u && st.fold.stores)
	{
		if (il) hipLaunchKernelGGL((marchTilesKernel<true, false, true, true>), dim3(64), dim3(64 * kTileWaves), 0, s, st, firstSlot);
		else hipLaunchKernelGGL((marchTilesKernel<true, false, false, true>), dim3(64), dim3(64 * kTileWaves), 0, s, st, firstSlot);
	}
	else if (il) hipLaunchKernelGGL((marchTilesKernel<true, false, true>), dim3(64), dim3(64 * kTileWaves), 0, s, st, firstSlot);
	else hipLaunchKernelGGL((marchTilesKernel<true, false, false>), dim3(64), dim3(64 * kTileWaves), 0, s, st, firstSlot);
}
} // namespace mv
