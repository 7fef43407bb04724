// experiments/mv_march_tiles_r05.hip -- the REGISTER-DIET form of the tile march (79 VGPRs = six wavefronts per SIMD: measured, not shipped) with
// the round-5 experiments' switches, built INSTEAD of
// it by csrc/build.sh when one of them is on its command line (measured, rejected: profiles/r05_instruction_cuts.txt):
//   -DMV_EMPTY_REGIONS=1       advance through all-empty 16^3 regions with the bare `t += stepScale` (emptyRegionEnd below)
//   -DMV_EMULATE_LIGHT_PAIRS   TIMING ONLY: the load and instruction stream of light cells kept as fp16 pairs (32-byte cells); images are garbage
//   -DMV_TILE_LDS_PAD=bytes    extra LDS per workgroup: the march at fewer wavefronts per SIMD
//   -DMV_DIAG_ITER             wave-iteration census into the timeline buffer (scripts/exp_wave_iterations.py)
// mv_march_tiles.hip -- a3 (+ the fused screen-space RayCast) as one wavefront per 8x8 tile: the reference's thread-group tile
// (Sh/CSRayMarch.hlsl:77), ray setup and sample loop in one kernel, wavefronts dispatched by the hardware in the cost-binned
// tile order.  This is the form the context uses when a frame has only a few rays per lane of the GPU (C3: 0.66 M rays on
// 0.33 M resident lanes), where a wavefront's whole life is two or three tiles and the ray queue of mv_march.hip cannot pay for
// its record traffic and refills ("tiles or ray queue").  Same arithmetic, same results, bit for bit.
#include "mv_device.hpp"
#include "mv_kernels.hpp"
// (built with -I csrc)

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
#ifndef MV_EMPTY_REGIONS
#define MV_EMPTY_REGIONS 0	// (measured: profiles/r05_instruction_cuts.txt) advance through 16^3 regions without an occupied brick with the bare `t += stepScale` (mv_device.hpp emptyRegionEnd)
#endif
#ifndef MV_TILE_MIN_WAVES
#define MV_TILE_MIN_WAVES 3	// occupancy bound given to the register allocator / scheduler. Measured at C3: 3 -> 0.360 ms, 5 -> 0.395, 6 -> 0.414,
							// 7 -> 0.46, 8 -> 0.60 (spills); all of 1..5 allocate about 90 VGPRs, the looser bound schedules loads earlier
#endif


// ---- round-5 experiments' device functions (not in the product headers) ----
// A 64-bit occupancy word that is ZERO says more than one sample's bit: every sample whose first taps lie in that 16^3-texel region is
// fetch-free, so until the ray leaves the region CSRayMarch.hlsl:117-153 does nothing but `t += stepScale` and its two exit tests.
// emptyRegionEnd returns a ray parameter below which the ray's samples provably (a) have their first taps in the region of the sample at
// `t` and (b) lie inside the box (the loop's `any(abs(pos) > 1)` test cannot fire): the march advances through them with the bare
// addition -- the same additions, the same counters -- instead of an address computation and an occupancy lookup per sample.
// In region units r = (u N - 0.5) / 16 per axis the region of the sample is [floor(r), floor(r) + 1), the box [-0.5, N - 0.5] / 16 (taps
// clamp: a coordinate below the first or beyond the last texel centre still belongs to the edge region); the exit distance along each
// axis, shortened by kEmptyMargin = 1/64 texel, over the ray's speed along it -- v_rcp_f32 is enough, the margin is a thousand times the
// rounding of everything here and of the loop's own `o + d t` (a few 1e-5 texels at N = 256).  Conservative is all it has to be: a
// sample left out takes the ordinary path, finds the same zero word and counts the same.
constexpr float kEmptyMargin = 1.0f / 1024.0f;	// in regions of 16 texels
__device__ __forceinline__ float emptyRegionEnd(vec3 uvw, vec3 d, float t, uint32_t N)
{
	const float n = (float)N, speed = n * (0.5f / 16.0f);	// regions per unit of t and unit direction component (pos = o + d t, u = pos / 2 + 1/2)
	const float boxLo = -0.5f / 16.0f, boxHi = (n - 0.5f) * 0.0625f;
	float tEnd = kFltMax;
#pragma unroll
	for (int a = 0; a < 3; ++a)
	{
		const float da = comp(d, a);
		const float r = (comp(uvw, a) * n - 0.5f) * 0.0625f;
		const float rl = __builtin_floorf(r);
		const float up = __builtin_fminf(rl + 1.0f, boxHi) - r, down = r - __builtin_fmaxf(rl, boxLo);
		const float dist = (da >= 0.0f ? up : down) - kEmptyMargin;
		tEnd = __builtin_fminf(tEnd, dist * __builtin_amdgcn_rcpf(__builtin_fabsf(da) * speed));	// (da = 0: +inf, or NaN when dist = 0 -- ignored by fmin)
	}
	return t + tEnd;
}


// TIMING EXPERIMENT ONLY: the instruction and load stream the view march would have with light cells kept as fp16
// pairs -- 32 bytes per cell {RG(x,y), RG(x+1,y), RG(x,y+1), RG(x+1,y+1) | B ...}, four 16-byte loads per footprint, no 11/10-bit decode.
// The buffers hold no such cells (the light march still writes 16-byte ones into a buffer allocated twice the size: mv_api.cpp under the
// same flag): the light values, hence the images, of such a build are garbage; the march's control flow (alpha only) and its counters are not.
struct LightPairFootprint { u32x4 q0, q1, b0, b1; float cx, cy, fz; };
__device__ __forceinline__ LightPairFootprint fetchLightPairs(const uint4* quads, uint32_t L, vec3 uvw, float fL, float fLp1)
{
	const GlobalBytes pool = (GlobalBytes)quads;
	const Taps tx = linearTaps(uvw.x, L, 0, fL, fLp1), ty = linearTaps(uvw.y, L, 0, fL, fLp1), tz = linearTaps(uvw.z, L, 0, fL, fLp1);
	const uint32_t row = (__umul24((uint32_t)ty.i0, L) + (uint32_t)tx.i0) << 4;
	const uint32_t slice = (L * L) << 4;
	const uint32_t a0 = (__umul24((uint32_t)tz.i0, slice) + row) << 1, a1 = (__umul24((uint32_t)tz.i1, slice) + row) << 1;
	LightPairFootprint f;
	f.q0 = *(GlobalUint4)(pool + a0); f.b0 = *(GlobalUint4)(pool + a0 + 16u);
	f.q1 = *(GlobalUint4)(pool + a1); f.b1 = *(GlobalUint4)(pool + a1 + 16u);
	f.cx = tx.i1 == tx.i0 ? 0.0f : tx.f; f.cy = ty.i1 == ty.i0 ? 0.0f : ty.f; f.fz = tz.f;
	return f;
}
__device__ __forceinline__ vec3 resolveLightPairs(const LightPairFootprint& f)
{
	vec3 o;
	o.x = lerpf(lerpf(lerpHalves<0, 0>(f.q0.x, f.q0.y, f.cx), lerpHalves<0, 0>(f.q0.z, f.q0.w, f.cx), f.cy),
		lerpf(lerpHalves<0, 0>(f.q1.x, f.q1.y, f.cx), lerpHalves<0, 0>(f.q1.z, f.q1.w, f.cx), f.cy), f.fz);
	o.y = lerpf(lerpf(lerpHalves<1, 1>(f.q0.x, f.q0.y, f.cx), lerpHalves<1, 1>(f.q0.z, f.q0.w, f.cx), f.cy),
		lerpf(lerpHalves<1, 1>(f.q1.x, f.q1.y, f.cx), lerpHalves<1, 1>(f.q1.z, f.q1.w, f.cx), f.cy), f.fz);
	o.z = lerpf(lerpf(lerpHalves<0, 0>(f.b0.x, f.b0.y, f.cx), lerpHalves<0, 0>(f.b0.z, f.b0.w, f.cx), f.cy),
		lerpf(lerpHalves<0, 0>(f.b1.x, f.b1.y, f.cx), lerpHalves<0, 0>(f.b1.z, f.b1.w, f.cx), f.cy), f.fz);
	return o;
}
#ifdef MV_EMULATE_LIGHT_PAIRS
#define MV_LIGHT_FOOTPRINT LightPairFootprint
#define MV_FETCH_LIGHT fetchLightPairs
#define MV_RESOLVE_LIGHT resolveLightPairs
#else
#define MV_LIGHT_FOOTPRINT LightFootprint
#define MV_FETCH_LIGHT fetchLightFootprint
#define MV_RESOLVE_LIGHT resolveLight
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
__device__ __forceinline__ void marchTilesBody(const DeviceState& st, uint32_t firstSlot, u32x4* sAxis, uint32_t marchBlock);

// PEER: the peer-store transport's instantiation: a folded wait in front (mv_kernels.hpp peerFoldWait), every marched texel also stored
// into the other ranks' cube maps, the last workgroup raises "cube faces landed" (peerFoldSignal); the others are untouched by it
template <bool OVERFLOW, bool LUT, bool IL, bool PEER = false>
__global__ __launch_bounds__(64 * kTileWaves, MV_TILE_MIN_WAVES) void marchTilesKernel(DeviceState st, uint32_t firstSlot)
{
#ifdef MV_TILE_LDS_PAD	// (experiment: extra LDS per workgroup, to measure the march at FEWER wavefronts per SIMD -- what a wavefront more would be worth)
	__shared__ u32x4 sAxis[(LUT ? 3 * (kLutMaxN + 1) : 1) + MV_TILE_LDS_PAD / 16];
#else
	__shared__ u32x4 sAxis[LUT ? 3 * (kLutMaxN + 1) : 1];
#endif
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
		// every rank's layers of the light maps relit in the frames BEFORE this one (rows 1 ..): long landed.  This frame's (row 0) is
		// waited for inside the body, by the workgroups whose tile reads the volume relit in this frame -- the others march at once,
		// beside the push above and the other ranks'
		peerFoldWait(st, 1u);
	}
	marchTilesBody<OVERFLOW, LUT, IL, PEER>(st, firstSlot, sAxis, (PEER && !OVERFLOW) ? blockIdx.x - st.fold.pushBlocks : blockIdx.x);
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
		if (first != 0xffffffffu) marchTilesBody<true, LUT, IL, true>(st, first, sAxis, blockIdx.x);
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
	const uint32_t m = blockIdx.x - pushAll, marchBlock = m / n;
	const DeviceState& st = p.s[m - marchBlock * n];
	peerFoldWait(st, 1u);
	marchTilesBody<false, LUT, IL, true>(st, 0u, sAxis, marchBlock);
	peerFoldSignal(st);
}

template <bool OVERFLOW, bool LUT, bool IL, bool PEER>
__device__ __forceinline__ void marchTilesBody(const DeviceState& st, uint32_t firstSlot, u32x4* sAxis, uint32_t marchBlock)
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
#ifdef MV_TILE_TIMELINE
	const unsigned long long tlStart = wall_clock64();
	unsigned long long tlLoop = 0;
#endif
#ifdef MV_TILE_ROTATE	// (with it: the first 4096 dispatch slots take each other's tiles, rotated by the frame -- the same tile on eight different CUs over eight frames)
	const uint32_t orderSlot = slot < 4096u && nTiles >= 4096u ? (slot + (st.frameTag & 7u) * 74u) & 4095u : slot;
#else
	const uint32_t orderSlot = slot;
#endif
	uint32_t costBin = 8u;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(orderedTile(st, __builtin_amdgcn_readfirstlane(orderSlot), &costBin));

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
			if (fresh || first) peerFoldWaitWave(st, 0u, 1u, first);	// (wave-uniform)
		}
	}

	// ---- ray setup: a cube-face texel (CSRayMarch.hlsl:84-107) or a screen pixel of a RayCast volume (RayCast.hlsli:43-54)
	// (register diet: the loop runs at the wavefronts per SIMD its registers allow -- 80 for six.  What the loop does not need is not carried
	// through it: the output index is derived again at the stores, lit and fetch-free samples count in the two halves of one register
	// (SmpCount is 16 bits: VolumeInfo), a ray is counted by `valid`)
	uint32_t nSamples = 0, nLitSkipped = 0;
	bool valid = false;
	vec3 rayOrigin = mul43(v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]), po.worldI);	// :89
	vec3 rayDir = v3(0.0f, 0.0f, 1.0f);
	float tMax = 0.0f;
	uint2 peerColour; peerColour.x = peerColour.y = 0u;
	// where this lane's result goes: a texel of cubeMaps / cubeDepths, or of screenLayers
	auto outIndexOf = [&]() -> size_t
	{
		if (!screenTile)
		{
			const uint32_t dims = cubeMipDim(st.N, mipLevel);
			const uint32_t tilesX = (dims + 7u) >> 3;
			const uint32_t px = (localTile % tilesX) * 8u + (lane & 7u), py = (localTile / tilesX) * 8u + (lane >> 3);
			return (size_t)st.cubeOffset[volumeId] + ((size_t)faceId * dims + py) * dims + px;
		}
		const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[volumeId];
		const uint32_t x1 = rect.z < st.W - 1u ? rect.z : st.W - 1u;
		const ScreenTileRows rows = screenTileRowsT<IL>(st, rect.y, rect.w < st.H - 1u ? rect.w : st.H - 1u);
		const uint32_t tx0 = rect.x >> 3, tw = (x1 >> 3) - tx0 + 1u;
		const uint32_t px = (tx0 + localTile % tw) * 8u + (lane & 7u), py = (rows.first + (localTile / tw) * rows.step) * 8u + (lane >> 3);
		return ((size_t)st.screenSlot[volumeId] * st.H + py) * st.W + px;
	};
	if (!screenTile)
	{
		const uint32_t dims = cubeMipDim(st.N, mipLevel);
		const uint32_t tilesX = (dims + 7u) >> 3;
		const uint32_t px = (localTile % tilesX) * 8u + (lane & 7u), py = (localTile / tilesX) * 8u + (lane >> 3);
		float peerDepth = 0.0f;	// (PEER) what this lane stored into the cube depth
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
				if constexpr (PEER) peerDepth = clip.z;
				tMax = __builtin_fminf(GetTMax(clip, rayOrigin, rayDir, po.wvpI), tMax);	// :106
				valid = true;
			}
		}
		if constexpr (PEER)
		{
			// Peer-store transport: the cube depth this wavefront wrote (CSRayMarch.hlsl:105), into every other rank's cube depths -- here, in
			// front of the march, which does not touch it: 16 bytes per store where four neighbouring lanes of a tile row hit (4 x 4 bytes),
			// single words otherwise (narrower write-through stores are a fabric write each)
			if (st.peerCount > 1u)	// (wave-uniform)
			{
				const size_t outIndex = outIndexOf();
				const float d1 = __shfl_down(peerDepth, 1), d2 = __shfl_down(peerDepth, 2), d3 = __shfl_down(peerDepth, 3);
				const unsigned long long hits = __ballot(valid);
				const bool quad = !(lane & 3u) && ((hits >> lane) & 0xfull) == 0xfull && !(outIndex & 3u);
				const bool inQuad = ((hits >> (lane & ~3u)) & 0xfull) == 0xfull && !((outIndex - (lane & 3u)) & 3u);
				for (uint32_t p = 0; p < st.peerCount; ++p)
				{
					if (p == st.shardIndex) continue;
					float* depths = st.peerCubeDepths[p];
					if (quad) peerStore16(depths + outIndex, make_uint4(__float_as_uint(peerDepth), __float_as_uint(d1), __float_as_uint(d2), __float_as_uint(d3)));
					else if (valid && !inQuad) peerStore(reinterpret_cast<uint32_t*>(depths + outIndex), __float_as_uint(peerDepth));
				}
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
			// (what the whole wavefront shares goes to scalar registers by hand -- pointers a wave-uniform index loaded; float constants come as
			// kernel arguments: the compiler kept fourteen such values in vector registers through the loop, which runs at the wavefronts its registers allow)
			const float stepScale = st.volConsts[volumeId].stepScale;	// g_maxDist / SmpCount (:109), divided by the cull: a scalar load
			const float fN = st.fN, fNm1 = st.fNm1, fL = st.fL, fLp1 = st.fLp1;
			const VolumeView vol = makeVolumeView(uniformPtr(st.volumes[volTexId]), uniformPtr(st.alphaCells[volTexId]), st.N, st.bricks);
			const uint4* lightQuads = uniformPtr(st.lightTable[volumeId]);
#ifdef MV_EMULATE_LIGHT_PAIRS
			lightQuads = st.lightBuffers + 2 * (lightQuads - st.lightBuffers);	// (timing experiment: the map's cells 32 bytes apart in a buffer twice the size)
#endif

			vec4 scatter; scatter.x = scatter.y = scatter.z = scatter.w = 0.0f;
			float t = 0.0f;
			float prevDensity = 0.0f;
			// The loop at :117-153.  One aligned 16-byte alpha-cell load decides the alpha test (:128); only lit samples
			// fetch the eight RGBA16F texels and the two light-map quads (issued together: one round trip).
			bool prevLit = false;
#ifdef MV_DIAG_ITER
			uint32_t dAllSkip = 0, dAllZero = 0, dLanes = 0;
#endif
#if MV_OCCUPANCY
			const GlobalBytes occupancy = (GlobalBytes)uniformPtr(st.occupancy[volTexId]);	// (scalar base + 32-bit offset: N <= 640 keeps the words below 2^20)
			const uint32_t regions = occupancyRegions(st.bricks);
			uint32_t occIndex = 0xffffffffu;
			uint64_t occWord = 0;
#endif
#if MV_OCCUPANCY && MV_EMPTY_REGIONS
			float tEmptyEnd = 0.0f;	// below it the ray is inside a 16^3 region without an occupied brick, and inside the box (emptyRegionEnd)
#endif
			for (uint32_t i = 0; i < smpCount; ++i)
			{
#if MV_OCCUPANCY && MV_EMPTY_REGIONS
				// Through an all-empty region: what :117-153 does for a sample whose alpha cannot pass :128 -- count it, advance, test :152 --
				// without the address arithmetic that would only find the same zero word again (the box test :120 cannot fire before tEmptyEnd)
				if (t < tEmptyEnd)
				{
					++nSamples;
					nLitSkipped += 0x10000u;
					prevLit = false;
					t += stepScale;
					if (t > tMax) break;	// :152
					continue;
				}
#endif
				const vec3 pos = rayAt(rayOrigin, rayDir, t);
				if (outsideBox(pos)) break;	// :120
				const vec3 uvw = LocalToTex3DSpace(pos);

				FootprintAddr fa;
				uint32_t occSum = 0u;
				if (LUT)
				{
					const u32x4 ex = sAxis[axisTableIndex(uvw.x, fN, fNm1, fa.fx)];
					const u32x4 ey = sAxis[(kLutMaxN + 1u) + axisTableIndex(uvw.y, fN, fNm1, fa.fy)];
					const u32x4 ez = sAxis[2u * (kLutMaxN + 1u) + axisTableIndex(uvw.z, fN, fNm1, fa.fz)];
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
					if (w != occIndex) { occWord = *(GlobalUint64)(occupancy + (w << 3)); occIndex = w; }
#ifdef MV_DIAG_ITER	// (diagnostic build, scripts/exp_wave_iterations.py: how many wave-iterations have no lane that fetches, and how many of those only zero words)
					{
						const bool dSkip = !((occWord >> (LUT ? occSum >> 24 : occupancyBit(fa.ix, fa.iy, fa.iz))) & 1ull);
						const unsigned long long act = __ballot(1), sk = __ballot(dSkip), zr = __ballot(dSkip && !occWord);
						dAllSkip += sk == act ? 1u : 0u; dAllZero += zr == act ? 1u : 0u;
						dLanes += sk == act ? (uint32_t)__popcll(act) : 0u;
					}
#endif
					if (!((occWord >> (LUT ? occSum >> 24 : occupancyBit(fa.ix, fa.iy, fa.iz))) & 1ull))
					{
#if MV_EMPTY_REGIONS
						if (!occWord) tEmptyEnd = emptyRegionEnd(uvw, rayDir, t, vol.N);
#endif
						++nSamples;
						nLitSkipped += 0x10000u;
						prevLit = false;
						t += stepScale;
						if (t > tMax) break;	// :152
						continue;
					}
				}
#endif
				// latency-oriented form: the eight RGBA texels are fetched unconditionally and alpha comes from them
				ColourFootprint cf = fetchColour(vol, fa);
				MV_LIGHT_FOOTPRINT lf;
				const bool early = MV_PREDICT_LIT && prevLit;
				if (early) lf = MV_FETCH_LIGHT(lightQuads, st.L, uvw, fL, fLp1);
				const float alpha = resolveAlphaFromTexels(cf);
				++nSamples;
				float newStep = stepScale;
				const bool lit = alpha > kZeroThreshold;	// :128
				prevLit = lit;
				if (lit)
				{
					if (!early) lf = MV_FETCH_LIGHT(lightQuads, st.L, uvw, fL, fLp1);
					const vec3 light = MV_RESOLVE_LIGHT(lf);
					vec3 color = resolveRGB(cf);
					++nLitSkipped;

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
#ifdef MV_DIAG_ITER
			if (!screenTile)
			{
				// every lane saw the same ballots while it lived: the longest-lived lane holds the tile's counts
				unsigned long long key = ((unsigned long long)nSamples << 32) | lane, best = key;
				for (int o = 32; o > 0; o >>= 1) { const unsigned long long m = __shfl_xor(best, o); best = m > best ? m : best; }
				if (key == best)
				{
					unsigned long long* tl = reinterpret_cast<unsigned long long*>(st.timeline) + (size_t)(slot % st.marchWaves) * 8u;
					atomicAdd(tl + 0, (unsigned long long)nSamples); atomicAdd(tl + 1, (unsigned long long)dAllSkip); atomicAdd(tl + 2, (unsigned long long)dAllZero); atomicAdd(tl + 3, (unsigned long long)dLanes);
				}
			}
#endif

			const size_t outIndex = outIndexOf();
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
		// Peer-store transport: the texel this wavefront wrote into its own cube map (CSRayMarch.hlsl:157), into every other rank's, 16 bytes
		// per store where neighbouring lanes allow it: a tile row is 8 lanes = 8 texels in a row of the face, so an even lane takes its right
		// neighbour's colour along (2 x 8 bytes); narrower write-through stores are a fabric write each.  (The depths went out in front of the march.)
		if (!screenTile && st.peerCount > 1u)	// (wave-uniform)
		{
			const size_t outIndex = outIndexOf();
			const bool have = valid;
			const uint2 right = make_uint2(__shfl_down(peerColour.x, 1), __shfl_down(peerColour.y, 1));
			const bool haveRight = __shfl_down(have ? 1 : 0, 1) != 0, haveLeft = __shfl_up(have ? 1 : 0, 1) != 0;
			const bool even = !(lane & 1u);
			const bool pair = even && have && haveRight && !(outIndex & 1u);	// (16-byte aligned: faces start at even texels whenever their edge is even)
			const bool single = have && !pair && !(!even && haveLeft && !((outIndex - 1u) & 1u));	// (an odd lane whose left neighbour took it along stores nothing)
			for (uint32_t p = 0; p < st.peerCount; ++p)
			{
				if (p == st.shardIndex) continue;
				uint2* colours = st.peerCubeMaps[p];
				if (pair) peerStore16(colours + outIndex, make_uint4(peerColour.x, peerColour.y, right.x, right.y));
				else if (single) peerStore8(colours + outIndex, peerColour);
			}
		}
	}
	const uint32_t totS = waveSumU32(nSamples), totL = waveSumU32(nLitSkipped & 0xffffu), totR = (uint32_t)__popcll(__ballot(valid && !screenTile)), totK = waveSumU32(nLitSkipped >> 16);
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
		st.tileCost[tileCostKeyT<IL>(st, volumeId, faceId, localTile, tile, nCubeTiles)] = maxS + 1u;	// (0 = never marched)
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
	const dim3 grid((uint32_t)blocks + (peer ? st.fold.pushBlocks : 0u)), block(64 * kTileWaves);
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
	const dim3 grid(((uint32_t)blocks + states[0].fold.pushBlocks) * n), block(64 * kTileWaves);
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
