// experiments/mv_march_tiles_experiments.hip -- the tile march WITH its measured-and-rejected variants (LDS brick staging, speculation,
// s_setprio by cost bin: profiles/r03_lds_staging.txt, r03_speculation.txt, r03_view_march_tail.txt).  Not part of the product: csrc/build.sh
// compiles this file in place of mv_march_tiles.hip only when one of -DMV_TILE_STAGE / -DMV_TILE_SPECULATE_BELOW / -DMV_TILE_PRIORITY is
// on its command line (scripts/check_switches.sh keeps them building and bit-identical).  No peer-store instantiation here.
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
#ifndef MV_TILE_MIN_WAVES
#define MV_TILE_MIN_WAVES 3	// occupancy bound given to the register allocator / scheduler. Measured at C3: 3 -> 0.360 ms, 5 -> 0.395, 6 -> 0.414,
							// 7 -> 0.46, 8 -> 0.60 (spills); all of 1..5 allocate about 90 VGPRs, the looser bound schedules loads earlier
#endif

typedef const __attribute__((address_space(1))) void* GlobalVoidPtr;
typedef __attribute__((address_space(3))) void* LdsVoidPtr;

#ifndef MV_TILE_PRIORITY
#define MV_TILE_PRIORITY 0	// s_setprio by cost bin: measured (profiles/r03_view_march_tail.txt), no effect -- the long tiles wait for memory, not for issue slots
#endif
#ifndef MV_TILE_PRIORITY_BINS
#define MV_TILE_PRIORITY_BINS 2	// bins 0 .. n - 1 at priority 3, the next n at 1, the rest at 0
#endif

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
// SPEC: the texels of the NEXT sample are requested before the current one is resolved (see "speculation" inside): for launches that
// leave most of the GPU's wavefront slots empty, where a tile lasts as long as its longest ray's chain of dependent samples.
// STAGE: the volume bricks a tile's 64 footprints of one step lie in are copied into LDS first (LDS-DMA) and the taps read from there
// (see "staging" inside): the north star's "LDS-staged volume bricks", built to be measured.
constexpr uint32_t kStageBricks = 32;	// bricks (512 B each) a wavefront stages per step: 16 KiB of LDS per wavefront
template <bool OVERFLOW, bool LUT, bool IL, bool SPEC = false, bool STAGE = false>
__global__ __launch_bounds__(64 * kTileWaves, MV_TILE_MIN_WAVES) void marchTilesKernel(DeviceState st, uint32_t firstSlot)
{
	__shared__ u32x4 sAxis[LUT ? 3 * (kLutMaxN + 1) : 1];
	__shared__ u32x4 sStage[STAGE ? kTileWaves * kStageBricks * 32u : 1];	// [wavefront][brick][32 x 16 bytes]
	__shared__ uint32_t sBox[STAGE ? kTileWaves * 8u : 1];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t nEntries = st.counters[2], nTiles = st.counters[3];
	if (LUT)
	{
		if (!OVERFLOW && blockIdx.x * kTileWaves >= nTiles) return;	// a surplus workgroup: nothing to march, no table to build
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
	for (uint32_t slot = firstSlot + blockIdx.x * kTileWaves + wave; slot < nTiles; slot += OVERFLOW ? gridDim.x * kTileWaves : 0xffffffffu - slot)
	{
	uint32_t costBin = 8u;
	const uint32_t tile = __builtin_amdgcn_readfirstlane(orderedTile(st, __builtin_amdgcn_readfirstlane(slot), &costBin));
#if MV_TILE_PRIORITY
	// The launch lasts as long as its longest rays: the tiles of the first cost bins (dispatched first) are on its critical path from
	// the first microsecond, while the short tiles only have to fit in somewhere.  A wavefront of a long tile therefore raises its
	// issue priority (s_setprio): on a SIMD shared by five wavefronts its instructions go first, the others fill the gaps.
	if (costBin < (uint32_t)MV_TILE_PRIORITY_BINS) __builtin_amdgcn_s_setprio(3);
	else if (costBin < 2u * (uint32_t)MV_TILE_PRIORITY_BINS) __builtin_amdgcn_s_setprio(1);
	else __builtin_amdgcn_s_setprio(0);
#endif

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
	const uint32_t localTile = faceId == 7u ? tile - st.workTileStart[entry] : cubeLocalTile(st, volumeId, tile - st.workTileStart[entry]);

	const ushort4 vi = reinterpret_cast<const ushort4*>(st.volumeInfo)[volumeId];	// CSRayMarch.hlsl:81
	const uint32_t mipLevel = vi.x, smpCount = vi.y, volTexId = vi.w;
	const PerObject& po = reinterpret_cast<const PerObject*>(st.perObject)[volumeId];
	const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
	const bool screenTile = faceId == 7u;	// wave-uniform

	// ---- ray setup: a cube-face texel (CSRayMarch.hlsl:84-107) or a screen pixel of a RayCast volume (RayCast.hlsli:43-54)
	uint32_t nSamples = 0, nLit = 0, nRays = 0, nSkipped = 0;
	bool valid = false;
	vec3 rayOrigin = mul43(v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]), po.worldI);	// :89
	vec3 rayDir = v3(0.0f, 0.0f, 1.0f);
	float tMax = 0.0f;
	size_t outIndex = 0;
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

	if (STAGE || valid)	// (STAGE: the whole wavefront takes part in the cooperative copies, lanes without a ray included)
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
			if constexpr (STAGE)
			{
				// ---- staging (round 3; BASELINE north_star "LDS-staged volume bricks for coalesced HBM reads") ----
				// Per step of the wavefront: (1) every lane that needs texels finds the bricks of its 2x2x2 footprint; the wavefront takes
				// the bounding box of all of them in brick coordinates (LDS min / max atomics); (2) if the box holds at most kStageBricks
				// bricks, the whole box goes into LDS with global_load_lds_dwordx4 -- two bricks per instruction, lane l's 16 bytes landing
				// at base + 16 l, fully coalesced 512-byte reads, no VGPRs -- and every lane reads its eight taps from LDS (ds_read_b64);
				// a larger box (a tile seen at a grazing angle, or rays far apart after adaptive steps) falls back to the direct gathers.
				// The loop is the reference's, lane for lane; lanes run it in lockstep over i with a per-lane `alive` instead of
				// break / continue, so that every lane reaches the cooperative copy.  Results are bit-identical.
				// MEASURED (profiles/r03_lds_staging.txt): slower than the direct gathers -- see there and DESIGN.md section 4.
				const uint32_t wave = threadIdx.x >> 6;
				u32x4* stage = sStage + (size_t)wave * kStageBricks * 32u;
				uint32_t* box = sBox + wave * 8u;
				const uint32_t strideYb = brickPitchY(st.bricks), strideZb = brickPitchZ(st.bricks);
				bool alive = valid;
				for (uint32_t i = 0; i < smpCount && __ballot(alive); ++i)
				{
					vec3 uvw = v3(0.0f, 0.0f, 0.0f);
					FootprintAddr fa;
					bool fetch = false;
					if (alive)
					{
						const vec3 pos = rayAt(rayOrigin, rayDir, t);
						if (outsideBox(pos)) alive = false;	// :120
						else
						{
							uvw = LocalToTex3DSpace(pos);
							fa = footprintAddr(vol, uvw);
							fetch = true;
#if MV_OCCUPANCY
							const uint32_t w = occupancyWord(fa.ix, fa.iy, fa.iz, regions);
							if (w != occIndex) { occWord = occupancy[w]; occIndex = w; }
							if (!((occWord >> occupancyBit(fa.ix, fa.iy, fa.iz)) & 1ull))
							{
								fetch = false;
								++nSamples; ++nSkipped;
								prevLit = false;
								t += stepScale;
								if (t > tMax) alive = false;	// :152
							}
#endif
						}
					}
					// (1) bounding box of the footprints' bricks
					const uint32_t N1 = vol.N - 1u;
					// second taps: min(i + 1, N - 1) -- or the first tap again where the sampler's two taps coincide (a coordinate below the first
					// texel centre: taps (0, 0); FootprintAddr marks that with a zero cell weight, and a zero weight makes the tap irrelevant)
					const uint32_t x1 = fa.cx == 0.0f ? fa.ix : (fa.ix + 1u < N1 ? fa.ix + 1u : N1);
					const uint32_t y1 = fa.cy == 0.0f ? fa.iy : (fa.iy + 1u < N1 ? fa.iy + 1u : N1);
					const uint32_t z1 = fa.cz == 0.0f ? fa.iz : (fa.iz + 1u < N1 ? fa.iz + 1u : N1);
					if (lane < 6u) box[lane] = lane < 3u ? 0xffffffffu : 0u;
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					if (fetch)
					{
						atomicMin(&box[0], fa.ix >> 2); atomicMin(&box[1], fa.iy >> 2); atomicMin(&box[2], fa.iz >> 2);
						atomicMax(&box[3], x1 >> 2); atomicMax(&box[4], y1 >> 2); atomicMax(&box[5], z1 >> 2);
					}
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					const uint32_t bx0 = __builtin_amdgcn_readfirstlane(box[0]), by0 = __builtin_amdgcn_readfirstlane(box[1]), bz0 = __builtin_amdgcn_readfirstlane(box[2]);
					const uint32_t bx1 = __builtin_amdgcn_readfirstlane(box[3]), by1 = __builtin_amdgcn_readfirstlane(box[4]), bz1 = __builtin_amdgcn_readfirstlane(box[5]);
					const bool any = bx0 != 0xffffffffu;
					const uint32_t dx = any ? bx1 - bx0 + 1u : 0u, dy = any ? by1 - by0 + 1u : 0u, dz = any ? bz1 - bz0 + 1u : 0u, nb = dx * dy * dz;
					const bool staged = any && nb <= kStageBricks;	// wave-uniform
					ColourFootprint cf;
					if (staged)
					{
						// (2) the box into LDS, two bricks per instruction
						for (uint32_t j = 0; j < nb; j += 2u)
						{
							const uint32_t b = j + (lane >> 5);
							if (b < nb)
							{
								const uint32_t bxi = b % dx, byi = (b / dx) % dy, bzi = b / (dx * dy);
								const size_t brick = (size_t)(bz0 + bzi) * strideZb + (size_t)(by0 + byi) * strideYb + (bx0 + bxi);
								__builtin_amdgcn_global_load_lds((GlobalVoidPtr)(vol.base + brick * 512u + (lane & 31u) * 16u), (LdsVoidPtr)(stage + (size_t)j * 32u), 16, 0, 0);
							}
						}
						asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
						if (fetch)
						{
							// (volatile: nothing tells the compiler that the copies above wrote these addresses)
							const volatile u32x2* ldsTexels = reinterpret_cast<const volatile u32x2*>(stage);
							auto tap = [&](uint32_t x, uint32_t y, uint32_t z) -> uint2
							{
								const uint32_t b = (((z >> 2) - bz0) * dy + ((y >> 2) - by0)) * dx + ((x >> 2) - bx0);
								const u32x2 v = ldsTexels[b * 64u + ((z & 3u) << 4) + ((y & 3u) << 2) + (x & 3u)];
								return make_uint2(v.x, v.y);
							};
							cf.t000 = tap(fa.ix, fa.iy, fa.iz); cf.t100 = tap(x1, fa.iy, fa.iz);
							cf.t010 = tap(fa.ix, y1, fa.iz); cf.t110 = tap(x1, y1, fa.iz);
							cf.t001 = tap(fa.ix, fa.iy, z1); cf.t101 = tap(x1, fa.iy, z1);
							cf.t011 = tap(fa.ix, y1, z1); cf.t111 = tap(x1, y1, z1);
							cf.fx = fa.fx; cf.fy = fa.fy; cf.fz = fa.fz;
						}
					}
					else if (fetch) cf = fetchColour(vol, fa);
					if (fetch)
					{
						LightFootprint lf;
						const bool early = MV_PREDICT_LIT && prevLit;
						if (early) lf = fetchLightFootprint(lightQuads, st.L, uvw);
						const float alpha = resolveAlphaFromTexels(cf);
						++nSamples;
						float newStep = stepScale;
						const bool lit = alpha > kZeroThreshold;	// :128
						prevLit = lit;
						bool saturated = false;
						if (lit)
						{
							if (!early) lf = fetchLightFootprint(lightQuads, st.L, uvw);
							const vec3 light = resolveLight(lf);
							vec3 color = resolveRGB(cf);
							++nLit;
							const float transm = 1.0f - scatter.w;
							const float dDensity = alpha - prevDensity;
							newStep = GetViewStep(dDensity, scatter.w, alpha, stepScale);
							prevDensity = alpha;
							const float a = transm * kAbsorption * alpha;
							scatter.x += a * color.x * light.x;
							scatter.y += a * color.y * light.y;
							scatter.z += a * color.z * light.z;
							scatter.w += a;
							saturated = transm < kZeroThreshold;	// :146
						}
						if (saturated) alive = false;
						else
						{
							t += newStep;
							if (t > tMax) alive = false;	// :152
						}
					}
				}
			}
			else
			if constexpr (SPEC)
			{
				// ---- speculation (round 3) ----
				// A ray is a chain of dependent samples: the next position needs this sample's step, which needs its alpha, which needs
				// its texels.  While the accumulated alpha is small -- and through empty space always -- the step is exactly stepScale
				// (GetStep, RayMarch.hlsli:182-192: max(1.5 scatter.w factorEv factorUi, 1) stepScale), so the footprint at t + stepScale is
				// requested BEFORE this sample is resolved and is there when the loop arrives; a wrong guess (t advanced by a longer step)
				// discards it and fetches at the real t.  `t + stepScale` is the very addition the loop performs, so a right guess is the
				// same float: arithmetic, order and results are the reference's, only requests are issued earlier (and some in vain).
				// It costs 40 registers (92 -> 132) and wasted requests.  MEASURED SLOWER in both regimes it was meant for -- a launch that fills
				// the GPU (a wavefront per SIMD less) and a launch that leaves it nearly empty (a rank's share of a frame in the multi-GPU
				// loop: 116 -> 158 us) -- so no launch selects it (MV_TILE_SPECULATE_BELOW = 0); kept as the experiment it was.
				struct Prepared { ColourFootprint cf; vec3 uvw; float t; bool outside, skip; };
				auto prepare = [&](float tt) -> Prepared
				{
					Prepared p;
					p.t = tt;
					const vec3 pos = rayAt(rayOrigin, rayDir, tt);
					p.outside = outsideBox(pos);
					p.skip = false;
					p.uvw = LocalToTex3DSpace(pos);
					if (p.outside) return p;
					FootprintAddr fa;
					uint32_t occSum = 0u;
					if (LUT)
					{
						const u32x4 ex = sAxis[axisTableIndex(p.uvw.x, vol.N, fa.fx)];
						const u32x4 ey = sAxis[(kLutMaxN + 1u) + axisTableIndex(p.uvw.y, vol.N, fa.fy)];
						const u32x4 ez = sAxis[2u * (kLutMaxN + 1u) + axisTableIndex(p.uvw.z, vol.N, fa.fz)];
						fa.x0 = ex.x; fa.x1 = ex.y;
						fa.r00 = ey.x + ez.x; fa.r10 = ey.y + ez.x; fa.r01 = ey.x + ez.y; fa.r11 = ey.y + ez.y;
						occSum = ex.z + ey.z + ez.z;
					}
					else fa = footprintAddr(vol, p.uvw);
#if MV_OCCUPANCY
					{
						const uint32_t w = LUT ? occSum & 0xffffffu : occupancyWord(fa.ix, fa.iy, fa.iz, regions);
						if (w != occIndex) { occWord = occupancy[w]; occIndex = w; }
						p.skip = !((occWord >> (LUT ? occSum >> 24 : occupancyBit(fa.ix, fa.iy, fa.iz))) & 1ull);
					}
#endif
					if (!p.skip) p.cf = fetchColour(vol, fa);
					else { p.cf.fx = fa.fx; p.cf.fy = fa.fy; p.cf.fz = fa.fz; }
					return p;
				};
				Prepared cur = prepare(t);
				for (uint32_t i = 0; i < smpCount; ++i)
				{
					if (cur.outside) break;	// :120
					const float tGuess = t + stepScale;
					const bool last = i + 1u >= smpCount || tGuess > tMax;	// (nothing follows this sample if the step is stepScale: no request)
					Prepared nxt;
					if (!last) nxt = prepare(tGuess);
					float newStep = stepScale;
					if (cur.skip)
					{
						++nSamples;
						++nSkipped;
						prevLit = false;
					}
					else
					{
						LightFootprint lf;
						const bool early = MV_PREDICT_LIT && prevLit;
						if (early) lf = fetchLightFootprint(lightQuads, st.L, cur.uvw);
						const float alpha = resolveAlphaFromTexels(cur.cf);
						++nSamples;
						const bool lit = alpha > kZeroThreshold;	// :128
						prevLit = lit;
						if (lit)
						{
							if (!early) lf = fetchLightFootprint(lightQuads, st.L, cur.uvw);
							const vec3 light = resolveLight(lf);
							vec3 color = resolveRGB(cur.cf);
							++nLit;
							const float transm = 1.0f - scatter.w;
							const float dDensity = alpha - prevDensity;
							newStep = GetViewStep(dDensity, scatter.w, alpha, stepScale);
							prevDensity = alpha;
							const float a = transm * kAbsorption * alpha;
							scatter.x += a * color.x * light.x;
							scatter.y += a * color.y * light.y;
							scatter.z += a * color.z * light.z;
							scatter.w += a;
							if (transm < kZeroThreshold) break;	// :146
						}
					}
					t += newStep;
					if (t > tMax) break;	// :152
					if (!last && t == tGuess) cur = nxt;
					else if (i + 1u < smpCount) cur = prepare(t);
				}
			}
			else
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

			if (!valid) {}	// (STAGE: a lane without a ray went through the loop for the copies' sake)
			else if (!screenTile)
			{
				uint2 o;
				o.x = (uint32_t)float_to_half_bits(scatter.x) | ((uint32_t)float_to_half_bits(scatter.y) << 16);
				o.y = (uint32_t)float_to_half_bits(scatter.z) | ((uint32_t)float_to_half_bits(scatter.w) << 16);
				st.cubeMaps[outIndex] = o;	// :157
			}
			else
			{
				float4 o; o.x = scatter.x; o.y = scatter.y; o.z = scatter.z; o.w = scatter.w;
				st.screenLayers[outIndex] = o;	// what RayCast returns (RayCast.hlsli:107); the composite applies PSCube.hlsl:56
			}
		}
	}

	const uint32_t totS = waveSumU32(nSamples), totL = waveSumU32(nLit), totR = waveSumU32(nRays), totK = waveSumU32(nSkipped);
	uint32_t maxS = nSamples;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const uint32_t m = __shfl_xor(maxS, o); maxS = m > maxS ? m : maxS; }
	const uint32_t nCubeTiles = st.counters[6];
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

uint32_t marchTilesResidentWaves(int device)
{
	int perCu = 0, cus = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, marchTilesKernel<false, true, false>, 64 * kTileWaves, 0) != hipSuccess || perCu <= 0) perCu = 8;
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
	return (uint32_t)perCu * (uint32_t)cus * kTileWaves;
}

#ifndef MV_TILE_STAGE
#define MV_TILE_STAGE 0
#endif
#ifndef MV_TILE_SPECULATE_BELOW
#define MV_TILE_SPECULATE_BELOW 0	// tiles per launch under which the next sample's texels are requested ahead (SPEC).  0 = never: measured (round 3,
										// profiles/r03_speculation.txt) a rank's share of C3 on 8 ranks, 1.5 k tiles, 116 us -> 158 us with it -- a lone wavefront's
										// iteration is its ~230 dependent VALU instructions, not the wait for its texels
#endif
uint32_t marchTilesBeyond(const DeviceState& st, uint32_t gridTiles)
{
	const uint64_t maxTiles = (uint64_t)st.maxTiles + st.maxScreenTiles;
	const uint64_t tiles = gridTiles < maxTiles ? (gridTiles ? gridTiles : 1u) : maxTiles;
	uint64_t blocks = (tiles + kTileWaves - 1) / kTileWaves;
	blocks = ((blocks + 7) / 8) * 8;
	return blocks * kTileWaves < maxTiles ? (uint32_t)(blocks * kTileWaves) : 0xffffffffu;
}

uint32_t launchMarchTiles(hipStream_t s, const DeviceState& st, uint32_t gridTiles)
{
	// gridTiles: the host's estimate of this frame's tile count (the cull's latest report + a margin; anything large when it has
	// none).  The grid covers min(estimate, largest possible frame); returns the first dispatch slot beyond it (what the overflow
	// launch has to cover), or 0xffffffff if the grid covers every frame there can be.
	const uint64_t maxTiles = (uint64_t)st.maxTiles + st.maxScreenTiles;
	const uint64_t tiles = gridTiles < maxTiles ? (gridTiles ? gridTiles : 1u) : maxTiles;
	uint64_t blocks = (tiles + kTileWaves - 1) / kTileWaves;
	blocks = ((blocks + 7) / 8) * 8;
	const bool il = st.flags & kFlagInterleaveRows;
	const bool spec = tiles < (uint64_t)MV_TILE_SPECULATE_BELOW;
	const dim3 grid((uint32_t)blocks), block(64 * kTileWaves);
#if MV_TILE_STAGE	// the LDS-staged variant instead of the direct gathers (an experiment: profiles/r03_lds_staging.txt)
	if (il) hipLaunchKernelGGL((marchTilesKernel<false, false, true, false, true>), grid, block, 0, s, st, 0u);
	else hipLaunchKernelGGL((marchTilesKernel<false, false, false, false, true>), grid, block, 0, s, st, 0u);
	return blocks * kTileWaves < maxTiles ? (uint32_t)(blocks * kTileWaves) : 0xffffffffu;
#endif
	if (st.N <= kLutMaxN && MV_TILE_LUT != 0)
	{
		if (spec)
		{
			if (il) hipLaunchKernelGGL((marchTilesKernel<false, true, true, true>), grid, block, 0, s, st, 0u);
			else hipLaunchKernelGGL((marchTilesKernel<false, true, false, true>), grid, block, 0, s, st, 0u);
		}
		else if (il) hipLaunchKernelGGL((marchTilesKernel<false, true, true>), grid, block, 0, s, st, 0u);
		else hipLaunchKernelGGL((marchTilesKernel<false, true, false>), grid, block, 0, s, st, 0u);
	}
	else if (il) hipLaunchKernelGGL((marchTilesKernel<false, false, true>), grid, block, 0, s, st, 0u);
	else hipLaunchKernelGGL((marchTilesKernel<false, false, false>), grid, block, 0, s, st, 0u);
	return blocks * kTileWaves < maxTiles ? (uint32_t)(blocks * kTileWaves) : 0xffffffffu;
}

// (several frames per launch -- mv_config.frames_per_launch in the multi-GPU loop -- exist for the product kernel only: these builds are A/B builds of the single-GPU march)
void launchMarchTilesBatch(hipStream_t, const DeviceState*, uint32_t, const uint32_t*, uint32_t*) { fprintf(stderr, "multivolumes_amd: this build of the tile march (csrc/experiments) has no launches of several frames\n"); abort(); }
void launchMarchTilesOverflowBatch(hipStream_t, const DeviceState*, uint32_t, const uint32_t*) { abort(); }
void launchMarchTilesOverflow(hipStream_t s, const DeviceState& st, uint32_t firstSlot)
{
	if (st.flags & kFlagInterleaveRows) hipLaunchKernelGGL((marchTilesKernel<true, false, true>), dim3(64), dim3(64 * kTileWaves), 0, s, st, firstSlot);
	else hipLaunchKernelGGL((marchTilesKernel<true, false, false>), dim3(64), dim3(64 * kTileWaves), 0, s, st, firstSlot);
}
} // namespace mv
