// experiments/mv_march_narrow.hip -- MEASURED AND REJECTED (round 4, profiles/r04_narrow_march.txt): bit-identical to the product's tile march
// on every parity test, and slower at every size -- C1 (64 tiles) 38 vs 37 us, T1 (192 tiles) 50 vs 44, C2 (768 tiles) 87 vs 62, a rank of
// eight at C3 (~1 000 tiles) 190 vs 110 us per frame.  Eight lanes per ray mean eight times the wavefronts, each issuing its (mostly redundant:
// position, address tables, occupancy) instructions at the full four cycles a wave64 instruction takes; from ~128 tiles on those wavefronts
// no longer find idle SIMDs and the launch becomes bound by instruction issue, and below that the cross-lane tree and the LDS / DPP
// latencies cost what the shorter per-lane decode saves.  Built only with -DMV_TILE_NARROW_BELOW=<tiles> (csrc/build.sh).  What follows
// is the design as it was written:
// a3 (+ the fused screen-space RayCast) for launches that CANNOT fill the GPU: a rank's share of a frame in the
// multi-GPU loop, BASELINE config C2 (768 tiles on 1 024 SIMDs).  Such a launch lasts as long as its longest ray, and a ray is a chain
// of dependent samples: in the one-wavefront-per-tile form (mv_march_tiles.hip) an iteration is ~230 dependent VALU instructions of a
// lone wavefront (0.45 us) plus the round trip of its texels (0.4-0.8 us) -- the other 63 rays of the wavefront add nothing to that
// latency and the rest of the GPU is idle.  Here the same loop (CSRayMarch.hlsl:117-153, same arithmetic, same order, same bits) is laid
// out for latency instead of throughput:
//   * TAP-PARALLEL: eight lanes per ray, one texel tap of the 2x2x2 footprint each (lane bits 0 / 1 / 2 = the x / y / z tap).  A lane
//     decodes ONE RGBA texel and ONE light-map texel instead of eight, and the trilinear tree runs across lanes -- x between lane
//     pairs (DPP quad_perm), y between pairs of pairs, z between the halves of the 8-lane group (DPP row_shl:4) -- with the very
//     operations of mv_device.hpp's trilerpChannel / resolveLight (v_fma_mix lerp on the fp16 bits for x, fma(w, b - a, a) for y and z),
//     so lane 0 of the group ends up with the bits the wide form computes.  A wavefront holds 8 rays (one row of the 8x8 tile), a
//     workgroup of 8 wavefronts the tile: eight times the wavefronts on a GPU that was empty anyway.
//   * SPECULATIVE: the texels of the NEXT sample are requested before the current one is resolved.  The next position needs this
//     sample's step, but GetStep (RayMarch.hlsli:182-192) returns exactly stepScale through empty space and while little has
//     accumulated, so t + stepScale -- the very addition the loop performs -- is requested ahead; a longer step discards the request and
//     fetches at the real t.  With one texel and one light word per lane the request costs three registers (in the wide form it cost
//     forty and was a loss: profiles/r03_speculation.txt); an iteration is then max(VALU, round trip) instead of their sum.
// Results, counters and tile costs are those of the wide form, bit for bit; the host picks the form by the launch's tile count
// (launchMarchTiles, MV_TILE_NARROW_BELOW).
#include "mv_device.hpp"
#include "mv_kernels.hpp"

namespace mv
{
#ifndef MV_PREDICT_LIT_NARROW
#define MV_PREDICT_LIT_NARROW 1	// request a sample's light-map texel together with its volume texel while the ray is inside lit space (as MV_PREDICT_LIT does in the wide form)
#endif
constexpr uint32_t kNarrowWaves = 8;			// wavefronts per workgroup = rows of the 8x8 tile
constexpr uint32_t kNarrowLutMaxN = 256;		// (the address tables in LDS: 3 (N + 1) x 16 B)

// value of lane (lane & ~7) in every lane of its 8-lane group: quad_perm [0,0,0,0], then the upper quad of every group takes the lower one's (row_shr:4 into banks 1 and 3)
__device__ __forceinline__ uint32_t groupLeader(uint32_t v)
{
	const uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x00, 0xf, 0xf, false);
	return (uint32_t)__builtin_amdgcn_update_dpp((int)q, (int)q, 0x114, 0xf, 0xa, false);
}
__device__ __forceinline__ float groupLeader(float v) { return __uint_as_float(groupLeader(__float_as_uint(v))); }
// the partner's value across one tap axis: x = the other lane of the pair, y = the other pair of the quad, z = the lane four on (valid in the lower quad of a group)
__device__ __forceinline__ uint32_t partnerX(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false); }	// quad_perm [1,0,3,2]
__device__ __forceinline__ float partnerY(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x4e, 0xf, 0xf, false)); }	// quad_perm [2,3,0,1]
__device__ __forceinline__ float partnerZ(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x104, 0xf, 0xf, false)); }	// row_shl:4

// One channel of the trilinear tree across the group (mv_device.hpp trilerpChannel, lane for lane): the x lerp on the fp16 bits of this
// lane's register and its x partner's, then y and z in fp32.  The result is valid in lane 0 of the group (taps 0,0,0).
template <int HI>
__device__ __forceinline__ float groupTrilerpHalves(uint32_t reg, float fx, float fy, float fz)
{
	const float x = lerpHalves<HI, HI>(reg, partnerX(reg), fx);	// valid where the x tap is 0
	const float y = lerpf(x, partnerY(x), fy);						// valid where x and y taps are 0
	return lerpf(y, partnerZ(y), fz);
}

#ifndef MV_TILE_MIN_WAVES_NARROW
#define MV_TILE_MIN_WAVES_NARROW 1
#endif
template <bool IL, bool PEER>
__global__ __launch_bounds__(64 * kNarrowWaves, MV_TILE_MIN_WAVES_NARROW) void marchTilesNarrowKernel(DeviceState st)
{
	__shared__ u32x4 sAxis[3 * (kNarrowLutMaxN + 1)];
	__shared__ uint32_t sMaxSamples;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t tapX = lane & 1u, tapY = (lane >> 1) & 1u, tapZ = (lane >> 2) & 1u, ray = lane >> 3;
	const bool leader = (lane & 7u) == 0u;
	const uint32_t groupShift = lane & ~7u;	// (bit of this group's leader in a ballot)
	if constexpr (PEER) peerFoldWait(st);
	const uint32_t nEntries = st.counters[2], nTiles = st.counters[3];
	const uint32_t slot = blockIdx.x;
	if (slot < nTiles)	// (workgroup-uniform; a surplus workgroup still reaches the epilogue)
	{
		{
			const uint32_t P = st.N + 1u, regions = occupancyRegions(st.bricks);
			const uint32_t strideY = brickPitchY(st.bricks) * 512u, strideZ = brickPitchZ(st.bricks) * 512u;
			for (uint32_t k = threadIdx.x; k < 3u * P; k += 64u * kNarrowWaves)
				sAxis[(k / P) * (kNarrowLutMaxN + 1u) + k % P] = axisTableEntry(k / P, k % P, st.N, strideY, strideZ, regions);
			if (threadIdx.x == 0u) sMaxSamples = 0u;
			__syncthreads();
		}
		const uint32_t tile = __builtin_amdgcn_readfirstlane(orderedTile(st, slot));
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
		const uint32_t mipLevel = vi.x, smpCount = vi.y, volTexId = vi.w;
		const PerObject& po = reinterpret_cast<const PerObject*>(st.perObject)[volumeId];
		const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
		const bool screenTile = faceId == 7u;	// wave-uniform

		// ---- ray setup, as in mv_march_tiles.hip (every lane of a group sets up its ray: the same values eight times) ----
		uint32_t nSamples = 0, nLit = 0, nRays = 0, nSkipped = 0;
		bool valid = false;
		vec3 rayOrigin = mul43(v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]), po.worldI);	// :89
		vec3 rayDir = v3(0.0f, 0.0f, 1.0f);
		float tMax = 0.0f, peerDepth = 0.0f;
		size_t outIndex = 0;
		if (!screenTile)
		{
			const uint32_t dims = cubeMipDim(st.N, mipLevel);
			const uint32_t tilesX = (dims + 7u) >> 3;
			const uint32_t px = (localTile % tilesX) * 8u + ray, py = (localTile / tilesX) * 8u + wave;
			if (px < dims && py < dims)
			{
				const vec3 target = GetLocalPos((float)px, (float)py, faceId, (float)dims);
				rayDir = normalize3(target - rayOrigin);
				if (ComputeRayOrigin(rayOrigin, rayDir))	// :96
				{
					tMax = ComputeTargetHit(rayOrigin, target, rayDir);
					outIndex = (size_t)st.cubeOffset[volumeId] + ((size_t)faceId * dims + py) * dims + px;
					vec3 clip;	// GetClipPos (:59-71)
					{
						vec4 hPos;
						hPos.x = rayOrigin.x + rayDir.x * 0.01f;
						hPos.y = rayOrigin.y + rayDir.y * 0.01f;
						hPos.z = rayOrigin.z + rayDir.z * 0.01f;
						hPos.w = 1.0f;
						hPos = mul44(hPos, po.wvp);
						const float cx = hPos.x / hPos.w, cy = hPos.y / hPos.w;
						const float u = cx * 0.5f + 0.5f;
						const float v = 0.5f - cy * 0.5f;
						clip = v3(cx, cy, samplePointR32F(st.depthMap, st.W, st.H, u, v));
					}
					if (leader) st.cubeDepths[outIndex] = clip.z;	// :105
					peerDepth = clip.z;
					tMax = __builtin_fminf(GetTMax(clip, rayOrigin, rayDir, po.wvpI), tMax);	// :106
					valid = true;
					nRays = 1;
				}
			}
		}
		else
		{
			const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[volumeId];
			const uint32_t x1 = rect.z < st.W - 1u ? rect.z : st.W - 1u;
			const ScreenTileRows rows = screenTileRowsT<IL>(st, rect.y, rect.w < st.H - 1u ? rect.w : st.H - 1u);
			const uint32_t tx0 = rect.x >> 3, tw = (x1 >> 3) - tx0 + 1u;
			const uint32_t px = (tx0 + localTile % tw) * 8u + ray, py = (rows.first + (localTile / tw) * rows.step) * 8u + wave;
			if (px < st.W && py < st.H)
			{
				float nx = ((float)px + 0.5f) / (float)st.W, ny = ((float)py + 0.5f) / (float)st.H;
				nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
				vec3 lpt;
				float z;
				outIndex = ((size_t)st.screenSlot[volumeId] * st.H + py) * st.W + px;
				if (findExitFragment(po, rayOrigin, pixelRayWorld(cb, nx, ny), lpt, z))
				{
					rayDir = normalize3(lpt - rayOrigin);	// PSCube.hlsl:33, 50
					float4 zero; zero.x = zero.y = zero.z = zero.w = 0.0f;
					if (leader) st.screenLayers[outIndex] = zero;	// RayCast returns 0 on a miss (RayCast.hlsli:46)
					if (ComputeRayOrigin(rayOrigin, rayDir))
					{
						const float depthPix = st.depthMap[(size_t)py * st.W + px];
						tMax = GetTMax(v3(nx, ny, depthPix), rayOrigin, rayDir, po.wvpI);	// RayCast.hlsli:50-51
						valid = true;
					}
				}
			}
		}

		vec4 scatter; scatter.x = scatter.y = scatter.z = scatter.w = 0.0f;
		if (valid)	// (uniform over a group)
		{
			const float stepScale = kMaxDist / (float)smpCount;
			const GlobalBytes texels = (GlobalBytes)st.volumes[volTexId];
			const GlobalBytes lightBytes = (GlobalBytes)st.lightTable[volumeId];
			const uint64_t* occupancy = st.occupancy[volTexId];
			const uint32_t L = st.L, lightSlice = (L * L) << 4;
			uint32_t occIndex = 0xffffffffu;
			uint64_t occWord = 0;
			float t = 0.0f, prevDensity = 0.0f;
			bool prevLit = false;

			// everything a sample needs that can be had before the previous one is resolved: where it lies, whether its brick can hold
			// density at all, this lane's texel, and -- while the ray is inside lit space -- this lane's light-map texel
			struct Prepared { float t, fx, fy, fz, lcx, lcy, lfz; uint32_t lightOff; u32x2 texel; uint32_t lightWord; bool outside, skip, haveLight; };
			auto lightTaps = [&](const vec3& uvw, Prepared& p)
			{
				// fetchLightFootprintAt (mv_device.hpp), this lane's texel of the two quads: slot tapX + 2 tapY of the cell at z tap tapZ
				const Taps tx = linearTaps(uvw.x, L, 0), ty = linearTaps(uvw.y, L, 0), tz = linearTaps(uvw.z, L, 0);
				const uint32_t row = (__umul24((uint32_t)ty.i0, L) + (uint32_t)tx.i0) << 4;
				p.lightOff = __umul24((uint32_t)(tapZ ? tz.i1 : tz.i0), lightSlice) + row + ((tapX + 2u * tapY) << 2);
				p.lcx = tx.i1 == tx.i0 ? 0.0f : tx.f; p.lcy = ty.i1 == ty.i0 ? 0.0f : ty.f; p.lfz = tz.f;
			};
			auto prepare = [&](float tt, bool wantLight) -> Prepared
			{
				Prepared p;
				p.t = tt; p.skip = false; p.haveLight = false;
				const vec3 pos = rayAt(rayOrigin, rayDir, tt);
				p.outside = outsideBox(pos);	// :120
				if (p.outside) return p;
				const vec3 uvw = LocalToTex3DSpace(pos);
				const u32x4 ex = sAxis[axisTableIndex(uvw.x, st.N, p.fx)];
				const u32x4 ey = sAxis[(kNarrowLutMaxN + 1u) + axisTableIndex(uvw.y, st.N, p.fy)];
				const u32x4 ez = sAxis[2u * (kNarrowLutMaxN + 1u) + axisTableIndex(uvw.z, st.N, p.fz)];
				const uint32_t occSum = ex.z + ey.z + ez.z;
				const uint32_t w = occSum & 0xffffffu;
				if (w != occIndex) { occWord = occupancy[w]; occIndex = w; }
				p.skip = !((occWord >> (occSum >> 24)) & 1ull);	// a clear bit proves alpha <= ZERO_THRESHOLD (mv_device.hpp "brick occupancy")
				if (p.skip) return p;
				p.texel = *(GlobalUint2)(texels + ((tapX ? ex.y : ex.x) + (tapY ? ey.y : ey.x) + (tapZ ? ez.y : ez.x)));
				if (wantLight)
				{
					lightTaps(uvw, p);
					p.lightWord = *(GlobalUint)(lightBytes + p.lightOff);
					p.haveLight = true;
				}
				return p;
			};

			Prepared cur = prepare(t, false);
			for (uint32_t i = 0; i < smpCount; ++i)
			{
				if (cur.outside) break;	// :120
				const float tGuess = t + stepScale;
				const bool last = i + 1u >= smpCount || tGuess > tMax;	// (nothing follows this sample if the step is stepScale: no request)
				Prepared nxt;
				if (!last) nxt = prepare(tGuess, MV_PREDICT_LIT_NARROW && prevLit);
				float newStep = stepScale;
				++nSamples;
				if (cur.skip)
				{
					++nSkipped;
					prevLit = false;
				}
				else
				{
					// alpha first (:128): channel 3 = the high half of the texel's second word
					const float alpha = groupTrilerpHalves<1>(cur.texel.y, cur.fx, cur.fy, cur.fz);
					const bool lit = ((__ballot(leader && alpha > kZeroThreshold) >> groupShift) & 1ull) != 0ull;	// the leader's verdict, for its group
					prevLit = lit;
					if (lit)
					{
						if (!cur.haveLight)
						{
							const vec3 uvw = LocalToTex3DSpace(rayAt(rayOrigin, rayDir, cur.t));
							lightTaps(uvw, cur);
							cur.lightWord = *(GlobalUint)(lightBytes + cur.lightOff);
						}
						vec3 color;
						color.x = groupTrilerpHalves<0>(cur.texel.x, cur.fx, cur.fy, cur.fz);
						color.y = groupTrilerpHalves<1>(cur.texel.x, cur.fx, cur.fy, cur.fz);
						color.z = groupTrilerpHalves<0>(cur.texel.y, cur.fx, cur.fy, cur.fz);
						// resolveLight (mv_device.hpp): R and G as fp16 bit patterns in the halves of one register, B in the low half of another
						const uint32_t p = cur.lightWord;
						const uint32_t rg = ((p << 4) & 0x7ff0u) | ((p << 9) & 0x7ff00000u), bb = (p >> 17) & 0x7fe0u;
						vec3 light;
						light.x = groupTrilerpHalves<0>(rg, cur.lcx, cur.lcy, cur.lfz);
						light.y = groupTrilerpHalves<1>(rg, cur.lcx, cur.lcy, cur.lfz);
						light.z = groupTrilerpHalves<0>(bb, cur.lcx, cur.lcy, cur.lfz);
						++nLit;
						const float transm = 1.0f - scatter.w;
						const float dDensity = alpha - prevDensity;
						newStep = GetViewStep(dDensity, scatter.w, alpha, stepScale);
						prevDensity = alpha;
						const float a = transm * kAbsorption * alpha;	// :140-146 as compiled
						scatter.x += a * color.x * light.x;
						scatter.y += a * color.y * light.y;
						scatter.z += a * color.z * light.z;
						scatter.w += a;
						if (((__ballot(leader && transm < kZeroThreshold) >> groupShift) & 1ull) != 0ull) break;	// :146
					}
				}
				t = groupLeader(t + newStep);	// (the leader's state is the ray's: its step, for every lane of the group)
				if (t > tMax) break;	// :152
				if (!last && t == tGuess) cur = nxt;
				else cur = prepare(t, MV_PREDICT_LIT_NARROW && prevLit);
			}
			scatter.x *= kInvTwoPi; scatter.y *= kInvTwoPi; scatter.z *= kInvTwoPi;	// :155
		}

		uint2 colour; colour.x = colour.y = 0u;
		if (valid && leader)
		{
			if (!screenTile)
			{
				colour.x = (uint32_t)float_to_half_bits(scatter.x) | ((uint32_t)float_to_half_bits(scatter.y) << 16);
				colour.y = (uint32_t)float_to_half_bits(scatter.z) | ((uint32_t)float_to_half_bits(scatter.w) << 16);
				st.cubeMaps[outIndex] = colour;	// :157
			}
			else
			{
				float4 o; o.x = scatter.x; o.y = scatter.y; o.z = scatter.z; o.w = scatter.w;
				st.screenLayers[outIndex] = o;	// what RayCast returns (RayCast.hlsli:107)
			}
		}
		if constexpr (PEER)
		{
			// the row's texels into the other ranks' cube maps (mv_march_tiles.hip has the same stores for its 8x8 lanes): the eight leaders
			// hold eight neighbouring texels -- an even ray takes its right neighbour's colour along, rays 0 and 4 their three neighbours' depths
			if (!screenTile && st.peerCount > 1u)
			{
				const bool have = valid && leader;
				const uint2 right = make_uint2(__shfl_down(colour.x, 8), __shfl_down(colour.y, 8));
				const unsigned long long hits = __ballot(have);
				const bool haveRight = ray < 7u && ((hits >> (lane + 8u)) & 1ull), haveLeft = ray > 0u && ((hits >> (lane - 8u)) & 1ull);
				const bool even = !(ray & 1u);
				const bool pair = have && even && haveRight && !(outIndex & 1u);
				const bool single = have && !pair && !(!even && haveLeft && !((outIndex - 1u) & 1u));
				const float d1 = __shfl_down(peerDepth, 8), d2 = __shfl_down(peerDepth, 16), d3 = __shfl_down(peerDepth, 24);
				const uint32_t quadBase = lane & 32u;	// rays 0-3 live in lanes 0-31, rays 4-7 in lanes 32-63
				const unsigned long long quadMask = 0x01010101ull << quadBase;
				const bool fullQuad = (hits & quadMask) == quadMask && !((outIndex - (ray & 3u)) & 3u);
				const bool quad = have && !(ray & 3u) && fullQuad;
				for (uint32_t p = 0; p < st.peerCount; ++p)
				{
					if (p == st.shardIndex) continue;
					uint2* colours = st.peerCubeMaps[p];
					float* depths = st.peerCubeDepths[p];
					if (pair) peerStore16(colours + outIndex, make_uint4(colour.x, colour.y, right.x, right.y));
					else if (single) peerStore8(colours + outIndex, colour);
					if (quad) peerStore16(depths + outIndex, make_uint4(__float_as_uint(peerDepth), __float_as_uint(d1), __float_as_uint(d2), __float_as_uint(d3)));
					else if (have && !fullQuad) peerStore(reinterpret_cast<uint32_t*>(depths + outIndex), __float_as_uint(peerDepth));
				}
			}
		}

		// counters: per RAY (the group's leader), summed over the wavefront's eight rays
		auto leaderSum = [&](uint32_t v) -> uint32_t
		{
			v = leader ? v : 0u;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
			return v;
		};
		const uint32_t totS = leaderSum(nSamples), totL = leaderSum(nLit), totR = leaderSum(nRays), totK = leaderSum(nSkipped);
		uint32_t maxS = leader ? nSamples : 0u;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { const uint32_t m = __shfl_xor(maxS, o); maxS = m > maxS ? m : maxS; }
		if (lane == 0)
		{
			atomicMax(&sMaxSamples, maxS);
			unsigned long long* s = reinterpret_cast<unsigned long long*>(st.waveStats) + (size_t)((slot * kNarrowWaves + wave) % st.marchWaves) * 8u;
			if (!screenTile)
			{
				atomicAdd(s + 0, (unsigned long long)totS); atomicAdd(s + 1, (unsigned long long)totL); atomicAdd(s + 2, (unsigned long long)totR);
				atomicAdd(s + 3, (unsigned long long)maxS * 8ull);	// ray slots the wavefront occupied
			}
			else { atomicAdd(s + 4, (unsigned long long)totS); atomicAdd(s + 5, (unsigned long long)totL); }
			if (totK) atomicAdd(s + 6, (unsigned long long)totK);
		}
		__syncthreads();
		if (threadIdx.x == 0u)
			st.tileCost[tileCostKeyT<IL>(st, volumeId, faceId, localTile, tile, st.counters[6])] = sMaxSamples + 1u;	// (0 = never marched)
	}
	if constexpr (PEER) peerFoldSignal(st);
}

// tiles per launch under which the narrow form marches (0: never).  A wide launch of T tiles is T wavefronts; the narrow one 8 T: it
// pays while those still find the GPU empty (256 CUs x 4 SIMDs x 8 wavefronts of this kernel)
#ifndef MV_TILE_NARROW_BELOW
#define MV_TILE_NARROW_BELOW 0
#endif
bool marchTilesNarrow(const DeviceState& st, uint64_t tiles) { return MV_TILE_NARROW_BELOW != 0 && tiles < (uint64_t)MV_TILE_NARROW_BELOW && st.N <= kNarrowLutMaxN; }

void launchMarchTilesNarrow(hipStream_t s, const DeviceState& st, uint32_t blocks)
{
	const bool il = st.flags & kFlagInterleaveRows;
	const dim3 grid(blocks), block(64 * kNarrowWaves);
	if (st.peerCount > 1u && st.fold.stores)
	{
		if (il) hipLaunchKernelGGL((marchTilesNarrowKernel<true, true>), grid, block, 0, s, st);
		else hipLaunchKernelGGL((marchTilesNarrowKernel<false, true>), grid, block, 0, s, st);
	}
	else if (il) hipLaunchKernelGGL((marchTilesNarrowKernel<true, false>), grid, block, 0, s, st);
	else hipLaunchKernelGGL((marchTilesNarrowKernel<false, false>), grid, block, 0, s, st);
}
} // namespace mv
