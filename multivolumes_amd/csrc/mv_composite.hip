// mv_composite.hip -- multi-GPU exchange pack/unpack and the K-buffer OIT composite (a7).
#include "mv_device.hpp"
#include "mv_kernels.hpp"

namespace mv
{
// ---------------------------------------------------------------------------------------------
// Exchange: rank r owns volumes v with v % shardCount == r; local slot j = v / shardCount.
// Slot = [6 d d RGBA16F texels][6 d d R32F depths], d = N >> (min selected mip); a volume at a coarser mip
// fills the front of each part.  All ranks agree on d because the cull is replicated.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exchangePackKernel(DeviceState st, uint32_t slotDim, uint8_t* __restrict__ dst)
{
	const uint32_t j = blockIdx.y;
	const uint32_t v = j * st.shardCount + st.shardIndex;
	if (v >= st.V) return;
	const uint32_t info = st.volumeFlags[v];
	const size_t slotTexels = (size_t)6 * slotDim * slotDim;
	uint2* colour = reinterpret_cast<uint2*>(dst + (size_t)j * slotTexels * 12);
	float* depth = reinterpret_cast<float*>(dst + (size_t)j * slotTexels * 12 + slotTexels * 8);
	const uint32_t dv = st.N >> ((info >> 2) & 7u);
	const size_t n = (info & 1u) ? (size_t)6 * dv * dv : 0;
	const size_t off = (info & 1u) ? (size_t)st.cubeOffset[v] : 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		colour[i] = st.cubeMaps[off + i];
		depth[i] = st.cubeDepths[off + i];
	}
}

__global__ __launch_bounds__(256) void exchangeUnpackKernel(DeviceState st, uint32_t slotDim, const uint8_t* __restrict__ src)
{
	const uint32_t perShard = (st.V + st.shardCount - 1u) / st.shardCount;
	const uint32_t r = blockIdx.y / perShard, j = blockIdx.y % perShard;
	if (r == st.shardIndex) return;	// own volumes are already in place
	const uint32_t v = j * st.shardCount + r;
	if (v >= st.V) return;
	const uint32_t info = st.volumeFlags[v];
	if (!(info & 1u)) return;
	const size_t slotTexels = (size_t)6 * slotDim * slotDim;
	const size_t slot = (size_t)r * perShard + j;
	const uint2* colour = reinterpret_cast<const uint2*>(src + slot * slotTexels * 12);
	const float* depth = reinterpret_cast<const float*>(src + slot * slotTexels * 12 + slotTexels * 8);
	const uint32_t dv = st.N >> ((info >> 2) & 7u);
	const size_t n = (size_t)6 * dv * dv;
	const size_t off = (size_t)st.cubeOffset[v];
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		st.cubeMaps[off + i] = colour[i];
		st.cubeDepths[off + i] = depth[i];
	}
}

void launchExchangePack(hipStream_t s, const DeviceState& st, uint32_t slotDim, void* dst)
{
	const uint32_t perShard = (st.V + st.shardCount - 1u) / st.shardCount;
	const size_t n = (size_t)6 * slotDim * slotDim;
	const uint32_t gx = (uint32_t)((n + 1023) / 1024);
	hipLaunchKernelGGL(exchangePackKernel, dim3(gx ? gx : 1, perShard), dim3(256), 0, s, st, slotDim, reinterpret_cast<uint8_t*>(dst));
}

void launchExchangeUnpack(hipStream_t s, const DeviceState& st, uint32_t slotDim, const void* src)
{
	const uint32_t perShard = (st.V + st.shardCount - 1u) / st.shardCount;
	const size_t n = (size_t)6 * slotDim * slotDim;
	const uint32_t gx = (uint32_t)((n + 1023) / 1024);
	hipLaunchKernelGGL(exchangeUnpackKernel, dim3(gx ? gx : 1, perShard * st.shardCount), dim3(256), 0, s, st, slotDim, reinterpret_cast<const uint8_t*>(src));
}

// ---------------------------------------------------------------------------------------------
// a7: K-buffer OIT composite = VSCube.hlsl:51-76 + PSDepthPeel.hlsl:12-24 + PSCube.hlsl:30-60 (CubeCast, PSCube.hlsli:51-108,
// or the screen-space RayCast, RayCast.hlsli:43-108) + PSResolveOIT.hlsl:12-26.
// The reference rasterises every visible cube with front-face culling and no depth test (MultiRayCaster.cpp:868-869,
// 897-898): one fragment per volume per pixel, where the pixel's view ray exits the cube.  CDNA4 has no rasteriser, so
// each pixel finds those fragments analytically (slab test in volume space), keeps the 8 nearest in registers (the
// InterlockedMin peel), shades them and resolves front to back.  One wavefront = one 8x8 pixel tile.
// ---------------------------------------------------------------------------------------------
struct CubeCoord { uint32_t face; float s, t; };

__device__ __forceinline__ CubeCoord cubeFaceCoords(vec3 d)	// D3D face selection, consistent with GetLocalPos (CSRayMarch.hlsl:36-52)
{
	const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
	CubeCoord c;
	float sc, tc, ma;
	if (ax >= ay && ax >= az) { c.face = d.x > 0.0f ? 0u : 1u; sc = d.x > 0.0f ? -d.z : d.z; tc = -d.y; ma = ax; }
	else if (ay >= az) { c.face = d.y > 0.0f ? 2u : 3u; sc = d.x; tc = d.y > 0.0f ? d.z : -d.z; ma = ay; }
	else { c.face = d.z > 0.0f ? 4u : 5u; sc = d.z > 0.0f ? d.x : -d.x; tc = -d.y; ma = az; }
	c.s = (sc / ma + 1.0f) * 0.5f;
	c.t = (tc / ma + 1.0f) * 0.5f;
	return c;
}

__device__ __forceinline__ int floorToIntClamped(float x, int lo, int hi)
{
	return (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(x), (float)lo), (float)hi);	// NaN -> lo
}

// texel (i, j) of `face`; a texel outside the face resolves to the texel its centre re-projects to on the neighbouring face
__device__ __forceinline__ uint32_t cubeTexelIndex(uint32_t face, int i, int j, int d)
{
	if (i < 0 || j < 0 || i >= d || j >= d)
	{
		const float a = ((float)i + 0.5f) / (float)d * 2.0f - 1.0f, b = ((float)j + 0.5f) / (float)d * 2.0f - 1.0f;
		vec3 dir;
		switch (face)
		{
		case 0: dir = v3(1.0f, -b, -a); break;
		case 1: dir = v3(-1.0f, -b, a); break;
		case 2: dir = v3(a, 1.0f, b); break;
		case 3: dir = v3(a, -1.0f, -b); break;
		case 4: dir = v3(a, -b, 1.0f); break;
		default: dir = v3(-a, -b, -1.0f); break;
		}
		const CubeCoord c = cubeFaceCoords(dir);
		face = c.face;
		i = floorToIntClamped(c.s * (float)d, 0, d - 1);
		j = floorToIntClamped(c.t * (float)d, 0, d - 1);
	}
	return (face * (uint32_t)d + (uint32_t)j) * (uint32_t)d + (uint32_t)i;
}

__device__ __forceinline__ float UnprojectZ(float depth)	// PSCube.hlsli:21-26
{
	return (kZNear * kZFar) / (depth * (kZNear - kZFar) + kZFar);
}

// CubeCast (PSCube.hlsli:51-108) with GetDomain (:31-46)
__device__ __forceinline__ vec4 CubeCast(const uint2* cubeMap, const float* cubeDepth, int d, float depthPix, vec3 pos, vec3 rayDir)
{
	const CubeCoord cc = cubeFaceCoords(pos);
	const float gridSize = (float)d;

	const float u = cc.s * gridSize - 0.5f, v = cc.t * gridSize - 0.5f;
	const float flu = __builtin_floorf(u), flv = __builtin_floorf(v);
	const int i0 = floorToIntClamped(u, -1, d), j0 = floorToIntClamped(v, -1, d);
	const float fx = u - flu, fy = v - flv;
	uint32_t idx[4];
	idx[0] = cubeTexelIndex(cc.face, i0, j0 + 1, d); idx[1] = cubeTexelIndex(cc.face, i0 + 1, j0 + 1, d);
	idx[2] = cubeTexelIndex(cc.face, i0 + 1, j0, d); idx[3] = cubeTexelIndex(cc.face, i0, j0, d);
	vec4 smp[4];
	float z[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) { smp[k] = decodeTexel(cubeMap[idx[k]]); z[k] = cubeDepth[idx[k]]; }

	// SampleLevel fallback colour (PSCube.hlsli:59): plain bilinear over the same footprint
	const float wl[4] = { (1.0f - fx) * fy, fx * fy, fx * (1.0f - fy), (1.0f - fx) * (1.0f - fy) };
	vec4 color;
	color.x = smp[0].x * wl[0] + smp[1].x * wl[1] + smp[2].x * wl[2] + smp[3].x * wl[3];
	color.y = smp[0].y * wl[0] + smp[1].y * wl[1] + smp[2].y * wl[2] + smp[3].y * wl[3];
	color.z = smp[0].z * wl[0] + smp[1].z * wl[1] + smp[2].z * wl[2] + smp[3].z * wl[3];
	color.w = smp[0].w * wl[0] + smp[1].w * wl[1] + smp[2].w * wl[2] + smp[3].w * wl[3];

	// GetDomain
	float uvx = cc.s * gridSize, uvy = cc.t * gridSize;
	float domx = (uvx + 0.5f) - __builtin_floorf(uvx + 0.5f), domy = (uvy + 0.5f) - __builtin_floorf(uvy + 0.5f);
	const float bound = gridSize - 1.0f;
	const vec3 axes = v3(pos.x * gridSize, pos.y * gridSize, pos.z * gridSize);
	if ((__builtin_fabsf(axes.x) > bound && axes.x * rayDir.x < 0.0f) || (__builtin_fabsf(axes.y) > bound && axes.y * rayDir.y < 0.0f)
		|| (__builtin_fabsf(axes.z) > bound && axes.z * rayDir.z < 0.0f))
	{
		uvx = __builtin_fminf(uvx, gridSize - 0.5f); uvy = __builtin_fminf(uvy, gridSize - 0.5f);
		domx = uvx < 0.5f ? 1.0f : 0.0f; domy = uvy < 0.5f ? 1.0f : 0.0f;
	}
	const float dix = 1.0f - domx, diy = 1.0f - domy;
	const float wb[4] = { dix * domy, domx * domy, domx * diy, dix * diy };

	const float depth = UnprojectZ(depthPix);
	vec4 result; result.x = result.y = result.z = result.w = 0.0f;
	float ws = 0.0f;
#pragma unroll
	for (int k = 0; k < 4; ++k)
	{
		const float zi = UnprojectZ(z[k]);
		float w = __builtin_fmaxf(1.0f - 0.5f * __builtin_fabsf(depth - zi), 0.0f);
		w *= wb[k];
		result.x += smp[k].x * w; result.y += smp[k].y * w; result.z += smp[k].z * w; result.w += smp[k].w * w;
		ws += w;
	}
	if (ws > 0.0f) { result.x /= ws; result.y /= ws; result.z /= ws; result.w /= ws; return result; }
	return color;
}

#ifndef MV_COMPOSITE_MIN_WAVES
#define MV_COMPOSITE_MIN_WAVES 4
#endif
__global__ __launch_bounds__(64, MV_COMPOSITE_MIN_WAVES) void compositeKernel(DeviceState st, uint32_t row0, uint32_t row1)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t tilesX = (st.W + 7u) >> 3;
	const uint32_t tileX0 = (blockIdx.x % tilesX) * 8u, tileY0 = row0 + (blockIdx.x / tilesX) * 8u;
	const uint32_t px = tileX0 + (lane & 7u), py = tileY0 + (lane >> 3);
	uint32_t nSamples = 0, nLit = 0;

	if (px < st.W && py < row1)
	{
		const PerObject* perObjects = reinterpret_cast<const PerObject*>(st.perObject);
		const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
		const vec3 eyePt = v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]);
		const uint32_t visibleCount = st.counters[0];

		// PSCube.hlsl:37-39
		float nx = ((float)px + 0.5f) / (float)st.W, ny = ((float)py + 0.5f) / (float)st.H;
		nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
		const float depthPix = st.depthMap[(size_t)py * st.W + px];
		const vec3 dirW = pixelRayWorld(cb, nx, ny);

		// PSDepthPeel.hlsl:12-24: the NUM_OIT_LAYERS nearest fragments, ascending depth; key = depth bits : list index
		unsigned long long keys[kNumOitLayers];
#pragma unroll
		for (uint32_t m = 0; m < kNumOitLayers; ++m) keys[m] = ~0ull;

		for (uint32_t k = 0; k < visibleCount; ++k)
		{
			const uint32_t v = st.visibleVolumes[k];
			// volumes whose screen rectangle misses this 8x8 tile cannot produce a fragment here (wave-uniform skip)
			const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[v];
			if (tileX0 + 7u < rect.x || tileX0 > rect.z || tileY0 + 7u < rect.y || tileY0 > rect.w) continue;
			const PerObject& po = perObjects[v];
			vec3 lpt;
			float z;
			if (!findExitFragment(po, mul43(eyePt, po.worldI), dirW, lpt, z)) continue;

			// sorted insert (stable for equal depths: the list index k is the low word)
			unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | k;
#pragma unroll
			for (uint32_t m = 0; m < kNumOitLayers; ++m)
			{
				const unsigned long long cur = keys[m];
				const bool smaller = key < cur;
				keys[m] = smaller ? key : cur;
				key = smaller ? cur : key;
			}
		}

		// PSCube.hlsl:30-60 per layer, PSResolveOIT.hlsl:12-26
		vec4 result; result.x = result.y = result.z = result.w = 0.0f;
		const uint64_t chain = cubeChainTexels(st.N);
#pragma unroll 1
		for (uint32_t m = 0; m < kNumOitLayers; ++m)
		{
			if (keys[m] == ~0ull) break;
			const uint32_t v = st.visibleVolumes[(uint32_t)(keys[m] & 0xffffffffull)];
			const PerObject& po = perObjects[v];
			const ushort4 vi = reinterpret_cast<const ushort4*>(st.volumeInfo)[v];
			const vec3 localSpaceEyePt = mul43(eyePt, po.worldI);

			vec4 color;
			const uint32_t slot = (vi.z & kCubeMapRayMarchBit) ? 0xffffffffu : st.screenSlot[v];
			if (!(vi.z & kCubeMapRayMarchBit) && slot != 0xffffffffu && py >= st.bandRow0 && py < st.bandRow1)
			{
				// RayCast (RayCast.hlsli:43-108) already marched for this pixel by marchViewKernel's screen tiles
				const float4 c = st.screenLayers[((size_t)slot * st.H + py) * st.W + px];
				color.x = c.x; color.y = c.y; color.z = c.z; color.w = c.w;
			}
			else
			{
				// recompute the exit point exactly as in the peel loop (cheaper than keeping 8 float3 in registers)
				vec3 lpt;
				float z;
				findExitFragment(po, localSpaceEyePt, dirW, lpt, z);
				const vec3 rayDir = lpt - localSpaceEyePt;
				if (!(vi.z & kCubeMapRayMarchBit))
				{
					// RayCast (RayCast.hlsli:43-108), inline fallback (sharded contexts, or more screen-space volumes than slots)
					vec3 ro = localSpaceEyePt;
					const vec3 rd = normalize3(rayDir);
					color.x = color.y = color.z = color.w = 0.0f;
					if (ComputeRayOrigin(ro, rd))
					{
						const float tMax = GetTMax(v3(nx, ny, depthPix), ro, rd, po.wvpI);
						const VolumeView vol = makeVolumeView(st.volumes[vi.w], st.alphaCells[vi.w], st.N, st.bricks);
						color = marchRay(vol, st.lightTable[v], st.L, ro, rd, tMax, vi.y, nSamples, nLit);
					}
				}
				else
				{
					const int dims = (int)(st.N >> vi.x);
					const size_t off = (size_t)v * chain + cubeMipOffset(st.N, vi.x);
					color = CubeCast(st.cubeMaps + off, st.cubeDepths + off, dims, depthPix, lpt, rayDir);
				}
			}
			if (color.w > 0.0f && color.w <= 1.0f)
			{
				// kColors is RGBA16F (MultiRayCaster.cpp:236): each layer is rounded to fp16 before the resolve
				const float cr = half_bits_to_float(float_to_half_bits(color.x)), cg = half_bits_to_float(float_to_half_bits(color.y));
				const float cbl = half_bits_to_float(float_to_half_bits(color.z)), ca = half_bits_to_float(float_to_half_bits(color.w));
				const float it = 1.0f - result.w;
				result.x += cr * it; result.y += cg * it; result.z += cbl * it; result.w += ca * it;
			}
		}
		result.w = __builtin_fminf(result.w, 0.9997f);	// PSResolveOIT.hlsl:23
		uint2 o;
		o.x = (uint32_t)float_to_half_bits(result.x) | ((uint32_t)float_to_half_bits(result.y) << 16);
		o.y = (uint32_t)float_to_half_bits(result.z) | ((uint32_t)float_to_half_bits(result.w) << 16);
		st.composite[(size_t)py * st.W + px] = o;
	}

	uint32_t tot = nSamples;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
	const uint32_t slot = ((row0 >> 3) * tilesX + blockIdx.x);
	if (lane == 0 && tot && slot < st.compositeSlots) st.compositeStats[slot] += tot;
}

void launchComposite(hipStream_t s, const DeviceState& st, uint32_t row0, uint32_t row1)
{
	const uint32_t tilesX = (st.W + 7u) >> 3, tilesY = (row1 - row0 + 7u) >> 3;
	hipLaunchKernelGGL(compositeKernel, dim3(tilesX * tilesY), dim3(64), 0, s, st, row0, row1);
}
} // namespace mv
