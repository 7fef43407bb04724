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
	const uint32_t dv = cubeMipDim(st.N, (info >> 2) & 7u);
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
	const uint32_t dv = cubeMipDim(st.N, (info >> 2) & 7u);
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
// Exchange, compact form (mv_dist_frame): a shard's buffer holds ONLY what the other ranks' composites can read -- the marched
// faces (MaskBits) of its visible volumes on the cube-map scheme, at the selected mip -- back to back in ascending volume order:
// per volume [k faces x d^2 RGBA16F texels][k faces x d^2 R32F depths], faces in mask order.  Every rank derives the same offsets
// from the replicated cull (volumeFlags, exchangeOff); the host only needs the size of the largest shard (counters[12]).
// C4 at the default camera: 10 cube-map volumes x 3 faces x 128^2 x 12 B = 5.9 MB per frame instead of 18.9 MB for whole cubes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t exchangeFaceTexels(uint32_t info, uint32_t N)
{
	if ((info & 3u) != 3u) return 0u;	// visible and on the cube-map scheme
	const uint32_t d = cubeMipDim(N, (info >> 2) & 7u);
	return (uint32_t)__popc((info >> 5) & 0x3fu) * d * d;
}

// One block row per volume.  Owner and offset inside the owner's shard come from the cull (volumeFlags bits 16-31, exchangeOff):
// v % shardCount, or the cost-balanced deal of the multi-GPU frame loop.
template <bool PACK>
__global__ __launch_bounds__(256) void exchangeCompactKernel(DeviceState st, uint8_t* __restrict__ buf, size_t shardBytes)
{
	const uint32_t v = blockIdx.y;
	const uint32_t info = st.volumeFlags[v];
	const uint32_t n = exchangeFaceTexels(info, st.N);
	if (!n) return;
	const uint32_t r = info >> 16;
	if (PACK ? r != st.shardIndex : r == st.shardIndex) return;	// pack what this rank marched; unpack what the others did
	const uint32_t d = cubeMipDim(st.N, (info >> 2) & 7u), mask = (info >> 5) & 0x3fu, face2 = d * d;
	uint8_t* shard = buf + (PACK ? 0 : (size_t)r * shardBytes) + (size_t)st.exchangeOff[v] * 12u;
	uint2* colour = reinterpret_cast<uint2*>(shard);
	float* depth = reinterpret_cast<float*>(shard + (size_t)n * 8u);
	const size_t cube = (size_t)st.cubeOffset[v];
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
	{
		// the k-th marched face
		uint32_t k = i / face2, m = mask, f = 0;
		for (;;) { f = (uint32_t)__ffs((int)m) - 1u; if (!k) break; m &= m - 1u; --k; }
		const size_t texel = cube + (size_t)f * face2 + (i % face2);
		if (PACK) { colour[i] = st.cubeMaps[texel]; depth[i] = st.cubeDepths[texel]; }
		else { st.cubeMaps[texel] = colour[i]; st.cubeDepths[texel] = depth[i]; }
	}
}

void launchExchangeCompact(hipStream_t s, const DeviceState& st, bool pack, void* buf, size_t shardBytes)
{
	const size_t n = (size_t)6 * st.N * st.N;
	const uint32_t gx = (uint32_t)((n + 2047) / 2048);
	if (pack) hipLaunchKernelGGL(exchangeCompactKernel<true>, dim3(gx ? gx : 1, st.V), dim3(256), 0, s, st, reinterpret_cast<uint8_t*>(buf), shardBytes);
	else hipLaunchKernelGGL(exchangeCompactKernel<false>, dim3(gx ? gx : 1, st.V), dim3(256), 0, s, st, reinterpret_cast<uint8_t*>(buf), shardBytes);
}

// ---------------------------------------------------------------------------------------------
// Rotating output sets, one history.  The reference never clears its cube maps: a face that a frame does not march (face bit clear,
// CSRayMarch.hlsl:84-85, a volume that is culled or on the screen-space scheme, another mip) keeps what the last frame that
// marched it left there, and a CubeCast gather across a cube edge can read it.  With frames in flight (and on every rank of the
// multi-GPU loop) consecutive frames write different output sets, so the newest data of such a face sits in the set of an earlier
// frame.  Every (volume, mip, face) carries a stamp per set -- the frame that last marched what the set holds -- and every frame
// first brings EVERY face it does not march up to date from the previous frame's set where that one is newer.  By induction the
// previous frame's set is complete, so one source set is enough however many sets rotate, and every set reads exactly like the
// reference's single one once its frame has marched.  In steady state nothing is copied: a face is copied once per set when it stops
// being marched.  (Texels of a MARCHED face whose ray misses the box by an ulp are not tracked.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copyForwardKernel(DeviceState st, const uint2* __restrict__ otherMaps, const float* __restrict__ otherDepths,
	const uint32_t* __restrict__ otherStamps, uint32_t tag, int marchFollows)
{
	const uint32_t f = blockIdx.x % 6u, mip = (blockIdx.x / 6u) % kNumCubeMip, v = blockIdx.x / (6u * kNumCubeMip);
	const uint32_t info = st.volumeFlags[v];
	const uint32_t idx = (v * kNumCubeMip + mip) * 6u + f;
	// visible, on the cube-map scheme, at this mip, face bit set: marched this frame (here, or by its owner and unpacked here) --
	// unless this frame has no view march at all (a composite-only frame: marchFollows == 0), which leaves every face to the copy
	if (marchFollows && (info & 3u) == 3u && ((info >> 2) & 7u) == mip && (((info >> 5) & 0x3fu) & (1u << f)))
	{
		if (threadIdx.x == 0) st.faceStamps[idx] = tag;
		return;
	}
	const uint32_t mine = st.faceStamps[idx], theirs = otherStamps[idx];
	if ((int32_t)(theirs - mine) <= 0) return;
	const uint32_t d = cubeMipDim(st.N, mip);
	const size_t base = (size_t)v * cubeChainTexels(st.N) + cubeMipOffset(st.N, mip) + (size_t)f * d * d;
	for (uint32_t i = threadIdx.x; i < d * d; i += blockDim.x)
	{
		st.cubeMaps[base + i] = otherMaps[base + i];
		st.cubeDepths[base + i] = otherDepths[base + i];
	}
	__syncthreads();
	if (threadIdx.x == 0) st.faceStamps[idx] = theirs;
}

// The copy-forwards of the frames of one launch group (mv_kernels.hpp DeviceStateBatch; the multi-GPU loop with several frames per
// launch), one after the other inside ONE kernel: a (volume, mip, face) is a workgroup's own through all of them -- frame i's copy reads
// what frame i - 1's wrote -- so the chain needs no launch boundary.  Frame `first`'s source is the set of the frame before the group
// (prev*); then, peers: every rank's marched faces of the group's frames must have landed in this rank's sets before frame i + 1 copies
// from frame i's (the folded wait of the launch: mv_kernels.hpp peerFoldWait; one wavefront per workgroup polls).
__global__ __launch_bounds__(256) void copyForwardBatchKernel(DeviceStateBatch p, uint32_t first, uint32_t n, const uint2* __restrict__ prevMaps, const float* __restrict__ prevDepths,
	const uint32_t* __restrict__ prevStamps, uint4 tags)
{
	const uint32_t f = blockIdx.x % 6u, mip = (blockIdx.x / 6u) % kNumCubeMip, v = blockIdx.x / (6u * kNumCubeMip);
	const uint32_t idx = (v * kNumCubeMip + mip) * 6u + f;
	const uint2* srcMaps = prevMaps; const float* srcDepths = prevDepths; const uint32_t* srcStamps = prevStamps;
#pragma unroll
	for (uint32_t i = 0; i < kMaxBatch; ++i)
	{
		if (i >= n) break;
		const DeviceState& st = p.s[i];
		if (i == 1u) peerFoldWait(p.s[0]);	// (from here on the sources are sets of this group)
		if (i >= first)
		{
			const uint32_t tag = i == 0u ? tags.x : i == 1u ? tags.y : i == 2u ? tags.z : tags.w;
			const uint32_t info = st.volumeFlags[v];
			if ((info & 3u) == 3u && ((info >> 2) & 7u) == mip && (((info >> 5) & 0x3fu) & (1u << f)))
			{
				if (threadIdx.x == 0) st.faceStamps[idx] = tag;	// marched this frame
			}
			else
			{
				const uint32_t mine = st.faceStamps[idx], theirs = srcStamps[idx];
				if ((int32_t)(theirs - mine) > 0)
				{
					const uint32_t d = cubeMipDim(st.N, mip);
					const size_t base = (size_t)v * cubeChainTexels(st.N) + cubeMipOffset(st.N, mip) + (size_t)f * d * d;
					for (uint32_t k = threadIdx.x; k < d * d; k += blockDim.x)
					{
						st.cubeMaps[base + k] = srcMaps[base + k];
						st.cubeDepths[base + k] = srcDepths[base + k];
					}
					__syncthreads();
					if (threadIdx.x == 0) st.faceStamps[idx] = theirs;
				}
			}
			__threadfence_block();
			__syncthreads();	// (what this workgroup wrote is what it reads as the next frame's source)
		}
		srcMaps = st.cubeMaps; srcDepths = st.cubeDepths; srcStamps = st.faceStamps;
	}
}

void launchCopyForwardBatch(hipStream_t s, const DeviceState* states, uint32_t first, uint32_t n, const uint2* prevMaps, const float* prevDepths, const uint32_t* prevStamps, const uint32_t* tags)
{
	DeviceStateBatch p;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i] = states[i < n ? i : n - 1u];
	const uint4 t = make_uint4(tags[0], n > 1u ? tags[1] : 0u, n > 2u ? tags[2] : 0u, n > 3u ? tags[3] : 0u);
	hipLaunchKernelGGL(copyForwardBatchKernel, dim3(states[0].V * kNumCubeMip * 6u), dim3(256), 0, s, p, first, n, prevMaps, prevDepths, prevStamps, t);
}

void launchCopyForward(hipStream_t s, const DeviceState& st, const uint2* otherMaps, const float* otherDepths, const uint32_t* otherStamps, uint32_t tag, bool marchFollows)
{
	hipLaunchKernelGGL(copyForwardKernel, dim3(st.V * kNumCubeMip * 6u), dim3(256), 0, s, st, otherMaps, otherDepths, otherStamps, tag, marchFollows ? 1 : 0);
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

#ifndef MV_COMPOSITE_LDS
#define MV_COMPOSITE_LDS 0
#endif
#ifndef MV_COMPOSITE_MIN_WAVES
#define MV_COMPOSITE_MIN_WAVES 4
#endif
// a layer's colour as the K-buffer holds it: kColors is RGBA16F (MultiRayCaster.cpp:236), so every layer is rounded to fp16 before
// the resolve; a colour that fails PSResolveOIT's test (alpha outside (0, 1]) is stored as zeros, which the resolve adds as nothing
__device__ __forceinline__ uint2 packLayer(vec4 color)
{
	uint2 o; o.x = 0u; o.y = 0u;
	if (color.w > 0.0f && color.w <= 1.0f)
	{
		o.x = (uint32_t)float_to_half_bits(color.x) | ((uint32_t)float_to_half_bits(color.y) << 16);
		o.y = (uint32_t)float_to_half_bits(color.z) | ((uint32_t)float_to_half_bits(color.w) << 16);
	}
	return o;
}
constexpr uint32_t kLayerNeedsMarch = 0xffff0000u;	// in .y: alpha = an fp16 NaN, which packLayer never produces

// MARCH = false: the launch over every 8x8 tile of the row range.  It leaves out the inline RayCast march (which alone costs the
// kernel half of its wavefronts per SIMD): a tile in which some surviving layer needs that march is appended to `retry`
// ({count, finished workgroups, tile ...}) and written by the MARCH = true launch behind it, which walks that list.
// PEER: the peer-store transport's instantiation with a folded wait (mv_kernels.hpp peerFoldWait); the others are untouched by it
template <bool MARCH, bool PEER>
__device__ __forceinline__ void compositeBody(const DeviceState& st, uint32_t row0, uint32_t row1, const uint2* __restrict__ background, uint32_t* __restrict__ retry);

// PEER (the peer-store transport's instantiation; the others are untouched by it): a folded wait in front (every rank's cube faces of this
// frame have landed in this rank's set) and every pixel the launch writes also stored into the other ranks' frame buffers -- rank 0's only
// with fold.rowsRoot: no kernel of its own for the row push (round 4's peerPushRowsKernel: 15 us alone, 60-190 us beside the marches of
// the other streams).  The flags that announce the rows are raised by a kernel of one wavefront BEHIND the composite's launches: a
// wavefront's s_endpgm waits for its stores, so the kernel boundary says they are acknowledged -- raising them from the composite's last
// workgroup (every wavefront drains, counts in) was measured slower (profiles/r05_rank_chain.txt)
template <bool MARCH, bool PEER = false>
__global__ __launch_bounds__(64, MARCH ? 4 : MV_COMPOSITE_MIN_WAVES) void compositeKernel(DeviceState st, uint32_t row0, uint32_t row1, const uint2* __restrict__ background,
	uint32_t* __restrict__ retry)
{
	if constexpr (PEER) if (!MARCH) peerFoldWait(st);
	compositeBody<MARCH, PEER>(st, row0, row1, background, retry);
}

// several frames in one launch (mv_kernels.hpp DeviceStateBatch; the multi-GPU loop's interleaved tile rows, no background): blockIdx.y is the frame
template <bool MARCH>
__global__ __launch_bounds__(64, MARCH ? 4 : MV_COMPOSITE_MIN_WAVES) void compositeBatchKernel(DeviceStateBatch p)
{
	const DeviceState& st = p.s[blockIdx.y];
	if (!MARCH) peerFoldWait(st);
	compositeBody<MARCH, true>(st, st.bandRow0, st.bandRow1, nullptr, st.compositeRetry);
}

template <bool MARCH, bool PEER>
__device__ __forceinline__ void compositeBody(const DeviceState& st, uint32_t row0, uint32_t row1, const uint2* __restrict__ background, uint32_t* __restrict__ retry)
{
	const uint32_t lane = threadIdx.x;
	const uint32_t tilesX = (st.W + 7u) >> 3;
	const uint32_t nRetry = MARCH ? retry[0] : 0u;
	for (uint32_t item = blockIdx.x; !MARCH || item < nRetry; item += gridDim.x)
	{
	const uint32_t tileIndex = MARCH ? retry[2u + item] : blockIdx.x;
	// rows [row0, row1) -- or, in the multi-GPU frame loop, this rank's tile rows (every shardCount-th; stored rank-major, see kFlagInterleaveRows)
	const bool interleaved = st.flags & kFlagInterleaveRows;
	const uint32_t tileRow = tileIndex / tilesX;
	const uint32_t tileX0 = (tileIndex % tilesX) * 8u, tileY0 = interleaved ? (st.shardIndex + tileRow * st.shardCount) * 8u : row0 + tileRow * 8u;
	const uint32_t storeY0 = interleaved ? (st.shardIndex * st.rowsPerRank + tileRow) * 8u : tileY0;
	if (interleaved) row1 = st.H;
	const uint32_t px = tileX0 + (lane & 7u), py = tileY0 + (lane >> 3);
	uint32_t nSamples = 0, nLit = 0;
	bool deferred = false;
	uint2 outPixel; outPixel.x = outPixel.y = 0u;	// (PEER) what this lane wrote into the frame buffer, if it did
	bool wrote = false;

	if (px < st.W && py < row1)
	{
		const PerObject* perObjects = reinterpret_cast<const PerObject*>(st.perObject);
		const CBPerFrame* cb = reinterpret_cast<const CBPerFrame*>(st.cbPerFrame);
		const vec3 eyePt = v3(cb->eyePos[0], cb->eyePos[1], cb->eyePos[2]);
		const uint32_t visibleCount = st.counters[0];
		const uint64_t chain = cubeChainTexels(st.N);

		// PSCube.hlsl:37-39
		float nx = ((float)px + 0.5f) / (float)st.W, ny = ((float)py + 0.5f) / (float)st.H;
		nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
		const float depthPix = st.depthMap[(size_t)py * st.W + px];
		const vec3 dirW = pixelRayWorld(cb, nx, ny);
		const bool inBand = interleaved || (py >= st.bandRow0 && py < st.bandRow1);

		// PSDepthPeel.hlsl:12-24: the NUM_OIT_LAYERS nearest fragments, ascending depth; key = depth bits : list index.
		// A fragment's colour (PSCube.hlsl:30-60) does not depend on the other fragments, so it is shaded here, where the volume
		// is wave-uniform (its constants come through scalar loads and the cube-map gathers of successive volumes do not wait
		// for one another), and travels through the sorted insert with its key; only the resolve below is ordered.  The one
		// expensive kind -- a screen-space volume without a marched layer -- is left to the layers that survive the peel.
		unsigned long long keys[kNumOitLayers];
		uint2 layers[kNumOitLayers];
#pragma unroll
		for (uint32_t m = 0; m < kNumOitLayers; ++m) { keys[m] = ~0ull; layers[m].x = layers[m].y = 0u; }
		bool needMarch = false;

		for (uint32_t k = 0; k < visibleCount; ++k)
		{
			const uint32_t v = st.visibleVolumes[k];
			// volumes whose screen rectangle misses this 8x8 tile cannot produce a fragment here (wave-uniform skip)
			const ushort4 rect = reinterpret_cast<const ushort4*>(st.volumeRects)[v];
			if (tileX0 + 7u < rect.x || tileX0 > rect.z || tileY0 + 7u < rect.y || tileY0 > rect.w) continue;
			const PerObject& po = perObjects[v];
			const vec3 localSpaceEyePt = mul43(eyePt, po.worldI);
			vec3 lpt;
			float z;
			if (!findExitFragment(po, localSpaceEyePt, dirW, lpt, z)) continue;

			const ushort4 vi = reinterpret_cast<const ushort4*>(st.volumeInfo)[v];
			uint2 layer;
			if (vi.z & kCubeMapRayMarchBit)
			{
				const int dims = (int)cubeMipDim(st.N, vi.x);
				const size_t off = (size_t)v * chain + cubeMipOffset(st.N, vi.x);
				layer = packLayer(CubeCast(st.cubeMaps + off, st.cubeDepths + off, dims, depthPix, lpt, lpt - localSpaceEyePt));
			}
			else
			{
				const uint32_t slot = st.screenSlot[v];
				if (slot != 0xffffffffu && inBand && !(st.flags & kFlagNoScreenLayers))
				{
					// RayCast (RayCast.hlsli:43-108) already marched for this pixel by the view march's screen tiles
					const float4 c = st.screenLayers[((size_t)slot * st.H + py) * st.W + px];
					vec4 color; color.x = c.x; color.y = c.y; color.z = c.z; color.w = c.w;
					layer = packLayer(color);
				}
				else { layer.x = 0u; layer.y = kLayerNeedsMarch; needMarch = true; }
			}

			// sorted insert (stable for equal depths: the list index k is the low word)
			unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | k;
#pragma unroll
			for (uint32_t m = 0; m < kNumOitLayers; ++m)
			{
				const unsigned long long cur = keys[m];
				const uint2 curLayer = layers[m];
				const bool smaller = key < cur;
				keys[m] = smaller ? key : cur;
				layers[m].x = smaller ? layer.x : curLayer.x; layers[m].y = smaller ? layer.y : curLayer.y;
				key = smaller ? cur : key;
				layer.x = smaller ? curLayer.x : layer.x; layer.y = smaller ? curLayer.y : layer.y;
			}
		}

		// RayCast (RayCast.hlsli:43-108) inline for the surviving layers of screen-space volumes that the view march did not march
		// for this pixel (sharded contexts outside their row band, or more screen-space volumes than slots): rare, per-lane volumes
		if (!MARCH) deferred = __any(needMarch);
		else if (__any(needMarch))
		{
#pragma unroll 1
			for (uint32_t m = 0; m < kNumOitLayers; ++m)
			{
				unsigned long long key = keys[0];
				uint32_t tag = layers[0].y;
#pragma unroll
				for (uint32_t j = 1; j < kNumOitLayers; ++j) if (m == j) { key = keys[j]; tag = layers[j].y; }
				if (key == ~0ull || tag != kLayerNeedsMarch) continue;
				const uint32_t v = st.visibleVolumes[(uint32_t)(key & 0xffffffffull)];
				const PerObject& po = perObjects[v];
				const ushort4 vi = reinterpret_cast<const ushort4*>(st.volumeInfo)[v];
				const vec3 localSpaceEyePt = mul43(eyePt, po.worldI);
				vec3 lpt;
				float z;
				findExitFragment(po, localSpaceEyePt, dirW, lpt, z);	// the exit point exactly as in the peel loop
				vec3 ro = localSpaceEyePt;
				const vec3 rd = normalize3(lpt - localSpaceEyePt);
				vec4 color; color.x = color.y = color.z = color.w = 0.0f;
				if (ComputeRayOrigin(ro, rd))
				{
					const float tMax = GetTMax(v3(nx, ny, depthPix), ro, rd, po.wvpI);
					const VolumeView vol = makeVolumeView(st.volumes[vi.w], st.alphaCells[vi.w], st.N, st.bricks);
					color = marchRay(vol, st.lightTable[v], st.L, ro, rd, tMax, vi.y, nSamples, nLit);
				}
				const uint2 layer = packLayer(color);
#pragma unroll
				for (uint32_t j = 0; j < kNumOitLayers; ++j) if (m == j) layers[j] = layer;
			}
		}

		if (!deferred)
		{
			// PSResolveOIT.hlsl:12-26: front to back over the layers
			vec4 result; result.x = result.y = result.z = result.w = 0.0f;
	#pragma unroll
			for (uint32_t m = 0; m < kNumOitLayers; ++m)
			{
				const uint2 c = layers[m];	// zeros where there is no layer
				const float cr = half_bits_to_float((uint16_t)(c.x & 0xffffu)), cg = half_bits_to_float((uint16_t)(c.x >> 16));
				const float cbl = half_bits_to_float((uint16_t)(c.y & 0xffffu)), ca = half_bits_to_float((uint16_t)(c.y >> 16));
				const float it = 1.0f - result.w;
				result.x += cr * it; result.y += cg * it; result.z += cbl * it; result.w += ca * it;
			}
			result.w = __builtin_fminf(result.w, 0.9997f);	// PSResolveOIT.hlsl:23
			if (background)
			{
				// the resolve is drawn over the scene's colour target with the PREMULTIPLITED blend preset (MultiRayCaster.cpp:931):
				// dst = src + dst * (1 - src.a), colour and alpha alike, blended in fp32 and rounded to the target's format once
				const vec4 dst = decodeTexel(background[(size_t)py * st.W + px]);
				const float inv = 1.0f - result.w;
				result.x = result.x + dst.x * inv; result.y = result.y + dst.y * inv; result.z = result.z + dst.z * inv; result.w = result.w + dst.w * inv;
			}
			uint2 o;
			o.x = (uint32_t)float_to_half_bits(result.x) | ((uint32_t)float_to_half_bits(result.y) << 16);
			o.y = (uint32_t)float_to_half_bits(result.z) | ((uint32_t)float_to_half_bits(result.w) << 16);
			st.composite[(size_t)(storeY0 + (lane >> 3)) * st.W + px] = o;
			if constexpr (PEER) { outPixel = o; wrote = true; }
		}
	}
	if constexpr (PEER)
	{
		// Peer-store transport: the pixels this wavefront wrote, into the other ranks' frame buffers (rank 0's only: fold.rowsRoot) at the same
		// place of the rank-major image -- 16 bytes per store where an even lane and its right neighbour both wrote (a tile row is 8 lanes = 8
		// pixels of a screen row: 2 x 8 bytes); narrower write-through stores are a fabric write each
		if (st.fold.stores && st.peerCount > 1u)	// (wave-uniform)
		{
			const size_t index = (size_t)(storeY0 + (lane >> 3)) * st.W + px;
			const uint2 right = make_uint2(__shfl_down(outPixel.x, 1), __shfl_down(outPixel.y, 1));
			const bool wroteRight = __shfl_down(wrote ? 1 : 0, 1) != 0, wroteLeft = __shfl_up(wrote ? 1 : 0, 1) != 0;
			const bool even = !(lane & 1u);
			const bool pair = even && wrote && wroteRight && !(index & 1u);
			const bool single = wrote && !pair && !(!even && wroteLeft && !((index - 1u) & 1u));	// (an odd lane whose left neighbour took it along stores nothing)
			const uint32_t p0 = st.fold.rowsRoot ? 0u : 0u, p1 = st.fold.rowsRoot ? 1u : st.peerCount;
			for (uint32_t p = p0; p < p1; ++p)
			{
				if (p == st.shardIndex) continue;
				uint2* dst = st.peerComposite[p];
				if (pair) peerStore16(dst + index, make_uint4(outPixel.x, outPixel.y, right.x, right.y));
				else if (single) peerStore8(dst + index, outPixel);
			}
		}
	}

	if (!MARCH)
	{
		if (__any(deferred) && lane == 0) retry[2u + atomicAdd(retry, 1u)] = tileIndex;
		break;
	}
	uint32_t tot = nSamples;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
	const uint32_t slot = ((row0 >> 3) * tilesX + tileIndex);
	if (lane == 0 && tot && slot < st.compositeSlots) st.compositeStats[slot] += tot;
	}
	if (MARCH && lane == 0)
	{
		// the last workgroup to finish empties the list for the next composite into this output set
		__threadfence();
		if (atomicAdd(retry + 1, 1u) == gridDim.x - 1u) { retry[0] = 0u; retry[1] = 0u; }
	}
}

uint32_t launchComposite(hipStream_t s, const DeviceState& stIn, uint32_t row0, uint32_t row1, const void* background)
{
	DeviceState st = stIn;
	const bool interleaved = st.flags & kFlagInterleaveRows;	// (row0 / row1 unused: the rank's own tile rows)
	const uint32_t tilesX = (st.W + 7u) >> 3, T = (st.H + 7u) >> 3;
	const uint32_t tilesY = interleaved ? (st.shardIndex < T ? (T - st.shardIndex + st.shardCount - 1u) / st.shardCount : 0u) : (row1 - row0 + 7u) >> 3;
	if (!tilesY) return 0u;
	// The inline march can only be asked for by a screen-space volume without a slot, by rows outside this context's band (what the
	// view march's screen tiles cover) or by a frame without a view march: when none of them can happen the second launch is left out.
	const bool second = st.screenSlots < st.V || (!interleaved && (row0 < st.bandRow0 || row1 > st.bandRow1)) || (st.flags & kFlagNoScreenLayers);
	// (peer-store transport: the first launch waits -- folded -- and both store their pixels to the other ranks)
	if (st.fold.nWait || st.fold.stores)
	{
		hipLaunchKernelGGL((compositeKernel<false, true>), dim3(tilesX * tilesY), dim3(64), MV_COMPOSITE_LDS, s, st, row0, row1, reinterpret_cast<const uint2*>(background), st.compositeRetry);
		if (!second) return 1u;
		st.fold.nWait = 0u;
		hipLaunchKernelGGL((compositeKernel<true, true>), dim3(512), dim3(64), 0, s, st, row0, row1, reinterpret_cast<const uint2*>(background), st.compositeRetry);
		return 2u;
	}
	hipLaunchKernelGGL(compositeKernel<false>, dim3(tilesX * tilesY), dim3(64), MV_COMPOSITE_LDS, s, st, row0, row1, reinterpret_cast<const uint2*>(background), st.compositeRetry);
	if (!second) return 1u;
	hipLaunchKernelGGL(compositeKernel<true>, dim3(512), dim3(64), 0, s, st, row0, row1, reinterpret_cast<const uint2*>(background), st.compositeRetry);
	return 2u;
}

uint32_t launchCompositeBatch(hipStream_t s, const DeviceState* states, uint32_t n)
{
	DeviceStateBatch p;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i] = states[i < n ? i : n - 1u];
	const DeviceState& st = states[0];
	const uint32_t tilesX = (st.W + 7u) >> 3, T = (st.H + 7u) >> 3;
	const uint32_t tilesY = st.shardIndex < T ? (T - st.shardIndex + st.shardCount - 1u) / st.shardCount : 0u;
	if (!tilesY) return 0u;
	bool second = false;
	for (uint32_t i = 0; i < n; ++i) second = second || states[i].screenSlots < states[i].V || (states[i].flags & kFlagNoScreenLayers);
	// (the first launch waits, both store their pixels to the other ranks)
	hipLaunchKernelGGL(compositeBatchKernel<false>, dim3(tilesX * tilesY, n), dim3(64), MV_COMPOSITE_LDS, s, p);
	if (!second) return 1u;
	for (uint32_t i = 0; i < kMaxBatch; ++i) p.s[i].fold.nWait = 0u;
	hipLaunchKernelGGL(compositeBatchKernel<true>, dim3(512, n), dim3(64), 0, s, p);
	return 2u;
}
} // namespace mv
