// mv_mesh.hip -- software rasteriser for the mesh depth / shadow maps the ray-march path reads.
//
// Stands in for ObjectRenderer's two raster passes (ObjectRenderer.cpp:209-262; VSDepth.hlsl:24-27 for the shadow map,
// VSBasePass.hlsl:44-60 for the depth map incl. the TAA jitter `Pos.xy += g_projBias * Pos.w`).  CDNA4 has no rasteriser:
// one thread per triangle walks the triangle's pixel bounding box with integer edge functions.
// Rules (mirrored by the oracle, oracle/mv_oracle_mesh.cpp): vertices snapped to 1/256 pixel (D3D's 8 sub-pixel bits),
// pixel centres at +0.5, top-left fill rule, both windings drawn (depth LESS makes back-face culling irrelevant for closed
// meshes), z interpolated in screen space from fp32 barycentrics, near/far clipped per pixel (0 <= z <= 1), triangles
// with a vertex at w <= 0 dropped.
#include "mv_mesh.hpp"

namespace mv
{
__device__ __forceinline__ bool topLeft(long long ex, long long ey)	// edge (ex, ey) = b - a, y down, clockwise interior
{
	return ey < 0 || (ey == 0 && ex > 0);
}

__global__ __launch_bounds__(256) void rasterDepthKernel(const float* __restrict__ positions, const uint32_t* __restrict__ indices, uint32_t numTris,
	const float* __restrict__ wvpT, float biasX, float biasY, uint32_t W, uint32_t H, uint32_t* __restrict__ depthBits)
{
	const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= numTris) return;

	float sx[3], sy[3], sz[3];
#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		const uint32_t vi = indices[3 * t + k];
		const float px = positions[3 * vi], py = positions[3 * vi + 1], pz = positions[3 * vi + 2];
		// mul(float4(Pos, 1), g_worldViewProj), left to right
		float cx = px * wvpT[0] + py * wvpT[1] + pz * wvpT[2] + wvpT[3];
		float cy = px * wvpT[4] + py * wvpT[5] + pz * wvpT[6] + wvpT[7];
		const float cz = px * wvpT[8] + py * wvpT[9] + pz * wvpT[10] + wvpT[11];
		const float cw = px * wvpT[12] + py * wvpT[13] + pz * wvpT[14] + wvpT[15];
		if (!(cw > 0.0f)) return;
		cx += biasX * cw; cy += biasY * cw;	// VSBasePass.hlsl:57
		const float nx = cx / cw, ny = cy / cw;
		sx[k] = (nx * 0.5f + 0.5f) * (float)W;
		sy[k] = (1.0f - (ny * 0.5f + 0.5f)) * (float)H;
		sz[k] = cz / cw;
	}
	// 24.8 fixed point; reject anything absurdly far outside the viewport (guard band) so the products fit 64 bits
	long long X[3], Y[3];
#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		if (!(__builtin_fabsf(sx[k]) < 1.0e6f) || !(__builtin_fabsf(sy[k]) < 1.0e6f)) return;
		X[k] = (long long)__builtin_rintf(sx[k] * 256.0f);
		Y[k] = (long long)__builtin_rintf(sy[k] * 256.0f);
	}
	long long area = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
	if (area == 0) return;
	int i1 = 1, i2 = 2;
	if (area < 0) { i1 = 2; i2 = 1; area = -area; }	// make the winding clockwise on screen (y down) so the interior is E >= 0
	const long long ax = X[0], ay = Y[0], bx = X[i1], by = Y[i1], cx2 = X[i2], cy2 = Y[i2];
	const float za = sz[0], zb = sz[i1], zc = sz[i2];

	long long minX = ax < bx ? ax : bx; minX = minX < cx2 ? minX : cx2;
	long long maxX = ax > bx ? ax : bx; maxX = maxX > cx2 ? maxX : cx2;
	long long minY = ay < by ? ay : by; minY = minY < cy2 ? minY : cy2;
	long long maxY = ay > by ? ay : by; maxY = maxY > cy2 ? maxY : cy2;
	// pixel p is covered if its centre (p + 0.5) * 256 lies inside: candidate range
	long long px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
	long long py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
	if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
	if (px1 > (long long)W - 1) px1 = (long long)W - 1; if (py1 > (long long)H - 1) py1 = (long long)H - 1;
	if (px0 > px1 || py0 > py1) return;

	// edge functions E_ab(p) = (bx - ax)(py - ay) - (by - ay)(px - ax), interior >= 0 for this winding
	const long long e0x = bx - ax, e0y = by - ay, e1x = cx2 - bx, e1y = cy2 - by, e2x = ax - cx2, e2y = ay - cy2;
	const long long bias0 = topLeft(e0x, e0y) ? 0 : -1, bias1 = topLeft(e1x, e1y) ? 0 : -1, bias2 = topLeft(e2x, e2y) ? 0 : -1;
	const float invArea = 1.0f / (float)area;
	for (long long py = py0; py <= py1; ++py)
	{
		const long long cy = py * 256 + 128;
		for (long long px = px0; px <= px1; ++px)
		{
			const long long cx = px * 256 + 128;
			const long long w0 = e0x * (cy - ay) - e0y * (cx - ax);	// opposite vertex c
			const long long w1 = e1x * (cy - by) - e1y * (cx - bx);	// opposite vertex a
			const long long w2 = e2x * (cy - cy2) - e2y * (cx - cx2);	// opposite vertex b
			if (w0 + bias0 < 0 || w1 + bias1 < 0 || w2 + bias2 < 0) continue;
			const float l1 = (float)w1 * invArea, l2 = (float)w2 * invArea, l0 = (float)w0 * invArea;
			const float z = l1 * za + l2 * zb + l0 * zc;
			if (!(z >= 0.0f && z <= 1.0f)) continue;
			atomicMin(depthBits + (size_t)py * W + (size_t)px, __float_as_uint(z));
		}
	}
}

__global__ __launch_bounds__(256) void fillU32Kernel(uint32_t* dst, uint32_t value, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = value;
}

// D16_UNORM store of a [0,1] depth: round to nearest (D3D float -> unorm: x * 65535 + 0.5, truncated)
__global__ __launch_bounds__(256) void depthBitsToD16Kernel(const uint32_t* __restrict__ depthBits, uint16_t* __restrict__ d16, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
	{
		const float z = __uint_as_float(depthBits[i]);
		d16[i] = (uint16_t)(z * 65535.0f + 0.5f);
	}
}

static inline uint32_t gridFor(size_t total, uint32_t block)
{
	const size_t g = (total + block - 1) / block;
	return (uint32_t)(g < 8192 ? (g ? g : 1) : 8192);
}

void launchRasterDepthBits(hipStream_t s, const float* positions, const uint32_t* indices, uint32_t numTris,
	const float* wvpT, float biasX, float biasY, uint32_t W, uint32_t H, uint32_t* depthBits)
{
	if (!numTris) return;
	hipLaunchKernelGGL(rasterDepthKernel, dim3((numTris + 255u) / 256u), dim3(256), 0, s, positions, indices, numTris, wvpT, biasX, biasY, W, H, depthBits);
}

void launchFillU32(hipStream_t s, uint32_t* dst, uint32_t value, size_t n)
{
	hipLaunchKernelGGL(fillU32Kernel, dim3(gridFor(n, 256)), dim3(256), 0, s, dst, value, n);
}

void launchDepthBitsToD16(hipStream_t s, const uint32_t* depthBits, uint16_t* d16, size_t n)
{
	hipLaunchKernelGGL(depthBitsToD16Kernel, dim3(gridFor(n, 256)), dim3(256), 0, s, depthBits, d16, n);
}
} // namespace mv
