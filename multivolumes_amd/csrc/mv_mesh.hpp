// mv_mesh.hpp -- launch interface of the mesh depth / shadow rasteriser (mv_mesh.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv
{
// Rasterise numTris indexed triangles through the row-vector matrix stored transposed in wvpT (f[4*c + r] = M[r][c]) into
// a W x H buffer of depth BITS (uint32 of the fp32 depth, LESS test by atomicMin; the caller clears to 0x3f800000).
void launchRasterDepthBits(hipStream_t s, const float* positions, const uint32_t* indices, uint32_t numTris,
	const float* wvpT /* 16 floats, device */, float biasX, float biasY, uint32_t W, uint32_t H, uint32_t* depthBits);
void launchFillU32(hipStream_t s, uint32_t* dst, uint32_t value, size_t n);
void launchDepthBitsToD16(hipStream_t s, const uint32_t* depthBits, uint16_t* d16, size_t n);
} // namespace mv
