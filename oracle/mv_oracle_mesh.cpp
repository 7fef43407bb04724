/*
 * mv_oracle_mesh.cpp -- CPU ORACLE of the mesh depth / shadow rasteriser (test infrastructure, NOT product code).
 * PARITY: the VERTEX stage (clipPosition below) is pinned to the reference's shipped compiled vertex shaders -- Bin/VSDepth.cso and
 * Bin/VSBasePass.cso run on the CPU through oracle/dxil give the same SV_Position bit for bit for every vertex of the reference scene's
 * mesh (tests/test_oracle_vs_reference_dxil.py::test_mesh_vertex_stage_equals_the_compiled_vertex_shaders).  The RASTER stage is
 * UNPINNED: the reference draws the mesh with the D3D12 rasteriser (ObjectRenderer.cpp:209-262); no reference output exists.  Its rules
 * here, the same as multivolumes_amd/csrc/mv_mesh.hip: 1/256-pixel snapping, pixel centres at +0.5, top-left rule, both windings,
 * fp32 barycentric z, LESS test on a buffer cleared to 1.0.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

extern "C" {

static inline bool topLeft(long long ex, long long ey) { return ey < 0 || (ey == 0 && ex > 0); }

/* VSDepth.hlsl:24-27 / VSBasePass.hlsl:44-60: mul(float4(Pos, 1), g_worldViewProj), then Pos.xy += g_projBias * Pos.w.  As compiled
 * (Bin/VSDepth.cso, Bin/VSBasePass.cso): m0 x, FMad(y, m1, .), FMad(z, m2, .), + m3 -- with FMad as a multiply and an add that is the
 * left-to-right sum below; the bias as bias w + xy.  wvpT: the transposed 4x4 the reference stores in cbPerObject. */
static inline void clipPosition(const float* p, const float* wvpT, float biasX, float biasY, float out[4])
{
	const float px = p[0], py = p[1], pz = p[2];
	float cx = px * wvpT[0] + py * wvpT[1] + pz * wvpT[2] + wvpT[3];
	float cy = px * wvpT[4] + py * wvpT[5] + pz * wvpT[6] + wvpT[7];
	const float cz = px * wvpT[8] + py * wvpT[9] + pz * wvpT[10] + wvpT[11];
	const float cw = px * wvpT[12] + py * wvpT[13] + pz * wvpT[14] + wvpT[15];
	cx = biasX * cw + cx; cy = biasY * cw + cy;
	out[0] = cx; out[1] = cy; out[2] = cz; out[3] = cw;
}

/* the vertex stage alone: SV_Position of every vertex (what VSDepth / VSBasePass output), for the check against the compiled shaders */
void mvo_mesh_clip_positions(const float* positions, uint32_t numVerts, const float* wvpT, float biasX, float biasY, float* out4)
{
	for (uint32_t i = 0; i < numVerts; ++i) clipPosition(positions + 3 * (size_t)i, wvpT, biasX, biasY, out4 + 4 * (size_t)i);
}

/* depth: W x H floats, pre-cleared by the caller (1.0f); wvpT: transposed 4x4 as the reference stores it */
void mvo_raster_depth(const float* positions, const uint32_t* indices, uint32_t numTris, const float* wvpT,
	float biasX, float biasY, uint32_t W, uint32_t H, float* depth)
{
	for (uint32_t t = 0; t < numTris; ++t)
	{
		float sx[3], sy[3], sz[3];
		bool drop = false;
		for (int k = 0; k < 3; ++k)
		{
			const uint32_t vi = indices[3 * t + k];
			float c[4];
			clipPosition(positions + 3 * (size_t)vi, wvpT, biasX, biasY, c);
			const float cx = c[0], cy = c[1], cz = c[2], cw = c[3];
			if (!(cw > 0.0f)) { drop = true; break; }
			const float nx = cx / cw, ny = cy / cw;
			sx[k] = (nx * 0.5f + 0.5f) * (float)W;
			sy[k] = (1.0f - (ny * 0.5f + 0.5f)) * (float)H;
			sz[k] = cz / cw;
		}
		if (drop) continue;
		long long X[3], Y[3];
		for (int k = 0; k < 3; ++k)
		{
			if (!(fabsf(sx[k]) < 1.0e6f) || !(fabsf(sy[k]) < 1.0e6f)) { drop = true; break; }
			X[k] = (long long)rintf(sx[k] * 256.0f);
			Y[k] = (long long)rintf(sy[k] * 256.0f);
		}
		if (drop) continue;
		long long area = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
		if (area == 0) continue;
		int i1 = 1, i2 = 2;
		if (area < 0) { i1 = 2; i2 = 1; area = -area; }
		const long long ax = X[0], ay = Y[0], bx = X[i1], by = Y[i1], cx2 = X[i2], cy2 = Y[i2];
		const float za = sz[0], zb = sz[i1], zc = sz[i2];
		long long minX = ax < bx ? ax : bx; minX = minX < cx2 ? minX : cx2;
		long long maxX = ax > bx ? ax : bx; maxX = maxX > cx2 ? maxX : cx2;
		long long minY = ay < by ? ay : by; minY = minY < cy2 ? minY : cy2;
		long long maxY = ay > by ? ay : by; maxY = maxY > cy2 ? maxY : cy2;
		long long px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
		long long py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
		if (px0 < 0) px0 = 0;
		if (py0 < 0) py0 = 0;
		if (px1 > (long long)W - 1) px1 = (long long)W - 1;
		if (py1 > (long long)H - 1) py1 = (long long)H - 1;
		if (px0 > px1 || py0 > py1) continue;
		const long long e0x = bx - ax, e0y = by - ay, e1x = cx2 - bx, e1y = cy2 - by, e2x = ax - cx2, e2y = ay - cy2;
		const long long bias0 = topLeft(e0x, e0y) ? 0 : -1, bias1 = topLeft(e1x, e1y) ? 0 : -1, bias2 = topLeft(e2x, e2y) ? 0 : -1;
		const float invArea = 1.0f / (float)area;
		for (long long py = py0; py <= py1; ++py)
		{
			const long long cy = py * 256 + 128;
			for (long long px = px0; px <= px1; ++px)
			{
				const long long cx = px * 256 + 128;
				const long long w0 = e0x * (cy - ay) - e0y * (cx - ax);
				const long long w1 = e1x * (cy - by) - e1y * (cx - bx);
				const long long w2 = e2x * (cy - cy2) - e2y * (cx - cx2);
				if (w0 + bias0 < 0 || w1 + bias1 < 0 || w2 + bias2 < 0) continue;
				const float l1 = (float)w1 * invArea, l2 = (float)w2 * invArea, l0 = (float)w0 * invArea;
				const float z = l1 * za + l2 * zb + l0 * zc;
				if (!(z >= 0.0f && z <= 1.0f)) continue;
				float* d = depth + (size_t)py * W + (size_t)px;
				if (z < *d) *d = z;
			}
		}
	}
}

void mvo_depth_to_d16(const float* depth, size_t n, uint16_t* d16)
{
	for (size_t i = 0; i < n; ++i) d16[i] = (uint16_t)(depth[i] * 65535.0f + 0.5f);
}

} // extern "C"
