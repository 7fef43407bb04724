// mv_mesh_api.cpp -- C-ABI of the mesh depth / shadow producer: what ObjectRenderer hands the path through SetRenderTargets
// (ObjectRenderer.cpp:209-262, VSDepth.hlsl:24-27, VSBasePass.hlsl:44-60), rasterised by the kernels of mv_mesh.hip.
#include "mv_ctx.hpp"

extern "C" {

// ---- mesh depth / shadow producer (ObjectRenderer.cpp:209-262) ----
int mv_mesh_upload(mv_ctx* ctx, const float* positions, uint32_t numVerts, const uint32_t* indices, uint32_t numTris)
{
	if (!ctx || !positions || !indices || !numVerts || !numTris) return MV_ERR_INVALID_ARG;
	for (size_t i = 0; i < (size_t)numTris * 3; ++i) MV_REQUIRE(ctx, indices[i] < numVerts, MV_ERR_RANGE, "mesh index out of range");
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	if (ctx->meshPositions) { (void)hipFree(ctx->meshPositions); ctx->meshPositions = nullptr; }
	if (ctx->meshIndices) { (void)hipFree(ctx->meshIndices); ctx->meshIndices = nullptr; }
	MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->meshPositions), (size_t)numVerts * 3 * sizeof(float)));
	MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->meshIndices), (size_t)numTris * 3 * sizeof(uint32_t)));
	if (!ctx->rasterBits)
	{
		const size_t n = std::max((size_t)ctx->st.W * ctx->st.H, (size_t)ctx->st.M * ctx->st.M);
		MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->rasterBits), n * sizeof(uint32_t)));
		MV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->dMeshMatrix), 16 * sizeof(float)));
	}
	MV_HIP(ctx, hipMemcpy(ctx->meshPositions, positions, (size_t)numVerts * 3 * sizeof(float), hipMemcpyHostToDevice));
	MV_HIP(ctx, hipMemcpy(ctx->meshIndices, indices, (size_t)numTris * 3 * sizeof(uint32_t), hipMemcpyHostToDevice));
	ctx->meshVerts = numVerts; ctx->meshTris = numTris;
	return MV_OK;
}

static int rasterInto(mv_ctx* ctx, const float matT[16], float bx, float by, uint32_t W, uint32_t H)
{
	MV_REQUIRE(ctx, ctx->meshTris > 0, MV_ERR_NOT_READY, "no mesh (mv_mesh_upload)");
	MV_DRAIN(ctx);	// depth / shadow maps may still be read by frames in flight
	MV_HIP(ctx, hipMemcpyAsync(ctx->dMeshMatrix, matT, 16 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	mv::launchFillU32(ctx->stream, ctx->rasterBits, 0x3f800000u, (size_t)W * H);	// cleared to 1.0
	mv::launchRasterDepthBits(ctx->stream, ctx->meshPositions, ctx->meshIndices, ctx->meshTris, ctx->dMeshMatrix, bx, by, W, H, ctx->rasterBits);
	MV_HIP(ctx, hipGetLastError());
	return MV_OK;
}

int mv_mesh_render_depth(mv_ctx* ctx, const float worldViewProjT[16], const float projBias[2])
{
	if (!ctx || !worldViewProjT) return MV_ERR_INVALID_ARG;
	const int rc = rasterInto(ctx, worldViewProjT, projBias ? projBias[0] : 0.0f, projBias ? projBias[1] : 0.0f, ctx->st.W, ctx->st.H);
	if (rc != MV_OK) return rc;
	// the depth bits ARE the D32F values
	MV_HIP(ctx, hipMemcpyAsync(ctx->ownDepth, ctx->rasterBits, (size_t)ctx->st.W * ctx->st.H * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	ctx->st.depthMap = ctx->ownDepth;
	ctx->haveDepth = true;
	return MV_OK;
}

int mv_mesh_render_shadow(mv_ctx* ctx, const float shadowWVPT[16])
{
	if (!ctx || !shadowWVPT) return MV_ERR_INVALID_ARG;
	const int rc = rasterInto(ctx, shadowWVPT, 0.0f, 0.0f, ctx->st.M, ctx->st.M);
	if (rc != MV_OK) return rc;
	mv::launchDepthBitsToD16(ctx->stream, ctx->rasterBits, ctx->ownShadow, (size_t)ctx->st.M * ctx->st.M);
	MV_HIP(ctx, hipGetLastError());
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	ctx->st.shadowMap = ctx->ownShadow;
	ctx->haveShadow = true;
	return MV_OK;
}

int mv_readback_depth_map(mv_ctx* ctx, float* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemcpy(dst, ctx->st.depthMap, (size_t)ctx->st.W * ctx->st.H * sizeof(float), hipMemcpyDeviceToHost));
	return MV_OK;
}

int mv_readback_shadow_map(mv_ctx* ctx, uint16_t* dst)
{
	if (!ctx || !dst) return MV_ERR_INVALID_ARG;
	MV_HIP(ctx, (hipError_t)syncAll(ctx));
	MV_HIP(ctx, hipMemcpy(dst, ctx->st.shadowMap, (size_t)ctx->st.M * ctx->st.M * sizeof(uint16_t), hipMemcpyDeviceToHost));
	return MV_OK;
}

} // extern "C"
