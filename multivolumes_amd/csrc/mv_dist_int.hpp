// mv_dist_int.hpp -- what the two translation units of the multi-GPU frame loop share (not part of the ABI): mv_dist.cpp (the loop, the
// collective transports, the exchange building blocks) and mv_dist_peer.cpp (the peer-store transport and its several-frames-per-launch form).
#ifndef MV_DIST_INT_HPP
#define MV_DIST_INT_HPP
#include "mv_ctx.hpp"

// ---- mv_dist.cpp ----
extern "C" {	// (defined inside mv_dist.cpp's extern "C" block)
MV_INTERNAL int distAlloc(mv_ctx* ctx, uint8_t** out, size_t bytes, hipStream_t zeroOn = nullptr);	// a device buffer of the loop (freed by distRelease)
MV_INTERNAL int distCommonInit(mv_ctx* ctx);	// what every transport's init ends with: buffers, reports, the flags of a sharded context (releases everything on failure)
}

// ---- mv_dist_peer.cpp ----
// stages a rank raises a flag for (rows of the flag block = stage x chain) and the launches that carry folded waits / signals
enum { kStageLight = 0, kStageCube = 1, kStageRows = 2, kStageDone = 3, kStageRebuilt = 4 /* local: this rank has rebuilt the light cells of the other ranks' layers from their texels */ };
enum { kFoldLight = 0, kFoldView = 1, kFoldComposite = 2, kFoldRebuild = 3 };
struct PeerWaits { uint32_t n = 0; int stage[8]; uint64_t frame[8]; };
MV_INTERNAL mv::PeerSync peerSyncBase(mv_ctx* ctx, uint32_t frameTag, uint32_t slot);
MV_INTERNAL void foldBase(mv_ctx* ctx, DeviceState& st, uint32_t kind, uint64_t frame, uint32_t frameTag, uint32_t slot);
MV_INTERNAL void foldWait(mv_ctx* ctx, DeviceState& st, int stage, uint64_t frame);
MV_INTERNAL void foldSignal(mv_ctx* ctx, DeviceState& st, int stage, uint64_t frame);
MV_INTERNAL void foldClear(DeviceState& st);
MV_INTERNAL int peerWaitStage(mv_ctx* ctx, hipStream_t s, int stage, uint64_t frame, uint32_t frameTag, uint32_t slot);
MV_INTERNAL int peerCheckReports(mv_ctx* ctx);
MV_INTERNAL int peerWaitFor(mv_ctx* ctx, hipStream_t s, DeviceState& st, uint32_t kind, const PeerWaits& w, uint64_t frame, uint32_t frameTag, uint32_t slot, bool compareCulls);
MV_INTERNAL DeviceState peerPushState(mv_ctx* ctx, const DeviceState& st, uint32_t kind, uint64_t frame, uint32_t frameTag, uint32_t slot);
MV_INTERNAL int peerLight(mv_ctx* ctx);
MV_INTERNAL int peerBeforeView(mv_ctx* ctx);
MV_INTERNAL int peerSelfTest(mv_ctx* ctx);
// several frames per launch
MV_INTERNAL bool batchMode(const mv_ctx* ctx);
MV_INTERNAL int distFrameBatch(mv_ctx* ctx, const void* cbPerFrame, const void* perObject, int with_composite);
MV_INTERNAL int batchClose(mv_ctx* ctx);	// light march + view march of the group being collected
MV_INTERNAL int batchTail(mv_ctx* ctx);	// copy-forwards, composite, row push of the group before it
#endif
