// dist_rank_cost.cpp -- what one rank of the multi-GPU frame loop costs per frame from a C++ host, WITHOUT a fabric: sharded contexts
// (rank r of `world`) on the one GPU there is, mv_dist_frame with do-nothing collectives.  The images are wrong (nothing is exchanged);
// the kernels a rank runs per frame and their scheduling are the real ones, and so is the host cost of the loop -- with no Python in it
// (scripts/exp_rank_cost.py is the same measurement through the Python mirror).  max over ranks = the frame time an ideal fabric would
// give; its ratio to world 1 is the ceiling of the strong-scaling curve.
// Build: g++ -O2 -std=c++17 examples/dist_rank_cost.cpp -Lmultivolumes_amd -lmultivolumes_amd -Wl,-rpath,$PWD/multivolumes_amd -o /tmp/dist_rank_cost
// Usage: /tmp/dist_rank_cost [frames = 400] [worlds = 1,2,4,8 as a bit mask: 15] [one rank only, -1 = all] [frames in flight = 3 (0)] [C3 | C5 | C2] [callbacks | peer | peer-nofold | peer4 | peer4-root | peer4x2 | peer4x4 | peer4x4-root]      (config C3: 16 x 256^3, 1024^2, 128 steps / 96 light steps)
// "peer": the peer-store transport in loopback (MV_DIST_PEER_LOOPBACK): the rank's kernels store into all `world` copies -- every copy being
// its own buffer -- and it raises the other ranks' flags itself: no pack / gather / unpack kernels, the flag kernels instead.
// Prints per rank: ms per frame, and in brackets the host's share (time spent inside DistFrame: constants + every HIP call of the frame).
#include "../multivolumes_amd/csrc/MultiRayCaster.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static int noGather(void*, const void*, void*, size_t, void*, int) { return 0; }
static int noReduce(void*, void*, size_t, void*, int) { return 0; }

int main(int argc, char** argv)
{
	const bool c5 = argc > 5 && std::string(argv[5]) == "C5";	// BASELINE config C5: 64 volumes, 2048^2, 256 steps (default C3)
	const bool c2 = argc > 5 && std::string(argv[5]) == "C2";	// BASELINE config C2: 4 volumes of 128^3, 512^2, 64 steps (a frame that cannot fill the GPU even on one rank)
	const uint32_t N = c2 ? 128 : 256, L = 96, V = c5 ? 64 : c2 ? 4 : 16, W = c5 ? 2048 : c2 ? 512 : 1024, H = W, M = 1024;
	const bool peer = argc > 6 && std::string(argv[6]).rfind("peer", 0) == 0;
	const bool noFold = argc > 6 && std::string(argv[6]) == "peer-nofold";
	const bool root = argc > 6 && std::string(argv[6]) == "peer4-root";	// ... and the rows to rank 0 only (MV_DIST_PEER_ROWS_TO_ROOT)
	// "peer4x2", "peer4x3", "peer4x4" (+ "-root"): ... and B = 2, 3, 4 frames per launch (mv_config.frames_per_launch): 4 x B frames in flight
	uint32_t perLaunch = 0;
	bool rootB = false;
	if (argc > 6 && std::string(argv[6]).rfind("peer4x", 0) == 0) { perLaunch = (uint32_t)std::atoi(argv[6] + 6); rootB = std::string(argv[6]).find("-root") != std::string::npos; }
	const bool four = root || perLaunch || (argc > 6 && std::string(argv[6]) == "peer4");	// four whole frames in flight on the four hardware queues: MV_FLAG_CULL_ON_VIEW_STREAM (no stream of its own for upload + cull), frames_in_flight = 4
	const bool withComposite = !std::getenv("RANK_COST_NO_COMPOSITE");	// (experiment: a frame without its composite + row gather)
	const int frames = argc > 1 ? std::atoi(argv[1]) : 400;
	const unsigned worlds = argc > 2 ? (unsigned)std::atoi(argv[2]) : 15u;
	const int onlyRank = argc > 3 ? std::atoi(argv[3]) : -1;	// (one rank only: for a kernel trace)
	const uint32_t inFlight = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 0u;	// mv_config.frames_in_flight (0 = 3)
	std::vector<std::vector<uint16_t>> vols(V, std::vector<uint16_t>((size_t)N * N * N * 4));
	for (uint32_t k = 0; k < V; ++k) mvh_synth_volume(N, k, vols[k].data());

	const float es = c5 ? 2.0f : 1.0f;	// the reference's camera, pulled back for the 8 x 8 grid (scene.py camera())
	const float eye[3] = { 4.0f * es, 16.0f * es, -80.0f * es }, focus[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 }, light[3] = { 75, 75, -75 };
	float view[16], proj[16], viewProj[16], shadowVP[16];
	mvh_look_at_lh(eye, focus, up, view);
	mvh_perspective_fov_lh(3.14159265f / 4, (float)W / H, 1.0f, 1000.0f, proj);
	mvh_multiply(view, proj, viewProj);
	mvh_shadow_view_proj(light, 10.03f * 1.8f, shadowVP);
	std::vector<float> depth((size_t)W * H);
	std::vector<uint16_t> shadow((size_t)M * M);
	mvh_synth_sphere_depth(W, H, viewProj, focus, 8.0f, depth.data());
	mvh_synth_sphere_shadow(M, shadowVP, focus, 8.0f, shadow.data());
	const float lc[3] = { 1.0f, 0.7f, 0.3f }, ac[3] = { 0.4f, 0.6f, 1.0f };

	double one = 0.0;
	for (uint32_t world : { 1u, 2u, 4u, 8u })
	{
		if (!(worlds & world)) continue;
		double worst = 0.0;
		std::printf("world %u: ms/frame per rank [", world);
		for (uint32_t rank = 0; rank < world; ++rank)
		{
			if (onlyRank >= 0 && (uint32_t)onlyRank != rank) continue;
			mvamd::MultiRayCaster rc;
			if (!rc.Init(N, L, V, V, W, H, M, 0, nullptr, rank, world, MV_FLAG_PIPELINE_FRAMES | (four ? MV_FLAG_CULL_ON_VIEW_STREAM : 0u), four ? 4u : inFlight, perLaunch)) { std::fprintf(stderr, "Init failed (no gfx950 device?)\n"); return 2; }
			for (uint32_t k = 0; k < V; ++k) if (!rc.LoadVolumeData(k, vols[k].data())) return 3;
			if (!rc.SetRenderTargetsHost(depth.data(), shadow.data())) return 4;
			rc.SetMaxSamples(c5 ? 256 : c2 ? 64 : 128, 96);
			rc.SetLight(light, lc, 3.0f * 3.14159265f);
			rc.SetAmbient(ac, 2.0f * 3.14159265f);
			if (peer)
			{
				if (mv_dist_init_peer(rc.Context(), nullptr, MV_DIST_PEER_LOOPBACK | (noFold ? MV_DIST_PEER_NO_FOLD : 0u) | ((root || rootB) ? MV_DIST_PEER_ROWS_TO_ROOT : 0u) | (std::getenv("RANK_COST_DEAL_VOLUMES") ? MV_DIST_PEER_DEAL_VOLUMES : 0u) | (std::getenv("RANK_COST_DEAL_TILES") ? MV_DIST_PEER_DEAL_TILES : 0u) | (std::getenv("RANK_COST_NO_ACQUIRE") ? MV_DIST_PEER_NO_ACQUIRE : 0u) | (std::getenv("RANK_COST_PUSH_CELLS") ? MV_DIST_PEER_PUSH_CELLS : 0u)) != MV_OK || mv_dist_selftest(rc.Context()) != MV_OK) { std::fprintf(stderr, "%s\n", rc.LastError()); return 5; }
			}
			else
			{
				if (mv_dist_init_callbacks(rc.Context(), noGather, noReduce, nullptr) != MV_OK) { std::fprintf(stderr, "%s\n", rc.LastError()); return 5; }
				mv_dist_set_inplace_gather(rc.Context(), 1);	// like RCCL: no staging copies
				mv_dist_set_cull_crosscheck(rc.Context(), 0);	// the collectives here deliver nothing: the other ranks' culls are not there to compare
			}
			for (int f = 0; f < std::min(200, frames); ++f) if (!rc.DistFrame(viewProj, shadowVP, eye, withComposite)) { std::fprintf(stderr, "%s\n", rc.LastError()); return 6; }
			rc.DistFlush(); mv_synchronize(rc.Context());
			double waited = 0.0;
			mv_host_wait_ms(rc.Context(), &waited, 1);
			const auto t0 = std::chrono::steady_clock::now();
			for (int f = 0; f < frames; ++f) if (!rc.DistFrame(viewProj, shadowVP, eye, withComposite)) { std::fprintf(stderr, "frame %d: %s\n", f, rc.LastError()); return 6; }
			mv_host_wait_ms(rc.Context(), &waited, 1);
			const double host = (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - waited) / frames;	// enqueueing, without the waits for a free frame slot
			if (!rc.DistFlush() || mv_synchronize(rc.Context()) != MV_OK) { std::fprintf(stderr, "flush: %s\n", rc.LastError()); return 7; }
			const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
			std::printf("%s%.4f (host %.4f)", rank ? ", " : "", ms, host);
			std::fflush(stdout);
			worst = std::max(worst, ms);
		}
		if (world == 1) one = worst;
		std::printf("]  -> max %.4f  ceiling %.2fx\n", worst, one / worst);
	}
	return 0;
}
