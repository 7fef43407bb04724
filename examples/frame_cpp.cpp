// frame_cpp.cpp -- C++ host driving the path through the header-only MultiRayCaster mirror (the reference's host
// language), with the reference's default scene: 4 volumes, eye (4,16,-80), FOV pi/4, light (75,75,-75).
// Build: g++ -O2 -std=c++17 examples/frame_cpp.cpp -Lmultivolumes_amd -lmultivolumes_amd -Wl,-rpath,$PWD/multivolumes_amd -o /tmp/frame_cpp
// Prints "OK <view samples> <rays> <checksum of the composite>" or exits non-zero.
#include "../multivolumes_amd/csrc/MultiRayCaster.hpp"

#include <cmath>
#include <cstdio>
#include <vector>

int main()
{
	const uint32_t N = 32, L = 24, V = 4, W = 256, H = 256, M = 128;
	mvamd::MultiRayCaster rc;
	if (!rc.Init(N, L, V, V, W, H, M)) { std::fprintf(stderr, "Init failed (no gfx950 device?)\n"); return 2; }

	std::vector<uint16_t> vol((size_t)N * N * N * 4);
	for (uint32_t k = 0; k < V; ++k) { mvh_synth_volume(N, k, vol.data()); if (!rc.LoadVolumeData(k, vol.data())) return 3; }

	const float eye[3] = { 4.0f, 16.0f, -80.0f }, focus[3] = { 0, 0, 0 }, up[3] = { 0, 1, 0 }, light[3] = { 75, 75, -75 };
	float view[16], proj[16], viewProj[16], shadowVP[16];
	mvh_look_at_lh(eye, focus, up, view);
	mvh_perspective_fov_lh(3.14159265f / 4, (float)W / H, 1.0f, 1000.0f, proj);
	mvh_multiply(view, proj, viewProj);
	mvh_shadow_view_proj(light, 10.03f * 1.8f, shadowVP);

	std::vector<float> depth((size_t)W * H);
	std::vector<uint16_t> shadow((size_t)M * M);
	mvh_synth_sphere_depth(W, H, viewProj, focus, 8.0f, depth.data());
	mvh_synth_sphere_shadow(M, shadowVP, focus, 8.0f, shadow.data());
	if (!rc.SetRenderTargetsHost(depth.data(), shadow.data())) return 4;
	rc.SetMaxSamples(48, 24);
	const float lc[3] = { 1.0f, 0.7f, 0.3f }, ac[3] = { 0.4f, 0.6f, 1.0f };
	rc.SetLight(light, lc, 3.0f * 3.14159265f);
	rc.SetAmbient(ac, 2.0f * 3.14159265f);

	for (uint32_t f = 0; f < V; ++f)
		if (!rc.UpdateFrame(viewProj, shadowVP, eye) || !rc.Render(true)) { std::fprintf(stderr, "frame failed: %s\n", rc.LastError()); return 5; }

	std::vector<uint16_t> img((size_t)W * H * 4);
	if (mv_readback_composite(rc.Context(), img.data()) != MV_OK) return 6;
	mv_stats st;
	if (mv_get_stats(rc.Context(), &st) != MV_OK) return 7;
	unsigned long long sum = 0;
	for (uint16_t v : img) sum = sum * 1315423911ull + v;
	std::printf("OK %llu %llu %llx\n", (unsigned long long)st.view_samples, (unsigned long long)st.rays, sum);
	return st.view_samples > 0 ? 0 : 8;
}
