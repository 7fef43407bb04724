// mv_host.cpp -- CPU-side helpers of include/multivolumes_host.h (no HIP): the arithmetic the reference's C++ does
// around the hot path (DirectXMath matrices, SetVolumesWorld, UpdateFrame, shadow view-projection) and the
// deterministic synthetic inputs used by bench.py and the tests.  Matrices are computed in double and rounded once.
#include "../../include/multivolumes_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <vector>
#include <thread>
#include <algorithm>

namespace
{
struct M4 { double m[4][4]; };

M4 load(const float* f) { M4 r; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r.m[i][j] = f[4 * i + j]; return r; }
void store(const M4& a, float* f) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) f[4 * i + j] = (float)a.m[i][j]; }
// XMStoreFloat4x4(XMMatrixTranspose(M)) and XMStoreFloat3x4(M) both put M[r][c] at f[4*c + r]
void storeT(const M4& a, float* f, int cols) { for (int c = 0; c < cols; ++c) for (int r = 0; r < 4; ++r) f[4 * c + r] = (float)a.m[r][c]; }

M4 mul(const M4& a, const M4& b)
{
	M4 r;
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j)
		{
			double s = 0.0;
			for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
			r.m[i][j] = s;
		}
	return r;
}

bool inverse(const M4& a, M4& out)
{
	// Gauss-Jordan with partial pivoting
	double w[4][8];
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j) { w[i][j] = a.m[i][j]; w[i][4 + j] = i == j ? 1.0 : 0.0; }
	for (int c = 0; c < 4; ++c)
	{
		int p = c;
		for (int r = c + 1; r < 4; ++r) if (fabs(w[r][c]) > fabs(w[p][c])) p = r;
		if (fabs(w[p][c]) < 1e-300) return false;
		if (p != c) for (int j = 0; j < 8; ++j) std::swap(w[p][j], w[c][j]);
		const double inv = 1.0 / w[c][c];
		for (int j = 0; j < 8; ++j) w[c][j] *= inv;
		for (int r = 0; r < 4; ++r)
			if (r != c)
			{
				const double f = w[r][c];
				if (f != 0.0) for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
			}
	}
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out.m[i][j] = w[i][4 + j];
	return true;
}

M4 worldFrom3x4(const float* f)	// XMLoadFloat3x4: f[4*c + r] = M[r][c], last column (0,0,0,1)
{
	M4 r;
	for (int i = 0; i < 4; ++i)
	{
		for (int j = 0; j < 3; ++j) r.m[i][j] = f[4 * j + i];
		r.m[i][3] = i == 3 ? 1.0 : 0.0;
	}
	return r;
}

uint16_t f2h(float f)	// fp32 -> fp16, round to nearest even
{
	uint32_t u; memcpy(&u, &f, 4);
	const uint32_t s = (u >> 16) & 0x8000u;
	u &= 0x7fffffffu;
	if (u >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
	if (u >= 0x477ff000u) return (uint16_t)(s | 0x7c00u);	// rounds to >= 65520 -> inf
	if (u < 0x33000001u) return (uint16_t)s;				// < 2^-25 (or exactly 2^-25, ties to even 0)
	const int e = (int)(u >> 23) - 127;
	uint32_t m = (u & 0x7fffffu) | 0x800000u;
	int shift = e >= -14 ? 13 : 13 + (-14 - e);
	uint32_t v = m >> shift;
	const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
	if (rem > half || (rem == half && (v & 1u))) ++v;
	if (e >= -14) v += (uint32_t)(e + 14) << 10;	// v already carries the implicit 1 at bit 10
	return (uint16_t)(s | v);
}

template <typename F>
void parallelFor(uint32_t n, F fn)
{
	const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
	const uint32_t nt = std::min(hw, std::max(1u, n));
	std::vector<std::thread> th;
	for (uint32_t t = 0; t < nt; ++t)
		th.emplace_back([=]() { for (uint32_t i = t; i < n; i += nt) fn(i); });
	for (auto& x : th) x.join();
}

inline uint32_t lcgStep(uint32_t prev) { return 1664525u * prev + 1013904223u; }	// Shaders/Random.hlsli:34-41

inline float latticeValue(uint32_t seed, int x, int y, int z, int o)
{
	uint32_t h = seed;
	h = lcgStep(h ^ (uint32_t)x);
	h = lcgStep(h ^ (uint32_t)y);
	h = lcgStep(h ^ (uint32_t)z);
	h = lcgStep(h ^ (uint32_t)o);
	h ^= h >> 16;
	h = lcgStep(h);
	return (float)((h >> 8) & 0x00ffffffu) / (float)0x01000000;
}

inline float smooth(float t) { return t * t * (3.0f - 2.0f * t); }

float valueNoise(uint32_t seed, float px, float py, float pz)	// p in [0,1)^3 -> [0,1]
{
	float sum = 0.0f, amp = 0.5f, norm = 0.0f;
	int freq = 4;
	for (int o = 0; o < 3; ++o)
	{
		const float x = px * freq, y = py * freq, z = pz * freq;
		const int ix = (int)floorf(x), iy = (int)floorf(y), iz = (int)floorf(z);
		const float fx = smooth(x - ix), fy = smooth(y - iy), fz = smooth(z - iz);
		float c[2][2][2];
		for (int k = 0; k < 2; ++k) for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) c[k][j][i] = latticeValue(seed, ix + i, iy + j, iz + k, o);
		const float c00 = c[0][0][0] + fx * (c[0][0][1] - c[0][0][0]), c10 = c[0][1][0] + fx * (c[0][1][1] - c[0][1][0]);
		const float c01 = c[1][0][0] + fx * (c[1][0][1] - c[1][0][0]), c11 = c[1][1][0] + fx * (c[1][1][1] - c[1][1][0]);
		const float c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
		sum += amp * (c0 + fz * (c1 - c0));
		norm += amp;
		amp *= 0.5f;
		freq *= 2;
	}
	return sum / norm;
}

struct D3 { double x, y, z; };

// project p through row-vector matrix, returning clip (x/w, y/w, z/w)
D3 project(const M4& m, D3 p)
{
	const double x = p.x * m.m[0][0] + p.y * m.m[1][0] + p.z * m.m[2][0] + m.m[3][0];
	const double y = p.x * m.m[0][1] + p.y * m.m[1][1] + p.z * m.m[2][1] + m.m[3][1];
	const double z = p.x * m.m[0][2] + p.y * m.m[1][2] + p.z * m.m[2][2] + m.m[3][2];
	const double w = p.x * m.m[0][3] + p.y * m.m[1][3] + p.z * m.m[2][3] + m.m[3][3];
	return D3{ x / w, y / w, z / w };
}

// nearest intersection parameter of segment a + t (b - a), t in [0,1], with a sphere; < 0 if none
double hitSphere(D3 a, D3 b, D3 c, double r)
{
	const D3 d{ b.x - a.x, b.y - a.y, b.z - a.z };
	const D3 o{ a.x - c.x, a.y - c.y, a.z - c.z };
	const double A = d.x * d.x + d.y * d.y + d.z * d.z;
	const double B = 2.0 * (o.x * d.x + o.y * d.y + o.z * d.z);
	const double C = o.x * o.x + o.y * o.y + o.z * o.z - r * r;
	const double disc = B * B - 4.0 * A * C;
	if (disc < 0.0 || A == 0.0) return -1.0;
	const double sq = sqrt(disc);
	double t = (-B - sq) / (2.0 * A);
	if (t < 0.0) t = (-B + sq) / (2.0 * A);
	return (t >= 0.0 && t <= 1.0) ? t : -1.0;
}
} // namespace

extern "C" {

void mvh_look_at_lh(const float eye[3], const float focus[3], const float up[3], float out[16])
{
	double z[3] = { (double)focus[0] - eye[0], (double)focus[1] - eye[1], (double)focus[2] - eye[2] };
	double l = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
	for (double& v : z) v /= l;
	double x[3] = { up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0] };
	l = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
	for (double& v : x) v /= l;
	const double y[3] = { z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0] };
	M4 m;
	for (int i = 0; i < 3; ++i) { m.m[i][0] = x[i]; m.m[i][1] = y[i]; m.m[i][2] = z[i]; m.m[i][3] = 0.0; }
	m.m[3][0] = -(x[0] * eye[0] + x[1] * eye[1] + x[2] * eye[2]);
	m.m[3][1] = -(y[0] * eye[0] + y[1] * eye[1] + y[2] * eye[2]);
	m.m[3][2] = -(z[0] * eye[0] + z[1] * eye[1] + z[2] * eye[2]);
	m.m[3][3] = 1.0;
	store(m, out);
}

void mvh_perspective_fov_lh(float fovY, float aspect, float zn, float zf, float out[16])
{
	const double h = 1.0 / tan(0.5 * (double)fovY), w = h / (double)aspect, q = (double)zf / ((double)zf - zn);
	M4 m; memset(&m, 0, sizeof(m));
	m.m[0][0] = w; m.m[1][1] = h; m.m[2][2] = q; m.m[2][3] = 1.0; m.m[3][2] = -q * zn;
	store(m, out);
}

void mvh_orthographic_lh(float w, float h, float zn, float zf, float out[16])
{
	const double q = 1.0 / ((double)zf - zn);
	M4 m; memset(&m, 0, sizeof(m));
	m.m[0][0] = 2.0 / w; m.m[1][1] = 2.0 / h; m.m[2][2] = q; m.m[3][2] = -q * zn; m.m[3][3] = 1.0;
	store(m, out);
}

void mvh_multiply(const float a[16], const float b[16], float out[16]) { store(mul(load(a), load(b)), out); }

int mvh_inverse(const float m[16], float out[16])
{
	M4 r;
	if (!inverse(load(m), r)) return -1;
	store(r, out);
	return 0;
}

// MultiRayCaster.cpp:297-303
void mvh_set_volume_world(float size, const float pos[3], float world[12])
{
	size *= 0.5f;
	M4 m; memset(&m, 0, sizeof(m));
	m.m[0][0] = m.m[1][1] = m.m[2][2] = size;
	m.m[3][0] = pos[0]; m.m[3][1] = pos[1]; m.m[3][2] = pos[2]; m.m[3][3] = 1.0;
	storeT(m, world, 3);
}

// MultiRayCaster.cpp:277-295
void mvh_set_volumes_world(uint32_t V, float size, const float center[3], float* worlds)
{
	memset(worlds, 0, (size_t)V * 12 * sizeof(float));
	const uint32_t rowLength = (uint32_t)ceilf(sqrtf((float)V));
	const uint32_t colLength = (uint32_t)ceilf((float)(V / rowLength));	// integer division, as in the reference
	float pos[3] = { center[0], center[1], center[2] };
	pos[2] -= (colLength / 2.0f - 0.5f) * size * 1.5f;
	for (uint32_t m = 0; m < colLength; ++m)
	{
		pos[0] = center[0] - (rowLength / 2.0f - 0.5f) * size * 1.5f;
		for (uint32_t n = 0; n < rowLength; ++n)
		{
			const uint32_t i = rowLength * m + n;
			if (i < V) mvh_set_volume_world(size, pos, worlds + 12 * (size_t)i);
			pos[0] += size * 1.5f;
		}
		pos[2] += size * 1.5f;
	}
}

// ObjectRenderer.cpp:176-185
void mvh_shadow_view_proj(const float lightPos[3], float sceneSize, float outT[16])
{
	const float zero[3] = { 0.0f, 0.0f, 0.0f }, up[3] = { 0.0f, 1.0f, 0.0f };
	float view[16], proj[16];
	mvh_look_at_lh(lightPos, zero, up, view);
	const float size = sceneSize * 1.5f;
	mvh_orthographic_lh(size, size, 1.0f, 200.0f, proj);
	// full-precision product, then the transposed store of XMStoreFloat4x4(&m_shadowVP, XMMatrixTranspose(lightViewProj))
	storeT(mul(load(view), load(proj)), outT, 4);
}

// MultiRayCaster.cpp:316-353
void mvh_update_frame(uint32_t V, const float* worlds, const float viewProj[16], const float shadowVPT[16],
	const float eye[3], float viewportW, float viewportH, const float lightPos[3], const float lightColor[4],
	const float ambient[4], uint32_t frameIdx, void* cbPerFrame, void* perObject)
{
	const M4 vp = load(viewProj);
	{
		float cb[53];
		memset(cb, 0, sizeof(cb));
		cb[0] = eye[0]; cb[1] = eye[1]; cb[2] = eye[2]; cb[3] = 1.0f;
		cb[4] = viewportW; cb[5] = viewportH; cb[6] = 0.0f; cb[7] = 0.0f;
		M4 projToWorld;
		if (inverse(vp, projToWorld)) storeT(projToWorld, cb + 8, 4);
		memcpy(cb + 24, shadowVPT, 16 * sizeof(float));
		cb[40] = lightPos[0]; cb[41] = lightPos[1]; cb[42] = lightPos[2]; cb[43] = 1.0f;
		memcpy(cb + 44, lightColor, 4 * sizeof(float));
		memcpy(cb + 48, ambient, 4 * sizeof(float));
		memcpy(cb + 52, &frameIdx, 4);
		memcpy(cbPerFrame, cb, 212);
	}
	// The inverse world matrices only change when a volume is moved (SetVolumesWorld / SetVolumeWorld): the last set is kept per calling
	// thread and reused while the world matrices compare equal (the same arithmetic either way -- a rank of the multi-GPU loop enqueues a
	// whole frame in ~50 us of host time, of which sixteen 4 x 4 inversions were a tenth).
	thread_local std::vector<float> cachedWorlds;
	thread_local std::vector<M4> cachedWorldI;
	const size_t nw = (size_t)V * 12;
	if (cachedWorlds.size() != nw || memcmp(cachedWorlds.data(), worlds, nw * sizeof(float)) != 0)
	{
		cachedWorlds.assign(worlds, worlds + nw);
		cachedWorldI.resize(V);
		for (uint32_t i = 0; i < V; ++i)
		{
			memset(&cachedWorldI[i], 0, sizeof(M4));
			inverse(worldFrom3x4(worlds + 12 * (size_t)i), cachedWorldI[i]);	// singular (unplaced) volumes keep zeros
		}
	}
	for (uint32_t i = 0; i < V; ++i)
	{
		float* po = reinterpret_cast<float*>(static_cast<uint8_t*>(perObject) + (size_t)i * 224);
		const M4 world = worldFrom3x4(worlds + 12 * (size_t)i);
		const M4 wvp = mul(world, vp);
		M4 wvpI;
		memset(&wvpI, 0, sizeof(wvpI));
		inverse(wvp, wvpI);
		storeT(wvp, po, 4);
		storeT(wvpI, po + 16, 4);
		storeT(cachedWorldI[i], po + 32, 3);
		storeT(world, po + 44, 3);
	}
}

void mvh_synth_volume(uint32_t N, uint32_t k, uint16_t* rgba)
{
	uint32_t seed = 0x2545F491u;
	for (uint32_t i = 0; i < k; ++i) seed = lcgStep(seed);
	const float gridSize = (float)N;
	parallelFor(N, [=](uint32_t z)
	{
		for (uint32_t y = 0; y < N; ++y)
			for (uint32_t x = 0; x < N; ++x)
			{
				// CSInitGridData.hlsl:16-24
				const float px = ((float)x + 0.5f) / gridSize * 2.0f - 1.0f, py = ((float)y + 0.5f) / gridSize * 2.0f - 1.0f, pz = ((float)z + 0.5f) / gridSize * 2.0f - 1.0f;
				const float r_sq = px * px + py * py + pz * pz;
				float a = 1.0f - r_sq;
				a *= a;
				a = fminf(fmaxf(a * a * 2.0f, 0.0f), 1.0f);
				const float s = fminf(fmaxf(py * 0.5f + 0.2f, 0.0f), 1.0f);
				const float r = fmaf(s, 1.0f - 0.5f, 0.5f), g = fmaf(s, 0.6f - 0.8f, 0.8f), b = fmaf(s, 0.0f - 1.0f, 1.0f);
				if (k > 0)
				{
					const float n = valueNoise(seed, ((float)x + 0.5f) / gridSize, ((float)y + 0.5f) / gridSize, ((float)z + 0.5f) / gridSize);
					a *= 0.25f + 0.75f * n;
				}
				uint16_t* o = rgba + 4 * (((size_t)z * N + y) * N + x);
				o[0] = f2h(r); o[1] = f2h(g); o[2] = f2h(b); o[3] = f2h(a);
			}
	});
}

void mvh_synth_sphere_depth(uint32_t W, uint32_t H, const float viewProj[16], const float center[3], float radius, float* depth)
{
	const M4 vp = load(viewProj);
	M4 vpI;
	if (!inverse(vp, vpI)) { for (size_t i = 0; i < (size_t)W * H; ++i) depth[i] = 1.0f; return; }
	const D3 c{ center[0], center[1], center[2] };
	parallelFor(H, [=](uint32_t y)
	{
		for (uint32_t x = 0; x < W; ++x)
		{
			const double nx = ((double)x + 0.5) / W * 2.0 - 1.0, ny = 1.0 - ((double)y + 0.5) / H * 2.0;
			const D3 a = project(vpI, D3{ nx, ny, 0.0 }), b = project(vpI, D3{ nx, ny, 1.0 });
			const double t = hitSphere(a, b, c, radius);
			float d = 1.0f;
			if (t >= 0.0)
			{
				const D3 p{ a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z) };
				d = (float)project(vp, p).z;
			}
			depth[(size_t)y * W + x] = d;
		}
	});
}

void mvh_synth_sphere_shadow(uint32_t M, const float lightViewProjT[16], const float center[3], float radius, uint16_t* d16)
{
	M4 vp;
	for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) vp.m[r][c] = lightViewProjT[4 * c + r];
	M4 vpI;
	if (!inverse(vp, vpI)) { for (size_t i = 0; i < (size_t)M * M; ++i) d16[i] = 65535; return; }
	const D3 c{ center[0], center[1], center[2] };
	parallelFor(M, [=](uint32_t y)
	{
		for (uint32_t x = 0; x < M; ++x)
		{
			const double nx = ((double)x + 0.5) / M * 2.0 - 1.0, ny = 1.0 - ((double)y + 0.5) / M * 2.0;
			const D3 a = project(vpI, D3{ nx, ny, 0.0 }), b = project(vpI, D3{ nx, ny, 1.0 });
			const double t = hitSphere(a, b, c, radius);
			uint16_t d = 65535;
			if (t >= 0.0)
			{
				const D3 p{ a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z) };
				const double z = std::min(std::max(project(vp, p).z, 0.0), 1.0);
				d = (uint16_t)lrint(z * 65535.0);
			}
			d16[(size_t)y * M + x] = d;
		}
	});
}

int mvh_load_obj(const char* path, float** positions, uint32_t* numVerts, uint32_t** indices, uint32_t* numTris, float aabbMin[3], float aabbMax[3])
{
	if (!path || !positions || !numVerts || !indices || !numTris) return -1;
	FILE* f = fopen(path, "r");
	if (!f) return -1;
	std::vector<float> v;
	std::vector<uint32_t> idx;
	char line[1024];
	while (fgets(line, sizeof(line), f))
	{
		if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t'))
		{
			float x, y, z;
			if (sscanf(line + 1, "%f %f %f", &x, &y, &z) == 3) { v.push_back(x); v.push_back(y); v.push_back(z); }
		}
		else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t'))
		{
			std::vector<uint32_t> poly;
			const char* p = line + 1;
			for (;;)
			{
				while (*p == ' ' || *p == '\t') ++p;
				if (*p == 0 || *p == '\n' || *p == '\r') break;
				char* end = nullptr;
				const long a = strtol(p, &end, 10);
				if (end == p) break;
				const long nv = (long)(v.size() / 3);
				const long i = a > 0 ? a - 1 : nv + a;	// 1-based, or relative to the vertices read so far
				if (i < 0 || i >= nv) { fclose(f); return -1; }
				poly.push_back((uint32_t)i);
				p = end;
				while (*p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r') ++p;	// skip /vt/vn
			}
			for (size_t k = 2; k < poly.size(); ++k) { idx.push_back(poly[0]); idx.push_back(poly[k - 1]); idx.push_back(poly[k]); }
		}
	}
	fclose(f);
	if (v.empty() || idx.empty()) return -1;
	*numVerts = (uint32_t)(v.size() / 3);
	*numTris = (uint32_t)(idx.size() / 3);
	*positions = static_cast<float*>(malloc(v.size() * sizeof(float)));
	*indices = static_cast<uint32_t*>(malloc(idx.size() * sizeof(uint32_t)));
	if (!*positions || !*indices) { free(*positions); free(*indices); return -1; }
	memcpy(*positions, v.data(), v.size() * sizeof(float));
	memcpy(*indices, idx.data(), idx.size() * sizeof(uint32_t));
	if (aabbMin && aabbMax)
		for (int c = 0; c < 3; ++c)
		{
			float lo = v[c], hi = v[c];
			for (size_t i = c; i < v.size(); i += 3) { lo = fminf(lo, v[i]); hi = fmaxf(hi, v[i]); }
			aabbMin[c] = lo; aabbMax[c] = hi;
		}
	return 0;
}

void mvh_free(void* p) { free(p); }

void mvh_mesh_world(float scale, const float pos[3], float world[16])
{
	M4 m; memset(&m, 0, sizeof(m));
	m.m[0][0] = m.m[1][1] = m.m[2][2] = scale;
	m.m[3][0] = pos[0]; m.m[3][1] = pos[1]; m.m[3][2] = pos[2]; m.m[3][3] = 1.0;
	store(m, world);
}

void mvh_transpose(const float m[16], float outT[16]) { storeT(load(m), outT, 4); }

} // extern "C"

// ---------------------------------------------------------------------------------------------
// Minimal DDS volume reader (the front end of LoadVolumeData, MultiRayCaster.cpp:168-180).  Layout per the public DDS
// specification: magic "DDS ", DDS_HEADER (124 bytes), optional DDS_HEADER_DXT10 (20 bytes) when the four-CC is "DX10".
// ---------------------------------------------------------------------------------------------
namespace
{
float halfToFloat(uint16_t h)
{
	const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
	uint32_t bits;
	if (e == 0)
	{
		if (m == 0) bits = s;
		else
		{
			int sh = 0; uint32_t mm = m;
			while (!(mm & 0x400u)) { mm <<= 1; ++sh; }
			bits = s | ((uint32_t)(127 - 15 + 1 - sh) << 23) | ((mm & 0x3ffu) << 13);
		}
	}
	else if (e == 31) bits = s | 0x7f800000u | (m << 13);
	else bits = s | ((e + 127 - 15) << 23) | (m << 13);
	float f; memcpy(&f, &bits, 4);
	return f;
}

enum class DdsChannel { F32, F16, U8, U16 };
}

int mvh_load_dds_volume(const char* path, float** texels, uint32_t* width, uint32_t* height, uint32_t* depth)
{
	if (!path || !texels || !width || !height || !depth) return -1;
	FILE* f = fopen(path, "rb");
	if (!f) return -1;
	uint32_t hdr[32];	// magic + DDS_HEADER
	if (fread(hdr, 4, 32, f) != 32 || hdr[0] != 0x20534444u /* "DDS " */ || hdr[1] != 124u) { fclose(f); return -1; }
	const uint32_t flags = hdr[2], h = hdr[3], w = hdr[4], d = hdr[6];
	const uint32_t pfFlags = hdr[20], fourCC = hdr[21], rgbBits = hdr[22], rMask = hdr[23], caps2 = hdr[28];
	bool isVolume = (flags & 0x800000u /* DDSD_DEPTH */) || (caps2 & 0x200000u /* DDSCAPS2_VOLUME */);
	DdsChannel ch = DdsChannel::F32;
	uint32_t stride = 0;	// bytes per texel; channel 0 is at offset 0
	auto dxgi = [&](uint32_t fmt) -> bool
	{
		switch (fmt)
		{
		case 41: ch = DdsChannel::F32; stride = 4; return true;		// R32_FLOAT
		case 54: ch = DdsChannel::F16; stride = 2; return true;		// R16_FLOAT
		case 61: ch = DdsChannel::U8; stride = 1; return true;		// R8_UNORM
		case 56: ch = DdsChannel::U16; stride = 2; return true;		// R16_UNORM
		case 2: ch = DdsChannel::F32; stride = 16; return true;		// R32G32B32A32_FLOAT
		case 10: ch = DdsChannel::F16; stride = 8; return true;		// R16G16B16A16_FLOAT
		case 28: ch = DdsChannel::U8; stride = 4; return true;		// R8G8B8A8_UNORM
		default: return false;
		}
	};
	bool ok = false;
	if ((pfFlags & 0x4u /* DDPF_FOURCC */) && fourCC == 0x30315844u /* "DX10" */)
	{
		uint32_t dx10[5];
		if (fread(dx10, 4, 5, f) != 5) { fclose(f); return -4; }
		isVolume = dx10[1] == 4u /* D3D10_RESOURCE_DIMENSION_TEXTURE3D */;
		ok = dxgi(dx10[0]);
	}
	else if (pfFlags & 0x4u)
	{
		switch (fourCC)	// legacy D3DFMT codes stored as four-CC
		{
		case 114: ok = dxgi(41); break;	// D3DFMT_R32F
		case 111: ok = dxgi(54); break;	// D3DFMT_R16F
		case 116: ok = dxgi(2); break;	// D3DFMT_A32B32G32R32F
		case 113: ok = dxgi(10); break;	// D3DFMT_A16B16G16R16F
		default: ok = false;
		}
	}
	else if ((pfFlags & 0x20000u /* DDPF_LUMINANCE */) && rgbBits == 8) ok = dxgi(61);
	else if ((pfFlags & 0x20000u) && rgbBits == 16) ok = dxgi(56);
	else if ((pfFlags & 0x40u /* DDPF_RGB */) && rgbBits == 32 && rMask == 0xffu) ok = dxgi(28);	// A8B8G8R8: red is byte 0
	if (!isVolume || w == 0 || h == 0 || d == 0) { fclose(f); return -2; }
	if (!ok) { fclose(f); return -3; }
	// The header is untrusted input: bound every extent (D3D12 caps 3D textures at 2048 per axis), so that w*h*d*stride
	// cannot wrap, and check the payload against what the file really holds before allocating anything.
	if (w > 2048u || h > 2048u || d > 2048u) { fclose(f); return -2; }
	const size_t n = (size_t)w * h * d;	// <= 2^33
	const size_t rawBytes = n * stride;	// <= 2^37
	{
		const long here = ftell(f);
		if (here < 0 || fseek(f, 0, SEEK_END) != 0) { fclose(f); return -4; }
		const long end = ftell(f);
		if (end < 0 || (size_t)(end - here) < rawBytes || fseek(f, here, SEEK_SET) != 0) { fclose(f); return -4; }	// truncated file
	}
	uint8_t* rawBuf = static_cast<uint8_t*>(malloc(rawBytes ? rawBytes : 1));	// no exception may cross the extern "C" boundary
	float* out = static_cast<float*>(malloc(n * sizeof(float)));
	if (!rawBuf || !out) { free(rawBuf); free(out); fclose(f); return -1; }
	if (fread(rawBuf, 1, rawBytes, f) != rawBytes) { free(rawBuf); free(out); fclose(f); return -4; }
	fclose(f);
	struct RawHolder { uint8_t* p; ~RawHolder() { free(p); } uint8_t* data() const { return p; } } raw = { rawBuf };
	for (size_t i = 0; i < n; ++i)
	{
		const uint8_t* p = raw.data() + i * stride;
		switch (ch)
		{
		case DdsChannel::F32: memcpy(&out[i], p, 4); break;
		case DdsChannel::F16: { uint16_t v; memcpy(&v, p, 2); out[i] = halfToFloat(v); break; }
		case DdsChannel::U8: out[i] = (float)p[0] / 255.0f; break;
		case DdsChannel::U16: { uint16_t v; memcpy(&v, p, 2); out[i] = (float)v / 65535.0f; break; }
		}
	}
	*texels = out; *width = w; *height = h; *depth = d;
	return 0;
}
