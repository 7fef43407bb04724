/*
 * mv_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code; see mv_oracle.h).
 *
 * PARITY: the cull, the light march, the view march and the composite's shaders (a1, a2, a3, a7) are PINNED to the reference's shipped compiled shaders
 * (the .cso files under Bin, run on the CPU through oracle/dxil -> oracle/_ref/libmv_dxref.so; tests/test_oracle_vs_reference_dxil.py: cull identical
 * in 400 / 400 cases, view march bit-identical, light maps bit-identical without the probe), with the fixed-function sampler and the
 * format conversions as a stated contract the bytecode does not contain.  For the composite the rasteriser is a contract too: the
 * fragments its three pixel shaders run on come from this file's analytic stand-in (exitFragment), the shaders are the reference's.
 * Line-for-line scalar fp32 restatement of the reference HLSL:
 *   Sh/Common.hlsli, SharedConsts.h, Sh/RayMarch.hlsli, Sh/CSRayMarch.hlsl (+CSRayMarchV.hlsl),
 *   Sh/CSRayMarchL.hlsl, Sh/CSVolumeCull.hlsl, Sh/VolumeCull.hlsli,
 *   XUSG/Shaders/SHIrradiance{,Typeless}.hlsli, Sh/CSInitGridData.hlsl, Sh/CSR32FToRGBA16F.hlsl,
 *   Sh/VSCube.hlsl, Sh/PSDepthPeel.hlsl, Sh/PSCube.hlsl(i), Sh/RayCast.hlsli, Sh/PSResolveOIT.hlsl.
 *
 * Numeric conventions (documented in DESIGN.md, mirrored by the HIP kernels):
 *   - `min16float` is a minimum-precision hint (SM6.3 without -enable-16bit-types): fp32 everywhere,
 *     fp16 / 11-11-10 only in storage; min16float CONSTANTS are the half values the reference's compiler folded them to.
 *   - build with -ffp-contract=off; a multiply-add is fused ONLY where fmaf() is written out
 *     (the sampler's lerps, the composite's analytic exit point); operation order follows the reference's bytecode.
 *   - normalize(v) = v * (1 / sqrt(dot(v,v))); dot and mul() accumulate left to right.
 *   - HLSL min/max return the non-NaN operand (fminf/fmaxf); saturate(NaN) = 0.
 *   - float -> uint casts saturate (D3D ftou).
 *   - pow(x, 0.25) is evaluated as sqrt(sqrt(x)) (correctly rounded steps, identical on CPU and GPU).
 *   - sampler: texel centres at +0.5, clamp addressing, exact fp32 filter weights
 *     (D3D hardware may quantise weights to >= 8 fractional bits; unpinned).
 *   - fp32 -> fp16 and fp32 -> 11/11/10 stores round to nearest even; negative 11/10-bit values -> 0.
 *   - D16 comparison sampler: reference clamped to [0,1], compared in fp32 against texel/65535,
 *     LESS_EQUAL, bilinear blend of the four 0/1 results.
 */
#include "mv_oracle.h"

#include <math.h>
#include <string.h>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
// The shaders declare these as min16float: the reference's compiler (the shipped Bin/*.cso, read through oracle/dxil) has folded them
// to half precision, and every implementation -- whether it then computes at 16 or 32 bits -- starts from the half values.
const float kMaxDist = 3.46484375f;			// RayMarch.hlsli:17   half(2.0 * sqrt(3.0)) = 0x42EE
const float kAbsorption = 0.7998046875f;		// RayMarch.hlsli:11   half(0.8) = 0x3A66
const float kZeroThreshold = 0.010002136230468750f;	// RayMarch.hlsli:12   half(0.01) = 0x211F
const float kInvTwoPi = 0.1591796875f;			// CSRayMarch.hlsl:155  half(1 / (2 PI)) = 0x3118: the division compiled into a multiplication
const float kPI = 3.1415926535897f;			// XUSG/Shaders/SHIrradiance.hlsli:5-7
const float kFltMax = 3.402823466e+38f;		// Common.hlsli:9
const float kSqrt3 = 1.7320508075688772f;
const float kZNear = 1.0f, kZFar = 1000.0f;	// SharedConsts.h:9-10
const uint32_t kCubeMapRayMarchBit = 1u << 15;	// Common.hlsli:11

struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

inline float3 f3(float x, float y, float z) { float3 r = { x, y, z }; return r; }
inline float3 operator+(float3 a, float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(float3 a, float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator*(float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline float3 operator/(float3 a, float3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline float idx(const float3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
inline float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 normalize3(float3 v) { const float r = 1.0f / sqrtf(dot3(v, v)); return f3(v.x * r, v.y * r, v.z * r); }	// compiled as v * rsqrt(dot(v, v))
inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }	// NaN -> 0
inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
inline float lerpf(float a, float b, float w) { return fmaf(w, b - a, a); }
inline float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float3 rayAt(float3 o, float3 d, float t) { return f3(t * d.x + o.x, t * d.y + o.y, t * d.z + o.z); }	// compiled as a multiplication and an addition

inline uint32_t f2u(float x)	// D3D ftou: NaN/negative -> 0, saturating
{
	if (!(x > 0.0f)) return 0u;
	if (x >= 4294967296.0f) return 0xffffffffu;
	return (uint32_t)x;
}

inline int32_t f2i_floor(float x)	// floor to int with saturation (texel addressing)
{
	if (!(x == x)) return 0;
	const float f = floorf(x);
	if (f <= -2147483648.0f) return INT32_MIN;
	if (f >= 2147483520.0f) return INT32_MAX - 127;
	return (int32_t)f;
}

inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- HLSL constant-buffer / structured-buffer decoding (Common.hlsli:28-49; MultiRayCaster.cpp:23-41, 341-351)
// float4x4 stored transposed via XMStoreFloat4x4(XMMatrixTranspose(M)) and read column-major: M[r][c] = f[4*c + r].
// float4x3 stored via XMStoreFloat3x4(M) and read column-major:                                M[r][c] = f[4*c + r], c < 3.
struct PerObject
{
	float wvp[16], wvpI[16], worldI[12], world[12];
};

struct CBPerFrame
{
	float eyePos[4];
	float viewport[4];
	float screenToWorld[16];
	float shadowViewProj[16];
	float lightPos[4];
	float lightColor[4];
	float ambient[4];
	uint32_t frameIdx;
};

static_assert(sizeof(PerObject) == MVO_PER_OBJECT_BYTES, "PerObject layout");
static_assert(sizeof(CBPerFrame) == MVO_CB_PER_FRAME_BYTES, "CBPerFrame layout");

inline float4 mul44(float4 p, const float* m)	// mul(float4, float4x4)
{
	float4 r;
	r.x = p.x * m[0] + p.y * m[1] + p.z * m[2] + p.w * m[3];
	r.y = p.x * m[4] + p.y * m[5] + p.z * m[6] + p.w * m[7];
	r.z = p.x * m[8] + p.y * m[9] + p.z * m[10] + p.w * m[11];
	r.w = p.x * m[12] + p.y * m[13] + p.z * m[14] + p.w * m[15];
	return r;
}

inline float3 mul43(float3 p, const float* m)	// mul(float4(p, 1.0), float4x3)
{
	return f3(p.x * m[0] + p.y * m[1] + p.z * m[2] + m[3],
		p.x * m[4] + p.y * m[5] + p.z * m[6] + m[7],
		p.x * m[8] + p.y * m[9] + p.z * m[10] + m[11]);
}

inline float3 mul33(float3 p, const float* m)	// mul(float3, (float3x3)float4x3)
{
	return f3(p.x * m[0] + p.y * m[1] + p.z * m[2],
		p.x * m[4] + p.y * m[5] + p.z * m[6],
		p.x * m[8] + p.y * m[9] + p.z * m[10]);
}

// ---------------------------------------------------------------------------------------------
// formats
// ---------------------------------------------------------------------------------------------
inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float h2f(uint16_t h)
{
	const uint32_t s = (uint32_t)(h & 0x8000u) << 16;
	const uint32_t e = (h >> 10) & 0x1fu;
	uint32_t m = h & 0x3ffu;
	if (e == 0)
	{
		if (m == 0) return asfloat(s);
		// denormal: m * 2^-24
		const float v = (float)m * 5.9604644775390625e-08f;
		return asfloat(asuint(v) | s);
	}
	if (e == 31) return asfloat(s | 0x7f800000u | (m << 13));
	return asfloat(s | ((e + 112u) << 23) | (m << 13));
}

// Generic fp32 -> small float (EB exponent bits = 5, MB mantissa bits) RNE; returns magnitude bits (no sign).
template <int MB>
uint32_t f32_to_small(uint32_t absBits)
{
	const int SH = 23 - MB;
	const uint32_t INF = 31u << MB;
	if (absBits > 0x7f800000u) return INF | (1u << (MB - 1));	// NaN -> quiet NaN
	if (absBits == 0x7f800000u) return INF;
	const int32_t e = (int32_t)(absBits >> 23) - 127;
	uint32_t m = absBits & 0x7fffffu;
	if (e > 15)
	{
		// may still round to max finite? no: e >= 16 is already >= 65536 > max finite -> inf under RNE
		return INF;
	}
	if (e >= -14)
	{
		// normal
		uint32_t v = ((uint32_t)(e + 15) << MB) | (m >> SH);
		const uint32_t rem = m & ((1u << SH) - 1);
		const uint32_t half = 1u << (SH - 1);
		if (rem > half || (rem == half && (v & 1u))) ++v;	// carry may roll into exponent (and to INF), which is correct
		return v;
	}
	// denormal: value = 1.m * 2^e, target unit = 2^(-14 - MB)
	if (e < -14 - MB - 1) return 0;
	m |= 0x800000u;
	const int shift = SH + (-14 - e);	// >= SH + 1
	if (shift > 31) return 0;
	uint32_t v = m >> shift;
	const uint32_t rem = m & ((1u << shift) - 1);
	const uint32_t half = 1u << (shift - 1);
	if (rem > half || (rem == half && (v & 1u))) ++v;
	return v;
}

uint16_t f2h(float f)
{
	const uint32_t u = asuint(f);
	const uint32_t s = (u >> 16) & 0x8000u;
	return (uint16_t)(s | f32_to_small<10>(u & 0x7fffffffu));
}

inline float small_to_f32(uint32_t bits, int MB)	// 5-bit exponent, MB mantissa bits, unsigned
{
	return h2f((uint16_t)(bits << (10 - MB)));
}

uint32_t pack_r11g11b10f(float r, float g, float b)
{
	// negative (incl. -0) -> 0 ; NaN keeps NaN ; RNE  (DirectXMath XMStoreFloat3PK semantics for sign; RNE assumed)
	const uint32_t ur = asuint(r), ug = asuint(g), ub = asuint(b);
	const uint32_t R = ((ur & 0x80000000u) && (ur & 0x7fffffffu) <= 0x7f800000u) ? 0u : f32_to_small<6>(ur & 0x7fffffffu);
	const uint32_t G = ((ug & 0x80000000u) && (ug & 0x7fffffffu) <= 0x7f800000u) ? 0u : f32_to_small<6>(ug & 0x7fffffffu);
	const uint32_t B = ((ub & 0x80000000u) && (ub & 0x7fffffffu) <= 0x7f800000u) ? 0u : f32_to_small<5>(ub & 0x7fffffffu);
	return (R & 0x7ffu) | ((G & 0x7ffu) << 11) | ((B & 0x3ffu) << 22);
}

inline float3 unpack_r11g11b10f(uint32_t p)
{
	return f3(small_to_f32(p & 0x7ffu, 6), small_to_f32((p >> 11) & 0x7ffu, 6), small_to_f32(p >> 22, 5));
}

// ---------------------------------------------------------------------------------------------
// a5: fixed-function sampler contract (samplers at MultiRayCaster.cpp:556-566)
// ---------------------------------------------------------------------------------------------
struct Taps { int32_t i0, i1; float f; };

// Optional modes (mvo_set_options; both off by default and off in every parity run):
//  * g_weightBits = b > 0: the scaled texel coordinate is snapped to b fractional bits (round to nearest) before the taps
//    and the filter weight are taken from it -- what the D3D11.3 functional spec allows a fixed-function sampler to do
//    ("at least 8 bits of fraction").  The reference's GPU may therefore differ from the exact-fp32 weights used here by up
//    to this mode's deviation; tests/test_oracle_kat.py and scripts/weight_bits_deviation.py report it.
//  * g_libRayMarch: the view march as the work-graph variant states it (LibRayMarch.hlsl:205-287): tMax = GetTMax only, no
//    ComputeTargetHit clamp (CSRayMarch.hlsl:102-106 has it).
int g_weightBits = 0;
bool g_libRayMarch = false;

inline Taps linearTaps(float coord01, uint32_t size, int32_t off)
{
	// LINEAR + CLAMP: c = u * size - 0.5 ; taps floor(c) and floor(c) + 1 clamped to [0, size-1]
	float c = coord01 * (float)size - 0.5f;
	if (g_weightBits > 0)
	{
		const float q = (float)(1u << g_weightBits);
		c = nearbyintf(c * q) / q;
	}
	const float fl = floorf(c);
	Taps t;
	t.f = c - fl;
	const int32_t i = f2i_floor(c);
	const int64_t a = (int64_t)i + off, b = a + 1;
	t.i0 = (int32_t)std::min<int64_t>(std::max<int64_t>(a, 0), (int64_t)size - 1);
	t.i1 = (int32_t)std::min<int64_t>(std::max<int64_t>(b, 0), (int64_t)size - 1);
	return t;
}

// Texture3D<float4>.SampleLevel(LINEAR_CLAMP, uvw, 0, offset) on RGBA16F -- RayMarch.hlsli:44-50, 74
float4 sampleRGBA16F(const uint16_t* tex, uint32_t N, float3 uvw, int ox = 0, int oy = 0, int oz = 0)
{
	const Taps tx = linearTaps(uvw.x, N, ox), ty = linearTaps(uvw.y, N, oy), tz = linearTaps(uvw.z, N, oz);
	const size_t sy = (size_t)N, sz = (size_t)N * N;
	const int32_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 }, zs[2] = { tz.i0, tz.i1 };
	float c[2][2][2][4];
	for (int k = 0; k < 2; ++k)
		for (int j = 0; j < 2; ++j)
			for (int i = 0; i < 2; ++i)
			{
				const uint16_t* p = tex + 4 * ((size_t)zs[k] * sz + (size_t)ys[j] * sy + (size_t)xs[i]);
				for (int ch = 0; ch < 4; ++ch) c[k][j][i][ch] = h2f(p[ch]);
			}
	float r[4];
	for (int ch = 0; ch < 4; ++ch)
	{
		const float c00 = lerpf(c[0][0][0][ch], c[0][0][1][ch], tx.f);
		const float c10 = lerpf(c[0][1][0][ch], c[0][1][1][ch], tx.f);
		const float c01 = lerpf(c[1][0][0][ch], c[1][0][1][ch], tx.f);
		const float c11 = lerpf(c[1][1][0][ch], c[1][1][1][ch], tx.f);
		const float c0 = lerpf(c00, c10, ty.f);
		const float c1 = lerpf(c01, c11, ty.f);
		r[ch] = lerpf(c0, c1, tz.f);
	}
	float4 o = { r[0], r[1], r[2], r[3] };
	return o;
}

// alpha only (same arithmetic as channel 3 above)
float sampleAlpha16F(const uint16_t* tex, uint32_t N, float3 uvw, int ox = 0, int oy = 0, int oz = 0)
{
	const Taps tx = linearTaps(uvw.x, N, ox), ty = linearTaps(uvw.y, N, oy), tz = linearTaps(uvw.z, N, oz);
	const size_t sy = (size_t)N, sz = (size_t)N * N;
	const int32_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 }, zs[2] = { tz.i0, tz.i1 };
	float c[2][2][2];
	for (int k = 0; k < 2; ++k)
		for (int j = 0; j < 2; ++j)
			for (int i = 0; i < 2; ++i)
				c[k][j][i] = h2f(tex[4 * ((size_t)zs[k] * sz + (size_t)ys[j] * sy + (size_t)xs[i]) + 3]);
	const float c00 = lerpf(c[0][0][0], c[0][0][1], tx.f);
	const float c10 = lerpf(c[0][1][0], c[0][1][1], tx.f);
	const float c01 = lerpf(c[1][0][0], c[1][0][1], tx.f);
	const float c11 = lerpf(c[1][1][0], c[1][1][1], tx.f);
	return lerpf(lerpf(c00, c10, ty.f), lerpf(c01, c11, ty.f), tz.f);
}

// Texture3D<float3>.SampleLevel(LINEAR_CLAMP, uvw, 0) on R11G11B10F -- RayMarch.hlsli:235-240
float3 sampleR11G11B10F(const uint32_t* tex, uint32_t L, float3 uvw)
{
	const Taps tx = linearTaps(uvw.x, L, 0), ty = linearTaps(uvw.y, L, 0), tz = linearTaps(uvw.z, L, 0);
	const size_t sy = (size_t)L, sz = (size_t)L * L;
	const int32_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 }, zs[2] = { tz.i0, tz.i1 };
	float3 c[2][2][2];
	for (int k = 0; k < 2; ++k)
		for (int j = 0; j < 2; ++j)
			for (int i = 0; i < 2; ++i)
				c[k][j][i] = unpack_r11g11b10f(tex[(size_t)zs[k] * sz + (size_t)ys[j] * sy + (size_t)xs[i]]);
	float r[3];
	for (int ch = 0; ch < 3; ++ch)
	{
		const float c00 = lerpf(idx(c[0][0][0], ch), idx(c[0][0][1], ch), tx.f);
		const float c10 = lerpf(idx(c[0][1][0], ch), idx(c[0][1][1], ch), tx.f);
		const float c01 = lerpf(idx(c[1][0][0], ch), idx(c[1][0][1], ch), tx.f);
		const float c11 = lerpf(idx(c[1][1][0], ch), idx(c[1][1][1], ch), tx.f);
		r[ch] = lerpf(lerpf(c00, c10, ty.f), lerpf(c01, c11, ty.f), tz.f);
	}
	return f3(r[0], r[1], r[2]);
}

// Texture2D<float>.SampleLevel(POINT_CLAMP, uv, 0) -- CSRayMarch.hlsl:68
float samplePointR32F(const float* tex, uint32_t W, uint32_t H, float u, float v)
{
	const int32_t x = clampi(f2i_floor(u * (float)W), 0, (int32_t)W - 1);
	const int32_t y = clampi(f2i_floor(v * (float)H), 0, (int32_t)H - 1);
	return tex[(size_t)y * W + x];
}

// Texture2D<float>.SampleCmpLevelZero(LINEAR_LESS_EQUAL, uv, ref) on D16_UNORM -- RayMarch.hlsli:109
float sampleCmpD16(const uint16_t* tex, uint32_t M, float u, float v, float ref)
{
	const Taps tx = linearTaps(u, M, 0), ty = linearTaps(v, M, 0);
	const float r = saturatef(ref);
	const float t00 = (float)tex[(size_t)ty.i0 * M + tx.i0] / 65535.0f;
	const float t10 = (float)tex[(size_t)ty.i0 * M + tx.i1] / 65535.0f;
	const float t01 = (float)tex[(size_t)ty.i1 * M + tx.i0] / 65535.0f;
	const float t11 = (float)tex[(size_t)ty.i1 * M + tx.i1] / 65535.0f;
	const float c00 = r <= t00 ? 1.0f : 0.0f, c10 = r <= t10 ? 1.0f : 0.0f;
	const float c01 = r <= t01 ? 1.0f : 0.0f, c11 = r <= t11 ? 1.0f : 0.0f;
	return lerpf(lerpf(c00, c10, tx.f), lerpf(c01, c11, tx.f), ty.f);
}

// ---------------------------------------------------------------------------------------------
// Sh/RayMarch.hlsli device functions
// ---------------------------------------------------------------------------------------------
// RayMarch.hlsli:170-177 (_TEXCOORD_INVERT_Y_ undefined)
inline float3 LocalToTex3DSpace(float3 pos) { return f3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f); }

// RayMarch.hlsli:128-155
bool ComputeRayOrigin(float3& rayOrigin, float3 rayDir)
{
	if (fabsf(rayOrigin.x) <= 1.0f && fabsf(rayOrigin.y) <= 1.0f && fabsf(rayOrigin.z) <= 1.0f) return true;

	float U = kFltMax;
	bool isHit = false;
	for (int i = 0; i < 3; ++i)
	{
		const float u = (-signf(idx(rayDir, i)) - idx(rayOrigin, i)) / idx(rayDir, i);
		if (u < 0.0f) continue;
		const int j = (i + 1) % 3, k = (i + 2) % 3;
		if (fabsf(idx(rayDir, j) * u + idx(rayOrigin, j)) > 1.0f) continue;
		if (fabsf(idx(rayDir, k) * u + idx(rayOrigin, k)) > 1.0f) continue;
		if (u < U)
		{
			U = u;
			isHit = true;
		}
	}
	rayOrigin = f3(clampf(rayDir.x * U + rayOrigin.x, -1.0f, 1.0f),
		clampf(rayDir.y * U + rayOrigin.y, -1.0f, 1.0f),
		clampf(rayDir.z * U + rayOrigin.z, -1.0f, 1.0f));
	return isHit;
}

// RayMarch.hlsli:160-165
inline float ComputeTargetHit(float3 rayOrigin, float3 target, float3 rayDir)
{
	const float3 u = (target - rayOrigin) / rayDir;
	return fmaxf(fmaxf(u.x, u.y), u.z);
}

// RayMarch.hlsli:82-97
float GetTMax(float3 pos, float3 rayOrigin, float3 rayDir, const float* worldViewProjI)
{
	if (pos.z >= 1.0f) return kFltMax;
	float4 p = { pos.x, pos.y, pos.z, 1.0f };
	const float4 hpos = mul44(p, worldViewProjI);
	pos = f3(hpos.x / hpos.w, hpos.y / hpos.w, hpos.z / hpos.w);
	const float3 t = (pos - rayOrigin) / rayDir;
	return fmaxf(fmaxf(t.x, t.y), t.z);
}

// RayMarch.hlsli:182-192
inline float GetStep(float dDensity, float transm, float density, float step)
{
	const float factorEv = fminf(1.0f / 256.0f / fabsf(dDensity), 2.0f);
	const float factorUi = fminf(1.0f - density, 1.0f);
	const float factorTh = 1.0f - transm;
	step *= fmaxf(factorTh * 1.5f * factorEv * factorUi, 1.0f);	// the compiled order: ((factorTh * 1.5) * factorEv) * factorUi
	return step;
}

// RayMarch.hlsli:197-230 (mipLevel = 0)
void CastLightRay(float& transm, const uint16_t* tex, uint32_t N, float3 rayOrigin, float3 rayDir,
	float stepScale, uint32_t numSamples, uint64_t& samples)
{
	float t = stepScale;
	float step = stepScale;
	float prevDensity = 0.0f;
	for (uint32_t i = 0; i < numSamples; ++i)
	{
		const float3 pos = rayAt(rayOrigin, rayDir, t);
		if (fabsf(pos.x) > 1.0f || fabsf(pos.y) > 1.0f || fabsf(pos.z) > 1.0f) break;
		const float3 uvw = LocalToTex3DSpace(pos);

		const float density = sampleAlpha16F(tex, N, uvw);
		++samples;

		const float dDensity = density - prevDensity;
		const float opacity = saturatef(density * step);
		const float newStep = GetStep(dDensity, transm, opacity, stepScale);
		// step = (step + newStep) * 0.5;	-- :219, dead: overwritten at :227
		prevDensity = density;

		transm *= 1.0f - density * kAbsorption;
		if (transm < kZeroThreshold) break;

		step = newStep;
		t += step;
	}
}

// RayMarch.hlsli:55-77
float3 GetDensityGradient(const uint16_t* tex, uint32_t N, float3 uvw)
{
	float q[6];
	q[0] = sampleAlpha16F(tex, N, uvw, -1, 0, 0);
	q[1] = sampleAlpha16F(tex, N, uvw, 1, 0, 0);
	q[2] = sampleAlpha16F(tex, N, uvw, 0, -1, 0);
	q[3] = sampleAlpha16F(tex, N, uvw, 0, 1, 0);
	q[4] = sampleAlpha16F(tex, N, uvw, 0, 0, -1);
	q[5] = sampleAlpha16F(tex, N, uvw, 0, 0, 1);
	return f3(q[1] - q[0], q[3] - q[2], q[5] - q[4]);
}

// XUSG/Shaders/SHIrradianceTypeless.hlsli:15-36
float3 EvaluateSHIrradiance(const float* sh /* 9 x float3 */, float3 norm)
{
	// XUSG/Shaders/SHIrradiance.hlsli as the reference's compiler emitted it inside CSRayMarchL.cso (oracle/dxil): the constants folded
	// to fp32 (3 c3 and 2 c1, 2 c2 as single constants), the sum in this order
	const float c1 = 0.4290427565574646f, c3x3 = 0.7431238889694214f, c3 = 0.24770796298980713f, c4 = 0.8862269520759583f;
	const float c1x2 = 0.8580855131149292f, c2x2 = 1.0233267545700073f;
	const float x = -norm.x, y = -norm.y, z = norm.z;
	float r[3];
	for (int ch = 0; ch < 3; ++ch)
	{
#define SHC(i) sh[3 * (i) + ch]
		const float t8 = (norm.x * norm.x - norm.y * norm.y) * c1 * SHC(8);
		const float t6 = (z * z * c3x3 + -c3) * SHC(6);
		const float t0 = SHC(0) * c4;
		const float q = ((z * SHC(5) + SHC(4) * x) * y + SHC(7) * x * z) * c1x2;
		const float l = (SHC(3) * x + SHC(1) * y + z * SHC(2)) * c2x2;
		const float v = t6 + t0 + t8 + l + q;
#undef SHC
		r[ch] = fmaxf(0.0f, v);
	}
	return f3(r[0], r[1], r[2]);
}

// CSRayMarch.hlsl:28-53
float3 GetLocalPos(float px, float py, uint32_t slice, float dims)
{
	float x = (px + 0.5f) / dims * 2.0f - 1.0f;
	float y = (py + 0.5f) / dims * 2.0f - 1.0f;
	y = -y;
	switch (slice)
	{
	case 0: return f3(1.0f, y, -x);
	case 1: return f3(-1.0f, y, x);
	case 2: return f3(x, 1.0f, -y);
	case 3: return f3(x, -1.0f, y);
	case 4: return f3(x, y, 1.0f);
	case 5: return f3(-x, y, -1.0f);
	default: return f3(0.0f, 0.0f, 0.0f);
	}
}

// edge of a cube map's mip as D3D defines it (what GetDimensions answers in CSRayMarch.hlsl:31 / PSCube.hlsli:55): max(1, N >> mip)
inline uint32_t mipDim(uint32_t N, uint32_t mip) { const uint32_t d = N >> mip; return d ? d : 1u; }

inline uint64_t chainTexels(uint32_t N)
{
	uint64_t s = 0;
	for (uint32_t m = 0; m < MVO_NUM_CUBE_MIP; ++m) { const uint64_t d = mipDim(N, m); s += 6 * d * d; }
	return s;
}

inline uint64_t mipOffset(uint32_t N, uint32_t mip)
{
	uint64_t s = 0;
	for (uint32_t m = 0; m < mip; ++m) { const uint64_t d = mipDim(N, m); s += 6 * d * d; }
	return s;
}

// Shared body of CSRayMarch.hlsl:111-155 and RayCast.hlsli:56-104 (identical loops; RayCast has no ComputeTargetHit clamp)
float4 marchLoop(const uint16_t* vol, uint32_t N, const uint32_t* lightMap, uint32_t L,
	float3 rayOrigin, float3 rayDir, float tMax, uint32_t smpCount, uint64_t& nSamples, uint64_t& nLit)
{
	const float stepScale = kMaxDist / (float)smpCount;
	float4 scatter = { 0.0f, 0.0f, 0.0f, 0.0f };

	float t = 0.0f;
	float step = stepScale;
	float prevDensity = 0.0f;
	for (uint32_t i = 0; i < smpCount; ++i)
	{
		const float3 pos = rayAt(rayOrigin, rayDir, t);
		if (fabsf(pos.x) > 1.0f || fabsf(pos.y) > 1.0f || fabsf(pos.z) > 1.0f) break;
		const float3 uvw = LocalToTex3DSpace(pos);

		float4 color = sampleRGBA16F(vol, N, uvw);
		++nSamples;
		float newStep = stepScale;

		if (color.w > kZeroThreshold)
		{
			const float3 light = sampleR11G11B10F(lightMap, L, f3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f));
			++nLit;

			const float transm = 1.0f - scatter.w;
			const float dDensity = color.w - prevDensity;
			// GetStep (RayMarch.hlsli:182-192) as compiled: factorTh = 1 - transm = 1 - (1 - scatter.w) was simplified to scatter.w
			// (fast math), and the product is taken in the order ((scatter.w * 1.5) * factorEv) * factorUi
			{
				const float factorEv = fminf(1.0f / 256.0f / fabsf(dDensity), 2.0f);
				const float factorUi = fminf(1.0f - color.w, 1.0f);
				newStep = fmaxf(scatter.w * 1.5f * factorEv * factorUi, 1.0f) * stepScale;
			}
			// step = (step + newStep) * 0.5;	-- CSRayMarch.hlsl:136, dead: overwritten at :150
			prevDensity = color.w;

			// :140-146 as compiled: a = (transm * ABSORPTION) * alpha, scatter.rgb += (a * color) * light, scatter.a += a
			const float a = transm * kAbsorption * color.w;
			scatter.x += a * color.x * light.x;
			scatter.y += a * color.y * light.y;
			scatter.z += a * color.z * light.z;
			scatter.w += a;

			if (transm < kZeroThreshold) break;
		}

		step = newStep;
		t += step;
		if (t > tMax) break;
	}
	(void)step;

	scatter.x *= kInvTwoPi; scatter.y *= kInvTwoPi; scatter.z *= kInvTwoPi;	// :155 `/ (2 PI)` compiled as a multiplication by half(1 / (2 PI))
	return scatter;
}
} // namespace

// =============================================================================================
// exported
// =============================================================================================
extern "C" {

uint16_t mvo_f32_to_f16(float f) { return f2h(f); }
float mvo_f16_to_f32(uint16_t h) { return h2f(h); }
uint32_t mvo_pack_r11g11b10f(float r, float g, float b) { return pack_r11g11b10f(r, g, b); }
void mvo_unpack_r11g11b10f(uint32_t p, float rgb[3]) { const float3 c = unpack_r11g11b10f(p); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z; }
uint64_t mvo_cube_chain_texels(uint32_t N) { return chainTexels(N); }
uint64_t mvo_cube_mip_offset(uint32_t N, uint32_t mip) { return mipOffset(N, mip); }

void mvo_sample_rgba16f(const uint16_t* tex, uint32_t N, const float uvw[3], const int32_t off[3], float out[4])
{
	const float4 c = sampleRGBA16F(tex, N, f3(uvw[0], uvw[1], uvw[2]), off ? off[0] : 0, off ? off[1] : 0, off ? off[2] : 0);
	out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}

void mvo_sample_r11g11b10f(const uint32_t* tex, uint32_t L, const float uvw[3], float out[3])
{
	const float3 c = sampleR11G11B10F(tex, L, f3(uvw[0], uvw[1], uvw[2]));
	out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

float mvo_sample_point_r32f(const float* tex, uint32_t W, uint32_t H, float u, float v) { return samplePointR32F(tex, W, H, u, v); }
float mvo_sample_cmp_d16(const uint16_t* tex, uint32_t M, float u, float v, float ref) { return sampleCmpD16(tex, M, u, v, ref); }

int mvo_compute_ray_origin(float origin[3], const float dir[3])
{
	float3 o = f3(origin[0], origin[1], origin[2]);
	const bool hit = ComputeRayOrigin(o, f3(dir[0], dir[1], dir[2]));
	origin[0] = o.x; origin[1] = o.y; origin[2] = o.z;
	return hit ? 1 : 0;
}

float mvo_compute_target_hit(const float origin[3], const float target[3], const float dir[3])
{
	return ComputeTargetHit(f3(origin[0], origin[1], origin[2]), f3(target[0], target[1], target[2]), f3(dir[0], dir[1], dir[2]));
}

float mvo_get_step(float dDensity, float transm, float density, float step) { return GetStep(dDensity, transm, density, step); }

void mvo_get_local_pos(float px, float py, uint32_t slice, uint32_t dims, float out[3])
{
	const float3 p = GetLocalPos(px, py, slice, (float)dims);
	out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

void mvo_cast_light_ray(float* transm, const uint16_t* tex, uint32_t N, const float origin[3],
	const float dir[3], float stepScale, uint32_t numSamples, uint64_t* samples)
{
	uint64_t n = 0;
	CastLightRay(*transm, tex, N, f3(origin[0], origin[1], origin[2]), f3(dir[0], dir[1], dir[2]), stepScale, numSamples, n);
	if (samples) *samples = n;
}

void mvo_eval_sh_irradiance(const float sh27[27], const float norm[3], float out[3])
{
	const float3 r = EvaluateSHIrradiance(sh27, f3(norm[0], norm[1], norm[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

int mvo_num_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

// Copy with every thread writing its own share of the pages first: on a multi-socket host the destination then lies spread over the
// memory of all sockets (first-touch placement) instead of on the one whose thread allocated it -- for arrays all threads read at random
// (the volumes, when the oracle is timed as the CPU baseline).  dst must be freshly allocated, untouched memory.
void mvo_parallel_copy(void* dst, const void* src, size_t bytes)
{
	const int64_t pages = (int64_t)((bytes + 4095) / 4096);
#pragma omp parallel for schedule(static)
	for (int64_t p = 0; p < pages; ++p)
	{
		const size_t off = (size_t)p * 4096, n = off + 4096 <= bytes ? 4096 : bytes - off;
		memcpy((char*)dst + off, (const char*)src + off, n);
	}
}

void mvo_set_num_threads(int n)
{
#ifdef _OPENMP
	if (n > 0) omp_set_num_threads(n);
#else
	(void)n;
#endif
}

// ---------------------------------------------------------------------------------------------
// a6: CSInitGridData.hlsl:10-27
// ---------------------------------------------------------------------------------------------
void mvo_init_grid_data(uint32_t N, uint16_t* rgba)
{
	const float gridSize = (float)N;
#pragma omp parallel for schedule(static)
	for (int64_t z = 0; z < (int64_t)N; ++z)
		for (uint32_t y = 0; y < N; ++y)
			for (uint32_t x = 0; x < N; ++x)
			{
				const float3 pos = f3(((float)x + 0.5f) / gridSize * 2.0f - 1.0f,
					((float)y + 0.5f) / gridSize * 2.0f - 1.0f,
					((float)z + 0.5f) / gridSize * 2.0f - 1.0f);
				const float r_sq = dot3(pos, pos);
				float a = 1.0f - r_sq;
				a *= a;
				a = saturatef(a * a * 2.0f);

				const float3 colorU = f3(1.0f, 0.6f, 0.0f);
				const float3 colorD = f3(0.5f, 0.8f, 1.0f);
				const float s = saturatef(pos.y * 0.5f + 0.2f);
				// lerp(x, y, s) = x + s * (y - x)
				const float3 color = f3(lerpf(colorD.x, colorU.x, s), lerpf(colorD.y, colorU.y, s), lerpf(colorD.z, colorU.z, s));

				uint16_t* o = rgba + 4 * (((size_t)z * N + y) * N + x);
				o[0] = f2h(color.x); o[1] = f2h(color.y); o[2] = f2h(color.z); o[3] = f2h(a);
			}
}

// CSR32FToRGBA16F.hlsl:16-26 -- Texture3D<float>.SampleLevel(LINEAR_CLAMP) of a w x h x d R32F source
// Texture3D<float>.SampleLevel(LINEAR_CLAMP, uvw, 0) on an R32F volume of w x h x d texels (the sampler contract, also used by oracle/dxil)
float mvo_sample_r32f_3d(const float* src, uint32_t w, uint32_t h, uint32_t d, const float uvw[3])
{
	const Taps tx = linearTaps(uvw[0], w, 0), ty = linearTaps(uvw[1], h, 0), tz = linearTaps(uvw[2], d, 0);
#define SRC(i, j, k) src[((size_t)(k) * h + (j)) * w + (i)]
	const float c00 = lerpf(SRC(tx.i0, ty.i0, tz.i0), SRC(tx.i1, ty.i0, tz.i0), tx.f);
	const float c10 = lerpf(SRC(tx.i0, ty.i1, tz.i0), SRC(tx.i1, ty.i1, tz.i0), tx.f);
	const float c01 = lerpf(SRC(tx.i0, ty.i0, tz.i1), SRC(tx.i1, ty.i0, tz.i1), tx.f);
	const float c11 = lerpf(SRC(tx.i0, ty.i1, tz.i1), SRC(tx.i1, ty.i1, tz.i1), tx.f);
#undef SRC
	return lerpf(lerpf(c00, c10, ty.f), lerpf(c01, c11, ty.f), tz.f);
}

void mvo_r32f_to_rgba16f(const float* src, uint32_t w, uint32_t h, uint32_t d, uint32_t N, uint16_t* rgba)
{
	const float gridSize = (float)N;
	const uint16_t one = f2h(1.0f);
#pragma omp parallel for schedule(static)
	for (int64_t z = 0; z < (int64_t)N; ++z)
		for (uint32_t y = 0; y < N; ++y)
			for (uint32_t x = 0; x < N; ++x)
			{
				const float uvw[3] = { ((float)x + 0.5f) / gridSize, ((float)y + 0.5f) / gridSize, ((float)z + 0.5f) / gridSize };
				const float a = mvo_sample_r32f_3d(src, w, h, d, uvw);
				uint16_t* o = rgba + 4 * (((size_t)z * N + y) * N + x);
				o[0] = one; o[1] = one; o[2] = one; o[3] = f2h(a * 0.25f);
			}
}

// ---------------------------------------------------------------------------------------------
// a1: CSVolumeCull.hlsl:13-78 + VolumeCull.hlsli
// ---------------------------------------------------------------------------------------------
void mvo_cull(uint32_t V, const void* perObjectBytes, const uint32_t* volumeDescs, const void* cbBytes,
	uint32_t maxRaySamples, uint16_t* volumeInfo, uint32_t* visibleVolumes, uint32_t* visibleCount,
	uint32_t* cubeMapVolumes, uint32_t* cubeMapCount)
{
	const PerObject* perObjects = (const PerObject*)perObjectBytes;
	CBPerFrame cb;
	memcpy(&cb, cbBytes, sizeof(cb));
	const float vpW = cb.viewport[0], vpH = cb.viewport[1];
	const float3 eyePt = f3(cb.eyePos[0], cb.eyePos[1], cb.eyePos[2]);

	// VolumeCull.hlsli:119-138 edge -> vertex-lane table ; :213-223 face -> edge table
	static const uint8_t edgeLanes[12][2] = { {0,1},{3,2},{1,3},{2,0},{6,7},{5,4},{4,6},{7,5},{4,0},{2,6},{7,3},{1,5} };
	static const uint8_t faceEdges[6][4] = { {8,3,9,6},{10,2,11,7},{0,8,5,11},{1,10,4,9},{0,2,1,3},{4,6,5,7} };

	uint32_t nVis = 0, nCube = 0;
	for (uint32_t volumeId = 0; volumeId < V; ++volumeId)
	{
		const PerObject& po = perObjects[volumeId];

		// ProjectToViewport (VolumeCull.hlsli:27-41), one vertex per lane
		float vx[8], vy[8], vz[8];
		bool anyInView = false;
		for (uint32_t i = 0; i < 8; ++i)
		{
			float4 p;
			p.x = (i & 1) ? 1.0f : -1.0f;
			p.y = ((i >> 1) & 1) ? 1.0f : -1.0f;
			p.z = (i >> 2) ? 1.0f : -1.0f;
			p.w = 1.0f;
			p = mul44(p, po.wvp);
			p.x /= p.w; p.y /= p.w; p.z /= p.w;
			// (as compiled, CSVolumeCull.cso %87-%92: y = 0.5 - y * 0.5 in ONE rounding -- the source's 1 - (y * 0.5 + 0.5) has two; found in round 6 by
			// walking the cull's decision boundaries, tests/test_oracle_vs_reference_dxil.py::test_cull_agrees_at_its_decision_boundaries)
			p.x = p.x * 0.5f + 0.5f;
			p.y = 0.5f - p.y * 0.5f;
			vx[i] = p.x * vpW; vy[i] = p.y * vpH; vz[i] = p.z;
			// CSVolumeCull.hlsl:32
			const bool isInView = (vx[i] <= vpW && vy[i] <= vpH && vx[i] >= 0.0f && vy[i] >= 0.0f) && vz[i] > 0.0f && vz[i] < 1.0f;
			anyInView = anyInView || isInView;
		}
		if (!anyInView) continue;	// :38

		const uint32_t volumeIn = volumeDescs[volumeId];
		uint32_t raySampleCount = maxRaySamples;

		// GenVisibilityMask (VolumeCull.hlsli:46-66)
		const float3 localSpaceEyePt = mul43(eyePt, po.worldI);
		uint32_t faceMask = 0;
		for (uint32_t face = 0; face < 6; ++face)
		{
			const float viewComp = idx(localSpaceEyePt, (int)(face >> 1));
			const bool vis = (face & 1) ? viewComp > -1.0f : viewComp < 1.0f;
			if (vis) faceMask |= 1u << face;
		}

		// GetCubeEdgePairPerLane (:164-181)
		float ex[12], ey[12];
		for (uint32_t e = 0; e < 12; ++e)
		{
			const uint32_t a = edgeLanes[e][0], b = edgeLanes[e][1];
			ex[e] = vx[b] - vx[a];
			ey[e] = vy[b] - vy[a];
		}

		// EstimateCubeMaxEdgeLength (:248-262): WaveActiveMax over lanes 0-5 of max(|e0|, |e1|)
		float maxEdge = 0.0f;
		for (uint32_t lane = 0; lane < 6; ++lane)
		{
			const float l0 = sqrtf(ex[2 * lane] * ex[2 * lane] + ey[2 * lane] * ey[2 * lane]);
			const float l1 = sqrtf(ex[2 * lane + 1] * ex[2 * lane + 1] + ey[2 * lane + 1] * ey[2 * lane + 1]);
			const float ms = fmaxf(l0, l1);
			maxEdge = lane == 0 ? ms : fmaxf(maxEdge, ms);
		}

		// EstimateCubeMapLOD (:267-294), upscale = 2, raySampleCountScale = 2
		const uint32_t cubeMapSize = volumeIn >> 18;
		const uint32_t numMips = (volumeIn >> 14) & 0xf;
		// (as compiled, %216-%218: s * 2 TIMES the constant 1 / sqrt(3) folded to fp32, not a division by sqrt(3))
		float s = maxEdge / 2.0f;
		float raySampleAmt = 2.0f * s * asfloat(0x3f13cd3au);
		const uint32_t raySampleCnt = f2u(ceilf(raySampleAmt));
		raySampleCount = std::min(raySampleCnt, raySampleCount);
		raySampleAmt = fminf(raySampleAmt, (float)raySampleCount);
		s = raySampleAmt / 2.0f * kSqrt3;
		const uint32_t level = f2u(fmaxf(log2f((float)cubeMapSize / s), 0.0f));
		const uint32_t mipLevel = std::min(level, numMips - 1);

		// EstimateProjCoverage (:299-322): WaveActiveSum over lanes 0-5 (summed in lane order here)
		float projCov = 0.0f;
		for (uint32_t face = 0; face < 6; ++face)
		{
			float fx[4], fy[4];
			for (uint32_t k = 0; k < 4; ++k) { fx[k] = ex[faceEdges[face][k]]; fy[k] = ey[faceEdges[face][k]]; }
			float faceArea = 0.0f;
			if (faceMask & (1u << face))
				faceArea = 0.5f * fabsf(fx[0] * fy[1] - fy[0] * fx[1]) + 0.5f * fabsf(fx[2] * fy[3] - fy[2] * fx[3]);
			projCov = face == 0 ? faceArea : projCov + faceArea;
		}

		// CSVolumeCull.hlsl:62-77
		const uint32_t edgeLength = cubeMapSize >> mipLevel;
		const float visibleFaceCount = (float)__builtin_popcount(faceMask);
		const float cubeMapPix = (float)(edgeLength * edgeLength) * visibleFaceCount;
		const bool useCubeMap = cubeMapPix <= projCov;
		const uint32_t maskBits = useCubeMap ? (faceMask | kCubeMapRayMarchBit) : faceMask;
		const uint32_t volTexId = volumeIn & 0x3fff;

		if (useCubeMap) cubeMapVolumes[nCube++] = volumeId;
		volumeInfo[4 * volumeId + 0] = (uint16_t)mipLevel;
		volumeInfo[4 * volumeId + 1] = (uint16_t)raySampleCount;
		volumeInfo[4 * volumeId + 2] = (uint16_t)maskBits;
		volumeInfo[4 * volumeId + 3] = (uint16_t)volTexId;
		visibleVolumes[nVis++] = volumeId;
	}
	*visibleCount = nVis;
	*cubeMapCount = nCube;
}

// ---------------------------------------------------------------------------------------------
// a2: CSRayMarchL.hlsl:20-121
// ---------------------------------------------------------------------------------------------
// Share of the light march a caller wants (the multi-GPU loop shares the voxel blocks of the relit map out over the ranks): only voxels of
// the 4x4x4 blocks [lo, hi), numbered x fastest, are marched and written.  Default: all.
static uint32_t g_lightBlockLo = 0u, g_lightBlockHi = 0xffffffffu;
void mvo_set_light_block_range(uint32_t lo, uint32_t hi) { g_lightBlockLo = lo; g_lightBlockHi = hi; }

uint32_t mvo_march_light(uint32_t V, uint32_t N, uint32_t L, uint32_t M,
	const uint16_t* const* volumes, const uint32_t* volumeDescs,
	const void* perObjectBytes, const void* cbBytes, const uint16_t* shadowMap,
	const float* sh27, const uint32_t* visibleVolumes, uint32_t visibleCount,
	uint32_t maxLightSamples, uint32_t* lightMaps, mvo_stats* stats)
{
	const PerObject* perObjects = (const PerObject*)perObjectBytes;
	CBPerFrame cb;
	memcpy(&cb, cbBytes, sizeof(cb));

	// :29-33
	uint32_t volumeId;
	if (visibleCount) volumeId = visibleVolumes[cb.frameIdx % visibleCount];
	else volumeId = cb.frameIdx % V;

	const uint32_t hasLightProbe = sh27 ? 1u : 0u;
	const float gStep = kMaxDist / (float)maxLightSamples;	// RayMarch.hlsli:18
	const float gridSize = (float)L;
	const uint32_t volTexId0 = volumeDescs[volumeId] & 0x3fff;
	const PerObject& po0 = perObjects[volumeId];
	uint32_t* out = lightMaps + (size_t)volumeId * L * L * L;
	uint64_t totalSamples = 0;

#pragma omp parallel for collapse(2) schedule(dynamic, 1) reduction(+ : totalSamples)
	for (int64_t zz = 0; zz < (int64_t)L; ++zz)
		for (uint32_t yy = 0; yy < L; ++yy)
			for (uint32_t xx = 0; xx < L; ++xx)
			{
				{
					const uint32_t B = (L + 3u) >> 2, b = (((uint32_t)zz >> 2) * B + (yy >> 2)) * B + (xx >> 2);
					if (b < g_lightBlockLo || b >= g_lightBlockHi) continue;
				}
				uint64_t nS = 0;
				float3 rayOrigin = f3(((float)xx + 0.5f) / gridSize * 2.0f - 1.0f,
					((float)yy + 0.5f) / gridSize * 2.0f - 1.0f,
					((float)zz + 0.5f) / gridSize * 2.0f - 1.0f);

				// LocalToTex3DSpace(rayOrigin) = ((x + 0.5) / L * 2 - 1) * 0.5 + 0.5, compiled (fast math) as (x + 0.5) / L
				const float3 uvw = f3(((float)xx + 0.5f) / gridSize, ((float)yy + 0.5f) / gridSize, ((float)zz + 0.5f) / gridSize);
				const float density = sampleAlpha16F(volumes[volTexId0], N, uvw);
				const bool hasDensity = density >= kZeroThreshold;	// :46

				const float3 localPos = rayOrigin;
				rayOrigin = mul43(rayOrigin, po0.world);	// light-map space to world space

				// ShadowTest (RayMarch.hlsli:103-112): mul(float4(pos,1), g_shadowViewProj).xyz
				float shadow;
				{
					// In the COMPILED order (CSRayMarchL.cso %109-%125; found in round 6 by reading the shader's fp32 `shadow` out through the light colour /
					// ambient constants, tests/test_oracle_vs_reference_dxil.py::test_light_march_shadow_in_full_precision): v = 0.5 - y * 0.5 (ONE rounding; the
					// source's 1 - (y * 0.5 + 0.5) has two), and the reference depth as (row.w - 0.0027) + dot3, not (dot3 + row.w) - 0.0027.
					const float4 p = { rayOrigin.x, rayOrigin.y, rayOrigin.z, 1.0f };
					const float4 ls = mul44(p, cb.shadowViewProj);
					const float* m = cb.shadowViewProj;
					const float su = ls.x * 0.5f + 0.5f;
					const float sv = 0.5f - ls.y * 0.5f;
					const float dz = p.x * m[8] + p.y * m[9] + p.z * m[10];
					const float ref = (m[11] + -0.0027f) + dz;
					shadow = sampleCmpD16(shadowMap, M, su, sv, ref);
				}

				float ao = 1.0f;
				float3 irradiance = f3(0.0f, 0.0f, 0.0f);

				if (hasDensity)
				{
					float3 aoRayDir = f3(0.0f, 0.0f, 0.0f);
					if (hasLightProbe)
					{
						aoRayDir = GetDensityGradient(volumes[volTexId0], N, uvw);
						aoRayDir = f3(-aoRayDir.x, -aoRayDir.y, -aoRayDir.z);
						const bool nz = fabsf(aoRayDir.x) > 0.0f || fabsf(aoRayDir.y) > 0.0f || fabsf(aoRayDir.z) > 0.0f;
						// :69 falls back to rayOrigin.xyz, which at that point is already the WORLD-space position (:47)
						aoRayDir = nz ? aoRayDir : rayOrigin;
						aoRayDir = mul33(aoRayDir, po0.world);
						aoRayDir = normalize3(aoRayDir);
						irradiance = EvaluateSHIrradiance(sh27, normalize3(aoRayDir));	// GetIrradiance normalises again (RayMarch.hlsli:121)
					}

					for (uint32_t n = 0; n < V; ++n)	// :77-110
					{
						const uint32_t volTexId = volumeDescs[n] & 0x3fff;
						const PerObject& po = perObjects[n];
						float3 localRayOrigin = mul43(rayOrigin, po.worldI);

						if (shadow >= kZeroThreshold)
						{
							const float3 localSpaceLightPt = mul33(f3(cb.lightPos[0], cb.lightPos[1], cb.lightPos[2]), po.worldI);
							const float3 rayDir = normalize3(localSpaceLightPt);
							if (!ComputeRayOrigin(localRayOrigin, rayDir)) continue;
							CastLightRay(shadow, volumes[volTexId], N, localRayOrigin, rayDir, gStep, maxLightSamples, nS);
						}

						if (hasLightProbe)
						{
							const float3 rayDir = normalize3(mul33(aoRayDir, po.worldI));
							if (!ComputeRayOrigin(localRayOrigin, rayDir)) continue;
							float transm = 1.0f;
							CastLightRay(transm, volumes[volTexId], N, localRayOrigin, rayDir, gStep, maxLightSamples, nS);
							ao *= n == volumeId ? transm : sqrtf(sqrtf(saturatef(transm + 0.5f)));	// pow(x, 0.25)
						}
					}
				}
				(void)localPos;

				const float3 lightColor = f3(cb.lightColor[0] * cb.lightColor[3], cb.lightColor[1] * cb.lightColor[3], cb.lightColor[2] * cb.lightColor[3]);
				float3 ambient = f3(cb.ambient[0] * cb.ambient[3], cb.ambient[1] * cb.ambient[3], cb.ambient[2] * cb.ambient[3]);
				if (hasLightProbe) ambient = f3(ao * irradiance.x, ao * irradiance.y, ao * irradiance.z);

				const float3 result = f3(shadow * lightColor.x + ambient.x, shadow * lightColor.y + ambient.y, shadow * lightColor.z + ambient.z);
				out[((size_t)zz * L + yy) * L + xx] = pack_r11g11b10f(result.x, result.y, result.z);
				totalSamples += nS;
			}

	if (stats) stats->light_samples += totalSamples;
	return volumeId;
}

// ---------------------------------------------------------------------------------------------
// a3: CSRayMarch.hlsl:77-158 compiled as CSRayMarchV
// ---------------------------------------------------------------------------------------------
void mvo_march_view(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H,
	const uint16_t* const* volumes, const uint32_t* lightMaps,
	const float* depthMap, const void* perObjectBytes, const void* cbBytes,
	const uint16_t* volumeInfo, const uint32_t* cubeMapVolumes, uint32_t cubeMapCount,
	uint16_t* cubeMaps, float* cubeDepths, mvo_stats* stats)
{
	(void)V;
	const PerObject* perObjects = (const PerObject*)perObjectBytes;
	CBPerFrame cb;
	memcpy(&cb, cbBytes, sizeof(cb));
	const float3 eyePt = f3(cb.eyePos[0], cb.eyePos[1], cb.eyePos[2]);
	const uint64_t chain = chainTexels(N);
	uint64_t nSamples = 0, nLit = 0, nRays = 0;

	// One parallel loop over every texel row of every marched face of every cube-map volume (round 3: a parallel region per face left
	// 128 threads with 128 rows of 128 short rays each and spent its time in fork / join: 40 M samples/s on 128 threads).
	struct Job { uint32_t g, faceId, dims; };
	std::vector<Job> jobs;
	std::vector<uint64_t> rowStart(1, 0);
	for (uint32_t g = 0; g < cubeMapCount; ++g)
	{
		const uint32_t volumeId = cubeMapVolumes[g];	// :80
		const uint32_t dims = mipDim(N, volumeInfo[4 * volumeId + 0]);	// rwCubeMap.GetDimensions (:31)
		for (uint32_t faceId = 0; faceId < 6; ++faceId)
			if (volumeInfo[4 * volumeId + 2] & (1u << faceId))	// :85
			{
				jobs.push_back({ g, faceId, dims });
				rowStart.push_back(rowStart.back() + dims);
			}
	}
	const int64_t totalRows = (int64_t)rowStart.back();

#pragma omp parallel for schedule(dynamic, 4) reduction(+ : nSamples, nLit, nRays)
	for (int64_t row = 0; row < totalRows; ++row)
	{
		const size_t j = (size_t)(std::upper_bound(rowStart.begin(), rowStart.end(), (uint64_t)row) - rowStart.begin()) - 1u;
		const uint32_t g = jobs[j].g, faceId = jobs[j].faceId, dims = jobs[j].dims;
		const int64_t ty = row - (int64_t)rowStart[j];
		const uint32_t volumeId = cubeMapVolumes[g];
		const uint32_t mipLevel = volumeInfo[4 * volumeId + 0];
		const uint32_t smpCount = volumeInfo[4 * volumeId + 1];
		const uint32_t volTexId = volumeInfo[4 * volumeId + 3];
		const PerObject& po = perObjects[volumeId];
		uint16_t* cubeMap = cubeMaps + 4 * ((size_t)volumeId * chain + mipOffset(N, mipLevel));
		float* cubeDepth = cubeDepths + ((size_t)volumeId * chain + mipOffset(N, mipLevel));
		const uint16_t* vol = volumes[volTexId];
		const uint32_t* lightMap = lightMaps + (size_t)volumeId * L * L * L;
		for (uint32_t tx = 0; tx < dims; ++tx)
		{
			float3 rayOrigin = mul43(eyePt, po.worldI);	// :89
			const float3 target = GetLocalPos((float)tx, (float)ty, faceId, (float)dims);
			const float3 rayDir = normalize3(target - rayOrigin);
			if (!ComputeRayOrigin(rayOrigin, rayDir)) continue;	// :96

			float tMax = g_libRayMarch ? kFltMax : ComputeTargetHit(rayOrigin, target, rayDir);	// CSRayMarch.hlsl:102; absent from LibRayMarch.hlsl:222-232
			const size_t texel = ((size_t)faceId * dims + (size_t)ty) * dims + tx;

			// GetClipPos (:59-71)
			float3 clip;
			{
				float4 hPos = { rayOrigin.x + rayDir.x * 0.01f, rayOrigin.y + rayDir.y * 0.01f, rayOrigin.z + rayDir.z * 0.01f, 1.0f };
				hPos = mul44(hPos, po.wvp);
				const float cx = hPos.x / hPos.w, cy = hPos.y / hPos.w;
				const float u = cx * 0.5f + 0.5f;
				const float v = 0.5f - cy * 0.5f;	// `1 - (y * 0.5 + 0.5)` compiled (fast math) as 0.5 - y * 0.5
				const float z = samplePointR32F(depthMap, W, H, u, v);
				clip = f3(cx, cy, z);
			}
			cubeDepth[texel] = clip.z;	// :105
			tMax = fminf(GetTMax(clip, rayOrigin, rayDir, po.wvpI), tMax);	// :106

			uint64_t sn = 0, ln = 0;
			const float4 scatter = marchLoop(vol, N, lightMap, L, rayOrigin, rayDir, tMax, smpCount, sn, ln);
			nSamples += sn; nLit += ln; ++nRays;

			uint16_t* o = cubeMap + 4 * texel;
			o[0] = f2h(scatter.x); o[1] = f2h(scatter.y); o[2] = f2h(scatter.z); o[3] = f2h(scatter.w);
		}
	}

	if (stats)
	{
		stats->view_samples += nSamples;
		stats->lit_samples += nLit;
		stats->rays += nRays;
	}
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// a7: K-buffer OIT composite -- VSCube.hlsl:51-76, PSDepthPeel.hlsl:12-24, PSCube.hlsl:30-60, PSCube.hlsli:21-108,
// RayCast.hlsli:43-108, PSResolveOIT.hlsl:12-26.
// The reference rasterises each visible volume's cube with front-face culling (MultiRayCaster.cpp:868, 897), no depth
// test (:869, :898): a pixel gets one fragment per volume, at the point where the pixel's view ray EXITS the cube.
// Here that fragment is found analytically (slab test) instead of by rasterisation; fragment depth = clip z/w of the
// exit point; fragments outside 0 <= z < 1 or with w <= 0 are clipped.  The 8 nearest fragments (PSDepthPeel) are
// shaded (CubeCast when MaskBits has CUBEMAP_RAYMARCH_BIT, else the screen-space RayCast), stored as RGBA16F
// (kColors format, MultiRayCaster.cpp:236) and resolved front to back.
// TextureCube Gather across a face edge takes the texel the out-of-face texel centre re-projects to (D3D leaves
// edge/corner filtering implementation-defined: unpinned).
// ---------------------------------------------------------------------------------------------
namespace
{
struct CubeCoord { uint32_t face; float s, t; };

inline CubeCoord cubeFaceCoords(float3 d)	// D3D cube-map face selection, consistent with GetLocalPos (CSRayMarch.hlsl:36-52)
{
	const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
	CubeCoord c;
	float sc, tc, ma;
	if (ax >= ay && ax >= az) { c.face = d.x > 0.0f ? 0u : 1u; sc = d.x > 0.0f ? -d.z : d.z; tc = -d.y; ma = ax; }
	else if (ay >= az) { c.face = d.y > 0.0f ? 2u : 3u; sc = d.x; tc = d.y > 0.0f ? d.z : -d.z; ma = ay; }
	else { c.face = d.z > 0.0f ? 4u : 5u; sc = d.z > 0.0f ? d.x : -d.x; tc = -d.y; ma = az; }
	c.s = (sc / ma + 1.0f) * 0.5f;
	c.t = (tc / ma + 1.0f) * 0.5f;
	return c;
}

inline size_t cubeTexelIndex(uint32_t face, int32_t i, int32_t j, int32_t d)
{
	if (i < 0 || j < 0 || i >= d || j >= d)
	{
		const float a = ((float)i + 0.5f) / (float)d * 2.0f - 1.0f, b = ((float)j + 0.5f) / (float)d * 2.0f - 1.0f;
		float3 dir;
		switch (face)
		{
		case 0: dir = f3(1.0f, -b, -a); break;
		case 1: dir = f3(-1.0f, -b, a); break;
		case 2: dir = f3(a, 1.0f, b); break;
		case 3: dir = f3(a, -1.0f, -b); break;
		case 4: dir = f3(a, -b, 1.0f); break;
		default: dir = f3(-a, -b, -1.0f); break;
		}
		const CubeCoord c = cubeFaceCoords(dir);
		face = c.face;
		i = clampi(f2i_floor(c.s * (float)d), 0, d - 1);
		j = clampi(f2i_floor(c.t * (float)d), 0, d - 1);
	}
	return ((size_t)face * d + (size_t)j) * d + (size_t)i;
}

inline float UnprojectZ(float depth)	// PSCube.hlsli:21-26
{
	return (kZNear * kZFar) / (depth * (kZNear - kZFar) + kZFar);
}

// CubeCast (PSCube.hlsli:51-108) + GetDomain (:31-46)
// The four texels a TextureCube Gather / bilinear SampleLevel at direction `dir` touches on a d x d cube map, in Gather order
// ((0,1), (1,1), (1,0), (0,0)), and the bilinear weights of the same footprint: the sampler contract shared with oracle/dxil.
void cubeFootprint(float3 dir, int32_t d, size_t idx[4], float wl[4])
{
	const CubeCoord cc = cubeFaceCoords(dir);
	const float gridSize = (float)d;
	const float u = cc.s * gridSize - 0.5f, v = cc.t * gridSize - 0.5f;
	const int32_t i0 = f2i_floor(u), j0 = f2i_floor(v);
	const float fx = u - floorf(u), fy = v - floorf(v);
	idx[0] = cubeTexelIndex(cc.face, i0, j0 + 1, d); idx[1] = cubeTexelIndex(cc.face, i0 + 1, j0 + 1, d);
	idx[2] = cubeTexelIndex(cc.face, i0 + 1, j0, d); idx[3] = cubeTexelIndex(cc.face, i0, j0, d);
	wl[0] = (1.0f - fx) * fy; wl[1] = fx * fy; wl[2] = fx * (1.0f - fy); wl[3] = (1.0f - fx) * (1.0f - fy);
}

float4 CubeCast(const uint16_t* cubeMap, const float* cubeDepth, int32_t d, float depthPix, float3 pos, float3 rayDir)
{
	const CubeCoord cc = cubeFaceCoords(pos);
	const float gridSize = (float)d;

	// gather footprint
	size_t idx[4];
	float wl[4];
	cubeFootprint(pos, d, idx, wl);
	float smp[4][4], z[4];
	for (int k = 0; k < 4; ++k)
	{
		for (int ch = 0; ch < 4; ++ch) smp[k][ch] = h2f(cubeMap[4 * idx[k] + ch]);
		z[k] = cubeDepth[idx[k]];
	}

	// SampleLevel fallback colour: plain bilinear over the same footprint
	float color[4];
	for (int ch = 0; ch < 4; ++ch) color[ch] = smp[0][ch] * wl[0] + smp[1][ch] * wl[1] + smp[2][ch] * wl[2] + smp[3][ch] * wl[3];

	// GetDomain
	float uvx = cc.s * gridSize, uvy = cc.t * gridSize;
	float domx = (uvx + 0.5f) - floorf(uvx + 0.5f), domy = (uvy + 0.5f) - floorf(uvy + 0.5f);
	const float bound = gridSize - 1.0f;
	const float3 axes = f3(pos.x * gridSize, pos.y * gridSize, pos.z * gridSize);
	if ((fabsf(axes.x) > bound && axes.x * rayDir.x < 0.0f) || (fabsf(axes.y) > bound && axes.y * rayDir.y < 0.0f) || (fabsf(axes.z) > bound && axes.z * rayDir.z < 0.0f))
	{
		uvx = fminf(uvx, gridSize - 0.5f); uvy = fminf(uvy, gridSize - 0.5f);
		domx = uvx < 0.5f ? 1.0f : 0.0f; domy = uvy < 0.5f ? 1.0f : 0.0f;
	}
	const float dix = 1.0f - domx, diy = 1.0f - domy;
	const float wb[4] = { dix * domy, domx * domy, domx * diy, dix * diy };

	const float depth = UnprojectZ(depthPix);
	float result[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
	float ws = 0.0f;
	for (int k = 0; k < 4; ++k)
	{
		const float zi = UnprojectZ(z[k]);
		float w = fmaxf(1.0f - 0.5f * fabsf(depth - zi), 0.0f);
		w *= wb[k];
		for (int ch = 0; ch < 4; ++ch) result[ch] += smp[k][ch] * w;
		ws += w;
	}
	float4 r;
	if (ws > 0.0f) { r.x = result[0] / ws; r.y = result[1] / ws; r.z = result[2] / ws; r.w = result[3] / ws; }
	else { r.x = color[0]; r.y = color[1]; r.z = color[2]; r.w = color[3]; }
	return r;
}

// The fragment the reference's rasteriser produces for one visible cube at one pixel (VSCube.hlsl:51-76 + front-face culling, no depth
// test: MultiRayCaster.cpp:868-869, 897-898): where the pixel's view ray leaves the cube.  Found analytically (slab test in volume
// space) instead of by rasterisation.  Returns false when there is none; lpt = the exit point in volume space (VSCube's LPt,
// interpolated), z = its depth (SV_Position.z).
bool exitFragment(const PerObject& po, float3 eyePt, float3 dirW, float3& lpt, float& z)
{
	const float3 o = mul43(eyePt, po.worldI);
	const float3 d = mul33(dirW, po.worldI);	// the pixel's world-space ray direction in volume space

	// slab test from the eye along d
	float tNear = -kFltMax, tFar = kFltMax;
	int farAxis = -1;
	bool miss = false;
	for (int a = 0; a < 3; ++a)
	{
		const float da = idx(d, a), oa = idx(o, a);
		if (da == 0.0f) { if (fabsf(oa) > 1.0f) miss = true; continue; }
		const float sgn = da > 0.0f ? 1.0f : -1.0f;
		const float tn = (-sgn - oa) / da, tf = (sgn - oa) / da;
		if (tn > tNear) tNear = tn;
		if (tf < tFar) { tFar = tf; farAxis = a; }
	}
	if (miss || farAxis < 0 || !(tFar > 0.0f) || !(tNear <= tFar)) return false;

	lpt = f3(fmaf(d.x, tFar, o.x), fmaf(d.y, tFar, o.y), fmaf(d.z, tFar, o.z));
	lpt = f3(clampf(lpt.x, -1.0f, 1.0f), clampf(lpt.y, -1.0f, 1.0f), clampf(lpt.z, -1.0f, 1.0f));
	const float snap = idx(d, farAxis) > 0.0f ? 1.0f : -1.0f;
	if (farAxis == 0) lpt.x = snap; else if (farAxis == 1) lpt.y = snap; else lpt.z = snap;

	float4 lp = { lpt.x, lpt.y, lpt.z, 1.0f };
	const float4 clip = mul44(lp, po.wvp);
	if (!(clip.w > 0.0f)) return false;
	z = clip.z / clip.w;
	return z >= 0.0f && z < 1.0f;
}

// the pixel's view ray in world space: unproject its near and far points with g_screenToWorld (Common.hlsli:43)
float3 pixelRayWorld(const CBPerFrame& cb, float nx, float ny)
{
	float4 c0 = { nx, ny, 0.0f, 1.0f }, c1 = { nx, ny, 1.0f, 1.0f };
	const float4 h0 = mul44(c0, cb.screenToWorld), h1 = mul44(c1, cb.screenToWorld);
	return f3(h1.x / h1.w, h1.y / h1.w, h1.z / h1.w) - f3(h0.x / h0.w, h0.y / h0.w, h0.z / h0.w);
}

struct Fragment { uint32_t zbits; uint32_t volume; float3 lpt; };
} // namespace

extern "C" {
float mvo_unproject_z(float depth) { return UnprojectZ(depth); }

void mvo_cube_footprint(const float dir[3], uint32_t d, uint64_t idx[4], float weights[4])
{
	size_t i[4];
	cubeFootprint(f3(dir[0], dir[1], dir[2]), (int32_t)d, i, weights);
	for (int k = 0; k < 4; ++k) idx[k] = i[k];
}

// every fragment of pixel (px, py): volume ids (in list order), exit points and depths; returns their number (<= visibleCount)
uint32_t mvo_pixel_fragments(uint32_t W, uint32_t H, uint32_t px, uint32_t py, const void* perObjectBytes, const void* cbBytes,
	const uint32_t* visibleVolumes, uint32_t visibleCount, uint32_t* volume, float* lpt3, float* z)
{
	const PerObject* perObjects = (const PerObject*)perObjectBytes;
	CBPerFrame cb;
	memcpy(&cb, cbBytes, sizeof(cb));
	float nx = ((float)px + 0.5f) / (float)W, ny = ((float)py + 0.5f) / (float)H;
	nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
	const float3 dirW = pixelRayWorld(cb, nx, ny);
	uint32_t n = 0;
	for (uint32_t k = 0; k < visibleCount; ++k)
	{
		float3 l;
		float zz;
		if (!exitFragment(perObjects[visibleVolumes[k]], f3(cb.eyePos[0], cb.eyePos[1], cb.eyePos[2]), dirW, l, zz)) continue;
		volume[n] = visibleVolumes[k]; lpt3[3 * n] = l.x; lpt3[3 * n + 1] = l.y; lpt3[3 * n + 2] = l.z; z[n] = zz;
		++n;
	}
	return n;
}

// CubeCast on one cube map / cube depth of edge d (layout uint16[6][d][d][4] / float[6][d][d]) for the KATs
void mvo_cube_cast(const uint16_t* cubeMap, const float* cubeDepth, uint32_t d, float depthPix, const float pos[3], const float rayDir[3], float out[4])
{
	const float4 r = CubeCast(cubeMap, cubeDepth, (int32_t)d, depthPix, f3(pos[0], pos[1], pos[2]), f3(rayDir[0], rayDir[1], rayDir[2]));
	out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// face / (s, t) of a direction and the texel a Gather tap (possibly outside the face) resolves to
void mvo_cube_face_coords(const float dir[3], uint32_t* face, float st[2])
{
	const CubeCoord c = cubeFaceCoords(f3(dir[0], dir[1], dir[2]));
	*face = c.face; st[0] = c.s; st[1] = c.t;
}
uint64_t mvo_cube_texel_index(uint32_t face, int32_t i, int32_t j, uint32_t d) { return cubeTexelIndex(face, i, j, (int32_t)d); }

void mvo_set_options(int weightBits, int libRayMarch)
{
	g_weightBits = weightBits < 0 ? 0 : (weightBits > 23 ? 23 : weightBits);
	g_libRayMarch = libRayMarch != 0;
}
}

extern "C" void mvo_composite_over(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
	const uint16_t* const* volumes, const uint32_t* lightMaps, const float* depthMap, const void* perObjectBytes, const void* cbBytes,
	const uint16_t* volumeInfo, const uint32_t* visibleVolumes, uint32_t visibleCount,
	const uint16_t* cubeMaps, const float* cubeDepths, const uint16_t* background, uint16_t* outRGBA, mvo_stats* stats);

extern "C" void mvo_composite(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
	const uint16_t* const* volumes, const uint32_t* lightMaps, const float* depthMap, const void* perObjectBytes, const void* cbBytes,
	const uint16_t* volumeInfo, const uint32_t* visibleVolumes, uint32_t visibleCount,
	const uint16_t* cubeMaps, const float* cubeDepths, uint16_t* outRGBA, mvo_stats* stats)
{
	mvo_composite_over(V, N, L, W, H, y0, y1, volumes, lightMaps, depthMap, perObjectBytes, cbBytes, volumeInfo, visibleVolumes, visibleCount,
		cubeMaps, cubeDepths, nullptr, outRGBA, stats);
}

// background: W x H RGBA16F the resolve is drawn over (PREMULTIPLITED blend preset, MultiRayCaster.cpp:931), or NULL
extern "C" void mvo_composite_over(uint32_t V, uint32_t N, uint32_t L, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1,
	const uint16_t* const* volumes, const uint32_t* lightMaps, const float* depthMap, const void* perObjectBytes, const void* cbBytes,
	const uint16_t* volumeInfo, const uint32_t* visibleVolumes, uint32_t visibleCount,
	const uint16_t* cubeMaps, const float* cubeDepths, const uint16_t* background, uint16_t* outRGBA, mvo_stats* stats)
{
	(void)V;
	const PerObject* perObjects = (const PerObject*)perObjectBytes;
	CBPerFrame cb;
	memcpy(&cb, cbBytes, sizeof(cb));
	const float3 eyePt = f3(cb.eyePos[0], cb.eyePos[1], cb.eyePos[2]);
	const uint64_t chain = chainTexels(N);
	uint64_t nSamples = 0;
	if (y1 > H) y1 = H;

#pragma omp parallel for schedule(dynamic, 2) reduction(+ : nSamples)
	for (int64_t py = y0; py < (int64_t)y1; ++py)
		for (uint32_t px = 0; px < W; ++px)
		{
			// PSCube.hlsl:37-39
			float nx = ((float)px + 0.5f) / (float)W, ny = ((float)py + 0.5f) / (float)H;
			nx = nx * 2.0f - 1.0f; ny = ny * 2.0f - 1.0f; ny = -ny;
			const float depthPix = depthMap[(size_t)py * W + px];
			const float3 dirW = pixelRayWorld(cb, nx, ny);

			Fragment frags[MVO_NUM_OIT_LAYERS];
			uint32_t nFrags = 0;
			for (uint32_t k = 0; k < visibleCount; ++k)
			{
				const uint32_t v = visibleVolumes[k];
				const PerObject& po = perObjects[v];
				float3 lpt;
				float z;
				if (!exitFragment(po, eyePt, dirW, lpt, z)) continue;

				// PSDepthPeel.hlsl:12-24: keep the NUM_OIT_LAYERS smallest depths, ascending
				Fragment f; f.zbits = asuint(z); f.volume = v; f.lpt = lpt;
				uint32_t pos = nFrags;
				while (pos > 0 && frags[pos - 1].zbits > f.zbits) --pos;
				if (pos >= MVO_NUM_OIT_LAYERS) continue;
				const uint32_t last = nFrags < MVO_NUM_OIT_LAYERS ? nFrags : MVO_NUM_OIT_LAYERS - 1;
				for (uint32_t m = last; m > pos; --m) frags[m] = frags[m - 1];
				frags[pos] = f;
				if (nFrags < MVO_NUM_OIT_LAYERS) ++nFrags;
			}

			// PSCube.hlsl:30-60 per layer, PSResolveOIT.hlsl:12-26
			float4 result = { 0.0f, 0.0f, 0.0f, 0.0f };
			for (uint32_t k = 0; k < nFrags; ++k)
			{
				const uint32_t v = frags[k].volume;
				const PerObject& po = perObjects[v];
				const uint32_t mipLevel = volumeInfo[4 * v + 0], smpCount = volumeInfo[4 * v + 1];
				const uint32_t maskBits = volumeInfo[4 * v + 2], volTexId = volumeInfo[4 * v + 3];
				const float3 localSpaceEyePt = mul43(eyePt, po.worldI);
				const float3 rayDir = frags[k].lpt - localSpaceEyePt;
				float4 color;
				if (!(maskBits & kCubeMapRayMarchBit))
				{
					// RayCast (RayCast.hlsli:43-108)
					float3 ro = localSpaceEyePt;
					const float3 rd = normalize3(rayDir);
					if (!ComputeRayOrigin(ro, rd)) { color.x = color.y = color.z = color.w = 0.0f; }
					else
					{
						const float tMax = GetTMax(f3(nx, ny, depthPix), ro, rd, po.wvpI);
						uint64_t s = 0, l = 0;
						color = marchLoop(volumes[volTexId], N, lightMaps + (size_t)v * L * L * L, L, ro, rd, tMax, smpCount, s, l);
						nSamples += s;
					}
				}
				else
				{
					const int32_t d = (int32_t)mipDim(N, mipLevel);	// txCubeMap.GetDimensions (PSCube.hlsli:55)
					const size_t off = (size_t)v * chain + mipOffset(N, mipLevel);
					color = CubeCast(cubeMaps + 4 * off, cubeDepths + off, d, depthPix, frags[k].lpt, rayDir);
				}
				if (color.w > 0.0f && color.w <= 1.0f)
				{
					// kColors is RGBA16F
					const float cr = h2f(f2h(color.x)), cg = h2f(f2h(color.y)), cbb = h2f(f2h(color.z)), ca = h2f(f2h(color.w));
					const float it = 1.0f - result.w;
					result.x += cr * it; result.y += cg * it; result.z += cbb * it; result.w += ca * it;
				}
			}
			result.w = fminf(result.w, 0.9997f);
			if (background)
			{
				// dst = src + dst * (1 - src.a) on all four channels (premultiplied alpha), fp32, one rounding at the store
				const uint16_t* b = background + 4 * ((size_t)py * W + px);
				const float inv = 1.0f - result.w;
				result.x = result.x + h2f(b[0]) * inv; result.y = result.y + h2f(b[1]) * inv; result.z = result.z + h2f(b[2]) * inv; result.w = result.w + h2f(b[3]) * inv;
			}
			uint16_t* o = outRGBA + 4 * ((size_t)py * W + px);
			o[0] = f2h(result.x); o[1] = f2h(result.y); o[2] = f2h(result.z); o[3] = f2h(result.w);
		}
	if (stats) stats->view_samples += nSamples;
}

