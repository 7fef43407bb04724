// mv_device.hpp -- device-side building blocks of the ray-march path (gfx950 only).
//
// CDNA has no texture units, so everything the reference leaves to the D3D12 fixed-function sampler
// (MultiRayCaster.cpp:556-566: LINEAR_CLAMP, POINT_CLAMP, LINEAR_LESS_EQUAL) is written out here:
// texel-centre addressing, clamp, fp16 / 11-11-10 decode, filter weights, comparison PCF.
//
// Numeric contract (DESIGN.md "Numerics"): fp32 arithmetic, fp16/packed formats only in storage;
// compiled with -ffp-contract=off, a multiply-add is fused only where __builtin_fmaf is written
// (the sampler's lerps); HLSL min/max = fminf/fmaxf (NaN-ignoring); normalize = v * (1 / sqrt(dot)); the order of operations and the
// half-folded constants follow the reference's compiled shaders.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv
{
// min16float constants: the reference's compiler folds them to half precision (its shipped Bin/*.cso read through oracle/dxil), so
// every implementation starts from the half values whatever precision it then computes in
constexpr float kMaxDist = 3.46484375f;			// Sh/RayMarch.hlsli:17  half(2.0 * sqrt(3.0)) = 0x42EE
constexpr float kAbsorption = 0.7998046875f;		// Sh/RayMarch.hlsli:11  half(0.8) = 0x3A66
constexpr float kZeroThreshold = 0.010002136230468750f;	// Sh/RayMarch.hlsli:12  half(0.01) = 0x211F
constexpr float kInvTwoPi = 0.1591796875f;			// Sh/CSRayMarch.hlsl:155  half(1 / (2 PI)) = 0x3118: `/ (2 PI)` is compiled as a multiplication
constexpr float kPI = 3.1415926535897f;			// XUSG/Shaders/SHIrradiance.hlsli:5-7
constexpr float kFltMax = 3.402823466e+38f;		// Sh/Common.hlsli:9
constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kZNear = 1.0f, kZFar = 1000.0f;	// SharedConsts.h:9-10
constexpr uint32_t kNumCubeMip = 5;				// SharedConsts.h:6
constexpr uint32_t kNumOitLayers = 8;			// SharedConsts.h:7
constexpr uint32_t kCubeMapRayMarchBit = 1u << 15;	// Sh/Common.hlsli:11

struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };

__device__ __forceinline__ vec3 v3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ vec3 operator/(vec3 a, vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ float comp(const vec3& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ float dot3(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ vec3 normalize3(vec3 v) { const float r = 1.0f / __builtin_sqrtf(dot3(v, v)); return v3(v.x * r, v.y * r, v.z * r); }	// compiled as v * rsqrt(dot(v, v))
__device__ __forceinline__ float saturatef(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
__device__ __forceinline__ float lerpf(float a, float b, float w) { return __builtin_fmaf(w, b - a, a); }
__device__ __forceinline__ float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ vec3 rayAt(vec3 o, vec3 d, float t) { return v3(t * d.x + o.x, t * d.y + o.y, t * d.z + o.z); }	// compiled as a multiplication and an addition (the build has -ffp-contract=off)
__device__ __forceinline__ bool outsideBox(vec3 p) { return __builtin_fabsf(p.x) > 1.0f || __builtin_fabsf(p.y) > 1.0f || __builtin_fabsf(p.z) > 1.0f; }

// A pointer every lane of the wavefront holds alike, moved to scalar registers: the compiler cannot prove that what a wave-uniform index
// loaded is itself uniform, and keeps such a pointer in two vector registers through a loop.  (Uniform FLOATS the kernel would convert from
// scalar integers -- (float)N and the like -- come as kernel arguments instead, DeviceState.fN ..: __builtin_amdgcn_readfirstlane is
// folded away where the compiler can see the value is uniform, and the converted float then stays in a vector register all the same.)
template <typename T>
__device__ __forceinline__ T* uniformPtr(T* p)
{
	const uint64_t v = reinterpret_cast<uint64_t>(p);
	const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
	return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ uint32_t f2u_sat(float x)	// D3D ftou
{
	if (!(x > 0.0f)) return 0u;
	if (x >= 4294967296.0f) return 0xffffffffu;
	return (uint32_t)x;
}

// min(max(v, lo), hi): the form the compiler turns into one v_med3_i32 (the ?: chain became min + compare + select)
__device__ __forceinline__ int clampi(int v, int lo, int hi) { const int m = v > lo ? v : lo; return m < hi ? m : hi; }

// ---- structured/constant buffer layouts (Sh/Common.hlsli:28-49; MultiRayCaster.cpp:23-41, 341-351) ----
// float4x4 written transposed by the host and read column-major: M[r][c] = f[4*c + r]; float4x3 likewise, c < 3.
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

static_assert(sizeof(PerObject) == 224, "PerObject must stay 224 bytes (MultiRayCaster.cpp:35-41)");
static_assert(sizeof(CBPerFrame) == 212, "CBPerFrame must stay 212 bytes (MultiRayCaster.cpp:23-33)");

__device__ __forceinline__ vec4 mul44(vec4 p, const float* m)	// mul(float4, float4x4)
{
	vec4 r;
	r.x = p.x * m[0] + p.y * m[1] + p.z * m[2] + p.w * m[3];
	r.y = p.x * m[4] + p.y * m[5] + p.z * m[6] + p.w * m[7];
	r.z = p.x * m[8] + p.y * m[9] + p.z * m[10] + p.w * m[11];
	r.w = p.x * m[12] + p.y * m[13] + p.z * m[14] + p.w * m[15];
	return r;
}

__device__ __forceinline__ vec3 mul43(vec3 p, const float* m)	// mul(float4(p, 1), float4x3)
{
	return v3(p.x * m[0] + p.y * m[1] + p.z * m[2] + m[3],
		p.x * m[4] + p.y * m[5] + p.z * m[6] + m[7],
		p.x * m[8] + p.y * m[9] + p.z * m[10] + m[11]);
}

__device__ __forceinline__ vec3 mul33(vec3 p, const float* m)	// mul(float3, (float3x3)float4x3)
{
	return v3(p.x * m[0] + p.y * m[1] + p.z * m[2],
		p.x * m[4] + p.y * m[5] + p.z * m[6],
		p.x * m[8] + p.y * m[9] + p.z * m[10]);
}

// ---- formats ----
__device__ __forceinline__ float half_bits_to_float(uint32_t h)
{
	return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
}

__device__ __forceinline__ uint16_t float_to_half_bits(float f)	// RNE (v_cvt_f16_f32)
{
	return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// R11G11B10_FLOAT texel -> fp32: 5-bit exponent fields line up with fp16 after a left shift.
__device__ __forceinline__ vec3 unpack_r11g11b10f(uint32_t p)
{
	return v3(half_bits_to_float((p & 0x7ffu) << 4), half_bits_to_float(((p >> 11) & 0x7ffu) << 4), half_bits_to_float((p >> 22) << 5));
}

// fp32 magnitude bits -> unsigned small float with 5 exponent / MB mantissa bits, round to nearest even
template <int MB>
__device__ __forceinline__ uint32_t f32_to_small(uint32_t a)
{
	constexpr int SH = 23 - MB;
	constexpr uint32_t INF = 31u << MB;
	if (a > 0x7f800000u) return INF | (1u << (MB - 1));
	if (a == 0x7f800000u) return INF;
	const int e = (int)(a >> 23) - 127;
	uint32_t m = a & 0x7fffffu;
	if (e > 15) return INF;
	if (e >= -14)
	{
		uint32_t v = ((uint32_t)(e + 15) << MB) | (m >> SH);
		const uint32_t rem = m & ((1u << SH) - 1u), half = 1u << (SH - 1);
		if (rem > half || (rem == half && (v & 1u))) ++v;
		return v;
	}
	if (e < -14 - MB - 1) return 0u;
	m |= 0x800000u;
	const int shift = SH + (-14 - e);
	if (shift > 31) return 0u;
	uint32_t v = m >> shift;
	const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
	if (rem > half || (rem == half && (v & 1u))) ++v;
	return v;
}

__device__ __forceinline__ uint32_t pack_r11g11b10f(float r, float g, float b)	// negative -> 0, RNE
{
	const uint32_t ur = __float_as_uint(r), ug = __float_as_uint(g), ub = __float_as_uint(b);
	const uint32_t R = ((ur & 0x80000000u) && (ur & 0x7fffffffu) <= 0x7f800000u) ? 0u : f32_to_small<6>(ur & 0x7fffffffu);
	const uint32_t G = ((ug & 0x80000000u) && (ug & 0x7fffffffu) <= 0x7f800000u) ? 0u : f32_to_small<6>(ug & 0x7fffffffu);
	const uint32_t B = ((ub & 0x80000000u) && (ub & 0x7fffffffu) <= 0x7f800000u) ? 0u : f32_to_small<5>(ub & 0x7fffffffu);
	return (R & 0x7ffu) | ((G & 0x7ffu) << 11) | ((B & 0x3ffu) << 22);
}

// ---- sampler addressing: LINEAR + CLAMP on one axis ----
struct Taps { int i0, i1; float f; };

// (fSize = (float)size, fSizeP1 = (float)size + 1: callers inside a loop hand them over from scalar registers: DeviceState.fN ..)
__device__ __forceinline__ Taps linearTaps(float coord01, uint32_t size, int off, float fSize, float fSizeP1)
{
	const float c = coord01 * fSize - 0.5f;
	const float fl = __builtin_floorf(c);
	Taps t;
	t.f = c - fl;
	// Clamp in float first (NaN -> -2) so the int conversion is always in range; after the integer clamp to
	// [0, size-1] this equals a saturating floor-to-int for every finite coordinate (|off| <= 1).
	const float flc = __builtin_fminf(__builtin_fmaxf(fl, -2.0f), fSizeP1);
	const int i = (int)flc + off;
	t.i0 = clampi(i, 0, (int)size - 1);
	t.i1 = clampi(i + 1, 0, (int)size - 1);
	return t;
}
__device__ __forceinline__ Taps linearTaps(float coord01, uint32_t size, int off) { return linearTaps(coord01, size, off, (float)size, (float)size + 1.0f); }

// ---- volume texel storage -------------------------------------------------------------------
// Volumes live in HBM as 4x4x4-texel bricks of RGBA16F (64 texels x 8 B = 512 B = four 128-B lines), bricks in
// x-fastest order.  A trilinear footprint (2x2x2 texels) then touches 1 brick 42 % of the time and at most 8,
// instead of always 4 rows that are N*8 bytes (and N*N*8 bytes) apart in a linear layout.
// Texel (x,y,z) -> element index (units of one RGBA16F texel):
// Brick pitches are padded (row of bricks + 1, plane + 17 bricks): with power-of-two pitches the bricks one x-slab of a
// march front touches are exactly 512 * bricks bytes apart and alias onto a few HBM channels (measured: a random tile
// order ran 11 % faster than list order before the padding).
#ifndef MV_BRICK_PAD
#define MV_BRICK_PAD 1
#endif
__host__ __device__ __forceinline__ uint32_t brickPitchY(uint32_t bricks) { return MV_BRICK_PAD ? bricks + 1u : bricks; }
__host__ __device__ __forceinline__ uint32_t brickPitchZ(uint32_t bricks) { return MV_BRICK_PAD ? bricks * (bricks + 1u) + 17u : bricks * bricks; }
__host__ __device__ __forceinline__ size_t volBrickCount(uint32_t bricks) { return (size_t)bricks * brickPitchZ(bricks); }	// allocation, in bricks
__host__ __device__ __forceinline__ size_t volBrickIndex(uint32_t bx, uint32_t by, uint32_t bz, uint32_t bricks)
{
	return (size_t)bz * brickPitchZ(bricks) + (size_t)by * brickPitchY(bricks) + bx;
}
__host__ __device__ __forceinline__ size_t volTexelIndex(uint32_t x, uint32_t y, uint32_t z, uint32_t bricksPerAxis)
{
	return volBrickIndex(x >> 2, y >> 2, z >> 2, bricksPerAxis) * 64 + ((z & 3u) << 4) + ((y & 3u) << 2) + (x & 3u);
}

// Global-address-space views: loads through them are global_load (SGPR base + 32-bit VGPR byte offset), not flat_load.
typedef const __attribute__((address_space(1))) uint8_t* GlobalBytes;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x2* GlobalUint2;
typedef const __attribute__((address_space(1))) u32x4* GlobalUint4;
typedef const __attribute__((address_space(1))) uint32_t* GlobalUint;
typedef const __attribute__((address_space(1))) uint64_t* GlobalUint64;

// Besides the RGBA16F bricks every volume source has an ALPHA-CELL companion: for every texel (x,y,z) the eight
// alphas of the 2x2x2 footprint that starts there, a(x..x+1, y..y+1, z..z+1) with the +1 neighbours clamped, as 8 halfs
// = one 16-byte aligned cell, cells bricked exactly like the texels (so cell offset = 2 x texel offset).  A density
// sample (CastLightRay, the alpha test of the view march, the gradient taps) is then ONE aligned 16-byte load instead of
// eight 2-byte gathers: these kernels are bound by L1 tag lookups (one per distinct line per load instruction), and
// 288 GB of HBM make the 8x replication of a 2-byte channel (16 B/texel next to the 8 B/texel RGBA) affordable.
struct VolumeView
{
	GlobalBytes base;		// bricked RGBA16F texels (8 bytes each)
	GlobalBytes cells;		// bricked alpha cells (16 bytes each); both stay < 4 GiB (mv_create enforces N <= 640)
	uint32_t N;				// texels per axis
	uint32_t strideY;		// texel bytes between bricks along y = bricks * 512
	uint32_t strideZ;		// texel bytes between bricks along z = bricks^2 * 512
};

__device__ __forceinline__ VolumeView makeVolumeView(const uint2* texels, const uint4* cells, uint32_t N, uint32_t bricks)
{
	VolumeView v;
	v.base = (GlobalBytes)texels;
	v.cells = (GlobalBytes)cells;
	v.N = N;
	v.strideY = brickPitchY(bricks) * 512u;
	v.strideZ = brickPitchZ(bricks) * 512u;
	return v;
}

// texel byte offsets contributed by one axis: brick part + in-brick part
__device__ __forceinline__ uint32_t offX(int i) { return (((uint32_t)i >> 2) << 9) + (((uint32_t)i & 3u) << 3); }
// 24-bit multiplies (v_mul_u32_u24, full rate; v_mul_lo_u32 is quarter rate): brick indices are < 2^8 and the strides < 2^24
// for every N <= 640 (pitchZ * 512 = 13.2 M at N = 640), which mv_create enforces.
__device__ __forceinline__ uint32_t offY(int i, uint32_t strideY) { return __umul24((uint32_t)i >> 2, strideY) + (((uint32_t)i & 3u) << 5); }
__device__ __forceinline__ uint32_t offZ(int i, uint32_t strideZ) { return __umul24((uint32_t)i >> 2, strideZ) + (((uint32_t)i & 3u) << 7); }

__device__ __forceinline__ uint2 loadTexelAt(const VolumeView& v, uint32_t byteOff)
{
	const u32x2 t = *(GlobalUint2)(v.base + byteOff);	// global_load_dwordx2, SGPR base + VGPR offset
	return make_uint2(t.x, t.y);
}

__device__ __forceinline__ vec4 decodeTexel(uint2 t)
{
	vec4 c;
	c.x = half_bits_to_float(t.x & 0xffffu);
	c.y = half_bits_to_float(t.x >> 16);
	c.z = half_bits_to_float(t.y & 0xffffu);
	c.w = half_bits_to_float(t.y >> 16);
	return c;
}

__device__ __forceinline__ float trilerp(float c000, float c100, float c010, float c110, float c001, float c101, float c011, float c111, float fx, float fy, float fz)
{
	return lerpf(lerpf(lerpf(c000, c100, fx), lerpf(c010, c110, fx), fy), lerpf(lerpf(c001, c101, fx), lerpf(c011, c111, fx), fy), fz);
}

// Footprint addressing shared by the density and the colour fetch of one sample.
struct FootprintAddr
{
	uint32_t x0, x1, r00, r10, r01, r11;	// texel byte offsets: x taps and the four (y, z) rows
	float fx, fy, fz;						// filter weights
	float cx, cy, cz;						// weights for a CELL fetch: 0 where the two taps of an axis coincide (the 16-byte and pair forms of fetchAlphaCell; experiments)
	uint32_t mx, mz;						// 0 where the two taps of the x / z axis coincide, all ones otherwise (the quad form of fetchAlphaCell: it SELECTS the taps)
	uint32_t ix, iy, iz;					// first tap of every axis (the footprint is {i, min(i + 1, N - 1)}^3)
};

__device__ __forceinline__ FootprintAddr footprintAddr(uint32_t N, uint32_t strideY, uint32_t strideZ, vec3 uvw, int ox = 0, int oy = 0, int oz = 0)
{
	const Taps tx = linearTaps(uvw.x, N, ox), ty = linearTaps(uvw.y, N, oy), tz = linearTaps(uvw.z, N, oz);
	FootprintAddr a;
	a.x0 = offX(tx.i0); a.x1 = offX(tx.i1);
	const uint32_t y0 = offY(ty.i0, strideY), y1 = offY(ty.i1, strideY);
	const uint32_t z0 = offZ(tz.i0, strideZ), z1 = offZ(tz.i1, strideZ);
	a.r00 = y0 + z0; a.r10 = y1 + z0; a.r01 = y0 + z1; a.r11 = y1 + z1;
	a.fx = tx.f; a.fy = ty.f; a.fz = tz.f;
	a.ix = (uint32_t)tx.i0; a.iy = (uint32_t)ty.i0; a.iz = (uint32_t)tz.i0;
	// A cell holds the clamped +1 neighbours of its own texel.  They are the sampler's second taps except when both
	// taps of an axis are the SAME texel (coordinate below the first texel centre: taps (0,0) but the cell holds (0,1)).
	// Rounds 1-5 zeroed the weight there (lerp(a, a, f) == a == lerp(a, b, 0)) -- true for FINITE b only: with b = inf or NaN
	// 0 x (b - a) is NaN where the sampler, which reads a twice, returns a (round 6: found by the sweep with special halfs in
	// the volumes, tests/test_random_scenes.py).  The quad form (the product's) therefore replaces the second taps by the
	// first where mx / mz say so and keeps the true weights; cx / cy / cz remain for the other forms.
	a.cx = tx.i1 == tx.i0 ? 0.0f : tx.f; a.cy = ty.i1 == ty.i0 ? 0.0f : ty.f; a.cz = tz.i1 == tz.i0 ? 0.0f : tz.f;
	a.mx = tx.i1 == tx.i0 ? 0u : 0xffffffffu; a.mz = tz.i1 == tz.i0 ? 0u : 0xffffffffu;
	return a;
}

__device__ __forceinline__ FootprintAddr footprintAddr(const VolumeView& v, vec3 uvw, int ox = 0, int oy = 0, int oz = 0)
{
	return footprintAddr(v.N, v.strideY, v.strideZ, uvw, ox, oy, oz);
}

// ---- address tables in LDS ----------------------------------------------------------------------
// What a texture unit's address path does per tap -- clamp, split into brick and in-brick index, multiply by the pitches -- depends
// only on ONE integer per axis, so a kernel can tabulate it: entry j of an axis holds, for the first tap i0 = max(j - 1, 0) and the
// second tap i1 = min(j, N - 1) (j = floor(coordinate) + 1 clamped to [0, N]: clamp addressing is in the table),
//   .x / .y = the byte offsets the two taps contribute to a texel address,
//   .z      = what i0 contributes to the brick-occupancy lookup: word index in bits 0-23, bit index in bits 24-29 (the three axes'
//             contributions simply add up: their fields cannot carry into each other),
//   .w      = 0 where both taps are the same texel, all ones otherwise (the mask of the weight for cell fetches).
// 3 (N + 1) entries of 16 bytes per workgroup: 12 KiB at N = 256.  One ds_read_b128 per axis replaces about twenty VALU
// instructions of clamping, shifting and multiplying, in a kernel that issues about 230 VALU instructions per wave-sample.
__device__ __forceinline__ u32x4 axisTableEntry(uint32_t axis, uint32_t j, uint32_t N, uint32_t strideY, uint32_t strideZ, uint32_t regions)
{
	const uint32_t i0 = j ? j - 1u : 0u, i1 = j < N - 1u ? j : N - 1u;
	u32x4 e;
	if (axis == 0u) { e.x = offX((int)i0); e.y = offX((int)i1); e.z = (i0 >> 4) | (((i0 >> 2) & 3u) << 24); }
	else if (axis == 1u) { e.x = offY((int)i0, strideY); e.y = offY((int)i1, strideY); e.z = ((i0 >> 4) * regions) | (((i0 >> 2) & 3u) << 26); }
	else { e.x = offZ((int)i0, strideZ); e.y = offZ((int)i1, strideZ); e.z = ((i0 >> 4) * regions * regions) | (((i0 >> 2) & 3u) << 28); }
	e.w = i1 == i0 ? 0u : 0xffffffffu;
	return e;
}

// index of a coordinate's table entry and its filter weight (LINEAR sampler: c = u * size - 0.5)
// (fSize = (float)size, fSizeM1 = (float)size - 1)
__device__ __forceinline__ uint32_t axisTableIndex(float coord01, float fSize, float fSizeM1, float& frac)
{
	const float c = coord01 * fSize - 0.5f;
	const float fl = __builtin_floorf(c);
	frac = c - fl;
	// float clamp first (NaN -> -1), so that the conversion is in range: floor + 1 in [0, size]
	return (uint32_t)((int)__builtin_fminf(__builtin_fmaxf(fl, -1.0f), fSizeM1) + 1);
}
__device__ __forceinline__ uint32_t axisTableIndex(float coord01, uint32_t size, float& frac) { return axisTableIndex(coord01, (float)size, (float)size - 1.0f, frac); }

// ---- brick occupancy ---------------------------------------------------------------------------
// One bit per 4x4x4 brick: set when ANY texel of the brick or of its +1 halo (the texels a trilinear footprint whose first
// taps lie in the brick can read) has alpha above kOccupancyLimit.  A clear bit proves that the interpolated alpha of such a
// sample cannot exceed ZERO_THRESHOLD (a convex combination of values <= limit, evaluated with three roundings of relative
// size 6e-8, against a limit 1e-3 below the threshold: the largest fp16 below 0.01 is 0.0099945), i.e. that the reference's
// loop does nothing for it but advance t -- so the eight texel loads, and the HBM lines behind them, are skipped.
// 64 bricks (a 16^3-texel region) share one 64-bit word: bit = bx & 3 | (by & 3) << 2 | (bz & 3) << 4.
constexpr float kOccupancyLimit = 0.00999f;
__host__ __device__ inline uint32_t occupancyRegions(uint32_t bricks) { return (bricks + 3u) >> 2; }
__device__ __forceinline__ uint32_t occupancyWord(uint32_t ix, uint32_t iy, uint32_t iz, uint32_t regions)
{
	return __umul24(__umul24(iz >> 4, regions) + (iy >> 4), regions) + (ix >> 4);
}
__device__ __forceinline__ uint32_t occupancyBit(uint32_t ix, uint32_t iy, uint32_t iz)
{
	return ((ix >> 2) & 3u) | (((iy >> 2) & 3u) << 2) | (((iz >> 2) & 3u) << 4);
}

// density of the footprint: one aligned 16-byte load of the alpha cell at (x0, y0, z0)
#ifndef MV_ALPHA_QUADS
#define MV_ALPHA_QUADS 1
#endif
#ifndef MV_EXACT_COINCIDENT_TAPS
#define MV_EXACT_COINCIDENT_TAPS 2	// 1: select the taps in every sample; 2: behind a scalar branch on the wavefront's ballot (the default: measured view march 0.2448 -> 0.2411-0.2426 ms alone, frame 0.318-0.33 -> 0.316; the results are the same)
#endif
#if MV_ALPHA_QUADS == 2
// Pair form: the companion holds 4 bytes per texel -- {a(x,y,z), a(x+1,y,z)}, +1 clamped -- in the texels' own brick layout
// (256 bytes per brick), and a footprint is the pairs of its four (y, z) rows: four aligned 4-byte loads, half the bytes of the quads.
struct AlphaFootprint { uint32_t p00, p10, p01, p11; float cx, cy, cz; };

__device__ __forceinline__ AlphaFootprint fetchAlphaCell(const VolumeView& v, const FootprintAddr& a)
{
	AlphaFootprint f;
	f.p00 = *(GlobalUint)(v.cells + ((a.r00 + a.x0) >> 1));
	f.p10 = *(GlobalUint)(v.cells + ((a.r10 + a.x0) >> 1));
	f.p01 = *(GlobalUint)(v.cells + ((a.r01 + a.x0) >> 1));
	f.p11 = *(GlobalUint)(v.cells + ((a.r11 + a.x0) >> 1));
	f.cx = a.cx; f.cy = a.fy; f.cz = a.fz;	// the y and z taps are the sampler's own: their own weights
	return f;
}

__device__ __forceinline__ float lerpRegisterHalves(uint32_t reg, float w)
{
	float d, r;
	asm("v_fma_mix_f32 %0, %1, 1.0, -%1 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(reg));
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(w), "v"(d), "v"(reg));
	return r;
}

__device__ __forceinline__ float resolveAlpha(const AlphaFootprint& f)
{
	const float x00 = lerpRegisterHalves(f.p00, f.cx), x10 = lerpRegisterHalves(f.p10, f.cx);
	const float x01 = lerpRegisterHalves(f.p01, f.cx), x11 = lerpRegisterHalves(f.p11, f.cx);
	return lerpf(lerpf(x00, x10, f.cy), lerpf(x01, x11, f.cy), f.cz);
}
#elif MV_ALPHA_QUADS
// Quad form (default): the companion holds 8 bytes per texel -- {a(x,y,z), a(x+1,y,z), a(x,y,z+1), a(x+1,y,z+1)}, +1 neighbours
// clamped -- in the texels' own brick layout, and a footprint is the quads of its two y rows (32 bytes apart inside one
// 128-byte line three times out of four).  Half the bytes of the 16-byte cell form behind every sample: the light march
// reads this structure almost completely every launch and is bound by HBM bandwidth (589 MB per launch with cells).
struct AlphaFootprint { u32x2 row0, row1; float cx, cy, cz; uint32_t mx, mz; };

__device__ __forceinline__ AlphaFootprint fetchAlphaCell(const VolumeView& v, const FootprintAddr& a)
{
	AlphaFootprint f;
	f.row0 = *(GlobalUint2)(v.cells + (a.r00 + a.x0));
	f.row1 = *(GlobalUint2)(v.cells + (a.r10 + a.x0));
	f.cx = a.fx; f.cy = a.fy; f.cz = a.fz;	// the sampler's own weights on every axis (the y taps are its own rows; x and z: see resolveAlpha)
	f.mx = a.mx; f.mz = a.mz;
	return f;
}

#ifndef MV_FMA_MIX
#define MV_FMA_MIX 1
#endif
#if MV_FMA_MIX
// lerp(lo, hi, w) of the two fp16 halves of ONE register through v_fma_mix_f32 (see lerpHalves below: same arithmetic as
// converting both and computing fma(w, hi - lo, lo), without the conversions as separate instructions)
__device__ __forceinline__ float lerpRegisterHalves(uint32_t reg, float w)
{
	float d, r;
	asm("v_fma_mix_f32 %0, %1, 1.0, -%1 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(reg));
	asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(w), "v"(d), "v"(reg));
	return r;
}
#endif

__device__ __forceinline__ float resolveAlpha(const AlphaFootprint& f)
{
	// Where the two x (z) taps are the same texel the sampler reads that texel twice; the quad holds its +1 neighbour instead: take the
	// first tap for both -- v_perm_b32 duplicates the low half of a word (selector 0x01000100; 0x03020100 leaves it alone), a select copies
	// the z words -- and lerp with the sampler's own weights, so that a NaN or an infinity in the neighbour stays out as it does in the
	// reference (lerp(a, a, f) is a, or NaN for an infinite a: whatever the arithmetic says, on the same operands).
	uint32_t r0x = f.row0.x, r1x = f.row1.x, r0y = f.row0.y, r1y = f.row1.y;
#if MV_EXACT_COINCIDENT_TAPS == 2
	if (__ballot(!f.mx || !f.mz))	// (only samples in the outermost half texel of the volume have such an axis: the other wavefronts skip this)
#endif
	{
		const uint32_t sel = (f.mx & 0x02020000u) | 0x01000100u;
		r0x = __builtin_amdgcn_perm(r0x, r0x, sel); r1x = __builtin_amdgcn_perm(r1x, r1x, sel);
		r0y = __builtin_amdgcn_perm(r0y, r0y, sel); r1y = __builtin_amdgcn_perm(r1y, r1y, sel);
		r0y = f.mz ? r0y : r0x; r1y = f.mz ? r1y : r1x;
	}
#if MV_FMA_MIX
	// x taps are the two halves of each quad word: {a000,a100 | a001,a101} in row0, {a010,a110 | a011,a111} in row1
	const float x00 = lerpRegisterHalves(r0x, f.cx), x10 = lerpRegisterHalves(r1x, f.cx);
	const float x01 = lerpRegisterHalves(r0y, f.cx), x11 = lerpRegisterHalves(r1y, f.cx);
	return lerpf(lerpf(x00, x10, f.cy), lerpf(x01, x11, f.cy), f.cz);
#else
	return trilerp(half_bits_to_float(r0x & 0xffffu), half_bits_to_float(r0x >> 16), half_bits_to_float(r1x & 0xffffu), half_bits_to_float(r1x >> 16),
		half_bits_to_float(r0y & 0xffffu), half_bits_to_float(r0y >> 16), half_bits_to_float(r1y & 0xffffu), half_bits_to_float(r1y >> 16),
		f.cx, f.cy, f.cz);
#endif
}
#else
struct AlphaFootprint { u32x4 cell; float cx, cy, cz; };

__device__ __forceinline__ AlphaFootprint fetchAlphaCell(const VolumeView& v, const FootprintAddr& a)
{
	AlphaFootprint f;
	f.cell = *(GlobalUint4)(v.cells + 2u * (a.r00 + a.x0));
	f.cx = a.cx; f.cy = a.cy; f.cz = a.cz;
	return f;
}

// cell halfs are ordered x fastest, then y, then z: {a000,a100 | a010,a110 | a001,a101 | a011,a111}
__device__ __forceinline__ float resolveAlpha(const AlphaFootprint& f)
{
	return trilerp(half_bits_to_float(f.cell.x & 0xffffu), half_bits_to_float(f.cell.x >> 16), half_bits_to_float(f.cell.y & 0xffffu), half_bits_to_float(f.cell.y >> 16),
		half_bits_to_float(f.cell.z & 0xffffu), half_bits_to_float(f.cell.z >> 16), half_bits_to_float(f.cell.w & 0xffffu), half_bits_to_float(f.cell.w >> 16),
		f.cx, f.cy, f.cz);
}
#endif

// Texture3D<float4>.SampleLevel(LINEAR_CLAMP, uvw, 0, offset).w -- Sh/RayMarch.hlsli:44-50, 74
__device__ __forceinline__ float sampleAlpha16F(const VolumeView& v, vec3 uvw, int ox = 0, int oy = 0, int oz = 0)
{
	return resolveAlpha(fetchAlphaCell(v, footprintAddr(v, uvw, ox, oy, oz)));
}

// colour of the footprint: the eight RGBA16F texels (only fetched for samples that pass the alpha test)
struct ColourFootprint { uint2 t000, t100, t010, t110, t001, t101, t011, t111; float fx, fy, fz; };

// `base` is either wave-uniform (SGPR base + VGPR byte offset: the offsets then carry the volume's offset inside the pool) or a
// per-lane pointer to the volume
__device__ __forceinline__ uint2 loadTexelFrom(GlobalBytes base, uint32_t byteOff)
{
	const u32x2 t = *(GlobalUint2)(base + byteOff);
	return make_uint2(t.x, t.y);
}

__device__ __forceinline__ ColourFootprint fetchColourAt(GlobalBytes base, const FootprintAddr& a)
{
	ColourFootprint f;
	f.t000 = loadTexelFrom(base, a.r00 + a.x0); f.t100 = loadTexelFrom(base, a.r00 + a.x1);
	f.t010 = loadTexelFrom(base, a.r10 + a.x0); f.t110 = loadTexelFrom(base, a.r10 + a.x1);
	f.t001 = loadTexelFrom(base, a.r01 + a.x0); f.t101 = loadTexelFrom(base, a.r01 + a.x1);
	f.t011 = loadTexelFrom(base, a.r11 + a.x0); f.t111 = loadTexelFrom(base, a.r11 + a.x1);
	f.fx = a.fx; f.fy = a.fy; f.fz = a.fz;
	return f;
}

__device__ __forceinline__ ColourFootprint fetchColour(const VolumeView& v, const FootprintAddr& a) { return fetchColourAt(v.base, a); }

#ifndef MV_FMA_MIX
#define MV_FMA_MIX 1
#endif
#if MV_FMA_MIX
// lerp(a, b, w) = fma(w, b - a, a) with a, b fp16 halves of 32-bit registers, through v_fma_mix_f32: the conversions ride on
// the two FMAs instead of being instructions of their own.  b - a is fma(b, 1.0, -a): b * 1.0 is exact, so the one rounding is
// that of the fp32 subtraction the plain form performs; the second FMA is the same operation with a converted on the fly.
// HI selects the upper half of the register that holds the texel channel.
template <int HI_A, int HI_B>
__device__ __forceinline__ float lerpHalves(uint32_t regA, uint32_t regB, float w)
{
	float d, r;
	if constexpr (HI_A == 0 && HI_B == 0) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(regB), "v"(regA));
	else if constexpr (HI_A == 1 && HI_B == 1) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(regB), "v"(regA));
	else static_assert(HI_A == HI_B, "both halves come from the same channel position");
	if constexpr (HI_A == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(w), "v"(d), "v"(regA));
	else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(w), "v"(d), "v"(regA));
	return r;
}
// channel C of the eight texels (0 = R: low half of .x, 1 = G: high half of .x, 2 = B: low half of .y, 3 = A: high half of .y)
template <int C>
__device__ __forceinline__ float trilerpChannel(const ColourFootprint& f)
{
	constexpr int HI = C & 1;
#define MV_REG(t) ((C >> 1) ? (t).y : (t).x)
	const float x00 = lerpHalves<HI, HI>(MV_REG(f.t000), MV_REG(f.t100), f.fx), x10 = lerpHalves<HI, HI>(MV_REG(f.t010), MV_REG(f.t110), f.fx);
	const float x01 = lerpHalves<HI, HI>(MV_REG(f.t001), MV_REG(f.t101), f.fx), x11 = lerpHalves<HI, HI>(MV_REG(f.t011), MV_REG(f.t111), f.fx);
#undef MV_REG
	return lerpf(lerpf(x00, x10, f.fy), lerpf(x01, x11, f.fy), f.fz);
}
#endif

__device__ __forceinline__ vec3 resolveRGB(const ColourFootprint& f)
{
	vec3 r;
#if MV_FMA_MIX
	r.x = trilerpChannel<0>(f); r.y = trilerpChannel<1>(f); r.z = trilerpChannel<2>(f);
	return r;
#endif
#define MV_R(t) half_bits_to_float((t).x & 0xffffu)
#define MV_G(t) half_bits_to_float((t).x >> 16)
#define MV_B(t) half_bits_to_float((t).y & 0xffffu)
	r.x = trilerp(MV_R(f.t000), MV_R(f.t100), MV_R(f.t010), MV_R(f.t110), MV_R(f.t001), MV_R(f.t101), MV_R(f.t011), MV_R(f.t111), f.fx, f.fy, f.fz);
	r.y = trilerp(MV_G(f.t000), MV_G(f.t100), MV_G(f.t010), MV_G(f.t110), MV_G(f.t001), MV_G(f.t101), MV_G(f.t011), MV_G(f.t111), f.fx, f.fy, f.fz);
	r.z = trilerp(MV_B(f.t000), MV_B(f.t100), MV_B(f.t010), MV_B(f.t110), MV_B(f.t001), MV_B(f.t101), MV_B(f.t011), MV_B(f.t111), f.fx, f.fy, f.fz);
#undef MV_R
#undef MV_G
#undef MV_B
	return r;
}

// alpha through the texels themselves (used by diagnostics and by the alpha-cell builder's self check)
__device__ __forceinline__ float resolveAlphaFromTexels(const ColourFootprint& f)
{
#if MV_FMA_MIX
	return trilerpChannel<3>(f);
#endif
#define MV_W(t) half_bits_to_float((t).y >> 16)
	return trilerp(MV_W(f.t000), MV_W(f.t100), MV_W(f.t010), MV_W(f.t110), MV_W(f.t001), MV_W(f.t101), MV_W(f.t011), MV_W(f.t111), f.fx, f.fy, f.fz);
#undef MV_W
}

// ---- light maps ---------------------------------------------------------------------------------
// Light maps are stored QUAD-replicated: element (x,y,z) is a 16-byte cell {t(x,y), t(x+1,y), t(x,y+1), t(x+1,y+1)} of
// slice z (+1 neighbours clamped), elements in plain [z][y][x] order.  Texture3D<float3>.SampleLevel(LINEAR_CLAMP)
// (GetLight, Sh/RayMarch.hlsli:235-240) is then two aligned 16-byte loads (slices z0 and z1) instead of eight 4-byte
// gathers.  The same coincident-tap rule as for alpha cells applies to x and y: where both taps of an axis are the same texel the
// cell's second member is replaced by its first (round 6; rounds 1-5 zeroed the weight, which assumed finite light-map texels -- an
// infinite alpha in a volume makes a light texel NaN, and a light map may be uploaded).
struct LightFootprint { u32x4 q0, q1; float cx, cy, fz; };

// `pool` + `mapOff`: the light map starts mapOff bytes into a (wave-uniform) pool; mapOff may differ from lane to lane
__device__ __forceinline__ LightFootprint fetchLightFootprintAt(GlobalBytes pool, uint32_t mapOff, uint32_t L, vec3 uvw, float fL, float fLp1)
{
	const Taps tx = linearTaps(uvw.x, L, 0, fL, fLp1), ty = linearTaps(uvw.y, L, 0, fL, fLp1), tz = linearTaps(uvw.z, L, 0, fL, fLp1);
	// 32-bit byte offsets and 24-bit multiplies: L <= 640 (mv_create), so L^2 * 16 < 2^24 and L^3 * 16 < 2^32
	const uint32_t row = ((__umul24((uint32_t)ty.i0, L) + (uint32_t)tx.i0) << 4) + mapOff;
	const uint32_t slice = (L * L) << 4;
	LightFootprint f;
	f.q0 = *(GlobalUint4)(pool + (__umul24((uint32_t)tz.i0, slice) + row));
	f.q1 = *(GlobalUint4)(pool + (__umul24((uint32_t)tz.i1, slice) + row));
	const bool sameX = tx.i1 == tx.i0, sameY = ty.i1 == ty.i0;
	// cell = {t(x,y), t(x+1,y), t(x,y+1), t(x+1,y+1)}: the sampler's taps where an axis' two taps coincide are its first members.
	// Only samples in the outermost half texel of the map have such an axis: a wavefront none of whose lanes has one skips the selects
	// (a scalar branch on the lanes' ballot).
#if MV_EXACT_COINCIDENT_TAPS == 2
	if (__ballot(sameX || sameY))
#endif
	{
		f.q0.y = sameX ? f.q0.x : f.q0.y; f.q0.w = sameX ? f.q0.z : f.q0.w; f.q1.y = sameX ? f.q1.x : f.q1.y; f.q1.w = sameX ? f.q1.z : f.q1.w;
		f.q0.z = sameY ? f.q0.x : f.q0.z; f.q0.w = sameY ? f.q0.y : f.q0.w; f.q1.z = sameY ? f.q1.x : f.q1.z; f.q1.w = sameY ? f.q1.y : f.q1.w;
	}
	f.cx = tx.f; f.cy = ty.f; f.fz = tz.f;
	return f;
}

__device__ __forceinline__ LightFootprint fetchLightFootprintAt(GlobalBytes pool, uint32_t mapOff, uint32_t L, vec3 uvw)
{
	return fetchLightFootprintAt(pool, mapOff, L, uvw, (float)L, (float)L + 1.0f);
}
__device__ __forceinline__ LightFootprint fetchLightFootprint(const uint4* quads, uint32_t L, vec3 uvw)
{
	return fetchLightFootprintAt((GlobalBytes)quads, 0u, L, uvw);
}
__device__ __forceinline__ LightFootprint fetchLightFootprint(const uint4* quads, uint32_t L, vec3 uvw, float fL, float fLp1)
{
	return fetchLightFootprintAt((GlobalBytes)quads, 0u, L, uvw, fL, fLp1);
}

#ifndef MV_LIGHT_FMA_MIX
#define MV_LIGHT_FMA_MIX 1	// the light footprint's x lerps on fp16 bits through v_fma_mix_f32 (0: convert every channel first, A/B)
#endif
__device__ __forceinline__ vec3 resolveLight(const LightFootprint& f)
{
#if MV_FMA_MIX && MV_LIGHT_FMA_MIX
	// An 11- / 10-bit float is an fp16 without its low mantissa bits (same exponent field, unpack_r11g11b10f above): R and G of a texel go,
	// as fp16 bit patterns, into the two halves of one register, B into the low half of another, and the four x lerps of a channel read
	// them through v_fma_mix_f32 exactly as the colour footprint's do (lerpHalves: the same two FMAs as lerpf, the conversion riding on
	// them) -- 24 conversions fewer per lit sample, the same bits.
#define MV_RG(p) ((((p) << 4) & 0x7ff0u) | (((p) << 9) & 0x7ff00000u))
#define MV_B(p) (((p) >> 17) & 0x7fe0u)
	const uint32_t rg000 = MV_RG(f.q0.x), rg100 = MV_RG(f.q0.y), rg010 = MV_RG(f.q0.z), rg110 = MV_RG(f.q0.w);
	const uint32_t rg001 = MV_RG(f.q1.x), rg101 = MV_RG(f.q1.y), rg011 = MV_RG(f.q1.z), rg111 = MV_RG(f.q1.w);
	const uint32_t b000 = MV_B(f.q0.x), b100 = MV_B(f.q0.y), b010 = MV_B(f.q0.z), b110 = MV_B(f.q0.w);
	const uint32_t b001 = MV_B(f.q1.x), b101 = MV_B(f.q1.y), b011 = MV_B(f.q1.z), b111 = MV_B(f.q1.w);
#undef MV_RG
#undef MV_B
	vec3 o;
	o.x = lerpf(lerpf(lerpHalves<0, 0>(rg000, rg100, f.cx), lerpHalves<0, 0>(rg010, rg110, f.cx), f.cy),
		lerpf(lerpHalves<0, 0>(rg001, rg101, f.cx), lerpHalves<0, 0>(rg011, rg111, f.cx), f.cy), f.fz);
	o.y = lerpf(lerpf(lerpHalves<1, 1>(rg000, rg100, f.cx), lerpHalves<1, 1>(rg010, rg110, f.cx), f.cy),
		lerpf(lerpHalves<1, 1>(rg001, rg101, f.cx), lerpHalves<1, 1>(rg011, rg111, f.cx), f.cy), f.fz);
	o.z = lerpf(lerpf(lerpHalves<0, 0>(b000, b100, f.cx), lerpHalves<0, 0>(b010, b110, f.cx), f.cy),
		lerpf(lerpHalves<0, 0>(b001, b101, f.cx), lerpHalves<0, 0>(b011, b111, f.cx), f.cy), f.fz);
	return o;
#endif
	const vec3 c000 = unpack_r11g11b10f(f.q0.x), c100 = unpack_r11g11b10f(f.q0.y), c010 = unpack_r11g11b10f(f.q0.z), c110 = unpack_r11g11b10f(f.q0.w);
	const vec3 c001 = unpack_r11g11b10f(f.q1.x), c101 = unpack_r11g11b10f(f.q1.y), c011 = unpack_r11g11b10f(f.q1.z), c111 = unpack_r11g11b10f(f.q1.w);
	vec3 r;
	r.x = trilerp(c000.x, c100.x, c010.x, c110.x, c001.x, c101.x, c011.x, c111.x, f.cx, f.cy, f.fz);
	r.y = trilerp(c000.y, c100.y, c010.y, c110.y, c001.y, c101.y, c011.y, c111.y, f.cx, f.cy, f.fz);
	r.z = trilerp(c000.z, c100.z, c010.z, c110.z, c001.z, c101.z, c011.z, c111.z, f.cx, f.cy, f.fz);
	return r;
}

__device__ __forceinline__ vec3 sampleR11G11B10F(const uint4* quads, uint32_t L, vec3 uvw)
{
	return resolveLight(fetchLightFootprint(quads, L, uvw));
}

// Texture2D<float>.SampleLevel(POINT_CLAMP, uv, 0) -- Sh/CSRayMarch.hlsl:68
__device__ __forceinline__ float samplePointR32F(const float* tex, uint32_t W, uint32_t H, float u, float v)
{
	// float clamp first (NaN -> 0), then the conversion is exact and in range
	const int x = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(u * (float)W), 0.0f), (float)(W - 1u));
	const int y = (int)__builtin_fminf(__builtin_fmaxf(__builtin_floorf(v * (float)H), 0.0f), (float)(H - 1u));
	return tex[(size_t)y * W + x];
}

// Texture2D<float>.SampleCmpLevelZero(LINEAR_LESS_EQUAL, uv, ref) on D16_UNORM -- Sh/RayMarch.hlsli:109
__device__ __forceinline__ float sampleCmpD16(const uint16_t* tex, uint32_t M, float u, float v, float ref)
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


// ---- Sh/RayMarch.hlsli device functions ----
__device__ __forceinline__ vec3 LocalToTex3DSpace(vec3 pos)	// :170-177, _TEXCOORD_INVERT_Y_ off
{
	return v3(pos.x * 0.5f + 0.5f, pos.y * 0.5f + 0.5f, pos.z * 0.5f + 0.5f);
}

// :128-155
__device__ __forceinline__ bool ComputeRayOrigin(vec3& rayOrigin, vec3 rayDir)
{
	if (__builtin_fabsf(rayOrigin.x) <= 1.0f && __builtin_fabsf(rayOrigin.y) <= 1.0f && __builtin_fabsf(rayOrigin.z) <= 1.0f) return true;

	float U = kFltMax;
	bool isHit = false;
#pragma unroll
	for (int i = 0; i < 3; ++i)
	{
		const int j = (i + 1) % 3, k = (i + 2) % 3;
		const float di = comp(rayDir, i), oi = comp(rayOrigin, i);
		const float u = (-signf(di) - oi) / di;
		const bool skip = (u < 0.0f)
			|| (__builtin_fabsf(comp(rayDir, j) * u + comp(rayOrigin, j)) > 1.0f)
			|| (__builtin_fabsf(comp(rayDir, k) * u + comp(rayOrigin, k)) > 1.0f);
		if (!skip && u < U)
		{
			U = u;
			isHit = true;
		}
	}
	rayOrigin = v3(clampf(rayDir.x * U + rayOrigin.x, -1.0f, 1.0f),
		clampf(rayDir.y * U + rayOrigin.y, -1.0f, 1.0f),
		clampf(rayDir.z * U + rayOrigin.z, -1.0f, 1.0f));
	return isHit;
}

// Conservative rejection in front of ComputeRayOrigin for the light march, where a voxel's ray is tested against every
// volume and misses nearly all of them: a ray whose LINE stays outside the sphere of radius^2 3.1 (the box's circumscribed
// sphere is radius^2 3, a 3 % margin, plus 1e-4 |o|^2 for the cancellation in |o|^2 - (o.d)^2 when the origin is far)
// cannot pass ComputeRayOrigin's tests, which accept only points within 1 + a few ulp of the box -- and an origin inside
// the box has |o|^2 <= 3.  rayDir is unit length (normalize3); NaNs compare false and fall through to the exact test.
__device__ __forceinline__ bool missesUnitBox(vec3 o, vec3 d)
{
	const float oo = o.x * o.x + o.y * o.y + o.z * o.z;
	const float od = o.x * d.x + o.y * d.y + o.z * d.z;
	return oo - od * od > 3.1f + 1e-4f * oo;
}

// :160-165
__device__ __forceinline__ float ComputeTargetHit(vec3 rayOrigin, vec3 target, vec3 rayDir)
{
	const vec3 u = (target - rayOrigin) / rayDir;
	return __builtin_fmaxf(__builtin_fmaxf(u.x, u.y), u.z);
}

// :82-97
__device__ __forceinline__ float GetTMax(vec3 pos, vec3 rayOrigin, vec3 rayDir, const float* worldViewProjI)
{
	if (pos.z >= 1.0f) return kFltMax;
	vec4 p; p.x = pos.x; p.y = pos.y; p.z = pos.z; p.w = 1.0f;
	const vec4 hpos = mul44(p, worldViewProjI);
	pos = v3(hpos.x / hpos.w, hpos.y / hpos.w, hpos.z / hpos.w);
	const vec3 t = (pos - rayOrigin) / rayDir;
	return __builtin_fmaxf(__builtin_fmaxf(t.x, t.y), t.z);
}

// factorEv = min(1/256 / |dDensity|, 2) (:186).  The numerator is a power of two, so the IEEE quotient is the correctly rounded
// reciprocal scaled exactly: v_rcp_f32 (1 ulp) + one fused Newton step gives it in 5 instructions instead of the compiler's 11 for
// a general division.  scripts/micro/rcp_exact.cpp compares the two forms over all 2^32 inputs on the GPU (profiles/r03_step_rcp.txt):
// identical for every |d| < 2^118, for +-0, +-inf and NaN (a density difference of half-float texels is at most 131008).
// The max(e, -1) only acts when e is NaN (d = 0: r0 = inf; d = inf: r0 = 0) and makes r1 = r0 - r0: NaN -> factor 2, or 0.
#ifndef MV_STEP_RCP
#define MV_STEP_RCP 1
#endif
__device__ __forceinline__ float StepFactorEv(float dDensity)
{
#if MV_STEP_RCP
	const float x = __builtin_fabsf(dDensity);
	const float r0 = __builtin_amdgcn_rcpf(x);
	const float e = __builtin_fmaxf(__builtin_fmaf(-x, r0, 1.0f), -1.0f);
	const float r1 = __builtin_fmaf(e, r0, r0);
	return __builtin_fminf(r1 * 0.00390625f, 2.0f);
#else
	return __builtin_fminf(1.0f / 256.0f / __builtin_fabsf(dDensity), 2.0f);
#endif
}

// :182-192
__device__ __forceinline__ float GetStep(float dDensity, float transm, float density, float step)
{
	const float factorEv = StepFactorEv(dDensity);
	const float factorUi = __builtin_fminf(1.0f - density, 1.0f);
	const float factorTh = 1.0f - transm;
	step *= __builtin_fmaxf(factorTh * 1.5f * factorEv * factorUi, 1.0f);	// the compiled order: ((factorTh * 1.5) * factorEv) * factorUi
	return step;
}

// The same inside the view march / RayCast loop, as the reference's compiler emitted it: factorTh = 1 - transm = 1 - (1 - scatter.w)
// was simplified to scatter.w (fast math), and the product is taken in the order ((scatter.w * 1.5) * factorEv) * factorUi.
__device__ __forceinline__ float GetViewStep(float dDensity, float scatterW, float density, float step)
{
	const float factorEv = StepFactorEv(dDensity);
	const float factorUi = __builtin_fminf(1.0f - density, 1.0f);
	return __builtin_fmaxf(scatterW * 1.5f * factorEv * factorUi, 1.0f) * step;
}

// Sh/CSRayMarch.hlsl:28-53
__device__ __forceinline__ vec3 GetLocalPos(float px, float py, uint32_t slice, float dims)
{
	const float x = (px + 0.5f) / dims * 2.0f - 1.0f;
	float y = (py + 0.5f) / dims * 2.0f - 1.0f;
	y = -y;
	switch (slice)
	{
	case 0: return v3(1.0f, y, -x);
	case 1: return v3(-1.0f, y, x);
	case 2: return v3(x, 1.0f, -y);
	case 3: return v3(x, -1.0f, y);
	case 4: return v3(x, y, 1.0f);
	case 5: return v3(-x, y, -1.0f);
	default: return v3(0.0f, 0.0f, 0.0f);
	}
}


// The fragment the reference's rasteriser produces for one volume at one pixel (VSCube.hlsl:51-76 with front-face
// culling, MultiRayCaster.cpp:868, 897): the point where the pixel's view ray EXITS the cube, found analytically.
// dirW = the pixel's world-space ray direction (from g_screenToWorld).  Returns false if the ray misses the cube or the
// fragment would be clipped (w <= 0 or z outside [0, 1)).  Used by the composite and by the fused screen-space RayCast.
__device__ __forceinline__ bool findExitFragment(const PerObject& po, vec3 eyeLocal, vec3 dirW, vec3& lpt, float& z)
{
	const vec3 d = mul33(dirW, po.worldI);
	float tNear = -kFltMax, tFar = kFltMax;
	int farAxis = -1;
	bool miss = false;
#pragma unroll
	for (int a = 0; a < 3; ++a)
	{
		const float da = comp(d, a), oa = comp(eyeLocal, a);
		if (da == 0.0f) { if (__builtin_fabsf(oa) > 1.0f) miss = true; continue; }
		const float sgn = da > 0.0f ? 1.0f : -1.0f;
		const float tn = (-sgn - oa) / da, tf = (sgn - oa) / da;
		if (tn > tNear) tNear = tn;
		if (tf < tFar) { tFar = tf; farAxis = a; }
	}
	if (miss || farAxis < 0 || !(tFar > 0.0f) || !(tNear <= tFar)) return false;

	// (the rasteriser's interpolated LPt in the reference: this stand-in keeps its fused form, like the oracle's)
	lpt = v3(__builtin_fmaf(d.x, tFar, eyeLocal.x), __builtin_fmaf(d.y, tFar, eyeLocal.y), __builtin_fmaf(d.z, tFar, eyeLocal.z));
	lpt = v3(clampf(lpt.x, -1.0f, 1.0f), clampf(lpt.y, -1.0f, 1.0f), clampf(lpt.z, -1.0f, 1.0f));
	const float snap = comp(d, farAxis) > 0.0f ? 1.0f : -1.0f;
	if (farAxis == 0) lpt.x = snap; else if (farAxis == 1) lpt.y = snap; else lpt.z = snap;

	vec4 lp; lp.x = lpt.x; lp.y = lpt.y; lp.z = lpt.z; lp.w = 1.0f;
	const vec4 clip = mul44(lp, po.wvp);
	if (!(clip.w > 0.0f)) return false;
	z = clip.z / clip.w;
	return z >= 0.0f && z < 1.0f;
}

// world-space direction of the view ray through NDC (nx, ny): unproject the near and far points with g_screenToWorld
__device__ __forceinline__ vec3 pixelRayWorld(const CBPerFrame* cb, float nx, float ny)
{
	vec4 c0; c0.x = nx; c0.y = ny; c0.z = 0.0f; c0.w = 1.0f;
	vec4 c1 = c0; c1.z = 1.0f;
	const vec4 h0 = mul44(c0, cb->screenToWorld), h1 = mul44(c1, cb->screenToWorld);
	return v3(h1.x / h1.w, h1.y / h1.w, h1.z / h1.w) - v3(h0.x / h0.w, h0.y / h0.w, h0.z / h0.w);
}

// Shared body of the view march (Sh/CSRayMarch.hlsl:111-155) and of the screen-space RayCast (Sh/RayCast.hlsli:56-104):
// front-to-back accumulation with the adaptive step of GetStep.  Returns scatter (rgb already divided by 2 pi).
// Per sample: one alpha-cell load decides :128; only lit samples fetch the 8 RGBA texels and the 2 light-map quads.
__device__ __forceinline__ vec4 marchRay(const VolumeView& vol, const uint4* lightQuads, uint32_t L, vec3 rayOrigin, vec3 rayDir,
	float tMax, uint32_t smpCount, uint32_t& nSamples, uint32_t& nLit)
{
	const float stepScale = kMaxDist / (float)smpCount;
	vec4 scatter; scatter.x = scatter.y = scatter.z = scatter.w = 0.0f;
	float t = 0.0f;
	float prevDensity = 0.0f;
	for (uint32_t i = 0; i < smpCount; ++i)
	{
		const vec3 pos = rayAt(rayOrigin, rayDir, t);
		if (outsideBox(pos)) break;
		const vec3 uvw = LocalToTex3DSpace(pos);

		const FootprintAddr fa = footprintAddr(vol, uvw);
		const float alpha = resolveAlpha(fetchAlphaCell(vol, fa));	// GetSample(...).w
		++nSamples;
		float newStep = stepScale;

		if (alpha > kZeroThreshold)
		{
			const LightFootprint lf = fetchLightFootprint(lightQuads, L, uvw);	// GetLight: same pos * 0.5 + 0.5
			const ColourFootprint cf = fetchColour(vol, fa);
			const vec3 light = resolveLight(lf);
			vec3 color = resolveRGB(cf);
			++nLit;

			const float transm = 1.0f - scatter.w;
			const float dDensity = alpha - prevDensity;
			newStep = GetViewStep(dDensity, scatter.w, alpha, stepScale);
			prevDensity = alpha;	// (:136 `step = (step + newStep) * 0.5` is dead: overwritten at :150)

			// :140-146 as compiled: a = (transm * ABSORPTION) * alpha, scatter.rgb += (a * color) * light, scatter.a += a
			const float a = transm * kAbsorption * alpha;
			scatter.x += a * color.x * light.x;
			scatter.y += a * color.y * light.y;
			scatter.z += a * color.z * light.z;
			scatter.w += a;

			if (transm < kZeroThreshold) break;
		}

		t += newStep;
		if (t > tMax) break;
	}
	scatter.x *= kInvTwoPi; scatter.y *= kInvTwoPi; scatter.z *= kInvTwoPi;
	return scatter;
}
} // namespace mv
