/* TEST INFRASTRUCTURE -- a second, independent statement of the fixed-function stages the reference's shaders call into:
 * the Direct3D 11.3 functional specification's texture addressing, filtering and format conversion, as far as the path uses them
 * (MultiRayCaster.cpp:556-566: LINEAR_CLAMP, POINT_CLAMP, LINEAR_LESS_EQUAL comparison; RGBA16F / R11G11B10F / R32F / D16 views).
 *
 * It is what answers dx.op.sampleLevel / sampleCmpLevelZero / textureGather / textureLoad / textureStore inside oracle/_ref/libmv_dxref.so
 * (dxil_rt.c).  It shares NO code with oracle/mv_oracle.cpp's sampler (mvo_sample_*): it is written from the specification's
 * definitions, integer-first where the oracle is float-first, so that comparing the oracle with the reference's compiled shaders no
 * longer compares one sampler with itself (round-2 review).  Two things remain common by necessity, and are stated, not hidden:
 *   - the evaluation form the specification leaves open is the contract of DESIGN.md section 2: separable filtering in the order
 *     x, y, z, each step a + w (b - a) with the multiply-add rounded once;
 *   - which texel a TextureCube tap BEYOND its face's edge resolves to is implementation-defined in D3D: cube_edge_rule below is the
 *     contract's rule (the texel the out-of-face texel centre re-projects to), restated here.
 * Switches for what the specification allows a device to do differently (d3ds_options):
 *   - frac_bits: texel coordinates snapped to a fixed-point grid with that many fractional bits before filtering (the specification
 *     requires at least 8; 0 = no snapping, weights in full fp32 -- the contract);
 *   - r11_truncate: float32 -> float11 / float10 conversion by truncation instead of round-to-nearest-even.
 * tests/test_oracle_vs_reference_dxil.py reports what each switch moves. */
#include <math.h>
#include <stdint.h>
#include <string.h>

static int g_frac_bits = 0;
static int g_r11_truncate = 0;
void d3ds_options(int frac_bits, int r11_truncate)
{
	g_frac_bits = frac_bits < 0 ? 0 : (frac_bits > 23 ? 23 : frac_bits);
	g_r11_truncate = r11_truncate != 0;
}

/* ---------------- formats ---------------- */
/* IEEE binary16 -> binary32: exact, denormals included */
float d3ds_f16_to_f32(uint16_t h)
{
	const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
	const int exp = (h >> 10) & 31;
	const uint32_t man = h & 0x3ffu;
	float mag;
	if (exp == 0) mag = ldexpf((float)man, -24);						/* zero / denormal: man * 2^-24 */
	else if (exp == 31) { const uint32_t b = 0x7f800000u | (man << 13); memcpy(&mag, &b, 4); }	/* inf / NaN */
	else mag = ldexpf((float)(man | 0x400u), exp - 25);				/* (1024 + man) * 2^(exp - 15 - 10) */
	uint32_t bits; memcpy(&bits, &mag, 4); bits |= sign;
	float out; memcpy(&out, &bits, 4);
	return out;
}

/* binary32 -> a small float with `mb` mantissa bits and a 5-bit exponent (bias 15), magnitude only; round to nearest even, or truncate */
static uint32_t to_small_float(float f, int mb, int truncate)
{
	uint32_t u; memcpy(&u, &f, 4);
	u &= 0x7fffffffu;
	const uint32_t inf = 31u << mb;
	if (u > 0x7f800000u) return inf | (1u << (mb - 1));	/* NaN */
	if (u == 0x7f800000u) return inf;
	if (u == 0u) return 0u;
	/* value = m * 2^e with m an integer of 24 bits (normal input) */
	int e;
	const float fr = frexpf(f < 0.0f ? -f : f, &e);		/* fr in [0.5, 1) */
	uint64_t m = (uint64_t)ldexpf(fr, 24);				/* 24-bit integer, top bit set */
	e -= 24;												/* value = m * 2^e exactly */
	/* target: normal numbers have mb + 1 significant bits; the smallest normal exponent is 2^-14, denormal step 2^(-14 - mb) */
	int shift = 24 - (mb + 1);								/* bits to drop for a normal result */
	int unbiased = e + 23;									/* floor(log2 value) */
	if (unbiased < -14) shift += -14 - unbiased;			/* denormal: drop more */
	if (shift > 40) return 0u;
	uint64_t q = m >> shift;
	if (!truncate)
	{
		const uint64_t rem = m & (((uint64_t)1 << shift) - 1u), half = (uint64_t)1 << (shift - 1);
		if (rem > half || (rem == half && (q & 1u))) ++q;
	}
	uint32_t out;
	if (unbiased < -14) out = (uint32_t)q;					/* denormal (a carry into 1 << mb becomes the smallest normal: same bits) */
	else
	{
		int be = unbiased + 15;
		if (q >> (mb + 1)) { q >>= 1; ++be; }				/* rounding carried into the next binade */
		if (be >= 31) return truncate ? (inf - 1u) : inf;	/* overflow: inf (nearest) or the largest finite (truncation) */
		out = ((uint32_t)be << mb) | ((uint32_t)q & ((1u << mb) - 1u));
	}
	return out;
}
uint16_t d3ds_f32_to_f16(float f)	/* typed UAV store to R16G16B16A16_FLOAT: round to nearest even (required by the specification) */
{
	uint32_t u; memcpy(&u, &f, 4);
	return (uint16_t)(((u >> 16) & 0x8000u) | to_small_float(f, 10, 0));
}
/* R11G11B10_FLOAT: no sign bits; negative values clamp to zero, NaN stays NaN, +inf stays inf */
uint32_t d3ds_pack_r11g11b10f(float r, float g, float b)
{
	const float c[3] = { r, g, b };
	uint32_t out[3];
	for (int i = 0; i < 3; ++i)
	{
		const int mb = i < 2 ? 6 : 5;
		const int neg = signbit(c[i]) && !isnan(c[i]);
		out[i] = neg ? 0u : to_small_float(c[i], mb, g_r11_truncate);
	}
	return (out[0] & 0x7ffu) | ((out[1] & 0x7ffu) << 11) | ((out[2] & 0x3ffu) << 22);
}
static float small_to_f32(uint32_t v, int mb)
{
	const int exp = (int)(v >> mb);
	const uint32_t man = v & ((1u << mb) - 1u);
	if (exp == 0) return ldexpf((float)man, -14 - mb);
	if (exp == 31) return man ? NAN : INFINITY;
	return ldexpf((float)(man | (1u << mb)), exp - 15 - mb);
}
void d3ds_unpack_r11g11b10f(uint32_t p, float out[3])
{
	out[0] = small_to_f32(p & 0x7ffu, 6); out[1] = small_to_f32((p >> 11) & 0x7ffu, 6); out[2] = small_to_f32(p >> 22, 5);
}

/* ---------------- addressing ---------------- */
/* One axis of a LINEAR + CLAMP fetch at normalised coordinate `u` on `size` texels with integer texel offset `off`:
 * the two texel indices and the weight of the second.  Texel centres sit at i + 0.5: the sample point in texel units is u size - 0.5. */
typedef struct { int64_t i0, i1; float w; } axis_taps;
static axis_taps linear_axis(float u, uint32_t size, int off)
{
	axis_taps t;
	float p = u * (float)size - 0.5f;
	int64_t base;
	if (!(p == p)) { base = -2; t.w = p; }				/* NaN coordinate: an out-of-range address (clamped below) and a NaN weight -> a NaN result */
	else if (g_frac_bits > 0)
	{
		/* the fixed-point form: round the coordinate to `frac_bits` fractional bits, split integer and fraction */
		const double scaled = floor((double)p * (double)(1 << g_frac_bits) + 0.5);
		const double ip = floor(scaled / (double)(1 << g_frac_bits));
		base = ip < -4.0e18 ? INT64_MIN / 2 : (ip > 4.0e18 ? INT64_MAX / 2 : (int64_t)ip);
		t.w = (float)((scaled - ip * (double)(1 << g_frac_bits)) / (double)(1 << g_frac_bits));
	}
	else
	{
		const float fl = floorf(p);
		base = fl < -4.0e18f ? INT64_MIN / 2 : (fl > 4.0e18f ? INT64_MAX / 2 : (int64_t)fl);
		t.w = p - fl;
	}
	const int64_t hi = (int64_t)size - 1;
	int64_t a = base + off, b = base + off + 1;
	t.i0 = a < 0 ? 0 : (a > hi ? hi : a);
	t.i1 = b < 0 ? 0 : (b > hi ? hi : b);
	return t;
}
static float mix1(float a, float b, float w) { return fmaf(w, b - a, a); }	/* a + w (b - a), the product-sum rounded once */

/* trilinear over eight values given as v[z][y][x] */
static float trilinear(float v[2][2][2], float wx, float wy, float wz)
{
	const float y0 = mix1(mix1(v[0][0][0], v[0][0][1], wx), mix1(v[0][1][0], v[0][1][1], wx), wy);
	const float y1 = mix1(mix1(v[1][0][0], v[1][0][1], wx), mix1(v[1][1][0], v[1][1][1], wx), wy);
	return mix1(y0, y1, wz);
}

/* Texture3D<float4> RGBA16F, SampleLevel(LINEAR_CLAMP, uvw, 0, offset) */
void d3ds_sample3d_rgba16f(const uint16_t* tex, uint32_t w, uint32_t h, uint32_t d, const float uvw[3], const int32_t off[3], float out[4])
{
	const axis_taps tx = linear_axis(uvw[0], w, off[0]), ty = linear_axis(uvw[1], h, off[1]), tz = linear_axis(uvw[2], d, off[2]);
	const int64_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 }, zs[2] = { tz.i0, tz.i1 };
	for (int c = 0; c < 4; ++c)
	{
		float v[2][2][2];
		for (int k = 0; k < 2; ++k) for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i)
			v[k][j][i] = d3ds_f16_to_f32(tex[4 * (((size_t)zs[k] * h + (size_t)ys[j]) * w + (size_t)xs[i]) + c]);
		out[c] = trilinear(v, tx.w, ty.w, tz.w);
	}
}
/* Texture3D<float3> R11G11B10F */
void d3ds_sample3d_r11g11b10f(const uint32_t* tex, uint32_t w, uint32_t h, uint32_t d, const float uvw[3], float out[3])
{
	const axis_taps tx = linear_axis(uvw[0], w, 0), ty = linear_axis(uvw[1], h, 0), tz = linear_axis(uvw[2], d, 0);
	const int64_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 }, zs[2] = { tz.i0, tz.i1 };
	float texel[2][2][2][3];
	for (int k = 0; k < 2; ++k) for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i)
		d3ds_unpack_r11g11b10f(tex[((size_t)zs[k] * h + (size_t)ys[j]) * w + (size_t)xs[i]], texel[k][j][i]);
	for (int c = 0; c < 3; ++c)
	{
		float v[2][2][2];
		for (int k = 0; k < 2; ++k) for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) v[k][j][i] = texel[k][j][i][c];
		out[c] = trilinear(v, tx.w, ty.w, tz.w);
	}
}
/* Texture3D<float> R32F (the DDS source of CSR32FToRGBA16F) */
float d3ds_sample3d_r32f(const float* tex, uint32_t w, uint32_t h, uint32_t d, const float uvw[3])
{
	const axis_taps tx = linear_axis(uvw[0], w, 0), ty = linear_axis(uvw[1], h, 0), tz = linear_axis(uvw[2], d, 0);
	const int64_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 }, zs[2] = { tz.i0, tz.i1 };
	float v[2][2][2];
	for (int k = 0; k < 2; ++k) for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i)
		v[k][j][i] = tex[((size_t)zs[k] * h + (size_t)ys[j]) * w + (size_t)xs[i]];
	return trilinear(v, tx.w, ty.w, tz.w);
}
/* Texture2D<float> R32F, SampleLevel(POINT_CLAMP): the texel whose cell contains the sample point */
float d3ds_point2d_r32f(const float* tex, uint32_t w, uint32_t h, float u, float v)
{
	const float px = floorf(u * (float)w), py = floorf(v * (float)h);
	const int64_t x = !(px >= 0.0f) ? 0 : (px > (float)(w - 1u) ? (int64_t)w - 1 : (int64_t)px);	/* (NaN -> 0) */
	const int64_t y = !(py >= 0.0f) ? 0 : (py > (float)(h - 1u) ? (int64_t)h - 1 : (int64_t)py);
	return tex[(size_t)y * w + (size_t)x];
}
/* Texture2D D16_UNORM, SampleCmpLevelZero(LINEAR, LESS_EQUAL): the reference value is clamped to the format's range, every one of the
 * four texels is compared (reference <= texel -> 1, else 0) and the four results are filtered bilinearly */
float d3ds_cmp2d_d16(const uint16_t* tex, uint32_t w, uint32_t h, float u, float v, float ref)
{
	const axis_taps tx = linear_axis(u, w, 0), ty = linear_axis(v, h, 0);
	const float r = !(ref == ref) ? 0.0f : (ref < 0.0f ? 0.0f : (ref > 1.0f ? 1.0f : ref));
	const int64_t xs[2] = { tx.i0, tx.i1 }, ys[2] = { ty.i0, ty.i1 };
	float c[2][2];
	for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i)
		c[j][i] = r <= (float)tex[(size_t)ys[j] * w + (size_t)xs[i]] / 65535.0f ? 1.0f : 0.0f;	/* UNORM16 -> float: n / 65535 */
	return mix1(mix1(c[0][0], c[0][1], tx.w), mix1(c[1][0], c[1][1], tx.w), ty.w);
}

/* ---------------- TextureCube ---------------- */
/* Face selection of the D3D cube map (major axis; faces +X -X +Y -Y +Z -Z) and the face coordinates in [0, 1]:
 *   +X: (sc, tc) = (-z, -y)   -X: (z, -y)   +Y: (x, z)   -Y: (x, -z)   +Z: (x, -y)   -Z: (-x, -y),   s = (sc / |ma| + 1) / 2 */
static void cube_face(const float dir[3], uint32_t* face, float* s, float* t)
{
	const float ax = fabsf(dir[0]), ay = fabsf(dir[1]), az = fabsf(dir[2]);
	float sc, tc, ma;
	if (ax >= ay && ax >= az) { const int pos = dir[0] > 0.0f; *face = pos ? 0u : 1u; sc = pos ? -dir[2] : dir[2]; tc = -dir[1]; ma = ax; }
	else if (ay >= az) { const int pos = dir[1] > 0.0f; *face = pos ? 2u : 3u; sc = dir[0]; tc = pos ? dir[2] : -dir[2]; ma = ay; }
	else { const int pos = dir[2] > 0.0f; *face = pos ? 4u : 5u; sc = pos ? dir[0] : -dir[0]; tc = -dir[1]; ma = az; }
	*s = (sc / ma + 1.0f) * 0.5f;
	*t = (tc / ma + 1.0f) * 0.5f;
}
/* the contract's rule for a tap beyond its face (D3D: implementation-defined): the texel the tap's centre re-projects to */
static uint64_t cube_edge_rule(uint32_t face, int64_t i, int64_t j, uint32_t d)
{
	if (i >= 0 && j >= 0 && i < (int64_t)d && j < (int64_t)d) return ((uint64_t)face * d + (uint64_t)j) * d + (uint64_t)i;
	const float a = ((float)i + 0.5f) / (float)d * 2.0f - 1.0f, b = ((float)j + 0.5f) / (float)d * 2.0f - 1.0f;
	float dir[3];
	switch (face)	/* inverse of cube_face on the face's plane */
	{
	case 0: dir[0] = 1.0f; dir[1] = -b; dir[2] = -a; break;
	case 1: dir[0] = -1.0f; dir[1] = -b; dir[2] = a; break;
	case 2: dir[0] = a; dir[1] = 1.0f; dir[2] = b; break;
	case 3: dir[0] = a; dir[1] = -1.0f; dir[2] = -b; break;
	case 4: dir[0] = a; dir[1] = -b; dir[2] = 1.0f; break;
	default: dir[0] = -a; dir[1] = -b; dir[2] = -1.0f; break;
	}
	uint32_t f2; float s, t;
	cube_face(dir, &f2, &s, &t);
	int64_t ii = (int64_t)floorf(s * (float)d), jj = (int64_t)floorf(t * (float)d);
	ii = ii < 0 ? 0 : (ii > (int64_t)d - 1 ? (int64_t)d - 1 : ii);
	jj = jj < 0 ? 0 : (jj > (int64_t)d - 1 ? (int64_t)d - 1 : jj);
	return ((uint64_t)f2 * d + (uint64_t)jj) * d + (uint64_t)ii;
}
/* The 2 x 2 footprint of a bilinear fetch / Gather at direction `dir` on a d x d cube map: texel indices into [face][y][x] storage in
 * Gather order -- (i0, j0 + 1), (i0 + 1, j0 + 1), (i0 + 1, j0), (i0, j0): the w, z, y, x components' texels, i.e. the order in which
 * Gather returns .x .y .z .w -- and the bilinear weights of the same four texels. */
void d3ds_cube_footprint(const float dir[3], uint32_t d, uint64_t idx[4], float wgt[4])
{
	uint32_t face; float s, t;
	cube_face(dir, &face, &s, &t);
	const float u = s * (float)d - 0.5f, v = t * (float)d - 0.5f;
	const float fu = floorf(u), fv = floorf(v);
	const int64_t i0 = (int64_t)fu, j0 = (int64_t)fv;
	const float fx = u - fu, fy = v - fv;
	idx[0] = cube_edge_rule(face, i0, j0 + 1, d); idx[1] = cube_edge_rule(face, i0 + 1, j0 + 1, d);
	idx[2] = cube_edge_rule(face, i0 + 1, j0, d); idx[3] = cube_edge_rule(face, i0, j0, d);
	wgt[0] = (1.0f - fx) * fy; wgt[1] = fx * fy; wgt[2] = fx * (1.0f - fy); wgt[3] = (1.0f - fx) * (1.0f - fy);
}
