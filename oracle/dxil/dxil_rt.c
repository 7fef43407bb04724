/* TEST INFRASTRUCTURE: the "device" under the reference's compiled shaders when they run on the CPU (oracle/_ref/libmv_dxref.so, see
 * cso2ll.py).  The translated DXIL calls rt_* for every dx.op intrinsic; this file answers them:
 *   - thread / group ids, resource handles, raw / structured / typed / constant buffer loads and stores: plain memory semantics;
 *   - dx.op.unary / binary / tertiary / dot: IEEE fp32, one operation = one rounding (FMad: see dxref_set_fused_mad);
 *   - wave intrinsics: the lanes of a wavefront run as threads that meet at every wave operation (dxref_dispatch);
 *   - texture sampling, typed texture loads / stores and their format conversion: oracle/dxil/d3d_sampler.c, a statement of the D3D11.3
 *     functional specification's addressing / filtering / conversion rules written independently of the oracle's sampler (round 2 answered
 *     these with the oracle's own mvo_sample_*: oracle-vs-_ref then compared that sampler with itself).  It is still a builder-written
 *     stand-in for hardware the bytecode does not contain; what it adds is that the oracle's sampler is now checked by a second statement.
 * Only tests/ (and scripts that make fixtures) may use this library. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the fixed-function stages: oracle/dxil/d3d_sampler.c, an independent statement of the D3D11.3 functional specification (NOT the oracle's sampler) */
void d3ds_options(int frac_bits, int r11_truncate);
float d3ds_f16_to_f32(uint16_t h);
uint16_t d3ds_f32_to_f16(float f);
uint32_t d3ds_pack_r11g11b10f(float r, float g, float b);
void d3ds_sample3d_rgba16f(const uint16_t* tex, uint32_t w, uint32_t h, uint32_t d, const float uvw[3], const int32_t off[3], float out[4]);
void d3ds_sample3d_r11g11b10f(const uint32_t* tex, uint32_t w, uint32_t h, uint32_t d, const float uvw[3], float out[3]);
float d3ds_sample3d_r32f(const float* tex, uint32_t w, uint32_t h, uint32_t d, const float uvw[3]);
float d3ds_point2d_r32f(const float* tex, uint32_t w, uint32_t h, float u, float v);
float d3ds_cmp2d_d16(const uint16_t* tex, uint32_t w, uint32_t h, float u, float v, float ref);
void d3ds_cube_footprint(const float dir[3], uint32_t d, uint64_t idx[4], float wgt[4]);

enum { DXR_RAW = 1, DXR_TYPED_BUF = 2, DXR_CBUF = 3, DXR_TEX2D = 4, DXR_TEX3D = 5, DXR_TEX2DARRAY = 6, DXR_SAMPLER = 7, DXR_TEXCUBE = 8 };
enum { DXF_NONE = 0, DXF_RGBA16F = 1, DXF_R11G11B10F = 2, DXF_R32F = 3, DXF_D16 = 4, DXF_RGBA16U = 5, DXF_R32U = 6, DXF_RGBA32F = 7 };

typedef struct dxr_res
{
	int kind, fmt;
	void* data;
	uint32_t w, h, d, stride;	/* stride: bytes per element of a structured buffer */
	uint32_t counter;			/* UAV counter (IncrementCounter) */
	int cmp_sampler;
} dxr_res;

#define DXR_MAX_BINDINGS 4096
typedef struct { uint32_t cls, range, index; dxr_res res; } dxr_binding;
static dxr_binding g_bind[DXR_MAX_BINDINGS];
static uint32_t g_nbind;
static int g_fused_mad = 0;
static dxr_res g_null_res;

/* per-lane context */
typedef struct
{
	uint32_t tid[3], gid[3], tig[3], flat, lane;
	struct dxr_wave* wave;
	const uint32_t (*in)[4];	/* pixel / vertex shader inputs: [signature element][component], raw bits */
	float out[8][4];			/* outputs: [signature element][component] */
} dxr_ctx;
static __thread dxr_ctx* t_ctx;

/* ---------------- wavefront rendezvous ----------------
 * The lanes of a wave run as threads.  A wave intrinsic blocks its lane; when EVERY lane that is still running is blocked, the
 * blocked lanes at the earliest call site (lowest code address: the shaders' wave intrinsics sit in forward control flow) form the
 * set of lanes that a SIMT machine would have active there -- lanes that skipped the enclosing branch are waiting further down, lanes
 * that left the shader no longer count -- and the operation completes for them.  `cluster` (dxref_dispatch) > 0 additionally splits
 * WaveActive* reductions by lane / cluster: CSVolumeCull runs them inside `for (i ...) if (i == wTid.y) return WaveActiveMax(x)`,
 * i.e. once per 8-lane volume cluster with only that cluster active. */
typedef struct dxr_wave
{
	pthread_mutex_t mu;
	pthread_cond_t cv;
	uint32_t size, running, blocked, cluster;
	void* site[64];			/* call site a lane is blocked at (NULL: running) */
	uint32_t val_u[64];
	float val_f[64];
	uint32_t out_u[64];
	float out_f[64];
	void* out_p[64];
	uint32_t ballot[64][4];
	uint8_t present[64];	/* lanes of the group being completed */
	int kind[64], op[64];
	uint32_t want[64];		/* ReadLaneAt: the lane whose value this lane asked for */
	/* registers published for ReadLaneAt (cso2ll.py): value and "has been written" per (id, lane) */
	uint32_t pub_u[256][64];
	uint8_t pub_set[256][64];
	uint8_t exited[64];
} dxr_wave;

enum { WK_READ = 1, WK_FIRST = 2, WK_BALLOT = 3, WK_ACTIVE = 4, WK_COUNT = 5, WK_PREFIX = 6, WK_ALLOC = 7 };
static void* node_alloc_records(uint32_t count);

static void wave_try_complete(dxr_wave* w)
{
	while (w->running && w->blocked == w->running)
	{
		void* s = NULL;
		for (uint32_t l = 0; l < w->size; ++l) if (w->site[l] && (!s || (uintptr_t)w->site[l] < (uintptr_t)s)) s = w->site[l];
		if (!s) return;
		int kind = 0, op = 0;
		memset(w->present, 0, sizeof(w->present));
		for (uint32_t l = 0; l < w->size; ++l) if (w->site[l] == s) { w->present[l] = 1; kind = w->kind[l]; op = w->op[l]; }
		if (kind == WK_READ)
		{
			/* every result goes to the asking lane's own slot: another group may complete before this one has read its results */
			for (uint32_t l = 0; l < w->size; ++l)
				if (w->present[l]) { const uint32_t q = w->want[l] < w->size ? w->want[l] : l; w->out_u[l] = w->val_u[q]; w->out_f[l] = w->val_f[q]; }
		}
		else if (kind == WK_FIRST)
		{
			uint32_t first = 0;
			while (first < w->size && !w->present[first]) ++first;
			for (uint32_t l = 0; l < w->size; ++l) if (w->present[l]) { w->out_u[l] = w->val_u[first]; w->out_f[l] = w->val_f[first]; }
		}
		else if (kind == WK_BALLOT)
		{
			uint32_t b[4] = { 0, 0, 0, 0 };
			for (uint32_t l = 0; l < w->size; ++l) if (w->present[l] && w->val_u[l]) b[l >> 5] |= 1u << (l & 31);
			for (uint32_t l = 0; l < w->size; ++l) if (w->present[l]) memcpy(w->ballot[l], b, sizeof(b));
		}
		else if (kind == WK_ACTIVE)
		{
			const uint32_t cl = w->cluster ? w->cluster : w->size;
			for (uint32_t base = 0; base < w->size; base += cl)
			{
				float acc = 0.0f;
				int have = 0;
				for (uint32_t l = base; l < base + cl && l < w->size; ++l)
				{
					if (!w->present[l]) continue;
					const float v = w->val_f[l];
					if (!have) { acc = v; have = 1; continue; }
					switch (op) { case 0: acc += v; break; case 1: acc *= v; break; case 2: acc = fminf(acc, v); break; case 3: acc = fmaxf(acc, v); break; default: abort(); }
				}
				for (uint32_t l = base; l < base + cl && l < w->size; ++l) if (w->present[l]) w->out_f[l] = acc;
			}
		}
		else if (kind == WK_COUNT || kind == WK_PREFIX)
		{
			/* WaveActiveCountBits / WavePrefixCountBits over the lanes that are active at this site */
			uint32_t n = 0;
			for (uint32_t l = 0; l < w->size; ++l)
				if (w->present[l]) { if (kind == WK_PREFIX) w->out_u[l] = n; n += w->val_u[l] ? 1u : 0u; }
			if (kind == WK_COUNT) for (uint32_t l = 0; l < w->size; ++l) if (w->present[l]) w->out_u[l] = n;
		}
		else if (kind == WK_ALLOC)
		{
			/* GetGroupNodeOutputRecords: one allocation for the group (here one wave), every lane gets the same records */
			uint32_t first = 0;
			while (first < w->size && !w->present[first]) ++first;
			void* recs = node_alloc_records(w->val_u[first]);
			for (uint32_t l = 0; l < w->size; ++l) if (w->present[l]) w->out_p[l] = recs;
		}
		if (getenv("DXREF_TRACE")) { fprintf(stderr, "wave op kind %d op %d site %p lanes:", kind, op, s); for (uint32_t l = 0; l < w->size; ++l) if (w->present[l]) fprintf(stderr, " %u(v=%g,u=%u,want=%u->%g)", l, w->val_f[l], w->val_u[l], w->want[l], w->out_f[l]); fprintf(stderr, "\n"); }
		for (uint32_t l = 0; l < w->size; ++l) if (w->present[l]) { w->site[l] = NULL; --w->blocked; }
		pthread_cond_broadcast(&w->cv);
	}
}
static void wave_meet(void* site, int kind, int op, uint32_t vu, float vf, uint32_t want)
{
	dxr_wave* w = t_ctx->wave;
	const uint32_t lane = t_ctx->lane;
	pthread_mutex_lock(&w->mu);
	w->val_u[lane] = vu; w->val_f[lane] = vf; w->kind[lane] = kind; w->op[lane] = op; w->want[lane] = want;
	w->site[lane] = site;
	++w->blocked;
	wave_try_complete(w);
	while (w->site[lane]) pthread_cond_wait(&w->cv, &w->mu);
	pthread_mutex_unlock(&w->mu);
}
static void wave_lane_exit(dxr_wave* w)
{
	pthread_mutex_lock(&w->mu);
	--w->running;
	w->exited[t_ctx->lane] = 1;
	pthread_cond_broadcast(&w->cv);	/* lanes waiting for a register of this one */
	wave_try_complete(w);
	pthread_mutex_unlock(&w->mu);
}

/* ---------------- binding ---------------- */
void dxref_reset(void) { g_nbind = 0; }
void dxref_set_fused_mad(int on) { g_fused_mad = on; }
/* what the specification lets a device do differently in the fixed-function stages: texel coordinates snapped to `frac_bits` fractional
 * bits (>= 8 allowed; 0 = none), float11 / float10 stores by truncation instead of round-to-nearest-even */
void dxref_set_sampler_options(int frac_bits, int r11_truncate) { d3ds_options(frac_bits, r11_truncate); }
int dxref_bind(uint32_t cls, uint32_t range, uint32_t index, int kind, int fmt, void* data, uint32_t w, uint32_t h, uint32_t d, uint32_t stride)
{
	if (g_nbind == DXR_MAX_BINDINGS) return -1;
	dxr_binding* b = &g_bind[g_nbind++];
	b->cls = cls; b->range = range; b->index = index;
	memset(&b->res, 0, sizeof(b->res));
	b->res.kind = kind; b->res.fmt = fmt; b->res.data = data; b->res.w = w; b->res.h = h; b->res.d = d; b->res.stride = stride;
	return 0;
}
uint32_t dxref_counter(uint32_t cls, uint32_t range, uint32_t index)
{
	for (uint32_t i = 0; i < g_nbind; ++i) if (g_bind[i].cls == cls && g_bind[i].range == range && g_bind[i].index == index) return g_bind[i].res.counter;
	return 0;
}

void* rt_createHandle(int32_t op, int32_t cls, int32_t range, int32_t index, int32_t nonuniform)
{
	(void)op; (void)nonuniform;
	for (uint32_t i = 0; i < g_nbind; ++i)
		if (g_bind[i].cls == (uint32_t)cls && g_bind[i].range == (uint32_t)range && g_bind[i].index == (uint32_t)index) return &g_bind[i].res;
	return &g_null_res;	/* an unbound descriptor: loads give 0, stores are dropped (D3D's null descriptor) */
}

/* ---------------- ids ---------------- */
int32_t rt_threadId_i32(int32_t op, int32_t c) { (void)op; return (int32_t)t_ctx->tid[c]; }
int32_t rt_groupId_i32(int32_t op, int32_t c) { (void)op; return (int32_t)t_ctx->gid[c]; }
int32_t rt_threadIdInGroup_i32(int32_t op, int32_t c) { (void)op; return (int32_t)t_ctx->tig[c]; }
int32_t rt_flattenedThreadIdInGroup_i32(int32_t op) { (void)op; return (int32_t)t_ctx->flat; }

/* ---------------- buffers ---------------- */
typedef struct { int32_t v[4]; int32_t status; } ret_i32;
typedef struct { float v[4]; int32_t status; } ret_f32;
typedef struct { float v[4]; } cbuf_f32;
typedef struct { int32_t v[4]; } cbuf_i32;
typedef struct { int32_t v[4]; } dims_t;

static const uint8_t* elem_ptr(const dxr_res* r, int32_t index, int32_t offset)
{
	if (!r->data) return NULL;
	if (!r->stride) return (const uint8_t*)r->data + (size_t)(uint32_t)index;	/* ByteAddressBuffer: the index is the byte address */
	return (const uint8_t*)r->data + (size_t)(uint32_t)index * r->stride + (size_t)(uint32_t)offset;
}
void rt_rawBufferLoad_i32(ret_i32* out, int32_t op, dxr_res* r, int32_t index, int32_t offset, int32_t mask, int32_t align)
{
	(void)op; (void)align;
	memset(out, 0, sizeof(*out));
	const uint8_t* p = elem_ptr(r, index, offset);
	if (!p) return;
	for (int c = 0, k = 0; c < 4; ++c) if (mask & (1 << c)) { memcpy(&out->v[c], p + 4 * k, 4); ++k; }
}
void rt_rawBufferLoad_f32(ret_f32* out, int32_t op, dxr_res* r, int32_t index, int32_t offset, int32_t mask, int32_t align)
{
	rt_rawBufferLoad_i32((ret_i32*)out, op, r, index, offset, mask, align);
}
void rt_rawBufferStore_i32(int32_t op, dxr_res* r, int32_t index, int32_t offset, int32_t v0, int32_t v1, int32_t v2, int32_t v3, int32_t mask, int32_t align)
{
	(void)op; (void)align;
	uint8_t* p = (uint8_t*)elem_ptr(r, index, offset);
	if (!p) return;
	const int32_t v[4] = { v0, v1, v2, v3 };
	for (int c = 0, k = 0; c < 4; ++c) if (mask & (1 << c)) { memcpy(p + 4 * k, &v[c], 4); ++k; }
}
void rt_bufferLoad_i32(ret_i32* out, int32_t op, dxr_res* r, int32_t index, int32_t woffset)
{
	(void)op; (void)woffset;
	memset(out, 0, sizeof(*out));
	if (!r->data) return;
	if (r->fmt == DXF_RGBA16U) { const uint16_t* p = (const uint16_t*)r->data + 4 * (size_t)(uint32_t)index; for (int c = 0; c < 4; ++c) out->v[c] = p[c]; }
	else if (r->fmt == DXF_R32U) out->v[0] = (int32_t)((const uint32_t*)r->data)[(uint32_t)index];
	else { fprintf(stderr, "dxil_rt: bufferLoad on format %d\n", r->fmt); abort(); }
}
void rt_bufferStore_i32(int32_t op, dxr_res* r, int32_t index, int32_t woffset, int32_t v0, int32_t v1, int32_t v2, int32_t v3, int32_t mask)
{
	(void)op; (void)woffset; (void)mask;
	if (!r->data) return;
	if (r->fmt == DXF_RGBA16U) { uint16_t* p = (uint16_t*)r->data + 4 * (size_t)(uint32_t)index; p[0] = (uint16_t)v0; p[1] = (uint16_t)v1; p[2] = (uint16_t)v2; p[3] = (uint16_t)v3; }
	else if (r->fmt == DXF_R32U) ((uint32_t*)r->data)[(uint32_t)index] = (uint32_t)v0;
	else { fprintf(stderr, "dxil_rt: bufferStore on format %d\n", r->fmt); abort(); }
}
int32_t rt_bufferUpdateCounter(int32_t op, dxr_res* r, int32_t inc)
{
	(void)op;
	/* IncrementCounter returns the value before, DecrementCounter the value after */
	if ((int8_t)inc > 0) return (int32_t)__atomic_fetch_add(&r->counter, 1u, __ATOMIC_RELAXED);
	return (int32_t)__atomic_sub_fetch(&r->counter, 1u, __ATOMIC_RELAXED);
}
void rt_cbufferLoadLegacy_f32(cbuf_f32* out, int32_t op, dxr_res* r, int32_t reg)
{
	(void)op;
	memset(out, 0, sizeof(*out));
	if (!r->data) return;
	const size_t off = (size_t)(uint32_t)reg * 16u, size = r->w;
	for (int c = 0; c < 4; ++c) if (off + 4u * c + 4u <= size) memcpy(&out->v[c], (const uint8_t*)r->data + off + 4u * c, 4);
}
void rt_cbufferLoadLegacy_i32(cbuf_i32* out, int32_t op, dxr_res* r, int32_t reg) { rt_cbufferLoadLegacy_f32((cbuf_f32*)out, op, r, reg); }
void rt_getDimensions(dims_t* out, int32_t op, dxr_res* r, int32_t mip)
{
	(void)op; (void)mip;
	out->v[0] = (int32_t)r->w; out->v[1] = (int32_t)r->h; out->v[2] = (int32_t)r->d; out->v[3] = 1;
	if (r->kind == DXR_RAW) { out->v[0] = (int32_t)r->w; out->v[1] = (int32_t)r->stride; }	/* element count, stride */
}

/* ---------------- textures: d3d_sampler.c (an independent statement of the D3D11.3 sampler / format rules) ---------------- */
static void cube_taps(const dxr_res* t, float x, float y, float z, uint64_t idx[4], float w[4]);
static float cube_texel(const dxr_res* t, uint64_t i, int ch);
void rt_sampleLevel_f32(ret_f32* out, int32_t op, dxr_res* t, dxr_res* smp, float c0, float c1, float c2, float c3, int32_t o0, int32_t o1, int32_t o2, float lod)
{
	(void)op; (void)smp; (void)c3; (void)lod;
	memset(out, 0, sizeof(*out));
	if (!t->data) return;
	if (t->kind == DXR_TEX3D && t->fmt == DXF_RGBA16F)
	{
		const float uvw[3] = { c0, c1, c2 };
		const int32_t off[3] = { o0, o1, o2 };
		d3ds_sample3d_rgba16f((const uint16_t*)t->data, t->w, t->h, t->d, uvw, off, out->v);
	}
	else if (t->kind == DXR_TEX3D && t->fmt == DXF_R11G11B10F)
	{
		const float uvw[3] = { c0, c1, c2 };
		d3ds_sample3d_r11g11b10f((const uint32_t*)t->data, t->w, t->h, t->d, uvw, out->v);
		out->v[3] = 1.0f;
	}
	else if (t->kind == DXR_TEX3D && t->fmt == DXF_R32F)
	{
		const float uvw[3] = { c0, c1, c2 };
		out->v[0] = d3ds_sample3d_r32f((const float*)t->data, t->w, t->h, t->d, uvw);
		out->v[3] = 1.0f;
	}
	else if (t->kind == DXR_TEXCUBE)
	{
		/* bilinear over the Gather footprint, summed tap 0..3 like the oracle's fallback colour (PSCube.hlsli:59) */
		uint64_t idx[4]; float w[4];
		cube_taps(t, c0, c1, c2, idx, w);
		for (int ch = 0; ch < 4; ++ch)
			out->v[ch] = cube_texel(t, idx[0], ch) * w[0] + cube_texel(t, idx[1], ch) * w[1] + cube_texel(t, idx[2], ch) * w[2] + cube_texel(t, idx[3], ch) * w[3];
	}
	else if (t->kind == DXR_TEX2D && t->fmt == DXF_R32F)
	{
		const float v = d3ds_point2d_r32f((const float*)t->data, t->w, t->h, c0, c1);
		out->v[0] = out->v[1] = out->v[2] = out->v[3] = v;	/* R32F: (r, 0, 0, 1); the shaders read .x only */
		out->v[1] = out->v[2] = 0.0f; out->v[3] = 1.0f;
	}
	else { fprintf(stderr, "dxil_rt: sampleLevel on kind %d format %d\n", t->kind, t->fmt); abort(); }
}
float rt_sampleCmpLevelZero_f32_scalar(dxr_res* t, float u, float v, float ref)
{
	if (!t->data) return 0.0f;
	if (t->kind == DXR_TEX2D && t->fmt == DXF_D16) return d3ds_cmp2d_d16((const uint16_t*)t->data, t->w, t->h, u, v, ref);
	fprintf(stderr, "dxil_rt: sampleCmpLevelZero on kind %d format %d\n", t->kind, t->fmt); abort();
}
void rt_sampleCmpLevelZero_f32(ret_f32* out, int32_t op, dxr_res* t, dxr_res* smp, float c0, float c1, float c2, float c3, int32_t o0, int32_t o1, int32_t o2, float ref)
{
	(void)op; (void)smp; (void)c2; (void)c3; (void)o0; (void)o1; (void)o2;
	memset(out, 0, sizeof(*out));
	out->v[0] = out->v[1] = out->v[2] = out->v[3] = rt_sampleCmpLevelZero_f32_scalar(t, c0, c1, ref);
}
void rt_textureStore_f32(int32_t op, dxr_res* t, int32_t x, int32_t y, int32_t z, float v0, float v1, float v2, float v3, int32_t mask)
{
	(void)op; (void)mask;
	if (!t->data) return;
	const uint32_t ux = (uint32_t)x, uy = (uint32_t)y, uz = (uint32_t)z;
	const uint32_t depth = t->kind == DXR_TEX2D ? 1u : t->d;
	if (ux >= t->w || uy >= t->h || (t->kind != DXR_TEX2D && uz >= depth)) return;	/* out-of-bounds UAV writes are dropped */
	const size_t i = ((size_t)(t->kind == DXR_TEX2D ? 0u : uz) * t->h + uy) * t->w + ux;
	switch (t->fmt)
	{
	case DXF_RGBA16F: { uint16_t* p = (uint16_t*)t->data + 4 * i; p[0] = d3ds_f32_to_f16(v0); p[1] = d3ds_f32_to_f16(v1); p[2] = d3ds_f32_to_f16(v2); p[3] = d3ds_f32_to_f16(v3); break; }
	case DXF_R11G11B10F: ((uint32_t*)t->data)[i] = d3ds_pack_r11g11b10f(v0, v1, v2); break;
	case DXF_R32F: ((float*)t->data)[i] = v0; break;
	default: fprintf(stderr, "dxil_rt: textureStore on format %d\n", t->fmt); abort();
	}
}

/* ---------------- graphics stages: inputs, outputs, typed loads, atomics, cube maps ---------------- */
float rt_loadInput_f32(int32_t op, int32_t sig, int32_t row, int32_t col, int32_t vertex)
{
	(void)op; (void)row; (void)vertex;
	float f; memcpy(&f, &t_ctx->in[sig][col & 3], 4); return f;
}
int32_t rt_loadInput_i32(int32_t op, int32_t sig, int32_t row, int32_t col, int32_t vertex)
{
	(void)op; (void)row; (void)vertex;
	return (int32_t)t_ctx->in[sig][col & 3];
}
void rt_storeOutput_f32(int32_t op, int32_t sig, int32_t row, int32_t col, float v) { (void)op; (void)row; t_ctx->out[sig & 7][col & 3] = v; }
void rt_storeOutput_i32(int32_t op, int32_t sig, int32_t row, int32_t col, int32_t v) { (void)op; (void)row; memcpy(&t_ctx->out[sig & 7][col & 3], &v, 4); }

static int tex_index(const dxr_res* t, int32_t x, int32_t y, int32_t z, size_t* i)
{
	const uint32_t ux = (uint32_t)x, uy = (uint32_t)y, uz = t->kind == DXR_TEX2D ? 0u : (uint32_t)z;
	const uint32_t depth = t->kind == DXR_TEX2D ? 1u : t->d;
	if (!t->data || ux >= t->w || uy >= t->h || uz >= depth) return 0;
	*i = ((size_t)uz * t->h + uy) * t->w + ux;
	return 1;
}
void rt_textureLoad_f32(ret_f32* out, int32_t op, dxr_res* t, int32_t mip, int32_t c0, int32_t c1, int32_t c2, int32_t o0, int32_t o1, int32_t o2)
{
	(void)op; (void)mip; (void)o0; (void)o1; (void)o2;
	memset(out, 0, sizeof(*out));
	size_t i;
	if (!tex_index(t, c0, c1, c2, &i)) return;	/* out of bounds: zeros */
	switch (t->fmt)
	{
	case DXF_R32F: out->v[0] = ((const float*)t->data)[i]; out->v[3] = 1.0f; break;
	case DXF_RGBA16F: { const uint16_t* p = (const uint16_t*)t->data + 4 * i; for (int c = 0; c < 4; ++c) out->v[c] = d3ds_f16_to_f32(p[c]); break; }
	default: fprintf(stderr, "dxil_rt: textureLoad.f32 on format %d\n", t->fmt); abort();
	}
}
void rt_textureLoad_i32(ret_i32* out, int32_t op, dxr_res* t, int32_t mip, int32_t c0, int32_t c1, int32_t c2, int32_t o0, int32_t o1, int32_t o2)
{
	(void)op; (void)mip; (void)o0; (void)o1; (void)o2;
	memset(out, 0, sizeof(*out));
	size_t i;
	if (!tex_index(t, c0, c1, c2, &i)) return;
	if (t->fmt == DXF_R32U) out->v[0] = (int32_t)((const uint32_t*)t->data)[i];
	else { fprintf(stderr, "dxil_rt: textureLoad.i32 on format %d\n", t->fmt); abort(); }
}
int32_t rt_atomicBinOp_i32(int32_t op, dxr_res* t, int32_t aop, int32_t c0, int32_t c1, int32_t c2, int32_t v)
{
	(void)op;
	size_t i;
	if (t->fmt != DXF_R32U || !tex_index(t, c0, c1, c2, &i)) return 0;
	uint32_t* p = (uint32_t*)t->data + i;
	const uint32_t old = *p, u = (uint32_t)v;
	switch (aop)
	{
	case 0: *p = old + u; break; case 1: *p = old & u; break; case 2: *p = old | u; break; case 3: *p = old ^ u; break;
	case 4: *p = (uint32_t)((int32_t)old < v ? (int32_t)old : v); break; case 5: *p = (uint32_t)((int32_t)old > v ? (int32_t)old : v); break;
	case 6: *p = old < u ? old : u; break; case 7: *p = old > u ? old : u; break; case 8: *p = u; break;
	default: fprintf(stderr, "dxil_rt: atomicBinOp %d\n", aop); abort();
	}
	return (int32_t)old;
}
/* TextureCube: d3d_sampler.c's statement of the footprint (face selection, bilinear weights; the texel a tap across a face edge resolves to
 * is implementation-defined in D3D: the contract's rule, restated there) */
static void cube_taps(const dxr_res* t, float x, float y, float z, uint64_t idx[4], float w[4])
{
	const float dir[3] = { x, y, z };
	d3ds_cube_footprint(dir, t->w, idx, w);
}
static float cube_texel(const dxr_res* t, uint64_t i, int ch)
{
	if (t->fmt == DXF_RGBA16F) return d3ds_f16_to_f32(((const uint16_t*)t->data)[4 * i + ch]);
	if (t->fmt == DXF_R32F) return ch == 0 ? ((const float*)t->data)[i] : (ch == 3 ? 1.0f : 0.0f);
	fprintf(stderr, "dxil_rt: cube map format %d\n", t->fmt); abort();
}
void rt_textureGather_f32(ret_f32* out, int32_t op, dxr_res* t, dxr_res* smp, float c0, float c1, float c2, float c3, int32_t o0, int32_t o1, int32_t channel)
{
	(void)op; (void)smp; (void)c3; (void)o0; (void)o1;
	memset(out, 0, sizeof(*out));
	if (!t->data) return;
	if (t->kind != DXR_TEXCUBE) { fprintf(stderr, "dxil_rt: textureGather on kind %d\n", t->kind); abort(); }
	uint64_t idx[4]; float w[4];
	cube_taps(t, c0, c1, c2, idx, w);
	for (int k = 0; k < 4; ++k) out->v[k] = cube_texel(t, idx[k], channel);
}

/* ---------------- arithmetic ---------------- */
float rt_unary_f32(int32_t op, float x)
{
	switch (op)
	{
	case 6: return fabsf(x);
	case 7: return x != x ? 0.0f : (x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x));	/* Saturate */
	case 12: return cosf(x);
	case 13: return sinf(x);
	case 21: return exp2f(x);
	case 22: return x - floorf(x);	/* Frc */
	case 23: return log2f(x);
	case 24: return sqrtf(x);
	case 25: return 1.0f / sqrtf(x);	/* Rsqrt: two correctly rounded operations */
	case 26: return nearbyintf(x);
	case 27: return floorf(x);
	case 28: return ceilf(x);
	case 29: return truncf(x);
	default: fprintf(stderr, "dxil_rt: unary.f32 opcode %d\n", op); abort();
	}
}
float rt_binary_f32(int32_t op, float a, float b)
{
	switch (op)
	{
	case 35: return fmaxf(a, b);	/* FMax: the operand that is not a NaN */
	case 36: return fminf(a, b);
	default: fprintf(stderr, "dxil_rt: binary.f32 opcode %d\n", op); abort();
	}
}
int32_t rt_binary_i32(int32_t op, int32_t a, int32_t b)
{
	switch (op)
	{
	case 37: return a > b ? a : b;
	case 38: return a < b ? a : b;
	case 39: return (uint32_t)a > (uint32_t)b ? a : b;
	case 40: return (uint32_t)a < (uint32_t)b ? a : b;
	default: fprintf(stderr, "dxil_rt: binary.i32 opcode %d\n", op); abort();
	}
}
int32_t rt_unaryBits_i32(int32_t op, int32_t a)
{
	switch (op)
	{
	case 30: { uint32_t v = (uint32_t)a, r = 0; for (int i = 0; i < 32; ++i) { r = (r << 1) | (v & 1u); v >>= 1; } return (int32_t)r; }	/* Bfrev */
	case 31: return __builtin_popcount((uint32_t)a);	/* Countbits */
	case 32: return a ? __builtin_ctz((uint32_t)a) : -1;	/* FirstbitLo */
	case 33: return a ? 31 - __builtin_clz((uint32_t)a) : -1;	/* FirstbitHi (position from the LSB, as HLSL's firstbithigh returns) */
	default: fprintf(stderr, "dxil_rt: unaryBits.i32 opcode %d\n", op); abort();
	}
}
float rt_tertiary_f32(int32_t op, float a, float b, float c)
{
	switch (op)
	{
	case 46: return g_fused_mad ? fmaf(a, b, c) : a * b + c;	/* FMad: fused or not is the implementation's choice in D3D */
	case 47: return fmaf(a, b, c);
	default: fprintf(stderr, "dxil_rt: tertiary.f32 opcode %d\n", op); abort();
	}
}
int32_t rt_tertiary_i32(int32_t op, int32_t a, int32_t b, int32_t c)
{
	switch (op)
	{
	case 48: case 49: return (int32_t)((uint32_t)a * (uint32_t)b + (uint32_t)c);
	default: fprintf(stderr, "dxil_rt: tertiary.i32 opcode %d\n", op); abort();
	}
}
float rt_dot2_f32(int32_t op, float ax, float ay, float bx, float by) { (void)op; return rt_tertiary_f32(46, ay, by, ax * bx); }
float rt_dot3_f32(int32_t op, float ax, float ay, float az, float bx, float by, float bz) { (void)op; return rt_tertiary_f32(46, az, bz, rt_tertiary_f32(46, ay, by, ax * bx)); }
float rt_dot4_f32(int32_t op, float ax, float ay, float az, float aw, float bx, float by, float bz, float bw)
{
	(void)op;
	return rt_tertiary_f32(46, aw, bw, rt_tertiary_f32(46, az, bz, rt_tertiary_f32(46, ay, by, ax * bx)));
}

/* mode "fp16": the same operations on IEEE half operands (computed in fp32 and rounded once: exact for min / max / abs / saturate) */
_Float16 rt_unary_f16(int32_t op, _Float16 x) { return (_Float16)rt_unary_f32(op, (float)x); }
_Float16 rt_binary_f16(int32_t op, _Float16 a, _Float16 b) { return (_Float16)rt_binary_f32(op, (float)a, (float)b); }
_Float16 rt_tertiary_f16(int32_t op, _Float16 a, _Float16 b, _Float16 c) { return (_Float16)rt_tertiary_f32(op, (float)a, (float)b, (float)c); }

/* ---------------- wave intrinsics ---------------- */
int32_t rt_waveGetLaneIndex(int32_t op) { (void)op; return (int32_t)t_ctx->lane; }
int32_t rt_waveGetLaneCount(int32_t op) { (void)op; return t_ctx->wave ? (int32_t)t_ctx->wave->size : 1; }
/* ReadLaneAt through published registers (see cso2ll.py) */
void rt_publish_i32(int32_t id, int32_t v)
{
	dxr_wave* w = t_ctx->wave;
	if (!w || (uint32_t)id >= 256u) return;
	pthread_mutex_lock(&w->mu);
	w->pub_u[id][t_ctx->lane] = (uint32_t)v; w->pub_set[id][t_ctx->lane] = 1;
	pthread_cond_broadcast(&w->cv);
	pthread_mutex_unlock(&w->mu);
}
void rt_publish_f32(int32_t id, float v) { int32_t u; memcpy(&u, &v, 4); rt_publish_i32(id, u); }
int32_t rt_readlane_i32(int32_t id, int32_t own, int32_t lane)
{
	dxr_wave* w = t_ctx->wave;
	if (!w) return own;
	if ((uint32_t)lane >= w->size || (uint32_t)id >= 256u) return 0;
	pthread_mutex_lock(&w->mu);
	/* the wanted lane defines the value at the same place in the program: it has, or will, unless it has left the shader.
	 * While waiting this lane counts as blocked for nobody: it waits for a lane that is RUNNING, not for a meeting. */
	while (!w->pub_set[id][lane] && !w->exited[lane] && lane < (int32_t)w->size) pthread_cond_wait(&w->cv, &w->mu);
	const int32_t r = w->pub_set[id][lane] ? (int32_t)w->pub_u[id][lane] : 0;
	pthread_mutex_unlock(&w->mu);
	return r;
}
float rt_readlane_f32(int32_t id, float own, int32_t lane)
{
	int32_t o, r; memcpy(&o, &own, 4);
	r = rt_readlane_i32(id, o, lane);
	float f; memcpy(&f, &r, 4);
	return f;
}
/* (the stubs in the translated IR are always inlined: the return address identifies the shader's call site) */
#define SITE() __builtin_extract_return_addr(__builtin_return_address(0))
int32_t rt_waveReadLaneAt_i32(int32_t op, int32_t v, int32_t lane)
{
	(void)op;
	if (!t_ctx->wave) return v;
	wave_meet(SITE(), WK_READ, 0, (uint32_t)v, 0.0f, (uint32_t)lane);
	return (int32_t)t_ctx->wave->out_u[t_ctx->lane];	/* this lane's own slot: stable until it blocks again */
}
float rt_waveReadLaneAt_f32(int32_t op, float v, int32_t lane)
{
	(void)op;
	if (!t_ctx->wave) return v;
	wave_meet(SITE(), WK_READ, 0, 0u, v, (uint32_t)lane);
	return t_ctx->wave->out_f[t_ctx->lane];
}
int32_t rt_waveReadLaneFirst_i32(int32_t op, int32_t v)
{
	(void)op;
	if (!t_ctx->wave) return v;
	wave_meet(SITE(), WK_FIRST, 0, (uint32_t)v, 0.0f, 0u);
	return (int32_t)t_ctx->wave->out_u[t_ctx->lane];
}
typedef struct { int32_t v[4]; } ballot_t;
void rt_waveActiveBallot(ballot_t* out, int32_t op, int32_t cond)
{
	(void)op;
	memset(out, 0, sizeof(*out));
	if (!t_ctx->wave) { out->v[0] = cond ? 1 : 0; return; }
	wave_meet(SITE(), WK_BALLOT, 0, cond ? 1u : 0u, 0.0f, 0u);
	for (int i = 0; i < 4; ++i) out->v[i] = (int32_t)t_ctx->wave->ballot[t_ctx->lane][i];
}
float rt_waveActiveOp_f32(int32_t opc, float v, int32_t op, int32_t sign)
{
	(void)opc; (void)sign;
	if (!t_ctx->wave) return v;
	wave_meet(SITE(), WK_ACTIVE, op, 0u, v, 0u);
	return t_ctx->wave->out_f[t_ctx->lane];
}

int32_t rt_waveAllOp(int32_t op, int32_t cond)
{
	if (op != 135) abort();	/* WaveAllBitCount */
	if (!t_ctx->wave) return cond ? 1 : 0;
	wave_meet(SITE(), WK_COUNT, 0, cond ? 1u : 0u, 0.0f, 0u);
	return (int32_t)t_ctx->wave->out_u[t_ctx->lane];
}
int32_t rt_wavePrefixOp(int32_t op, int32_t cond)
{
	if (op != 136) abort();	/* WavePrefixBitCount */
	if (!t_ctx->wave) return 0;
	wave_meet(SITE(), WK_PREFIX, 0, cond ? 1u : 0u, 0.0f, 0u);
	return (int32_t)t_ctx->wave->out_u[t_ctx->lane];
}

/* ---------------- work-graph library (lib_6_x: Bin/LibRayMarch.cso) ----------------
 * Resources of a library are globals that hold handles (cso2ll.py translate_lib names them lib_<name>): descriptor slots here, bound
 * by name.  The node intrinsics are answered by a minimal scheduler: the records a node allocates are appended to one list, which
 * the test then feeds to the consumer node, one dispatch per record with the record's SV_DispatchGrid. */
#define LIB_SLOTS 4096
dxr_res* lib_g_smpLinear = &g_null_res; dxr_res* lib_g_smpPoint = &g_null_res; dxr_res* lib_g_txDepth = &g_null_res;
dxr_res* lib_g_roVolumes = &g_null_res; dxr_res* lib_g_rwVisibleVolumes = &g_null_res; dxr_res* lib_g_rwVolumes = &g_null_res;
dxr_res* lib_cbSampleRes = &g_null_res; dxr_res* lib_h_cbPerFrame = &g_null_res; dxr_res* lib_h_g_roPerObject = &g_null_res;
dxr_res* lib_g_txLightMaps[LIB_SLOTS]; dxr_res* lib_g_txGrids[LIB_SLOTS]; dxr_res* lib_g_rwCubeMaps[LIB_SLOTS]; dxr_res* lib_g_rwCubeDepths[LIB_SLOTS];
static const struct { const char* name; dxr_res** slot; uint32_t n; } g_lib_slots[] = {
	{ "g_smpLinear", &lib_g_smpLinear, 1 }, { "g_smpPoint", &lib_g_smpPoint, 1 }, { "g_txDepth", &lib_g_txDepth, 1 },
	{ "g_roVolumes", &lib_g_roVolumes, 1 }, { "g_rwVisibleVolumes", &lib_g_rwVisibleVolumes, 1 }, { "g_rwVolumes", &lib_g_rwVolumes, 1 },
	{ "cbSampleRes", &lib_cbSampleRes, 1 }, { "cbPerFrame", &lib_h_cbPerFrame, 1 }, { "g_roPerObject", &lib_h_g_roPerObject, 1 },
	{ "g_txLightMaps", lib_g_txLightMaps, LIB_SLOTS }, { "g_txGrids", lib_g_txGrids, LIB_SLOTS },
	{ "g_rwCubeMaps", lib_g_rwCubeMaps, LIB_SLOTS }, { "g_rwCubeDepths", lib_g_rwCubeDepths, LIB_SLOTS } };
/* every slot back to the null descriptor; the resources themselves are dxref_bind entries (dxref_reset forgets those) */
void dxref_lib_reset(void)
{
	for (size_t i = 0; i < sizeof(g_lib_slots) / sizeof(g_lib_slots[0]); ++i)
		for (uint32_t k = 0; k < g_lib_slots[i].n; ++k) g_lib_slots[i].slot[k] = &g_null_res;
}
/* slot `name`[index] = the resource bound LAST with dxref_bind */
int dxref_lib_bind(const char* name, uint32_t index)
{
	if (!g_nbind) return -1;
	for (size_t i = 0; i < sizeof(g_lib_slots) / sizeof(g_lib_slots[0]); ++i)
		if (!strcmp(g_lib_slots[i].name, name))
		{
			if (index >= g_lib_slots[i].n) return -2;
			g_lib_slots[i].slot[index] = &g_bind[g_nbind - 1].res;
			return 0;
		}
	return -3;
}
uint32_t dxref_last_counter(void) { return g_nbind ? g_bind[g_nbind - 1].res.counter : 0; }

#define NODE_MAX_RECORD_BYTES (1u << 22)
static const void* g_node_in;
static uint8_t* g_node_out;
static uint32_t g_node_out_stride, g_node_out_count;
/* before a node's dispatch: its input record, and the size of the records its output allocates */
void dxref_node_begin(const void* input_record, uint32_t output_stride)
{
	g_node_in = input_record;
	g_node_out_stride = output_stride;
	if (!g_node_out) g_node_out = (uint8_t*)malloc(NODE_MAX_RECORD_BYTES);
}
void dxref_node_clear_output(void) { g_node_out_count = 0; }
uint32_t dxref_node_records(const void** base) { *base = g_node_out; return g_node_out_count; }
static void* node_alloc_records(uint32_t count)
{
	if (((size_t)g_node_out_count + count) * g_node_out_stride > NODE_MAX_RECORD_BYTES) abort();
	void* p = g_node_out + (size_t)g_node_out_count * g_node_out_stride;
	g_node_out_count += count;
	return p;
}
void rt_createNodeInputRecordHandle(void** out, int32_t op, int32_t index) { (void)op; (void)index; *out = (void*)g_node_in; }
void rt_createNodeOutputHandle(void** out, int32_t op, int32_t index) { (void)op; (void)index; *out = &g_node_out_stride; }
void rt_allocateNodeOutputRecords(void** out, int32_t op, void* node, int32_t count, int32_t per_thread)
{
	(void)op; (void)node;
	if (per_thread || !t_ctx->wave) abort();	/* the library only uses GetGroupNodeOutputRecords, from a one-wave group */
	wave_meet(SITE(), WK_ALLOC, 0, (uint32_t)count, 0.0f, 0u);
	*out = t_ctx->wave->out_p[t_ctx->lane];
}
void rt_outputComplete(int32_t op, void* records) { (void)op; (void)records; }

/* ---------------- dispatch ---------------- */
/* n invocations of a pixel (or vertex) shader, one after the other: inputs[i] = 8 signature elements x 4 components (raw bits),
 * outputs[i] = 4 floats of output element 0 (may be NULL); dxref_run_invocations_full returns all 8 x 4 output components */
int dxref_run_invocations(void (*fn)(void), uint32_t n, const uint32_t* inputs, float* outputs)
{
	dxr_ctx c;
	memset(&c, 0, sizeof(c));
	t_ctx = &c;
	for (uint32_t i = 0; i < n; ++i)
	{
		c.in = (const uint32_t (*)[4])(inputs + (size_t)i * 32u);
		memset(c.out, 0, sizeof(c.out));
		fn();
		if (outputs) memcpy(outputs + (size_t)i * 4u, c.out[0], 16);
	}
	return 0;
}
int dxref_run_invocations_full(void (*fn)(void), uint32_t n, const uint32_t* inputs, float* outputs)
{
	dxr_ctx c;
	memset(&c, 0, sizeof(c));
	t_ctx = &c;
	for (uint32_t i = 0; i < n; ++i)
	{
		c.in = (const uint32_t (*)[4])(inputs + (size_t)i * 32u);
		memset(c.out, 0, sizeof(c.out));
		fn();
		memcpy(outputs + (size_t)i * 32u, c.out, sizeof(c.out));
	}
	return 0;
}

typedef void (*shader_fn)(void);
typedef struct { shader_fn fn; dxr_ctx ctx; } lane_arg;
static void* lane_main(void* p)
{
	lane_arg* a = (lane_arg*)p;
	t_ctx = &a->ctx;
	a->fn();
	if (a->ctx.wave) wave_lane_exit(a->ctx.wave);
	return NULL;
}

/* Runs fn for every thread of a gx x gy x gz dispatch of (tx, ty, tz)-thread groups.  wave_size = 0: lanes one after the other on
 * the calling thread (shaders whose wave intrinsics only broadcast group-uniform values: ReadLaneAt / ReadLaneFirst return the
 * caller's value).  wave_size > 0: the group is cut into waves of that many lanes, each lane a thread, meeting at wave intrinsics
 * (see "wavefront rendezvous"; `cluster`: lanes per WaveActive* reduction group, 0 = the whole wave). */
/* wave_size == 0: the groups are independent (no wave intrinsic that couples lanes, no group-shared memory in these shaders) and are
 * spread over `dxref_set_threads` host threads */
static int g_threads = 1;
void dxref_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
typedef struct { shader_fn fn; uint32_t gx, gy, gz, tx, ty, tz; uint64_t first, last; uint64_t* next; } group_range;
static void run_group_serial(shader_fn fn, uint32_t x, uint32_t y, uint32_t z, uint32_t tx, uint32_t ty, uint32_t tz)
{
	dxr_ctx c;
	memset(&c, 0, sizeof(c));
	t_ctx = &c;
	const uint32_t per_group = tx * ty * tz;
	for (uint32_t f = 0; f < per_group; ++f)
	{
		c.flat = f; c.lane = 0; c.wave = NULL;
		c.tig[0] = f % tx; c.tig[1] = (f / tx) % ty; c.tig[2] = f / (tx * ty);
		c.gid[0] = x; c.gid[1] = y; c.gid[2] = z;
		c.tid[0] = x * tx + c.tig[0]; c.tid[1] = y * ty + c.tig[1]; c.tid[2] = z * tz + c.tig[2];
		fn();
	}
}
static void* group_worker(void* p)
{
	const group_range* r = (const group_range*)p;
	for (;;)
	{
		/* chunks of 8 groups from a shared counter: the cost of a group varies with what its rays meet */
		const uint64_t g0 = r->next ? __atomic_fetch_add(r->next, 8u, __ATOMIC_RELAXED) : r->first;
		const uint64_t g1 = r->next ? (g0 + 8u < r->last ? g0 + 8u : r->last) : r->last;
		for (uint64_t g = g0; g < g1; ++g)
			run_group_serial(r->fn, (uint32_t)(g % r->gx), (uint32_t)((g / r->gx) % r->gy), (uint32_t)(g / ((uint64_t)r->gx * r->gy)), r->tx, r->ty, r->tz);
		if (!r->next || g1 >= r->last) break;
	}
	return NULL;
}

int dxref_dispatch(shader_fn fn, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t tx, uint32_t ty, uint32_t tz, uint32_t wave_size, uint32_t cluster)
{
	const uint32_t per_group = tx * ty * tz;
	if (wave_size > 64) return -1;
	if (!wave_size)
	{
		const uint64_t groups = (uint64_t)gx * gy * gz;
		const int nt = (uint64_t)g_threads < groups ? g_threads : (int)groups;
		if (nt <= 1) { group_range r = { fn, gx, gy, gz, tx, ty, tz, 0, groups, NULL }; group_worker(&r); return 0; }
		static group_range ranges[256];
		pthread_t th[256];
		uint64_t next = 0;
		for (int t = 0; t < nt; ++t)
		{
			const group_range r = { fn, gx, gy, gz, tx, ty, tz, 0, groups, &next };
			ranges[t] = r;
			if (pthread_create(&th[t], NULL, group_worker, &ranges[t]) != 0) return -2;
		}
		for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
		return 0;
	}
	for (uint32_t z = 0; z < gz; ++z)
		for (uint32_t y = 0; y < gy; ++y)
			for (uint32_t x = 0; x < gx; ++x)
			{
				for (uint32_t base = 0; base < per_group; base += wave_size)
				{
					const uint32_t n = per_group - base < wave_size ? per_group - base : wave_size;
					dxr_wave w;
					memset(&w, 0, sizeof(w));
					pthread_mutex_init(&w.mu, NULL); pthread_cond_init(&w.cv, NULL);
					w.size = wave_size; w.running = n; w.cluster = cluster;
					lane_arg args[64];
					pthread_t th[64];
					for (uint32_t l = 0; l < n; ++l)
					{
						const uint32_t f = base + l;
						memset(&args[l], 0, sizeof(args[l]));
						args[l].fn = fn;
						dxr_ctx* c = &args[l].ctx;
						c->flat = f; c->lane = l; c->wave = &w;
						c->tig[0] = f % tx; c->tig[1] = (f / tx) % ty; c->tig[2] = f / (tx * ty);
						c->gid[0] = x; c->gid[1] = y; c->gid[2] = z;
						c->tid[0] = x * tx + c->tig[0]; c->tid[1] = y * ty + c->tig[1]; c->tid[2] = z * tz + c->tig[2];
						if (pthread_create(&th[l], NULL, lane_main, &args[l]) != 0) return -2;
					}
					for (uint32_t l = 0; l < n; ++l) pthread_join(th[l], NULL);
					pthread_mutex_destroy(&w.mu); pthread_cond_destroy(&w.cv);
				}
			}
	return 0;
}
