"""TEST INFRASTRUCTURE: runs the reference's SHIPPED compiled shaders (DXIL from /root/reference/Bin/*.cso, translated to x86-64 by
oracle/dxil/cso2ll.py into oracle/_ref/libmv_dxref.so) on the same inputs as the oracle's functions, with the same signatures, so
that tests can compare the oracle's restatement with the reference's own compiled code.

What is the reference's: control flow, arithmetic, constants, order of operations, early-outs of every shader.
What is NOT: the fixed-function stages -- sampler, format conversion of typed loads / stores -- answered by oracle/dxil/d3d_sampler.c, a
statement of the D3D11.3 rules written independently of the oracle's sampler (the library does not link libmv_oracle.so).
Register ranges below are the shaders' own (dx.resources metadata of each .cso, printed by `python oracle/dxil/cso2ll.py`).
"""
import ctypes as C
import os

import numpy as np

from . import mv_oracle as O

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.environ.get("MV_DXREF_LIB") or os.path.join(_DIR, "_ref", "libmv_dxref.so")  # (MV_DXREF_LIB: scripts/sanitize_cpu.sh, the runtime built with ASan / UBSan)
_lib = None

SRV, UAV, CBV, SAMPLER = 0, 1, 2, 3
RAW, TYPED_BUF, CBUF, TEX2D, TEX3D, TEX2DARRAY, SAMPLER_KIND = 1, 2, 3, 4, 5, 6, 7
F_NONE, F_RGBA16F, F_R11G11B10F, F_R32F, F_D16, F_RGBA16U, F_R32U = 0, 1, 2, 3, 4, 5, 6


def available():
    return os.path.exists(_PATH)


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(_PATH, mode=C.RTLD_GLOBAL)
        lib.dxref_set_sampler_options.restype = None
        lib.dxref_set_sampler_options.argtypes = [C.c_int, C.c_int]
        lib.dxref_bind.restype = C.c_int
        lib.dxref_bind.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.dxref_dispatch.restype = C.c_int
        lib.dxref_dispatch.argtypes = [C.c_void_p] + [C.c_uint32] * 8
        lib.dxref_counter.restype = C.c_uint32
        lib.dxref_counter.argtypes = [C.c_uint32] * 3
        _lib = lib
    return _lib


def set_sampler_options(frac_bits=0, r11_truncate=False):
    """What D3D lets a device do differently in the fixed-function stages: texel coordinates snapped to `frac_bits` fractional bits
    (the specification requires >= 8; 0 = exact fp32 weights, the contract), float11 / float10 stores truncated instead of rounded."""
    load().dxref_set_sampler_options(int(frac_bits), 1 if r11_truncate else 0)


def _bind(cls, rng, index, kind, fmt, arr, w=0, h=0, d=0, stride=0):
    ptr = arr.ctypes.data if arr is not None else None
    assert load().dxref_bind(cls, rng, index, kind, fmt, ptr, w, h, d, stride) == 0


def _entry(shader, mode):
    return C.cast(getattr(load(), f"dxil_{shader}_{mode}"), C.c_void_p)


def _dispatch(shader, mode, groups, threads, wave=0, cluster=0):
    rc = load().dxref_dispatch(_entry(shader, mode), groups[0], groups[1], groups[2], threads[0], threads[1], threads[2], wave, cluster)
    assert rc == 0, rc


def _bind_cube_targets(N, V, cm, cd):
    """RWTexture2DArray g_rwCubeMaps[] / g_rwCubeDepths[] (u0.., space 0 / 1): descriptor 5 v + mip, six slices of (N >> mip)^2."""
    for v in range(V):
        for mip in range(O.MVO_NUM_CUBE_MIP if hasattr(O, "MVO_NUM_CUBE_MIP") else 5):
            dim = max(1, N >> mip)   # (a mip's edge as D3D defines it: what GetDimensions answers)
            off = O.cube_mip_offset(N, mip)
            _bind(UAV, 0, 5 * v + mip, TEX2DARRAY, F_RGBA16F, cm[v][off:], dim, dim, 6)
            _bind(UAV, 1, 5 * v + mip, TEX2DARRAY, F_R32F, cd[v][off:], dim, dim, 6)


def march_view(V, N, L, W, H, volumes, lightMaps, depthMap, perObject, cbPerFrame, volumeInfo, cubeMapVolumes,
               cubeMaps=None, cubeDepths=None, mode="fp32", fused_mad=False):
    """CSRayMarchV.cso over the cube-map volume list: same arguments and outputs as mv_oracle.march_view (no counters).
    Dispatch = (N / 8, N / 8, number of cube-map volumes) groups of 8 x 8 x 6 threads (MultiRayCaster.cpp:1329-1368)."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_set_fused_mad(1 if fused_mad else 0)
    chain = O.cube_chain_texels(N)
    vols = [np.ascontiguousarray(v).view(np.uint16) for v in volumes]
    cm = np.zeros((V, chain, 4), np.uint16) if cubeMaps is None else np.ascontiguousarray(cubeMaps, np.uint16).copy()
    cd = np.zeros((V, chain), np.float32) if cubeDepths is None else np.ascontiguousarray(cubeDepths, np.float32).copy()
    lm = np.ascontiguousarray(lightMaps, np.uint32)
    dm = np.ascontiguousarray(depthMap, np.float32)
    po = np.ascontiguousarray(perObject).view(np.uint8).copy()
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8).copy()
    vi = np.ascontiguousarray(volumeInfo, np.uint16)
    cl = np.ascontiguousarray(cubeMapVolumes, np.uint32)
    # SRVs (CSRayMarchV.cso dx.resources): range 0 = Texture3D light maps t3.. space0 (index 3 + volume), range 1 = Texture3D grids
    # space1 (index = VolTexId), range 2 = depth Texture2D, range 3 = PerObject structured (224 B), range 4 = volume list (uint),
    # range 5 = VolumeInfo typed buffer (R16G16B16A16_UINT)
    for v in range(V):
        _bind(SRV, 0, 3 + v, TEX3D, F_R11G11B10F, lm[v], L, L, L)
    for i, g in enumerate(vols):
        _bind(SRV, 1, i, TEX3D, F_RGBA16F, g, N, N, N)
    _bind(SRV, 2, 0, TEX2D, F_R32F, dm, W, H, 1)
    _bind(SRV, 3, 0, RAW, F_NONE, po, V, 0, 0, 224)
    _bind(SRV, 4, 1, RAW, F_NONE, cl, len(cl), 0, 0, 4)
    _bind(SRV, 5, 2, TYPED_BUF, F_RGBA16U, vi, V)
    _bind_cube_targets(N, V, cm, cd)
    _bind(CBV, 0, 0, CBUF, F_NONE, cb, len(cb))
    _bind(SAMPLER, 0, 0, SAMPLER_KIND, F_NONE, None)
    _bind(SAMPLER, 1, 1, SAMPLER_KIND, F_NONE, None)
    g = (N + 7) // 8
    if len(cl):
        _dispatch("CSRayMarchV", mode, (g, g, len(cl)), (8, 8, 6))
    return cm, cd


def march_light(V, N, L, M, volumes, descs, perObject, cbPerFrame, shadowMap, sh27, visible, maxLightSamples, lightMaps=None,
                mode="fp32", fused_mad=False):
    """CSRayMarchL.cso (the shipped one has _HAS_SHADOW_MAP_ and _HAS_LIGHT_PROBE_ compiled in; the probe is switched by
    cbSampleRes.g_hasLightProbe): same arguments as mv_oracle.march_light; returns the light maps [V, L, L, L] u32."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_set_fused_mad(1 if fused_mad else 0)
    vols = [np.ascontiguousarray(v).view(np.uint16) for v in volumes]
    lm = np.zeros((V, L, L, L), np.uint32) if lightMaps is None else np.ascontiguousarray(lightMaps, np.uint32).copy()
    po = np.ascontiguousarray(perObject).view(np.uint8).copy()
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8).copy()
    d = np.ascontiguousarray(descs, np.uint32)
    sm = np.ascontiguousarray(shadowMap, np.uint16)
    vis = np.ascontiguousarray(visible, np.uint32)
    cnt = np.array([len(vis)], np.uint32)
    if len(vis) == 0:
        vis = np.zeros(1, np.uint32)
    sh = np.zeros((9, 3), np.float32) if sh27 is None else np.ascontiguousarray(sh27, np.float32).reshape(9, 3).copy()
    cb1 = np.array([maxLightSamples, 0 if sh27 is None else 1, 0, 0], np.uint32)
    # SRVs (CSRayMarchL.cso dx.resources): range 0 = grids (space1), 1 = shadow map (space2 t0), 2 = SH coefficients (space2 t1),
    # 3 = PerObject, 4 = VolumeDesc, 5 = visible volumes, 6 = visible volume count; UAV range 0 = light maps; CBV 0 / 1
    for i, g in enumerate(vols):
        _bind(SRV, 0, i, TEX3D, F_RGBA16F, g, N, N, N)
    _bind(SRV, 1, 0, TEX2D, F_D16, sm, M, M, 1)
    _bind(SRV, 2, 1, RAW, F_NONE, sh, 9, 0, 0, 12)
    _bind(SRV, 3, 0, RAW, F_NONE, po, V, 0, 0, 224)
    _bind(SRV, 4, 1, RAW, F_NONE, d, V, 0, 0, 4)
    _bind(SRV, 5, 2, RAW, F_NONE, vis, len(vis), 0, 0, 4)
    _bind(SRV, 6, 3, RAW, F_NONE, cnt, 1, 0, 0, 4)
    for v in range(V):
        _bind(UAV, 0, v, TEX3D, F_R11G11B10F, lm[v], L, L, L)
    _bind(CBV, 0, 0, CBUF, F_NONE, cb, len(cb))
    _bind(CBV, 1, 1, CBUF, F_NONE, cb1, 8)
    _bind(SAMPLER, 0, 0, SAMPLER_KIND, F_NONE, None)
    _bind(SAMPLER, 1, 1, SAMPLER_KIND, F_NONE, None)
    g = (L + 3) // 4
    _dispatch("CSRayMarchL", mode, (g, g, g), (4, 4, 4))
    return lm


def cull(V, perObject, descs, cbPerFrame, maxRaySamples, volumeInfo=None, mode="fp32", fused_mad=False):
    """CSVolumeCull.cso: groups of 8 x 4 threads = four volumes x eight cube vertices in one 32-lane wave (the shader's wave intrinsics
    need the lanes to run side by side: dxref_dispatch's wave mode).  Same arguments and results as mv_oracle.cull; the two append
    buffers come back sorted (on a GPU their order is whatever the atomics' order was)."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_set_fused_mad(1 if fused_mad else 0)
    info = np.zeros((V, 4), np.uint16) if volumeInfo is None else np.ascontiguousarray(volumeInfo, np.uint16).copy()
    vis = np.zeros(max(V, 1), np.uint32)
    cube = np.zeros(max(V, 1), np.uint32)
    po = np.ascontiguousarray(perObject).view(np.uint8).copy()
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8).copy()
    d = np.ascontiguousarray(descs, np.uint32)
    cb1 = np.array([maxRaySamples, 0, 0, 0], np.uint32)
    _bind(SRV, 0, 0, RAW, F_NONE, po, V, 0, 0, 224)
    _bind(SRV, 1, 1, RAW, F_NONE, d, V, 0, 0, 4)
    _bind(UAV, 0, 0, RAW, F_NONE, vis, V, 0, 0, 4)
    _bind(UAV, 1, 1, TYPED_BUF, F_RGBA16U, info, V)
    _bind(UAV, 2, 2, RAW, F_NONE, cube, V, 0, 0, 4)
    _bind(CBV, 0, 0, CBUF, F_NONE, cb, len(cb))
    _bind(CBV, 1, 1, CBUF, F_NONE, cb1, 8)
    _dispatch("CSVolumeCull", mode, ((V + 3) // 4, 1, 1), (8, 4, 1), wave=32, cluster=8)
    nv, nc = int(lib.dxref_counter(UAV, 0, 0)), int(lib.dxref_counter(UAV, 2, 2))
    return info, np.sort(vis[:nv]), np.sort(cube[:nc])


def lib_ray_march(V, N, L, W, H, volumes, lightMaps, depthMap, perObject, descs, cbPerFrame, maxRaySamples, volumeInfo=None,
                  cubeMaps=None, cubeDepths=None, mode="fp32", fused_mad=False):
    """Bin/LibRayMarch.cso, the work-graph statement of cull + view march (LibRayMarch.hlsl:39-134, 202-287), scheduled as
    MultiRayCaster.cpp:1421-1428 dispatches it: the entry node VolumeCull over DIV_UP(V, 4) groups of 8 x 4 threads (one 32-lane
    wave) with the record { V }; every RayMarchRecord it emits starts the RayMarch node over the record's SV_DispatchGrid of
    8 x 8 x 6-thread groups.  Returns (volumeInfo [V,4] u16, visible volumes sorted, records [n,7] u32 sorted by volume,
    cubeMaps, cubeDepths)."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_lib_reset()
    lib.dxref_set_fused_mad(1 if fused_mad else 0)
    chain = O.cube_chain_texels(N)
    keep = []
    slot = [0]

    def bind(name, index, kind, fmt, arr, w=0, h=0, d=0, stride=0):
        keep.append(arr)
        _bind(SRV, 99, slot[0], kind, fmt, arr, w, h, d, stride)
        slot[0] += 1
        assert lib.dxref_lib_bind(name.encode(), index) == 0, name

    vols = [np.ascontiguousarray(v).view(np.uint16) for v in volumes]
    cm = np.zeros((V, chain, 4), np.uint16) if cubeMaps is None else np.ascontiguousarray(cubeMaps, np.uint16).copy()
    cd = np.zeros((V, chain), np.float32) if cubeDepths is None else np.ascontiguousarray(cubeDepths, np.float32).copy()
    lm = np.ascontiguousarray(lightMaps, np.uint32)
    dm = np.ascontiguousarray(depthMap, np.float32)
    po = np.ascontiguousarray(perObject).view(np.uint8).copy()
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8).copy()
    d = np.ascontiguousarray(descs, np.uint32)
    info = np.zeros((V, 4), np.uint16) if volumeInfo is None else np.ascontiguousarray(volumeInfo, np.uint16).copy()
    vis = np.zeros(max(V, 1), np.uint32)
    cb1 = np.array([maxRaySamples, 0, 0, 0], np.uint32)
    for v in range(V):
        bind("g_txLightMaps", v, TEX3D, F_R11G11B10F, lm[v], L, L, L)
    for i, g in enumerate(vols):
        bind("g_txGrids", i, TEX3D, F_RGBA16F, g, N, N, N)
    bind("g_txDepth", 0, TEX2D, F_R32F, dm, W, H, 1)
    bind("g_roPerObject", 0, RAW, F_NONE, po, V, 0, 0, 224)
    bind("g_roVolumes", 0, RAW, F_NONE, d, V, 0, 0, 4)
    bind("g_rwVolumes", 0, TYPED_BUF, F_RGBA16U, info, V)
    bind("cbPerFrame", 0, CBUF, F_NONE, cb, len(cb))
    bind("cbSampleRes", 0, CBUF, F_NONE, cb1, 8)
    bind("g_smpLinear", 0, SAMPLER_KIND, F_NONE, None)
    bind("g_smpPoint", 0, SAMPLER_KIND, F_NONE, None)
    for v in range(V):
        for mip in range(5):
            dim = max(1, N >> mip)   # (a mip's edge as D3D defines it: what GetDimensions answers)
            off = O.cube_mip_offset(N, mip)
            bind("g_rwCubeMaps", 5 * v + mip, TEX2DARRAY, F_RGBA16F, cm[v][off:], dim, dim, 6)
            bind("g_rwCubeDepths", 5 * v + mip, TEX2DARRAY, F_R32F, cd[v][off:], dim, dim, 6)
    bind("g_rwVisibleVolumes", 0, RAW, F_NONE, vis, V, 0, 0, 4)   # bound last: dxref_last_counter reads its append counter
    lib.dxref_node_begin.argtypes = [C.c_void_p, C.c_uint32]
    lib.dxref_node_records.argtypes = [C.POINTER(C.c_void_p)]
    lib.dxref_node_records.restype = C.c_uint32
    lib.dxref_last_counter.restype = C.c_uint32
    count = np.array([V], np.uint32)
    lib.dxref_node_clear_output()
    lib.dxref_node_begin(count.ctypes.data, 28)
    _dispatch("LibRayMarch_VolumeCull", mode, ((V + 3) // 4, 1, 1), (8, 4, 1), wave=32, cluster=8)
    base = C.c_void_p()
    n = int(lib.dxref_node_records(C.byref(base)))
    recs = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint32)), shape=(max(n, 1), 7))[:n].copy()
    nv = int(lib.dxref_last_counter())
    for r in recs:
        lib.dxref_node_begin(r.ctypes.data, 0)
        _dispatch("LibRayMarch_RayMarch", mode, (int(r[0]), int(r[1]), 1), (8, 8, 6))
    return info, np.sort(vis[:nv]), recs[np.argsort(recs[:, 2], kind="stable")], cm, cd


TEXCUBE = 8


def composite(V, N, L, W, H, volumes, lightMaps, depthMap, perObject, cbPerFrame, volumeInfo, visible, cubeMaps, cubeDepths,
              mode="fp32", fused_mad=False, fragments=None):
    """The K-buffer OIT composite through the reference's three compiled pixel shaders: PSDepthPeel.cso (InterlockedMin peel into 8
    depth layers), PSCube.cso (CubeCast or the screen-space RayCast per fragment, stored into the layer whose depth matches) and
    PSResolveOIT.cso (front-to-back resolve), one invocation per fragment / pixel.  What the rasteriser would feed them -- one
    fragment per visible cube where the pixel's view ray leaves it, with VSCube's interpolated outputs -- comes from the oracle's
    analytic stand-in (mvo_pixel_fragments; UVW.xy = the exit point's cube-face (s, t) as VSCube.hlsl:69 defines it): that part,
    the TextureCube footprint across face edges and the samplers are the oracle's contract, everything inside the shaders is the
    reference's.  `fragments`: a list of (px, py, volume, lpt[3], z, u, v, face) to use instead (rasterised_fragments below).
    Returns the RGBA16F image [H, W, 4] u16 like mv_oracle.composite."""
    lib = load()
    olib = O.load()
    lib.dxref_run_invocations.restype = C.c_int
    lib.dxref_run_invocations.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    olib.mvo_pixel_fragments.restype = C.c_uint32
    olib.mvo_pixel_fragments.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 3
    olib.mvo_cube_face_coords.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    chain = O.cube_chain_texels(N)
    vols = [np.ascontiguousarray(v).view(np.uint16) for v in volumes]
    lm = np.ascontiguousarray(lightMaps, np.uint32)
    dm = np.ascontiguousarray(depthMap, np.float32)
    po = np.ascontiguousarray(perObject).view(np.uint8).copy()
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8).copy()
    vi = np.ascontiguousarray(volumeInfo, np.uint16)
    vis = np.ascontiguousarray(visible, np.uint32)
    cm = np.ascontiguousarray(cubeMaps, np.uint16)
    cd = np.ascontiguousarray(cubeDepths, np.float32)

    # fragments of every pixel (list order within a pixel)
    frags = list(fragments) if fragments is not None else None
    if frags is None:
        vol_buf, lpt_buf, z_buf = np.zeros(max(V, 1), np.uint32), np.zeros((max(V, 1), 3), np.float32), np.zeros(max(V, 1), np.float32)
        frags = []
        face, st = C.c_uint32(), np.zeros(2, np.float32)
        for py in range(H):
            for px in range(W):
                n = olib.mvo_pixel_fragments(W, H, px, py, po.ctypes.data, cb.ctypes.data, vis.ctypes.data, len(vis),
                                             vol_buf.ctypes.data, lpt_buf.ctypes.data, z_buf.ctypes.data)
                for k in range(n):
                    olib.mvo_cube_face_coords(lpt_buf[k].ctypes.data, C.byref(face), st.ctypes.data)
                    frags.append((px, py, int(vol_buf[k]), lpt_buf[k].copy(), float(z_buf[k]), float(st[0]), float(st[1]), int(face.value)))
    nf = len(frags)
    inputs = np.zeros((max(nf, 1), 8, 4), np.uint32)
    fv = inputs.view(np.float32)
    for i, (px, py, v, lpt, z, s, t, f) in enumerate(frags):
        fv[i, 0] = (px + 0.5, py + 0.5, z, 1.0)                      # SV_Position
        fv[i, 1, :3] = (s, t, float(f))                              # TEXCOORD = UVW
        fv[i, 2, :3] = lpt                                           # POSLOCAL
        inputs[i, 3, 0] = v                                          # VOLUMEID
        inputs[i, 4, 0] = 5 * v + int(vi[v, 0])                      # SRVINDEX = NUM_CUBE_MIP * volume + mip (VSCube.hlsl:73)
        inputs[i, 5, 0] = int(vi[v, 3])                              # VOLTEXID
        inputs[i, 6, 0] = 0 if (int(vi[v, 2]) & 0x8000) else int(vi[v, 1])  # SAMPLECOUNT (0 = cube-map scheme)

    k_depths = np.full((8, H, W), 0xffffffff, np.uint32)
    k_colors = np.zeros((8, H, W, 4), np.uint16)
    lib.dxref_set_fused_mad(1 if fused_mad else 0)

    # pass 1: depth peel
    lib.dxref_reset()
    _bind(UAV, 0, 0, TEX2DARRAY, F_R32U, k_depths, W, H, 8)
    if nf:
        assert lib.dxref_run_invocations(_entry("PSDepthPeel", mode), nf, inputs.ctypes.data, None) == 0

    # pass 2: colour of every fragment into its layer
    lib.dxref_reset()
    for v in range(V):
        _bind(SRV, 0, 3 + v, TEX3D, F_R11G11B10F, lm[v], L, L, L)
        for mip in range(5):
            dim = max(1, N >> mip)
            off = O.cube_mip_offset(N, mip)
            _bind(SRV, 4, 5 * v + mip, TEXCUBE, F_RGBA16F, cm[v][off:], dim, dim, 6)
            _bind(SRV, 5, 5 * v + mip, TEXCUBE, F_R32F, cd[v][off:], dim, dim, 6)
    for i, g in enumerate(vols):
        _bind(SRV, 1, i, TEX3D, F_RGBA16F, g, N, N, N)
    _bind(SRV, 2, 0, TEX2D, F_R32F, dm, W, H, 1)
    _bind(SRV, 3, 0, RAW, F_NONE, po, V, 0, 0, 224)
    _bind(SRV, 6, 1, TEX2DARRAY, F_R32U, k_depths, W, H, 8)
    _bind(UAV, 0, 0, TEX2DARRAY, F_RGBA16F, k_colors, W, H, 8)
    _bind(CBV, 0, 0, CBUF, F_NONE, cb, len(cb))
    _bind(SAMPLER, 0, 0, SAMPLER_KIND, F_NONE, None)
    if nf:
        assert lib.dxref_run_invocations(_entry("PSCube", mode), nf, inputs.ctypes.data, None) == 0

    # pass 3: resolve, one invocation per pixel
    lib.dxref_reset()
    _bind(SRV, 0, 0, TEX2DARRAY, F_RGBA16F, k_colors, W, H, 8)
    pin = np.zeros((W * H, 8, 4), np.uint32)
    pf = pin.view(np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    pf[:, 0, 0] = xs.ravel() + 0.5
    pf[:, 0, 1] = ys.ravel() + 0.5
    pf[:, 0, 3] = 1.0
    out = np.zeros((W * H, 4), np.float32)
    assert lib.dxref_run_invocations(_entry("PSResolveOIT", mode), W * H, pin.ctypes.data, out.ctypes.data) == 0
    img = np.array([olib.mvo_f32_to_f16(float(x)) for x in out.ravel()], np.uint16).reshape(H, W, 4)
    return img


def init_grid_data(N, mode="fp32"):
    """CSInitGridData.cso into an RGBA16F volume of edge N (groups of 4 x 4 x 4): same result as mv_oracle.init_grid_data."""
    lib = load()
    lib.dxref_reset()
    out = np.zeros((N, N, N, 4), np.uint16)
    _bind(UAV, 0, 0, TEX3D, F_RGBA16F, out, N, N, N)
    g = (N + 3) // 4
    _dispatch("CSInitGridData", mode, (g, g, g), (4, 4, 4))
    return out


def r32f_to_rgba16f(src, N, mode="fp32"):
    """CSR32FToRGBA16F.cso: an R32F volume [d, h, w] resampled (linear sampler) into an RGBA16F volume of edge N."""
    lib = load()
    lib.dxref_reset()
    src = np.ascontiguousarray(src, np.float32)
    d, h, w = src.shape
    out = np.zeros((N, N, N, 4), np.uint16)
    _bind(SRV, 0, 0, TEX3D, F_R32F, src, w, h, d)
    _bind(UAV, 0, 0, TEX3D, F_RGBA16F, out, N, N, N)
    _bind(SAMPLER, 0, 0, SAMPLER_KIND, F_NONE, None)
    g = (N + 3) // 4
    _dispatch("CSR32FToRGBA16F", mode, (g, g, g), (4, 4, 4))
    return out


def vs_cube(V, perObject, volumeInfo, visible, mode="fp32"):
    """VSCube.cso for the 24 vertices of every visible cube: returns [len(visible), 24, 8, 4] raw output components (uint32 view of
    float / int): element 0 SV_Position, 1 UVW, 2 LPt, 3 VolId, 4 SrvId, 5 TexId, 6 SmpCnt (VSCube.hlsl:11-20)."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_run_invocations_full.restype = C.c_int
    lib.dxref_run_invocations_full.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    po = np.ascontiguousarray(perObject).view(np.uint8).copy()
    vi = np.ascontiguousarray(volumeInfo, np.uint16)
    vis = np.ascontiguousarray(visible, np.uint32)
    _bind(SRV, 0, 0, RAW, F_NONE, po, V, 0, 0, 224)
    _bind(SRV, 1, 1, RAW, F_NONE, vis, len(vis), 0, 0, 4)
    _bind(SRV, 2, 2, TYPED_BUF, F_RGBA16U, vi, V)
    n = len(vis) * 24
    inputs = np.zeros((n, 8, 4), np.uint32)
    for i in range(len(vis)):
        for vid in range(24):
            inputs[i * 24 + vid, 0, 0] = vid  # SV_VertexID
            inputs[i * 24 + vid, 1, 0] = i    # SV_InstanceID
    out = np.zeros((n, 8, 4), np.float32)
    assert lib.dxref_run_invocations_full(_entry("VSCube", mode), n, inputs.ctypes.data, out.ctypes.data) == 0
    return out.view(np.uint32).reshape(len(vis), 24, 8, 4)


def vs_depth(positions, wvp_t, mode="fp32", fused_mad=False):
    """VSDepth.cso (the shadow pass's vertex shader, VSDepth.hlsl:24-27) for every vertex: SV_Position [n, 4] float32.
    wvp_t: the 16 floats the reference stores in cbPerObject (ObjectRenderer.cpp:186: XMStoreFloat4x4 of the TRANSPOSED world x lightViewProj)."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_set_fused_mad(1 if fused_mad else 0)
    lib.dxref_run_invocations_full.restype = C.c_int
    lib.dxref_run_invocations_full.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    cb = np.ascontiguousarray(wvp_t, np.float32).reshape(16).view(np.uint8).copy()
    _bind(CBV, 0, 0, CBUF, F_NONE, cb, len(cb))
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    n = len(pos)
    inputs = np.zeros((n, 8, 4), np.uint32)
    inputs[:, 0, :3] = pos.view(np.uint32)  # POSITION; NORMAL (element 1) is not read
    out = np.zeros((n, 8, 4), np.float32)
    assert lib.dxref_run_invocations_full(_entry("VSDepth", mode), n, inputs.ctypes.data, out.ctypes.data) == 0
    return out[:, 0, :].copy()


def vs_base_pass(positions, wvp_t, proj_bias=(0.0, 0.0), mode="fp32", fused_mad=False):
    """VSBasePass.cso (the depth / colour pass's vertex shader, VSBasePass.hlsl:44-60) for every vertex: SV_Position [n, 4] float32 --
    mul(pos, g_worldViewProj) with `Pos.xy += g_projBias * Pos.w` (the TAA jitter).  cbPerObject (VSBasePass.hlsl:27-34): g_worldViewProj
    (c0-3), g_worldViewProjPrev (c4-7), g_world float4x3 (c8-10), g_shadowWVP (c11-14), g_projBias (c15.xy); only the first and the last
    reach SV_Position."""
    lib = load()
    lib.dxref_reset()
    lib.dxref_set_fused_mad(1 if fused_mad else 0)
    lib.dxref_run_invocations_full.restype = C.c_int
    lib.dxref_run_invocations_full.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    cbf = np.zeros(64, np.float32)
    cbf[0:16] = np.asarray(wvp_t, np.float32).reshape(16)
    cbf[60:62] = proj_bias
    cb = cbf.view(np.uint8).copy()
    _bind(CBV, 0, 0, CBUF, F_NONE, cb, len(cb))
    pos = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    n = len(pos)
    inputs = np.zeros((n, 8, 4), np.uint32)
    inputs[:, 0, :3] = pos.view(np.uint32)
    out = np.zeros((n, 8, 4), np.float32)
    assert lib.dxref_run_invocations_full(_entry("VSBasePass", mode), n, inputs.ctypes.data, out.ctypes.data) == 0
    return out[:, 0, :].copy()


def rasterised_fragments(V, W, H, perObject, volumeInfo, visible, mode="fp32", subpixel_bits=8):
    """What a D3D rasteriser produces from VSCube.cso's vertices: for every visible cube its 12 triangles (index list
    MultiRayCaster.cpp:436-444), front faces culled (CULL_FRONT, clockwise = front), vertex positions snapped to 1/256 pixel (`subpixel_bits`), pixel
    centres at +0.5, top-left rule, z interpolated linearly in screen space, UVW / LPt perspective-correct -- evaluated in float64 and
    rounded to fp32 (an ideal interpolator; hardware's precision is its own).  Needs every vertex in front of the eye (w > 0: no
    clipping here).  Returns the fragment list composite(fragments=...) takes."""
    verts = vs_cube(V, perObject, volumeInfo, visible, mode).view(np.float32)
    tris = [(0, 1, 2), (3, 2, 1)]
    frags = []
    for i, v in enumerate(np.asarray(visible)):
        for face in range(6):
            q = verts[i, 4 * face:4 * face + 4].astype(np.float64)
            clip = q[:, 0, :4]
            assert (clip[:, 3] > 0).all(), "a cube vertex behind the eye: this emulation does not clip"
            ndc = clip[:, :3] / clip[:, 3:4]
            one = 1 << subpixel_bits                                               # D3D: 8 (1/256 pixel); more = a finer snap
            sx = np.rint((ndc[:, 0] * 0.5 + 0.5) * W * one).astype(np.int64)
            sy = np.rint((-ndc[:, 1] * 0.5 + 0.5) * H * one).astype(np.int64)
            for a, b, c in tris:
                area = (sx[b] - sx[a]) * (sy[c] - sy[a]) - (sy[b] - sy[a]) * (sx[c] - sx[a])
                if area >= 0:   # clockwise on a y-down screen = front face: culled (and degenerate)
                    continue
                x0, x1 = max(int(min(sx[a], sx[b], sx[c]) >> subpixel_bits), 0), min(int(max(sx[a], sx[b], sx[c]) >> subpixel_bits) + 1, W - 1)
                y0, y1 = max(int(min(sy[a], sy[b], sy[c]) >> subpixel_bits), 0), min(int(max(sy[a], sy[b], sy[c]) >> subpixel_bits) + 1, H - 1)
                if x0 > x1 or y0 > y1:
                    continue
                px = (np.arange(x0, x1 + 1, dtype=np.int64) * one + one // 2)[None, :]
                py = (np.arange(y0, y1 + 1, dtype=np.int64) * one + one // 2)[:, None]
                idx = (a, b, c)
                inside = np.ones((y1 - y0 + 1, x1 - x0 + 1), bool)
                lam = []
                for k in range(3):
                    p, r = idx[(k + 1) % 3], idx[(k + 2) % 3]      # edge opposite vertex k
                    ex, ey = sx[r] - sx[p], sy[r] - sy[p]
                    e = (px - sx[p]) * ey - (py - sy[p]) * ex      # same sign as `area` inside... with this orientation: e = -cross
                    e = -e
                    # counter-clockwise triangle (area < 0): inside where e <= 0; top-left rule decides e == 0
                    top_left = (ey == 0 and ex < 0) or (ey < 0)     # for the flipped orientation
                    inside &= (e < 0) | ((e == 0) & top_left)
                    lam.append(e.astype(np.float64) / float(area))
                ys, xs = np.nonzero(inside)
                for yy, xx in zip(ys, xs):
                    l = np.array([lam[0][yy, xx], lam[1][yy, xx], lam[2][yy, xx]])
                    w = clip[list(idx), 3]
                    z = float((l * ndc[list(idx), 2]).sum())
                    if not (0.0 <= z <= 1.0):
                        continue
                    pw = l / w
                    pw = pw / pw.sum()
                    uvw = (pw[:, None] * q[list(idx), 1, :3]).sum(axis=0)
                    lpt = (pw[:, None] * q[list(idx), 2, :3]).sum(axis=0).astype(np.float32)
                    frags.append((int(x0 + xx), int(y0 + yy), int(v), lpt, float(np.float32(z)), float(np.float32(uvw[0])), float(np.float32(uvw[1])), int(round(uvw[2]))))
    return frags


def analytic_fragments(V, W, H, px, py, perObject, cbPerFrame, visible):
    """The oracle's stand-in fragments of one pixel (mvo_pixel_fragments): [(volume, lpt[3], z)]."""
    olib = O.load()
    olib.mvo_pixel_fragments.restype = C.c_uint32
    olib.mvo_pixel_fragments.argtypes = [C.c_uint32] * 4 + [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 3
    po, cb = np.ascontiguousarray(perObject).view(np.uint8), np.ascontiguousarray(cbPerFrame).view(np.uint8)
    vis = np.ascontiguousarray(visible, np.uint32)
    vol, lpt, z = np.zeros(max(V, 1), np.uint32), np.zeros((max(V, 1), 3), np.float32), np.zeros(max(V, 1), np.float32)
    n = olib.mvo_pixel_fragments(W, H, px, py, po.ctypes.data, cb.ctypes.data, vis.ctypes.data, len(vis), vol.ctypes.data, lpt.ctypes.data, z.ctypes.data)
    return [(int(vol[k]), lpt[k].copy(), float(z[k])) for k in range(n)]
