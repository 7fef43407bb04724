"""ctypes wrapper of the CPU oracle (oracle/libmv_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the product package.  Parity: a1 / a2 / a3 / a6 and the composite's shaders pinned to the reference's shipped compiled shaders
(oracle/dxref.py); rasteriser and sampler are a stated contract: see oracle/mv_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmv_oracle.so")
NUM_CUBE_MIP = 5


class mvo_stats(C.Structure):
    _fields_ = [("view_samples", C.c_uint64), ("lit_samples", C.c_uint64), ("rays", C.c_uint64), ("light_samples", C.c_uint64)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("mv_oracle.cpp", "mv_oracle_mesh.cpp", "mv_oracle.h", "Makefile")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", _HERE, "-B", "libmv_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_P, _U32, _F = C.c_void_p, C.c_uint32, C.c_float


_budget = None


def host_cpu_budget():
    """(computed once, when this module is imported: an OpenMP runtime loaded later with OMP_PROC_BIND set pins the main thread to one
    CPU, and the affinity mask would then say 1)
    The CPUs this process can actually keep busy: the smaller of the affinity mask and the cgroup's CFS quota (a GPU box of this pool
    shows 256 logical CPUs and grants 16: `cpu.max 1600000 100000`).  More threads than that are throttled, and the throughput of a
    sustained run collapses (profiles/r03_cpu_scaling.txt) -- so this is the thread count of both CPU legs of bench.py's cpu_baseline
    and the oracle's default."""
    global _budget
    if _budget is not None:
        return _budget
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.999)))
    _budget = max(1, n)
    return _budget


host_cpu_budget()


def load():
    global _lib
    if _lib is not None:
        return _lib
    alt = os.environ.get("MV_ORACLE_LIB")  # scripts/sanitize_cpu.sh: the same sources built with ASan / UBSan
    if alt:
        lib = C.CDLL(alt)
    else:
        build()
        lib = C.CDLL(LIB_PATH)
    sig = {
        "mvo_f32_to_f16": (C.c_uint16, [_F]), "mvo_f16_to_f32": (_F, [C.c_uint16]),
        "mvo_pack_r11g11b10f": (_U32, [_F, _F, _F]), "mvo_unpack_r11g11b10f": (None, [_U32, _P]),
        "mvo_cube_chain_texels": (C.c_uint64, [_U32]), "mvo_cube_mip_offset": (C.c_uint64, [_U32, _U32]),
        "mvo_sample_rgba16f": (None, [_P, _U32, _P, _P, _P]), "mvo_sample_r11g11b10f": (None, [_P, _U32, _P, _P]),
        "mvo_sample_point_r32f": (_F, [_P, _U32, _U32, _F, _F]), "mvo_sample_cmp_d16": (_F, [_P, _U32, _F, _F, _F]),
        "mvo_compute_ray_origin": (C.c_int, [_P, _P]), "mvo_compute_target_hit": (_F, [_P, _P, _P]),
        "mvo_get_step": (_F, [_F, _F, _F, _F]), "mvo_get_local_pos": (None, [_F, _F, _U32, _U32, _P]),
        "mvo_cast_light_ray": (None, [_P, _P, _U32, _P, _P, _F, _U32, _P]),
        "mvo_eval_sh_irradiance": (None, [_P, _P, _P]),
        "mvo_init_grid_data": (None, [_U32, _P]), "mvo_r32f_to_rgba16f": (None, [_P, _U32, _U32, _U32, _U32, _P]),
        "mvo_cull": (None, [_U32, _P, _P, _P, _U32, _P, _P, _P, _P, _P]),
        "mvo_march_light": (_U32, [_U32, _U32, _U32, _U32, _P, _P, _P, _P, _P, _P, _P, _U32, _U32, _P, _P]),
        "mvo_march_view": (None, [_U32, _U32, _U32, _U32, _U32, _P, _P, _P, _P, _P, _P, _P, _U32, _P, _P, _P]),
        "mvo_num_threads": (C.c_int, []), "mvo_set_num_threads": (None, [C.c_int]),
        "mvo_parallel_copy": (None, [C.c_void_p, C.c_void_p, C.c_size_t]),
        "mvo_set_light_block_range": (None, [_U32, _U32]),
        "mvo_unproject_z": (_F, [_F]), "mvo_cube_cast": (None, [_P, _P, _U32, _F, _P, _P, _P]),
        "mvo_cube_face_coords": (None, [_P, C.POINTER(_U32), _P]), "mvo_cube_texel_index": (C.c_uint64, [_U32, C.c_int32, C.c_int32, _U32]),
        "mvo_set_options": (None, [C.c_int, C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if hasattr(lib, "mvo_composite"):
        lib.mvo_composite.restype = None
        lib.mvo_composite.argtypes = [_U32, _U32, _U32, _U32, _U32, _U32, _U32, _P, _P, _P, _P, _P, _P, _P, _U32, _P, _P, _P, _P]
        lib.mvo_composite_over.restype = None
        lib.mvo_composite_over.argtypes = [_U32, _U32, _U32, _U32, _U32, _U32, _U32, _P, _P, _P, _P, _P, _P, _P, _U32, _P, _P, _P, _P, _P]
    if "OMP_NUM_THREADS" not in os.environ:
        lib.mvo_set_num_threads(host_cpu_budget())
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _vol_ptrs(volumes):
    vols = [np.ascontiguousarray(v).view(np.uint16) for v in volumes]
    arr = (C.c_void_p * len(vols))(*[v.ctypes.data for v in vols])
    return vols, arr


def cube_chain_texels(N):
    return int(load().mvo_cube_chain_texels(N))


def cube_mip_offset(N, mip):
    return int(load().mvo_cube_mip_offset(N, mip))


def f16_to_f32(a):
    """uint16 bit patterns -> float32 through the oracle's own decoder."""
    lib = load()
    a = np.asarray(a, np.uint16)
    return np.array([lib.mvo_f16_to_f32(int(x)) for x in a.ravel()], np.float32).reshape(a.shape)


def init_grid_data(N):
    out = np.empty((N, N, N, 4), np.uint16)
    load().mvo_init_grid_data(N, _ptr(out))
    return out


def r32f_to_rgba16f(src, N):
    src = np.ascontiguousarray(src, np.float32)
    d, h, w = src.shape
    out = np.empty((N, N, N, 4), np.uint16)
    load().mvo_r32f_to_rgba16f(_ptr(src), w, h, d, N, _ptr(out))
    return out


def cull(V, perObject, descs, cbPerFrame, maxRaySamples, volumeInfo=None):
    """Returns (volumeInfo[V,4] u16, visible list, cube-map list). volumeInfo rows of culled volumes keep their input."""
    info = np.zeros((V, 4), np.uint16) if volumeInfo is None else np.ascontiguousarray(volumeInfo, np.uint16).copy()
    vis = np.zeros(V, np.uint32)
    cube = np.zeros(V, np.uint32)
    nv, nc = C.c_uint32(), C.c_uint32()
    po = np.ascontiguousarray(perObject).view(np.uint8)
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8)
    d = np.ascontiguousarray(descs, np.uint32)
    load().mvo_cull(V, _ptr(po), _ptr(d), _ptr(cb), maxRaySamples, _ptr(info), _ptr(vis), C.byref(nv), _ptr(cube), C.byref(nc))
    return info, vis[:nv.value].copy(), cube[:nc.value].copy()


def march_light(V, N, L, M, volumes, descs, perObject, cbPerFrame, shadowMap, sh27, visible, maxLightSamples, lightMaps=None):
    """Returns (volumeId, lightMaps[V, L,L,L] u32, light sample count)."""
    vols, arr = _vol_ptrs(volumes)
    lm = np.zeros((V, L, L, L), np.uint32) if lightMaps is None else np.ascontiguousarray(lightMaps, np.uint32).copy()
    po = np.ascontiguousarray(perObject).view(np.uint8)
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8)
    d = np.ascontiguousarray(descs, np.uint32)
    sm = np.ascontiguousarray(shadowMap, np.uint16)
    sh = None if sh27 is None else np.ascontiguousarray(sh27, np.float32)
    vis = np.ascontiguousarray(visible, np.uint32)
    st = mvo_stats()
    vid = load().mvo_march_light(V, N, L, M, arr, _ptr(d), _ptr(po), _ptr(cb), _ptr(sm), _ptr(sh), _ptr(vis), len(vis),
                                 maxLightSamples, _ptr(lm), C.byref(st))
    return int(vid), lm, int(st.light_samples)


def march_view(V, N, L, W, H, volumes, lightMaps, depthMap, perObject, cbPerFrame, volumeInfo, cubeMapVolumes,
               cubeMaps=None, cubeDepths=None, inplace=False):
    """Returns (cubeMaps[V, chain, 4] u16, cubeDepths[V, chain] f32, stats dict). inplace=True writes into the given arrays."""
    vols, arr = _vol_ptrs(volumes)
    chain = cube_chain_texels(N)
    if inplace:
        cm, cd = cubeMaps, cubeDepths
        assert cm.dtype == np.uint16 and cd.dtype == np.float32 and cm.flags.c_contiguous and cd.flags.c_contiguous
    else:
        cm = np.zeros((V, chain, 4), np.uint16) if cubeMaps is None else np.ascontiguousarray(cubeMaps, np.uint16).copy()
        cd = np.zeros((V, chain), np.float32) if cubeDepths is None else np.ascontiguousarray(cubeDepths, np.float32).copy()
    lm = np.ascontiguousarray(lightMaps, np.uint32)
    dm = np.ascontiguousarray(depthMap, np.float32)
    po = np.ascontiguousarray(perObject).view(np.uint8)
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8)
    vi = np.ascontiguousarray(volumeInfo, np.uint16)
    cl = np.ascontiguousarray(cubeMapVolumes, np.uint32)
    st = mvo_stats()
    load().mvo_march_view(V, N, L, W, H, arr, _ptr(lm), _ptr(dm), _ptr(po), _ptr(cb), _ptr(vi), _ptr(cl), len(cl),
                          _ptr(cm), _ptr(cd), C.byref(st))
    return cm, cd, dict(view_samples=int(st.view_samples), lit_samples=int(st.lit_samples), rays=int(st.rays))


def spread(a):
    """A copy of `a` whose pages were first touched by all OpenMP threads (spread over the sockets' memory): for timing runs."""
    a = np.ascontiguousarray(a)
    out = np.empty(a.shape, a.dtype)
    load().mvo_parallel_copy(_ptr(out), _ptr(a), a.nbytes)
    return out


def cube_face_view(chain_array, N, mip):
    """Slice mip `mip` out of a per-volume chain array: [chain, ...] -> [6, d, d, ...]."""
    d = max(1, N >> mip)   # (a mip's edge as D3D defines it)
    off = cube_mip_offset(N, mip)
    a = chain_array[off:off + 6 * d * d]
    return a.reshape((6, d, d) + a.shape[1:])


def composite(V, N, L, W, H, volumes, lightMaps, depthMap, perObject, cbPerFrame, volumeInfo, visible, cubeMaps, cubeDepths,
              y0=0, y1=None, out=None, background=None):
    """K-buffer OIT composite over rows [y0, y1). Returns (out[H, W, 4] u16, RayCast sample count)."""
    vols, arr = _vol_ptrs(volumes)
    y1 = H if y1 is None else y1
    o = np.zeros((H, W, 4), np.uint16) if out is None else out
    lm = np.ascontiguousarray(lightMaps, np.uint32)
    dm = np.ascontiguousarray(depthMap, np.float32)
    po = np.ascontiguousarray(perObject).view(np.uint8)
    cb = np.ascontiguousarray(cbPerFrame).view(np.uint8)
    vi = np.ascontiguousarray(volumeInfo, np.uint16)
    vl = np.ascontiguousarray(visible, np.uint32)
    cm = np.ascontiguousarray(cubeMaps, np.uint16)
    cd = np.ascontiguousarray(cubeDepths, np.float32)
    st = mvo_stats()
    bg = None if background is None else np.ascontiguousarray(background, np.uint16)
    load().mvo_composite_over(V, N, L, W, H, y0, y1, arr, _ptr(lm), _ptr(dm), _ptr(po), _ptr(cb), _ptr(vi), _ptr(vl), len(vl),
                              _ptr(cm), _ptr(cd), _ptr(bg), _ptr(o), C.byref(st))
    return o, int(st.view_samples)


def raster_depth(positions, indices, wvp_t, W, H, bias=(0.0, 0.0)):
    """CPU oracle of the mesh depth rasteriser: returns float32[H, W] (1.0 where nothing was drawn)."""
    lib = load()
    lib.mvo_raster_depth.restype = None
    lib.mvo_raster_depth.argtypes = [_P, _P, _U32, _P, _F, _F, _U32, _U32, _P]
    p = np.ascontiguousarray(positions, np.float32)
    i = np.ascontiguousarray(indices, np.uint32)
    m = np.ascontiguousarray(wvp_t, np.float32)
    out = np.ones((H, W), np.float32)
    lib.mvo_raster_depth(_ptr(p), _ptr(i), i.size // 3, _ptr(m), bias[0], bias[1], W, H, _ptr(out))
    return out


def depth_to_d16(depth):
    lib = load()
    lib.mvo_depth_to_d16.restype = None
    lib.mvo_depth_to_d16.argtypes = [_P, C.c_size_t, _P]
    d = np.ascontiguousarray(depth, np.float32)
    out = np.empty(d.shape, np.uint16)
    lib.mvo_depth_to_d16(_ptr(d), d.size, _ptr(out))
    return out
