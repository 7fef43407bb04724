"""ctypes loader for libmultivolumes_amd.so (the C ABI of include/multivolumes.h + include/multivolumes_host.h).

The library is built in-tree by multivolumes_amd/csrc/build.sh (hipcc, gfx950 only).  There is no Python or CPU
fallback for the device entry points: if the shared object is missing and cannot be built, importing fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmultivolumes_amd.so")
BUILD_SCRIPT = os.path.join(_HERE, "csrc", "build.sh")

MV_OK = 0
MV_ERR_INVALID_ARG = -1
MV_ERR_HIP = -2
MV_ERR_NO_DEVICE = -3
MV_ERR_OUT_OF_MEMORY = -4
MV_ERR_NOT_READY = -5
MV_ERR_RANGE = -6
MV_ERR_TIMEOUT = -7

MV_FLAG_FORCE_MIP0_ALL_FACES = 1
MV_FLAG_PIPELINE_FRAMES = 2
MV_FLAG_SINGLE_VIEW_STREAM = 4
MV_FLAG_LIST_ORDER = 8
MV_FLAG_RAY_QUEUE = 16
MV_FLAG_CULL_ON_VIEW_STREAM = 32
MV_FLAG_FINEGRAINED_PEER_BUFFERS = 64

MV_BUF_VOLUME, MV_BUF_LIGHT_MAP, MV_BUF_CUBE_MAP, MV_BUF_CUBE_DEPTH = 0, 1, 2, 3
MV_BUF_VOLUME_INFO, MV_BUF_COMPOSITE, MV_BUF_DEPTH_MAP, MV_BUF_SHADOW_MAP = 4, 5, 6, 7
MV_BUF_VOLUME_FLAGS, MV_BUF_EXCHANGE_OFFSETS = 10, 11

MV_DIST_ID_BYTES = 640
MV_DIST_PEER_BYTES = 8192
MV_DIST_PEER_LOOPBACK = 1
MV_DIST_PEER_FOLD = 2
MV_DIST_PEER_NO_FOLD = 4
MV_DIST_PEER_ROWS_TO_ROOT = 8
MV_DIST_PEER_DEAL_VOLUMES = 16
MV_DIST_PEER_DEAL_TILES = 32
MV_DIST_PEER_NO_ACQUIRE = 64
MV_DIST_PEER_PUSH_CELLS = 128
MV_ABI_VERSION = 6  # include/multivolumes.h
DIST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int)
DIST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int)

NUM_CUBE_MIP = 5
PER_OBJECT_BYTES = 224
CB_PER_FRAME_BYTES = 212


class mv_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("grid_size", C.c_uint32), ("light_grid_size", C.c_uint32),
        ("num_volumes", C.c_uint32), ("num_volume_srcs", C.c_uint32),
        ("viewport_w", C.c_uint32), ("viewport_h", C.c_uint32), ("shadow_map_size", C.c_uint32),
        ("device", C.c_int32), ("flags", C.c_uint32), ("stream", C.c_void_p),
        ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("frames_in_flight", C.c_uint32), ("frames_per_launch", C.c_uint32),
    ]


class mv_stats(C.Structure):
    _fields_ = [
        ("view_samples", C.c_uint64), ("lit_samples", C.c_uint64), ("rays", C.c_uint64),
        ("light_samples", C.c_uint64), ("composite_samples", C.c_uint64), ("reserved", C.c_uint64 * 3),
        ("ms_cull", C.c_double), ("ms_march_light", C.c_double), ("ms_march_view", C.c_double), ("ms_composite", C.c_double),
        ("n_cull", C.c_uint64), ("n_march_light", C.c_uint64), ("n_march_view", C.c_uint64), ("n_composite", C.c_uint64),
    ]


_P = C.c_void_p
_U32 = C.c_uint32
_SZ = C.c_size_t

# name -> (restype, argtypes).  Every symbol include/multivolumes.h and include/multivolumes_host.h declare.
ABI = {
    "mv_error_string": (C.c_char_p, [C.c_int]),
    "mv_last_error": (C.c_char_p, [_P]),
    "mv_get_warning": (C.c_char_p, [_P]),
    "mv_abi_version": (_U32, []),
    "mv_create": (C.c_int, [C.POINTER(mv_config), C.POINTER(_P)]),
    "mv_destroy": (None, [_P]),
    "mv_volume_upload_rgba16f": (C.c_int, [_P, _U32, _P]),
    "mv_volume_upload_rgba16f_device": (C.c_int, [_P, _U32, _P]),
    "mv_volume_init_procedural": (C.c_int, [_P, _U32]),
    "mv_volume_load_r32f": (C.c_int, [_P, _U32, _P, _U32, _U32, _U32]),
    "mv_set_volume_descs": (C.c_int, [_P, _P]),
    "mv_set_depth_map": (C.c_int, [_P, _P]),
    "mv_set_shadow_map": (C.c_int, [_P, _P]),
    "mv_set_depth_map_device": (C.c_int, [_P, _P]),
    "mv_set_shadow_map_device": (C.c_int, [_P, _P]),
    "mv_set_sh": (C.c_int, [_P, _P]),
    "mv_set_max_samples": (C.c_int, [_P, _U32, _U32]),
    "mv_update_frame": (C.c_int, [_P, _P, _P]),
    "mv_cull": (C.c_int, [_P]),
    "mv_march_light": (C.c_int, [_P]),
    "mv_march_view": (C.c_int, [_P]),
    "mv_composite": (C.c_int, [_P, _U32, _U32]),
    "mv_composite_over": (C.c_int, [_P, _U32, _U32, _P]),
    "mv_render": (C.c_int, [_P, C.c_int]),
    "mv_readback_volume_info": (C.c_int, [_P, _P]),
    "mv_readback_visible_lists": (C.c_int, [_P, _P, C.POINTER(_U32), _P, C.POINTER(_U32)]),
    "mv_readback_volume": (C.c_int, [_P, _U32, _P]),
    "mv_readback_light_map": (C.c_int, [_P, _U32, _P]),
    "mv_upload_light_map": (C.c_int, [_P, _U32, _P]),
    "mv_readback_cube_map": (C.c_int, [_P, _U32, _U32, _P]),
    "mv_readback_cube_depth": (C.c_int, [_P, _U32, _U32, _P]),
    "mv_readback_composite": (C.c_int, [_P, _P]),
    "mv_clear_cube_maps": (C.c_int, [_P]),
    "mv_get_stats": (C.c_int, [_P, C.POINTER(mv_stats)]),
    "mv_reset_stats": (C.c_int, [_P]),
    "mv_set_profiling": (C.c_int, [_P, C.c_int]),
    "mv_synchronize": (C.c_int, [_P]),
    "mv_get_stream": (_P, [_P]),
    "mv_get_buffer": (C.c_int, [_P, _U32, _U32, C.POINTER(_P), C.POINTER(_SZ)]),
    "mv_cube_chain_texels": (C.c_uint64, [_U32]),
    "mv_cube_mip_offset": (C.c_uint64, [_U32, _U32]),
    "mv_exchange_layout": (C.c_int, [_P, C.POINTER(_U32), C.POINTER(_SZ), C.POINTER(_SZ)]),
    "mv_shard_rows": (C.c_int, [_P, C.POINTER(_U32), C.POINTER(_U32)]),
    "mv_exchange_pack": (C.c_int, [_P, _P, _U32]),
    "mv_exchange_unpack": (C.c_int, [_P, _P, _U32]),
    "mv_readback_cull": (C.c_int, [_P, _P, _P, C.POINTER(_U32), _P, C.POINTER(_U32), C.POINTER(_U32)]),
    "mv_dist_unique_id": (C.c_int, [_P]),
    "mv_dist_init_rccl": (C.c_int, [_P, _P]),
    "mv_dist_init_callbacks": (C.c_int, [_P, _P, _P, _P]),
    "mv_dist_peer_export": (C.c_int, [_P, _P]),
    "mv_dist_init_peer": (C.c_int, [_P, _P, _U32]),
    "mv_dist_set_peer_timeout_ms": (C.c_int, [_P, _U32]),
    "mv_dist_peer_coherence": (C.c_int, [_P, _U32, C.POINTER(_U32), C.POINTER(_U32)]),
    "mv_dealt_tiles": (_U32, [_U32, _U32, _U32, _U32, C.POINTER(_U32), _U32]),
    "mv_light_rebuild_range": (C.c_int, [_U32, _U32, _U32, C.POINTER(_U32)]),
    "mv_dist_selftest": (C.c_int, [_P]),
    "mv_dist_shutdown": (C.c_int, [_P]),
    "mv_dist_set_inplace_gather": (C.c_int, [_P, C.c_int]),
    "mv_dist_set_cull_crosscheck": (C.c_int, [_P, C.c_int]),
    "mv_dist_frame": (C.c_int, [_P, _P, _P, C.c_int]),
    "mv_dist_flush": (C.c_int, [_P]),
    "mv_dist_exchanged_bytes": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "mv_dist_rccl_available": (C.c_int, []),
    "mv_dist_info": (C.c_int, [_P, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    "mv_host_wait_ms": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int]),
    "mv_set_latency_sampling": (C.c_int, [_P, _U32]),
    "mv_get_latency": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "mv_dist_peer_info": (C.c_int, [_P, C.POINTER(_U32)]),
    "mv_set_viewport": (C.c_int, [_P, _U32, _U32]),
    "mv_mesh_upload": (C.c_int, [_P, _P, _U32, _P, _U32]),
    "mv_mesh_render_depth": (C.c_int, [_P, _P, _P]),
    "mv_mesh_render_shadow": (C.c_int, [_P, _P]),
    "mv_readback_depth_map": (C.c_int, [_P, _P]),
    "mv_readback_shadow_map": (C.c_int, [_P, _P]),
    # host helpers
    "mvh_load_obj": (C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(_U32), C.POINTER(_P), C.POINTER(_U32), _P, _P]),
    "mvh_free": (None, [_P]),
    "mvh_load_dds_volume": (C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    "mvh_mesh_world": (None, [C.c_float, _P, _P]),
    "mvh_transpose": (None, [_P, _P]),
    "mvh_look_at_lh": (None, [_P, _P, _P, _P]),
    "mvh_perspective_fov_lh": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "mvh_orthographic_lh": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "mvh_multiply": (None, [_P, _P, _P]),
    "mvh_inverse": (C.c_int, [_P, _P]),
    "mvh_set_volume_world": (None, [C.c_float, _P, _P]),
    "mvh_set_volumes_world": (None, [_U32, C.c_float, _P, _P]),
    "mvh_shadow_view_proj": (None, [_P, C.c_float, _P]),
    "mvh_update_frame": (None, [_U32, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _U32, _P, _P]),
    "mvh_synth_volume": (None, [_U32, _U32, _P]),
    "mvh_synth_sphere_depth": (None, [_U32, _U32, _P, _P, C.c_float, _P]),
    "mvh_synth_sphere_shadow": (None, [_U32, _P, _P, C.c_float, _P]),
}

_lib = None


def _source_hash():
    """Content hash of everything the library is built from (mtimes do not survive the copy to the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    for d in (src, inc):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".cpp", ".hpp", ".h", ".sh")):
                h.update(f.encode())
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


HASH_PATH = LIB_PATH + ".srchash"


def _stale():
    """True when the shared object was not built from the sources that are in the tree now."""
    try:
        with open(HASH_PATH) as fh:
            return fh.read().strip() != _source_hash()
    except OSError:
        return True


def build(force=False):
    """Compile the HIP extension in-tree (hipcc --offload-arch=gfx950).  One builder at a time (ranks / xdist workers share the tree)."""
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not os.path.exists(LIB_PATH) or _stale():
            subprocess.run(["bash", BUILD_SCRIPT], check=True)  # writes HASH_PATH itself
    return LIB_PATH


def load():
    """Load the shared library and set ctypes signatures; rebuilds it first if it is missing or older than its sources,
    raises if that fails (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    alt = os.environ.get("MV_LIB_PATH")  # A/B measurements: a library built with other switches (scripts/exp_build.sh)
    if alt:
        lib = C.CDLL(alt)
    else:
        if not os.path.exists(LIB_PATH) or _stale():
            build()
        lib = C.CDLL(LIB_PATH)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class MVError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = load().mv_error_string(code).decode()
        super().__init__(f"{where}: {msg} ({code})" + (f": {detail}" if detail else ""))
