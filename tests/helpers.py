"""Shared scene plumbing for the tests: builds the CBPerFrame / PerObject blobs with the product's host helpers
and runs the ORACLE pipeline on them (the oracle is the checker, never the thing under test in -m gpu tests)."""
import numpy as np

from multivolumes_amd import _lib, scene
from multivolumes_amd import raycaster as rc


def default_descs(V, S, N):
    return np.array([(i % S) | (5 << 14) | (N << 18) for i in range(V)], np.uint32)


def host_frame(cfg, frame_idx=0, eye=None, worlds=None, light=None):
    """(eye, viewProj, shadowVP^T, cbPerFrame bytes, perObject bytes) for a config, reference defaults."""
    V = cfg["V"]
    lib = _lib.load()
    if worlds is None:
        worlds = np.zeros((V, 12), np.float32)
        lib.mvh_set_volumes_world(V, scene.VOLUME_SIZE, rc._ptr(rc._f32((0, 0, 0), 3)), rc._ptr(worlds))
    eye, vp = scene.camera(cfg, eye)
    svp = scene.shadow_vp(cfg)
    cb = np.zeros(212, np.uint8)
    po = np.zeros(V * 224, np.uint8)
    lc = np.array([*scene.LIGHT_COLOR, scene.LIGHT_INTENSITY], np.float32)
    am = np.array([*scene.AMBIENT_COLOR, scene.AMBIENT_INTENSITY], np.float32)
    lib.mvh_update_frame(V, rc._ptr(worlds), rc._ptr(rc._f32(vp, 16)), rc._ptr(rc._f32(svp, 16)), rc._ptr(eye),
                         float(cfg["W"]), float(cfg["H"]), rc._ptr(rc._f32(scene.LIGHT_PT if light is None else light, 3)), rc._ptr(lc), rc._ptr(am),
                         frame_idx, rc._ptr(cb), rc._ptr(po))
    return eye, vp, svp, cb, po


def oracle_pipeline(o, cfg, volumes, M, frames=None, sh=None, eye=None, worlds=None, light=None):
    """cull -> marchL for `frames` consecutive frame indices (default: one per volume) -> marchV, all on the oracle.
    worlds: V x 12 floats (mvh_set_volume_world per volume) instead of the reference's grid; light: another light position."""
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    S = len(volumes)
    descs = default_descs(V, S, N)
    eye_, vp, svp, cb, po = host_frame(cfg, 0, eye, worlds, light)
    depth, shadow = scene.depth_and_shadow(cfg, vp, svp, M)
    info, vis, cube = o.cull(V, po, descs, cb, cfg["maxRay"])
    lms = None
    nls = 0
    frames = V if frames is None else frames
    for f in range(frames):
        _, _, _, cb, po = host_frame(cfg, f, eye, worlds, light)
        _, lms, n = o.march_light(V, N, L, M, volumes, descs, po, cb, shadow, sh, vis, cfg["maxLight"], lms)
        nls += n
    cm, cd, st = o.march_view(V, N, L, cfg["W"], cfg["H"], volumes, lms, depth, po, cb, info, cube)
    st["light_samples"] = nls
    return dict(info=info, vis=vis, cube=cube, lightMaps=lms, cubeMaps=cm, cubeDepths=cd, stats=st,
                depth=depth, shadow=shadow, descs=descs, cb=cb, po=po, viewProj=vp, shadowVP=svp, eye=eye_)


def pack_r11g11b10f_const(o, r, g, b, L):
    return np.full((L, L, L), o.load().mvo_pack_r11g11b10f(r, g, b), np.uint32)


def replicating_allgather(world):
    """An all-gather callback for ONE rank alone on the GPU (no peers to exchange with): every slot of the receive buffer gets this rank's
    own share, through hipMemcpyAsync on the collective's stream.  The loop's cross-rank check of the culls (the trailers of the light
    exchange) then sees `world` equal trailers -- with a callback that copies nothing it would read whatever the buffer held."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.restype = C.c_int
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]

    def allgather(send, recv, nbytes, stream, which):
        for r in range(world):
            dst = recv + r * nbytes
            if dst != send and hip.hipMemcpyAsync(dst, send, nbytes, 3, stream) != 0:  # 3 = hipMemcpyDeviceToDevice
                return 1
        return 0
    return allgather
