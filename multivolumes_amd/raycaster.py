"""Python mirror of the reference's hot-path owner, MultiRayCaster (MultiVolumes/Content/MultiRayCaster.h:28-50),
driving the gfx950 kernels through the C ABI of include/multivolumes.h.

Method names, argument meaning and defaults follow the reference so that tests read like calls into it:
Init -> __init__, LoadVolumeData / InitVolumeData, SetRenderTargets, SetSH, SetMaxSamples, SetVolumesWorld /
SetVolumeWorld, SetLight, SetAmbient, UpdateFrame, Render.  Creation failures raise (the reference returns false,
XUSG.h:12-15); stage launches are stream-ordered and asynchronous like the reference's command recording.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MVError, mv_config, mv_stats


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


# ---- DirectXMath-convention helpers (row vectors, row-major 4x4), computed by libmultivolumes_amd's host code ----
def look_at_lh(eye, focus, up=(0.0, 1.0, 0.0)):
    out = np.empty(16, np.float32)
    _lib.load().mvh_look_at_lh(_ptr(_f32(eye, 3)), _ptr(_f32(focus, 3)), _ptr(_f32(up, 3)), _ptr(out))
    return out.reshape(4, 4)


def perspective_fov_lh(fov_y, aspect, zn, zf):
    out = np.empty(16, np.float32)
    _lib.load().mvh_perspective_fov_lh(fov_y, aspect, zn, zf, _ptr(out))
    return out.reshape(4, 4)


def orthographic_lh(w, h, zn, zf):
    out = np.empty(16, np.float32)
    _lib.load().mvh_orthographic_lh(w, h, zn, zf, _ptr(out))
    return out.reshape(4, 4)


def multiply(a, b):
    out = np.empty(16, np.float32)
    _lib.load().mvh_multiply(_ptr(_f32(a, 16)), _ptr(_f32(b, 16)), _ptr(out))
    return out.reshape(4, 4)


def shadow_view_proj(light_pos, scene_size):
    """ObjectRenderer.cpp:176-185; returned TRANSPOSED, as the reference stores m_shadowVP."""
    out = np.empty(16, np.float32)
    _lib.load().mvh_shadow_view_proj(_ptr(_f32(light_pos, 3)), scene_size, _ptr(out))
    return out.reshape(4, 4)


def synth_volume(N, k):
    out = np.empty((N, N, N, 4), np.uint16)
    _lib.load().mvh_synth_volume(N, k, _ptr(out))
    return out


def synth_sphere_depth(W, H, view_proj, center, radius):
    out = np.empty((H, W), np.float32)
    _lib.load().mvh_synth_sphere_depth(W, H, _ptr(_f32(view_proj, 16)), _ptr(_f32(center, 3)), radius, _ptr(out))
    return out


def synth_sphere_shadow(M, light_view_proj_t, center, radius):
    out = np.empty((M, M), np.uint16)
    _lib.load().mvh_synth_sphere_shadow(M, _ptr(_f32(light_view_proj_t, 16)), _ptr(_f32(center, 3)), radius, _ptr(out))
    return out


def load_obj(path):
    """(positions[n,3] float32, indices[m,3] uint32, aabb_min[3], aabb_max[3]) of a Wavefront OBJ (mvh_load_obj)."""
    lib = _lib.load()
    pos, idx = C.c_void_p(), C.c_void_p()
    nv, nt = C.c_uint32(), C.c_uint32()
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    if lib.mvh_load_obj(str(path).encode(), C.byref(pos), C.byref(nv), C.byref(idx), C.byref(nt), _ptr(lo), _ptr(hi)) != 0:
        raise IOError(f"cannot load OBJ {path}")
    try:
        p = np.ctypeslib.as_array(C.cast(pos, C.POINTER(C.c_float)), (nv.value, 3)).copy()
        i = np.ctypeslib.as_array(C.cast(idx, C.POINTER(C.c_uint32)), (nt.value, 3)).copy()
    finally:
        lib.mvh_free(pos)
        lib.mvh_free(idx)
    return p, i, lo, hi


def load_dds_volume(path):
    """float32 [depth, height, width] of the volume texture in a .dds file (mvh_load_dds_volume)."""
    lib = _lib.load()
    tex = C.c_void_p()
    w, h, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc_ = lib.mvh_load_dds_volume(str(path).encode(), C.byref(tex), C.byref(w), C.byref(h), C.byref(d))
    if rc_ != 0:
        raise IOError(f"cannot load DDS volume {path}: " + {-1: "unreadable or not a DDS file", -2: "not a volume texture", -3: "unsupported pixel format", -4: "truncated"}.get(rc_, str(rc_)))
    try:
        return np.ctypeslib.as_array(C.cast(tex, C.POINTER(C.c_float)), (d.value, h.value, w.value)).copy()
    finally:
        lib.mvh_free(tex)


def mesh_world(scale, pos):
    """ObjectRenderer::SetWorld (ObjectRenderer.cpp:147-154): Scale * Translate, row-major 4x4."""
    out = np.empty(16, np.float32)
    _lib.load().mvh_mesh_world(float(scale), _ptr(_f32(pos, 3)), _ptr(out))
    return out.reshape(4, 4)


def transpose_store(m):
    out = np.empty(16, np.float32)
    _lib.load().mvh_transpose(_ptr(_f32(m, 16)), _ptr(out))
    return out.reshape(4, 4)


def cube_chain_texels(N):
    return int(_lib.load().mv_cube_chain_texels(N))


def cube_mip_offset(N, mip):
    return int(_lib.load().mv_cube_mip_offset(N, mip))


class MultiRayCaster:
    FrameCount = 3  # MultiRayCaster.h:52

    def __init__(self, gridSize, lightGridSize, numVolumes, numVolumeSrcs, viewport, shadowMapSize=1024,
                 device=0, stream=None, flags=0, shard=(0, 1), framesInFlight=0, framesPerLaunch=0):
        """MultiRayCaster::Init (MultiRayCaster.cpp:79-166) + SetViewport (:226-241)."""
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self.gridSize, self.lightGridSize = int(gridSize), int(lightGridSize)
        self.numVolumes, self.numVolumeSrcs = int(numVolumes), int(numVolumeSrcs)
        self.viewport = (int(viewport[0]), int(viewport[1]))
        self.shadowMapSize = int(shadowMapSize)
        self.framesInFlight = int(framesInFlight) or 3
        self.framesPerLaunch = int(framesPerLaunch) or 1  # mv_config.frames_per_launch: B frames of the multi-GPU loop (peer stores) per kernel launch
        cfg = mv_config(C.sizeof(mv_config), self.gridSize, self.lightGridSize, self.numVolumes, self.numVolumeSrcs,
                        self.viewport[0], self.viewport[1], self.shadowMapSize, int(device), int(flags),
                        C.c_void_p(stream) if stream else None, int(shard[0]), int(shard[1]), int(framesInFlight), int(framesPerLaunch))
        rc = self._lib.mv_create(C.byref(cfg), C.byref(self._ctx))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise MVError(rc, "mv_create")
        # what mv_create found worth saying although it succeeded (include/multivolumes.h mv_get_warning): a context with more chains than the
        # process has hardware queues needs GPU_MAX_HW_QUEUES >= framesInFlight + 1 in the environment BEFORE torch / HIP start (bench.py sets 8)
        self.warning = self._lib.mv_get_warning(self._ctx).decode()
        if self.warning:
            import warnings
            warnings.warn(self.warning, RuntimeWarning, stacklevel=2)
        # host-side state of the reference class (MultiRayCaster.cpp:58-72)
        self.m_maxRaySamples, self.m_maxLightSamples = 256, 96
        self.m_lightPt = np.array([75.0, 75.0, -75.0], np.float32)
        self.m_lightColor = np.array([1.0, 0.7, 0.3, 1.0], np.float32)
        self.m_ambient = np.array([0.0, 0.3, 1.0, 0.4], np.float32)
        self.m_frameIdx = 0
        self.m_volumeWorlds = np.zeros((self.numVolumes, 12), np.float32)
        self.SetVolumesWorld(20.0, (0.0, 0.0, 0.0))  # MultiRayCaster.cpp:141
        self.cbPerFrame = np.zeros(_lib.CB_PER_FRAME_BYTES, np.uint8)
        self._vp, self._svp, self._eye = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(3, np.float32)
        self._pcache = {}
        self.perObject = np.zeros(self.numVolumes * _lib.PER_OBJECT_BYTES, np.uint8)

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.mv_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, where):
        if rc != 0:
            raise MVError(rc, where, self._lib.mv_last_error(self._ctx).decode())

    # ---- volume data ----
    def LoadVolumeData(self, i, rgba16f):
        """Result of LoadVolumeData (MultiRayCaster.cpp:168-209): N^3 x 4 halfs (uint16 bit patterns or float16)."""
        a = np.ascontiguousarray(rgba16f)
        if a.dtype == np.float16:
            a = a.view(np.uint16)
        N = self.gridSize
        if a.dtype != np.uint16 or a.size != N * N * N * 4:
            raise ValueError("volume must be N^3 x 4 uint16/float16")
        self._check(self._lib.mv_volume_upload_rgba16f(self._ctx, i, _ptr(a)), "mv_volume_upload_rgba16f")

    def LoadVolumeFile(self, i, fileName):
        """LoadVolumeData(pCommandList, i, fileName, ...) (MultiRayCaster.cpp:168-209): DDS volume -> CSR32FToRGBA16F."""
        self.LoadVolumeDataR32F(i, load_dds_volume(fileName))

    def LoadVolumeDataR32F(self, i, a):
        """LoadVolumeData's CSR32FToRGBA16F pass on a (d, h, w) float32 array."""
        a = np.ascontiguousarray(a, np.float32)
        d, h, w = a.shape
        self._check(self._lib.mv_volume_load_r32f(self._ctx, i, _ptr(a), w, h, d), "mv_volume_load_r32f")

    def LoadVolumeDataDevice(self, i, device_ptr):
        self._check(self._lib.mv_volume_upload_rgba16f_device(self._ctx, i, C.c_void_p(device_ptr)), "mv_volume_upload_rgba16f_device")

    def InitVolumeData(self, i):
        """InitVolumeData (MultiRayCaster.cpp:243-264)."""
        self._check(self._lib.mv_volume_init_procedural(self._ctx, i), "mv_volume_init_procedural")

    def SetVolumeDescs(self, packed):
        a = np.ascontiguousarray(packed, np.uint32)
        if a.size != self.numVolumes:
            raise ValueError("need one VolumeDesc per volume")
        self._check(self._lib.mv_set_volume_descs(self._ctx, _ptr(a)), "mv_set_volume_descs")

    # ---- inputs ----
    def SetViewport(self, width, height):
        """SetViewport (MultiRayCaster.cpp:226-241): the W x H buffers (composite image, RayCast layers, own depth map) are
        re-created; volumes, light maps and cube maps stay.  The depth map has to be supplied again."""
        self._check(self._lib.mv_set_viewport(self._ctx, int(width), int(height)), "mv_set_viewport")
        self.viewport = (int(width), int(height))

    def SetRenderTargets(self, depthMap=None, shadowMap=None, depthMapDevice=None, shadowMapDevice=None):
        """SetRenderTargets (MultiRayCaster.cpp:211-224): the DEPTH_MAP / SHADOW_MAP the path reads."""
        W, H = self.viewport
        if depthMap is not None:
            a = np.ascontiguousarray(depthMap, np.float32)
            if a.size != W * H:
                raise ValueError("depth map must be H x W float32")
            self._check(self._lib.mv_set_depth_map(self._ctx, _ptr(a)), "mv_set_depth_map")
        if shadowMap is not None:
            a = np.ascontiguousarray(shadowMap, np.uint16)
            if a.size != self.shadowMapSize ** 2:
                raise ValueError("shadow map must be M x M uint16")
            self._check(self._lib.mv_set_shadow_map(self._ctx, _ptr(a)), "mv_set_shadow_map")
        if depthMapDevice is not None:
            self._check(self._lib.mv_set_depth_map_device(self._ctx, C.c_void_p(depthMapDevice)), "mv_set_depth_map_device")
        if shadowMapDevice is not None:
            self._check(self._lib.mv_set_shadow_map_device(self._ctx, C.c_void_p(shadowMapDevice)), "mv_set_shadow_map_device")

    # ---- mesh depth / shadow producer (what ObjectRenderer hands to SetRenderTargets) ----
    def SetMesh(self, positions, indices):
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        self._check(self._lib.mv_mesh_upload(self._ctx, _ptr(p), len(p), _ptr(i), len(i)), "mv_mesh_upload")

    def RenderMeshDepth(self, worldViewProj, projBias=None):
        """ObjectRenderer::Render's depth output (VSBasePass.hlsl:44-60). worldViewProj: row-major world * view * proj."""
        t = transpose_store(worldViewProj)
        b = None if projBias is None else _f32(projBias, 2)
        self._check(self._lib.mv_mesh_render_depth(self._ctx, _ptr(t), _ptr(b)), "mv_mesh_render_depth")

    def RenderMeshShadow(self, shadowWVP):
        """ObjectRenderer::RenderShadow (VSDepth.hlsl:24-27). shadowWVP: row-major world * lightView * lightProj."""
        t = transpose_store(shadowWVP)
        self._check(self._lib.mv_mesh_render_shadow(self._ctx, _ptr(t)), "mv_mesh_render_shadow")

    def ReadBackDepthMap(self):
        W, H = self.viewport
        out = np.empty((H, W), np.float32)
        self._check(self._lib.mv_readback_depth_map(self._ctx, _ptr(out)), "mv_readback_depth_map")
        return out

    def ReadBackShadowMap(self):
        M = self.shadowMapSize
        out = np.empty((M, M), np.uint16)
        self._check(self._lib.mv_readback_shadow_map(self._ctx, _ptr(out)), "mv_readback_shadow_map")
        return out

    def SetSH(self, coeffSH):
        a = None if coeffSH is None else _f32(coeffSH, 27)
        self._check(self._lib.mv_set_sh(self._ctx, _ptr(a)), "mv_set_sh")

    def SetMaxSamples(self, maxRaySamples, maxLightSamples):
        self.m_maxRaySamples, self.m_maxLightSamples = int(maxRaySamples), int(maxLightSamples)
        self._check(self._lib.mv_set_max_samples(self._ctx, self.m_maxRaySamples, self.m_maxLightSamples), "mv_set_max_samples")

    def SetVolumesWorld(self, size, center):
        self._lib.mvh_set_volumes_world(self.numVolumes, float(size), _ptr(_f32(center, 3)), _ptr(self.m_volumeWorlds))

    def SetVolumeWorld(self, i, size, pos):
        self._lib.mvh_set_volume_world(float(size), _ptr(_f32(pos, 3)), _ptr(self.m_volumeWorlds[i]))

    def SetLight(self, pos, color, intensity):
        self.m_lightPt = _f32(pos, 3)
        self.m_lightColor = np.array([color[0], color[1], color[2], intensity], np.float32)

    def SetAmbient(self, color, intensity):
        self.m_ambient = np.array([color[0], color[1], color[2], intensity], np.float32)

    def _pp(self, name):
        """Cached pointer of a persistent member array (rebuilt when the member is replaced): a.ctypes.data_as costs microseconds,
        and a frame of the multi-GPU loop is enqueued in a hundred."""
        arr = getattr(self, name)
        c = self._pcache.get(name)
        if c is None or c[0] is not arr:
            c = (arr, _ptr(arr))
            self._pcache[name] = c
        return c[1]

    def _frame_constants(self, viewProj, shadowVP, eyePt):
        """mvh_update_frame into self.cbPerFrame / self.perObject through staging arrays with cached pointers."""
        np.copyto(self._vp, np.asarray(viewProj, np.float32).reshape(16))
        np.copyto(self._svp, np.asarray(shadowVP, np.float32).reshape(16))
        np.copyto(self._eye, np.asarray(eyePt, np.float32).reshape(3))
        W, H = self.viewport
        pp = self._pp
        self._lib.mvh_update_frame(self.numVolumes, pp("m_volumeWorlds"), pp("_vp"), pp("_svp"), pp("_eye"), float(W), float(H),
                                   pp("m_lightPt"), pp("m_lightColor"), pp("m_ambient"), self.m_frameIdx, pp("cbPerFrame"), pp("perObject"))

    def UpdateFrame(self, viewProj, shadowVP, eyePt):
        """UpdateFrame (MultiRayCaster.cpp:316-353). shadowVP is the TRANSPOSED light view-projection (m_shadowVP)."""
        self._frame_constants(viewProj, shadowVP, eyePt)
        self._check(self._lib.mv_update_frame(self._ctx, self._pp("cbPerFrame"), self._pp("perObject")), "mv_update_frame")

    def UpdateFrameRaw(self, cbPerFrame, perObject):
        cb = np.ascontiguousarray(cbPerFrame).view(np.uint8)
        po = np.ascontiguousarray(perObject).view(np.uint8)
        if cb.size != _lib.CB_PER_FRAME_BYTES or po.size != self.numVolumes * _lib.PER_OBJECT_BYTES:
            raise ValueError("CBPerFrame must be 212 bytes and PerObject V x 224 bytes")
        self._check(self._lib.mv_update_frame(self._ctx, _ptr(cb), _ptr(po)), "mv_update_frame")

    # ---- the path ----
    def cullVolumes(self):
        self._check(self._lib.mv_cull(self._ctx), "mv_cull")

    def rayMarchL(self):
        self._check(self._lib.mv_march_light(self._ctx), "mv_march_light")

    def rayMarchV(self):
        self._check(self._lib.mv_march_view(self._ctx), "mv_march_view")

    def composite(self, row0=0, row1=0, backgroundDevice=None):
        """cubeDepthPeel + renderCube + resolveOIT; backgroundDevice: device pointer of a W x H RGBA16F colour target to draw over."""
        if backgroundDevice is None:
            self._check(self._lib.mv_composite(self._ctx, row0, row1), "mv_composite")
        else:
            self._check(self._lib.mv_composite_over(self._ctx, row0, row1, C.c_void_p(backgroundDevice)), "mv_composite_over")

    def Render(self, composite=False):
        """Render (MultiRayCaster.cpp:355-385), non-work-graph branch; increments m_frameIdx like :384."""
        self._check(self._lib.mv_render(self._ctx, 1 if composite else 0), "mv_render")
        self.m_frameIdx += 1

    def synchronize(self):
        self._check(self._lib.mv_synchronize(self._ctx), "mv_synchronize")

    # ---- readback ----
    def ReadBackVolumeInfo(self):
        out = np.zeros((self.numVolumes, 4), np.uint16)
        self._check(self._lib.mv_readback_volume_info(self._ctx, _ptr(out)), "mv_readback_volume_info")
        return out

    def ReadBackVisibleLists(self):
        vis = np.zeros(self.numVolumes, np.uint32)
        cube = np.zeros(self.numVolumes, np.uint32)
        nv, nc = C.c_uint32(), C.c_uint32()
        self._check(self._lib.mv_readback_visible_lists(self._ctx, _ptr(vis), C.byref(nv), _ptr(cube), C.byref(nc)), "mv_readback_visible_lists")
        return vis[:nv.value].copy(), cube[:nc.value].copy()

    def ReadBackVolume(self, src):
        N = self.gridSize
        out = np.empty((N, N, N, 4), np.uint16)
        self._check(self._lib.mv_readback_volume(self._ctx, src, _ptr(out)), "mv_readback_volume")
        return out

    def ReadBackLightMap(self, volume):
        L = self.lightGridSize
        out = np.empty((L, L, L), np.uint32)
        self._check(self._lib.mv_readback_light_map(self._ctx, volume, _ptr(out)), "mv_readback_light_map")
        return out

    def UploadLightMap(self, volume, data):
        a = np.ascontiguousarray(data, np.uint32)
        if a.size != self.lightGridSize ** 3:
            raise ValueError("light map must be L^3 uint32")
        self._check(self._lib.mv_upload_light_map(self._ctx, volume, _ptr(a)), "mv_upload_light_map")

    def ReadBackCubeMap(self, volume, mip):
        d = max(1, self.gridSize >> mip)  # (a mip's edge as D3D defines it)
        out = np.empty((6, d, d, 4), np.uint16)
        self._check(self._lib.mv_readback_cube_map(self._ctx, volume, mip, _ptr(out)), "mv_readback_cube_map")
        return out

    def ReadBackCubeDepth(self, volume, mip):
        d = max(1, self.gridSize >> mip)  # (a mip's edge as D3D defines it)
        out = np.empty((6, d, d), np.float32)
        self._check(self._lib.mv_readback_cube_depth(self._ctx, volume, mip, _ptr(out)), "mv_readback_cube_depth")
        return out

    def ReadBackComposite(self):
        W, H = self.viewport
        out = np.empty((H, W, 4), np.uint16)
        self._check(self._lib.mv_readback_composite(self._ctx, _ptr(out)), "mv_readback_composite")
        return out

    def ClearCubeMaps(self):
        self._check(self._lib.mv_clear_cube_maps(self._ctx), "mv_clear_cube_maps")

    # ---- measurement / plumbing ----
    def GetStats(self):
        s = mv_stats()
        self._check(self._lib.mv_get_stats(self._ctx, C.byref(s)), "mv_get_stats")
        d = {k: getattr(s, k) for k, _ in mv_stats._fields_ if k != "reserved"}
        d["reserved0"] = s.reserved[0]  # marchV lane slots (64 x longest ray per wavefront)
        d["fetch_free_samples"] = s.reserved[2]  # view / RayCast samples whose texel fetch the brick occupancy bits saved
        d["fused_raycast_lit"] = s.reserved[1]  # lit samples of the RayCast tiles marched inside the view-march launch
        return d

    def ResetStats(self):
        self._check(self._lib.mv_reset_stats(self._ctx), "mv_reset_stats")

    def SetProfiling(self, on):
        self._check(self._lib.mv_set_profiling(self._ctx, 1 if on else 0), "mv_set_profiling")

    def GetStream(self):
        return self._lib.mv_get_stream(self._ctx)

    def GetBuffer(self, which, index=0):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.mv_get_buffer(self._ctx, which, index, C.byref(p), C.byref(n)), "mv_get_buffer")
        return p.value, n.value

    def ShardRows(self):
        """Screen rows [row0, row1) this shard composites and marches screen-space RayCast tiles for (mv_shard_rows)."""
        a, b = C.c_uint32(), C.c_uint32()
        self._check(self._lib.mv_shard_rows(self._ctx, C.byref(a), C.byref(b)), "mv_shard_rows")
        return a.value, b.value

    def ExchangeLayout(self):
        d, sb, tb = C.c_uint32(), C.c_size_t(), C.c_size_t()
        self._check(self._lib.mv_exchange_layout(self._ctx, C.byref(d), C.byref(sb), C.byref(tb)), "mv_exchange_layout")
        return d.value, sb.value, tb.value

    def ExchangePack(self, device_ptr, slot_dim):
        self._check(self._lib.mv_exchange_pack(self._ctx, C.c_void_p(device_ptr), slot_dim), "mv_exchange_pack")

    def ExchangeUnpack(self, device_ptr, slot_dim):
        self._check(self._lib.mv_exchange_unpack(self._ctx, C.c_void_p(device_ptr), slot_dim), "mv_exchange_unpack")

    # ---- the multi-GPU frame loop inside the library (mv_dist_*) ----
    @staticmethod
    def DistUniqueId():
        """Two RCCL unique ids (rank 0 calls this and hands the bytes to every rank)."""
        buf = (C.c_uint8 * _lib.MV_DIST_ID_BYTES)()
        rc_ = _lib.load().mv_dist_unique_id(buf)
        if rc_ != 0:
            raise _lib.MVError(rc_, "mv_dist_unique_id")
        return bytes(buf)

    @staticmethod
    def DistRcclAvailable():
        """librccl.so and the entry points the frame loop needs can be opened in this process (a local check: no collective)."""
        return _lib.load().mv_dist_rccl_available() == 1

    def DistInitRccl(self, id_bytes):
        buf = (C.c_uint8 * _lib.MV_DIST_ID_BYTES).from_buffer_copy(id_bytes)
        self._check(self._lib.mv_dist_init_rccl(self._ctx, buf), "mv_dist_init_rccl")

    def DistInitCallbacks(self, allgather, allreduce):
        """allgather(send_ptr, recv_ptr, nbytes, stream, which) / allreduce(ptr, count_u32, stream, which) -> 0 on success."""
        self._dist_cbs = (_lib.DIST_ALLGATHER_FN(lambda u, s_, r_, n, st, w: allgather(s_, r_, n, st, w)),
                          _lib.DIST_ALLREDUCE_FN(lambda u, b, n, st, w: allreduce(b, n, st, w)))  # kept alive with the context
        self._check(self._lib.mv_dist_init_callbacks(self._ctx, self._dist_cbs[0], self._dist_cbs[1], None), "mv_dist_init_callbacks")

    def DistPeerExport(self):
        """This rank's handles for the peer-store transport (MV_DIST_PEER_BYTES): gather every rank's, rank order, hand them to DistInitPeer."""
        buf = (C.c_uint8 * _lib.MV_DIST_PEER_BYTES)()
        self._check(self._lib.mv_dist_peer_export(self._ctx, buf), "mv_dist_peer_export")
        return bytes(buf)

    def DistInitPeer(self, all_blobs=None, loopback=False, fold=None, rows_to_root=False, deal_volumes=None, acquire=True, push_cells=False):
        """Attach the frame loop with the peer-store transport: every rank's kernels store into every rank's buffers (no collectives).
        loopback: a rank measured alone (every peer is the context itself).  fold: True / False force the flags into / out of the
        marches and the composite (None: the library decides -- folded when every rank owns its GPU)."""
        flags = 0 if fold is None else (_lib.MV_DIST_PEER_FOLD if fold else _lib.MV_DIST_PEER_NO_FOLD)
        if rows_to_root:  # only rank 0 assembles the whole frame
            flags |= _lib.MV_DIST_PEER_ROWS_TO_ROOT
        if deal_volumes is not None:  # whole cube-map volumes per rank, dealt by cost / every volume's tiles shared out (None: the library's choice -- volumes with one frame per launch, tiles with several)
            flags |= _lib.MV_DIST_PEER_DEAL_VOLUMES if deal_volumes else _lib.MV_DIST_PEER_DEAL_TILES
        if not acquire:  # round 4's folded waits: no invalidate behind the poll (A/B timing)
            flags |= _lib.MV_DIST_PEER_NO_ACQUIRE
        if push_cells:  # round 4's light push: whole quad-replicated cells instead of plain texels rebuilt by the receiver (A/B timing)
            flags |= _lib.MV_DIST_PEER_PUSH_CELLS
        if loopback:
            self._check(self._lib.mv_dist_init_peer(self._ctx, None, _lib.MV_DIST_PEER_LOOPBACK | flags), "mv_dist_init_peer")
            return
        buf = (C.c_uint8 * len(all_blobs)).from_buffer_copy(all_blobs)
        self._check(self._lib.mv_dist_init_peer(self._ctx, buf, flags), "mv_dist_init_peer")

    def DistPeerCoherence(self, acquire=True):
        """The in-kernel coherence test of the peer-store transport (collective: every rank calls it): (stale words behind a wait inside a
        running kernel, words the first read got wrong).  (0, 0) is what the folded waits need."""
        stale, first = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.mv_dist_peer_coherence(self._ctx, 1 if acquire else 0, C.byref(stale), C.byref(first)), "mv_dist_peer_coherence")
        return int(stale.value), int(first.value)

    def DistSetPeerTimeoutMs(self, ms):
        self._check(self._lib.mv_dist_set_peer_timeout_ms(self._ctx, int(ms)), "mv_dist_set_peer_timeout_ms")

    def DistSetInPlaceGather(self, enabled=True):
        """Caller-supplied collectives: the all-gather accepts send == recv + rank * bytes (NCCL's in-place convention)."""
        self._check(self._lib.mv_dist_set_inplace_gather(self._ctx, 1 if enabled else 0), "mv_dist_set_inplace_gather")

    def DistSetCullCrosscheck(self, enabled=True):
        """The loop's per-frame check that every rank's cull decided the same (on by default); off for a rank driven alone with
        collectives that deliver nothing (scripts/exp_rank_cost.py)."""
        self._check(self._lib.mv_dist_set_cull_crosscheck(self._ctx, 1 if enabled else 0), "mv_dist_set_cull_crosscheck")

    def DistShutdown(self):
        self._check(self._lib.mv_dist_shutdown(self._ctx), "mv_dist_shutdown")
        self._dist_cbs = None

    def DistSelfTest(self):
        self._check(self._lib.mv_dist_selftest(self._ctx), "mv_dist_selftest")

    def DistFrame(self, viewProj, shadowVP, eyePt, composite=True):
        """One frame of the multi-GPU loop: begins this frame (upload .. view march) and finishes the previous one (exchange, composite)."""
        self._frame_constants(viewProj, shadowVP, eyePt)
        self._check(self._lib.mv_dist_frame(self._ctx, self._pp("cbPerFrame"), self._pp("perObject"), 1 if composite else 0), "mv_dist_frame")
        self.m_frameIdx += 1

    def DistFlush(self):
        self._check(self._lib.mv_dist_flush(self._ctx), "mv_dist_flush")

    def DistInfo(self):
        """{"transport": "none" | "rccl" | "callbacks" | "peer", "ranks": communicator size as RCCL reports it, "communicators": n}"""
        t, r, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._lib.mv_dist_info(self._ctx, C.byref(t), C.byref(r), C.byref(c)), "mv_dist_info")
        return {"transport": ("none", "rccl", "callbacks", "peer")[t.value], "ranks": r.value, "communicators": c.value}

    def DistPeerInfo(self):
        """The peer-store transport's settings and its last coherence verdict (all zero / False on the other transports)."""
        out = (C.c_uint32 * 8)()
        self._check(self._lib.mv_dist_peer_info(self._ctx, out), "mv_dist_peer_info")
        return {"folded": bool(out[0]), "acquire": bool(out[1]), "rows_to_root": bool(out[2]), "deal_volumes": bool(out[3]), "light_push_bytes_per_texel": int(out[4]),
                "flags_memory": ("plain", "fine-grained", "uncached")[min(int(out[5]), 2)], "stale_words": int(out[6]), "first_read_bad": int(out[7])}

    def SetLatencySampling(self, every_n_frames):
        """Every n-th frame: host clock at the hand-over of its constants -> host clock behind its last kernel (0 = off)."""
        self._check(self._lib.mv_set_latency_sampling(self._ctx, int(every_n_frames)), "mv_set_latency_sampling")

    def GetLatency(self, reset=False):
        mean, mx, n = C.c_double(), C.c_double(), C.c_uint64()
        self._check(self._lib.mv_get_latency(self._ctx, C.byref(mean), C.byref(mx), C.byref(n), 1 if reset else 0), "mv_get_latency")
        return {"mean_ms": mean.value, "max_ms": mx.value, "samples": int(n.value)}

    def DistExchangedBytes(self, reset=False):
        n = C.c_uint64()
        self._check(self._lib.mv_dist_exchanged_bytes(self._ctx, C.byref(n), 1 if reset else 0), "mv_dist_exchanged_bytes")
        return n.value

    def ReadBackCull(self):
        """(VolumeInfo[V,4], visible list, cube-map list, min selected mip) with a single stream synchronisation."""
        info = np.zeros((self.numVolumes, 4), np.uint16)
        vis = np.zeros(self.numVolumes, np.uint32)
        cube = np.zeros(self.numVolumes, np.uint32)
        nv, nc, mm = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._lib.mv_readback_cull(self._ctx, _ptr(info), _ptr(vis), C.byref(nv), _ptr(cube), C.byref(nc), C.byref(mm)), "mv_readback_cull")
        return info, vis[:nv.value].copy(), cube[:nc.value].copy(), mm.value
