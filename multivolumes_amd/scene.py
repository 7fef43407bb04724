"""The benchmark / test scenes of BASELINE.json (C1..C5) built from the reference's own defaults
(MultiVolumes.cpp:38-47, 264-279, 339-346; all{4,16,64}.bat) and the synthetic inputs of SURVEY.md 8(d)."""
import math
import os

import numpy as np

from . import raycaster as rc

FOV_ANGLE_Y = math.pi / 4  # MultiVolumes.cpp:21
Z_NEAR, Z_FAR = 1.0, 1000.0  # SharedConsts.h:9-10
LIGHT_PT = (75.0, 75.0, -75.0)  # MultiVolumes.cpp:339
LIGHT_COLOR, LIGHT_INTENSITY = (1.0, 0.7, 0.3), 3.0 * math.pi  # :340, :342
AMBIENT_COLOR, AMBIENT_INTENSITY = (0.4, 0.6, 1.0), 2.0 * math.pi  # :341-342
VOLUME_SIZE = 20.0  # m_volPosScale.w * 2 (MultiVolumes.cpp:45, 222)
MESH_SCENE_SIZE = 10.03 * 1.8  # bunny.obj max AABB extent x meshPosScale.w (ObjectRenderer.cpp:76)
SPHERE_RADIUS = 8.0  # analytic stand-in for the bunny

CONFIGS = {
    # name: N, L, V, viewport, maxRaySamples, maxLightSamples, mesh depth
    "C1": dict(N=64, L=48, V=1, W=256, H=256, maxRay=32, maxLight=32, mesh=False),
    "C2": dict(N=128, L=96, V=4, W=512, H=512, maxRay=64, maxLight=96, mesh=False),
    "C3": dict(N=256, L=96, V=16, W=1024, H=1024, maxRay=128, maxLight=96, mesh=True),
    "C5": dict(N=256, L=96, V=64, W=2048, H=2048, maxRay=256, maxLight=96, mesh=True),
    # small stand-ins with the same structure, for parity tests the oracle finishes in seconds
    "T1": dict(N=32, L=24, V=4, W=256, H=256, maxRay=48, maxLight=24, mesh=True),
    "T2": dict(N=48, L=32, V=16, W=512, H=512, maxRay=64, maxLight=32, mesh=True),
}


def camera(cfg, eye=None):
    """View-projection of the reference's default camera (MultiVolumes.cpp:264-279), pulled back for big grids."""
    row_length = math.ceil(math.sqrt(cfg["V"]))
    if eye is None:
        s = max(1.0, row_length / 4.0)
        eye = (4.0 * s, 16.0 * s, -80.0 * s)
    view = rc.look_at_lh(eye, (0.0, 0.0, 0.0))
    proj = rc.perspective_fov_lh(FOV_ANGLE_Y, cfg["W"] / float(cfg["H"]), Z_NEAR, Z_FAR)
    return np.asarray(eye, np.float32), rc.multiply(view, proj)


def shadow_vp(cfg):
    scene = MESH_SCENE_SIZE if cfg["mesh"] else 1.0  # ObjectRenderer.cpp:46 default m_sceneSize
    return rc.shadow_view_proj(LIGHT_PT, scene)


def depth_and_shadow(cfg, view_proj, shadow_vp_t, M):
    W, H = cfg["W"], cfg["H"]
    if cfg["mesh"]:
        depth = rc.synth_sphere_depth(W, H, view_proj, (0.0, 0.0, 0.0), SPHERE_RADIUS)
        shadow = rc.synth_sphere_shadow(M, shadow_vp_t, (0.0, 0.0, 0.0), SPHERE_RADIUS)
    else:
        depth = np.ones((H, W), np.float32)
        shadow = np.full((M, M), 65535, np.uint16)
    return depth, shadow


MESH_POS, MESH_SCALE = (0.0, -9.0, 0.0), 1.8  # `-mesh Assets/bunny.obj 0.0 -9.0 0.0 1.8` (Bin/all16.bat:1, all64.bat:1)
BUNNY_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bunny_mesh.npz")


def load_bunny():
    """(positions[n, 3] f32, indices[m, 3] u32) of the reference scenes' mesh (fixture made by tests/golden/make_bunny_fixture.py)."""
    d = np.load(BUNNY_FIXTURE)
    return d["positions"], d["indices"]


def mesh_matrices(cfg, view_proj, scale=MESH_SCALE, pos=MESH_POS):
    """(world * viewProj, world * lightViewProj) of the mesh, row-major: what ObjectRenderer::UpdateFrame uploads
    (ObjectRenderer.cpp:171-207, no TAA jitter)."""
    world = rc.mesh_world(scale, pos)
    svp_t = shadow_vp(dict(cfg, mesh=True))  # transposed light view-projection
    return rc.multiply(world, view_proj), rc.multiply(world, svp_t.T.copy())


def render_mesh_targets(caster, cfg, view_proj, mesh=None):
    """Depth + shadow maps of the reference scenes' mesh, rasterised on the GPU by the context (mv_mesh_render_*)."""
    if mesh is not None:
        caster.SetMesh(*mesh)
    wvp, swvp = mesh_matrices(cfg, view_proj)
    caster.RenderMeshDepth(wvp)
    caster.RenderMeshShadow(swvp)


DEFAULT_SH = np.array([
    0.80, 0.90, 1.00, 0.10, 0.15, 0.25, 0.30, 0.28, 0.25, -0.05, -0.02, 0.04,
    0.02, 0.01, -0.03, 0.06, 0.05, 0.02, 0.09, 0.07, 0.05, -0.04, -0.03, -0.01, 0.03, 0.05, 0.08,
], np.float32)  # fixed 27-float "sky" for the AO-path parity run


def setup_raycaster(caster, cfg, volumes=None, use_sh=False, eye=None):
    """Apply a config to a MultiRayCaster the way MultiVolumes::LoadAssets / OnUpdate do. Returns the frame inputs."""
    caster.SetVolumesWorld(VOLUME_SIZE, (0.0, 0.0, 0.0))
    caster.SetMaxSamples(cfg["maxRay"], cfg["maxLight"])
    caster.SetLight(LIGHT_PT, LIGHT_COLOR, LIGHT_INTENSITY)
    caster.SetAmbient(AMBIENT_COLOR, AMBIENT_INTENSITY)
    eye, vp = camera(cfg, eye)
    svp = shadow_vp(cfg)
    depth, shadow = depth_and_shadow(cfg, vp, svp, caster.shadowMapSize)
    caster.SetRenderTargets(depth, shadow)
    caster.SetSH(DEFAULT_SH if use_sh else None)
    if volumes is not None:
        for i, v in enumerate(volumes):
            caster.LoadVolumeData(i, v)
    return dict(eye=eye, viewProj=vp, shadowVP=svp, depth=depth, shadow=shadow, sh=DEFAULT_SH if use_sh else None)
