"""Fuzz of the call sequences a host may make on ONE GPU (round 6): the same seeded random sequence -- frames of every kind (whole, without light march,
without view march, composite only, composite before the march, idle), input setters between them (depth map as a host copy or a borrowed device pointer,
shadow map, light probe on / off, sample caps, light, a volume re-upload, a light-map upload, a viewport change), readbacks at random points -- on a one-stream
context and on the contexts that keep frames in flight (three chains + a cull stream; four chains with the culls on them: bench.py's default).  Everything read
back must be identical: the frames in flight, their rotating output sets, light buffers and frame slots, and every setter's synchronisation are invisible.
MV_API_FUZZ=n runs n seeds (default 4), MV_API_FUZZ_FIRST=k starts at seed k."""
import os

import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, _lib, scene
from multivolumes_amd import raycaster as rc

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("MV_API_FUZZ", "4"))
FIRST = int(os.environ.get("MV_API_FUZZ_FIRST", "0"))
KINDS = ["full", "full", "full", "full_no_composite", "no_light", "no_march", "composite_only", "idle", "composite_then_march"]
SETTERS = ["depth_host", "depth_device", "shadow", "sh", "max_samples", "light", "volume", "light_map", "viewport", "readback"]


# the second fuzz (later in round 6): the producers and the remaining setters between the frames -- the mesh rasterised into the context's own depth / shadow
# map, a volume's world transform, the ambient term, the volume -> texture table, mv_clear_cube_maps, profiling on / off, mv_reset_stats -- and composites in
# row bands or over a background
KINDS_X = KINDS + ["bands", "bands", "over", "over_bands"]
SETTERS_X = SETTERS + ["mesh_depth", "mesh_shadow", "world", "ambient", "descs", "clear_cube", "profiling", "reset_stats", "mesh"]


@pytest.mark.parametrize("seed", range(FIRST, FIRST + N_SEEDS))
def test_random_call_sequences_read_the_same_with_and_without_frames_in_flight(seed):
    run_sequence(seed)


@pytest.mark.parametrize("seed", range(FIRST, FIRST + N_SEEDS))
def test_random_call_sequences_with_the_producers_and_partial_composites(seed):
    run_sequence(seed, extended=True)


def make_script(seed, extended=False):
    """The whole sequence, drawn up front (so that leaving ops out by hand does not change the rest): (cfg, [(setters, kind, compare)] per frame)."""
    r = np.random.default_rng((7000 if not extended else 8000000) + seed)
    SETTERS, KINDS = (SETTERS_X, KINDS_X) if extended else (globals()["SETTERS"], globals()["KINDS"])
    cfg = dict(scene.CONFIGS["T2"], N=int(r.choice([24, 32, 40, 48])), L=int(r.integers(8, 40)), V=int(r.choice([4, 9, 16])),
               W=int(r.integers(96, 300)), H=int(r.integers(64, 220)), maxRay=int(r.choice([32, 48, 64])), maxLight=int(r.choice([16, 24, 32])))
    frames = []
    for f in range(int(r.integers(30, 60))):
        setters = []
        for _ in range(int(r.integers(0, 3))):
            what = SETTERS[int(r.integers(len(SETTERS)))]
            arg = {"depth_host": float(r.uniform(4.0, 12.0)), "depth_device": float(r.uniform(4.0, 12.0)), "shadow": float(r.uniform(4.0, 12.0)), "sh": bool(r.integers(2)),
                   "max_samples": (int(r.choice([16, 32, 48, 64, 96])), int(r.choice([8, 16, 24, 32]))),
                   "light": (float(r.uniform(40, 90)), float(r.uniform(40, 90)), -float(r.uniform(40, 90))),
                   "volume": (int(r.integers(cfg["V"])), int(r.integers(0, 40))), "light_map": (int(r.integers(cfg["V"])), int(r.integers(1 << 30))),
                   "viewport": (int(r.integers(96, 300)), int(r.integers(64, 220))), "readback": int(r.integers(cfg["V"])),
                   "mesh_depth": (float(r.uniform(0.5, 3.0)), bool(r.integers(2))), "mesh_shadow": float(r.uniform(0.5, 3.0)), "mesh": int(r.integers(1 << 30)),
                   "world": (int(r.integers(cfg["V"])), float(r.uniform(6.0, 30.0)), tuple(float(x) for x in r.normal(0.0, 14.0, size=3))),
                   "ambient": (float(r.uniform(0.0, 1.0)), float(r.uniform(0.0, 1.0)), float(r.uniform(0.0, 1.0)), float(r.uniform(0.0, 2.0))),
                   "descs": int(r.integers(1 << 30)), "clear_cube": None, "profiling": bool(r.integers(2)), "reset_stats": None}[what]
            setters.append((what, arg))
        frames.append((setters, KINDS[int(r.integers(len(KINDS)))], bool(r.random() < 0.4), float(r.random())))
    return cfg, frames


def random_mesh(seed):
    """A few dozen triangles around the origin (the rasteriser's own fuzz is tests/test_mesh.py): (positions, indices)."""
    r = np.random.default_rng(seed)
    n = int(r.integers(4, 60))
    centres = r.normal(0.0, 5.0, size=(n, 1, 3))
    pos = (centres + r.normal(0.0, 3.0, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    return pos, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def run_sequence(seed, disable=(), verbose=False, only_kinds=None, variants=None, stop_after=None, script=None, extended=False):
    """(also driven by hand: python -c "from tests.test_api_fuzz import run_sequence; run_sequence(0, disable=('viewport',), verbose=True)")"""
    import torch
    cfg, drawn = make_script(seed, extended)
    script = drawn if script is None else script
    if stop_after is not None:
        script = script[:stop_after]
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    if variants is None:
        P, CV, RQ = _lib.MV_FLAG_PIPELINE_FRAMES, _lib.MV_FLAG_CULL_ON_VIEW_STREAM, _lib.MV_FLAG_RAY_QUEUE
        variants = [(0, 0), (P, 0), (P | CV, 4)]
        # every third seed also: the other forms of a context -- the ray-queue march (one stream and with frames in flight), list order instead of the
        # cost-binned tile order, two and five frames in flight, the single view stream of the multi-GPU loop's other schedule
        if seed % 3 == 0 or os.environ.get("MV_API_FUZZ_ALL_CONTEXTS") == "1":
            variants += [(RQ, 0), (P | RQ, 0), (P | _lib.MV_FLAG_LIST_ORDER, 0), (P, 2), (P, 5), (P | CV, 3), (P | _lib.MV_FLAG_SINGLE_VIEW_STREAM, 0),
                         (P | CV, 4, 3), (P | CV, 4, 4)]  # (... and B = 3, 4 consecutive frames per view stream: mv_config.frames_per_launch outside the multi-GPU loop)
    cs = [MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=v[0], framesInFlight=v[1], framesPerLaunch=v[2] if len(v) > 2 else 0) for v in variants]
    label = f"seed {seed}: {cfg}"
    print(label)
    try:
        frs = [scene.setup_raycaster(c, cfg, vols, use_sh=True) for c in cs]
        W, H = cfg["W"], cfg["H"]
        shadow_vp = frs[0]["shadowVP"]
        keep = []  # device tensors the contexts borrow: alive until the end
        bg = None
        compared = 0
        if extended:
            for c in cs:
                c.SetMesh(*random_mesh(seed))
        for f, (setters, kind, compare, *rest) in enumerate(script):
            a = 0.05 * f + 0.3 * seed
            s = 1.0 - 0.5 * np.sin(0.21 * f) ** 2
            e = np.array([85.0 * s * np.sin(a), 16.0 + 14.0 * np.sin(0.37 * a), -85.0 * s * np.cos(a)], np.float32)
            proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, W / H, 1.0, 1000.0)
            vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
            # ---- setters in front of the frame ----
            for what, arg in setters:
                if verbose:
                    print(f"  frame {f}: setter {what} {arg}" + (" (disabled)" if what in disable else ""), flush=True)
                if what in disable:
                    continue
                if what == "depth_host":
                    d = rc.synth_sphere_depth(W, H, vp, (0, 0, 0), arg)
                    for c in cs:
                        c.SetRenderTargets(d)
                elif what == "depth_device":
                    d = torch.from_numpy(rc.synth_sphere_depth(W, H, vp, (0, 0, 0), arg)).cuda()
                    keep.append(d)
                    for c in cs:
                        c.SetRenderTargets(depthMapDevice=d.data_ptr())
                elif what == "shadow":
                    sh = rc.synth_sphere_shadow(64, shadow_vp, (0, 0, 0), arg)
                    for c in cs:
                        c.SetRenderTargets(shadowMap=sh)
                elif what == "sh":
                    for c in cs:
                        c.SetSH(scene.DEFAULT_SH if arg else None)
                elif what == "max_samples":
                    for c in cs:
                        c.SetMaxSamples(*arg)
                elif what == "light":
                    for c in cs:
                        c.SetLight(arg, scene.LIGHT_COLOR, scene.LIGHT_INTENSITY)
                elif what == "volume":
                    v = rc.synth_volume(N, arg[1])
                    for c in cs:
                        c.LoadVolumeData(arg[0], v)
                elif what == "light_map":
                    lm = np.random.default_rng(arg[1]).integers(0, 2 ** 32, size=(L, L, L), dtype=np.uint64).astype(np.uint32) & np.uint32(0x3BFFF7FF)  # (finite 11/11/10-bit floats)
                    for c in cs:
                        c.UploadLightMap(arg[0], lm)
                elif what == "viewport":
                    W, H = arg
                    proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, W / H, 1.0, 1000.0)
                    vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
                    d = rc.synth_sphere_depth(W, H, vp, (0, 0, 0), 8.0)
                    for c in cs:
                        c.SetViewport(W, H)
                        c.SetRenderTargets(d)
                elif what == "mesh":
                    for c in cs:
                        c.SetMesh(*random_mesh(arg))
                elif what == "mesh_depth":  # the context rasterises its own depth map (ObjectRenderer's depth pass), with or without the TAA jitter
                    wvp = rc.multiply(rc.mesh_world(arg[0], (0.0, 0.0, 0.0)), vp)
                    for c in cs:
                        c.RenderMeshDepth(wvp, (0.25 / W, -0.4 / H) if arg[1] else None)
                elif what == "mesh_shadow":
                    swvp = rc.multiply(rc.mesh_world(arg, (0.0, 0.0, 0.0)), np.asarray(shadow_vp, np.float32).reshape(4, 4).T.copy())
                    for c in cs:
                        c.RenderMeshShadow(swvp)
                elif what == "world":
                    for c in cs:
                        c.SetVolumeWorld(*arg)
                elif what == "ambient":
                    for c in cs:
                        c.SetAmbient(arg[:3], arg[3])
                elif what == "descs":  # another volume -> texture table (VolTexId = a random source each; mask bits and size as before)
                    tex = np.random.default_rng(arg).integers(0, V, size=V)
                    for c in cs:
                        c.SetVolumeDescs(np.array([int(t) | (5 << 14) | (N << 18) for t in tex], np.uint32))
                elif what == "clear_cube":
                    for c in cs:
                        c.ClearCubeMaps()
                elif what == "profiling":
                    for c in cs:
                        c.SetProfiling(arg)
                elif what == "reset_stats":
                    for c in cs:
                        c.ResetStats()
                else:
                    got = [(c.ReadBackVolumeInfo(), c.ReadBackLightMap(arg)) for c in cs]
                    for g in got[1:]:
                        assert np.array_equal(g[0], got[0][0]) and np.array_equal(g[1], got[0][1]), f"{label}: frame {f}: a readback between frames differs"
            # ---- the frame ----
            if only_kinds is not None and kind not in only_kinds:
                kind = "full"
            if verbose:
                print(f"frame {f}: {kind}" + (" + compare" if compare else ""), flush=True)
            for c in cs:
                c.UpdateFrame(vp, shadow_vp, e)
                c.cullVolumes()
                if kind in ("full", "full_no_composite", "no_march", "composite_then_march"):
                    c.rayMarchL()
                if kind == "composite_then_march":
                    c.composite()
                    c.rayMarchV()
                else:
                    if kind in ("full", "full_no_composite", "no_light"):
                        c.rayMarchV()
                    if kind in ("full", "no_light", "no_march", "composite_only"):
                        c.composite()
                if kind in ("bands", "over", "over_bands"):  # (whole frames whose composite comes in row bands and / or over a colour target)
                    c.rayMarchL()
                    c.rayMarchV()
                    if kind != "bands":
                        if bg is None or tuple(bg.shape[:2]) != (H, W):
                            bg = torch.from_numpy(np.random.default_rng(seed).random((H, W, 4), dtype=np.float32).astype(np.float16)).cuda()
                            keep.append(bg)
                    back = None if kind == "bands" else bg.data_ptr()
                    if kind == "over":
                        c.composite(backgroundDevice=back)
                    else:
                        cut = sorted({0, H, 1 + int(rest[0] * (H - 1)), 1 + int((rest[0] * 7.3 % 1.0) * (H - 1))})
                        order = list(zip(cut[:-1], cut[1:]))
                        for r0, r1 in (order[::-1] if f % 2 else order):
                            c.composite(r0, r1, backgroundDevice=back)
                c.m_frameIdx += 1
            # ---- compare at random points (a readback drains the frames in flight: most frames go by without one) ----
            # (behind a frame that composited: the image belongs to the frame's output set, include/multivolumes.h mv_readback_composite)
            if kind in ("full", "no_light", "no_march", "composite_only", "composite_then_march", "bands", "over", "over_bands") and compare:
                imgs = [c.ReadBackComposite() for c in cs]
                infos = [c.ReadBackVolumeInfo() for c in cs]
                for i in range(1, len(cs)):
                    assert np.array_equal(infos[i], infos[0]), f"{label}: frame {f} ({kind}): VolumeInfo of context {i}"
                    assert np.array_equal(imgs[i], imgs[0]), f"{label}: frame {f} ({kind}): {int((imgs[i] != imgs[0]).sum())} composite halfs of context {i} differ"
                compared += 1
        assert compared >= 3 or stop_after is not None or script is not drawn
        # the end: a whole frame (cube maps are read from the set of a frame that marched), then every light map, every cube map and depth of the
        # cube-map volumes at their mip, the counters
        imgs = []
        for c in cs:
            c.UpdateFrame(vp, shadow_vp, e)
            c.Render(composite=True)
            imgs.append(c.ReadBackComposite())
        for i in range(1, len(cs)):
            assert np.array_equal(imgs[i], imgs[0]), f"{label}: the final frame of context {i}"
        info = cs[0].ReadBackVolumeInfo()
        _, cube = cs[0].ReadBackVisibleLists()
        st = [c.GetStats() for c in cs]
        for i in range(1, len(cs)):
            for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples"):
                assert st[i][k] == st[0][k], f"{label}: counter {k} of context {i}"
            for v in range(V):
                assert np.array_equal(cs[i].ReadBackLightMap(v), cs[0].ReadBackLightMap(v)), f"{label}: light map {v} of context {i}"
            for v in (int(x) for x in cube):
                mip = int(info[v, 0])
                assert np.array_equal(cs[i].ReadBackCubeMap(v, mip), cs[0].ReadBackCubeMap(v, mip)), f"{label}: cube map {v} of context {i}"
                assert np.array_equal(cs[i].ReadBackCubeDepth(v, mip), cs[0].ReadBackCubeDepth(v, mip)), f"{label}: cube depth {v} of context {i}"
    finally:
        for c in cs:
            c.close()
