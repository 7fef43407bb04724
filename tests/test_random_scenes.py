"""Randomised parity sweep: the whole path (cull, V light marches, view march, composite) against the oracle on
seeded random scenes -- ragged grid sizes, random eyes (outside, inside and between the volumes), sparse / dense / noisy
volumes, SH on and off, with and without mesh depth.  Everything must match bit for bit, counters included."""
import os

import numpy as np
import pytest

from multivolumes_amd import scene
from multivolumes_amd import raycaster as rc
from tests.helpers import oracle_pipeline
from tests.test_gpu_parity import compare_cube, compare_light, make_caster, oracle_composite, run_gpu

pytestmark = pytest.mark.gpu


def random_volume(rng, N, kind):
    v = rc.synth_volume(N, int(rng.integers(0, 6))).view(np.float16).copy()
    if kind == "sparse":
        keep = rng.random((-(-N // 4),) * 3) < rng.uniform(0.1, 0.6)
        mask = np.repeat(np.repeat(np.repeat(keep, 4, 0), 4, 1), 4, 2)[:N, :N, :N]
        v[..., 3] = np.where(mask, v[..., 3], np.float16(0))
    elif kind == "noise":
        v[..., 3] = (rng.random((N, N, N)) ** 3).astype(np.float16)
        v[..., :3] = rng.random((N, N, N, 3)).astype(np.float16)
    elif kind == "thin":
        v[..., 3] = np.where(rng.random((N, N, N)) < 0.02, np.float16(1.0), np.float16(0.004))
    elif kind == "special":
        # halfs no authoring tool would write, sprinkled over every channel: +-inf, NaNs, denormals, -0, +-65504, +-1, and the three halfs around ZERO_THRESHOLD
        # (0x211f is the folded constant itself).  The reference's arithmetic decides what comes out (NaN-ignoring min / max, 0 x inf inside GetStep): both
        # sides must agree texel for texel, NaNs at the same places.
        special = np.array([0x7c00, 0xfc00, 0x7e00, 0xfe00, 0x0001, 0x8001, 0x03ff, 0x8000, 0x7bff, 0xfbff, 0x3c00, 0xbc00, 0x211f, 0x2120, 0x211e], np.uint16)
        u = v.view(np.uint16)
        n = int(rng.integers(1, max(2, N * N)))
        idx = rng.integers(0, N, size=(n, 3))
        u[idx[:, 0], idx[:, 1], idx[:, 2], rng.integers(0, 4, size=n)] = special[rng.integers(0, len(special), size=n)]
    return v.view(np.uint16)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MV_RANDOM_SEEDS", "12"))))  # (a longer sweep: MV_RANDOM_SEEDS=300)
def test_random_scene(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([12, 17, 20, 24, 30, 36, 40]))
    L = int(rng.choice([6, 9, 12, 16]))
    V = int(rng.choice([1, 2, 4, 9]))
    W, H = int(rng.choice([96, 130, 200])), int(rng.choice([64, 100, 160]))
    cfg = dict(N=N, L=L, V=V, W=W, H=H, maxRay=int(rng.choice([16, 40, 100])), maxLight=int(rng.choice([8, 20, 48])), mesh=bool(rng.integers(0, 2)))
    kinds = rng.choice(["plain", "sparse", "noise", "thin"] + (["special"] if seed % 4 == 3 else []), size=V)  # (every fourth seed may draw the special halfs: the earlier seeds keep their scenes)
    vols = [random_volume(rng, N, k) for k in kinds]
    r = rng.uniform(8, 140)
    d = rng.normal(size=3)
    eye = tuple((d / np.linalg.norm(d) * r).astype(np.float32))
    use_sh = bool(rng.integers(0, 2))
    M = 64
    ref = oracle_pipeline(oracle, cfg, vols, M, sh=scene.DEFAULT_SH if use_sh else None, eye=eye)
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols, use_sh=use_sh, eye=eye)
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], ref["info"][ref["vis"]]), "VolumeInfo of the visible volumes"
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        if len(ref["cube"]):
            worst, mism, total = compare_cube(oracle, c, ref, cfg, tol=0.0)
            assert mism == 0
        # the K-buffer OIT composite of the same frame (CubeCast layers + screen-space RayCast layers)
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        got = c.ReadBackComposite()
        gf, ef = got.view(np.float16), np.asarray(exp).view(np.float16)
        assert np.array_equal(np.isnan(gf), np.isnan(ef))
        assert np.array_equal(np.where(np.isnan(gf), 0, got), np.where(np.isnan(ef), 0, np.asarray(exp))), (cfg, kinds, eye)
        st = c.GetStats()
    assert st["composite_samples"] == exp_samples
    for k in ("rays", "view_samples", "lit_samples", "light_samples"):
        assert st[k] == ref["stats"][k], (k, cfg, kinds, eye)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MV_RANDOM_PLACED_SEEDS", "8"))))
def test_random_scene_with_volumes_placed_at_will(oracle, seed):
    """What the reference's grid of equal cubes never shows: volumes of different sizes anywhere -- overlapping, nested, one containing the eye, far apart
    (SetVolumeWorld per volume: MultiRayCaster.cpp:297-303 takes any size and position) --, fewer volume textures than volumes (VolTexId = i % S,
    MultiRayCaster.cpp:471-479) and another light position.  Light rays cross several boxes at once, fragments of overlapping cubes interleave in the
    K-buffer, cube maps of aliased sources differ only by their transforms.  Everything against the oracle, bit for bit."""
    from multivolumes_amd import _lib
    from tests.helpers import host_frame
    rng = np.random.default_rng(9000 + seed)
    N = int(rng.choice([12, 17, 24, 30, 36]))
    L = int(rng.choice([6, 9, 12, 16]))
    V = int(rng.choice([2, 3, 5, 8]))
    S = int(rng.integers(1, V + 1))
    W, H = int(rng.choice([96, 130, 200])), int(rng.choice([64, 100, 160]))
    cfg = dict(N=N, L=L, V=V, W=W, H=H, maxRay=int(rng.choice([16, 40, 100])), maxLight=int(rng.choice([8, 20, 48])), mesh=bool(rng.integers(0, 2)))
    vols = [random_volume(rng, N, k) for k in rng.choice(["plain", "sparse", "noise", "thin", "special"], size=S)]
    worlds = np.zeros((V, 12), np.float32)
    lib = _lib.load()
    placed = []
    for i in range(V):
        size = float(np.exp(rng.uniform(np.log(4.0), np.log(60.0))))
        pos = rng.normal(0.0, 18.0, size=3).astype(np.float32)
        if i and rng.random() < 0.4:  # on top of / inside an earlier one
            pos = (placed[int(rng.integers(len(placed)))][1] + rng.normal(0.0, 3.0, size=3)).astype(np.float32)
        placed.append((size, pos))
        lib.mvh_set_volume_world(size, rc._ptr(rc._f32(pos, 3)), rc._ptr(worlds[i]))
    r = rng.uniform(8, 140)
    d = rng.normal(size=3)
    eye = tuple((d / np.linalg.norm(d) * r).astype(np.float32))
    if rng.random() < 0.25:  # inside one of the volumes
        eye = tuple((placed[0][1] + rng.uniform(-0.3, 0.3, size=3) * placed[0][0]).astype(np.float32))
    light = tuple(float(x) for x in rng.normal(0.0, 60.0, size=3) + np.array([0.0, 40.0, 0.0]))
    use_sh = bool(rng.integers(0, 2))
    M = 64
    ref = oracle_pipeline(oracle, cfg, vols, M, sh=scene.DEFAULT_SH if use_sh else None, eye=eye, worlds=worlds, light=light)
    with make_caster(cfg, M, S=S) as c:
        fr = scene.setup_raycaster(c, cfg, vols, use_sh=use_sh, eye=eye)
        for i, (size, pos) in enumerate(placed):
            c.SetVolumeWorld(i, size, pos)
        c.SetLight(light, scene.LIGHT_COLOR, scene.LIGHT_INTENSITY)
        for f in range(V):
            c.m_frameIdx = f
            c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            c.cullVolumes()
            c.rayMarchL()
        c.rayMarchV()
        c.synchronize()
        assert np.array_equal(np.frombuffer(c.perObject, np.uint8), np.frombuffer(ref["po"], np.uint8)), "the two sides were handed the same PerObject blobs"
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], ref["info"][ref["vis"]]), "VolumeInfo of the visible volumes"
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        if len(ref["cube"]):
            assert compare_cube(oracle, c, ref, cfg, tol=0.0)[1] == 0
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        got = c.ReadBackComposite()
        gf, ef = got.view(np.float16), np.asarray(exp).view(np.float16)
        assert np.array_equal(np.isnan(gf), np.isnan(ef))
        assert np.array_equal(np.where(np.isnan(gf), 0, got), np.where(np.isnan(ef), 0, np.asarray(exp))), (cfg, placed, eye)
        st = c.GetStats()
    assert st["composite_samples"] == exp_samples
    for k in ("rays", "view_samples", "lit_samples", "light_samples"):
        assert st[k] == ref["stats"][k], (k, cfg, eye)


def test_sixty_four_volumes(oracle):
    """V = 64 (eight cull workgroups + the ticket, an 8 x 8 grid of volumes): with a small viewport most volumes are on the
    screen-space scheme, more than the 32 whose RayCast joins the view-march launch, so the composite's inline fallback runs
    beside the pre-marched layers; everything against the oracle."""
    cfg = dict(N=16, L=8, V=64, W=192, H=128, maxRay=24, maxLight=8, mesh=True)
    vols = [rc.synth_volume(cfg["N"], k % 7) for k in range(cfg["V"])]
    eye = (16.0, 64.0, -320.0)  # the default eye scaled by rowLength / 4 (MultiVolumes.cpp:264-279), pulled back a little
    M = 64
    ref = oracle_pipeline(oracle, cfg, vols, M, frames=8, eye=eye)
    scr = [v for v in ref["vis"] if not (int(ref["info"][v, 2]) & 0x8000)]
    assert len(ref["vis"]) >= 48 and len(scr) > 32, (len(ref["vis"]), len(scr))
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols, frames=8, eye=eye)
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], ref["info"][ref["vis"]])
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        compare_light(oracle, c, ref, cfg, [v for v in ref["vis"] if ref["lightMaps"][v].any()], tol=0.0)
        if len(ref["cube"]):
            assert compare_cube(oracle, c, ref, cfg, tol=0.0)[1] == 0
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        got = c.ReadBackComposite()
        gf, ef = got.view(np.float16), np.asarray(exp).view(np.float16)
        assert np.array_equal(np.isnan(gf), np.isnan(ef))
        assert np.array_equal(np.where(np.isnan(gf), 0, got), np.where(np.isnan(ef), 0, np.asarray(exp)))
        assert c.GetStats()["composite_samples"] == exp_samples > 0


@pytest.mark.parametrize("V,eye", [(256, (16.0, 64.0, -320.0)), (1024, (32.0, 128.0, -640.0)), (100, (10.0, 40.0, -200.0))])  # (the default eye x rowLength / 4, MultiVolumes.cpp:264-279)
def test_many_volumes(oracle, V, eye):
    """V = 256 and V = MV_MAX_VOLUMES = 1024 (a 16 x 16 and a 32 x 32 grid; V = 100: a 10 x 10 grid): 32 and 128 cull workgroups, volume indices beyond a
    wavefront's 64 lanes everywhere they are ballots or bit masks (the light march's candidate words, the composite's per-tile skip, the work list), hundreds of
    entries in the march's work list, far more screen-space volumes than RayCast slots.  Small volumes and a small viewport keep the oracle in seconds."""
    cfg = dict(N=8, L=4, V=V, W=160, H=96, maxRay=16, maxLight=8, mesh=True)
    vols = [rc.synth_volume(cfg["N"], k % 5) for k in range(8)]
    M = 64
    frames = 12
    ref = oracle_pipeline(oracle, cfg, vols, M, frames=frames, eye=eye)
    assert len(ref["vis"]) > V // 3, len(ref["vis"])
    with make_caster(cfg, M, S=8) as c:
        run_gpu(c, cfg, vols, frames=frames, eye=eye)
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], ref["info"][ref["vis"]])
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        compare_light(oracle, c, ref, cfg, [v for v in ref["vis"] if ref["lightMaps"][v].any()], tol=0.0)
        if len(ref["cube"]):
            assert compare_cube(oracle, c, ref, cfg, tol=0.0)[1] == 0
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        got = c.ReadBackComposite()
        gf, ef = got.view(np.float16), np.asarray(exp).view(np.float16)
        assert np.array_equal(np.isnan(gf), np.isnan(ef))
        assert np.array_equal(np.where(np.isnan(gf), 0, got), np.where(np.isnan(ef), 0, np.asarray(exp)))
        st = c.GetStats()
        assert st["composite_samples"] == exp_samples
        for k in ("rays", "view_samples", "lit_samples", "light_samples"):
            assert st[k] == ref["stats"][k], k


def special_inputs(rng, cfg, M, vp):
    """Inputs no tool would hand over (round 6): a depth map with NaN, +-inf, negatives and values beyond 1; uploaded light maps with the 11- / 10-bit patterns
    of infinity and NaN, the largest finite value, denormals and zeros; a light probe with huge, negative and non-finite coefficients."""
    W, H, L, V = cfg["W"], cfg["H"], cfg["L"], cfg["V"]
    depth = rc.synth_sphere_depth(W, H, vp, (0, 0, 0), 8.0)
    n = int(rng.integers(1, max(2, W * H // 4 + 1)))
    ys, xs = rng.integers(0, H, size=n), rng.integers(0, W, size=n)
    depth[ys, xs] = np.array([np.nan, np.inf, -np.inf, -0.5, 0.0, 1.0, 1.5, 0.999999, 1e-30, -0.0], np.float32)[rng.integers(0, 10, size=n)]
    lms = rng.integers(0, 2 ** 32, size=(V, L, L, L), dtype=np.uint64).astype(np.uint32) & np.uint32(0x3BFFF7FF)   # finite 11/11/10-bit floats ...
    k = int(rng.integers(1, L * L))
    idx = rng.integers(0, L, size=(k, 3))
    vsel = rng.integers(0, V, size=k)
    special = np.array([0x7C0, 0x7FF, 0x7BF, 0x001, 0x03F, 0x000, 0x3C0], np.uint32)   # inf, NaN, largest finite, smallest denormal, largest denormal, 0, 1.0 (11-bit patterns)
    r11, g11, b10 = special[rng.integers(0, 7, size=k)], special[rng.integers(0, 7, size=k)], special[rng.integers(0, 7, size=k)] >> 1
    lms[vsel, idx[:, 0], idx[:, 1], idx[:, 2]] = r11 | (g11 << 11) | (b10 << 22)     # ... and these
    sh = scene.DEFAULT_SH.copy()
    sh[rng.integers(0, 27, size=4)] = np.array([1e30, -1e30, np.inf, np.nan], np.float32)[rng.integers(0, 4, size=4)]
    return depth, lms, sh


@pytest.mark.parametrize("seed", range(int(os.environ.get("MV_SPECIAL_INPUT_SEEDS", "8"))))
def test_special_inputs_against_the_oracle(oracle, seed):
    """View march, composite and light march fed non-finite and extreme inputs (special_inputs above), extreme sample caps (1 and 65535) and viewports of a
    few pixels: whatever the reference's arithmetic makes of them -- NaN-ignoring min / max, saturate(NaN) = 0, comparisons that are false -- both sides make the
    same: NaNs at the same halfs, every other bit equal."""
    from tests.helpers import host_frame
    rng = np.random.default_rng(12000 + seed)
    N = int(rng.choice([12, 17, 24, 32]))
    L = int(rng.choice([4, 7, 12]))
    V = int(rng.choice([1, 2, 4]))
    W, H = ((int(rng.integers(1, 12)), int(rng.integers(1, 12))) if seed % 4 == 0 else (int(rng.choice([64, 130])), int(rng.choice([48, 100]))))
    cfg = dict(N=N, L=L, V=V, W=W, H=H, maxRay=int(rng.choice([1, 2, 40, 65535])), maxLight=int(rng.choice([1, 3, 20, 65535])), mesh=True)
    vols = [random_volume(rng, N, k) for k in rng.choice(["plain", "noise", "special"], size=V)]
    r = rng.uniform(8, 100)
    d = rng.normal(size=3)
    eye = tuple((d / np.linalg.norm(d) * r).astype(np.float32))
    M = 64
    descs = np.array([(i % V) | (5 << 14) | (N << 18) for i in range(V)], np.uint32)
    eye_, vp, svp, cb, po = host_frame(cfg, 0, eye)
    depth, lms, sh = special_inputs(rng, cfg, M, vp)
    shadow = rc.synth_sphere_shadow(M, svp, (0, 0, 0), 8.0)
    info, vis, cube = oracle.cull(V, po, descs, cb, cfg["maxRay"])
    cm, cd, st = oracle.march_view(V, N, L, W, H, vols, lms, depth, po, cb, info, cube)
    img, comp_samples = oracle.composite(V, N, L, W, H, vols, lms, depth, po, cb, info, vis, cm, cd)
    # the light march with the bad probe (its own frame: it overwrites one of the uploaded maps)
    vid, lms2, nls = oracle.march_light(V, N, L, M, vols, descs, po, cb, shadow, sh, vis, cfg["maxLight"], lms.copy()) if len(vis) else (0, lms, 0)

    def same(a, b, what):
        fa, fb = np.asarray(a).view(np.float16), np.asarray(b).view(np.float16)
        assert np.array_equal(np.isnan(fa), np.isnan(fb)), f"seed {seed}: NaNs of {what} at other places"
        assert np.array_equal(np.where(np.isnan(fa), 0, np.asarray(a)), np.where(np.isnan(fb), 0, np.asarray(b))), f"seed {seed}: {what} differs"

    with make_caster(cfg, M) as c:
        fr = scene.setup_raycaster(c, cfg, vols, eye=eye)
        c.SetRenderTargets(depth, shadow)
        for v in range(V):
            c.UploadLightMap(v, lms[v])
        c.m_frameIdx = 0
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.cullVolumes()
        c.rayMarchV()
        c.composite()
        assert np.array_equal(c.ReadBackVolumeInfo()[vis], info[vis])
        for v in cube:
            mip = int(info[v, 0])
            same(c.ReadBackCubeMap(int(v), mip), oracle.cube_face_view(cm[v], N, mip), f"cube map {v}")
            gd, ed = c.ReadBackCubeDepth(int(v), mip), oracle.cube_face_view(cd[v], N, mip)
            assert np.array_equal(np.isnan(gd), np.isnan(ed)) and np.array_equal(np.where(np.isnan(gd), 0, gd), np.where(np.isnan(ed), 0, ed)), f"seed {seed}: cube depth {v}"
        same(c.ReadBackComposite(), img, "the composite")
        g = c.GetStats()
        assert (g["view_samples"], g["lit_samples"], g["rays"], g["composite_samples"]) == (st["view_samples"], st["lit_samples"], st["rays"], comp_samples)
        if len(vis):
            c.SetSH(sh)
            c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            c.cullVolumes()
            c.rayMarchL()
            c.synchronize()
            assert np.array_equal(c.ReadBackLightMap(int(vid)), lms2[vid]), f"seed {seed}: light map {vid} with a non-finite probe"
            assert c.GetStats()["light_samples"] == nls
