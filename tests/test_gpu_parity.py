"""Parity of the gfx950 kernels (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): cull outputs bit-exact; marched buffers within 1e-3 (fp16 storage) of the oracle,
with the outlier count reported.  Kernel and oracle share the numeric contract (fp32, explicit fma only in
lerp / ray-point evaluation, -ffp-contract=off), so in practice almost every texel matches bit for bit.
"""
import os

import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, scene
from multivolumes_amd import raycaster as rc
from tests.conftest import half_to_float
from tests.helpers import default_descs, oracle_pipeline

pytestmark = pytest.mark.gpu

TOL = 1e-3  # fp16 tolerance of the north star


from multivolumes_amd import _lib


def make_caster(cfg, M, S=None, flags=0, shard=(0, 1)):
    return MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], S or cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=M, flags=flags, shard=shard)


def run_gpu(caster, cfg, volumes, frames=None, use_sh=False, eye=None):
    fr = scene.setup_raycaster(caster, cfg, volumes, use_sh=use_sh, eye=eye)
    frames = cfg["V"] if frames is None else frames
    for f in range(frames):
        caster.m_frameIdx = f
        caster.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        caster.cullVolumes()
        caster.rayMarchL()
    caster.rayMarchV()
    caster.synchronize()
    return fr


def compare_cube(o, caster, ref, cfg, tol=TOL):
    N = cfg["N"]
    worst = 0.0
    mism = 0
    total = 0
    for v in ref["cube"]:
        mip = int(ref["info"][v, 0])
        mask = int(ref["info"][v, 2])
        got = half_to_float(caster.ReadBackCubeMap(int(v), mip))
        exp = half_to_float(o.cube_face_view(ref["cubeMaps"][v], N, mip))
        gotd = caster.ReadBackCubeDepth(int(v), mip)
        expd = o.cube_face_view(ref["cubeDepths"][v], N, mip)
        for f in range(6):
            if not (mask >> f) & 1:
                # invisible faces are never written (creation state = 0)
                assert not got[f].any() and not gotd[f].any()
                continue
            # NaNs (0 x inf inside GetStep / the accumulation for degenerate inputs) must appear at the same texels; their sign
            # and payload are not arithmetic (x86 and gfx950 propagate different default NaNs)
            gn, en = np.isnan(got[f]), np.isnan(exp[f])
            assert np.array_equal(gn, en), f"NaN texels differ (volume {v}, face {f})"
            diff = np.abs(np.where(gn, 0.0, got[f]) - np.where(en, 0.0, exp[f]))
            worst = max(worst, float(diff.max()))
            mism += int(((got[f] != exp[f]) & ~gn).sum())
            total += diff.size
            assert np.array_equal(gotd[f], expd[f]), f"cube depth differs (volume {v}, face {f})"
    assert worst <= tol, f"max |gpu - oracle| = {worst}"
    return worst, mism, total


def compare_light(o, caster, ref, cfg, volumes_to_check, tol=TOL):
    """tol == 0.0: every R11G11B10F word identical.  Otherwise the stated tolerance of the packed format: differing voxels at
    most one 11-bit ulp apart, at most 0.1 % of a map."""
    lib = o.load()
    worst = 0.0
    for v in volumes_to_check:
        got = caster.ReadBackLightMap(int(v))
        exp = ref["lightMaps"][v]
        if np.array_equal(got, exp):
            continue
        assert tol != 0.0, f"light map of volume {v}: {int((got != exp).sum())} of {got.size} words differ (bit-exact comparison)"
        idx = np.nonzero(got != exp)
        g = np.zeros(3, np.float32)
        e = np.zeros(3, np.float32)
        for a, b in zip(got[idx][:2000], exp[idx][:2000]):
            lib.mvo_unpack_r11g11b10f(int(a), rc._ptr(g))
            lib.mvo_unpack_r11g11b10f(int(b), rc._ptr(e))
            rel = np.max(np.abs(g - e) / np.maximum(np.abs(e), 1.0))
            worst = max(worst, float(rel))
        # 6-bit mantissas: one unit in the last place is 1/64 relative
        assert worst <= 1.0 / 32, f"light map differs by more than one 11-bit ulp: {worst}"
        assert len(idx[0]) <= 1e-3 * got.size, f"{len(idx[0])} light-map voxels differ"
    return worst


@pytest.mark.parametrize("name", ["C1", "T1"])
def test_pipeline_parity(oracle, name):
    cfg = scene.CONFIGS[name]
    M = 256
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, M)
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols)
        info = c.ReadBackVolumeInfo()
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(info, ref["info"]), "VolumeInfo must be bit-exact"
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        compare_light(oracle, c, ref, cfg, ref["vis"])
        worst, mism, total = compare_cube(oracle, c, ref, cfg)
        st = c.GetStats()
        # sample counters: identical control flow => identical counts (allow the rare threshold flip)
        assert abs(st["rays"] - ref["stats"]["rays"]) == 0
        assert abs(st["view_samples"] - ref["stats"]["view_samples"]) <= 1e-4 * ref["stats"]["view_samples"] + 2
        assert abs(st["lit_samples"] - ref["stats"]["lit_samples"]) <= 1e-4 * ref["stats"]["lit_samples"] + 2
        assert abs(st["light_samples"] - ref["stats"]["light_samples"]) <= 1e-4 * ref["stats"]["light_samples"] + 2
        print(f"{name}: max abs err {worst:.3g}, {mism}/{total} halfs differ")


def test_pipeline_parity_16_volumes_with_sh(oracle):
    """16 volumes (4 culled by the default camera), mesh depth + shadow, SH/AO path on."""
    cfg = scene.CONFIGS["T2"]
    M = 512
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, M, sh=scene.DEFAULT_SH)
    assert 0 < len(ref["vis"]) < cfg["V"]
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols, use_sh=True)
        assert np.array_equal(c.ReadBackVolumeInfo(), ref["info"])
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        # frames 0..V-1 relight visible[f % count]: every visible volume got a light map
        compare_light(oracle, c, ref, cfg, ref["vis"])
        compare_cube(oracle, c, ref, cfg)


def test_cull_bit_exact_many_cameras(oracle):
    """VolumeInfo / lists bit-exact over a sweep of eye points, including eyes inside the grid and odd viewports."""
    cfg = dict(scene.CONFIGS["T2"], W=640, H=360)
    V, N = cfg["V"], cfg["N"]
    M = 64
    rng = np.random.default_rng(3)
    eyes = [(4, 16, -80), (0, 0, -200), (30, 5, 10), (0, 60, 0.1), (-70, 20, 40), (5, 5, 5)] + [tuple(rng.uniform(-120, 120, 3)) for _ in range(24)]
    from tests.helpers import host_frame
    with make_caster(cfg, M, S=1) as c:
        c.SetMaxSamples(cfg["maxRay"], cfg["maxLight"])
        c.SetVolumesWorld(scene.VOLUME_SIZE, (0, 0, 0))
        for eye in eyes:
            _, vp, svp, cb, po = host_frame(cfg, 0, eye)
            sentinel = np.zeros((V, 4), np.uint16)
            # start both from the same VolumeInfo state: culled volumes keep stale rows (reference behaviour)
            prev = c.ReadBackVolumeInfo()
            info, vis, cube = oracle.cull(V, po, default_descs(V, 1, N), cb, cfg["maxRay"], prev)
            c.UpdateFrameRaw(cb, po)
            c.cullVolumes()
            got = c.ReadBackVolumeInfo()
            gvis, gcube = c.ReadBackVisibleLists()
            assert np.array_equal(got[:, 2], info[:, 2]), f"MaskBits differ for eye {eye}"
            assert np.array_equal(got, info), f"VolumeInfo differs for eye {eye}"
            assert np.array_equal(gvis, vis) and np.array_equal(gcube, cube)
            del sentinel


def test_volume_producers(oracle):
    N = 20  # not a multiple of the 4^3 brick: exercises the padded border bricks
    cfg = dict(N=N, L=8, V=1, W=64, H=64)
    with MultiRayCaster(N, 8, 1, 3, (64, 64), shadowMapSize=16) as c:
        c.InitVolumeData(0)
        assert np.array_equal(c.ReadBackVolume(0), oracle.init_grid_data(N)), "CSInitGridData must be bit-exact"
        rng = np.random.default_rng(5)
        src = rng.random((7, 9, 11), np.float32)
        c.LoadVolumeDataR32F(1, src)
        assert np.array_equal(c.ReadBackVolume(1), oracle.r32f_to_rgba16f(src, N)), "CSR32FToRGBA16F must be bit-exact"
        raw = rng.integers(0, 0x7C00, (N, N, N, 4), dtype=np.uint16)
        c.LoadVolumeData(2, raw)
        assert np.array_equal(c.ReadBackVolume(2), raw), "upload -> brick -> readback round trip"
    del cfg


def test_edge_cases(oracle):
    """Nothing visible, empty volume, single-texel faces (mip 4 of N=16), ragged sizes."""
    from tests.helpers import host_frame
    # camera looking away: nothing visible -> no work, stale buffers untouched, marchL falls back to frameIdx % V
    cfg = dict(scene.CONFIGS["T1"])
    M = 32
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    with make_caster(cfg, M) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        eye = np.array([0, 0, 300], np.float32)
        view = rc.look_at_lh(eye, (0, 0, 600))
        vp = rc.multiply(view, rc.perspective_fov_lh(scene.FOV_ANGLE_Y, 1.0, 1.0, 1000.0))
        c.m_frameIdx = 5
        c.UpdateFrame(vp, fr["shadowVP"], eye)
        c.Render()
        c.synchronize()
        vis, cube = c.ReadBackVisibleLists()
        assert len(vis) == 0 and len(cube) == 0
        st = c.GetStats()
        assert st["view_samples"] == 0 and st["rays"] == 0
        for v in range(cfg["V"]):
            assert not c.ReadBackCubeMap(v, 0).any()
        # CSRayMarchL.hlsl:31: no visible volume -> volume frameIdx % V is relit
        descs = default_descs(cfg["V"], cfg["V"], cfg["N"])
        vid, lms, _ = oracle.march_light(cfg["V"], cfg["N"], cfg["L"], M, vols, descs, c.perObject, c.cbPerFrame, fr["shadow"], None, [], cfg["maxLight"])
        assert vid == 5 % cfg["V"]
        assert np.array_equal(c.ReadBackLightMap(vid), lms[vid])
    # empty volume (alpha = 0): exact zeros, but rays are counted and cube depth is written
    cfg = dict(scene.CONFIGS["C1"], N=16, L=8, maxRay=16, maxLight=8)
    empty = np.zeros((16, 16, 16, 4), np.float16)
    empty[..., :3] = 1.0
    ref = oracle_pipeline(oracle, cfg, [empty.view(np.uint16)], M)
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, [empty.view(np.uint16)])
        compare_cube(oracle, c, ref, cfg, tol=0.0)
        st = c.GetStats()
        assert st["lit_samples"] == 0 and st["rays"] == ref["stats"]["rays"] and st["view_samples"] == ref["stats"]["view_samples"]
    # far camera: coarse mip, faces smaller than an 8x8 tile
    cfg = dict(scene.CONFIGS["C1"], N=16, L=8, maxRay=16, maxLight=8)
    vols = [rc.synth_volume(16, 1)]
    eye = (40.0, 160.0, -800.0)
    ref = oracle_pipeline(oracle, cfg, vols, M, eye=eye)
    assert ref["info"][0, 0] >= 2
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols, eye=eye)
        assert np.array_equal(c.ReadBackVolumeInfo(), ref["info"])
        compare_cube(oracle, c, ref, cfg)
    # ragged: N and L not multiples of the brick / block sizes
    cfg = dict(scene.CONFIGS["C1"], N=22, L=10, maxRay=24, maxLight=12, W=300, H=200)
    vols = [rc.synth_volume(22, 2)]
    ref = oracle_pipeline(oracle, cfg, vols, M)
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols)
        assert np.array_equal(c.ReadBackVolumeInfo(), ref["info"])
        compare_light(oracle, c, ref, cfg, ref["vis"])
        compare_cube(oracle, c, ref, cfg)


def test_shared_volume_sources(oracle):
    """VolumeDesc.VolTexId = i % numVolumeSrcs (MultiRayCaster.cpp:471-479): 4 volumes over 2 sources."""
    cfg = scene.CONFIGS["T1"]
    M = 64
    vols = [rc.synth_volume(cfg["N"], k) for k in range(2)]
    ref = oracle_pipeline(oracle, cfg, vols, M)
    assert list(ref["info"][:, 3]) == [0, 1, 0, 1]
    with make_caster(cfg, M, S=2) as c:
        run_gpu(c, cfg, vols)
        assert np.array_equal(c.ReadBackVolumeInfo(), ref["info"])
        compare_cube(oracle, c, ref, cfg)


def test_error_behaviour():
    """Creation failures return codes (the reference returns false, XUSG.h:12-15); stages need their inputs."""
    from multivolumes_amd import MVError, _lib
    with pytest.raises(MVError) as e:
        MultiRayCaster(1, 8, 1, 1, (8, 8))
    assert e.value.code == _lib.MV_ERR_INVALID_ARG
    with MultiRayCaster(8, 8, 1, 1, (8, 8), shadowMapSize=8) as c:
        with pytest.raises(MVError) as e:
            c.cullVolumes()
        assert e.value.code == _lib.MV_ERR_NOT_READY
        with pytest.raises(MVError) as e:
            c.InitVolumeData(3)
        assert e.value.code == _lib.MV_ERR_RANGE
        with pytest.raises(MVError) as e:
            c.SetVolumeDescs([5 | (5 << 14) | (8 << 18)])
        assert e.value.code == _lib.MV_ERR_RANGE
        # a frame's marches need THAT frame's cull (its slot holds the outputs of four frames ago until then)
        cfg = dict(scene.CONFIGS["C1"], N=8, L=8, W=8, H=8)
        fr = scene.setup_raycaster(c, cfg, [rc.synth_volume(8, 0)])
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render()
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        for stage in (c.rayMarchL, c.rayMarchV, c.composite, c.ReadBackCull):
            with pytest.raises(MVError) as e:
                stage()
            assert e.value.code == _lib.MV_ERR_NOT_READY
        c.cullVolumes()
        c.rayMarchL(); c.rayMarchV(); c.composite(); c.synchronize()
        with pytest.raises(MVError) as e:
            MultiRayCaster(8, 700, 1, 1, (8, 8))  # light grid beyond the 32-bit offsets of the quad-replicated layout
        assert e.value.code == _lib.MV_ERR_INVALID_ARG


@pytest.mark.parametrize("flags", [0, _lib.MV_FLAG_RAY_QUEUE], ids=["tiles", "ray_queue"])
def test_config_c2_matches_oracle_at_full_size(oracle, flags):
    """BASELINE.json configs[1] -- 4 x 128^3, 512^2, 64 steps/ray, no mesh -- the whole frame against the oracle, bit for bit, like the
    headline config below: VolumeInfo, both lists, every relit light map, every cube map and cube depth, the counters, and the
    composite of the WHOLE screen; both forms of the view march.  Then the size-independent properties: a second view march of the same
    frame leaves identical buffers and counts the same samples, alpha stays in [0, 1], every texel of a marched face hits the box."""
    cfg = scene.CONFIGS["C2"]
    V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
    M = 1024
    vols = [rc.synth_volume(N, k) for k in range(V)]
    ref = oracle_pipeline(oracle, cfg, vols, M)
    with make_caster(cfg, M, flags=flags) as c:
        run_gpu(c, cfg, vols)
        info = c.ReadBackVolumeInfo()
        assert np.array_equal(info, ref["info"])
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"]) and len(cube) >= 1
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        worst, mism, total = compare_cube(oracle, c, ref, cfg, tol=0.0)
        assert mism == 0 and total > 40_000
        st1 = c.GetStats()
        for k in ("rays", "view_samples", "lit_samples", "light_samples"):
            assert st1[k] == ref["stats"][k], k
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        assert np.array_equal(c.ReadBackComposite(), exp), "composite differs from the oracle"
        assert c.GetStats()["composite_samples"] == exp_samples
        # idempotence and bounds
        first = {int(v): (c.ReadBackCubeMap(int(v), int(info[v, 0])), c.ReadBackCubeDepth(int(v), int(info[v, 0]))) for v in cube}
        c.ResetStats()
        c.rayMarchV()
        c.synchronize()
        st2 = c.GetStats()
        expect_rays = 0
        for v in cube:
            cm, cd = first[int(v)]
            assert np.array_equal(c.ReadBackCubeMap(int(v), int(info[v, 0])), cm) and np.array_equal(c.ReadBackCubeDepth(int(v), int(info[v, 0])), cd), "marchV must be idempotent"
            a = half_to_float(cm)[..., 3]
            assert np.all((a >= 0) & (a <= 1.0)) and np.all(np.isfinite(half_to_float(cm)))
            d = max(1, N >> int(info[v, 0]))
            expect_rays += bin(int(info[v, 2]) & 0x3F).count("1") * d * d
        assert st2["rays"] == expect_rays  # every texel of a visible face of a volume the eye is outside of hits the box
        assert st2["view_samples"] == st1["view_samples"] and st2["lit_samples"] == st1["lit_samples"]
        assert st2["view_samples"] <= expect_rays * cfg["maxRay"]


def oracle_composite(o, cfg, vols, ref, lightMaps=None, cubeMaps=None, cubeDepths=None, y0=0, y1=None):
    return o.composite(cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"], vols, ref["lightMaps"] if lightMaps is None else lightMaps,
                       ref["depth"], ref["po"], ref["cb"], ref["info"], ref["vis"],
                       ref["cubeMaps"] if cubeMaps is None else cubeMaps, ref["cubeDepths"] if cubeDepths is None else cubeDepths, y0, y1)


def oracle_composite_over(o, cfg, vols, ref, background):
    return o.composite(cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"], ref["info"], ref["vis"],
                       ref["cubeMaps"], ref["cubeDepths"], background=background)


@pytest.mark.parametrize("name,W,H", [("T1", 256, 256), ("T2", 320, 200), ("T1", 48, 48)])
def test_composite_parity(oracle, name, W, H):
    """K-buffer OIT composite (a7): CubeCast layers for cube-map volumes, screen-space RayCast for the others
    (the 48x48 viewport makes cubeMapPix > projCov for the far volumes, which then take the RayCast path)."""
    cfg = dict(scene.CONFIGS[name], W=W, H=H)
    M = 128
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, M)
    exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
    if W == 48:
        assert not (ref["info"][ref["vis"], 2] & 0x8000).all(), "expected at least one volume on the screen-space scheme"
        assert exp_samples > 0
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols)
        c.composite()  # (the RayCast layers were marched by rayMarchV's screen tiles; composite_samples counts both places)
        got = c.ReadBackComposite()
        g, e = half_to_float(got), half_to_float(exp)
        diff = np.abs(g - e)
        assert (e[..., 3] > 0).sum() > 0.1 * W * H, "scene should cover a good part of the screen"
        assert diff.max() <= TOL, f"max |gpu - oracle| = {diff.max()} at {np.unravel_index(diff.argmax(), diff.shape)}"
        assert c.GetStats()["composite_samples"] == exp_samples
        # drawn over a colour target (the reference's resolve blends over the scene, MultiRayCaster.cpp:931)
        import torch
        rng = np.random.default_rng(7)
        bg = rng.random((H, W, 4), np.float32).astype(np.float16)
        d_bg = torch.from_numpy(bg.view(np.uint16).astype(np.int16)).cuda()
        torch.cuda.synchronize()
        c.composite(backgroundDevice=d_bg.data_ptr())
        over = c.ReadBackComposite()
        exp_over, _ = oracle_composite_over(oracle, cfg, vols, ref, bg.view(np.uint16))
        assert np.array_equal(over, exp_over), f"{(over != exp_over).sum()} halfs differ when drawn over a background"
        assert (half_to_float(over)[..., 3] >= half_to_float(got)[..., 3]).all()
        c.composite()
        # row-range form (what a rank composites of its screen tile): only rows [r0, r1) are written
        c2 = c.ReadBackComposite()
        r0, r1 = H // 3, (2 * H) // 3 + 1
        c.composite(r0, r1)
        c3 = c.ReadBackComposite()
        assert np.array_equal(c2, c3), "re-compositing a row band must reproduce the same pixels"
        print(f"composite {name} {W}x{H}: max abs err {diff.max():.3g}, {(got != exp).sum()} halfs differ")


def test_sharded_contexts_reproduce_unsharded_frame(oracle):
    """Two sharded contexts on one GPU (shard 0/2 and 1/2) + the exchange kernels = the unsharded frame.
    The all-gather itself is emulated by concatenating the two packed shards in rank order (what
    all_gather_into_tensor produces); tests/test_distributed_cpu.py covers the collective over gloo."""
    import torch
    from multivolumes_amd import distributed as D
    cfg = dict(scene.CONFIGS["T2"], W=256, H=160)
    V, N, H = cfg["V"], cfg["N"], cfg["H"]
    M = 64
    world = 2
    vols = [rc.synth_volume(N, k) for k in range(V)]
    with make_caster(cfg, M) as full:
        fr = run_gpu(full, cfg, vols)
        full.composite()
        info = full.ReadBackVolumeInfo()
        vis, cube = full.ReadBackVisibleLists()
        ref_img = full.ReadBackComposite()
        ref_cm = {int(v): full.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube}
        ref_cd = {int(v): full.ReadBackCubeDepth(int(v), int(info[v, 0])) for v in cube}
        ref_lm = {int(v): full.ReadBackLightMap(int(v)) for v in vis}
        ref_stats = full.GetStats()
    shards = [make_caster(cfg, M, shard=(r, world)) for r in range(world)]
    try:
        packed = []
        tot_samples = tot_rays = tot_raycast = 0
        for r, c in enumerate(shards):
            scene.setup_raycaster(c, cfg, vols)
            owned_vis = [int(v) for v in vis if int(v) % world == r]
            for f in range(len(vis)):  # every context runs every frame; frame f relights visible[f % count] on the shard that owns it
                c.m_frameIdx = f
                c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
                c.cullVolumes()
                c.rayMarchL()
            # the light-map exchange: the view march also marches this rank's rows of EVERY screen-space volume (RayCast reads
            # that volume's light map), so the maps the other rank relit must be here before it
            for v in vis:
                if int(v) % world != r:
                    c.UploadLightMap(int(v), ref_lm[int(v)])
            assert c.ShardRows() == D.row_range(H, r, world)
            c.rayMarchV()
            assert np.array_equal(c.ReadBackVolumeInfo(), info), "cull is replicated"
            for v in owned_vis:
                assert np.array_equal(c.ReadBackLightMap(v), ref_lm[v])
            d, slot_bytes, shard_bytes = c.ExchangeLayout()
            assert (d, slot_bytes, shard_bytes) == D.exchange_layout(N, info, vis, V, world)
            buf = torch.zeros(shard_bytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # torch fills on ITS stream; the pack below runs on the context's
            c.ExchangePack(buf.data_ptr(), d)
            info2, vis2, cube2, min_mip = c.ReadBackCull()
            assert np.array_equal(info2, info) and np.array_equal(vis2, vis) and np.array_equal(cube2, cube) and max(1, N >> min_mip) == d
            c.synchronize()
            packed.append(buf)
            st = c.GetStats()
            tot_samples += st["view_samples"]
            tot_rays += st["rays"]
            tot_raycast += st["composite_samples"]
            # a rank marches only its own volumes
            for v in cube:
                got = c.ReadBackCubeMap(int(v), int(info[v, 0]))
                if int(v) % world == r:
                    assert np.array_equal(got, ref_cm[int(v)])
                else:
                    assert not got.any()
        assert tot_samples == ref_stats["view_samples"] and tot_rays == ref_stats["rays"]
        assert tot_raycast == ref_stats["composite_samples"] > 0, "the RayCast pixels are split by rows, none marched twice"
        gathered = torch.cat(packed)
        torch.cuda.synchronize()  # torch.cat ran on torch's stream; the unpack below runs on the context's
        for r, c in enumerate(shards):
            c.ExchangeUnpack(gathered.data_ptr(), d)
            for v in cube:
                assert np.array_equal(c.ReadBackCubeMap(int(v), int(info[v, 0])), ref_cm[int(v)])
                assert np.array_equal(c.ReadBackCubeDepth(int(v), int(info[v, 0])), ref_cd[int(v)])
            r0, r1 = D.row_range(H, r, world)
            c.composite(r0, r1)
            img = c.ReadBackComposite()
            assert np.array_equal(img[r0:r1], ref_img[r0:r1]), "row band must match the unsharded composite"
    finally:
        for c in shards:
            c.close()


def test_device_views_alias_context_buffers():
    """distributed._device_view wraps a context-owned buffer as a torch tensor without copying (RCCL plumbing)."""
    import torch
    from multivolumes_amd import distributed as D
    cfg = dict(scene.CONFIGS["T1"], W=64, H=48)
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    with make_caster(cfg, 32) as c:
        run_gpu(c, cfg, vols, frames=1)
        c.composite()
        c.synchronize()
        ptr, n = c.GetBuffer(5)
        view = D._device_view(ptr, n)
        torch.cuda.synchronize()
        assert np.array_equal(view.cpu().numpy().view(np.uint16).reshape(48, 64, 4), c.ReadBackComposite())
        view.zero_()
        torch.cuda.synchronize()
        assert not c.ReadBackComposite().any()
        ptr, n = c.GetBuffer(1, 0)
        # device layout of a light map: quad-replicated cells {t(x,y), t(x+1,y), t(x,y+1), t(x+1,y+1)}, +1 clamped
        L = cfg["L"]
        assert n == L ** 3 * 16
        q = D._device_view(ptr, n).cpu().numpy().view(np.uint32).reshape(L, L, L, 4)
        lm = c.ReadBackLightMap(0)
        x1 = np.minimum(np.arange(L) + 1, L - 1)
        assert np.array_equal(q[..., 0], lm) and np.array_equal(q[..., 1], lm[:, :, x1])
        assert np.array_equal(q[..., 2], lm[:, x1, :]) and np.array_equal(q[..., 3], lm[:, x1, :][:, :, x1])
        # upload path rebuilds the same cells
        c.UploadLightMap(1, lm)
        ptr1, _ = c.GetBuffer(1, 1)
        assert np.array_equal(D._device_view(ptr1, n).cpu().numpy().view(np.uint32).reshape(L, L, L, 4), q)


def test_pipelined_frames_match_sequential():
    """MV_FLAG_PIPELINE_FRAMES (frame i+1's upload / cull / light march on a second stream while frame i's view march and
    composite run) must not change a single bit: same Render() sequence on a pipelined and on a plain context."""
    from multivolumes_amd import _lib
    cfg = dict(scene.CONFIGS["T2"], W=256, H=192)
    V = cfg["V"]
    vols = [rc.synth_volume(cfg["N"], k) for k in range(V)]
    outs = []
    for flags in (0, _lib.MV_FLAG_PIPELINE_FRAMES, _lib.MV_FLAG_PIPELINE_FRAMES | _lib.MV_FLAG_SINGLE_VIEW_STREAM):
        with make_caster(cfg, 64, flags=flags) as c:
            fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
            eye = np.array(fr["eye"], np.float32)
            frames = []
            for f in range(V + 5):
                # move the camera a little every frame so that cull results, light maps and cube maps all change
                e = eye + np.array([0.7 * f, -0.3 * f, 1.1 * f], np.float32)
                view = rc.look_at_lh(e, (0, 0, 0))
                vp = rc.multiply(view, rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0))
                c.SetRenderTargets(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), scene.SPHERE_RADIUS))
                c.UpdateFrame(vp, fr["shadowVP"], e)
                if f % 3 == 1:
                    # the multi-GPU call order: cull, cull readback (waits for the cull only), light march, light-map pointer
                    # (no host wait: the context's stream waits for the light march), view march, composite
                    c.cullVolumes()
                    info_f, vis_f, _, _ = c.ReadBackCull()
                    c.rayMarchL()
                    if len(vis_f):
                        c.GetBuffer(1, int(vis_f[0]))
                    c.rayMarchV()
                    c.composite()
                    c.m_frameIdx += 1
                else:
                    c.Render(composite=True)
                if f % 4 == 3 or f == V + 4:
                    frames.append((c.ReadBackComposite(), c.ReadBackVolumeInfo()))
            info = c.ReadBackVolumeInfo()
            vis, cube = c.ReadBackVisibleLists()
            lms = [c.ReadBackLightMap(v) for v in range(V)]
            cms = [c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube]
            st = c.GetStats()
            outs.append((frames, info, vis, cube, lms, cms, {k: st[k] for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples")}))
    a = outs[0]
    for b in outs[1:]:
        assert a[6] == b[6]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        for x, y in zip(a[0], b[0]):
            assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
        for x, y in zip(a[4], b[4]):
            assert np.array_equal(x, y)
        for x, y in zip(a[5], b[5]):
            assert np.array_equal(x, y)
    assert sum(int(l.any()) for l in a[4]) >= len(a[2]), "every visible volume has been relit at least once"


@pytest.fixture(scope="module")
def c3_reference(oracle):
    cfg = scene.CONFIGS["C3"]
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    return vols, oracle_pipeline(oracle, cfg, vols, 1024)


@pytest.mark.parametrize("flags", [0, _lib.MV_FLAG_RAY_QUEUE], ids=["tiles", "ray_queue"])
def test_headline_config_c3_matches_oracle_at_full_size(oracle, c3_reference, flags):
    """BASELINE.json configs[2] -- 16 x 256^3, 1024^2, 128 steps/ray, mesh depth -- cull, 16 light marches and the view march
    of the whole frame against the oracle, bit for bit (the sizes the metric is quoted on, not a reduced stand-in); then the
    K-buffer composite the bench's timed step ends with, a 64-row band across the middle of the screen against the oracle's
    (PSDepthPeel / PSCube.hlsl:30-60 / PSResolveOIT).  Both forms of the view march: one wavefront per tile, and the ray queue
    with ballot compaction (CSRayMarch.hlsl:77-158 either way)."""
    cfg = scene.CONFIGS["C3"]
    V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
    M = 1024
    vols, ref = c3_reference
    assert len(ref["vis"]) >= 10 and len(ref["cube"]) >= 8
    with make_caster(cfg, M, flags=flags) as c:
        run_gpu(c, cfg, vols)
        assert np.array_equal(c.ReadBackVolumeInfo(), ref["info"])
        vis, cube = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube, ref["cube"])
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        worst, mism, total = compare_cube(oracle, c, ref, cfg, tol=0.0)
        assert mism == 0 and total > 1_500_000
        st = c.GetStats()
        for k in ("rays", "view_samples", "lit_samples", "light_samples"):
            assert st[k] == ref["stats"][k], k
        assert st["view_samples"] > 15_000_000 and 0 < st["fetch_free_samples"] < st["view_samples"]
        # a7 at the headline size: rows H/2 - 32 .. H/2 + 32 (the near volumes on the screen-space scheme and the mesh are there)
        c.composite()
        img = c.ReadBackComposite()
        y0, y1 = H // 2 - 32, H // 2 + 32
        exp, _ = oracle.composite(V, N, L, W, H, vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"], ref["info"], ref["vis"],
                                  ref["cubeMaps"], ref["cubeDepths"], y0, y1)
        assert np.array_equal(exp[y0:y1], img[y0:y1]), "composite band differs from the oracle"
        assert (half_to_float(img[y0:y1])[..., 3] > 0).mean() > 0.2, "the band is covered"


@pytest.mark.parametrize("N,W,H,eye", [(12, 9, 2, (30.0, 40.0, -200.0)), (8, 16, 8, (4.0, 16.0, -400.0)), (12, 32, 24, (4.0, 16.0, -400.0)), (4, 16, 8, (4.0, 16.0, -400.0)), (12, 16, 8, (20.0, 20.0, -120.0))])
def test_one_texel_mips(oracle, N, W, H, eye):
    """Cube maps smaller than 16 texels: their last mips are ONE texel (max(1, N >> mip), as D3D lays a mip chain out and GetDimensions tells the reference's
    shaders), not empty.  Rounds 1-5 laid them out empty, and a composite that selected one divided by an edge of zero: a GPU memory fault at a 9 x 2 viewport
    (round 6, found by tests/test_random_scenes.py).  Cull, the march into 1 x 1 faces, CubeCast from them and the image against the oracle -- which
    tests/test_oracle_vs_reference_dxil.py::test_one_texel_mips_against_the_compiled_shaders holds against the reference's compiled shaders on these very scenes."""
    cfg = dict(scene.CONFIGS["T1"], N=N, L=6, V=4, W=W, H=H, maxRay=16, maxLight=8)
    M = 32
    vols = [rc.synth_volume(N, k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, M, eye=eye)
    assert len(ref["cube"]) and min(max(1, N >> int(ref["info"][v, 0])) for v in ref["cube"]) == 1
    with make_caster(cfg, M) as c:
        run_gpu(c, cfg, vols, eye=eye)
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], ref["info"][ref["vis"]])
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        worst, mism, total = compare_cube(oracle, c, ref, cfg, tol=0.0)
        assert mism == 0 and total > 0
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        assert np.array_equal(c.ReadBackComposite(), exp)
        st = c.GetStats()
        assert st["view_samples"] == ref["stats"]["view_samples"] and st["rays"] == ref["stats"]["rays"] > 0 and st["composite_samples"] == exp_samples


@pytest.mark.parametrize("name,over", [("T2", dict(W=256, H=160)), ("T1", dict(N=40, W=200, H=200))])
def test_stress_variant_against_the_oracle(oracle, name, over):
    """MV_FLAG_FORCE_MIP0_ALL_FACES (SURVEY 8d's stress variant, `bench.py --stress`): the cull forces mip 0, all six faces, the cube-map scheme and the
    whole sample cap for every visible volume.  Not a reference mode, so the oracle's cull is amended the same way here and its marches fed that VolumeInfo:
    cube maps of all six faces (rays through the back faces too), depths, light maps, the composite and the counters, bit for bit."""
    cfg = dict(scene.CONFIGS[name], **over)
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    M = 64
    vols = [rc.synth_volume(N, k) for k in range(V)]
    ref = oracle_pipeline(oracle, cfg, vols, M)   # cull + V light marches (the light marches do not read VolumeInfo)
    info = ref["info"].copy()
    for v in ref["vis"]:
        info[v] = (0, cfg["maxRay"], 0x3f | 0x8000, info[v, 3])
    cube = ref["vis"].copy()
    cm, cd, st = oracle.march_view(V, N, L, cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"], info, cube)
    ref2 = dict(ref, info=info, cube=cube, cubeMaps=cm, cubeDepths=cd)
    with make_caster(cfg, M, flags=_lib.MV_FLAG_FORCE_MIP0_ALL_FACES) as c:
        run_gpu(c, cfg, vols)
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], info[ref["vis"]])
        vis, cube_g = c.ReadBackVisibleLists()
        assert np.array_equal(vis, ref["vis"]) and np.array_equal(cube_g, cube)
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        worst, mism, total = compare_cube(oracle, c, ref2, cfg, tol=0.0)
        assert mism == 0 and total >= len(cube) * 6 * N * N * 0.2
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref2)
        c.composite()
        assert np.array_equal(c.ReadBackComposite(), exp)
        g = c.GetStats()
        assert g["view_samples"] == st["view_samples"] > 0 and g["lit_samples"] == st["lit_samples"] and g["rays"] == st["rays"] and g["composite_samples"] == exp_samples


@pytest.mark.parametrize("L", [255, 320])
def test_large_light_grids(oracle, L):
    """Light grids far beyond the benchmark's 96 (an odd one and 320: 33 M voxels, 0.5 GB of cells per buffer): the voxel-block dispatch order, the
    quad-replicated layout's 64-bit offsets and the light fetch's clamps at scale, every light map and the frame that reads them against the oracle."""
    cfg = dict(scene.CONFIGS["T1"], N=24, L=L, V=2, W=128, H=96, maxRay=32, maxLight=12)
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, 64)
    with make_caster(cfg, 64) as c:
        run_gpu(c, cfg, vols)
        assert np.array_equal(c.ReadBackVolumeInfo()[ref["vis"]], ref["info"][ref["vis"]])
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        if len(ref["cube"]):
            assert compare_cube(oracle, c, ref, cfg, tol=0.0)[1] == 0
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        assert np.array_equal(c.ReadBackComposite(), exp)
        assert c.GetStats()["light_samples"] == ref["stats"]["light_samples"] > 0


@pytest.mark.parametrize("N", [512, 640])
def test_large_grid_offsets(oracle, N):
    """N = 512 (1 GiB of texels per volume) and N = 640 (2 GiB: the largest grid mv_create accepts): the 32-bit byte offsets and 24-bit
    stride multiplies of the bricked layout near and at the top of their range, against the oracle."""
    cfg = dict(scene.CONFIGS["C1"], N=N, L=16, W=640 if N == 512 else 400, H=640 if N == 512 else 400, maxRay=40, maxLight=24)
    vols = [rc.synth_volume(cfg["N"], 1)]
    ref = oracle_pipeline(oracle, cfg, vols, 64)
    with make_caster(cfg, 64) as c:
        run_gpu(c, cfg, vols)
        assert np.array_equal(c.ReadBackVolumeInfo(), ref["info"])
        compare_light(oracle, c, ref, cfg, ref["vis"], tol=0.0)
        worst, mism, total = compare_cube(oracle, c, ref, cfg, tol=0.0)
        assert mism == 0 and total > 0
        exp, exp_samples = oracle_composite(oracle, cfg, vols, ref)
        c.composite()
        assert np.array_equal(c.ReadBackComposite(), exp)
        st = c.GetStats()
        assert st["view_samples"] == ref["stats"]["view_samples"] > 0 and st["composite_samples"] == exp_samples


@pytest.mark.gpu
def test_light_march_shadow_in_full_precision_on_the_gpu(oracle):
    """The light map keeps six mantissa bits; this reads the kernel's fp32 `shadow` out through the frame constants (tests/helpers.py
    shadow_in_full_precision: light colour (1, 0, 0), ambient (-c, 0, 0), bisect c) and compares it with the oracle's, voxel by voxel, to the last bit --
    ShadowTest in the compiled order (round 6) and the light ray through every volume behind it."""
    from tests.helpers import default_descs, full_precision_light_scene, host_frame, shadow_in_full_precision
    voxels = fractional = 0
    for seed in range(int(os.environ.get("MV_GPU_PEEL_SEEDS", "12"))):
        cfg, vols, worlds, light, frame, shadow, M = full_precision_light_scene(seed, L=4)
        V, N, L = cfg["V"], cfg["N"], cfg["L"]
        descs = default_descs(V, V, N)
        _, vp, svp, cb, po = host_frame(cfg, frame, (30.0, 20.0, -60.0), worlds, light)
        with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=M) as c:
            for i, v in enumerate(vols):
                c.LoadVolumeData(i, v)
            c.SetMaxSamples(cfg["maxRay"], cfg["maxLight"])
            c.SetRenderTargets(np.ones((cfg["H"], cfg["W"]), np.float32), shadow)
            c.UpdateFrameRaw(cb, po)
            c.cullVolumes()
            vis, _ = c.ReadBackVisibleLists()
            vis = np.sort(np.asarray(vis, np.uint32))
            vid = oracle.march_light(V, N, L, M, vols, descs, po, cb, shadow, None, vis, cfg["maxLight"])[0]

            def on_gpu(cbb):
                c.UpdateFrameRaw(cbb, po)
                c.cullVolumes()
                c.rayMarchL()
                return c.ReadBackLightMap(int(vid))
            a = shadow_in_full_precision(lambda cbb: oracle.march_light(V, N, L, M, vols, descs, po, cbb, shadow, None, vis, cfg["maxLight"])[1][vid], cb, L)
            b = shadow_in_full_precision(on_gpu, cb, L)
        assert np.array_equal(a, b), (seed, [hex(int(x)) for x in a[a != b]][:8], [hex(int(x)) for x in b[a != b]][:8])
        voxels += len(a)
        fractional += int(((a != 0) & (a != int(np.float32(1.0).view(np.uint32)))).sum())
    assert fractional > voxels // 4, (fractional, voxels)


@pytest.mark.gpu
def test_cull_agrees_with_the_oracle_at_its_decision_boundaries(oracle):
    """The cull kernel at the adjacent input floats between which the oracle's cull changes a decision, and at their outer neighbours
    (tests/helpers.py cull_decision_boundaries): VolumeInfo of the visible volumes and both lists equal -- the kernel's float intermediates are the
    oracle's (and, through tests/test_oracle_vs_reference_dxil.py, the compiled shader's) to the last bit where it matters."""
    from tests.helpers import cull_decision_boundaries
    found = 0
    casters = {}
    try:
        for seed in range(int(os.environ.get("MV_GPU_CULL_BOUNDARY_SEEDS", "150"))):
            for cfg, descs, cb, po in cull_decision_boundaries(oracle, seed):
                V, N = cfg["V"], cfg["N"]
                k = (N, V, cfg["W"], cfg["H"], cfg["maxRay"])
                if k not in casters:
                    if len(casters) >= 6:  # (a context of eight 256^3 volumes holds 2 GiB: a long sweep keeps a handful, not one per config)
                        casters.pop(next(iter(casters))).close()
                    casters[k] = MultiRayCaster(N, 8, V, V, (cfg["W"], cfg["H"]), shadowMapSize=32)
                    casters[k].SetMaxSamples(cfg["maxRay"], 8)
                c = casters[k]
                c.UpdateFrameRaw(cb, po)
                c.cullVolumes()
                info, vis, cube = oracle.cull(V, po, descs, cb, cfg["maxRay"])
                gvis, gcube = c.ReadBackVisibleLists()
                assert np.array_equal(np.sort(gvis), np.sort(vis)) and np.array_equal(np.sort(gcube), np.sort(cube)), (seed, cfg)
                assert np.array_equal(c.ReadBackVolumeInfo()[np.sort(vis)], info[np.sort(vis)]), (seed, cfg, c.ReadBackVolumeInfo().tolist(), np.asarray(info).tolist())
                found += 1
    finally:
        for c in casters.values():
            c.close()
    assert found >= 100, found


@pytest.mark.gpu
def test_view_march_agrees_with_the_oracle_at_its_boundaries():
    """scripts/r06_gpu_view_walk.py: cull + view march on the GPU at the adjacent input floats between which any half or cube depth the ORACLE stores changes
    (tests/helpers.py output_boundaries) -- VolumeInfo, every marched face and depth equal there too."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r06_gpu_view_walk.py"), "40"], capture_output=True, text=True, cwd=root, timeout=300)
    assert r.returncode == 0 and "GPU differs from the oracle at 0," in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
