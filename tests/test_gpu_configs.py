"""BASELINE.json configs at their FULL sizes on the GPU box: C5 (64 x 256^3, 2048^2, 256 steps/ray -- Bin/all64.bat:1)
and the C4 shape (C3's 16 x 256^3 frame sharded over 4 contexts on one GPU, bit for bit against the unsharded frame),
plus the moving-depth-map schedule of frames in flight at C3 size.

The oracle cannot march a whole C5 frame in seconds, so C5 is checked through size-independent properties (idempotence,
counters against the cull, finite alpha in [0, 1]) and oracle checks of five relit light maps (64 volumes x 256^3), of four volumes'
view march fed the oracle's light maps, and of one composite row band.
"""
import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, _lib, scene
from multivolumes_amd import distributed as D
from multivolumes_amd import raycaster as rc
from tests.conftest import half_to_float

pytestmark = pytest.mark.gpu


def make_caster(cfg, M, flags=0, shard=(0, 1)):
    return MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=M, flags=flags, shard=shard)


def relight_all(c, fr, frames):
    for f in range(frames):
        c.m_frameIdx = f
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.cullVolumes()
        c.rayMarchL()


@pytest.mark.parametrize("flags", [0, _lib.MV_FLAG_RAY_QUEUE], ids=["tiles", "ray_queue_wide"])
def test_c5_full_size_properties(oracle, flags):
    """BASELINE.json configs[4]: 64 volumes of 256^3, 2048^2 viewport, 256 steps/ray, early termination + OIT composite.
    With MV_FLAG_RAY_QUEUE the march is the ray-queue form with 64-bit pool pointers (`WIDE`: 64 x 130 MiB of texels exceed
    the 4 GiB a 32-bit offset reaches) -- the same oracle checks."""
    cfg = scene.CONFIGS["C5"]
    V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
    M = 1024
    with make_caster(cfg, M, flags=flags) as c:
        for k in range(V):
            c.LoadVolumeData(k, rc.synth_volume(N, k))
        fr = scene.setup_raycaster(c, cfg)
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.cullVolumes()
        vis, cube = c.ReadBackVisibleLists()
        assert len(vis) >= 32 and len(cube) >= 16, "the default C5 camera sees most of the 8 x 8 grid"
        relight_all(c, fr, len(vis))  # every visible volume relit once
        c.ResetStats()
        c.rayMarchV()
        c.composite()
        c.synchronize()
        info = c.ReadBackVolumeInfo()
        st1 = c.GetStats()
        img1 = c.ReadBackComposite()
        first = {int(v): (c.ReadBackCubeMap(int(v), int(info[v, 0])), c.ReadBackCubeDepth(int(v), int(info[v, 0]))) for v in cube}

        # idempotence: the same frame again writes the same bits and counts the same samples
        c.ResetStats()
        c.rayMarchV()
        c.composite()
        c.synchronize()
        st2 = c.GetStats()
        assert np.array_equal(c.ReadBackComposite(), img1)
        expect_rays = 0
        for v in cube:
            cm, cd = first[int(v)]
            mip = int(info[v, 0])
            assert np.array_equal(c.ReadBackCubeMap(int(v), mip), cm) and np.array_equal(c.ReadBackCubeDepth(int(v), mip), cd)
            f = half_to_float(cm)
            assert np.all(np.isfinite(f)) and np.all((f[..., 3] >= 0) & (f[..., 3] <= 1.0))
            d = max(1, N >> mip)
            expect_rays += bin(int(info[v, 2]) & 0x3F).count("1") * d * d
        for k in ("view_samples", "lit_samples", "rays", "composite_samples"):
            assert st1[k] == st2[k], k
        # the eye is outside every volume: every texel of a visible face hits its box -- up to the odd ray that grazes a cube edge
        # and fails ComputeRayOrigin's |p| <= 1 tests by an ulp (RayMarch.hlsli:141-146; such a texel is simply not written)
        assert expect_rays - 16 <= st1["rays"] <= expect_rays
        assert 0 < st1["lit_samples"] <= st1["view_samples"] <= expect_rays * cfg["maxRay"]
        assert all(int(info[v, 1]) <= cfg["maxRay"] for v in vis)
        a = half_to_float(img1)
        assert np.all(np.isfinite(a)) and np.all((a[..., 3] >= 0) & (a[..., 3] <= 1.0)) and (a[..., 3] > 0).mean() > 0.2

        # Oracle checks at the full size, bit for bit.
        # (1) the light march (CSRayMarchL.hlsl:77-110: the loop over all 64 volumes, pool addresses beyond 4 GiB): the ORACLE relights five
        # visible volumes -- frame f relights vis[f % count], and a relit map depends on the volumes' alpha only, not on other light
        # maps -- among them the four the view march is checked on below; each equals the GPU's map of that volume.
        vols = [rc.synth_volume(N, k) for k in range(V)]
        from tests.helpers import default_descs, host_frame
        descs = default_descs(V, V, N)
        _, vp0, svp0, _, _ = host_frame(cfg, 0)
        _, shadow = scene.depth_and_shadow(cfg, vp0, svp0, M)
        by_cost = sorted((int(u) for u in cube), key=lambda u: (int(info[u, 0]), -bin(int(info[u, 2]) & 0x3F).count("1")))
        picked = [by_cost[0], by_cost[len(by_cost) // 3], by_cost[2 * len(by_cost) // 3], by_cost[-1]]  # near and far, mip 0 and beyond
        extra = next(int(u) for u in vis if int(u) not in picked)
        vis_list = [int(u) for u in vis]
        lms_oracle = np.zeros((V, L, L, L), np.uint32)
        for u in picked + [extra]:
            f = vis_list.index(u)
            _, _, _, cb_f, po_f = host_frame(cfg, f)
            vid, lm_f, n_f = oracle.march_light(V, N, L, M, vols, descs, po_f, cb_f, shadow, None, vis, cfg["maxLight"])
            assert vid == u and n_f > 0
            assert np.array_equal(lm_f[u], c.ReadBackLightMap(u)), f"light map of volume {u} (relit in frame {f}) differs from the oracle's"
            lms_oracle[u] = lm_f[u]
        # (2) the view march of those four cube-map volumes, fed the ORACLE's light maps (not the GPU's own)
        cm, cd, _ = oracle.march_view(V, N, L, W, H, vols, lms_oracle, fr["depth"], c.perObject, c.cbPerFrame, info, picked)
        for v in picked:
            mip = int(info[v, 0])
            assert np.array_equal(oracle.cube_face_view(cm[v], N, mip), first[v][0]), v
            assert np.array_equal(oracle.cube_face_view(cd[v], N, mip), first[v][1]), v
        # (3) a band of the composite (the oracle composites from the GPU's cube maps and light maps: compositeKernel at 64 volumes, 2048^2)
        lms = np.zeros((V, L, L, L), np.uint32)
        for u in vis:
            lms[u] = c.ReadBackLightMap(int(u))
        chain = oracle.cube_chain_texels(N)
        cms = np.zeros((V, chain, 4), np.uint16)
        cds = np.zeros((V, chain), np.float32)
        for u in cube:
            m = int(info[u, 0])
            off, n = oracle.cube_mip_offset(N, m), 6 * max(1, N >> m) ** 2
            cms[u, off:off + n] = first[int(u)][0].reshape(n, 4)
            cds[u, off:off + n] = first[int(u)][1].reshape(n)
        y0, y1 = H // 2 - 16, H // 2 + 16
        exp, _ = oracle.composite(V, N, L, W, H, vols, lms, fr["depth"], c.perObject, c.cbPerFrame, info, vis, cms, cds, y0, y1)
        assert np.array_equal(exp[y0:y1], img1[y0:y1]), "composite band differs from the oracle"
        print(f"C5: {len(vis)} visible / {len(cube)} cube-map volumes, {st1['view_samples']} view samples, {st1['composite_samples']} RayCast samples")


def check_sharded(cfg, M, world):
    """`world` sharded contexts on one GPU + the exchange kernels reproduce the unsharded frame bit for bit (cube maps,
    cube depths, light maps of the owners, row-split RayCast, composite bands).  The all-gather is the concatenation of
    the packed shards in rank order (what ncclAllGather produces)."""
    import torch
    V, N, H = cfg["V"], cfg["N"], cfg["H"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    with make_caster(cfg, M) as full:
        fr = scene.setup_raycaster(full, cfg, vols)
        relight_all(full, fr, V)
        full.rayMarchV()
        full.composite()
        info = full.ReadBackVolumeInfo()
        vis, cube = full.ReadBackVisibleLists()
        ref_img = full.ReadBackComposite()
        ref_cm = {int(v): full.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube}
        ref_cd = {int(v): full.ReadBackCubeDepth(int(v), int(info[v, 0])) for v in cube}
        ref_lm = {int(v): full.ReadBackLightMap(int(v)) for v in vis}
        ref_stats = full.GetStats()
    shards = []
    try:
        packed = []
        tot = dict(view_samples=0, rays=0, composite_samples=0, light_samples=0)
        for r in range(world):
            c = make_caster(cfg, M, shard=(r, world))
            shards.append(c)
            scene.setup_raycaster(c, cfg, vols)
            relight_all(c, fr, V)  # frame f relights visible[f % count] on the shard that owns it
            for v in vis:
                if int(v) % world != r:
                    c.UploadLightMap(int(v), ref_lm[int(v)])  # the light-map exchange (RayCast rows read every such map)
                else:
                    assert np.array_equal(c.ReadBackLightMap(int(v)), ref_lm[int(v)])
            assert c.ShardRows() == D.row_range(H, r, world)
            c.rayMarchV()  # (same frame as the last light march, like the unsharded run above)
            assert np.array_equal(c.ReadBackVolumeInfo(), info), "cull is replicated"
            d, slot_bytes, shard_bytes = c.ExchangeLayout()
            assert (d, slot_bytes, shard_bytes) == D.exchange_layout(N, info, vis, V, world)
            buf = torch.zeros(shard_bytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # torch fills on ITS stream; the pack below runs on the context's
            c.ExchangePack(buf.data_ptr(), d)
            c.synchronize()
            packed.append(buf)
            st = c.GetStats()
            for k in tot:
                tot[k] += st[k]
            for v in cube:
                got = c.ReadBackCubeMap(int(v), int(info[v, 0]))
                if int(v) % world == r:
                    assert np.array_equal(got, ref_cm[int(v)])
                else:
                    assert not got.any(), "a shard marches only its own volumes"
        assert tot["view_samples"] == ref_stats["view_samples"] and tot["rays"] == ref_stats["rays"]
        assert tot["composite_samples"] == ref_stats["composite_samples"], "RayCast pixels are split by rows, none marched twice"
        assert tot["light_samples"] == ref_stats["light_samples"], "every light march ran on exactly one shard"
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
            assert np.array_equal(img[r0:r1], ref_img[r0:r1]), f"row band of shard {r} differs from the unsharded composite"
        return dict(visible=len(vis), cube=len(cube), exchange_bytes=int(gathered.numel()), raycast=ref_stats["composite_samples"])
    finally:
        for c in shards:
            c.close()


def test_c4_shape_four_shards_of_c3_match_unsharded_frame():
    """BASELINE.json configs[3] on the hardware at hand: the full-size C3 frame (16 x 256^3, 1024^2, 128 steps) split over
    FOUR sharded contexts (4 volumes each) on one GPU -- the partitioning, pack / unpack and row-band composite of the 8-GPU
    run, every byte compared with the unsharded frame.  (8 physical GPUs are the driver's to launch.)"""
    r = check_sharded(scene.CONFIGS["C3"], 1024, 4)
    assert r["visible"] >= 10 and r["cube"] >= 8 and r["raycast"] > 0
    print("C4 shape:", r)


def test_three_shards_uneven_split():
    """V = 16 over 3 shards (6 / 5 / 5 volumes, uneven row bands): padding slots and ragged bands."""
    check_sharded(dict(scene.CONFIGS["T2"], W=200, H=136), 64, 3)


@pytest.mark.parametrize("route", ["host_copy", "device_pointer"])
def test_frames_in_flight_with_a_new_depth_map_every_frame_c3(route):
    """C3-sized frames with a DIFFERENT depth map every frame (a moving camera): frames in flight must give the one-stream
    result.  `host_copy` = mv_set_depth_map per frame (the setter waits for the frames that still read the old map);
    `device_pointer` = a borrowed device buffer per frame (mv_set_depth_map_device), which keeps frames in flight."""
    import torch
    cfg = scene.CONFIGS["C3"]
    V, N, W, H = cfg["V"], cfg["N"], cfg["W"], cfg["H"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    frames = 12
    cams = []
    eye0 = np.array(scene.camera(cfg)[0], np.float32)
    for f in range(frames):
        e = eye0 + np.array([1.5 * f, -0.4 * f, 2.0 * f], np.float32)
        vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), rc.perspective_fov_lh(scene.FOV_ANGLE_Y, W / H, 1.0, 1000.0))
        # the occluder moves too, so that consecutive depth maps differ over many pixels
        cams.append((e, vp, rc.synth_sphere_depth(W, H, vp, (0.6 * f, 0.0, 0.0), scene.SPHERE_RADIUS)))
    outs = []
    for flags in (0, _lib.MV_FLAG_PIPELINE_FRAMES):
        with make_caster(cfg, 1024, flags=flags) as c:
            fr = scene.setup_raycaster(c, cfg, vols)
            dev = [torch.from_numpy(d).cuda() for (_, _, d) in cams] if route == "device_pointer" else None
            torch.cuda.synchronize()
            imgs = []
            for f, (e, vp, depth) in enumerate(cams):
                if route == "host_copy":
                    c.SetRenderTargets(depth)
                else:
                    c.SetRenderTargets(depthMapDevice=dev[f].data_ptr())
                c.UpdateFrame(vp, fr["shadowVP"], e)
                c.Render(composite=True)
                if f % 4 == 3:
                    imgs.append(c.ReadBackComposite())
            info = c.ReadBackVolumeInfo()
            vis, cube = c.ReadBackVisibleLists()
            cms = [c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube]
            cds = [c.ReadBackCubeDepth(int(v), int(info[v, 0])) for v in cube]
            st = c.GetStats()
            outs.append((imgs, info, cms, cds, {k: st[k] for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples")}))
    a, b = outs
    assert a[4] == b[4]
    assert np.array_equal(a[1], b[1])
    for x, y in zip(a[0] + a[2] + a[3], b[0] + b[2] + b[3]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("flags", [0, _lib.MV_FLAG_PIPELINE_FRAMES])
def test_a_frame_without_light_march_keeps_the_light_maps(flags):
    """The reference's light maps simply keep their contents when rayMarchL is not recorded in a frame.  Here every volume has
    rotating light buffers and the cull pre-selects the relit volume's next one: a frame that leaves mv_march_light out must
    still read the maps that were last written."""
    cfg = dict(scene.CONFIGS["T2"], W=128, H=96)
    V = cfg["V"]
    vols = [rc.synth_volume(cfg["N"], k) for k in range(V)]
    with make_caster(cfg, 64, flags=flags) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        relight_all(c, fr, V)
        c.rayMarchV()
        c.composite()
        info = c.ReadBackVolumeInfo()
        vis, cube = c.ReadBackVisibleLists()
        ref = [c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube]
        ref_lm = [c.ReadBackLightMap(int(v)) for v in vis]
        ref_img = c.ReadBackComposite()
        for f in range(2 * len(vis) + 1):  # frameIdx sweeps every visible volume: each would have been "relit" once or twice
            c.m_frameIdx = f
            c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            c.cullVolumes()
            c.rayMarchV()
            c.composite()
        for v, a in zip(cube, ref):
            assert np.array_equal(c.ReadBackCubeMap(int(v), int(info[v, 0])), a)
        assert np.array_equal(c.ReadBackComposite(), ref_img)
        for v, a in zip(vis, ref_lm):
            assert np.array_equal(c.ReadBackLightMap(int(v)), a)
        from multivolumes_amd._lib import MVError
        with pytest.raises(MVError):
            c.rayMarchL()  # after the frame's view march: refused (the frame already reads the kept maps)


@pytest.mark.parametrize("flags", [0, _lib.MV_FLAG_PIPELINE_FRAMES, _lib.MV_FLAG_RAY_QUEUE])
def test_tile_count_jumps_between_frames(flags):
    """The view-march grid is sized from the tile count of an EARLIER frame (the host never waits for the current cull): a
    camera that jumps between far and near multiplies the tile count from one frame to the next, far beyond the margin, and
    the overflow launch must march the rest.  Every frame is compared with a fresh context (no earlier report: upper-bound
    grid) rendering the same camera."""
    cfg = dict(scene.CONFIGS["T1"], W=320, H=240, N=64)
    V = cfg["V"]
    vols = [rc.synth_volume(cfg["N"], k) for k in range(V)]
    eye0 = np.array(scene.camera(cfg)[0], np.float32)
    eyes = [eye0 * s for s in (3.0, 0.45, 3.0, 0.5, 2.5, 0.45)]

    def render(c, fr, e):
        vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0))
        c.UpdateFrame(vp, fr["shadowVP"], e)
        c.Render(composite=True)
        info = c.ReadBackVolumeInfo()
        vis, cube = c.ReadBackVisibleLists()
        return info, [c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube], c.ReadBackComposite(), c.GetStats()["rays"]

    with make_caster(cfg, 64, flags=flags) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        seq = []
        for e in eyes:
            c.ClearCubeMaps()
            c.ResetStats()
            c.m_frameIdx = 0
            seq.append(render(c, fr, e))
    rays = [s[3] for s in seq]
    assert max(rays) > 2.5 * min(rays), f"the tile count must jump: {rays}"
    for e, got in zip(eyes, seq):
        with make_caster(cfg, 64) as ref:
            fr = scene.setup_raycaster(ref, cfg, vols)
            ref.m_frameIdx = 0
            exp = render(ref, fr, e)
        assert np.array_equal(exp[0], got[0]) and exp[3] == got[3]
        for a, b in zip(exp[1], got[1]):
            assert np.array_equal(a, b)
        assert np.array_equal(exp[2], got[2])


def test_c3_with_the_reference_scenes_mesh(oracle):
    """BASELINE.json configs[2] with the mesh of the reference's own scene (`-mesh Assets/bunny.obj 0 -9 0 1.8`, Bin/all16.bat:1)
    as the source of the depth and shadow maps: rasterised on the GPU (mv_mesh_render_*) bit for bit like the oracle rasteriser,
    then the whole C3 frame -- cull, 16 light marches, view march -- against the oracle on those maps."""
    from tests.helpers import default_descs, host_frame
    cfg = scene.CONFIGS["C3"]
    V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
    M = 1024
    pos, idx = scene.load_bunny()
    assert len(idx) == 69666
    vols = [rc.synth_volume(N, k) for k in range(V)]
    with make_caster(cfg, M) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        scene.render_mesh_targets(c, cfg, fr["viewProj"], (pos, idx))
        depth, shadow = c.ReadBackDepthMap(), c.ReadBackShadowMap()
        wvp, swvp = scene.mesh_matrices(cfg, fr["viewProj"])
        exp_d = oracle.raster_depth(pos, idx, rc.transpose_store(wvp), W, H)
        exp_s = oracle.depth_to_d16(oracle.raster_depth(pos, idx, rc.transpose_store(swvp), M, M))
        assert (exp_d < 1).sum() > 20000 and (exp_s < 65535).sum() > 20000, "the bunny covers a good part of both maps"
        assert np.array_equal(depth, exp_d) and np.array_equal(shadow, exp_s)

        relight_all(c, fr, V)
        c.rayMarchV()
        c.synchronize()
        info = c.ReadBackVolumeInfo()
        vis, cube = c.ReadBackVisibleLists()

        # the oracle's frame on the rasterised maps
        descs = default_descs(V, V, N)
        _, _, _, cb, po = host_frame(cfg, 0)
        oinfo, ovis, ocube = oracle.cull(V, po, descs, cb, cfg["maxRay"])
        assert np.array_equal(info, oinfo) and np.array_equal(vis, ovis) and np.array_equal(cube, ocube)
        lms = None
        for f in range(V):
            _, _, _, cbf, pof = host_frame(cfg, f)
            _, lms, _ = oracle.march_light(V, N, L, M, vols, descs, pof, cbf, exp_s, None, ovis, cfg["maxLight"], lms)
        for v in vis:
            assert np.array_equal(c.ReadBackLightMap(int(v)), lms[v]), f"light map {v}"
        _, _, _, cbl, pol = host_frame(cfg, V - 1)
        cm, cd, st = oracle.march_view(V, N, L, W, H, vols, lms, exp_d, pol, cbl, oinfo, ocube)
        for v in cube:
            mip = int(info[v, 0])
            assert np.array_equal(c.ReadBackCubeMap(int(v), mip), oracle.cube_face_view(cm[v], N, mip)), f"cube map {v}"
            assert np.array_equal(c.ReadBackCubeDepth(int(v), mip), oracle.cube_face_view(cd[v], N, mip)), f"cube depth {v}"
        g = c.GetStats()
        assert g["view_samples"] == st["view_samples"] and g["lit_samples"] == st["lit_samples"] and g["rays"] == st["rays"]


@pytest.mark.parametrize("extra,in_flight,per_launch", [(0, 0, 0), (_lib.MV_FLAG_CULL_ON_VIEW_STREAM, 4, 0), (_lib.MV_FLAG_CULL_ON_VIEW_STREAM, 4, 3)],
                         ids=["three_chains", "four_chains_cull_on_view_stream", "four_chains_three_frames_per_stream"])
def test_orbit_with_frames_in_flight_reads_like_one_output_set(extra, in_flight, per_launch):
    """The reference never clears its cube maps: a face a frame does not march keeps older data, and a CubeCast gather across a cube
    edge can read it.  With frames in flight three output sets rotate; the copy-forward of unmarched faces must make every one
    of them read exactly like the single set of a one-stream context.  An orbit that keeps changing face masks, mips and schemes, composites
    compared every few frames (scripts/soak.py runs the same for 40 000 frames).  Also with four output sets on four view streams and
    every frame's upload and cull on its own view stream (MV_FLAG_CULL_ON_VIEW_STREAM) -- and with mv_config.frames_per_launch = 3 outside
    the multi-GPU loop: three consecutive frames share a view stream, twelve output sets and light buffers rotate, nothing else changes."""
    cfg = dict(scene.CONFIGS["T2"], W=320, H=200)
    V, N = cfg["V"], cfg["N"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0)
    cs = [make_caster(cfg, 64), MultiRayCaster(cfg["N"], cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=_lib.MV_FLAG_PIPELINE_FRAMES | extra, framesInFlight=in_flight, framesPerLaunch=per_launch)]
    try:
        frs = [scene.setup_raycaster(c, cfg, vols, use_sh=True) for c in cs]
        differing = 0
        for f in range(1, 601):
            a = 0.02 * f  # ten times the soak's angular speed: about two turns
            e = np.array([90.0 * np.sin(a), 20.0 + 15.0 * np.sin(0.37 * a), -90.0 * np.cos(a)], np.float32)
            vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
            for c, fr in zip(cs, frs):
                c.UpdateFrame(vp, fr["shadowVP"], e)
                c.Render(composite=True)
            if f % 7 == 0:
                a_img, b_img = cs[0].ReadBackComposite(), cs[1].ReadBackComposite()
                differing += int((a_img != b_img).sum())
        assert differing == 0, f"{differing} composite halfs differ between one stream and frames in flight"
        sa, sb = cs[0].GetStats(), cs[1].GetStats()
        for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples"):
            assert sa[k] == sb[k], k
    finally:
        for c in cs:
            c.close()


@pytest.mark.parametrize("L", [10, 24])
def test_light_march_block_order_for_any_light_direction(oracle, L):
    """The light march's voxel blocks are dealt to the XCDs in an order built from the light direction (lightOrderKernel: blocks
    sheared along the dominant axis into tubes, counting sort, one eighth of the order per XCD).  Whatever the direction -- every
    dominant axis, both signs of every slope, axis-aligned, ragged L -- the order must be a permutation of the blocks: the light map
    and the sample count equal the oracle's.  Every light stays for five frames, so that the fifth one finds the order of the first
    still valid in its frame slot (the kernel's early exit)."""
    from tests.helpers import default_descs
    cfg = dict(scene.CONFIGS["C1"], N=24, L=L, V=2, maxRay=16, maxLight=24, W=64, H=64)
    N, M, V = cfg["N"], 32, cfg["V"]
    vols = [rc.synth_volume(N, 3), rc.synth_volume(N, 5)]
    lights = [(75, 75, -75), (0, 100, 0), (100, 0, 0), (0, 0, -100), (-60, 20, 5), (10, -80, 30), (-5, -7, 90), (33, -33, 33), (1, 100, -2), (-90, -89, 88)]
    shadow = np.full((M, M), 0xffff, np.uint16)
    with make_caster(cfg, M) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        c.SetRenderTargets(np.ones((cfg["H"], cfg["W"]), np.float32), shadow)
        descs = default_descs(V, V, N)
        expected = 0
        c.ResetStats()
        lm = np.zeros((V, L, L, L), np.uint32)
        for i, lp in enumerate(lights):
            c.SetLight(lp, scene.LIGHT_COLOR, scene.LIGHT_INTENSITY)
            for rep in range(5):
                c.m_frameIdx = i * 5 + rep
                c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
                c.cullVolumes()
                c.rayMarchL()
                c.synchronize()
                vis, _ = c.ReadBackVisibleLists()
                vid, lm, n = oracle.march_light(V, N, L, M, vols, descs, c.perObject, c.cbPerFrame, shadow, None, vis, cfg["maxLight"], lightMaps=lm)
                expected += n
                assert np.array_equal(c.ReadBackLightMap(vid), lm[vid]), f"light {lp}, frame {rep}: light map of volume {vid} differs"
        assert c.GetStats()["light_samples"] == expected


def test_frames_that_skip_the_view_march_or_only_composite():
    """Call patterns other than mv_render's: frames without a view march (their output set stays behind), frames that only composite
    (the reference would show the cube maps as the last marches left them), frames with neither.  With frames in flight the rotating
    output sets must still read like the one set of a one-stream context: same composites, same cube maps at the end."""
    cfg = dict(scene.CONFIGS["T2"], W=320, H=200)
    V, N = cfg["V"], cfg["N"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0)
    #            cull  light march composite
    kinds = {"full": (1, 1, 1, 1), "no_march": (1, 1, 0, 0), "composite_only": (1, 0, 0, 1), "idle": (1, 0, 0, 0), "march_no_composite": (1, 1, 1, 0),
             "composite_then_march": (1, 1, 2, 1)}
    pattern = ["full", "full", "full", "composite_only", "full", "no_march", "full", "idle", "idle", "full", "march_no_composite", "composite_only",
               "composite_only", "full", "no_march", "no_march", "composite_only", "full", "composite_then_march", "full", "idle", "composite_only", "full"]
    cs = [make_caster(cfg, 64, flags=f) for f in (0, _lib.MV_FLAG_PIPELINE_FRAMES)]
    try:
        frs = [scene.setup_raycaster(c, cfg, vols, use_sh=True) for c in cs]
        checked = 0
        for f in range(1, 8 * len(pattern) + 1):
            kind = pattern[f % len(pattern)]
            cull, light, march, comp = kinds[kind]
            a = 0.03 * f
            e = np.array([90.0 * np.sin(a), 20.0 + 15.0 * np.sin(0.37 * a), -90.0 * np.cos(a)], np.float32)
            vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
            imgs = []
            for c, fr in zip(cs, frs):
                c.UpdateFrame(vp, fr["shadowVP"], e)
                c.cullVolumes()
                if light:
                    c.rayMarchL()
                if march == 2:  # composite first, view march afterwards (the composite shows the previous marches; the march then writes this frame's faces)
                    c.composite()
                    c.rayMarchV()
                else:
                    if march:
                        c.rayMarchV()
                    if comp:
                        c.composite()
                if comp and f % 3 == 0:
                    imgs.append(c.ReadBackComposite())
            if len(imgs) == 2:
                checked += 1
                assert np.array_equal(imgs[0], imgs[1]), f"frame {f} ({kind}): {int((imgs[0] != imgs[1]).sum())} composite halfs differ"
        assert checked > 20
        # a final full frame, then every cube map of the visible cube-map volumes
        for c, fr in zip(cs, frs):
            c.UpdateFrame(vp, fr["shadowVP"], e)
            c.Render(composite=True)
        info = [c.ReadBackVolumeInfo() for c in cs]
        assert np.array_equal(info[0], info[1])
        vis, cube = cs[0].ReadBackVisibleLists()
        for v in cube:
            for mip in range(5):
                assert np.array_equal(cs[0].ReadBackCubeMap(int(v), mip), cs[1].ReadBackCubeMap(int(v), mip)), f"cube map of volume {v}, mip {mip}"
    finally:
        for c in cs:
            c.close()


def test_set_viewport_mid_sequence_equals_a_fresh_context(oracle):
    """SetViewport (MultiRayCaster.cpp:226-241, from SetRenderTargets :211-224 and MultiVolumes.cpp:382 on a resize) re-creates what
    is W x H and keeps volumes, light maps and cube maps: after a resize the context renders what a context created at the new size
    renders (cull decisions, mips and RayCast rectangles follow the viewport), and a second resize back gives the first image again."""
    base = scene.CONFIGS["T2"]
    cfg_a, cfg_b = dict(base, W=320, H=200), dict(base, W=200, H=136)   # (the smaller one moves volumes to another mip / scheme)
    V, N = base["V"], base["N"]
    M = 128
    vols = [rc.synth_volume(N, k) for k in range(V)]

    def frames(c, cfg, fr):
        relight_all(c, fr, V)
        c.rayMarchV()
        c.composite()
        info = c.ReadBackVolumeInfo()
        vis, cube = c.ReadBackVisibleLists()
        return dict(img=c.ReadBackComposite(), info=info, vis=vis, cube=cube,
                    cms=[c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube], lms=[c.ReadBackLightMap(int(v)) for v in vis])

    def fresh(cfg):
        with make_caster(cfg, M) as c:
            return frames(c, cfg, scene.setup_raycaster(c, cfg, vols))

    want_a, want_b = fresh(cfg_a), fresh(cfg_b)
    assert not np.array_equal(want_a["info"], want_b["info"]), "the two viewports must lead to different cull decisions"
    with make_caster(cfg_a, M) as c:
        fr = scene.setup_raycaster(c, cfg_a, vols)
        got = frames(c, cfg_a, fr)
        assert np.array_equal(got["img"], want_a["img"])
        for cfg, want in ((cfg_b, want_b), (cfg_a, want_a), (cfg_b, want_b)):
            c.SetViewport(cfg["W"], cfg["H"])
            c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            c.cullVolumes()
            with pytest.raises(Exception):
                c.rayMarchV()   # the depth map has to be supplied again
            eye, vp = scene.camera(cfg)
            fr = dict(fr, eye=eye, viewProj=vp)
            c.SetRenderTargets(scene.depth_and_shadow(cfg, vp, fr["shadowVP"], M)[0])
            got = frames(c, cfg, fr)
            assert np.array_equal(got["info"], want["info"]) and np.array_equal(got["vis"], want["vis"]) and np.array_equal(got["cube"], want["cube"])
            assert got["img"].shape == (cfg["H"], cfg["W"], 4) and np.array_equal(got["img"], want["img"]), "composite after the resize"
            for x, y in zip(got["cms"], want["cms"]):
                assert np.array_equal(x, y)
            for x, y in zip(got["lms"], want["lms"]):
                assert np.array_equal(x, y)
        # the volumes themselves were never uploaded again
        assert np.array_equal(c.ReadBackVolume(3), vols[3].reshape(c.ReadBackVolume(3).shape))


def test_step_factor_equals_the_ieee_quotient_for_every_input(tmp_path):
    """GetStep's factorEv = min(1/256 / |d|, 2) (RayMarch.hlsli:186) is taken from v_rcp_f32 + one Newton step in the kernels
    (mv_device.hpp StepFactorEv) instead of the compiler's IEEE division.  scripts/micro/rcp_exact.cpp includes that header and compares
    the two over ALL 2^32 float bit patterns on the GPU; it fails if any input below 2^118 in magnitude, or any infinity / NaN, differs
    (a density difference of half-float texels is at most 131008)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "rcp_exact.bin")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-w", "-o", exe,
                    os.path.join(root, "scripts", "micro", "rcp_exact.cpp")], check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 infinities / NaNs" in r.stdout
