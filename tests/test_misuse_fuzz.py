"""Misuse fuzz (round 6): calls in any order -- stages before their inputs or before the frame's cull, the light march behind the view march, two culls, no
update, indices out of range, setters at any point, a viewport change in the middle of a frame -- must each return MV_OK or a negative code (the reference's
creation functions return false, its record functions cannot fail: include/multivolumes.h), never crash, hang or poison the context: a whole frame afterwards
equals the same frame on a fresh context.  On a one-stream context and on the four-chain context.  MV_MISUSE_FUZZ=n seeds (default 6)."""
import os

import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, MVError, _lib, scene
from multivolumes_amd import raycaster as rc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("MV_MISUSE_FUZZ", "6"))))
def test_calls_in_any_order_fail_cleanly_and_leave_the_context_usable(seed):
    r = np.random.default_rng(4000 + seed)
    cfg = dict(scene.CONFIGS["T2"], N=int(r.choice([16, 24, 32])), L=int(r.integers(6, 20)), V=int(r.choice([1, 4, 9])), W=int(r.integers(64, 200)), H=int(r.integers(48, 150)),
               maxRay=32, maxLight=16)
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    flags, in_flight = ((0, 0), (_lib.MV_FLAG_PIPELINE_FRAMES | _lib.MV_FLAG_CULL_ON_VIEW_STREAM, 4), (_lib.MV_FLAG_PIPELINE_FRAMES, 0))[seed % 3]
    eye, vp = scene.camera(cfg)
    svp = scene.shadow_vp(cfg)
    depth, shadow = scene.depth_and_shadow(cfg, vp, svp, 64)

    def whole_frame(c, W, H):
        e2, vp2 = scene.camera(dict(cfg, W=W, H=H))
        c.SetRenderTargets(rc.synth_sphere_depth(W, H, vp2, (0, 0, 0), 8.0), shadow)
        c.SetSH(None)
        c.SetMaxSamples(cfg["maxRay"], cfg["maxLight"])
        c.SetLight(scene.LIGHT_PT, scene.LIGHT_COLOR, scene.LIGHT_INTENSITY)
        for k, v in enumerate(vols):
            c.LoadVolumeData(k, v)
        for f in range(V + 2):   # every volume relit with these inputs
            c.m_frameIdx = f
            c.UpdateFrame(vp2, svp, e2)
            c.Render(composite=True)
        vis, _ = c.ReadBackVisibleLists()
        return c.ReadBackComposite(), np.stack([c.ReadBackLightMap(int(v)) for v in vis]), c.ReadBackVolumeInfo()[vis], vis

    c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight)
    W, H = cfg["W"], cfg["H"]
    ok = errors = 0
    try:
        c.SetVolumesWorld(scene.VOLUME_SIZE, (0.0, 0.0, 0.0))
        ops = ["update", "cull", "light", "view", "composite", "render", "depth", "shadow", "sh", "volume", "bad_volume", "init_bad", "descs_bad", "light_map", "readbacks", "viewport",
               "max_samples_bad", "sync", "stats", "composite_rows_bad", "readback_bad", "upload_bad", "clear"]
        for step in range(int(r.integers(40, 120))):
            op = ops[int(r.integers(len(ops)))]
            try:
                if op == "update":
                    c.UpdateFrame(vp, svp, eye)
                elif op == "cull":
                    c.cullVolumes()
                elif op == "light":
                    c.rayMarchL()
                elif op == "view":
                    c.rayMarchV()
                elif op == "composite":
                    c.composite()
                elif op == "render":
                    c.Render(composite=bool(r.integers(2)))
                elif op == "depth":
                    c.SetRenderTargets(rc.synth_sphere_depth(W, H, vp, (0, 0, 0), float(r.uniform(4, 12))))
                elif op == "shadow":
                    c.SetRenderTargets(shadowMap=shadow)
                elif op == "sh":
                    c.SetSH(scene.DEFAULT_SH if r.integers(2) else None)
                elif op == "volume":
                    c.LoadVolumeData(int(r.integers(V)), vols[int(r.integers(V))])
                elif op == "bad_volume":
                    c._check(c._lib.mv_volume_upload_rgba16f(c._ctx, V + int(r.integers(0, 5)), rc._ptr(vols[0])), "mv_volume_upload_rgba16f")
                elif op == "init_bad":
                    c.InitVolumeData(V + int(r.integers(0, 1000)))
                elif op == "descs_bad":
                    c.SetVolumeDescs([(V + 3) | (5 << 14) | (N << 18)] * V)
                elif op == "light_map":
                    c.UploadLightMap(int(r.integers(V)), np.zeros((L, L, L), np.uint32))
                elif op == "readbacks":
                    c.ReadBackVolumeInfo(); c.ReadBackVisibleLists(); c.ReadBackLightMap(int(r.integers(V))); c.ReadBackComposite(); c.ReadBackCubeMap(int(r.integers(V)), int(r.integers(5))); c.ReadBackCull()
                elif op == "viewport":
                    W, H = int(r.integers(8, 260)), int(r.integers(8, 200))
                    c.SetViewport(W, H)   # (no depth map now: the stages that need one must say so)
                    eye, vp = scene.camera(dict(cfg, W=W, H=H))
                elif op == "max_samples_bad":
                    c.SetMaxSamples(int(r.choice([0, 70000])), 16)
                elif op == "sync":
                    c.synchronize()
                elif op == "stats":
                    c.GetStats(); c.ResetStats()
                elif op == "composite_rows_bad":
                    c.composite(int(r.integers(H, H + 50)), int(r.integers(1, H + 1)))
                elif op == "readback_bad":
                    c.ReadBackCubeMap(V + 1, 0)
                elif op == "upload_bad":
                    c.UploadLightMap(V, np.zeros((L, L, L), np.uint32))
                else:
                    c.ClearCubeMaps()
                ok += 1
            except MVError as ex:
                assert ex.code < 0, ex
                errors += 1
            except ValueError:   # (the Python mirror's own size checks)
                errors += 1
        assert ok > 5 and errors > 5, (ok, errors)
        # the context is still a context: a whole frame with known inputs ...
        c.SetViewport(cfg["W"], cfg["H"])
        got = whole_frame(c, cfg["W"], cfg["H"])
    finally:
        c.close()
    # ... equals the same on a fresh one (a fresh context starts from zero cube maps and light maps: clear this one's history the same way is not
    # possible for light maps, hence the V + 2 relighting frames above; cube faces that are not marched keep history -- compare what the frame wrote)
    with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight) as f:
        f.SetVolumesWorld(scene.VOLUME_SIZE, (0.0, 0.0, 0.0))
        exp = whole_frame(f, cfg["W"], cfg["H"])
    # (VolumeInfo rows and light maps of the VISIBLE volumes: a culled volume's row and map keep what earlier frames left, as in the reference)
    assert np.array_equal(got[3], exp[3]) and np.array_equal(got[2], exp[2]), "visible list / VolumeInfo"
    assert np.array_equal(got[1], exp[1]), f"seed {seed}: light maps after the misuse differ from a fresh context's"
    assert np.array_equal(got[0], exp[0]) or _only_history_differs(got[0], exp[0]), f"seed {seed}: {(got[0] != exp[0]).sum()} composite halfs differ"


def _only_history_differs(a, b):
    """The composite's CubeCast gathers across cube edges may read faces a frame does not march, which hold the history of earlier frames (the reference never
    clears its cube maps): allowed to differ in a thin set of pixels along silhouettes, not in bulk."""
    d = (a != b).any(axis=2)
    return d.sum() <= 0.002 * d.size


@pytest.mark.parametrize("seed", range(int(os.environ.get("MV_GARBAGE_FUZZ", "4"))))
def test_garbage_frame_constants_never_fault(seed):
    """CBPerFrame / PerObject are the caller's bytes (mv_update_frame takes them raw, as the reference's upload heap does): NaNs, infinities, huge and denormal
    values in them, or plain random bytes, may make a frame's RESULTS meaningless -- never its memory accesses.  Every float that becomes an index passes a clamp
    first, every count derived from a projected rectangle is bounded: hundreds of frames of garbage on the one-stream and the four-chain context, the process lives,
    and a proper frame afterwards equals a fresh context's.  (A GPU memory fault aborts the process: the test then dies instead of failing -- run it alone.)"""
    r = np.random.default_rng(6000 + seed)
    cfg = dict(scene.CONFIGS["T2"], N=int(r.choice([12, 24, 32])), L=int(r.integers(4, 16)), V=int(r.choice([1, 4, 9])), W=int(r.integers(8, 200)), H=int(r.integers(8, 150)), maxRay=32, maxLight=16)
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    flags, in_flight = ((0, 0), (_lib.MV_FLAG_PIPELINE_FRAMES | _lib.MV_FLAG_CULL_ON_VIEW_STREAM, 4))[seed % 2]
    BAD = np.array([np.nan, np.inf, -np.inf, 1e38, -1e38, 1e-45, -1e-45, 0.0, -0.0, 3.4e38, 1e20, -1e20, 65536.0, -1.0], np.float32)

    def proper(c):
        fr = scene.setup_raycaster(c, cfg, vols)
        c.ClearCubeMaps()   # (faces a frame does not march keep what earlier frames left -- here garbage: back to the creation state, as on the fresh context)
        for f in range(V + 2):
            c.m_frameIdx = f
            c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            c.Render(composite=True)
        vis, _ = c.ReadBackVisibleLists()
        return c.ReadBackComposite(), np.stack([c.ReadBackLightMap(int(v)) for v in vis] + [np.zeros((L, L, L), np.uint32)]), c.ReadBackVolumeInfo()[vis]

    with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight) as c:
        fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        good_cb, good_po = c.cbPerFrame.copy(), c.perObject.copy()
        for k in range(150):
            kind = int(r.integers(4))
            cb, po = good_cb.copy(), good_po.copy()
            if kind == 0:      # a few floats of the good constants replaced
                for blob in (cb[:208].view(np.float32), po.view(np.float32)):
                    n = int(r.integers(1, 6))
                    blob[r.integers(0, blob.size, size=n)] = BAD[r.integers(0, len(BAD), size=n)]
            elif kind == 1:    # every float of one of the two from the bad set
                blob = (cb[:208].view(np.float32), po.view(np.float32))[int(r.integers(2))]
                blob[:] = BAD[r.integers(0, len(BAD), size=blob.size)]
            elif kind == 2:    # random bytes
                cb[:] = r.integers(0, 256, size=cb.size, dtype=np.uint8)
                po[:] = r.integers(0, 256, size=po.size, dtype=np.uint8)
            else:              # good matrices scaled by wild factors (volumes of size 1e-20 .. 1e20, the eye at 1e30)
                po.view(np.float32)[:] *= np.float32(10.0) ** r.integers(-25, 25)
                cb[:16].view(np.float32)[:3] *= np.float32(10.0) ** r.integers(-20, 30)
            cb[208:212].view(np.uint32)[0] = int(r.integers(0, 2 ** 32))   # FrameIdx: any
            c.UpdateFrameRaw(cb, po)
            c.Render(composite=bool(r.integers(2)))
            if k % 25 == 24:
                c.synchronize()
                c.ReadBackComposite(); c.GetStats()
        c.synchronize()
        got = proper(c)
    with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight) as f:
        exp = proper(f)
    assert np.array_equal(got[2], exp[2]) and np.array_equal(got[1], exp[1])
    assert np.array_equal(got[0], exp[0]), f"seed {seed}: {(got[0] != exp[0]).sum()} composite halfs differ after the garbage"
