"""Dispatch order of the view march (launchTileOrder): it may only change WHICH wavefront marches a tile, never the result.

Same moving-camera frame sequence on a context with the cost-binned order and on one with list order (MV_FLAG_LIST_ORDER);
the cube maps are cleared before every frame, so a tile the order skipped (or marched with another tile's
parameters) would show up as zeros / wrong texels, and the sample counters would differ.
"""
import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, _lib, scene
from multivolumes_amd import raycaster as rc

pytestmark = pytest.mark.gpu


def sequence(cfg, vols, ordered, frames):
    c = MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=64,
                       flags=0 if ordered else _lib.MV_FLAG_LIST_ORDER)
    out = []
    with c:
        fr = scene.setup_raycaster(c, cfg, vols)
        eye = np.array(fr["eye"], np.float32)
        for f in range(frames):
            # three still frames (the cost table becomes exact), then a camera that moves and dollies in (mips change)
            k = max(0, f - 3)
            e = eye * (1.0 - 0.06 * k) + np.array([1.3 * k, -0.4 * k, 0.9 * k], np.float32)
            vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0))
            c.SetRenderTargets(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), scene.SPHERE_RADIUS))
            c.ClearCubeMaps()
            c.UpdateFrame(vp, fr["shadowVP"], e)
            c.Render(composite=True)
            info = c.ReadBackVolumeInfo()
            vis, cube = c.ReadBackVisibleLists()
            out.append((info, cube, [c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube],
                        [c.ReadBackCubeDepth(int(v), int(info[v, 0])) for v in cube], c.ReadBackComposite()))
        st = c.GetStats()
    return out, {k: st[k] for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples")}


@pytest.mark.parametrize("name,over", [("T2", dict(W=320, H=200)), ("T1", dict(W=200, H=120, N=40))])
def test_cost_binned_order_changes_no_bit(name, over):
    cfg = dict(scene.CONFIGS[name], **over)
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    a, sa = sequence(cfg, vols, True, 9)
    b, sb = sequence(cfg, vols, False, 9)
    assert sa == sb and sa["rays"] > 0
    mips = set()
    for x, y in zip(a, b):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and len(x[1]) > 0
        mips.update(int(m) for m in x[0][x[1], 0])
        for p, q in zip(x[2], y[2]):
            assert np.array_equal(p, q) and p.any()
        for p, q in zip(x[3], y[3]):
            assert np.array_equal(p, q)
        assert np.array_equal(x[4], y[4])
    assert len(mips) > 1, "the sequence must cross a mip change (stale cost keys)"
