"""Analytic known-answer tests for the COMPOSITE half of the CPU oracle and for its two optional
modes (8-bit filter weights; the LibRayMarch statement of the view march).

The reference ships no vectors ("parity unpinned"); these closed-form cases are derived by hand from
Sh/PSDepthPeel.hlsl:12-24, Sh/PSCube.hlsl:30-60, Sh/PSCube.hlsli:21-108, Sh/PSResolveOIT.hlsl:12-26 and
Sh/LibRayMarch.hlsl:205-287.  What they do NOT pin is listed in DESIGN.md section 2 (rasteriser coverage at silhouette
pixels, interpolation rounding of LPt, TextureCube filtering across face edges -- all implementation-defined in D3D).
"""
import ctypes as C
import math

import numpy as np
import pytest

from multivolumes_amd import _lib
from multivolumes_amd import raycaster as rc
from tests.conftest import half_to_float

F = np.float32
CUBE_BIT = 0x8000


def scene(positions, eye=(0.0, 0.0, -5.0), W=64, H=64, size=2.0):
    """Unit cubes (local == world up to a translation) at `positions`, camera on `eye` looking at the origin."""
    lib = _lib.load()
    V = len(positions)
    worlds = np.zeros((V, 12), np.float32)
    for i, p in enumerate(positions):
        lib.mvh_set_volume_world(F(size), rc._ptr(rc._f32(p, 3)), rc._ptr(worlds[i]))
    vp = rc.multiply(rc.look_at_lh(eye, (0, 0, 0), (0, 1, 0)), rc.perspective_fov_lh(math.pi / 4, W / H, 1.0, 1000.0))
    svp = rc.shadow_view_proj((75, 75, -75), 1.0)
    cb = np.zeros(212, np.uint8)
    po = np.zeros(V * 224, np.uint8)
    one4 = np.ones(4, np.float32)
    lib.mvh_update_frame(V, rc._ptr(worlds), rc._ptr(rc._f32(vp, 16)), rc._ptr(rc._f32(svp, 16)), rc._ptr(rc._f32(eye, 3)),
                         float(W), float(H), rc._ptr(rc._f32((75, 75, -75), 3)), rc._ptr(one4), rc._ptr(one4), 0, rc._ptr(cb), rc._ptr(po))
    return cb, po


def constant_cube_maps(oracle, V, N, colours, depth=1.0):
    """Cube maps whose every texel (every mip, every face) of volume v is colours[v]; cube depths = `depth`."""
    chain = oracle.cube_chain_texels(N)
    cm = np.zeros((V, chain, 4), np.float16)
    for v, c in enumerate(colours):
        cm[v, :, :] = np.asarray(c, np.float16)
    return cm.view(np.uint16), np.full((V, chain), depth, np.float32)


def run_composite(oracle, positions, colours, W=64, H=64, N=8, visible=None, depth_map=None, cube_depth=1.0, flags=None):
    V = len(positions)
    cb, po = scene(positions, W=W, H=H)
    cm, cd = constant_cube_maps(oracle, V, N, colours, cube_depth)
    info = np.zeros((V, 4), np.uint16)
    for v in range(V):
        info[v] = (0, 16, (CUBE_BIT | 0x3F) if flags is None else flags[v], 0)
    vols = [np.zeros((N, N, N, 4), np.uint16)]
    lms = np.zeros((V, 4, 4, 4), np.uint32)
    depth_map = np.ones((H, W), np.float32) if depth_map is None else depth_map
    vis = np.arange(V, dtype=np.uint32) if visible is None else np.asarray(visible, np.uint32)
    img, n = oracle.composite(V, N, 4, W, H, vols, lms, depth_map, po, cb, info, vis, cm, cd)
    return half_to_float(img), n


def test_kat_a7_i_one_volume_homogeneous_cube_map(oracle):
    """(i) One cube, eye on its axis, every cube-map texel = c: CubeCast returns c whatever the weights (normalised), the
    resolve over one layer is c * (1 - 0); pixels off the silhouette stay 0.  The silhouette is the projection of the near
    face: |ndc| <= (1/4) / tan(22.5 deg) = 0.6036 => pixel centres 13..50 of 64, i.e. 38^2 covered pixels."""
    c = (0.25, 0.5, 0.75, 0.5)
    img, n = run_composite(oracle, [(0, 0, 0)], [c])
    assert n == 0, "no RayCast on the cube-map scheme"
    covered = img[..., 3] > 0
    assert covered.sum() == 38 * 38
    assert covered[13:51, 13:51].all()
    assert np.all(img[covered] == np.asarray(c, np.float32)), "every covered pixel is exactly the texel value"
    assert not img[~covered].any()


def test_kat_a7_ii_two_volumes_peel_order_and_resolve(oracle):
    """(ii) Two cubes on the view axis: fragments are the EXIT points (front-face culling, MultiRayCaster.cpp:868), ordered by
    asuint(z) (PSDepthPeel.hlsl:12-24) whatever the order of the visible list; resolve = result += src * (1 - result.a)."""
    c0, c1 = (0.5, 0.25, 0.125, 0.25), (0.125, 0.5, 0.75, 0.5)  # near, far
    want = np.array([0.5 + 0.125 * 0.75, 0.25 + 0.5 * 0.75, 0.125 + 0.75 * 0.75, 0.25 + 0.5 * 0.75], np.float32)  # exact in fp16
    for order in ([0, 1], [1, 0]):
        img, _ = run_composite(oracle, [(0, 0, 0), (0, 0, 4)], [c0, c1], visible=order)
        assert np.array_equal(img[32, 32], want)
        assert np.array_equal(img[30:35, 30:35], np.broadcast_to(want, (5, 5, 4)))
    # swapped positions: the other one is in front
    img, _ = run_composite(oracle, [(0, 0, 4), (0, 0, 0)], [c0, c1])
    want2 = np.array([0.125 + 0.5 * 0.5, 0.5 + 0.25 * 0.5, 0.75 + 0.125 * 0.5, 0.5 + 0.25 * 0.5], np.float32)
    assert np.array_equal(img[32, 32], want2)


def test_kat_a7_eight_layers_alpha_rule_and_clamp(oracle):
    """NUM_OIT_LAYERS = 8 nearest fragments are kept (SharedConsts.h:7); a layer is stored only if 0 < alpha <= 1
    (PSCube.hlsl:57); the resolved alpha is clamped to 0.9997 (PSResolveOIT.hlsl:23)."""
    V = 10
    pos = [(0, 0, 3.0 * k) for k in range(V)]
    cols = [(2.0 ** -(k % 4 + 1), 0.25, 0.5, 0.25) for k in range(V)]
    cols[2] = (1.0, 1.0, 1.0, 1.5)   # alpha > 1: dropped, but it still occupies one of the 8 peeled layers
    cols[4] = (1.0, 1.0, 1.0, 0.0)   # alpha == 0: dropped likewise
    img, _ = run_composite(oracle, pos, cols, visible=list(reversed(range(V))))
    r = np.zeros(4, np.float32)
    for k in range(8):  # volumes 8 and 9 are beyond the eighth layer
        c = np.asarray(cols[k], np.float16).astype(np.float32)
        if 0.0 < c[3] <= 1.0:
            r = r + c * (F(1.0) - r[3])
    r[3] = min(r[3], F(0.9997))
    assert np.array_equal(img[32, 32], r.astype(np.float16).astype(np.float32))
    # alpha clamp: opaque layers
    img, _ = run_composite(oracle, [(0, 0, 0), (0, 0, 3)], [(0.5, 0.5, 0.5, 1.0), (0.25, 0.25, 0.25, 1.0)])
    assert img[32, 32, 3] == np.float32(np.float16(0.9997)) and img[32, 32, 0] == 0.5, "the far layer is hidden by (1 - 1)"


def test_kat_a7_iii_unproject_z(oracle):
    """(iii) UnprojectZ(d) = zN zF / (d (zN - zF) + zF), zN = 1, zF = 1000 (PSCube.hlsli:21-26, SharedConsts.h:9-10)."""
    lib = oracle.load()
    assert lib.mvo_unproject_z(0.0) == 1.0
    assert lib.mvo_unproject_z(1.0) == 1000.0
    assert lib.mvo_unproject_z(0.5) == pytest.approx(1000.0 / 500.5, rel=1e-6)
    # inverse of the projection: view z -> depth -> view z
    for zv in (1.5, 6.0, 10.0, 250.0):
        d = (1000.0 / 999.0) * (1.0 - 1.0 / zv)
        assert lib.mvo_unproject_z(F(d)) == pytest.approx(zv, rel=2e-4)


def depth_of(zv):
    return F((1000.0 / zv - 1000.0) / (1.0 - 1000.0))


def planted_face(oracle, d=2):
    """A d = 2 cube map whose face +Z holds four distinct colours and four distinct depths (view z 10, 10.5, 11, 13)."""
    cm = np.zeros((6, d, d, 4), np.float16)
    cd = np.ones((6, d, d), np.float32)
    cols = {(0, 0): (0.5, 0.25, 0.125, 0.5), (1, 0): (0.25, 0.5, 0.75, 0.25), (0, 1): (0.75, 0.125, 0.5, 0.75), (1, 1): (1.0, 1.0, 1.0, 1.0)}
    zs = {(0, 0): 10.0, (1, 0): 10.5, (0, 1): 11.0, (1, 1): 13.0}
    for (i, j), c in cols.items():
        cm[4, j, i] = c
        cd[4, j, i] = depth_of(zs[(i, j)])
    return cm, cd, cols, zs


def cube_cast(oracle, cm, cd, depth_pix, pos, ray_dir):
    out = np.zeros(4, np.float32)
    p, r = np.asarray(pos, np.float32), np.asarray(ray_dir, np.float32)
    cmu = np.ascontiguousarray(cm.view(np.uint16))
    oracle.load().mvo_cube_cast(rc._ptr(cmu), rc._ptr(np.ascontiguousarray(cd)), cm.shape[1], F(depth_pix), rc._ptr(p), rc._ptr(r), rc._ptr(out))
    return out


def test_kat_a7_iv_depth_weights(oracle):
    """(iv) CubeCast at the centre of face +Z of a 2 x 2 map: the four Gather taps have bilinear weight 1/4 each and depth
    weight max(1 - |Z(mesh) - Z(cube)| / 2, 0) (PSCube.hlsli:88-90) = 1, 0.75, 0.5, 0 for the planted view depths against a
    mesh at view z = 10."""
    cm, cd, cols, zs = planted_face(oracle)
    got = cube_cast(oracle, cm, cd, depth_of(10.0), (0, 0, 1), (0, 0, 1))
    w = {k: max(1.0 - 0.5 * abs(10.0 - z), 0.0) * 0.25 for k, z in zs.items()}
    want = sum(np.asarray(cols[k], np.float64) * w[k] for k in cols) / sum(w.values())
    assert got == pytest.approx(want, rel=2e-4)
    # the tap with weight 0 does not contribute at all: change its colour, same bits out
    cm2 = cm.copy()
    cm2[4, 1, 1] = (0.0, 0.0, 0.0, 0.0)
    assert np.array_equal(cube_cast(oracle, cm2, cd, depth_of(10.0), (0, 0, 1), (0, 0, 1)), got)
    # a far mesh in front of nothing: the sky pixel (depth 1.0 -> Z = 1000) against cube depths of 1.0 weighs every tap 1
    cd1 = np.ones_like(cd)
    got = cube_cast(oracle, cm, cd1, 1.0, (0, 0, 1), (0, 0, 1))
    want = sum(np.asarray(c, np.float64) for c in cols.values()) / 4.0
    assert got == pytest.approx(want, rel=1e-6)


def test_kat_a7_v_zero_weight_fallback(oracle):
    """(v) Every depth weight 0 (mesh at view z = 100): `ws > 0 ? result / ws : color` (PSCube.hlsli:105) falls back to the
    plain bilinear SampleLevel colour."""
    cm, cd, cols, _ = planted_face(oracle)
    got = cube_cast(oracle, cm, cd, depth_of(100.0), (0, 0, 1), (0, 0, 1))
    want = sum(np.asarray(c, np.float64) for c in cols.values()) / 4.0
    assert got == pytest.approx(want, rel=1e-6)
    # off-centre: bilinear weights (s, t) = (0.625, 0.5) -> fx = 0.75, fy = 0.5
    got = cube_cast(oracle, cm, cd, depth_of(100.0), (0.25, 0.0, 1.0), (0, 0, 1))
    fx, fy = 0.75, 0.5
    want = (np.asarray(cols[(0, 0)], np.float64) * (1 - fx) * (1 - fy) + np.asarray(cols[(1, 0)], np.float64) * fx * (1 - fy)
            + np.asarray(cols[(0, 1)], np.float64) * (1 - fx) * fy + np.asarray(cols[(1, 1)], np.float64) * fx * fy)
    assert got == pytest.approx(want, rel=1e-6)


def test_kat_a7_face_selection_and_edge_rule(oracle):
    """Face selection agrees with GetLocalPos (CSRayMarch.hlsl:28-53): the direction of texel (x, y) of face f maps back to
    (f, (x + 0.5) / d, (y + 0.5) / d).  A Gather tap outside its face resolves to the texel its centre re-projects to on the
    neighbouring face (the rule this build states for D3D's implementation-defined edge filtering): from +Z, one texel past
    the s = 1 edge lands in column 0 of +X, the same row."""
    lib = oracle.load()
    d = 4
    for f in range(6):
        for (x, y) in ((0, 0), (3, 1), (2, 3)):
            p = np.zeros(3, np.float32)
            lib.mvo_get_local_pos(F(x), F(y), f, d, rc._ptr(p))
            face, st = C.c_uint32(), np.zeros(2, np.float32)
            lib.mvo_cube_face_coords(rc._ptr(p), C.byref(face), rc._ptr(st))
            assert face.value == f and st[0] == pytest.approx((x + 0.5) / d) and st[1] == pytest.approx((y + 0.5) / d)
    for j in range(d):
        assert lib.mvo_cube_texel_index(4, d, j, d) == (0 * d + j) * d + 0
        assert lib.mvo_cube_texel_index(4, -1, j, d) == (1 * d + j) * d + (d - 1), "past the s = 0 edge of +Z: last column of -X"
        assert lib.mvo_cube_texel_index(4, 2, j, d) == (4 * d + j) * d + 2, "inside the face: unchanged"


def test_kat_a7_exterior_edge_domain_clamp(oracle):
    """GetDomain (PSCube.hlsli:31-46): on the outermost texel band, when the view ray points back across the edge
    (axes * rayDir < 0), the domain collapses onto the interior texel: uv = min(uv, d - 0.5); domain = uv < 0.5 ? 1 : 0."""
    cm, cd, cols, _ = planted_face(oracle)
    cd1 = np.ones_like(cd)
    # pos.x * d = 1.98 > bound = 1 and rayDir.x < 0: uv = (min(1.99, 1.5), 1) -> domain (0, 0) -> only the (i0, j0) tap
    got = cube_cast(oracle, cm, cd1, 1.0, (0.99, 0.0, 1.0), (-0.5, 0.0, 1.0))
    assert got == pytest.approx(np.asarray(cols[(1, 0)], np.float64), rel=1e-6)
    # the same point looked at from the other side keeps the bilinear domain (frac(1.99 + 0.5), frac(1 + 0.5)) = (0.49, 0.5)
    got2 = cube_cast(oracle, cm, cd1, 1.0, (0.99, 0.0, 1.0), (0.5, 0.0, 1.0))
    assert not np.allclose(got2, got)


def test_libraymarch_statement_differs_only_by_the_target_clamp(oracle):
    """Sh/LibRayMarch.hlsl:205-287 states the view march a second time, without `tMax = min(tMax, ComputeTargetHit)`
    (CSRayMarch.hlsl:102-106).  The target IS the exit point of the ray, so without the clamp a ray ends at its next sample
    (`any(abs(pos) > 1)`) instead of at `t > tMax`: both statements take the same samples, except that a ray whose last step
    lands within rounding of the exit face may take one more.  Everything else must agree bit for bit."""
    from tests.helpers import oracle_pipeline
    from multivolumes_amd import scene as S
    cfg = dict(S.CONFIGS["T1"])
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, 64)
    oracle.load().mvo_set_options(0, 1)
    try:
        cm, cd, st = oracle.march_view(cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"],
                                       ref["po"], ref["cb"], ref["info"], ref["cube"])
    finally:
        oracle.load().mvo_set_options(0, 0)
    assert st["rays"] == ref["stats"]["rays"]
    extra = st["view_samples"] - ref["stats"]["view_samples"]
    assert 0 <= extra <= 0.02 * st["rays"], f"{extra} extra samples over {st['rays']} rays"
    assert np.array_equal(cd, ref["cubeDepths"])
    a, b = half_to_float(cm), half_to_float(ref["cubeMaps"])
    differing = int((np.abs(a - b).max(axis=-1) > 0).sum())
    assert differing <= extra, "only rays that took the extra sample may differ"
    print(f"LibRayMarch statement: {extra} extra samples, {differing} texels differ, max |diff| {np.abs(a - b).max():.3g}")


def test_eight_bit_filter_weights_bound(oracle):
    """D3D lets the fixed-function sampler snap texel coordinates to 8 fractional bits; this build filters with exact fp32
    weights.  The optional 8-bit mode of the oracle bounds what a capture of the reference could differ by because of that."""
    lib = oracle.load()
    N = 8
    ramp = np.zeros((N, N, N, 4), np.float16)
    ramp[..., 3] = (np.arange(N, dtype=np.float32) / N)[None, None, :]
    ramp = ramp.view(np.uint16)
    zero3 = np.zeros(3, np.int32)

    def alpha(u):
        out = np.zeros(4, np.float32)
        lib.mvo_sample_rgba16f(rc._ptr(ramp), N, rc._ptr(np.array([u, 0.5, 0.5], np.float32)), rc._ptr(zero3), rc._ptr(out))
        return float(out[3])

    us = np.linspace(0.07, 0.93, 173)
    exact = np.array([alpha(u) for u in us])
    lib.mvo_set_options(8, 0)
    try:
        snapped = np.array([alpha(u) for u in us])
        # at a coordinate that IS a multiple of 1/256 texel both modes agree exactly
        u0 = (3.0 + 0.5 + 64.0 / 256.0) / N
        a8 = alpha(u0)
    finally:
        lib.mvo_set_options(0, 0)
    assert a8 == alpha(u0)
    err = np.abs(snapped - exact).max()
    assert 0 < err <= (1.0 / N) * (0.5 / 256.0) * 1.001, "at most half a 1/256 texel step times the texel-to-texel difference"

    # whole pipeline at T1: report the deviation (marchV threshold flips make it larger than the filter error alone)
    from tests.helpers import oracle_pipeline
    from multivolumes_amd import scene as S
    cfg = dict(S.CONFIGS["T1"])
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, 64)
    lib.mvo_set_options(8, 0)
    try:
        q = oracle_pipeline(oracle, cfg, vols, 64)
    finally:
        lib.mvo_set_options(0, 0)
    assert np.array_equal(q["info"], ref["info"]), "the cull does not sample"
    d = np.abs(half_to_float(q["cubeMaps"]) - half_to_float(ref["cubeMaps"]))
    frac = float((d.max(axis=-1) > 1e-3).mean())
    print(f"8-bit filter weights at T1: max |cube map diff| {d.max():.3g}, mean {d.mean():.3g}, texels beyond 1e-3: {100 * frac:.2f} %")
    # a ray whose alpha crosses ZERO_THRESHOLD or whose step changes under the snapped weights takes other samples: single texels
    # move by far more than the filter error; the bulk does not
    assert d.mean() < 2e-3 and frac < 0.25


def test_kat_a7_screen_space_raycast_layer(oracle):
    """A volume on the screen-space scheme (MaskBits without CUBEMAP_RAYMARCH_BIT, VSCube.hlsl:73 -> SmpCnt > 0) is marched per pixel by
    RayCast (RayCast.hlsli:43-108): the same loop as the view march, from the box entry along normalize(LPt - eye), no target clamp,
    tMax from the pixel's own depth.  Homogeneous volume + constant light: the centre pixel equals the closed-form recurrence of KAT 1,
    and a mesh plane in front of the far face cuts the march (GetTMax, RayMarch.hlsli:82-97)."""
    from tests.test_oracle_kat import emulate_march, homogeneous_volume, MAX_DIST
    lib = oracle.load()
    N, L, smp, W, H = 8, 4, 16, 64, 64
    alpha, c = F(np.float16(0.3)), F(np.float16(0.5))
    vol = homogeneous_volume(N, c, alpha)
    cb, po = scene([(0, 0, 0)], W=W, H=H)
    info = np.array([[0, smp, 0x3F, 0]], np.uint16)  # all faces, screen-space scheme
    lm = np.full((1, L, L, L), lib.mvo_pack_r11g11b10f(1.0, 1.0, 1.0), np.uint32)
    chain = oracle.cube_chain_texels(N)
    cm, cd = np.zeros((1, chain, 4), np.uint16), np.ones((1, chain), np.float32)
    vis = np.array([0], np.uint32)

    def centre(depth_map):
        img, n = oracle.composite(1, N, L, W, H, [vol], lm, depth_map, po, cb, info, vis, cm, cd)
        return half_to_float(img)[32, 32], n

    # the pixel's ray: eye (0, 0, -5), through ndc (1/64, -1/64) -> enters at z = -1
    t = math.tan(math.pi / 8)
    d = np.array([(32.5 / 32 - 1) * t, -(32.5 / 32 - 1) * t, 1.0], np.float64)
    d /= np.linalg.norm(d)
    o = np.array([0, 0, -5], np.float64) + d * (4.0 / d[2])
    rgb, A, it = emulate_march(alpha, c, 1.0, smp, F(3.0e38), o.astype(F), d.astype(F))
    px, n = centre(np.ones((H, W), np.float32))
    assert n > 0
    assert px[3] == pytest.approx(float(A), abs=1e-3) and px[0] == pytest.approx(float(rgb), abs=1e-3) and px[0] == px[1] == px[2]
    # closed form: A_n = 1 - (1 - 0.8 a)^n with n = iterations of the recurrence
    assert float(A) == pytest.approx(1 - (1 - 0.8 * float(alpha)) ** it, abs=2e-3)

    # a mesh plane at view z = 5.0 (local z = 0: half way through the cube) stops the ray there
    zv = 5.0
    plane = np.full((H, W), (1000.0 / 999.0) * (1.0 - 1.0 / zv), np.float32)
    t_max = (0.0 - o[2]) / d[2]
    rgb2, A2, it2 = emulate_march(alpha, c, 1.0, smp, F(t_max), o.astype(F), d.astype(F))
    px2, _ = centre(plane)
    assert it2 < it and px2[3] == pytest.approx(float(A2), abs=1e-3) and px2[0] == pytest.approx(float(rgb2), abs=1e-3)


def test_kat_a7_resolve_over_a_background(oracle):
    """The resolve pass is drawn over the scene's colour target with premultiplied-alpha blending (MultiRayCaster.cpp:931, 1615-1633):
    out = src + dst (1 - src.a) on all four channels; pixels the volumes do not cover keep the background."""
    V, N, W, H = 1, 8, 64, 64
    c = (0.25, 0.5, 0.75, 0.5)
    cb, po = scene([(0, 0, 0)], W=W, H=H)
    cm, cd = constant_cube_maps(oracle, V, N, [c])
    info = np.array([[0, 16, CUBE_BIT | 0x3F, 0]], np.uint16)
    bg = np.zeros((H, W, 4), np.float16)
    bg[...] = (0.5, 0.25, 1.0, 1.0)
    img, _ = oracle.composite(V, N, 4, W, H, [np.zeros((N, N, N, 4), np.uint16)], np.zeros((V, 4, 4, 4), np.uint32), np.ones((H, W), np.float32),
                              po, cb, info, np.array([0], np.uint32), cm, cd, background=bg.view(np.uint16))
    img = half_to_float(img)
    assert np.array_equal(img[32, 32], np.array([0.25 + 0.5 * 0.5, 0.5 + 0.25 * 0.5, 0.75 + 1.0 * 0.5, 0.5 + 1.0 * 0.5], np.float32))
    assert np.array_equal(img[0, 0], np.array([0.5, 0.25, 1.0, 1.0], np.float32)), "uncovered pixel: the background, untouched"
