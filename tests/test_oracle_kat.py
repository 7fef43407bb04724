"""Analytic known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors ("parity unpinned"), so these closed-form cases derived from the
shader math are what anchors the oracle before the HIP kernels are compared against it.
"""
import ctypes as C
import math

import numpy as np
import pytest

from multivolumes_amd import _lib
from multivolumes_amd import raycaster as rc
from tests.conftest import half_to_float
from tests.helpers import default_descs

F = np.float32
TWO_PI = F(2.0) * F(3.1415926535897)
# min16float constants as the reference's compiler folded them (half precision; oracle/dxil reads them out of the shipped shaders)
MAX_DIST = F(3.46484375)        # half(2 sqrt 3) = 0x42EE
ABSORPTION = F(0.7998046875)    # half(0.8) = 0x3A66
ZERO_THRESHOLD = F(0.01000213623046875)  # half(0.01) = 0x211F


def unit_scene(V=1, eye=(0.0, 0.0, -5.0), W=64, H=64, frame_idx=0, size=2.0, pos=(0.0, 0.0, 0.0)):
    """One unit cube at the origin (local == world), camera on `eye` looking at the origin."""
    lib = _lib.load()
    worlds = np.zeros((V, 12), np.float32)
    for i in range(V):
        lib.mvh_set_volume_world(F(size), rc._ptr(rc._f32(pos, 3)), rc._ptr(worlds[i]))
    view = rc.look_at_lh(eye, (0, 0, 0), (0, 1, 0))
    proj = rc.perspective_fov_lh(math.pi / 4, W / H, 1.0, 1000.0)
    vp = rc.multiply(view, proj)
    svp = rc.shadow_view_proj((75, 75, -75), 1.0)
    cb = np.zeros(212, np.uint8)
    po = np.zeros(V * 224, np.uint8)
    one4 = np.ones(4, np.float32)
    lib.mvh_update_frame(V, rc._ptr(worlds), rc._ptr(rc._f32(vp, 16)), rc._ptr(rc._f32(svp, 16)), rc._ptr(rc._f32(eye, 3)),
                         float(W), float(H), rc._ptr(rc._f32((75, 75, -75), 3)), rc._ptr(one4), rc._ptr(one4), frame_idx,
                         rc._ptr(cb), rc._ptr(po))
    return vp, cb, po


def homogeneous_volume(N, c, a):
    v = np.empty((N, N, N, 4), np.float16)
    v[..., :3] = c
    v[..., 3] = a
    return v.view(np.uint16)


def get_step(dD, transm, density, step):
    with np.errstate(divide="ignore"):
        ev = min(F(1.0) / F(256.0) / abs(F(dD)), F(2.0))
    ui = min(F(1.0) - F(density), F(1.0))
    th = F(1.0) - F(transm)
    return F(step) * max(F(1.5) * ev * ui * th, F(1.0))


def emulate_march(alpha, c, light, smp, t_max, origin, direction):
    """float32 emulation of CSRayMarch.hlsl:111-155 for a homogeneous volume; returns (rgb, A, iterations)."""
    step_scale = MAX_DIST / F(smp)
    A = F(0.0)
    rgb = F(0.0)
    t = F(0.0)
    prev = F(0.0)
    it = 0
    o = np.asarray(origin, F)
    d = np.asarray(direction, F)
    for _ in range(smp):
        pos = (d * F(t)).astype(F) + o  # a multiplication and an addition, as compiled
        if np.any(np.abs(pos) > 1.0):
            break
        it += 1
        new_step = step_scale
        if alpha > ZERO_THRESHOLD:
            transm = F(1.0) - A
            new_step = get_step(F(alpha) - prev, transm, alpha, step_scale)
            prev = F(alpha)
            rgb = rgb + F(c) * F(alpha) * F(light) * ABSORPTION * transm
            A = A + F(alpha) * ABSORPTION * transm
            if transm < ZERO_THRESHOLD:
                break
        t = t + new_step
        if t > t_max:
            break
    return rgb / TWO_PI, A, it


def test_kat1_homogeneous_volume_recurrence(oracle):
    """pins GetStep, the accumulate order, early termination before the t update, and the /2pi."""
    N, L, smp = 8, 4, 16
    alpha, c = F(np.float16(0.3)), F(np.float16(0.5))
    vol = homogeneous_volume(N, c, alpha)
    lm = np.full((1, L, L, L), oracle.load().mvo_pack_r11g11b10f(1.0, 1.0, 1.0), np.uint32)
    vp, cb, po = unit_scene()
    depth = np.ones((64, 64), np.float32)
    info = np.array([[0, smp, 1 << 4, 0]], np.uint16)  # mip 0, face +Z only
    cm, cd, st = oracle.march_view(1, N, L, 64, 64, [vol], lm, depth, po, cb, info, [0])
    face = half_to_float(oracle.cube_face_view(cm[0], N, 0))[4]
    assert st["rays"] == N * N
    lib = oracle.load()
    tot_it = 0
    for ty in range(N):
        for tx in range(N):
            tgt = np.zeros(3, np.float32)
            lib.mvo_get_local_pos(F(tx), F(ty), 4, N, rc._ptr(tgt))
            o = np.array([0, 0, -5], np.float32)
            d = tgt - o
            d = (d / np.sqrt(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])).astype(F)
            hit = lib.mvo_compute_ray_origin(rc._ptr(o), rc._ptr(d))
            assert hit == 1 and abs(o[2] + 1.0) < 1e-6
            t_max = lib.mvo_compute_target_hit(rc._ptr(o), rc._ptr(tgt), rc._ptr(d))
            rgb, A, it = emulate_march(alpha, c, 1.0, smp, t_max, o, d)
            tot_it += it
            assert face[ty, tx, 3] == pytest.approx(float(A), abs=1e-3)
            assert face[ty, tx, 0] == pytest.approx(float(rgb), abs=1e-3)
            assert face[ty, tx, 0] == face[ty, tx, 1] == face[ty, tx, 2]
    assert st["view_samples"] == tot_it
    assert st["lit_samples"] == tot_it
    # closed form on the axis: A_{k+1} = A_k + 0.8 a (1 - A_k)  =>  A_n = 1 - (1 - 0.8 a)^n
    n = round(math.log(1 - float(face[N // 2, N // 2, 3])) / math.log(1 - float(ABSORPTION) * float(alpha)))
    assert face[N // 2, N // 2, 3] == pytest.approx(1 - (1 - float(ABSORPTION) * float(alpha)) ** n, abs=2e-3)


def test_kat2_empty_volume(oracle):
    """alpha = 0 gives exactly 0; iterations = steps until box exit / tMax / SmpCount."""
    N, L, smp = 8, 4, 24
    vol = homogeneous_volume(N, 1.0, 0.0)
    lm = np.zeros((1, L, L, L), np.uint32)
    vp, cb, po = unit_scene()
    depth = np.ones((64, 64), np.float32)
    info = np.array([[0, smp, 1 << 4, 0]], np.uint16)
    cm, cd, st = oracle.march_view(1, N, L, 64, 64, [vol], lm, depth, po, cb, info, [0])
    assert not cm.any()
    assert st["lit_samples"] == 0 and st["rays"] == N * N
    # on-axis-ish ray: t_k = k * stepScale; iteration k runs while t_k <= tMax (~2/d.z) and inside the box
    step = float(MAX_DIST / F(smp))
    per_ray_min = math.floor(2.0 / step) + 1
    assert N * N * per_ray_min <= st["view_samples"] <= N * N * (per_ray_min + 1)
    # depth written for every ray = the depth map value
    assert np.all(oracle.cube_face_view(cd[0], N, 0)[4] == 1.0)


def test_kat3_get_local_pos(oracle):
    lib = oracle.load()
    out = np.zeros(3, np.float32)
    axes = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    for f, ax in enumerate(axes):
        lib.mvo_get_local_pos(0.0, 0.0, f, 1, rc._ptr(out))
        assert tuple(out) == ax
    lib.mvo_get_local_pos(0.0, 0.0, 0, 2, rc._ptr(out))
    assert tuple(out) == (1.0, 0.5, 0.5)
    # the six tables of CSRayMarch.hlsl:36-52 at a generic texel: u = -0.75, v -> +0.25 after the flip
    lib.mvo_get_local_pos(0.0, 1.0, 0, 4, rc._ptr(out))
    u, v = -0.75, 0.25
    expect = [(1, v, -u), (-1, v, u), (u, 1, -v), (u, -1, v), (u, v, 1), (-u, v, -1)]
    for f, e in enumerate(expect):
        lib.mvo_get_local_pos(0.0, 1.0, f, 4, rc._ptr(out))
        assert tuple(out) == tuple(np.float32(x) for x in e)


def test_kat4_compute_ray_origin(oracle):
    lib = oracle.load()
    o = np.array([0, 0, -5], np.float32)
    assert lib.mvo_compute_ray_origin(rc._ptr(o), rc._ptr(np.array([0, 0, 1], np.float32))) == 1
    assert tuple(o) == (0.0, 0.0, -1.0)
    o = np.array([0, 0, -5], np.float32)
    assert lib.mvo_compute_ray_origin(rc._ptr(o), rc._ptr(np.array([1, 0, 0], np.float32))) == 0
    o = np.array([0.25, -0.5, 0.75], np.float32)
    assert lib.mvo_compute_ray_origin(rc._ptr(o), rc._ptr(np.array([0, 1, 0], np.float32))) == 1
    assert tuple(o) == (0.25, -0.5, 0.75)
    # diagonal entry through the -X face; result clamped onto the surface
    o = np.array([-3, 0.5, 0.25], np.float32)
    d = np.array([1, 0, 0], np.float32)
    assert lib.mvo_compute_ray_origin(rc._ptr(o), rc._ptr(d)) == 1
    assert tuple(o) == (-1.0, 0.5, 0.25)
    # pointing away
    o = np.array([0, 0, -5], np.float32)
    assert lib.mvo_compute_ray_origin(rc._ptr(o), rc._ptr(np.array([0, 0, -1], np.float32))) == 0
    # ComputeTargetHit: max over components, NaN (0/0) ignored
    t = lib.mvo_compute_target_hit(rc._ptr(np.array([0, 0, -1], np.float32)), rc._ptr(np.array([0, 0, 1], np.float32)),
                                   rc._ptr(np.array([0, 0, 1], np.float32)))
    assert t == 2.0


def test_kat5_depth_clamp(oracle):
    """a planar depth in front of the box exit shortens the march; z >= 1 leaves tMax alone."""
    N, L, smp = 8, 4, 64
    vol = homogeneous_volume(N, 1.0, 0.0)
    lm = np.zeros((1, L, L, L), np.uint32)
    W = H = 64
    vp, cb, po = unit_scene(W=W, H=H)
    info = np.array([[0, smp, 1 << 4, 0]], np.uint16)
    far = np.ones((H, W), np.float32)
    _, _, st_far = oracle.march_view(1, N, L, W, H, [vol], lm, far, po, cb, info, [0])
    # plane at view distance 5 (through the cube centre): z_ndc = zf (d - zn) / (d (zf - zn))
    d = 5.0
    z = 1000.0 * (d - 1.0) / (d * 999.0)
    near = np.full((H, W), z, np.float32)
    cm, cd, st_near = oracle.march_view(1, N, L, W, H, [vol], lm, near, po, cb, info, [0])
    assert np.all(oracle.cube_face_view(cd[0], N, 0)[4] == np.float32(z))
    # rays run from z=-1 to z~0 instead of z=+1: about half the samples (+1 for the t > tMax test after the step)
    step = float(MAX_DIST / F(smp))
    lo = N * N * (math.floor(1.0 / step) + 1)
    assert lo <= st_near["view_samples"] <= lo + 2 * N * N
    assert st_far["view_samples"] > 1.8 * st_near["view_samples"]
    # a plane behind the cube (d = 7) must not change anything
    z7 = 1000.0 * (7.0 - 1.0) / (7.0 * 999.0)
    _, _, st_b = oracle.march_view(1, N, L, W, H, [vol], lm, np.full((H, W), z7, np.float32), po, cb, info, [0])
    assert st_b["view_samples"] == st_far["view_samples"]


def test_kat6_cull(oracle):
    """face mask rule, off-screen volumes, SmpCount / MipLevel from the projected edge length."""
    N = 256
    W = H = 1024
    # eye on the -Z axis at local (0,0,-8): faces with eye<1 (even) / eye>-1 (odd) = +X,-X,+Y,-Y,+Z -> 0x1F
    vp, cb, po = unit_scene(eye=(0, 0, -8), W=W, H=H)
    descs = default_descs(1, 1, N)
    info, vis, cube = oracle.cull(1, po, descs, cb, 128)
    assert list(vis) == [0]
    assert info[0, 2] & 0x3F == 0x1F
    # projected edge length: the near face (z=-1, distance 7) has edge 2 / 7 / tan(pi/8) * H/2 px
    edge = 2.0 / 7.0 / math.tan(math.pi / 8) * H / 2
    amt = 2 * (edge / 2) / math.sqrt(3)
    smp = min(math.ceil(amt), 128)
    s2 = min(amt, smp) / 2 * math.sqrt(3)
    mip = min(int(max(math.log2(N / s2), 0)), 4)
    assert info[0, 1] == smp and info[0, 0] == mip and info[0, 3] == 0
    # cube-map scheme bit: cubeMapPix = 5 (N >> mip)^2 vs projected area of the 5 visible faces
    # (near face edge^2 ... here the far face is what +Z shows; side faces are thin trapezoids)
    assert bool(info[0, 2] & 0x8000) == (5 * (N >> mip) ** 2 <= _proj_cov(vp, W, H))
    # a volume far off to the side: no vertex inside the viewport -> not appended, VolumeInfo untouched
    vp2, cb2, po2 = unit_scene(eye=(0, 0, -8), W=W, H=H, pos=(40.0, 0.0, 0.0))
    sentinel = np.full((1, 4), 0xABCD, np.uint16)
    info2, vis2, cube2 = oracle.cull(1, po2, descs, cb2, 128, sentinel)
    assert len(vis2) == 0 and len(cube2) == 0 and np.all(info2 == 0xABCD)
    # behind the camera
    vp3, cb3, po3 = unit_scene(eye=(0, 0, -8), W=W, H=H, pos=(0.0, 0.0, -30.0))
    info3, vis3, _ = oracle.cull(1, po3, descs, cb3, 128, sentinel)
    assert len(vis3) == 0
    # eye inside the box: all six interior faces visible
    vp4, cb4, po4 = unit_scene(eye=(0.2, 0.1, -0.3), W=W, H=H)
    info4, vis4, _ = oracle.cull(1, po4, descs, cb4, 128)
    assert (len(vis4) == 0) or (info4[0, 2] & 0x3F == 0x3F)


def _proj_cov(vp, W, H):
    """float64 restatement of EstimateProjCoverage for the unit cube seen through vp (faces 0-4 visible)."""
    vp = np.asarray(vp, np.float64)
    verts = []
    for i in range(8):
        p = np.array([1.0 if i & 1 else -1.0, 1.0 if (i >> 1) & 1 else -1.0, 1.0 if i >> 2 else -1.0, 1.0])
        q = p @ vp
        q = q[:3] / q[3]
        verts.append(((q[0] * 0.5 + 0.5) * W, (1 - (q[1] * 0.5 + 0.5)) * H))
    lanes = [(0, 1), (3, 2), (1, 3), (2, 0), (6, 7), (5, 4), (4, 6), (7, 5), (4, 0), (2, 6), (7, 3), (1, 5)]
    faces = [(8, 3, 9, 6), (10, 2, 11, 7), (0, 8, 5, 11), (1, 10, 4, 9), (0, 2, 1, 3), (4, 6, 5, 7)]
    e = [np.subtract(verts[b], verts[a]) for a, b in lanes]
    cov = 0.0
    for f in range(5):
        q = [e[k] for k in faces[f]]
        cov += 0.5 * abs(q[0][0] * q[1][1] - q[0][1] * q[1][0]) + 0.5 * abs(q[2][0] * q[3][1] - q[2][1] * q[3][0])
    return cov


def test_kat7_trilinear_sampler(oracle):
    lib = oracle.load()
    N = 5
    rng = np.random.default_rng(7)
    vol = rng.random((N, N, N, 4), np.float32).astype(np.float16)
    out = np.zeros(4, np.float32)
    # texel centres reproduce texels exactly
    for (x, y, z) in [(0, 0, 0), (4, 4, 4), (1, 2, 3), (3, 0, 2)]:
        uvw = np.array([(x + 0.5) / N, (y + 0.5) / N, (z + 0.5) / N], np.float32)
        lib.mvo_sample_rgba16f(rc._ptr(vol.view(np.uint16)), N, rc._ptr(uvw), None, rc._ptr(out))
        assert np.allclose(out, vol[z, y, x].astype(np.float32), atol=1e-6)
    # clamp at 0 and 1
    lib.mvo_sample_rgba16f(rc._ptr(vol.view(np.uint16)), N, rc._ptr(np.zeros(3, np.float32)), None, rc._ptr(out))
    assert np.array_equal(out, vol[0, 0, 0].astype(np.float32))
    lib.mvo_sample_rgba16f(rc._ptr(vol.view(np.uint16)), N, rc._ptr(np.ones(3, np.float32)), None, rc._ptr(out))
    assert np.array_equal(out, vol[4, 4, 4].astype(np.float32))
    # a linear ramp along x is reproduced between the first and last texel centres
    ramp = np.zeros((N, N, N, 4), np.float16)
    ramp[..., 3] = (np.arange(N, dtype=np.float32) * 0.125)[None, None, :]
    for u in np.linspace(0.5 / N, 1 - 0.5 / N, 17):
        uvw = np.array([u, 0.3, 0.6], np.float32)
        lib.mvo_sample_rgba16f(rc._ptr(ramp.view(np.uint16)), N, rc._ptr(uvw), None, rc._ptr(out))
        assert out[3] == pytest.approx((u * N - 0.5) * 0.125, abs=1e-6)
    # integer texel offsets (gradient taps) shift the footprint and clamp at the border
    off = np.array([1, 0, 0], np.int32)
    uvw = np.array([(1 + 0.5) / N, 0.5 / N, 0.5 / N], np.float32)
    lib.mvo_sample_rgba16f(rc._ptr(ramp.view(np.uint16)), N, rc._ptr(uvw), rc._ptr(off), rc._ptr(out))
    assert out[3] == pytest.approx(2 * 0.125, abs=1e-6)
    uvw = np.array([(4 + 0.5) / N, 0.5 / N, 0.5 / N], np.float32)
    lib.mvo_sample_rgba16f(rc._ptr(ramp.view(np.uint16)), N, rc._ptr(uvw), rc._ptr(off), rc._ptr(out))
    assert out[3] == pytest.approx(4 * 0.125, abs=1e-6)
    # point sampler + comparison sampler
    d = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert lib.mvo_sample_point_r32f(rc._ptr(d), 4, 3, 0.49, 0.34) == d[1, 1]
    assert lib.mvo_sample_point_r32f(rc._ptr(d), 4, 3, -3.0, 9.0) == d[2, 0]
    sm = np.array([[0, 65535], [65535, 65535]], np.uint16)
    assert lib.mvo_sample_cmp_d16(rc._ptr(sm), 2, 0.25, 0.25, 0.5) == 0.0  # centre of the near texel: occluded
    assert lib.mvo_sample_cmp_d16(rc._ptr(sm), 2, 0.75, 0.75, 0.5) == 1.0
    assert lib.mvo_sample_cmp_d16(rc._ptr(sm), 2, 0.5, 0.25, 0.5) == pytest.approx(0.5)
    assert lib.mvo_sample_cmp_d16(rc._ptr(sm), 2, 0.5, 0.5, 0.5) == pytest.approx(0.75)


def test_kat8_cast_light_ray(oracle):
    """homogeneous alpha: transm = (1 - 0.8 a)^k with k from the step schedule."""
    lib = oracle.load()
    N = 8
    alpha = F(np.float16(0.05))
    vol = homogeneous_volume(N, 1.0, alpha)
    n_samples = 32
    step_scale = MAX_DIST / F(n_samples)
    o = np.array([0, 0, -1], np.float32)
    d = np.array([0, 0, 1], np.float32)
    transm = C.c_float(1.0)
    cnt = C.c_uint64(0)
    lib.mvo_cast_light_ray(C.byref(transm), rc._ptr(vol), N, rc._ptr(o), rc._ptr(d), step_scale, n_samples, C.byref(cnt))
    # emulate the schedule (RayMarch.hlsli:203-228)
    t, step, prev, tr, k = step_scale, step_scale, F(0), F(1), 0
    for _ in range(n_samples):
        if abs(F(-1) + t) > 1:
            break
        k += 1
        opacity = min(max(alpha * step, F(0)), F(1))
        new_step = get_step(alpha - prev, tr, opacity, step_scale)
        prev = alpha
        tr = tr * (F(1) - alpha * ABSORPTION)
        if tr < ZERO_THRESHOLD:
            break
        step = new_step
        t = t + step
    assert cnt.value == k
    assert transm.value == pytest.approx((1 - float(ABSORPTION) * float(alpha)) ** k, rel=1e-5)
    assert transm.value == pytest.approx(float(tr), rel=1e-6)


def test_kat9_formats(oracle):
    lib = oracle.load()
    # every fp16 bit pattern decodes like numpy's IEEE half
    bits = np.arange(65536, dtype=np.uint16)
    mine = np.array([lib.mvo_f16_to_f32(int(b)) for b in bits], np.float32)
    ref = bits.view(np.float16).astype(np.float32)
    assert np.array_equal(mine.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    assert np.all(np.isnan(mine[np.isnan(ref)]))
    # encode: RNE like numpy, incl. denormals / overflow / ties
    rng = np.random.default_rng(9)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, 6.0975e-5,
                  1.0009765625, 1.00048828125, 1.00146484375, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        ref16 = vals.astype(np.float16).view(np.uint16)
    mine16 = np.array([lib.mvo_f32_to_f16(float(v)) for v in vals], np.uint16)
    assert np.array_equal(mine16, ref16)
    # R11G11B10F: exact round trips for representable values, negative -> 0, RNE, unpack = shifted half
    out = np.zeros(3, np.float32)
    for r, g, b in [(1.0, 1.0, 1.0), (0.0, 0.5, 2.0), (65024.0, 64512.0, 64512.0), (1.015625, 1.03125, 1.03125)]:
        lib.mvo_unpack_r11g11b10f(lib.mvo_pack_r11g11b10f(r, g, b), rc._ptr(out))
        assert tuple(out) == (r, g, b)
    lib.mvo_unpack_r11g11b10f(lib.mvo_pack_r11g11b10f(-1.0, -0.0, -1e-30), rc._ptr(out))
    assert tuple(out) == (0.0, 0.0, 0.0)
    # 6-bit mantissa: 1 + 1/128 is a tie between 1 and 1 + 1/64 -> even (1.0); 1 + 3/128 -> 1 + 2/64
    lib.mvo_unpack_r11g11b10f(lib.mvo_pack_r11g11b10f(1.0078125, 1.0234375, 1.015625), rc._ptr(out))
    assert tuple(out) == (1.0, 1.03125, 1.0)
    # overflow -> inf, max finite 11-bit = 65024, 10-bit = 64512
    lib.mvo_unpack_r11g11b10f(lib.mvo_pack_r11g11b10f(1e6, 65024.0, 64512.0), rc._ptr(out))
    assert out[0] == np.inf and out[1] == 65024.0 and out[2] == 64512.0
    # denormals: smallest 11-bit denormal = 2^-20, 10-bit = 2^-19
    lib.mvo_unpack_r11g11b10f(lib.mvo_pack_r11g11b10f(2.0 ** -20, 2.0 ** -21, 2.0 ** -19), rc._ptr(out))
    assert tuple(out) == (2.0 ** -20, 0.0, 2.0 ** -19)
    # bit layout: R in bits 0-10, G 11-21, B 22-31
    assert lib.mvo_pack_r11g11b10f(1.0, 0.0, 0.0) == 15 << 6
    assert lib.mvo_pack_r11g11b10f(0.0, 1.0, 0.0) == (15 << 6) << 11
    assert lib.mvo_pack_r11g11b10f(0.0, 0.0, 1.0) == (15 << 5) << 22


def test_kat10_init_grid_data(oracle):
    """the procedural field of CSInitGridData.hlsl:10-27."""
    N = 16
    v = half_to_float(oracle.init_grid_data(N))
    c = (np.arange(N, dtype=np.float32) + F(0.5)) / F(N) * F(2) - F(1)
    z, y, x = np.meshgrid(c, c, c, indexing="ij")
    r2 = x * x + y * y + z * z
    a = (F(1) - r2) ** 2
    a = np.clip(a * a * F(2), 0, 1).astype(np.float32)
    assert np.array_equal(v[..., 3], a.astype(np.float16).astype(np.float32))
    assert v[N // 2, N // 2, N // 2, 3] == 1.0  # centre
    assert v[0, 0, 0, 3] == 1.0  # the corner rise (|p|^2 > 1)
    shell = np.abs(r2 - 1.0) < 0.02
    assert shell.any() and np.all(v[..., 3][shell] < 2e-6)
    s = np.clip(y * F(0.5) + F(0.2), 0, 1)
    for ch, (d0, u0) in enumerate([(0.5, 1.0), (0.8, 0.6), (1.0, 0.0)]):
        expect = (F(d0) + s * (F(u0) - F(d0))).astype(np.float16).astype(np.float32)
        assert np.max(np.abs(v[..., ch] - expect)) <= 1e-3
    low = y <= -0.4
    assert np.allclose(v[..., 0][low], 0.5) and np.allclose(v[..., 2][low], 1.0)  # colorD below p.y = -0.4
    # CSR32FToRGBA16F: (1, 1, 1, 0.25 * linear(src))
    src = np.linspace(0, 1, 4 * 4 * 4, dtype=np.float32).reshape(4, 4, 4)
    r = half_to_float(oracle.r32f_to_rgba16f(src, 4))
    assert np.all(r[..., :3] == 1.0)
    assert np.array_equal(r[..., 3], (src * F(0.25)).astype(np.float16).astype(np.float32))
    up = half_to_float(oracle.r32f_to_rgba16f(src, 8))
    assert up[0, 0, 0, 3] == np.float16(src[0, 0, 0] * 0.25) and up[7, 7, 7, 3] == np.float16(src[3, 3, 3] * 0.25)


def test_sh_irradiance(oracle):
    """EvaluateSHIrradiance (SHIrradianceTypeless.hlsli:15-36): constant term and a first-order lobe."""
    lib = oracle.load()
    sh = np.zeros(27, np.float32)
    sh[0:3] = (1.0, 2.0, 3.0)
    out = np.zeros(3, np.float32)
    lib.mvo_eval_sh_irradiance(rc._ptr(sh), rc._ptr(np.array([0, 0, 1], np.float32)), rc._ptr(out))
    assert np.allclose(out, 0.886226925 * np.array([1, 2, 3]), rtol=1e-6)
    sh[:] = 0
    sh[3 * 2:3 * 2 + 3] = 1.0  # L10 (z)
    lib.mvo_eval_sh_irradiance(rc._ptr(sh), rc._ptr(np.array([0, 0, 1], np.float32)), rc._ptr(out))
    assert np.allclose(out, 2 * 0.511663354, rtol=1e-6)
    lib.mvo_eval_sh_irradiance(rc._ptr(sh), rc._ptr(np.array([0, 0, -1], np.float32)), rc._ptr(out))
    assert np.all(out == 0.0)  # max(0, .)
