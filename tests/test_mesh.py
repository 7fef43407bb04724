"""Mesh depth / shadow producer: OBJ loader, oracle rasteriser sanity (CPU), HIP rasteriser parity (GPU)."""
import os

import numpy as np
import pytest

from multivolumes_amd import scene
from multivolumes_amd import raycaster as rc

BUNNY = "/root/reference/Bin/Assets/bunny.obj"  # present in the authoring container only; never on the GPU box


def icosphere(radius=8.0, subdiv=3):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return (np.array(v) * radius).astype(np.float32), np.array(f, np.uint32)


def test_obj_loader(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# comment\nmtllib x.mtl\nv 0 0 0\nv 1 0 0\nv 1 2 0\nv 0 2 0\nvn 0 0 1\nvt 0 0\n"
                 "f 1/1/1 2/1/1 3/1/1 4/1/1\nv 0 0 3\nf -1 -5 -4\nf 1//1 3//1 5//1\n")
    pos, idx, lo, hi = rc.load_obj(p)
    assert pos.shape == (5, 3) and idx.tolist() == [[0, 1, 2], [0, 2, 3], [4, 0, 1], [0, 2, 4]]
    assert lo.tolist() == [0, 0, 0] and hi.tolist() == [1, 2, 3]
    with pytest.raises(IOError):
        rc.load_obj(tmp_path / "missing.obj")
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(IOError):
        rc.load_obj(tmp_path / "bad.obj")


@pytest.mark.skipif(not os.path.exists(BUNNY), reason="reference asset not present")
def test_obj_loader_on_the_reference_mesh():
    pos, idx, lo, hi = rc.load_obj(BUNNY)
    assert pos.shape == (34835, 3) and idx.shape == (69666, 3)  # 34 835 v / 69 666 f
    assert np.allclose(lo, [-5.015, -0.044, -3.887], atol=2e-3) and np.allclose(hi, [5.015, 9.898, 3.887], atol=2e-3)
    assert abs(max(hi - lo) - 10.03) < 1e-2  # the sceneSize the shadow projection is built from (ObjectRenderer.cpp:76)


def mesh_matrices(cfg, scale=1.0, pos=(0.0, 0.0, 0.0)):
    eye, vp = scene.camera(cfg)
    world = rc.mesh_world(scale, pos)
    wvp = rc.multiply(world, vp)
    svp_t = scene.shadow_vp(dict(cfg, mesh=True))  # transposed light view-projection
    shadow_wvp = rc.multiply(world, svp_t.T.copy())
    return eye, vp, wvp, shadow_wvp


def test_oracle_rasteriser_matches_the_analytic_sphere(oracle):
    """A finely tessellated sphere rasterised by the oracle = the analytic sphere depth, up to the tessellation error."""
    cfg = dict(scene.CONFIGS["T1"], W=160, H=120)
    pos, idx = icosphere(8.0, 4)
    eye, vp, wvp, shadow_wvp = mesh_matrices(cfg)
    d = oracle.raster_depth(pos, idx, rc.transpose_store(wvp), cfg["W"], cfg["H"])
    ref = rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), 8.0)
    both = (d < 1) & (ref < 1)
    assert both.sum() > 150 and ((d < 1) ^ (ref < 1)).sum() <= 0.15 * both.sum()  # silhouettes agree up to the facets
    # depth is z/w with zn = 1, zf = 1000: compare view distances instead (d = zf zn / (zf - z (zf - zn)))
    lin = lambda z: 1000.0 / (1000.0 - z.astype(np.float64) * 999.0)
    assert np.max(np.abs(lin(d[both]) - lin(ref[both]))) < 0.35
    # top-left rule / shared edges: a closed mesh covers every interior pixel exactly once -> no 1.0 holes inside
    inner = rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), 7.5) < 1
    assert np.all(d[inner] < 1)
    # the jitter of VSBasePass.hlsl:57 moves the image by bias * W / 2 pixels
    dj = oracle.raster_depth(pos, idx, rc.transpose_store(wvp), cfg["W"], cfg["H"], bias=(4.0 / cfg["W"], 0.0))
    assert np.array_equal((dj < 1)[:, 2:], (d < 1)[:, :-2])


@pytest.mark.gpu
def test_gpu_rasteriser_parity_and_use(oracle):
    from multivolumes_amd import MultiRayCaster
    cfg = dict(scene.CONFIGS["T1"], W=320, H=200)
    M = 256
    pos, idx = icosphere(8.0, 3)
    # a second, off-centre, partly off-screen object and a degenerate triangle
    pos2 = pos * 0.4 + np.array([30.0, 5.0, -20.0], np.float32)
    allpos = np.concatenate([pos, pos2, np.zeros((3, 3), np.float32)])
    allidx = np.concatenate([idx, idx + len(pos), [[len(pos) * 2, len(pos) * 2 + 1, len(pos) * 2 + 2]]]).astype(np.uint32)
    eye, vp, wvp, shadow_wvp = mesh_matrices(cfg, scale=1.25, pos=(0.0, -1.0, 0.0))
    bias = ((2 * 0.3 - 1) / cfg["W"], (2 * 0.7 - 1) / cfg["H"])
    exp_d = oracle.raster_depth(allpos, allidx, rc.transpose_store(wvp), cfg["W"], cfg["H"], bias)
    exp_s = oracle.depth_to_d16(oracle.raster_depth(allpos, allidx, rc.transpose_store(shadow_wvp), M, M))
    with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=M) as c:
        c.SetMesh(allpos, allidx)
        c.RenderMeshDepth(wvp, bias)
        c.RenderMeshShadow(shadow_wvp)
        got_d, got_s = c.ReadBackDepthMap(), c.ReadBackShadowMap()
        assert (exp_d < 1).sum() > 1000 and (exp_s < 65535).sum() > 1000
        assert np.array_equal(got_d, exp_d), f"{(got_d != exp_d).sum()} depth pixels differ"
        assert np.array_equal(got_s, exp_s)
        # the rasterised maps feed the path like uploaded ones: same frame as uploading the oracle's maps
        vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
        fr = scene.setup_raycaster(c, cfg, vols)  # uploads analytic maps ...
        c.RenderMeshDepth(wvp, bias)              # ... replaced by the rasterised ones
        c.RenderMeshShadow(shadow_wvp)
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render(composite=True)
        a = (c.ReadBackComposite(), c.ReadBackLightMap(0))
    with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=M) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        c.SetRenderTargets(exp_d, exp_s)
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render(composite=True)
        b = (c.ReadBackComposite(), c.ReadBackLightMap(0))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("MV_RASTER_FUZZ", "6"))))
def test_gpu_rasteriser_equals_the_oracle_on_random_triangle_soups(oracle, seed):
    """Round 6 fuzz of the depth / shadow producer (csrc/mv_mesh.hip against mvo_raster_depth): triangle soups no modelling tool would make -- slivers, triangles
    larger than the screen, triangles that cross the screen's edges, the eye plane and the far plane, duplicates at equal depth, zero-area triangles, vertices on
    pixel centres -- at odd viewports, with and without the jitter.  Depth and D16 shadow maps bit for bit."""
    from multivolumes_amd import MultiRayCaster
    r = np.random.default_rng(500 + seed)
    cfg = dict(scene.CONFIGS["T1"], W=int(r.integers(33, 400)), H=int(r.integers(17, 300)))
    M = int(r.choice([64, 100, 256]))
    eye, vp, wvp, shadow_wvp = mesh_matrices(cfg, scale=float(r.uniform(0.5, 2.0)), pos=(float(r.uniform(-5, 5)), float(r.uniform(-5, 5)), float(r.uniform(-5, 5))))
    n = int(r.integers(50, 600))
    centre = r.normal(0.0, 25.0, size=(n, 1, 3))
    size = np.exp(r.uniform(np.log(0.01), np.log(80.0), size=(n, 1, 1)))
    tri = (centre + r.normal(0.0, 1.0, size=(n, 3, 3)) * size).astype(np.float32)
    k = n // 10
    tri[:k, 2] = tri[:k, 1]                                   # zero area
    tri[k:2 * k] = tri[2 * k:3 * k]                           # duplicates (equal depth: atomicMin either way)
    tri[3 * k:4 * k, :, :] += np.array(eye, np.float32)       # around and behind the eye
    tri[4 * k:5 * k] = np.round(tri[4 * k:5 * k])             # lattice positions
    pos = tri.reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    bias = (0.0, 0.0) if seed % 2 else ((2 * float(r.random()) - 1) / cfg["W"], (2 * float(r.random()) - 1) / cfg["H"])
    exp_d = oracle.raster_depth(pos, idx, rc.transpose_store(wvp), cfg["W"], cfg["H"], bias)
    exp_s = oracle.depth_to_d16(oracle.raster_depth(pos, idx, rc.transpose_store(shadow_wvp), M, M))
    with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=M) as c:
        c.SetMesh(pos, idx)
        c.RenderMeshDepth(wvp, bias)
        c.RenderMeshShadow(shadow_wvp)
        got_d, got_s = c.ReadBackDepthMap(), c.ReadBackShadowMap()
    assert (exp_d < 1).sum() > 50, "the soup must cover something"
    assert np.array_equal(got_d, exp_d), f"seed {seed}: {(got_d != exp_d).sum()} depth pixels differ"
    assert np.array_equal(got_s, exp_s), f"seed {seed}: {(got_s != exp_s).sum()} shadow texels differ"
