#!/usr/bin/env python3
"""How far is the analytic fragment stand-in (oracle: mvo_pixel_fragments, DESIGN.md section 2 "contract" items 1-2) from what a D3D
rasteriser hands the reference's pixel shaders?  Runs the small frame of tests/golden/make_dxil_golden.py twice through
PSDepthPeel / PSCube / PSResolveOIT .cso: once with the analytic fragments (ray through the pixel centre against the cube's exit
face, what the oracle and the HIP composite use), once with fragments from an emulated rasteriser fed by VSCube.cso's vertices
(oracle/dxref.py rasterised_fragments: CULL_FRONT, 1/256 snapping, top-left rule, perspective-correct interpolation in float64).
Reports the pixels whose fragment sets differ (silhouette coverage) and the image difference over the others.
Needs oracle/_ref (make -C oracle ref).  Test infrastructure only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import make_dxil_golden as G  # noqa: E402
from multivolumes_amd import scene  # noqa: E402
from multivolumes_amd import raycaster as rc  # noqa: E402
from oracle import dxref, mv_oracle as o  # noqa: E402
from tests.conftest import half_to_float  # noqa: E402
from tests.helpers import default_descs, host_frame  # noqa: E402


def main():
    assert dxref.available(), "oracle/_ref/libmv_dxref.so missing: make -C oracle ref"
    o.load()
    c = G.CFG
    V, N, L, W, H = c["V"], c["N"], c["L"], c["W"], c["H"]
    vols = [rc.synth_volume(N, k + 1) for k in range(V)]
    descs = default_descs(V, V, N)
    _, vp, svp, cb, po = host_frame(c, 0, G.EYE)
    depth, shadow = scene.depth_and_shadow(c, vp, svp, G.M)
    info, vis, cube = o.cull(V, po, descs, cb, c["maxRay"])
    lms = np.zeros((V, L, L, L), np.uint32)
    for f in range(V):
        _, _, _, cbf, pof = host_frame(c, f, G.EYE)
        lms = o.march_light(V, N, L, G.M, vols, descs, pof, cbf, shadow, None, vis, c["maxLight"], lightMaps=lms)[1]
    cm, cd = o.march_view(V, N, L, W, H, vols, lms, depth, po, cb, info, cube)[:2]
    vis = np.sort(vis)
    analytic = dxref.composite(V, N, L, W, H, vols, lms, depth, po, cb, info, vis, cm, cd)
    bits = int(os.environ.get("RASTER_SUBPIXEL_BITS", "8"))
    frags = dxref.rasterised_fragments(V, W, H, po, info, vis, subpixel_bits=bits)
    raster = dxref.composite(V, N, L, W, H, vols, lms, depth, po, cb, info, vis, cm, cd, fragments=frags)

    # per-pixel fragment sets of both
    cov_r = {}
    for f in frags:
        cov_r.setdefault((f[0], f[1]), []).append(f)
    same, differ, dz, dl = 0, 0, 0.0, 0.0
    for py in range(H):
        for px in range(W):
            an = dxref.analytic_fragments(V, W, H, px, py, po, cb, vis)
            r = cov_r.get((px, py), [])
            if sorted(f[0] for f in an) == sorted(f[2] for f in r):
                same += len(an) > 0
                for vol, lpt, z in an:
                    f = next(f for f in r if f[2] == vol)
                    dz = max(dz, abs(z - f[4]))
                    dl = max(dl, float(np.abs(lpt - f[3]).max()))
            else:
                differ += 1
    a, r = half_to_float(analytic), half_to_float(raster)
    d = np.abs(a - r).max(axis=-1)
    covered = (analytic[..., 3] != 0) | (raster[..., 3] != 0)
    print(f"viewport {W}x{H}, {len(vis)} visible cubes, {len(frags)} rasterised fragments (vertices snapped to 1/{1 << bits} pixel)")
    print(f"pixels with fragments: {same + differ}; same fragment set in both: {same}; different set (silhouette coverage): {differ}")
    print(f"over same-set pixels: max |z| diff {dz:.3e}, max |LPt| diff {dl:.3e} (cube half-extent 1)")
    hist = np.histogram(d[covered], bins=[0, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1, 10])[0]
    print(f"image over the {int(covered.sum())} pixels with non-zero alpha, max abs difference per pixel: ==0: {hist[0]}, <1e-4: {hist[1]}, <1e-3: {hist[2]}, "
          f"<1e-2: {hist[3]}, <1e-1: {hist[4]}, more: {hist[5]} (max {d.max():.3e}: PSCube's in-shader adaptive RayCast is discontinuous in the ray)")
    if os.environ.get("RASTER_DEBUG"):
        ys, xs = np.nonzero(d > 1e-2)
        for yy, xx in list(zip(ys, xs))[:12]:
            print(" ", xx, yy, "analytic", a[yy, xx], "raster", r[yy, xx], "raster frags", [(f[2], f[7], f[3].tolist(), f[4], f[5], f[6]) for f in cov_r.get((xx, yy), [])],
                  "analytic frags", [(f[0], f[1].tolist(), f[2]) for f in dxref.analytic_fragments(V, W, H, int(xx), int(yy), po, cb, vis)])
    print(f"bit-identical pixels: {(np.all(analytic == raster, axis=-1)).sum()} of {W * H}")


if __name__ == "__main__":
    main()
