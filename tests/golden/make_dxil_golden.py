#!/usr/bin/env python3
"""Regenerates tests/golden/dxil_small.npz: outputs of the REFERENCE'S SHIPPED COMPILED SHADERS (Bin/CSVolumeCull.cso, CSRayMarchL.cso,
CSRayMarchV.cso, PSDepthPeel.cso, PSCube.cso, PSResolveOIT.cso run on the CPU through oracle/_ref/libmv_dxref.so, see oracle/dxil) for a small seeded case, so that the pin of the
oracle to the reference's own code travels to machines that have neither /root/reference nor the library.  Inputs are not stored: the
test rebuilds them from the same seeds (scene.py / raycaster.synth_volume / tests.helpers.host_frame).
Run from the repo root, where /root/reference exists:  make -C oracle ref && python tests/golden/make_dxil_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multivolumes_amd import scene  # noqa: E402
from multivolumes_amd import raycaster as rc  # noqa: E402
from oracle import dxref, mv_oracle as o  # noqa: E402
from tests.helpers import default_descs, host_frame  # noqa: E402

CFG = dict(N=24, L=12, V=4, W=128, H=96, maxRay=40, maxLight=20, mesh=True)
M = 64
EYE = (30.0, 22.0, -70.0)


def run(pipeline_cull, pipeline_light, pipeline_view, pipeline_composite):
    """The frame sequence of the fixture on any implementation with the oracle's signatures."""
    V, N, L = CFG["V"], CFG["N"], CFG["L"]
    vols = [rc.synth_volume(N, k + 1) for k in range(V)]
    descs = default_descs(V, V, N)
    _, vp, svp, cb, po = host_frame(CFG, 0, EYE)
    depth, shadow = scene.depth_and_shadow(CFG, vp, svp, M)
    info, vis, cube = pipeline_cull(V, po, descs, cb, CFG["maxRay"])
    lms = np.zeros((V, L, L, L), np.uint32)
    for f in range(V):
        _, _, _, cb, po = host_frame(CFG, f, EYE)
        lms = pipeline_light(V, N, L, M, vols, descs, po, cb, shadow, None, vis, CFG["maxLight"], lms)
    cm, cd = pipeline_view(V, N, L, CFG["W"], CFG["H"], vols, lms, depth, po, cb, info, cube)
    img = pipeline_composite(V, N, L, CFG["W"], CFG["H"], vols, lms, depth, po, cb, info, np.sort(vis), cm, cd)
    return dict(info=info, vis=np.sort(vis), cube=np.sort(cube), lightMaps=lms, cubeMaps=cm, cubeDepths=cd, composite=img)


def oracle_run():
    o.load()
    return run(o.cull,
               lambda *a: o.march_light(*a[:-1], lightMaps=a[-1])[1],
               lambda *a: o.march_view(*a)[:2],
               lambda *a: o.composite(*a)[0])


def main():
    assert dxref.available(), "oracle/_ref/libmv_dxref.so missing: make -C oracle ref"
    out = run(dxref.cull, lambda *a: dxref.march_light(*a[:-1], lightMaps=a[-1]), dxref.march_view, dxref.composite)
    path = os.path.join(ROOT, "tests", "golden", "dxil_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; info", out["info"].tolist(), "cube-map volumes", out["cube"].tolist())


if __name__ == "__main__":
    main()
