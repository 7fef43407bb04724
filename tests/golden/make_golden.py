#!/usr/bin/env python3
"""Regenerates tests/golden/small.npz: inputs + the CPU oracle's outputs for a small case
(4 volumes: 2 on the cube-map scheme, 2 on the screen-space scheme; procedural x noise density, analytic-sphere depth / shadow, SH on), small enough to commit.

The reference has no golden vectors of its own and cannot run here, so these vectors pin the ORACLE
(regressions in oracle/mv_oracle.cpp) and give the HIP kernels a fixture that does not need the oracle at run time.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multivolumes_amd import scene  # noqa: E402
from multivolumes_amd import raycaster as rc  # noqa: E402
from oracle import mv_oracle as o  # noqa: E402
from tests.helpers import oracle_pipeline  # noqa: E402

CFG = dict(N=24, L=12, V=4, W=128, H=96, maxRay=40, maxLight=20, mesh=True)
M = 64


def main():
    vols = [rc.synth_volume(CFG["N"], k + 1) for k in range(CFG["V"])]
    ref = oracle_pipeline(o, CFG, vols, M, sh=scene.DEFAULT_SH)
    img, ns = o.composite(CFG["V"], CFG["N"], CFG["L"], CFG["W"], CFG["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"],
                          ref["info"], ref["vis"], ref["cubeMaps"], ref["cubeDepths"])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small.npz")
    np.savez_compressed(
        out, cfg=np.array([CFG[k] for k in ("N", "L", "V", "W", "H", "maxRay", "maxLight")] + [M], np.uint32),
        volumes=np.stack(vols), depth=ref["depth"], shadow=ref["shadow"], sh=scene.DEFAULT_SH, cb=ref["cb"], po=ref["po"], descs=ref["descs"],
        info=ref["info"], vis=ref["vis"], cube=ref["cube"], lightMaps=ref["lightMaps"], cubeMaps=ref["cubeMaps"], cubeDepths=ref["cubeDepths"],
        composite=img, stats=np.array([ref["stats"]["view_samples"], ref["stats"]["lit_samples"], ref["stats"]["rays"], ref["stats"]["light_samples"], ns], np.uint64))
    print("wrote", out, os.path.getsize(out), "bytes;", "info", ref["info"].tolist(), "stats", ref["stats"], "composite samples", ns)


if __name__ == "__main__":
    main()
