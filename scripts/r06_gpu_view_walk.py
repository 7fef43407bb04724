#!/usr/bin/env python3
"""Round 6: the view march on the GPU against the oracle at the oracle's boundaries (tests/helpers.py output_boundaries): the adjacent input floats between which any
stored half or cube depth of the oracle's cull + view march changes.  python scripts/r06_gpu_view_walk.py [seeds]   (through gpurun)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multivolumes_amd import MultiRayCaster  # noqa: E402
from oracle import mv_oracle as oracle  # noqa: E402
from tests.helpers import halfs_without_nan_payloads, output_boundaries, small_march_scene  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    t0 = time.time()
    found = bad = 0
    for seed in range(seeds):
        rng, cfg, vols, descs, cb, po, depth, lms = small_march_scene(seed)
        V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
        info0, vis0, cube0 = oracle.cull(V, po, descs, cb, cfg["maxRay"])
        if len(cube0) == 0:
            continue

        def on_oracle(c, p):
            info, vis, cube = oracle.cull(V, p, descs, c, cfg["maxRay"])
            cm, cd = oracle.march_view(V, N, L, W, H, vols, lms, depth, p, c, info, cube)[:2]
            out = [np.asarray(info).tobytes()]
            for v in np.sort(cube):
                mip = int(info[v, 0])
                out += [halfs_without_nan_payloads(oracle.cube_face_view(cm[v], N, mip)).tobytes(), np.asarray(oracle.cube_face_view(cd[v], N, mip)).tobytes()]
            return b"".join(out)
        with MultiRayCaster(N, L, V, V, (W, H), shadowMapSize=32) as g:
            for i, v in enumerate(vols):
                g.LoadVolumeData(i, v)
            g.SetMaxSamples(cfg["maxRay"], cfg["maxLight"])
            g.SetRenderTargets(depth)
            for v in range(V):
                g.UploadLightMap(v, lms[v])

            def on_gpu(c, p):
                g.ClearCubeMaps()
                g.UpdateFrameRaw(c, p)
                g.cullVolumes()
                g.rayMarchV()
                info = g.ReadBackVolumeInfo()
                _, cube = g.ReadBackVisibleLists()
                full = np.zeros_like(np.asarray(info))
                vis, _ = g.ReadBackVisibleLists()
                full[np.sort(vis)] = info[np.sort(vis)]
                out = [full.tobytes()]
                for v in np.sort(cube):
                    mip = int(info[v, 0])
                    out += [halfs_without_nan_payloads(g.ReadBackCubeMap(int(v), mip)).tobytes(), np.asarray(g.ReadBackCubeDepth(int(v), mip)).tobytes()]
                return b"".join(out)
            for c, p in output_boundaries(rng, cb, po, cube0, on_oracle):
                found += 1
                if on_oracle(c, p) != on_gpu(c, p):
                    bad += 1
                    print("seed", seed, "differs", flush=True)
    print(f"view-march boundaries {found}, GPU differs from the oracle at {bad}, {time.time() - t0:.1f} s")
    return 1 if bad or not found else 0


if __name__ == "__main__":
    sys.exit(main())
