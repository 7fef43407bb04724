#!/usr/bin/env python3
"""Where a view-march launch spends its time (needs a -DMV_TIMELINE build loaded through MV_LIB_PATH): per-wavefront wall-clock
stamps -> how many wavefronts are alive over the launch, time inside refills, iterations / reservations per wavefront."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
from multivolumes_amd import distributed as D

cfg = scene.CONFIGS["C3"]
with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=1024) as c:
    for k in range(cfg["V"]):
        c.LoadVolumeData(k, rc.synth_volume(cfg["N"], k))
    fr = scene.setup_raycaster(c, cfg)
    for f in range(24):
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render()
        c.synchronize()
    ptr, n = c.GetBuffer(8)
    t = D._device_view(ptr, n).cpu().numpy().view(np.uint64).reshape(-1, 8).astype(np.int64)
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    start, end, refill, nref, iters, res, last, rays = [t[:, i] for i in range(8)]
    us = lambda x: x / 100.0
    span = us(end.max() - t0)
    print(f"wavefronts {len(t)}, launch span {span:.1f} us; start spread {us(start.max() - t0):.1f} us")
    print(f"per wavefront: life {us((end - start).mean()):.1f} us (min {us((end - start).min()):.1f}, max {us((end - start).max()):.1f}); "
          f"in refills {us(refill.mean()):.1f} us over {nref.mean():.1f} refills ({us(refill.sum()) / max(1, nref.sum()):.2f} us each); "
          f"iterations {iters.mean():.1f} (max {iters.max()}), tiles {res.mean():.2f}, us per iteration {us((end - start - refill).sum()) / iters.sum():.2f}")
    edges = np.linspace(0, span, 11)
    alive = [int(((us(start - t0) <= x) & (us(end - t0) > x)).sum()) for x in edges[:-1] + span / 20]
    print("wavefronts alive at 5%,15%..95% of the span:", alive)
    print(f"after its last successful refill a wavefront lives {us((end - last)[last > 0].mean()):.1f} us on average (the drain), max {us((end - last)[last > 0].max()):.1f}")
