#!/usr/bin/env python3
"""How the view march's wave-iterations split (needs a -DMV_DIAG_ITER -DMV_EMPTY_REGIONS=0 build loaded through MV_LIB_PATH): an iteration of a
wavefront costs what its most expensive lane costs, so what a cheaper fetch-free sample can save is bounded by the iterations in which NO lane
fetches.  Prints, for one C3 frame's cube tiles: wave-iterations, those with no fetching lane, those whose lanes all sit in all-empty 16^3 regions."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multivolumes_amd import MultiRayCaster, scene
from multivolumes_amd import raycaster as rc
from multivolumes_amd import distributed as D

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = scene.CONFIGS[name]
with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=1024) as c:
    for k in range(cfg["V"]):
        c.LoadVolumeData(k, rc.synth_volume(cfg["N"], k))
    fr = scene.setup_raycaster(c, cfg)
    ptr, n = c.GetBuffer(8)
    view = D._device_view(ptr, n)
    for f in range(3):
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.synchronize(); view.zero_(); torch.cuda.synchronize()
        c.Render()
        c.synchronize()
    t = view.cpu().numpy().view(np.uint64).reshape(-1, 8).astype(np.int64).sum(axis=0)
    st = c.GetStats()
    it, skip, zero, lanes = t[0], t[1], t[2], t[3]
    print(f"{name}: wave-iterations {it} (view samples {st['view_samples']}, fetch-free {st['fetch_free_samples']}); "
          f"no lane fetches in {skip} ({skip / it:.3f}), lanes in them {lanes} = {lanes / max(1, st['fetch_free_samples']):.3f} of the fetch-free samples; "
          f"all lanes in all-empty regions in {zero} ({zero / it:.3f})")
