"""Dump the cube maps of one small frame (MV_LIB_PATH selects the library): python scripts/exp_dump_frame.py out.npz [config]; compare two dumps with --cmp a.npz b.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        d = x != y
        print(k, x.shape, "differing", int(d.sum()))
        if d.any() and x.ndim == 4:
            idx = np.argwhere(d.any(axis=-1))
            print("  faces", sorted(set(idx[:, 0].tolist())), "rows", idx[:, 1].min(), idx[:, 1].max(), "cols", idx[:, 2].min(), idx[:, 2].max(), "first", idx[:8].tolist())
            f, r, c = idx[0]
            print("  a", x[f, r, c].view(np.float16), "b", y[f, r, c].view(np.float16))
    sys.exit(0)
from multivolumes_amd import MultiRayCaster, scene
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C1"])
vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
with MultiRayCaster(cfg["N"], cfg["L"], cfg["V"], cfg["V"], (cfg["W"], cfg["H"]), shadowMapSize=256) as c:
    fr = scene.setup_raycaster(c, cfg, vols)
    for f in range(cfg["V"]):
        c.m_frameIdx = f
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.cullVolumes(); c.rayMarchL()
    c.rayMarchV(); c.synchronize()
    info = c.ReadBackVolumeInfo(); vis, cube = c.ReadBackVisibleLists()
    out = {f"cm{int(v)}": c.ReadBackCubeMap(int(v), int(info[v, 0])) for v in cube}
    st = c.GetStats()
    print({k: st[k] for k in ("view_samples", "lit_samples", "rays", "fetch_free_samples")})
    np.savez(sys.argv[1], **out)
