"""How much does the tile cost table (longest ray's samples per tile) change between the frame that wrote it and the frame that
uses it four frames later, with bench.py's moving camera?  Reads the four per-slot tables after every frame."""
import sys, os, ctypes as C
import numpy as np
import torch; torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS["C3"]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
vols = [rc.synth_volume(N, k) for k in range(V)]
c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024, flags=0)
fr = scene.setup_raycaster(c, cfg, vols)
e0 = np.asarray(fr["eye"], np.float64)
r, a0 = float(np.hypot(e0[0], e0[2])), float(np.arctan2(e0[0], -e0[2]))
proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / float(cfg["H"]), scene.Z_NEAR, scene.Z_FAR)
def table(slot):
    p, n = C.c_void_p(), C.c_size_t()
    assert c._lib.mv_get_buffer(c._ctx, 9, slot, C.byref(p), C.byref(n)) == 0
    from multivolumes_amd.distributed import _device_view
    return _device_view(p.value, n.value).view(torch.int32).cpu().numpy().copy()
step = float(sys.argv[1]) if len(sys.argv) > 1 else 0.004
prev = {}
for i in range(64):
    a = a0 + step * i
    e = np.array([r * np.sin(a), e0[1] * (1.0 + 0.25 * np.sin(3.0 * (a - a0))), -r * np.cos(a)], np.float32)
    vp = rc.multiply(rc.look_at_lh(e, (0.0, 0.0, 0.0)), proj)
    d = torch.from_numpy(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0.0, 0.0, 0.0), scene.SPHERE_RADIUS)).cuda()
    c.SetRenderTargets(depthMapDevice=d.data_ptr()); c.UpdateFrame(vp, fr["shadowVP"], e); c.Render(composite=True); c.synchronize()
    slot = (i + 1) % 4   # mv_update_frame advances the slot before the frame
    t = table(slot)
    if slot in prev and i >= 8:
        old = prev[slot]
        both = (t > 0) & (old > 0)
        binw = 128 // 8
        moved = ((t[both] - 1) // binw) != ((old[both] - 1) // binw)
        if i % 8 == 0:
            print("frame %2d: tiles with a cost %d, in both tables %d, bin changed %.1f %%, mean |delta| %.2f samples, new tiles %d" % (i, int((t > 0).sum()), int(both.sum()), 100.0 * moved.mean(), float(np.abs(t[both].astype(np.int64) - old[both]).mean()), int(((t > 0) & (old == 0)).sum())))
    prev[slot] = t
