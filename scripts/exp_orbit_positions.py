"""Is the moving-camera bench slower because the camera moves (stale tile costs, changing grids) or because the positions on the
arc are heavier than the default one?  View-march time (one frame at a time) with the camera STANDING at positions of bench.py's
orbit, against the same positions visited in motion."""
import sys, os, time
import numpy as np
import torch; torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS["C3"]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
vols = [rc.synth_volume(N, k) for k in range(V)]
c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024, flags=int(os.environ.get("MV_EXP_FLAGS", "0")))
fr = scene.setup_raycaster(c, cfg, vols)
e0 = np.asarray(fr["eye"], np.float64)
r, a0 = float(np.hypot(e0[0], e0[2])), float(np.arctan2(e0[0], -e0[2]))
proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / float(cfg["H"]), scene.Z_NEAR, scene.Z_FAR)
pos = []
for j in range(128):
    a = a0 + 0.004 * j
    e = np.array([r * np.sin(a), e0[1] * (1.0 + 0.25 * np.sin(3.0 * (a - a0))), -r * np.cos(a)], np.float32)
    vp = rc.multiply(rc.look_at_lh(e, (0.0, 0.0, 0.0)), proj)
    d = torch.from_numpy(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0.0, 0.0, 0.0), scene.SPHERE_RADIUS)).cuda()
    pos.append((e, vp, d))
torch.cuda.synchronize()
def run(seq, n):
    c.synchronize(); c.ResetStats(); c.SetProfiling(True)
    for i in range(n):
        e, vp, d = seq(i)
        c.SetRenderTargets(depthMapDevice=d.data_ptr()); c.UpdateFrame(vp, fr["shadowVP"], e); c.Render(composite=True); c.synchronize()
    st = c.GetStats(); c.SetProfiling(False)
    return st["ms_march_view"] / max(1, st["n_march_view"]), st["view_samples"] / n, st["rays"] / n
standing = []
for j in (0, 16, 32, 48, 64, 80, 96, 112, 127):
    for _ in range(8): run(lambda i: pos[j], 1)   # warm the cost table at this position
    ms, smp, rays = run(lambda i: pos[j], 24)
    standing.append(ms)
    print("standing at %3d: march %.4f ms  %.2f M samples  %.0f rays" % (j, ms, smp / 1e6, rays))
def moving(i):
    k = i % 254
    return pos[k if k < 128 else 254 - k]
run(moving, 254)
ms, smp, rays = run(moving, 508)
print("moving over the arc: march %.4f ms  %.2f M samples  %.0f rays   (mean of the standing positions %.4f ms)" % (ms, smp / 1e6, rays, sum(standing) / len(standing)))
if os.environ.get("MV_EXP_PER_FRAME"):
    # per-frame march time while moving, beside the standing time at the same position
    seq = []
    run(moving, 254)
    for i in range(0, 128, 4):
        pass
    c.synchronize(); c.ResetStats(); c.SetProfiling(True)
    last = 0.0
    for i in range(254):
        e, vp, d = moving(i)
        c.SetRenderTargets(depthMapDevice=d.data_ptr()); c.UpdateFrame(vp, fr["shadowVP"], e); c.Render(composite=True); c.synchronize()
        st = c.GetStats()
        seq.append(st["ms_march_view"] - last); last = st["ms_march_view"]
    print("moving, per frame (position: ms):", " ".join("%d:%.3f" % (i if i < 128 else 254 - i, seq[i]) for i in range(0, 254, 3)))
if os.environ.get("MV_EXP_OSC"):
    for j in (0, 48, 96):
        for _ in range(8): run(lambda i: pos[j], 1)
        a = run(lambda i: pos[j], 48)[0]
        b = run(lambda i: pos[j + (i & 1)], 48)[0]           # two positions one step apart, alternating
        d = run(lambda i: pos[j + (i % 8)], 48)[0]           # eight positions, round and round (every slot's table sees a 4-step-old one)
        print("position %3d: standing %.4f  alternating j / j+1 %.4f  cycling j..j+7 %.4f" % (j, a, b, d))
