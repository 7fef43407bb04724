"""Long-run stability check: 40 000 frames with an orbiting camera on a context with frames in flight and on a one-stream
context; the composites and the counters must agree bit for bit at every checkpoint (event / slot / buffer rotation, counter
width, nothing drifts).

The reference never clears its cube maps, so a texel a frame does not write keeps old data (CSRayMarch.hlsl:84-85 returns early
for faces that are not visible) and a 4-tap CubeCast gather across a cube edge can read it; with frames in flight the rotating
output sets are kept reading like one by the copy-forward of unmarched faces (mv_composite.hip copyForwardKernel), which this run checks
over tens of thousands of frames.  tests/test_gpu_configs.py holds a 600-frame version."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS[os.environ.get("MV_SOAK_CONFIG", "T2")]) if os.environ.get("MV_SOAK_CONFIG") else dict(scene.CONFIGS["T2"], W=320, H=200)
V, N = cfg["V"], cfg["N"]
vols = [rc.synth_volume(N, k) for k in range(V)]
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
check = int(sys.argv[2]) if len(sys.argv) > 2 else frames // 8
# MV_SOAK_FOUR_CHAINS=1: the pipelined context as bench.py creates it on one GPU since round 4 (four chains, every frame's upload and cull on its own view stream);
# MV_SOAK_FRAMES_PER_LAUNCH=B: ... and B consecutive frames per view stream (mv_config.frames_per_launch outside the multi-GPU loop)
four = os.environ.get("MV_SOAK_FOUR_CHAINS") == "1"
per_launch = int(os.environ.get("MV_SOAK_FRAMES_PER_LAUNCH", "0"))
cs = [MultiRayCaster(N, cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64),
      MultiRayCaster(N, cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=_lib.MV_FLAG_PIPELINE_FRAMES | (_lib.MV_FLAG_CULL_ON_VIEW_STREAM if four else 0),
                     framesInFlight=4 if four else 0, framesPerLaunch=per_launch)]
frs = [scene.setup_raycaster(c, cfg, vols, use_sh=True) for c in cs]
proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0)
t0 = time.time()
for f in range(1, frames + 1):
    a = 0.002 * f
    e = np.array([90.0 * np.sin(a), 20.0 + 15.0 * np.sin(0.37 * a), -90.0 * np.cos(a)], np.float32)
    vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
    for c, fr in zip(cs, frs):
        c.UpdateFrame(vp, fr["shadowVP"], e)
        c.Render(composite=True)
    if f % check == 0:
        imgs = [c.ReadBackComposite() for c in cs]
        sts = [c.GetStats() for c in cs]
        same = np.array_equal(imgs[0], imgs[1]) and all(sts[0][k] == sts[1][k] for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples"))
        print("frame %6d  %5.1f s  composite equal %s  view samples %d  nonzero px %d" % (f, time.time() - t0, same, sts[1]["view_samples"], int((imgs[1][..., 3] != 0).sum())), flush=True)
        if not same:
            print("   image halfs differing:", int((imgs[0] != imgs[1]).sum()), {k: (sts[0][k], sts[1][k]) for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples")})
            info = [c.ReadBackVolumeInfo() for c in cs]
            print("   VolumeInfo equal:", np.array_equal(info[0], info[1]))
            lm = [[c.ReadBackLightMap(v) for v in range(V)] for c in cs]
            print("   light maps differing:", [v for v in range(V) if not np.array_equal(lm[0][v], lm[1][v])])
        assert same
for c in cs: c.close()
print("soak ok")
