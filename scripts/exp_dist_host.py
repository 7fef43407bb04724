import sys, os, time
import torch; torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
cfg = dict(scene.CONFIGS["C3"]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
vols = [rc.synth_volume(N, k) for k in range(V)]
flags = _lib.MV_FLAG_PIPELINE_FRAMES | (_lib.MV_FLAG_SINGLE_VIEW_STREAM if os.environ.get("SINGLE") == "1" else 0)
for world, rank in (((8, 2), (1, 0)) if os.environ.get("ONLY8") != "1" else ((8, 2),)):
    with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024, flags=flags, shard=(rank, world)) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        c.DistInitCallbacks(lambda s, r, n, st, w: 0, lambda b, n, st, w: 0); c.DistSetInPlaceGather(True); c.DistSetCullCrosscheck(False)  # like RCCL: no staging copies; nothing is delivered, so no cross-rank check
        for _ in range(200): c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.DistFlush(); c.synchronize()
        # enqueue-only: few enough frames that the host never waits for a slot (4 slots)... measure per call anyway
        ts = []
        for i in range(300):
            a = time.perf_counter(); c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); ts.append(time.perf_counter() - a)
        c.DistFlush(); c.synchronize()
        ts.sort()
        print("world %d: DistFrame host time us: min %.1f median %.1f p90 %.1f" % (world, ts[0] * 1e6, ts[len(ts) // 2] * 1e6, ts[int(len(ts) * 0.9)] * 1e6))
        # the same frames with a device sync after each: GPU time of one frame's work alone
        t0 = time.perf_counter()
        for i in range(100):
            c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"]); c.synchronize()
        print("   one frame at a time (sync after each): %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
