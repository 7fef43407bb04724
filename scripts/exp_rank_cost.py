"""What one rank of the multi-GPU loop costs per frame WITHOUT the collectives: a sharded context (rank r of `world`) on this one
GPU, mv_dist_frame with do-nothing collectives.  The images are wrong (nothing is exchanged); the kernels a rank runs per frame
(replicated cull + orders, its light marches, its share of the view march, pack / unpack, its rows of the composite, copy-forward)
and their scheduling are the real ones.  max over ranks = the frame time an ideal fabric would give; the ratio to the one-GPU frame
is the ceiling of the strong-scaling curve (run through gpurun: python scripts/exp_rank_cost.py [C3])."""
import sys, os, time
import torch; torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = dict(scene.CONFIGS[name]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
vols = [rc.synth_volume(N, k) for k in range(V)]
frames = int(os.environ.get("FRAMES", "400"))
only = os.environ.get("MODE")  # "chains" / "single": one schedule only
for mode, flags in (("chains (whole frames on three view streams)", _lib.MV_FLAG_PIPELINE_FRAMES),
                    ("single view stream", _lib.MV_FLAG_PIPELINE_FRAMES | _lib.MV_FLAG_SINGLE_VIEW_STREAM)):
    if only and not mode.startswith(only):
        continue
    print(mode, flush=True)
    for world in (1, 2, 4, 8):
        per_rank = []
        for rank in range(world):
            with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024, flags=flags, shard=(rank, world)) as c:
                fr = scene.setup_raycaster(c, cfg, vols)
                c.DistInitCallbacks(lambda s, r, n, st, w: 0, lambda b, n, st, w: 0); c.DistSetInPlaceGather(True); c.DistSetCullCrosscheck(False)  # like RCCL: no staging copies; nothing is delivered, so no cross-rank check
                for _ in range(min(200, frames)): c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
                c.DistFlush(); c.synchronize()
                t0 = time.perf_counter()
                for _ in range(frames): c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
                c.DistFlush(); c.synchronize()
                per_rank.append((time.perf_counter() - t0) / frames * 1e3)
        print("  world %d: ms/frame per rank %s  -> max %.4f" % (world, [round(x, 4) for x in per_rank], max(per_rank)), flush=True)
