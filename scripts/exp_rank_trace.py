"""One rank of the multi-GPU loop with do-nothing collectives (see exp_rank_cost.py), for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_rank -- python3 scripts/exp_rank_trace.py 8 2 [single]; python3 scripts/timeline.py /tmp/prof_rank"""
import sys, os
import torch; torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multivolumes_amd import MultiRayCaster, scene, _lib
from multivolumes_amd import raycaster as rc
world, rank = int(sys.argv[1]), int(sys.argv[2])
cfg = dict(scene.CONFIGS["C3"]); V, N, L = cfg["V"], cfg["N"], cfg["L"]
vols = [rc.synth_volume(N, k) for k in range(V)]
flags = _lib.MV_FLAG_PIPELINE_FRAMES | (_lib.MV_FLAG_SINGLE_VIEW_STREAM if len(sys.argv) > 3 and sys.argv[3] == "single" else 0)
with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=1024, flags=flags, shard=(rank, world)) as c:
    fr = scene.setup_raycaster(c, cfg, vols)
    c.DistInitCallbacks(lambda s, r, n, st, w: 0, lambda b, n, st, w: 0); c.DistSetInPlaceGather(True); c.DistSetCullCrosscheck(False)  # like RCCL: no staging copies; nothing is delivered, so no cross-rank check
    for _ in range(600): c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
    c.DistFlush(); c.synchronize()
