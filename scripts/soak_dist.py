#!/usr/bin/env python3
"""Soak of the multi-GPU frame loop: tests/test_dist_frame_loop.py's moving-camera sequence, long (default 900 frames), on `world` ranks
sharing this GPU over gloo callbacks, compared with the single-GPU run at every checkpoint (composites, VolumeInfo, every light map
at the end, the summed counters): python scripts/soak_dist.py [world] [frames] [every] [chains|single]   (through gpurun)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_dist_frame_loop import _free_port, run_group, single_gpu  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 900
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    mode = sys.argv[4] if len(sys.argv) > 4 else "chains"
    out = f"/tmp/soak_dist_{world}.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), mode, "1"]
    r = run_group(cmd, 1500, env)
    if r.returncode != 0:
        why = [l for l in r.stderr.splitlines() if "unavailable" in l or "MVError" in l]
        raise SystemExit("ranks failed: " + " | ".join(why[:4]) + "\n" + r.stderr[-1500:])
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, True)
    bad = [i for i in range(len(imgs)) if not np.array_equal(d["imgs"][i], imgs[i])]
    print(f"world {world}, {mode}, {frames} frames, checkpoint every {every}: {len(imgs)} checkpoints, differing composites {bad}, "
          f"VolumeInfo equal {np.array_equal(d['infos'], infos)}, light maps equal {np.array_equal(d['lms'], lms)}, "
          f"counters equal {np.array_equal(d['counts'], counts)} {counts.tolist()}, bytes exchanged per frame {int(d['nbytes']) // frames}")
    assert not bad and np.array_equal(d["infos"], infos) and np.array_equal(d["lms"], lms) and np.array_equal(d["counts"], counts)


if __name__ == "__main__":
    main()
