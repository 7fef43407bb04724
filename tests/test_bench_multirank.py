"""bench.py's multi-rank path on ONE GPU: two ranks share cuda:0 and the collectives go through gloo (MV_BENCH_SHARE_GPU=1, a
functional mode -- RCCL refuses two ranks on one device).  Exercises the in-library frame loop (mv_dist_frame: light-map all-reduce,
cube-face all-gather, row-band composite and its gather, frame k - 1's exchange beside frame k's march) end to end through bench.py;
rank 0's cpu_baseline leg checks its own cube maps against the oracle.  tests/test_dist_frame_loop.py compares pixels."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--orbit"]])
def test_two_ranks_on_one_gpu(extra):
    env = dict(os.environ, MV_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "T2", "--steps", "6", "--warmup", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange_bytes_per_frame"] > 0
    assert d["cpu_baseline"]["gpu_vs_oracle_max_abs_err"] == 0.0
    # the same frame on one rank: the sharded run marches the same samples
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "T2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    s = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("view_samples_per_frame", "lit_samples_per_frame", "rays_per_frame", "light_samples_per_frame"):
        assert d["config"][k] == s["config"][k], k
