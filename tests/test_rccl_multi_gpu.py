"""The RCCL transport with more than one rank: one GPU per rank, the library's own communicators (three with whole-frame chains).
Needs a box with at least two GPUs -- skipped otherwise (RCCL refuses two ranks on one device: on one GPU the protocol runs over gloo
callbacks, tests/test_dist_frame_loop.py).  Every multi-rank run sits in a process group of its own under a watchdog: ranks that hang
in a collective are killed and the test fails with what they had printed (fresh child processes, nothing is exec'ed from a process
that has touched the GPU)."""
import json
import os
import sys

import numpy as np
import pytest

from tests.test_dist_frame_loop import ROOT, _free_port, run_group, single_gpu

pytestmark = pytest.mark.gpu


def _gpus():
    import torch
    return torch.cuda.device_count()  # (does not initialise the GPU)


@pytest.mark.parametrize("mode", ["chains", "single"])
def test_frame_loop_over_rccl_matches_single_gpu(tmp_path, mode):
    world = min(_gpus(), 8)
    if world < 2:
        pytest.skip("one GPU: RCCL with world size > 1 needs a GPU per rank")
    frames, every = 26, 13
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MV_TEST_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), mode, "1"]
    r = run_group(cmd, 600, env)  # the watchdog: a hung rank group is killed, the failure carries the ranks' output
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, True)
    assert np.array_equal(d["infos"], infos)
    for i in range(len(imgs)):
        assert np.array_equal(d["imgs"][i], imgs[i]), f"composite of checkpoint {i} differs ({(d['imgs'][i] != imgs[i]).sum()} halfs)"
    assert np.array_equal(d["lms"], lms) and np.array_equal(d["counts"], counts)


def test_bench_starts_its_own_ranks_over_rccl():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: the bench starts N ranks itself (child processes), every rank's
    supervisor starts the measuring process, and rank 0's line says how many ranks RCCL saw."""
    world = min(_gpus(), 8)
    if world < 2:
        pytest.skip("one GPU: RCCL with world size > 1 needs a GPU per rank")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", "T2", "--steps", "20", "--warmup", "5"], 900, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world and d["value"] > 0
    # whichever rung of the ladder ran (bench.py LEVELS: the peer-store transport first), the line says so; RCCL's own rank count only
    # exists on the rungs that open RCCL
    cfg = d["config"]
    assert cfg["transport"] in ("peer", "rccl", "torch-callbacks"), cfg
    assert cfg["transport_ladder"][cfg["fallback_level"]].startswith({"peer": "peer", "rccl": "rccl", "torch-callbacks": ("torch-nccl", "gloo")}[cfg["transport"]]), cfg
    if cfg["transport"] == "rccl":
        assert cfg["rccl_ranks"] == world
    if cfg["transport"] == "peer":
        assert cfg["ranks_hold_the_same_frame"] in (True, None) and cfg["outbound_bytes_per_frame"] > 0
    assert d["cpu_baseline"]["gpu_vs_oracle_max_abs_err"] == 0.0


def test_bench_over_rccl_when_asked():
    """--transport rccl: the library's own communicators, whatever the ladder would have picked first."""
    world = min(_gpus(), 8)
    if world < 2:
        pytest.skip("one GPU: RCCL with world size > 1 needs a GPU per rank")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--config", "T2", "--steps", "20", "--warmup", "5", "--transport", "rccl"], 900, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["transport"] == "rccl" and d["config"]["rccl_ranks"] == world


@pytest.mark.parametrize("variant", ["peer", "peer-fold", "peer-x4"])
def test_frame_loop_over_peer_stores_on_separate_gpus(tmp_path, variant):
    """The peer-store transport with one GPU per rank: the stores cross the fabric, the flags are polled by another device than the one that
    raised them, and the folded waits (the library's choice when every rank owns its GPU) read other GPUs' write-through stores.  Bit for bit
    against the single-GPU frames; the in-kernel coherence test's verdict is part of the output."""
    world = min(_gpus(), 8)
    if world < 2:
        pytest.skip("one GPU: ranks on separate devices need a GPU per rank (tests/test_dist_frame_loop.py runs the transport with the ranks sharing cuda:0)")
    frames, every = 26, 13
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MV_TEST_TRANSPORT="peer", MV_TEST_SEPARATE_GPUS="1", MV_TEST_COHERENCE="1", GPU_MAX_HW_QUEUES="8")
    if variant != "peer":
        env.update(MV_TEST_FOUR_CHAINS="1", MV_TEST_PEER_FOLD="1")
    if variant == "peer-x4":
        env["MV_TEST_FRAMES_PER_LAUNCH"] = "4"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), "chains", "1"]
    r = run_group(cmd, 900, env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("COHERENCE")]
    assert line and " acquire stale 0 first_bad 0 " in line[0] + " ", line
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, True)
    assert np.array_equal(d["infos"], infos)
    for i in range(len(imgs)):
        assert np.array_equal(d["imgs"][i], imgs[i]), f"composite of checkpoint {i} differs ({(d['imgs'][i] != imgs[i]).sum()} halfs)"
    assert np.array_equal(d["lms"], lms) and np.array_equal(d["counts"], counts)
