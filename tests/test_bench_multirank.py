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

def run_group(cmd, timeout, env=None, _retry=True):
    """subprocess.run in a process group of its own: on a timeout the whole group is killed (torchrun's workers would otherwise outlive
    it and keep the GPU), and the failure carries what the ranks had printed.  A rendezvous port that was taken between _free_port() and
    torchrun's bind (EADDRINUSE: seen once in some hundred launches) gets one more try on another port."""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError(f"timed out after {timeout} s; stderr tail:\n{err[-3000:]}\nstdout tail:\n{out[-1000:]}")
    if p.returncode != 0 and _retry and "EADDRINUSE" in err and "--master-port" in cmd:
        cmd = list(cmd)
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        return run_group(cmd, timeout, env, _retry=False)
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)



@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--orbit"], ["--orbit", "--dist-single-stream"], ["--orbit", "--frames-per-launch", "4"]])
def test_two_ranks_on_one_gpu(extra):
    env = dict(os.environ, MV_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "T2", "--steps", "6", "--warmup", "2", "--steady-frames", "24", "--latency-frames", "24"] + extra
    r = run_group(cmd, 240, env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["exchange_bytes_per_frame"] > 0
    # the first rung of the ladder: the peer-store transport (IPC mappings work between ranks that share a GPU) -- except for the
    # single-view-stream schedule, which it does not run; every rank ends with the same frame, and the line says what every rank took
    assert d["config"]["transport"] == ("torch-callbacks" if "--dist-single-stream" in extra else "peer"), d["config"]["transport"]
    assert d["config"]["ranks_hold_the_same_frame"] is True and len(d["config"]["ms_per_step_per_rank"]) == 2
    assert d["cpu_baseline"]["gpu_vs_oracle_max_abs_err"] == 0.0
    # depth and what it costs a frame; the steady-state figure beside the K timed frames
    per_launch = 4 if "--frames-per-launch" in extra else 1
    assert d["frames_in_flight_total"] == (2 if "--dist-single-stream" in extra else (4 * per_launch if d["config"]["transport"] == "peer" else 3)), d["frames_in_flight_total"]
    # (latency has its own pass behind the timed frames: on ranks that share a GPU with gloo in between, either pass can hit a stall the other does not)
    assert d["latency_samples"] > 0 and d["latency_ms"] > 0 and d["latency_ms_max"] >= d["latency_ms"]
    assert d["steady_state_ms_per_step"] > 0 and d["steady_state_frames"] == 24
    # every stage of rank 0's frame is timed, whichever path enqueued it (round 6: the composite of the loop's one-frame tail was not)
    assert all(d["kernel_ms"][k] > 0 for k in ("cull", "march_light", "march_view", "composite")), d["kernel_ms"]
    if d["config"]["transport"] == "peer":
        c = d["config"]
        assert c["peer"]["rows_to_root"] and c["peer"]["acquire"] and c["peer"]["stale_words"] == 0
        assert len(c["outbound_bytes_per_frame_per_rank"]) == 2 and c["outbound_bytes_per_frame"] == max(c["outbound_bytes_per_frame_per_rank"]) > 0
        assert c["outbound_gb_per_s_at_this_frame_time"] > 0 and "light map" in c["ranks_compared"]
    # the same frame on one rank: the sharded run marches the same samples
    one = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "T2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--steady-frames", "0"] + [e for e in extra if e in ("--orbit",)], 240)
    assert one.returncode == 0, one.stderr[-2000:]
    s = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("view_samples_per_frame", "lit_samples_per_frame", "rays_per_frame", "light_samples_per_frame"):
        assert d["config"][k] == s["config"][k], k


@pytest.mark.gpu
def test_bench_starts_and_supervises_its_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the bench starts two ranks as child processes (torch.distributed.run), every rank is a
    supervisor that starts the measuring process and watches its progress marks, rank 0's line comes back with n_gpus 2 (here the two
    ranks share cuda:0 over gloo, MV_BENCH_SHARE_GPU=1).  And a WORLD_SIZE that contradicts --gpus is an error, not a warning."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MV_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "T2", "--steps", "6", "--warmup", "2", "--steady-frames", "24", "--latency-frames", "24"], 400, env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["transport"] == "peer" and d["config"]["fallback_level"] == 0
    bad = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "T2", "--steps", "2", "--warmup", "1"], 120,
                    dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["fail", pytest.param("hang", marks=pytest.mark.skipif(os.environ.get("MV_SLOW_TESTS") != "1", reason="100 s inside the whole suite: MV_SLOW_TESTS=1 (passes: docs/STATUS.md)"))])
def test_a_rung_that_fails_or_hangs_on_one_rank_moves_every_rank_to_the_next(how):
    """The safety net of the first run on a node with more than one GPU: bench.py's ladder of transports.  Rank 1 of the first rung (peer stores) leaves with
    an error, or stops making progress in its warm-up (MV_BENCH_INJECT, with the supervisors' patience for a warm-up cut to 25 s): the supervisors kill the rung's
    processes on EVERY rank, say so, start the next rung together, and rank 0's line comes from it (config.fallback_level 1) with exit status 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MV_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MV_BENCH_INJECT=f"{how}:peer:1", MV_BENCH_PEER_TIMEOUT_MS="3000")  # (a killed rank leaves the GPU when its polling kernels give up)
    if how == "hang":
        env["MV_BENCH_WARMUP_TIMEOUT_S"] = "25"
    r = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "T2", "--steps", "6", "--warmup", "2", "--steady-frames", "0", "--latency-frames", "0", "--no-cpu-baseline"], 400, env)
    assert r.returncode == 0, "\n".join(l for l in r.stderr.splitlines() if l.startswith("bench.py") or "Error" in l)[-3000:]
    assert f"injected {how} on rank 1 of rung peer" in r.stderr and "transport 'peer' failed or stopped making progress" in r.stderr and "trying the next one" in r.stderr, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["fallback_level"] == 1 and d["config"]["transport"] == "torch-callbacks" and d["value"] > 0, d["config"]
    assert d["config"]["ranks_hold_the_same_frame"] is True


@pytest.mark.gpu
def test_one_rank_bench_line_carries_both_cpu_baselines():
    """The default (N = 1) bench line: `cpu_baseline` is the reference's compiled shader on the host cores when oracle/_ref travelled
    with the tree (kind "reference", equal to the oracle and to the GPU), else the oracle ("port") -- never a silently failed leg."""
    r = run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "T2", "--steps", "6", "--warmup", "2"], 300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cb = d["cpu_baseline"]
    assert "reference_error" not in cb, cb
    assert cb["gpu_vs_oracle_max_abs_err"] == 0.0
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libmv_dxref.so")):
        assert cb["kind"] == "reference" and cb["reference_equals_oracle"] and cb["gpu_vs_reference_max_abs_err"] == 0.0
    else:
        assert cb["kind"] == "port"
    assert d["frames_in_flight_total"] == 4 and d["latency_samples"] > 0 and d["latency_ms"] > 0 and d["steady_state_ms_per_step"] > 0   # (separate passes: their figures are not compared)
    r = d["roofline"]
    assert 0 < r["frac"] <= r["frac_with_fused_raycast"] and 0 <= r["fetch_free_share"] < 1 and d["n_gpus"] == 1
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9, "frac is achieved / peak: SURVEY 8(d)'s algorithmic bytes of the a3 loop over the launch time"
    assert cb["single_thread"] > 0 and 0 < cb["parallel_efficiency"] <= 1.5
