#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: volume Msamples/s (+ frames/s) of the cube-map-space ray-march path
at 1024^2 viewport, 16 x 256^3 volumes, 128 steps/ray + mesh-depth occlusion (config C3), on N MI355X GPUs.

One "step" = one frame of the hot path exactly as MultiRayCaster::Render records it (MultiRayCaster.cpp:355-385):
UpdateFrame upload -> cull -> marchL (one volume, round-robin) -> marchV [-> composite].  Inputs (volumes, depth,
shadow map) are resident in HBM before the timed region.  With --gpus N > 1 the 16 volumes are sharded v % N across
ranks (strong scaling, BASELINE config C4) and the selected-mip cube faces are all-gathered over RCCL every frame.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (marchViewKernel) with the algorithmic bytes of
SURVEY.md 8(d): 64 B/view sample + 32 B/lit sample + 16 B/ray, divided by its HIP-event launch time.  On one GPU the
timed region keeps several frames in flight, so two view-march launches share the GPU and a launch's event-to-event time
(`launch_ms_in_flight`) is not the kernel's own speed; `launch_ms` therefore comes from a second pass of the same frames
run one at a time (synchronise after every frame), still HIP events inside this process.
`cpu_baseline` is the reference's own compiled view-march shader run on the host cores (oracle/_ref, kind "reference") when that
library is there, with the CPU oracle's rate beside it ("port"); the oracle alone otherwise.  Both march the same frame's marchV.

Defaults: 200 warm-up + 1000 timed frames (about 0.5 s of GPU time).  The HIP runtime stalls once for 30-50 ms around the
140th frame of a process (its command / signal pools grow; measured, one-time): the setup phase below runs 176 frames for
that reason, before the caller's warm-up.
"""
import argparse
import json
import os
import sys
import time

# more than three whole frames in flight need more than the HIP runtime's default of four hardware queues per process: must be in
# the environment before the runtime starts (torch starts it), so before anything else (include/multivolumes.h, mv_config.frames_in_flight)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (no OMP_PROC_BIND here: libgomp then pins the MAIN thread to one CPU when it loads, and every pthread created afterwards -- the group
# workers of the reference leg of cpu_baseline, oracle/dxil/dxil_rt.c -- inherits that one-CPU mask: that leg ran at single-thread speed)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (cpu_baseline leg only) how many CPUs the box really grants -- affinity and cgroup quota, read NOW: once an OpenMP runtime is loaded
# with OMP_PROC_BIND set (torch's, the oracle's) the main thread is pinned to one CPU and the mask would say 1
def _host_cpu_budget():
    """min(CPUs of the affinity mask, cgroup CPU quota): a GPU box of this pool shows 256 logical CPUs and grants 16 (cpu.max)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


CPU_BUDGET = _host_cpu_budget()

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def roofline_fraction(alg_bytes, requested_bytes, traffic_bytes, launch_ms):
    """roofline.achieved / frac of the dominant kernel: SURVEY 8(d)'s algorithmic bytes per launch over the launch's duration while that
    is a fraction (<= 1: the headline config); otherwise the bytes the launch requested from the memory system (the algorithmic bytes
    minus the samples whose brick-occupancy bit spared the fetch), and if cache hits make even that exceed the peak, the measured HBM
    traffic."""
    if not launch_ms or launch_ms <= 0:
        return {"achieved": 0.0, "frac": None, "frac_basis": None}
    gbs = lambda b: b / (launch_ms * 1e-3) / 1e9
    alg = gbs(alg_bytes)
    if alg <= HBM_PEAK_GBS:
        return {"achieved": alg, "frac": alg / HBM_PEAK_GBS, "frac_basis": "algorithmic bytes (SURVEY 8d: 64 B / view sample + 32 B / lit sample + 16 B / ray)"}
    out = {"frac_algorithmic_uncapped": alg / HBM_PEAK_GBS, "achieved_algorithmic_uncapped": alg}
    req = gbs(requested_bytes)
    if req <= HBM_PEAK_GBS:
        out.update({"achieved": req, "frac": req / HBM_PEAK_GBS, "frac_basis": "requested bytes (algorithmic bytes without the fetch-free samples): the algorithmic figure exceeds the peak on this workload"})
    elif traffic_bytes:
        out.update({"achieved": gbs(traffic_bytes), "frac": gbs(traffic_bytes) / HBM_PEAK_GBS, "frac_basis": "measured HBM traffic (rocprofv3 PMC): algorithmic and requested bytes both exceed the peak (cache hits)"})
    else:
        # nothing measured that is a fraction of the roofline: say so instead of inventing one
        out.update({"achieved": None, "frac": None, "frac_basis": "none: algorithmic and requested bytes exceed the peak and no traffic measurement matches these sources (scripts/make_profiles.sh)"})
    return out


def march_source_hash():
    """Hash of the sources that decide the view march's memory traffic: profiles/traffic.json carries the hash it was measured
    at, and a stale measurement is not reported (scripts/make_profiles.sh regenerates it)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("mv_march_tiles.hip", "mv_march.hip", "mv_device.hpp", "mv_kernels.hpp", "mv_kernels.hip"):
        with open(os.path.join(ROOT, "multivolumes_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--config", default="C3", help="C1/C2/C3/C5/T1/T2 (default C3 = the config the metric is quoted on)")
    p.add_argument("--stress", action="store_true", help="C3-stress: force mip 0, all 6 faces (SURVEY 8d)")
    p.add_argument("--no-composite", action="store_true", help="leave the K-buffer OIT composite out of the step")
    p.add_argument("--no-pipeline", action="store_true", help="one stream, no overlap of frame i+1's light march with frame i's view march")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--dist-single-stream", action="store_true", help="--gpus > 1: the multi-GPU loop's single-view-stream schedule instead of whole frames on three view streams")
    p.add_argument("--orbit", action="store_true", help="the camera moves every frame (an arc around the scene, there and back) instead of standing still: "
                   "cull results, mips and the tile order's cost feedback (four frames old) then change from frame to frame")
    p.add_argument("--mesh", action="store_true", help="depth / shadow maps rasterised on the GPU from the reference scenes' mesh (bunny.obj at 0 -9 0 x1.8, "
                   "Bin/all16.bat:1) instead of the analytic sphere")
    p.add_argument("--ray-queue", action="store_true", help="MV_FLAG_RAY_QUEUE: the two-stage ray-queue march instead of one wavefront per tile")
    p.add_argument("--no-light", action="store_true", help="DIAGNOSTIC (not a valid bench line): leave the light march out of the timed frames, to price it")
    p.add_argument("--profile-every", type=int, default=8, help="HIP-event pairs around the kernel launches of every n-th timed frame (each pair costs the pipelined frame ~0.5 %%)")
    p.add_argument("--roofline-frames", type=int, default=64, help="frames of the one-at-a-time pass that measures roofline.launch_ms")
    p.add_argument("--cpu-volumes", type=int, default=0, help="volumes the CPU baseline marches (0 = auto-bounded)")
    p.add_argument("--rows-to-root", action="store_true", help="(the default since round 5; kept for old command lines)")
    p.add_argument("--rows-to-all", action="store_true", help="--gpus > 1, peer-store transport: every rank gets every rank's screen rows (as with the all-gathers) instead of rank 0 only "
                   "(SURVEY 8e: 'gather tiles to rank 0' -- a seventh of the row traffic; the default)")
    p.add_argument("--peer-no-acquire", action="store_true", help="DIAGNOSTIC: the folded waits of the peer-store transport without their acquire (round 4's form; MV_DIST_PEER_NO_ACQUIRE)")
    p.add_argument("--peer-push-cells", action="store_true", help="DIAGNOSTIC: the peer-store transport pushes whole light cells (16 B per texel, round 4) instead of plain texels that the receivers rebuild (MV_DIST_PEER_PUSH_CELLS)")
    p.add_argument("--finegrained", action="store_true", help="MV_FLAG_FINEGRAINED_PEER_BUFFERS: the buffers other ranks store into allocated fine-grained (what it costs on one GPU: profiles/r05_finegrained.txt)")
    p.add_argument("--latency-frames", type=int, default=128, help="frames of the latency pass behind the timed region (every 8th sampled); 0 = none")
    p.add_argument("--steady-frames", type=int, default=400, help="frames of the steady-state pass behind the timed region (its own barrier-to-barrier time: `steady_state_ms_per_step`); 0 = none")
    p.add_argument("--cull-on-view-stream", action="store_true", help="MV_FLAG_CULL_ON_VIEW_STREAM: every frame's upload and cull on its own view stream (no stream of their own); "
                   "the default of the peer rung of --gpus > 1, an experiment on one GPU")
    p.add_argument("--frames-in-flight", type=int, default=0, help="whole frames in flight = view streams = output sets (2..5; 0 = the library's default, 3)")
    p.add_argument("--cull-on-stream-a", action="store_true", help="one GPU: three chains and a stream of its own for upload + cull (the library's default) instead of four chains with the culls on them")
    p.add_argument("--frames-per-launch", type=int, default=0, help="--gpus N > 1, peer stores: B = 1..4 consecutive frames go through every long kernel of a rank's chain together "
                   "(mv_config.frames_per_launch; 0 = 4 on the peer-store rung, 1 elsewhere): a rank of eight holds an eighth of a frame, which leaves its GPU to latency")
    p.add_argument("--transport", default="auto", choices=["auto", "peer", "peer-one-frame-per-launch", "peer-no-fold", "rccl", "rccl-single-stream", "torch-nccl", "gloo"],
                   help="--gpus > 1: auto = the peer-store transport (no collectives: every rank's kernels store into every rank's buffers over "
                        "IPC mappings, flags instead of all-gathers), and -- only if that fails its self-test, fails or stops making progress -- "
                        "the next of: the library's own RCCL communicators with one view stream (two communicators, one collective in flight "
                        "per stream), RCCL with whole frames on three view streams, torch.distributed's NCCL group through callbacks, gloo through "
                        "host memory (the line says which one ran: config.transport, config.fallback_level)")
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# Launching and supervising the ranks of a multi-GPU run.
#
# `python bench.py --gpus N` (no WORLD_SIZE in the environment) starts N ranks itself, as CHILD processes
# (`python -m torch.distributed.run ...`), before anything here touches the GPU, relays rank 0's JSON line and exits with the
# ranks' status.  Under torch.distributed.run (the driver's command, or the one just described) every rank process is a SUPERVISOR:
# it never initialises the GPU, it starts the measuring process as a child and watches its progress marks.  The in-library
# multi-GPU loop keeps three RCCL communicators busy on three streams per device, which no box with more than one GPU was available
# to exercise while it was written: if a rank dies or stops making progress, every supervisor kills its child (a fresh process each
# time: nothing is exec'ed from a process that has touched the GPU) and all move to the next, more conservative transport together.
# ---------------------------------------------------------------------------------------------------------------------
# (the conservative RCCL schedule in front of the one that keeps three communicators busy on three streams of a device, until a box with
# more than one GPU has run tests/test_rccl_multi_gpu.py)
# "peer": four frames per launch, waits folded into the kernels; then the same transport with a frame per launch; then with every wait a
# kernel of its own in front of its consumer ("peer-no-fold": a kernel boundary between another GPU's stores and this GPU's loads of them --
# the rung for a node on which the in-kernel coherence self-test of the folded waits fails); then the collectives
LEVELS = ["peer", "peer-one-frame-per-launch", "peer-no-fold", "rccl-single-stream", "rccl", "torch-nccl", "gloo"]
PHASE_TIMEOUT_S = {"start": 420.0, "setup": 420.0, "dist-init": 180.0, "warmup": 240.0, "timed": 240.0, "roofline": 240.0, "report": 600.0}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kill_group(proc):
    import signal
    try:
        os.killpg(proc.pid, signal.SIGKILL)  # the exact process group this supervisor started (start_new_session below)
    except ProcessLookupError:
        pass
    try:
        proc.wait(timeout=30)
    except Exception:
        pass


def launch_ranks(args):
    """No WORLD_SIZE: start --gpus ranks as child processes and relay rank 0's line."""
    import subprocess
    import torch
    if os.environ.get("MV_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < args.gpus:  # (counting devices does not initialise the GPU)
        print(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} GPU(s)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), start_new_session=True)
    try:
        return proc.wait()
    except KeyboardInterrupt:
        _kill_group(proc)
        return 130


def supervise(args, world, rank):
    """One rank's supervisor (see above).  CPU only: gloo between the supervisors, the measuring process is a child."""
    import subprocess
    import threading
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    share = os.environ.get("MV_BENCH_SHARE_GPU") == "1"
    levels = (["peer", "gloo"] if args.transport == "auto" else [args.transport]) if share else LEVELS if args.transport == "auto" else [args.transport]
    if args.dist_single_stream:
        levels = [l for l in levels if not l.startswith("peer")] or levels  # (the peer-store transport runs whole-frame chains only)
    rc = 1
    for level, transport in enumerate(levels):
        port = [_free_port() if rank == 0 else 0]
        dist.broadcast_object_list(port, src=0)
        env = dict(os.environ, MV_BENCH_WORKER="1", MV_BENCH_TRANSPORT=transport, MV_BENCH_LEVEL=str(level), MASTER_PORT=str(port[0]),
                   TORCHELASTIC_USE_AGENT_STORE="False", HSA_ENABLE_IPC_MODE_LEGACY="0")
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True, start_new_session=True)
        state = {"phase": "start", "since": time.monotonic()}

        def pump():
            for line in proc.stderr:
                if line.startswith("MV_PHASE "):
                    state["phase"], state["since"] = line.split()[1], time.monotonic()
                else:
                    sys.stderr.write(line)
        th = threading.Thread(target=pump, daemon=True)
        th.start()
        # Every few seconds the supervisors agree on what they see: everybody done, somebody failed or stuck, or keep waiting.
        while True:
            time.sleep(0.5 if proc.poll() is not None else 2.0)
            code = proc.poll()
            limit = PHASE_TIMEOUT_S.get(state["phase"], 240.0)
            if state["phase"] == "warmup" and os.environ.get("MV_BENCH_WARMUP_TIMEOUT_S") and f"hang:{transport}:" in os.environ.get("MV_BENCH_INJECT", ""):
                limit = float(os.environ["MV_BENCH_WARMUP_TIMEOUT_S"])  # (test hook: the injected hang sits in this rung's warm-up; the other rungs keep their patience)
            stuck = code is None and time.monotonic() - state["since"] > limit
            mine = 0 if code == 0 else 2 if (stuck or (code is not None and code != 0)) else 1  # 0 done, 1 running, 2 failed
            seen = [None] * world
            dist.all_gather_object(seen, (mine, state["phase"]))
            if any(m == 2 for m, _ in seen):
                if rank == 0:
                    print(f"bench.py: transport '{transport}' failed or stopped making progress (rank states {seen}); "
                          + ("trying the next one" if level + 1 < len(levels) else "no transport left"), file=sys.stderr)
                _kill_group(proc)
                rc = 1
                break
            if all(m == 0 for m, _ in seen):
                rc = 0
                break
        th.join(timeout=5)
        if rc == 0:
            break
    dist.barrier()
    dist.destroy_process_group()
    return rc


def phase(name):
    """Progress mark for the supervisor (stderr; a plain run ignores it)."""
    if os.environ.get("MV_BENCH_WORKER") == "1":
        print(f"MV_PHASE {name}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to print a line for another rank count", file=sys.stderr)
        return 2
    if world > 1 and os.environ.get("MV_BENCH_WORKER") != "1":
        return supervise(args, world, rank)
    import torch
    import torch.distributed as dist

    transport = os.environ.get("MV_BENCH_TRANSPORT", "peer" if args.transport == "auto" else args.transport)
    peer_fold = None  # (the library decides: folded when every rank owns its GPU)
    if transport == "peer-one-frame-per-launch":  # (the ladder's second rung: what round 4 measured first -- no launch holds more than one frame)
        transport = "peer"
        if args.frames_per_launch == 0:
            args.frames_per_launch = 1
    if transport == "peer-no-fold":  # (third rung: every wait a one-wavefront kernel in front of its consumer; several frames per launch exist folded only)
        transport, peer_fold = "peer", False
        args.frames_per_launch = 1
    rows_to_root = not args.rows_to_all
    if transport == "rccl-single-stream":
        args.dist_single_stream = True
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the ray-march path has no CPU fallback")
    # MV_BENCH_SHARE_GPU=1 is a functional test mode for boxes with fewer GPUs than ranks: every rank uses cuda:0 and the
    # collectives go through gloo (RCCL refuses two ranks on one device). Never used for reported numbers.
    share_gpu = os.environ.get("MV_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu or transport in ("gloo", "peer"):  # (peer stores: torch.distributed only carries the IPC handles and the final sums -- no RCCL anywhere)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    phase("setup")

    from multivolumes_amd import MultiRayCaster, _lib, scene
    from multivolumes_amd import raycaster as rc

    cfg = dict(scene.CONFIGS[args.config])
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    M = 1024
    flags = _lib.MV_FLAG_FORCE_MIP0_ALL_FACES if args.stress else 0
    if args.finegrained:
        flags |= _lib.MV_FLAG_FINEGRAINED_PEER_BUFFERS
    if args.ray_queue:
        flags |= _lib.MV_FLAG_RAY_QUEUE
    pipelined = not args.no_pipeline
    if pipelined:
        flags |= _lib.MV_FLAG_PIPELINE_FRAMES  # frames in flight, like the reference's upload ring (MultiRayCaster.h:52)
        if world == 1 and args.frames_in_flight == 0 and not args.cull_on_stream_a and not args.ray_queue:
            # four whole frames on the four hardware queues, every frame's upload and cull on its own view stream: 0.3124-0.3144 ms per C3
            # frame against 0.3158-0.3185 with three chains + a stream for the culls (scripts/r04_one_gpu_chains_ab.sh, three alternations)
            args.cull_on_view_stream = True
            args.frames_in_flight = 4
        if args.cull_on_view_stream:
            flags |= _lib.MV_FLAG_CULL_ON_VIEW_STREAM
    if world > 1:
        # the in-library frame loop (mv_dist_frame): whole frames on three view streams, a communicator each; --dist-single-stream
        # selects its other schedule (one view stream, the exchange beside it)
        if not pipelined and rank == 0:
            print("warning: --no-pipeline is ignored with --gpus > 1", file=sys.stderr)
        pipelined = True
        flags |= _lib.MV_FLAG_PIPELINE_FRAMES | (_lib.MV_FLAG_SINGLE_VIEW_STREAM if args.dist_single_stream else 0)
        if transport == "peer" and not args.dist_single_stream and args.frames_in_flight == 0:
            # a rank's frame is a chain of short kernels: four whole frames in flight on the four hardware queues, every frame's upload
            # and cull on its own view stream (profiles/r04_peer_rank_cost.txt)
            flags |= _lib.MV_FLAG_CULL_ON_VIEW_STREAM
            args.frames_in_flight = 4
            if args.frames_per_launch == 0 and not share_gpu:  # (ranks that share a GPU -- tests -- fill it with each other's waiting workgroups at full size: only when asked)
                # ... and four frames per launch on each: sixteen frames in flight per rank (profiles/r04_frames_per_launch.txt:
                # a rank of eight 0.095 -> 0.060 ms per C3 frame, of four 0.129 -> 0.099, of two 0.201 -> 0.175)
                args.frames_per_launch = 4
    if not (world > 1 and transport == "peer" and (flags & _lib.MV_FLAG_CULL_ON_VIEW_STREAM)):
        args.frames_per_launch = 0  # (groups of frames per launch are the peer-store loop's)
    # a real (non-blocking) torch stream, made current so that the RCCL collectives of the multi-GPU path are ordered on it too:
    # torch's default is the legacy null stream, whose implicit synchronisation rules cost the pipelined frame ~20 %
    if os.environ.get("MV_BENCH_NULL_STREAM", "0") != "1":
        torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))
    stream = torch.cuda.current_stream().cuda_stream
    caster = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=M, device=local_rank,
                            stream=stream, flags=flags, shard=(rank, world), framesInFlight=args.frames_in_flight, framesPerLaunch=args.frames_per_launch)

    # ---- synthetic inputs, resident in HBM before the timed region ----
    t0 = time.time()
    vols_host = {}
    for k in range(V):
        v = rc.synth_volume(N, k)
        caster.LoadVolumeData(k, v)
        if rank == 0 and not args.no_cpu_baseline:
            vols_host[k] = v
    fr = scene.setup_raycaster(caster, cfg)
    mesh = scene.load_bunny() if (args.mesh and cfg["mesh"]) else None
    if mesh is not None:
        scene.render_mesh_targets(caster, cfg, fr["viewProj"], mesh)  # replaces the analytic sphere's maps
        fr["depth"] = caster.ReadBackDepthMap()
    # --orbit: ORBIT_N camera positions 0.004 rad apart on an arc through the default eye point, visited there and back; the depth
    # map of every position is made once, kept in HBM, and handed over per frame as a borrowed device pointer (as the reference
    # hands over ObjectRenderer's depth buffer): no upload and no host wait inside the timed frames.
    orbit = None
    if args.orbit:
        ORBIT_N = 128
        e0 = np.asarray(fr["eye"], np.float64)
        r, a0 = float(np.hypot(e0[0], e0[2])), float(np.arctan2(e0[0], -e0[2]))
        proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / float(cfg["H"]), scene.Z_NEAR, scene.Z_FAR)
        orbit = []
        for j in range(ORBIT_N):
            a = a0 + 0.004 * j
            e = np.array([r * np.sin(a), e0[1] * (1.0 + 0.25 * np.sin(3.0 * (a - a0))), -r * np.cos(a)], np.float32)
            vp = rc.multiply(rc.look_at_lh(e, (0.0, 0.0, 0.0)), proj)
            if mesh is not None:
                scene.render_mesh_targets(caster, cfg, vp)
                d = torch.from_numpy(caster.ReadBackDepthMap()).cuda()
            elif cfg["mesh"]:
                d = torch.from_numpy(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0.0, 0.0, 0.0), scene.SPHERE_RADIUS)).cuda()
            else:
                d = torch.ones((cfg["H"], cfg["W"]), dtype=torch.float32, device="cuda")
            orbit.append((e, vp, d))
        if mesh is not None:
            scene.render_mesh_targets(caster, cfg, fr["viewProj"])  # (the shadow map does not depend on the camera)
        torch.cuda.synchronize()
    t_setup = time.time() - t0
    do_composite = not args.no_composite

    from multivolumes_amd import distributed as D
    # N > 1: the per-frame protocol runs inside the library (mv_dist_frame); torch.distributed only distributes the RCCL ids
    phase("dist-init")
    exchange = D.init_frame_loop(caster, prefer_rccl=transport in ("rccl", "rccl-single-stream"), prefer_peer=transport == "peer",
                                 peer_timeout_ms=int(os.environ.get("MV_BENCH_PEER_TIMEOUT_MS", "20000")), peer_rows_to_root=rows_to_root, peer_fold=peer_fold, peer_acquire=not args.peer_no_acquire, peer_push_cells=args.peer_push_cells) if world > 1 else None
    if world > 1 and transport == "peer" and exchange != "peer":
        # (asked for by name or by the ladder: a rank that could not attach it leaves, and the supervisors move everybody to the next level)
        print(f"bench.py: the peer-store transport could not be attached (got {exchange})", file=sys.stderr)
        return 3
    phase("warmup")
    # (test hook, tests/test_bench_multirank.py: MV_BENCH_INJECT="fail:<rung>:<rank>" / "hang:<rung>:<rank>" -- that rank of that rung of the ladder leaves
    # with an error / stops making progress here: the supervisors must move every rank to the next rung together)
    for inj in filter(None, os.environ.get("MV_BENCH_INJECT", "").split(",")):
        how, rung, who = inj.split(":")
        if rung == os.environ.get("MV_BENCH_TRANSPORT", transport) and int(who) == rank:
            print(f"bench.py: injected {how} on rank {rank} of rung {rung}", file=sys.stderr, flush=True)
            if how == "fail":
                return 5
            time.sleep(3600.0)
    H, W = cfg["H"], cfg["W"]
    band = D.row_range(H, rank, world)
    assert band == caster.ShardRows()
    xbytes = [0]
    skip_light = [False]

    nframe = [0]
    last_depth = [None]

    def frame(with_exchange=True):
        e, vp = fr["eye"], fr["viewProj"]
        if orbit is not None:
            j = nframe[0] % (2 * len(orbit) - 2)
            e, vp, d = orbit[j if j < len(orbit) else 2 * len(orbit) - 2 - j]
            nframe[0] += 1
            caster.SetRenderTargets(depthMapDevice=d.data_ptr())
            last_depth[0] = d
        if world > 1:
            # begins this frame (upload, cull, light march + light-map exchange, view march) and finishes the previous one (cube-face
            # all-gather, composite of this rank's rows, row-band all-gather): MultiRayCaster.cpp:355-385 across the ranks
            caster.DistFrame(vp, fr["shadowVP"], e, composite=do_composite)
            return
        caster.UpdateFrame(vp, fr["shadowVP"], e)
        if world == 1 and skip_light[0]:
            caster.cullVolumes()
            caster.rayMarchV()
            caster.m_frameIdx += 1
        elif world == 1:
            caster.Render(composite=False)  # cull -> marchL -> marchV, stream ordered
        if do_composite:
            caster.composite(band[0], band[1])

    # Setup frames: relight every visible volume at least once so the light maps are real -- and get past the HIP runtime's
    # one-time 30-50 ms stall around the 140th frame of a process (see the module docstring) whatever --warmup the caller gives.
    for _ in range(max(V, 176)):
        frame(with_exchange=False)
    skip_light[0] = args.no_light
    for _ in range(args.warmup):
        frame()
    if world > 1:
        caster.DistFlush()  # (several frames per launch: the last group of warm-up frames goes out now, not inside the timed region)
    caster.synchronize()
    caster.ResetStats()
    caster.SetProfiling(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        caster.DistFlush()
        caster.synchronize()
        caster.DistExchangedBytes(reset=True)  # exchange bytes of the timed frames only
    barrier()
    phase("timed")
    t0 = time.perf_counter()
    trace = [] if os.environ.get("MV_BENCH_TRACE") else None
    for i in range(args.steps):
        caster.SetProfiling(i % args.profile_every == 0)  # HIP-event pairs around the launches of a sample of the timed frames
        frame()
        if trace is not None:
            trace.append(time.perf_counter())
    if world > 1:
        caster.DistFlush()  # the last frame's exchange + composite
    barrier()
    if trace:
        d = sorted(((b - a) * 1e3, i + 1) for i, (a, b) in enumerate(zip(trace, trace[1:])))
        print("host ms per frame: median %.3f, largest %s" % (d[len(d) // 2][0], [(round(x, 2), i) for x, i in d[-6:]]), file=sys.stderr)
    dt = time.perf_counter() - t0
    xbytes_timed = caster.DistExchangedBytes() if world > 1 else 0
    caster.SetProfiling(False)
    st = caster.GetStats()

    # steady state: the same loop for --steady-frames more frames with its own barriers.  The driver's 20 timed frames are little more than one
    # depth of the pipeline when sixteen frames are in flight (four chains x four frames per launch): mostly fill and drain.
    dt_steady = None
    if args.steady_frames > 0:
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steady_frames):
            frame()
        if world > 1:
            caster.DistFlush()
        barrier()
        dt_steady = time.perf_counter() - t1
    # latency: a pass of its own (a host function behind the last kernel of every 8th frame holds that frame's stream until it has run: inside
    # the timed region that cost the 20-frame figure 4 %), the same loop, frames in flight as in the timed region
    latency = {"mean_ms": 0.0, "max_ms": 0.0, "samples": 0}
    if args.latency_frames > 0:
        caster.SetLatencySampling(8)
        barrier()
        for i in range(args.latency_frames):
            frame()
        if world > 1:
            caster.DistFlush()
        barrier()
        latency = caster.GetLatency(reset=True)
        caster.SetLatencySampling(0)

    # roofline pass: the same frames, one at a time, so that a launch's HIP-event duration is the kernel's own time
    phase("roofline")
    st_iso = None
    if pipelined and args.roofline_frames > 0:
        caster.ResetStats()
        caster.SetProfiling(True)
        for _ in range(args.roofline_frames):
            frame()
            if world > 1:
                caster.DistFlush()
            caster.synchronize()
        caster.SetProfiling(False)
        st_iso = caster.GetStats()

    # max time over ranks, sum of samples over ranks
    phase("report")
    per_rank_ms, ranks_agree = None, None
    if world > 1:
        # every rank's own time for the K frames (the line's ms_per_step is their maximum), and whether the ranks ended up with the same
        # frame: with the peer-store transport and the all-gathers alike every rank holds the whole image
        import zlib
        mine = [None] * world
        to_root = rows_to_root and exchange == "peer"
        # what EVERY rank holds whoever produced it: every light map and the marched faces of every cube map (and, unless the rows go to
        # rank 0 only, the frame itself) -- a transport that left any rank with other bytes is a failed transport
        info_ = caster.ReadBackVolumeInfo()
        vis_, cube_ = caster.ReadBackVisibleLists()
        crc = 0
        for v in range(V):
            crc = zlib.crc32(caster.ReadBackLightMap(v).tobytes(), crc)
        for v in sorted(int(x) for x in cube_):
            crc = zlib.crc32(caster.ReadBackCubeMap(v, int(info_[v, 0])).tobytes(), crc)
            crc = zlib.crc32(caster.ReadBackCubeDepth(v, int(info_[v, 0])).tobytes(), crc)
        # the frame: with the rows pushed to every rank, every rank's whole image; with the rows to rank 0 only (the default), rank r's OWN tile
        # rows as rank r composited them against the same rows of rank 0's assembled image -- the bytes the composite's peer stores carried
        # (that image, the only complete one, was compared with nothing)
        row_crcs = None
        if do_composite:
            img_ = caster.ReadBackComposite()
            if not to_root:
                crc = zlib.crc32(img_.tobytes(), crc)
            else:
                def rows_crc(r_):
                    c_ = 0
                    for t_ in D.tile_rows_of_rank(H, r_, world):
                        c_ = zlib.crc32(img_[8 * t_:min(8 * t_ + 8, H)].tobytes(), c_)
                    return c_
                row_crcs = [rows_crc(r_) for r_ in range(world)] if rank == 0 else rows_crc(rank)
        # bytes this rank sends per frame (peer stores): its marched texels (8 B colour + 4 B depth) and its layers of the relit light map to
        # every other rank, its screen rows to every other rank or to rank 0
        pinfo = caster.DistPeerInfo()
        rows_mine = len(D.tile_rows_of_rank(H, rank, world)) * 8 * W * 8 if hasattr(D, "tile_rows_of_rank") else (band[1] - band[0]) * W * 8
        out_bytes = ((st["rays"] / max(1, args.steps)) * 12 + (L ** 3 // world) * pinfo["light_push_bytes_per_texel"]) * (world - 1) \
            + rows_mine * ((0 if rank == 0 else 1) if to_root else (world - 1)) if exchange == "peer" else None
        dist.all_gather_object(mine, (dt / args.steps * 1e3, crc, out_bytes, (dt_steady / args.steady_frames * 1e3) if dt_steady else None, latency, row_crcs))
        per_rank_ms = [round(m[0], 5) for m in mine]
        ranks_agree = len(set(m[1] for m in mine)) == 1
        rows_agree = True
        if to_root and do_composite:  # rank 0's assembled image holds every rank's rows as that rank wrote them
            rows_agree = all(mine[0][5][r_] == (mine[r_][5] if r_ else mine[0][5][0]) for r_ in range(world))
        if not ranks_agree or not rows_agree:
            # a transport that delivered different images to different ranks is a failed transport: no line, and the supervisors move on
            if rank == 0:
                what = "the ranks do not hold the same light maps / cube faces" if not ranks_agree else "rank 0's assembled frame does not hold the rows the ranks composited"
                print(f"bench.py: {what} after the run (transport {exchange}): {[hex(m[1]) for m in mine]}"
                      + ("" if rows_agree else f"; rows at rank 0 {[hex(x) for x in mine[0][5]]}, at their ranks {[hex(m[5]) if i else '-' for i, m in enumerate(mine)]}"), file=sys.stderr)
            caster.close()
            return 4
        t = torch.tensor([dt, dt_steady or 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_steady = float(t[0].item()), (float(t[1].item()) or None)
        out_per_rank = [m[2] for m in mine]
        latency = {"mean_ms": max(m[4]["mean_ms"] for m in mine), "max_ms": max(m[4]["max_ms"] for m in mine), "samples": min(m[4]["samples"] for m in mine)}
        cnt = torch.tensor([st["view_samples"], st["lit_samples"], st["rays"], st["light_samples"], st["fetch_free_samples"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_view, tot_lit, tot_rays, tot_light, tot_free = [int(x) for x in cnt.tolist()]
    else:
        tot_view, tot_lit, tot_rays, tot_light, tot_free = st["view_samples"], st["lit_samples"], st["rays"], st["light_samples"], st["fetch_free_samples"]
        out_per_rank = None

    if rank == 0:
        K = args.steps
        info = caster.ReadBackVolumeInfo()
        vis, cube = caster.ReadBackVisibleLists()
        def kernel_ms(s_):
            return {"cull": s_["ms_cull"] / max(1, s_["n_cull"]), "march_light": s_["ms_march_light"] / max(1, s_["n_march_light"]),
                    "march_view": s_["ms_march_view"] / max(1, s_["n_march_view"]), "composite": s_["ms_composite"] / max(1, s_["n_composite"])}
        k_in_flight = kernel_ms(st)
        k_alone = kernel_ms(st_iso) if st_iso else k_in_flight
        ms_v = k_alone["march_view"]
        # rank-0 marchV launch: algorithmic bytes per launch (the counters cover every timed frame) / average launch duration
        # the view-march launch also marches the screen-space RayCast tiles (composite_samples) on single-GPU contexts
        fused = st["composite_samples"]  # the screen-space RayCast samples of this rank's rows, marched by the same launch
        alg_a3 = (64 * st["view_samples"] + 32 * st["lit_samples"] + 16 * st["rays"]) / K  # SURVEY 8(d)'s unit: the a3 loop over cube-face texels
        alg_bytes = (64 * (st["view_samples"] + fused) + 32 * (st["lit_samples"] + st["fused_raycast_lit"]) + 16 * st["rays"]) / K
        achieved = alg_bytes / (ms_v * 1e-3) / 1e9 if ms_v > 0 else 0.0
        # measured HBM traffic of the launch: PMC counters cannot be read from inside this process, so the figure comes from the
        # rocprofv3 passes of scripts/make_profiles.sh -- and only if it was measured on the sources that are in the tree now
        traffic, traffic_note = None, "no measurement under profiles/traffic.json"
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        key = f"{args.config}{'-stress' if args.stress else ''}" + ("-rayqueue" if args.ray_queue else "")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if tj.get("source_hash") != march_source_hash():
                    traffic_note = "profiles/traffic.json was measured on other kernel sources: not reported (rerun scripts/make_profiles.sh)"
                elif key in tj:
                    traffic = tj[key].get("marchViewKernel_hbm_bytes_per_launch")
                    traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/make_profiles.sh on these sources (2 x FETCH_SIZE + WRITE_SIZE)"
            except Exception:
                traffic = None
        # SURVEY 8(d) asks for two more reference points beside the spec peak: what a plain device copy reaches on this box, and the
        # compulsory footprint of the frame (every visible volume and light map read once, outputs written once)
        copy_gbs = None
        try:
            a = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
            b = torch.empty_like(a)
            b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                b.copy_(a)
            e1.record()
            e1.synchronize()
            copy_gbs = 8 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9  # bytes read + bytes written
            del a, b
        except Exception:
            pass
        compulsory = len(vis) * (8 * N ** 3 + 4 * L ** 3) + 12 * (tot_rays // K) + 4 * cfg["W"] * cfg["H"]
        out = {
            "metric": "volume_samples_per_sec", "value": tot_view / dt / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
            "frames_per_s": K / dt, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # depth: whole frames the host may run ahead of the GPU (chains x frames per launch), and what a frame pays for it -- host clock at
            # the hand-over of its constants to host clock behind its last kernel, every 8th frame of a pass of --latency-frames frames behind the timed region (max over ranks)
            "frames_in_flight_total": (2 if (world > 1 and args.dist_single_stream) else (args.frames_in_flight or 3) * (args.frames_per_launch or 1)) if pipelined else 1,
            "latency_ms": latency["mean_ms"] if latency["samples"] else None, "latency_ms_max": latency["max_ms"] if latency["samples"] else None, "latency_samples": latency["samples"],
            # the same loop over --steady-frames more frames, its own barriers (max over ranks): what the K timed frames would say if K were large
            "steady_state_ms_per_step": (dt_steady / args.steady_frames * 1e3) if dt_steady else None, "steady_state_frames": args.steady_frames if dt_steady else 0,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.config}{'-stress' if args.stress else ''}: {V}x{N}^3 RGBA16F volumes, {cfg['W']}x{cfg['H']} viewport, "
                            f"{cfg['maxRay']} steps/ray, light grid {L}^3 x {cfg['maxLight']} light steps"
                            + ((", mesh depth occlusion (bunny mesh rasterised on the GPU)" if args.mesh else ", mesh depth occlusion (analytic sphere)") if cfg["mesh"] else "")
                            + (", orbiting camera" if args.orbit else ""),
                "visible_volumes": int(len(vis)), "cube_map_volumes": int(len(cube)),
                "selected_mips": sorted(set(int(info[v, 0]) for v in vis)),
                "samples_per_ray_cap": [int(info[v, 1]) for v in vis][:4],
                "view_samples_per_frame": tot_view // K, "lane_utilisation": (st["view_samples"] / st["reserved0"]) if st.get("reserved0") else None, "lit_samples_per_frame": tot_lit // K, "rays_per_frame": tot_rays // K,
                "light_samples_per_frame": tot_light // K,
                "fetch_free_samples_per_frame": tot_free // K,
                "frames_per_launch": (args.frames_per_launch or 1) if world > 1 else None,
                "parallelism": (f"{world} ranks, peer stores (no collectives): " + ("every cube-map volume's 8x8 tiles dealt over the ranks" if (args.frames_per_launch or 1) > 1 else "cube-map volumes dealt per frame by estimated march cost") + ", the view march stores its texels into every rank's cube maps; screen-space RayCast and composite dealt by 8-row tile row, " + ("rows stored by the composite into rank 0's frame buffer" if rows_to_root else "rows stored by the composite into every rank's frame buffer") + "; the frame's light march shared out by layers of voxel blocks, " + ("whole cells pushed into every rank's light map" if args.peer_push_cells else "plain texels pushed into every rank's staging map and the cells rebuilt by the receivers") + "; flags in uncached device memory order them; frame loop inside the library (mv_dist_frame)" if exchange == "peer"
                                else f"{world} ranks: cube-map volumes dealt per frame by estimated march cost + all-gather of the marched cube faces, screen-space RayCast and composite dealt by 8-row tile row + in-place all-gather of the rows, the frame's light march shared out by voxel blocks + all-gather of the shares; frame loop inside the library (mv_dist_frame), transport: {exchange}") if world > 1 else "single GPU",
                "frames_in_flight": ((f"up to {args.frames_in_flight or 3} whole frames (" + ("upload, cull, " if args.cull_on_view_stream else "") + f"light march, view march, composite) on {args.frames_in_flight or 3} view streams"
                                      + ("" if args.cull_on_view_stream else "; upload + cull on a stream of their own")) if world == 1
                                     else "upload + cull + light march ahead of the view march / exchange / composite stream" if args.dist_single_stream
                                     else (f"{args.frames_in_flight} chains (view streams) x {args.frames_per_launch or 1} frames per launch: upload, cull, light-march layers, view march (stores its texels and the light layers into every rank's buffers), composite, row push per group of frames") if exchange == "peer"
                                     else "up to 3 whole frames (light-march share + all-gather, view march, cube-face all-gather, composite, row all-gather) on 3 view streams with a communicator each") if pipelined else 1,
                "step": "UpdateFrame+cull+marchL(1 volume)+marchV" + (("+peer stores" if exchange == "peer" else "+cube-face all-gather") if world > 1 else "") + ("+OIT composite" if do_composite else ""),
                "exchange_bytes_per_frame": xbytes_timed // max(1, K),
                # peer stores: what a rank SENDS per frame (the largest rank's; every rank's beside it) and the rate that is at this run's frame time --
                # against 7 links x 76.8 GB/s one way per GPU (DESIGN.md 5.5 has the budget)
                "outbound_bytes_per_frame": int(max(out_per_rank)) if out_per_rank and out_per_rank[0] is not None else None,
                "outbound_bytes_per_frame_per_rank": [int(x) for x in out_per_rank] if out_per_rank and out_per_rank[0] is not None else None,
                "outbound_gb_per_s_at_this_frame_time": (max(out_per_rank) / (dt / K) / 1e9) if out_per_rank and out_per_rank[0] is not None else None,
                "peer": caster.DistPeerInfo() if (world > 1 and exchange == "peer") else None,
                "transport": exchange, "fallback_level": int(os.environ.get("MV_BENCH_LEVEL", "0")),
                "transport_ladder": LEVELS if world > 1 else None,
                "rccl_ranks": caster.DistInfo()["ranks"] if world > 1 else None,  # (the size of the transport's group as the library reports it: ncclCommCount on the RCCL rungs, the mapped peers on the peer-store rungs)
                "transport_ranks": caster.DistInfo()["ranks"] if world > 1 else None,
                "ms_per_step_per_rank": per_rank_ms, "ranks_hold_the_same_frame": ranks_agree,
                "ranks_compared": ("every light map, the marched faces of every cube map" + (", every rank's own rows of the composite against rank 0's assembled frame" if (rows_to_root and exchange == "peer") else ", the composite")) if world > 1 else None,
            },
            "kernel_ms": k_alone, "kernel_ms_in_flight": k_in_flight,
            "composite_msamples_per_s": st["composite_samples"] / dt / 1e6,
            "light_msamples_per_s": tot_light / dt / 1e6,
            "roofline": {"bound": "hbm", "kernel": "marchViewKernel (mv_march.hip ray queue)" if args.ray_queue else "marchTilesKernel",
                         # SURVEY 8(d)'s figure: the a3 loop over cube-face texels -- 64 B per view sample + 32 B per lit sample + 16 B per ray, counted
                         # for the launches of the timed region -- over the launch's HIP-event duration.  A fraction of a roofline cannot exceed 1:
                         # on workloads whose samples are mostly served without HBM (C3-stress: six faces at mip 0, 60 % of the samples fetch-free
                         # or cache hits) the algorithmic figure does, and the line then prices the bytes the launch REQUESTED (frac_basis says
                         # which; the algorithmic one stays beside it as frac_algorithmic_uncapped)
                         **roofline_fraction(alg_a3, alg_bytes - 64 * st.get("fetch_free_samples", 0) / K, traffic, ms_v),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "traffic": traffic, "traffic_source": traffic_note,
                         # `achieved` / `frac` price ALGORITHMIC bytes (64 B per sample whether or not its texels were fetched or came from
                         # cache): they can exceed what HBM delivers.  What HBM really moved per launch is `traffic`:
                         "hbm_achieved": (traffic / (ms_v * 1e-3) / 1e9) if (traffic and ms_v > 0) else None,
                         "hbm_frac": (traffic / (ms_v * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and ms_v > 0) else None,
                         # the same launch also marches the screen-space RayCast tiles of the volumes on that scheme (RayCast.hlsli:43-108): with
                         # their samples priced the same way
                         # (the same rule as `frac`: never above 1 -- C3-stress has no screen-space volume, its fused figure is the a3 one)
                         "frac_with_fused_raycast": roofline_fraction(alg_bytes, alg_bytes - 64 * st.get("fetch_free_samples", 0) / K, traffic, ms_v)["frac"],
                         "frac_with_fused_raycast_algorithmic_uncapped": achieved / HBM_PEAK_GBS,
                         # share of the launch's samples (cube-face + RayCast) whose brick occupancy bit proved alpha <= ZERO_THRESHOLD: counted and
                         # priced at 64 B like the others (the reference's loop reads them), but no texel was requested for them
                         "fetch_free_share": (st.get("fetch_free_samples", 0) / max(1, st["view_samples"] + fused)),
                         "algorithmic_bytes_per_launch": alg_a3, "algorithmic_bytes_with_fused_raycast_per_launch": alg_bytes, "launch_ms": ms_v,
                         "requested_bytes_per_launch": alg_bytes - 64 * st.get("fetch_free_samples", 0) / K,
                         "compulsory_footprint_bytes": compulsory, "device_copy_measured": copy_gbs, "device_copy_unit": "GB/s (read + write)",
                         "launch_ms_in_flight": k_in_flight["march_view"],
                         "launch_ms_from": (f"{args.roofline_frames} frames run one at a time after the timed region (HIP events)" if st_iso
                                            else f"every {args.profile_every}th frame of the timed region (HIP events)")},
            "setup_s": t_setup,
        }
        if args.no_light:
            out["INVALID"] = "--no-light: the light march was left out of the timed frames (diagnostic run)"
        if not args.no_cpu_baseline:
            if last_depth[0] is not None:
                fr["depth"] = last_depth[0].cpu().numpy()  # the depth map of the frame the buffers hold
            out["cpu_baseline"] = cpu_baseline(args, cfg, caster, fr, info, cube, vols_host)
        print(json.dumps(out))
    caster.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, cfg, caster, fr, info, cube, vols_host):
    """Time the CPU oracle's marchV on (a bounded part of) the same frame, and check the GPU against it."""
    from oracle import mv_oracle as o
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    # (sharded: after the frame's exchange every rank holds every cube map, whoever marched it -- the check then covers the exchange too)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    mine = sorted(int(v) for v in cube)
    n = args.cpu_volumes or len(mine)
    sel = mine[:n]
    if not sel:
        return None
    lms = np.zeros((V, L, L, L), np.uint32)
    for v in sel:
        lms[v] = caster.ReadBackLightMap(v)
    # (what every thread reads at random -- volumes, light maps -- spread over the memory of all sockets: first-touch copies)
    spread = {k: o.spread(v) for k, v in vols_host.items()}
    vols = [spread.get(k, spread[sel[0]]) for k in range(V)]
    lms = o.spread(lms)
    chain = o.cube_chain_texels(N)
    cm = np.zeros((V, chain, 4), np.uint16)
    cd = np.zeros((V, chain), np.float32)
    o.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, fr["depth"], caster.perObject, caster.cbPerFrame, info, sel[:1], cm, cd, inplace=True)  # warm caches / thread pool
    # a bounded sample of the same frame's marchV: the whole frame on every thread, repeated until ~3 s have passed (the oracle marches
    # C3's 17 M samples in a fraction of a second on 128 threads), then one volume on ONE thread for the per-thread rate
    lib = o.load()
    budget = CPU_BUDGET  # affinity and cgroup quota: the box shows 256 CPUs and grants 16; threads beyond the grant are throttled
    threads = min(lib.mvo_num_threads(), budget) if "OMP_NUM_THREADS" not in os.environ else lib.mvo_num_threads()
    lib.mvo_set_num_threads(threads)
    reps, dt, st = 0, 0.0, None
    t0 = time.perf_counter()
    while reps == 0 or (world == 1 and dt < 3.0 and reps < 64):  # (N > 1: one pass -- there the leg is the oracle CHECK of what the transport delivered to rank 0, not a timing)
        _, _, st = o.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, fr["depth"], caster.perObject, caster.cbPerFrame, info, sel, cm, cd, inplace=True)
        reps += 1
        dt = time.perf_counter() - t0
    dt /= reps
    def marched(a, v):
        # the faces this frame marched (VolumeInfo.MaskBits): the others keep what an earlier frame left there -- the reference never clears
        # its cube maps -- which a moving camera makes differ from the oracle's zero-initialised buffer
        return a[[f for f in range(6) if (int(info[v, 2]) >> f) & 1]]
    worst = 0.0
    for v in sel:
        mip = int(info[v, 0])
        exp = marched(o.cube_face_view(cm[v], N, mip), v).view(np.float16).astype(np.float32)
        got = marched(caster.ReadBackCubeMap(v, mip), v).view(np.float16).astype(np.float32)
        worst = max(worst, float(np.max(np.abs(exp - got))))
    lib.mvo_set_num_threads(1)
    t1 = time.perf_counter()
    _, _, st1 = o.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, fr["depth"], caster.perObject, caster.cbPerFrame, info, sel[:1], np.zeros_like(cm), np.zeros_like(cd), inplace=True)
    dt1 = time.perf_counter() - t1
    lib.mvo_set_num_threads(threads)
    single = st1["view_samples"] / dt1 / 1e6
    rate = st["view_samples"] / dt / 1e6
    port = {"value": rate, "unit": "Msamples/s", "cores": threads, "kind": "port", "host_cpus": os.cpu_count(), "cpu_budget": budget,
            "build": "oracle/mv_oracle.cpp, g++ -O3 -ffp-contract=off -fopenmp",
            "sample": f"marchV of {len(sel)} of the {len(cube)} cube-map volumes of the same frame ({st['view_samples']} samples) x {reps} repetitions, {dt * reps:.1f} s; "
                      f"single thread: one volume ({st1['view_samples']} samples, {dt1:.1f} s)",
            "single_thread": single, "parallel_efficiency": rate / (single * threads) if single > 0 else None,
            "gpu_vs_oracle_max_abs_err": worst}
    # The reference's own code on the host cores, when oracle/_ref is there (built from the reference's shipped CSRayMarchV.cso through LLVM,
    # oracle/dxil: its compiled DXIL on the CPU, the stand-in for the DX12 WARP path north_star asks for): the same volumes of the same
    # frame, all cores.  Its cube maps must equal the oracle's bit for bit (they do: tests/test_oracle_vs_reference_dxil.py).
    try:
        from oracle import dxref
        if world == 1 and dxref.available():  # (N = 1 only: the multi-rank runs keep the oracle check of rank 0's cube maps and nothing else)
            cores = threads
            dxref.load().dxref_set_threads(cores)
            dxref.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, fr["depth"], caster.perObject, caster.cbPerFrame, info, sel[:1])  # warm
            reps_r, dtr = 0, 0.0
            t0 = time.perf_counter()
            while dtr < 3.0 and reps_r < 16:  # (a bounded sample, like the oracle's: the whole marchV, repeated for about 3 s)
                rm, _ = dxref.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, fr["depth"], caster.perObject, caster.cbPerFrame, info, sel)
                reps_r += 1
                dtr = time.perf_counter() - t0
            dtr /= reps_r
            same = all(np.array_equal(rm[v], cm[v]) for v in sel)
            worst_ref = 0.0
            for v in sel:  # the GPU's cube maps against the reference's own, no oracle in between
                mip = int(info[v, 0])
                exp = marched(o.cube_face_view(rm[v], N, mip), v).view(np.float16).astype(np.float32)
                got = marched(caster.ReadBackCubeMap(v, mip), v).view(np.float16).astype(np.float32)
                worst_ref = max(worst_ref, float(np.max(np.abs(exp - got))))
            # one group-worker thread (the lanes of a wavefront still run as threads that meet at every wave intrinsic: the emulation's cost)
            dxref.load().dxref_set_threads(1)
            t1 = time.perf_counter()
            dxref.march_view(V, N, L, cfg["W"], cfg["H"], vols, lms, fr["depth"], caster.perObject, caster.cbPerFrame, info, sel[:1])
            dtr1 = time.perf_counter() - t1
            dxref.load().dxref_set_threads(cores)
            single_ref = st1["view_samples"] / dtr1 / 1e6
            rate_ref = st["view_samples"] / dtr / 1e6
            return {"value": rate_ref, "unit": "Msamples/s", "cores": min(cores, 256), "kind": "reference", "host_cpus": os.cpu_count(), "cpu_budget": budget,
                    "build": "oracle/_ref/libmv_dxref.so: the reference's shipped CSRayMarchV.cso (DXIL) translated to x86-64 by LLVM (clang -O1), "
                             "resource intrinsics answered by oracle/dxil/dxil_rt.c, thread groups spread over the host threads; inside a group the 64 lanes of a "
                             "wavefront run as threads that rendezvous on a mutex at every wave intrinsic (the emulation's cost, not the shader's)",
                    "sample": f"marchV of {len(sel)} of the {len(cube)} cube-map volumes of the same frame ({st['view_samples']} samples) x {reps_r} repetitions, {dtr * reps_r:.1f} s; "
                              f"one group-worker thread: one volume ({st1['view_samples']} samples, {dtr1:.1f} s)",
                    "single_thread": single_ref, "parallel_efficiency": rate_ref / (single_ref * min(cores, 256)) if single_ref > 0 else None,
                    "reference_equals_oracle": bool(same), "gpu_vs_reference_max_abs_err": worst_ref, "gpu_vs_oracle_max_abs_err": worst,
                    "port": {k: port[k] for k in ("value", "cores", "build", "single_thread", "parallel_efficiency", "sample")}}
    except Exception as e:  # the checker's checker must never take the bench line down
        port["reference_error"] = str(e)[:200]
    return port


if __name__ == "__main__":
    sys.exit(main() or 0)
