"""The multi-GPU frame loop inside the library (mv_dist_frame): a moving-camera sequence on 2 and 3 ranks (sharing cuda:0, gloo
collectives handed to the library as callbacks) must reproduce the single-GPU composites, light maps and counters bit for bit --
including the frames in which a volume changes between the cube-map and the screen-space scheme (every rank marches its own rows of
such a volume and needs its light map whoever relit it)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


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
        try:
            out, err = p.communicate(timeout=60)
        except subprocess.TimeoutExpired:  # (ranks stuck in a GPU kernel can take long to die: do not wait for their pipes)
            out, err = "", "(the killed ranks have not closed their pipes)"
        raise AssertionError(f"timed out after {timeout} s; stderr tail:\n{err[-3000:]}\nstdout tail:\n{out[-1000:]}")
    if p.returncode != 0 and _retry and "EADDRINUSE" in err and "--master-port" in cmd:
        cmd = list(cmd)
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        return run_group(cmd, timeout, env, _retry=False)
    return subprocess.CompletedProcess(cmd, p.returncode, out, err)



def single_gpu(world, frames, every, static_depth=False, all_frames=False):
    from multivolumes_amd import MultiRayCaster, _lib, scene
    from multivolumes_amd import raycaster as rc
    from tests.dist_worker import apply_setter, camera_sequence, setter_script, test_config, test_volumes
    cfg = test_config(world)
    V = cfg["V"]
    vols = test_volumes(cfg)
    with MultiRayCaster(cfg["N"], cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64) as c:
        fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
        imgs, infos = [], []
        script = setter_script(frames, every, cfg)  # (MV_TEST_SETTER_SEED: the setters the ranks make between their frames, frames without a composite)
        for f, (e, vp) in enumerate(camera_sequence(cfg, frames)):
            for what, arg in script[f][0]:
                apply_setter(c, cfg, fr, what, arg)
            if not static_depth or f == 0:
                c.SetRenderTargets(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), scene.SPHERE_RADIUS))
            c.UpdateFrame(vp, fr["shadowVP"], e)
            c.Render(composite=script[f][1])
            if all_frames or f % every == every - 1 or f == frames - 1:
                imgs.append(c.ReadBackComposite())
                infos.append(c.ReadBackVolumeInfo())
        st = c.GetStats()
        lms = np.stack([c.ReadBackLightMap(v) for v in range(V)])
    return np.stack(imgs), np.stack(infos), np.array([st[k] for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples")]), lms


@pytest.mark.parametrize("world,mode,every,static_depth", [(2, "chains", 3, False), (3, "chains", 13, True), (2, "chains", 26, True),
                                                           (2, "single", 3, False), (3, "single", 13, True),
                                                           (2, "peer", 3, False), (3, "peer", 13, True), (2, "peer", 26, True), (5, "peer", 13, True),
                                                           (2, "peer-folded", 3, False), (3, "peer-folded", 13, True), (2, "peer-folded", 26, True),
                                                           (2, "peer-four", 3, False), (3, "peer-four", 13, True), (2, "peer-four", 26, True), (3, "chains-four", 13, True),
                                                           (3, "peer-root", 13, True), (3, "peer-tiles", 13, True), (2, "peer-tiles", 3, False), (3, "peer-x2-volumes", 13, True),
                                                           (2, "peer-x2", 3, False), (3, "peer-x2", 13, True), (2, "peer-x2", 26, True), (2, "peer-x3", 13, True), (3, "peer-x4", 7, True),
                                                           (5, "peer-x4", 13, True),
                                                           # ranks that own NO layer of the light map and whose slab would start beyond it (B = 8 block layers,
                                                           # two per rank: ranks 4.. idle, rank 5's slab "starts" at z = 40 > L = 32; round 5's ADVICE: the
                                                           # rebuild's range wrapped there); L = 24 on five ranks: rank 4 at z = 32 > 24
                                                           (6, "peer", 13, True), (7, "peer-x2", 13, True), (5, "peer-folded-L24", 13, True)])
def test_dist_frame_loop_matches_single_gpu(tmp_path, world, mode, every, static_depth, monkeypatch):
    """Both schedules of the loop (include/multivolumes.h): whole frames on three view streams with a communicator each ("chains"),
    and one view stream with the exchange beside it ("single").  With `static_depth` nothing inside the sequence drains the frames
    in flight (a depth-map setter does), so up to three frames -- collectives included -- really overlap between checkpoints.
    "peer": the chains schedule on the peer-store transport -- no collectives, every rank's kernels store into every rank's buffers
    through IPC mappings, flags instead of all-gathers (csrc/mv_peer.hip).  "peer-folded": the form ranks that own their GPU run -- the
    waits folded into the marches and the composite -- forced here on two and three ranks that share one.  "peer-xB": B frames per
    launch (mv_config.frames_per_launch): a group of B frames goes through one light march, one view march, one composite and one row
    push (blockIdx.y = the frame); checkpoints that do not fall on a group's end close groups early."""
    frames = 26
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode.endswith("-L24"):
        mode = mode[:-len("-L24")]
        env["MV_TEST_L"] = "24"
        monkeypatch.setenv("MV_TEST_L", "24")  # (single_gpu() below reads the same config)
    if mode == "peer-root":  # the rows to rank 0 only (it is rank 0 whose frames the test compares)
        mode, env["MV_TEST_ROWS_TO_ROOT"] = "peer", "1"
    if mode == "peer-tiles":  # one frame per launch with every cube-map volume's tiles shared out (MV_DIST_PEER_DEAL_TILES; the default there: whole volumes per rank)
        mode, env["MV_TEST_DEAL_VOLUMES"] = "peer", "0"
    if mode.endswith("-volumes"):  # several frames per launch with whole volumes per rank (MV_DIST_PEER_DEAL_VOLUMES; the default there: tiles)
        mode, env["MV_TEST_DEAL_VOLUMES"] = mode[:-len("-volumes")], "1"
    if "-x" in mode:  # mv_config.frames_per_launch = 2, 3, 4: groups of frames through every long kernel together (four chains, folded flags); a checkpoint or a setter closes a group early
        env["MV_TEST_FRAMES_PER_LAUNCH"] = mode[-1]
        mode, env["MV_TEST_FOUR_CHAINS"] = "peer-folded", "1"
    if mode.endswith("-four"):  # MV_FLAG_CULL_ON_VIEW_STREAM + four frames in flight: upload and cull on the frames' own view streams
        env["MV_TEST_FOUR_CHAINS"] = "1"
    if mode.startswith("peer"):
        env["MV_TEST_PEER_FOLD"] = "1" if mode == "peer-folded" else "0"
        mode, env["MV_TEST_TRANSPORT"] = "chains", "peer"
    mode = mode.replace("-four", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), mode, "1" if static_depth else "0"]
    r = run_group(cmd, 300, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, static_depth)
    assert np.array_equal(d["infos"], infos), "the cull is replicated"
    schemes = {tuple((infos[i][:, 2] & 0x8000).tolist()) for i in range(len(infos))}
    assert len(schemes) > 1 or len(infos) == 1, "the sequence must move volumes between the cube-map and the screen-space scheme"
    for i in range(len(imgs)):
        assert np.array_equal(d["imgs"][i], imgs[i]), f"composite of checkpoint {i} differs ({(d['imgs'][i] != imgs[i]).sum()} halfs)"
    assert np.array_equal(d["lms"], lms), "every rank holds every light map, current"
    assert np.array_equal(d["counts"], counts), "the ranks together march exactly the single-GPU samples"
    assert int(d["nbytes"]) > 0


@pytest.mark.parametrize("world,viewport", [(2, None), (5, None), (4, (250, 173))])
def test_in_kernel_coherence_selftest(tmp_path, world, viewport):
    """The self-test of what the folded waits rest on (mv_dist_peer_coherence; run by mv_dist_selftest whenever the waits are folded): ONE
    launch per rank reads every rank's region, waits inside the kernel, every rank overwrites its region everywhere, the launch waits
    inside the kernel again, acquires as the folded waits do and reads once more with plain loads.  The ranks share cuda:0 here, but their
    workgroups sit on eight XCDs with an L2 each, and on 256 CUs with an L1 each that no store of another CU ever refreshes: a wait without
    the acquire DOES see stale words on this one GPU (reported, not asserted: how many depends on what stayed resident) -- with it, none.
    The odd viewport puts the ranks' regions on odd workgroup boundaries: a word's writer and its reader are workgroups of different XCDs on
    the writing rank too (round 5's soak: the test's own plain-stored copy read stale there and refused a sound transport)."""
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MV_TEST_TRANSPORT="peer", MV_TEST_PEER_FOLD="1", MV_TEST_COHERENCE="1")
    if viewport:
        env.update(MV_TEST_W=str(viewport[0]), MV_TEST_H=str(viewport[1]), MV_TEST_FOUR_CHAINS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, "4", "4", "chains", "1"]
    r = run_group(cmd, 300, env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("COHERENCE")]
    assert line, r.stdout[-2000:]
    print(line[0])
    assert " acquire stale 0 first_bad 0 " in line[0] + " ", line[0]
    assert " first_bad 0" in line[0].split("|")[1], line[0]


@pytest.mark.parametrize("mode", ["chains", "single", "peer"])
def test_eight_ranks_at_c3_size_match_single_gpu(tmp_path, mode, monkeypatch):
    """BASELINE.json configs[3] (C4) at its real shape: the full C3 frame -- 16 x 256^3, 1024^2, 128 steps/ray, mesh depth -- through
    mv_dist_frame on EIGHT ranks (sharing cuda:0, gloo callbacks: RCCL refuses two ranks on one device; 8 x ~7 GiB of the 288 GB), both
    schedules, a 14-frame orbit + dolly with nothing that drains the frames in flight: composites, every light map (each marched by
    all eight ranks, an eighth of the voxel blocks each), VolumeInfo and the summed counters equal the single-GPU run bit for bit."""
    monkeypatch.setenv("MV_TEST_CONFIG", "C3")
    world, frames, every = 8, 14, 7
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MV_TEST_CONFIG="C3", OMP_NUM_THREADS="4")
    if mode == "peer":  # (eight processes, each mapping the other seven's light maps, cube maps and frame buffers)
        mode, env["MV_TEST_TRANSPORT"] = "chains", "peer"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), mode, "1"]
    r = run_group(cmd, 900, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, True)
    assert np.array_equal(d["infos"], infos), "the cull is replicated"
    for i in range(len(imgs)):
        assert imgs[i].shape == (1024, 1024, 4)
        assert np.array_equal(d["imgs"][i], imgs[i]), f"composite of checkpoint {i} differs ({(d['imgs'][i] != imgs[i]).sum()} halfs)"
    assert np.array_equal(d["lms"], lms), "every rank holds every light map, current"
    assert np.array_equal(d["counts"], counts) and counts[0] > 10_000_000 * frames, "the ranks together march exactly the single-GPU samples"
    # per frame: the light shares (8 x 1 728 blocks x 256 B), the marched cube faces, the rows of the image
    assert int(d["nbytes"]) // frames > 8 * 1024 * 1024


def test_one_rank_at_c3_size_four_frames_per_launch(tmp_path, monkeypatch):
    """mv_config.frames_per_launch = 4 at the full C3 frame: the launches that hold four frames each (light march, view march, composite:
    blockIdx.y = the frame), sixteen frames in flight, on ONE rank through mv_dist_frame -- bit for bit the frames of the plain single-GPU
    path.  (Ranks of this form that SHARE a GPU only work while their launches leave it mostly empty -- the small configs above: the
    tiles that read a map relit in their launch spin until the other ranks' layers land, and two ranks' worth of a full C3 frame fills
    the GPU with them.  On a GPU of its own a rank of eight has a quarter of the GPU's workgroup slots in such a launch.)"""
    monkeypatch.setenv("MV_TEST_CONFIG", "C3")
    world, frames, every = 1, 14, 7
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MV_TEST_CONFIG="C3", OMP_NUM_THREADS="4", MV_TEST_TRANSPORT="peer",
               MV_TEST_FRAMES_PER_LAUNCH="4", MV_TEST_FOUR_CHAINS="1", MV_TEST_PEER_FOLD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), "chains", "1"]
    r = run_group(cmd, 600, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, True)
    assert np.array_equal(d["infos"], infos)
    for i in range(len(imgs)):
        assert np.array_equal(d["imgs"][i], imgs[i]), f"composite of checkpoint {i} differs ({(d['imgs'][i] != imgs[i]).sum()} halfs)"
    assert np.array_equal(d["lms"], lms)
    assert np.array_equal(d["counts"], counts) and counts[0] > 10_000_000 * frames


@pytest.mark.parametrize("mode", ["chains", "single", "peer", "peer-x2"])
def test_ranks_that_were_handed_different_frames_get_an_error_not_a_hang(tmp_path, mode):
    """The sizes of a frame's collectives follow from the replicated cull; ranks whose callers passed different frame constants would
    enter collectives that do not match.  Every rank's share of the frame's light exchange carries what its cull decided, the unpack
    compares them, and the NEXT mv_dist_frame -- before it sizes that frame's cube-face all-gather -- fails on every rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "peer-x2":  # (two frames per launch: the group that holds the divergent frame is compared as a whole)
        env.update(MV_TEST_FRAMES_PER_LAUNCH="2", MV_TEST_FOUR_CHAINS="1", MV_TEST_PEER_FOLD="1")
        mode = "peer"
    if mode == "peer":  # (the light stage's wait compares the ranks' cull words there)
        mode, env["MV_TEST_TRANSPORT"] = "chains", "peer"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path / "unused.npz"), "12", "1", mode, "3"]
    r = run_group(cmd, 300, env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("divergence reported at frame") == 2, r.stdout[-2000:] + r.stderr[-2000:]


_SLOW = pytest.mark.skipif(os.environ.get("MV_SLOW_TESTS") != "1", reason="MV_SLOW_TESTS=1 (passes: docs/STATUS.md)")


@pytest.mark.parametrize("mode", [pytest.param("peer", marks=_SLOW), pytest.param("peer-folded", marks=_SLOW), "peer-x4"])
def test_a_stalled_rank_is_reported_soon_after_the_timeout(tmp_path, mode):
    """Rank 1's host stops handing over frames behind frame 8 (it stays alive, its buffers mapped).  Rank 0's waits -- a one-wavefront kernel, or
    the prologue of EVERY workgroup of the consumer when the waits are folded -- give up after the transport's timeout (2 s here) and the next call
    returns MV_ERR_TIMEOUT naming the rank.  Round 6 (first run of this test): the report said "the ranks' culls disagree" (a wait that gave up went on
    to compare cull words the stalled rank never wrote; timeouts are now looked at first), or named rank 0 itself (a wait for this rank's own rebuild of
    the light cells, whose kernel was waiting for rank 1: local rows now have twice the patience, and a report that names another rank comes first);
    a give-up is also seen by the workgroups and launches of the slot dispatched behind it (mv_kernels.hpp peerGaveUp)."""
    import re
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MV_TEST_TRANSPORT="peer", MV_TEST_PEER_TIMEOUT_MS="2000", MV_TEST_STALL_S="12", MV_TEST_W="1024", MV_TEST_H="1024")
    if mode == "peer-x4":
        env.update(MV_TEST_FRAMES_PER_LAUNCH="4", MV_TEST_FOUR_CHAINS="1", MV_TEST_PEER_FOLD="1")
    elif mode == "peer-folded":
        env.update(MV_TEST_FOUR_CHAINS="1", MV_TEST_PEER_FOLD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path / "unused.npz"), "40", "1", "chains", "4"]
    r = run_group(cmd, 200, env)
    assert r.returncode == 0, r.stderr[-3000:]
    m = re.search(r"rank 0: stall reported at frame (\d+) after ([0-9.]+) s", r.stdout)
    assert m, r.stdout[-2000:] + r.stderr[-2000:]
    print(m.group(0))
    assert float(m.group(2)) < 15.0, m.group(0)  # (the timeout, plus the frames rank 0 could still finish, plus the groups in flight)


@pytest.mark.parametrize("mode", ["chains", "single", "peer", "peer-x3"])
def test_input_setters_inside_the_loop_leave_the_pending_frame_its_inputs(tmp_path, mode):
    """mv_dist_frame(k) defers frame k - 1's exchange, composite and row gather to the next call.  A caller that hands over a new host
    depth map before every frame (SetRenderTargets(depth_k); DistFrame(k)) must still get frame k - 1 composited against depth_{k-1}:
    the setter enqueues the pending tail before it overwrites the map.  Every frame is read back right after the NEXT frame's setter,
    with no flush anywhere, and compared with the single-GPU frame."""
    frames, world = 12, 2
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if mode == "peer-x3":  # (three frames per launch: the setter closes the group it interrupts -- here every group holds one frame)
        env.update(MV_TEST_FRAMES_PER_LAUNCH="3", MV_TEST_FOUR_CHAINS="1", MV_TEST_PEER_FOLD="1")
        mode = "peer"
    if mode == "peer":
        mode, env["MV_TEST_TRANSPORT"] = "chains", "peer"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), "1", mode, "2"]
    r = run_group(cmd, 300, env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, 1, all_frames=True)
    assert len(d["imgs"]) == frames == len(imgs)
    for i in range(frames):
        assert np.array_equal(d["infos"][i], infos[i])
        assert np.array_equal(d["imgs"][i], imgs[i]), f"frame {i} was not composited against its own depth map ({(d['imgs'][i] != imgs[i]).sum()} halfs differ)"
    assert np.array_equal(d["lms"], lms) and np.array_equal(d["counts"], counts)


def test_rccl_transport_on_one_rank():
    """What can be checked of the RCCL transport on a one-GPU box: the library opens librccl itself, creates its two communicators
    from the ids (world size 1), and every collective the frame loop uses passes the verified self-test; a short mv_dist_frame
    sequence then equals the plain single-GPU frames.  (More than one rank per GPU is refused by RCCL: the protocol itself is
    tested over gloo above; 8 physical GPUs are the driver's to launch.)"""
    import torch.distributed as dist
    from multivolumes_amd import MultiRayCaster, _lib, scene
    from multivolumes_amd import raycaster as rc
    from multivolumes_amd import distributed as D
    from tests.dist_worker import camera_sequence
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = dict(scene.CONFIGS["T2"], W=256, H=168)
        V = cfg["V"]
        vols = [rc.synth_volume(cfg["N"], k) for k in range(V)]
        frames = 9
        flags = _lib.MV_FLAG_PIPELINE_FRAMES  # whole frames on three view streams, three communicators
        with MultiRayCaster(cfg["N"], cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, shard=(0, 1)) as c:
            fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
            assert D.init_frame_loop(c) == "rccl"
            for f, (e, vp) in enumerate(camera_sequence(cfg, frames)):
                c.SetRenderTargets(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), scene.SPHERE_RADIUS))
                c.DistFrame(vp, fr["shadowVP"], e)
            c.DistFlush()
            got = c.ReadBackComposite()
        imgs, _, _, _ = single_gpu(2, frames, frames)
        assert np.array_equal(got, imgs[-1])
        # the fallback init_frame_loop takes when RCCL fails on any rank: detach (mv_dist_shutdown), attach the other transport
        # (here on the other schedule: one view stream, two communicators)
        flags |= _lib.MV_FLAG_SINGLE_VIEW_STREAM
        with MultiRayCaster(cfg["N"], cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, shard=(0, 1)) as c:
            fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
            assert D.init_frame_loop(c) == "rccl"
            c.DistShutdown()
            c.DistShutdown()  # nothing attached: a no-op
            assert D.init_frame_loop(c, prefer_rccl=False) == "torch-callbacks"
            for f, (e, vp) in enumerate(camera_sequence(cfg, frames)):
                c.SetRenderTargets(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), scene.SPHERE_RADIUS))
                c.DistFrame(vp, fr["shadowVP"], e)
            c.DistFlush()
            assert np.array_equal(c.ReadBackComposite(), imgs[-1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V,world", [(16, 4), (64, 8), (64, 3), (4, 8), (1, 2), (9, 5), (33, 7)])  # (also fewer volumes than ranks)
def test_the_culls_deal_equals_its_host_restatement(V, world):
    """What cullKernel's last workgroup decides inside the loop -- the owner of every cube-map volume (volumeFlags bits 16-31), the
    layout of the frame's cube-face exchange (exchangeOff) -- equals multivolumes_amd/distributed.py's numpy restatement of the rule
    (which tests/test_distributed_cpu.py runs as a protocol over gloo), on every rank."""
    import torch
    from multivolumes_amd import MultiRayCaster, _lib, scene
    from multivolumes_amd import raycaster as rc
    from multivolumes_amd import distributed as D
    from tests.helpers import replicating_allgather
    cfg = dict(scene.CONFIGS["T2"], V=V, W=256, H=168)
    N, L = cfg["N"], cfg["L"]
    vols = [rc.synth_volume(N, k % 5) for k in range(V)]
    owners = []
    for rank in range(world):
        flags = _lib.MV_FLAG_PIPELINE_FRAMES
        with MultiRayCaster(N, L, V, 5, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, shard=(rank, world)) as c:
            fr = scene.setup_raycaster(c, cfg, vols[:5])
            c.DistInitCallbacks(replicating_allgather(world), lambda b, n, st, w: 0)  # (a rank alone: its own share in every slot)
            c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            c.DistFlush()
            c.synchronize()
            info = c.ReadBackVolumeInfo()
            vis, cube = c.ReadBackVisibleLists()
            pf, nb = c.GetBuffer(_lib.MV_BUF_VOLUME_FLAGS)
            po_, nb2 = c.GetBuffer(_lib.MV_BUF_EXCHANGE_OFFSETS)
            vflags = D._device_view(pf, nb).view(torch.int32).cpu().numpy().view(np.uint32)
            offs = D._device_view(po_, nb2).view(torch.int32).cpu().numpy().view(np.uint32)
        owner = D.deal_cube_volumes(info, vis, V, N, world)
        assert np.array_equal(vflags >> 16, owner), (rank, (vflags >> 16).tolist(), owner.tolist())
        off, biggest = D.compact_exchange_layout(info, vis, owner, V, N, world)
        for v in (int(x) for x in cube):
            assert offs[v] == off[v], (rank, v)
        owners.append(owner)
        assert len(cube) >= (2 if V >= 9 else 0)  # (the scene must give the deal something to do; the small cases are about fewer volumes than ranks)
    assert all(np.array_equal(owners[0], o) for o in owners), "every rank makes the same deal"
    assert len(set(int(owners[0][int(v)]) for v in cube)) >= min(world, len(cube)) - 1
