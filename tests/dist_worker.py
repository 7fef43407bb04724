"""Worker of tests/test_dist_frame_loop.py: one rank of a moving-camera sequence through the in-library multi-GPU frame loop
(mv_dist_frame) with gloo collectives (ranks share cuda:0; RCCL refuses two ranks on one device) -- or, MV_TEST_TRANSPORT=peer, with the
peer-store transport (IPC mappings between the rank processes: works with every rank on one GPU).  Rank 0 saves the composites."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def camera_sequence(cfg, frames):
    from multivolumes_amd import raycaster as rc, scene
    eye0 = np.array(scene.camera(cfg)[0], np.float32)
    proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0)
    out = []
    for f in range(frames):
        # orbit + dolly: volumes change scheme (cube map <-> screen-space RayCast), mip and visibility along the way
        a = 0.11 * f
        s = 1.0 - 0.55 * np.sin(0.35 * f) ** 2
        e = np.array([eye0[0] * s + 30.0 * np.sin(a), eye0[1] * s, eye0[2] * s * np.cos(0.5 * a)], np.float32)
        vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
        out.append((e, vp))
    return out


def test_config(world):
    """T2 at a small viewport (MV_TEST_W / MV_TEST_H: scripts/soak_dist.py with odd viewports), or a whole BASELINE config (MV_TEST_CONFIG=C3)."""
    from multivolumes_amd import scene
    name = os.environ.get("MV_TEST_CONFIG")
    if name:
        return dict(scene.CONFIGS[name])
    cfg = dict(scene.CONFIGS["T2"], W=int(os.environ.get("MV_TEST_W", "256")), H=int(os.environ.get("MV_TEST_H", "168" if world != 3 else "200")))
    if os.environ.get("MV_TEST_L"):  # another light grid: how many block layers the ranks share out (idle ranks whose slab would start beyond the map)
        cfg["L"] = int(os.environ["MV_TEST_L"])
    # (tests/test_dist_fuzz.py: any of the config's sizes -- N = volume and cube-face edge, V volumes, ray / light sample caps)
    for key, env in (("N", "MV_TEST_N"), ("V", "MV_TEST_V"), ("maxRay", "MV_TEST_MAXRAY"), ("maxLight", "MV_TEST_MAXLIGHT")):
        if os.environ.get(env):
            cfg[key] = int(os.environ[env])
    return cfg


def test_volumes(cfg):
    """The config's synthetic volumes; MV_TEST_SPECIAL_VOLUMES=seed sprinkles halfs no authoring tool writes over them (+-inf, NaNs, denormals, -0, +-65504, the
    halfs around ZERO_THRESHOLD: tests/test_random_scenes.py random_volume "special"): every rank and the single-GPU run make the same ones."""
    from multivolumes_amd import raycaster as rc
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    seed = os.environ.get("MV_TEST_SPECIAL_VOLUMES")
    if seed:
        r = np.random.default_rng(int(seed))
        special = np.array([0x7c00, 0xfc00, 0x7e00, 0xfe00, 0x0001, 0x8001, 0x03ff, 0x8000, 0x7bff, 0xfbff, 0x3c00, 0xbc00, 0x211f, 0x2120, 0x211e], np.uint16)
        N = cfg["N"]
        for v in vols:
            n = int(r.integers(1, N * N))
            idx = r.integers(0, N, size=(n, 3))
            v[idx[:, 0], idx[:, 1], idx[:, 2], r.integers(0, 4, size=n)] = special[r.integers(0, len(special), size=n)]
    return vols


def setter_script(frames, every, cfg):
    """(tests/test_dist_fuzz.py, MV_TEST_SETTER_SEED) per frame: the input setters in front of it -- collective inside the loop: every rank makes the same
    calls -- and whether the frame composites (checkpoint frames always do: a readback shows the image of a frame that composited)."""
    seed = os.environ.get("MV_TEST_SETTER_SEED")
    if not seed:
        return [([], True)] * frames
    r = np.random.default_rng(int(seed))
    out = []
    for f in range(frames):
        setters = []
        if r.random() < 0.35:
            what = ["sh", "max_samples", "light", "volume", "light_map", "shadow"][int(r.integers(6))]
            arg = {"sh": bool(r.integers(2)), "max_samples": (int(r.choice([16, 32, 48, 64])), int(r.choice([8, 16, 24, 32]))),
                   "light": (float(r.uniform(40, 90)), float(r.uniform(40, 90)), -float(r.uniform(40, 90))), "volume": (int(r.integers(cfg["V"])), int(r.integers(0, 40))),
                   "light_map": (int(r.integers(cfg["V"])), int(r.integers(1 << 30))), "shadow": float(r.uniform(4.0, 12.0))}[what]
            setters.append((what, arg))
        checkpoint = f % every == every - 1 or f == frames - 1
        out.append((setters, True if checkpoint else bool(r.random() < 0.75)))
    return out


def apply_setter(c, cfg, fr, what, arg):
    from multivolumes_amd import raycaster as rc, scene
    if what == "sh":
        c.SetSH(scene.DEFAULT_SH if arg else None)
    elif what == "max_samples":
        c.SetMaxSamples(*arg)
    elif what == "light":
        c.SetLight(arg, scene.LIGHT_COLOR, scene.LIGHT_INTENSITY)
    elif what == "volume":
        c.LoadVolumeData(arg[0], rc.synth_volume(cfg["N"], arg[1]))
    elif what == "light_map":
        L = cfg["L"]
        c.UploadLightMap(arg[0], np.random.default_rng(arg[1]).integers(0, 2 ** 32, size=(L, L, L), dtype=np.uint64).astype(np.uint32) & np.uint32(0x3BFFF7FF))
    elif what == "shadow":
        c.SetRenderTargets(shadowMap=rc.synth_sphere_shadow(64, fr["shadowVP"], (0, 0, 0), arg))


def main():
    import torch
    import torch.distributed as dist
    from multivolumes_amd import MultiRayCaster, _lib, scene
    from multivolumes_amd import raycaster as rc
    from multivolumes_amd import distributed as D
    out_path, frames, every = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    mode = sys.argv[4] if len(sys.argv) > 4 else "chains"           # "chains" (three view streams) or "single" (one view stream)
    static_depth = len(sys.argv) > 5 and sys.argv[5] == "1"          # no setter (= no drain of the frames in flight) inside the sequence
    diverge = len(sys.argv) > 5 and sys.argv[5] == "3"               # rank 1 is handed another camera in frame 5: the loop must REPORT it, not hang
    stalled = len(sys.argv) > 5 and sys.argv[5] == "4"               # rank 1's host stops handing over frames behind frame 8: rank 0 must get MV_ERR_TIMEOUT soon after its timeout, not hang
    after_setter = len(sys.argv) > 5 and sys.argv[5] == "2"          # a new host depth map every frame and NO flush: frame f - 1 is read back right after the setter of frame f
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = os.environ.get("MV_TEST_RCCL") == "1"  # one GPU per rank and the library's own RCCL communicators (boxes with >= world GPUs)
    # MV_TEST_SEPARATE_GPUS=1: one GPU per rank for the peer-store transport too (boxes with >= world GPUs: the stores then cross the fabric,
    # and the folded waits meet caches that are not the writer's)
    separate = rccl or (os.environ.get("MV_TEST_SEPARATE_GPUS") == "1" and torch.cuda.device_count() >= world)
    device = int(os.environ.get("LOCAL_RANK", "0")) if separate else 0
    torch.cuda.set_device(device)
    if rccl:
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo")
    cfg = test_config(world)
    V = cfg["V"]
    vols = test_volumes(cfg)
    flags = _lib.MV_FLAG_PIPELINE_FRAMES | (_lib.MV_FLAG_SINGLE_VIEW_STREAM if mode == "single" else 0)
    four = os.environ.get("MV_TEST_FOUR_CHAINS") == "1"  # four whole frames in flight, upload + cull on the frames' own view streams
    if four:
        flags |= _lib.MV_FLAG_CULL_ON_VIEW_STREAM
    per_launch = int(os.environ.get("MV_TEST_FRAMES_PER_LAUNCH", "0"))  # B > 1: groups of B frames go through every long kernel together (needs the four-chain flags, folded peer stores)
    with MultiRayCaster(cfg["N"], cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, shard=(rank, world), device=device, framesInFlight=4 if four else 0,
                        framesPerLaunch=per_launch) as c:
        fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
        peer = os.environ.get("MV_TEST_TRANSPORT") == "peer"     # the peer-store transport: IPC mappings, works with every rank on one GPU
        fold = {"1": True, "0": False}.get(os.environ.get("MV_TEST_PEER_FOLD"))  # (None: the library decides -- ranks sharing a GPU keep the flag kernels)
        kind = D.init_frame_loop(c, prefer_peer=peer, peer_timeout_ms=int(os.environ.get("MV_TEST_PEER_TIMEOUT_MS", "120000")), peer_fold=fold, peer_rows_to_root=os.environ.get("MV_TEST_ROWS_TO_ROOT") == "1",
                                 peer_deal_volumes={"1": True, "0": False}.get(os.environ.get("MV_TEST_DEAL_VOLUMES")),  # (None: the library's choice)
                                 peer_push_cells=os.environ.get("MV_TEST_PUSH_CELLS") == "1")  # (round 4's light push: whole cells; the default is plain texels rebuilt by the receiver)
        assert kind == ("peer" if peer else "rccl" if rccl else "torch-callbacks"), kind
        if peer:
            info = c.DistInfo()
            assert info["transport"] == "peer" and info["ranks"] == world, info
            if os.environ.get("MV_TEST_COHERENCE") == "1":
                # the in-kernel coherence test by itself, with and without the acquire behind the wait (collective); rank 0 reports
                with_acq, without = c.DistPeerCoherence(True), c.DistPeerCoherence(False)
                res = torch.tensor([with_acq[0], with_acq[1], without[0], without[1]], dtype=torch.int64)
                dist.all_reduce(res)
                if rank == 0:
                    print("COHERENCE acquire stale %d first_bad %d | no acquire stale %d first_bad %d" % tuple(res.tolist()), flush=True)
        if rccl:
            info = c.DistInfo()
            assert info["transport"] == "rccl" and info["ranks"] == world and info["communicators"] == (2 if mode == "single" else 3), info
        imgs, infos = [], []
        script = setter_script(frames, every, cfg)
        # (MV_TEST_SKEW_SEED: the ranks' hosts stall at different frames for different times -- a rank runs groups ahead of another as far as the flags let it)
        skew = np.random.default_rng(int(os.environ["MV_TEST_SKEW_SEED"]) * 64 + rank) if os.environ.get("MV_TEST_SKEW_SEED") else None
        for f, (e, vp) in enumerate(camera_sequence(cfg, frames)):
            if skew is not None and skew.random() < 0.25:
                import time
                time.sleep(float(skew.uniform(0.0, 0.03)))
            for what, arg in script[f][0]:
                apply_setter(c, cfg, fr, what, arg)
            if (not static_depth and not diverge) or f == 0:
                # (inside the loop the setter first enqueues the pending tail of frame f - 1 -- its exchange, composite and row gather --
                # so that frame is composited against ITS depth map, not this one; a collective call, made by every rank)
                c.SetRenderTargets(rc.synth_sphere_depth(cfg["W"], cfg["H"], vp, (0, 0, 0), scene.SPHERE_RADIUS))
                if after_setter and f > 0:
                    imgs.append(c.ReadBackComposite())
                    infos.append(c.ReadBackVolumeInfo())
            if stalled:
                import time
                if rank == 1 and f == 8:
                    c.synchronize()
                    print("rank 1: stops here", flush=True)
                    time.sleep(float(os.environ.get("MV_TEST_STALL_S", "40")))  # (alive, its buffers mapped, handing over nothing)
                    os._exit(0)
                if f == 8:
                    t_stall = time.time()
                try:
                    c.DistFrame(vp, fr["shadowVP"], e)
                    if f >= 8:
                        c.synchronize()
                except Exception as ex:
                    assert f >= 8 and "gave up waiting for rank 1" in str(ex), (f, str(ex))
                    print(f"rank {rank}: stall reported at frame {f} after {time.time() - t_stall:.1f} s: {ex}", flush=True)
                    os._exit(0)
                continue
            if diverge:
                if f == 5 and rank == 1:
                    e = np.array([3.0, 9.0, -12.0], np.float32)  # inside the grid of volumes: another visible set, other shard sizes
                    vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0))
                try:
                    c.DistFrame(vp, fr["shadowVP"], e)
                    c.synchronize()  # (this test only: let the frame's light exchange, which carries the check, finish before the next call)
                except Exception as ex:
                    # (collectives: the call after the divergent frame, before it sizes that frame's all-gather; peer stores: already the
                    # divergent frame's own synchronize -- the light stage's wait has compared the ranks' cull words by then)
                    # (several frames per launch: the group that holds frame 5 is compared as a whole, when it goes out)
                    late = tuple(range(5, 6 + 2 * per_launch)) if per_launch > 1 else ()
                    assert f in (late or ((5, 6) if peer else (6,))) and "disagree" in str(ex), (f, str(ex))
                    print(f"rank {rank}: divergence reported at frame {f}: {ex}", flush=True)
                    os._exit(0)  # nothing more can be exchanged; the ranks leave without a collective
                assert f <= 5 + 2 * max(per_launch, 1) - 2, "a frame after the divergent one went through"
                continue
            c.DistFrame(vp, fr["shadowVP"], e, composite=script[f][1])
            if after_setter and f != frames - 1:
                continue
            if f % every == every - 1 or f == frames - 1:
                c.DistFlush()
                imgs.append(c.ReadBackComposite())
                infos.append(c.ReadBackVolumeInfo())
        st = c.GetStats()
        lms = np.stack([c.ReadBackLightMap(v) for v in range(V)])
        nbytes = c.DistExchangedBytes()
    cnt = torch.tensor([st[k] for k in ("view_samples", "lit_samples", "rays", "light_samples", "composite_samples")], dtype=torch.int64)
    dist.all_reduce(cnt)
    if not diverge:
        # EVERY rank holds every light map, current: rank 0's are compared with the single-GPU run by the test, the other ranks' with rank 0's
        # here (a rank that rebuilds the others' layers wrongly corrupts its own maps only)
        import zlib
        crcs = torch.zeros(world, dtype=torch.int64)
        crcs[rank] = zlib.crc32(lms.tobytes())
        dist.all_reduce(crcs)
        assert len(set(crcs.tolist())) == 1, f"the ranks' light maps differ: {crcs.tolist()}"
    if rank == 0:
        np.savez(out_path, imgs=np.stack(imgs), infos=np.stack(infos), counts=cnt.numpy(), lms=lms, nbytes=nbytes)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
