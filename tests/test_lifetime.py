"""Lifetime of a context (round 6): create -> a few frames -> (viewport changes, the multi-GPU loop attached and detached) -> destroy, many times over, in every
form of a context: device memory returns to where it was (nothing the context allocated survives mv_destroy), no stream / event / pinned-buffer creation starts to
fail, and a context destroyed with frames still in flight takes nothing down with it."""
import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, _lib, scene
from multivolumes_amd import raycaster as rc
from tests.helpers import replicating_allgather

pytestmark = pytest.mark.gpu


def test_contexts_come_and_go_without_leaking_device_memory():
    import torch
    cfg = dict(scene.CONFIGS["T2"], N=32, L=16, V=4, W=200, H=120)
    V, N, L = cfg["V"], cfg["N"], cfg["L"]
    vols = [rc.synth_volume(N, k) for k in range(V)]
    P, CV, RQ, SV = _lib.MV_FLAG_PIPELINE_FRAMES, _lib.MV_FLAG_CULL_ON_VIEW_STREAM, _lib.MV_FLAG_RAY_QUEUE, _lib.MV_FLAG_SINGLE_VIEW_STREAM
    forms = [(0, 0, 0), (P, 0, 0), (P | CV, 4, 0), (P | RQ, 0, 0), (P, 5, 0), (P | SV, 0, 0), (P | CV, 4, 3)]

    def cycle(i):
        flags, in_flight, per_launch = forms[i % len(forms)]
        c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight, framesPerLaunch=per_launch)
        fr = scene.setup_raycaster(c, cfg, vols, use_sh=bool(i & 1))
        loop = (flags & P) and i % 3 == 0 and not (flags & RQ)
        if loop:  # the in-library frame loop on caller-supplied collectives (one rank alone), detached again before the end on every other pass
            c.close()
            c = MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight, shard=(0, 2))
            fr = scene.setup_raycaster(c, cfg, vols)
            c.DistInitCallbacks(replicating_allgather(2), lambda b, n, st, w: 0)
            c.DistSetCullCrosscheck(False)
        for f in range(7):
            if loop:
                c.DistFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
            else:
                c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
                c.Render(composite=True)
            if f == 3 and not loop and i % 2 == 0:
                c.SetViewport(cfg["W"] + 8 * (i % 5), cfg["H"] + 8 * (i % 3))
                c.SetRenderTargets(rc.synth_sphere_depth(c.viewport[0], c.viewport[1], fr["viewProj"], (0, 0, 0), 8.0))
        if loop and i % 2:
            c.DistFlush()
            c.DistShutdown()
        if i % 4 != 3:
            c.synchronize()   # (every fourth context is destroyed with its frames still in flight)
        c.close()

    for i in range(2 * len(forms)):   # warm-up: the runtime's pools (command buffers, signals, the allocator's own arenas) reach their steady state
        cycle(i)
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    n = 12 * len(forms)
    for i in range(n):
        cycle(i)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    # a context of this size holds ~100 MB; one leaked buffer per cycle would be n x (KBs .. MBs).  The runtime may keep a few MB of pools.
    assert free0 - free1 < 24 * 1024 * 1024, f"{(free0 - free1) / 2 ** 20:.1f} MiB of device memory did not come back after {n} contexts"
    # and the box still works
    with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=P | CV, framesInFlight=4) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render(composite=True)
        assert c.ReadBackComposite().any()


def test_independent_contexts_driven_from_different_threads():
    """include/multivolumes.h: "functions are not re-entrant per context, independent contexts may be driven from different threads".  Four threads, a context
    each (different sizes and forms), frames with readbacks, setters and a viewport change, all at once -- every thread's images equal the ones the same
    sequence produced alone."""
    import threading
    P, CV, RQ = _lib.MV_FLAG_PIPELINE_FRAMES, _lib.MV_FLAG_CULL_ON_VIEW_STREAM, _lib.MV_FLAG_RAY_QUEUE
    jobs = [(dict(N=32, L=16, V=4, W=200, H=120), P | CV, 4), (dict(N=24, L=12, V=9, W=160, H=100), P, 0), (dict(N=40, L=20, V=4, W=128, H=128), 0, 0),
            (dict(N=32, L=10, V=16, W=240, H=90), P | RQ, 0)]

    def sequence(job, out, rounds):
        over, flags, in_flight = job
        cfg = dict(scene.CONFIGS["T2"], **over)
        V, N, L = cfg["V"], cfg["N"], cfg["L"]
        vols = [rc.synth_volume(N, k) for k in range(V)]
        try:
            for _ in range(rounds):
                imgs = []
                with MultiRayCaster(N, L, V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight) as c:
                    fr = scene.setup_raycaster(c, cfg, vols, use_sh=True)
                    proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, cfg["W"] / cfg["H"], 1.0, 1000.0)
                    for f in range(40):
                        a = 0.07 * f
                        e = np.array([85.0 * np.sin(a), 18.0, -85.0 * np.cos(a)], np.float32)
                        vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
                        if f % 9 == 4:
                            c.SetRenderTargets(rc.synth_sphere_depth(c.viewport[0], c.viewport[1], vp, (0, 0, 0), 6.0 + f % 5))
                        if f == 20:
                            c.SetViewport(cfg["W"] + 16, cfg["H"] + 8)
                            proj = rc.perspective_fov_lh(scene.FOV_ANGLE_Y, c.viewport[0] / c.viewport[1], 1.0, 1000.0)
                            vp = rc.multiply(rc.look_at_lh(e, (0, 0, 0)), proj)
                            c.SetRenderTargets(rc.synth_sphere_depth(c.viewport[0], c.viewport[1], vp, (0, 0, 0), 8.0))
                        c.UpdateFrame(vp, fr["shadowVP"], e)
                        c.Render(composite=True)
                        if f % 7 == 6:
                            imgs.append(c.ReadBackComposite())
                    imgs.append(np.stack([c.ReadBackLightMap(v) for v in range(V)]))
                out.append(imgs)
        except Exception as ex:  # (a thread's failure must reach the test)
            out.append(ex)

    alone = []
    for job in jobs:
        r = []
        sequence(job, r, 1)
        assert not isinstance(r[0], Exception), r[0]
        alone.append(r[0])
    results = [[] for _ in jobs]
    threads = [threading.Thread(target=sequence, args=(job, results[i], 3)) for i, job in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "a thread hung"
    for i, rounds in enumerate(results):
        assert len(rounds) == 3, rounds
        for got in rounds:
            assert not isinstance(got, Exception), got
            assert len(got) == len(alone[i])
            for a, b in zip(got, alone[i]):
                assert np.array_equal(a, b), f"context {i} read other bytes while the other threads ran"
