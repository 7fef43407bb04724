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
