"""Round 6: the frame-latency sampler counts a frame ONCE, at its last mark (a frame is marked behind its march and behind its
composite: round 5 could count it at the first and let the second write into another frame's sample), and mv_create's hardware-queue warning
travels through mv_get_warning, never through mv_last_error."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from multivolumes_amd import MultiRayCaster, _lib, scene
from multivolumes_amd import raycaster as rc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags,in_flight", [(0, 0), (_lib.MV_FLAG_PIPELINE_FRAMES, 0), (_lib.MV_FLAG_PIPELINE_FRAMES | _lib.MV_FLAG_CULL_ON_VIEW_STREAM, 4)])
def test_latency_samples_count_every_sampled_frame_once_and_cover_the_composite(flags, in_flight):
    cfg = dict(scene.CONFIGS["T2"], W=256, H=168)
    V = cfg["V"]
    vols = [rc.synth_volume(cfg["N"], k) for k in range(V)]
    with MultiRayCaster(cfg["N"], cfg["L"], V, V, (cfg["W"], cfg["H"]), shadowMapSize=64, flags=flags, framesInFlight=in_flight) as c:
        fr = scene.setup_raycaster(c, cfg, vols)
        for every, frames in ((1, 40), (3, 41), (8, 64)):
            c.SetLatencySampling(every)
            for f in range(frames):
                c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
                c.Render(composite=True)
            lat = c.GetLatency(reset=True)
            # every `every`-th frame is sampled (the frame counter runs on from the context's creation: one more or less at the ends), each exactly once
            assert abs(lat["samples"] - frames / every) <= 1.0, (every, frames, lat)
            assert 0.0 < lat["mean_ms"] <= lat["max_ms"] < 10_000.0, lat
            assert c.GetLatency()["samples"] == 0, "reset"
        # a sample ends behind the frame's LAST kernel: a frame with its composite cannot be reported faster than the GPU takes for it
        c.SetLatencySampling(1)
        c.synchronize()
        t0 = time.perf_counter()
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render(composite=True)
        c.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        one = c.GetLatency(reset=True)
        assert one["samples"] == 1 and 0.0 < one["mean_ms"] <= wall_ms * 1.05 + 0.05, (one, wall_ms)
        c.SetLatencySampling(0)
        c.UpdateFrame(fr["viewProj"], fr["shadowVP"], fr["eye"])
        c.Render(composite=True)
        assert c.GetLatency()["samples"] == 0
        assert c._lib.mv_last_error(c._ctx) == b""


def test_queue_warning_is_not_an_error():
    """More chains than hardware queues (GPU_MAX_HW_QUEUES unset = the runtime's four): mv_create succeeds, mv_last_error stays "", mv_get_warning
    names the variable.  In a process of its own, so that the variable is really unset when the HIP runtime starts."""
    code = (
        "import warnings, sys\n"
        "from multivolumes_amd import MultiRayCaster, _lib\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    c = MultiRayCaster(32, 24, 4, 4, (128, 96), shadowMapSize=64, flags=_lib.MV_FLAG_PIPELINE_FRAMES, framesInFlight=5)\n"
        "    quiet = MultiRayCaster(32, 24, 4, 4, (128, 96), shadowMapSize=64, flags=_lib.MV_FLAG_PIPELINE_FRAMES)\n"
        "assert c._lib.mv_last_error(c._ctx) == b'', c._lib.mv_last_error(c._ctx)\n"
        "assert 'GPU_MAX_HW_QUEUES' in c.warning and c._lib.mv_get_warning(c._ctx).decode() == c.warning, c.warning\n"
        "assert any('GPU_MAX_HW_QUEUES' in str(x.message) for x in w), [str(x.message) for x in w]\n"
        "assert quiet.warning == '' and quiet._lib.mv_get_warning(quiet._ctx) == b''\n"
        "c.close(); quiet.close(); print('WARNING_OK')\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "WARNING_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
