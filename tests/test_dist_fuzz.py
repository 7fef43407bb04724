"""Fuzz of the multi-GPU frame loop (mv_dist_frame), round 6: a seeded random config (volume edge N, light grid L, volume count V, viewport, sample
caps), world size and transport form against the single-GPU frames of the same config -- composites, VolumeInfo, every light map on EVERY rank, summed
counters, bit for bit.  Round 5's review found a range that wrapped for light grids and world sizes no hand-picked case had combined (L = 48 on 8 ranks,
T2 on 6 or 7): the hand-picked cases stay in tests/test_dist_frame_loop.py, this walks the space between them.  Ranks share cuda:0 (IPC mappings and
gloo callbacks work between processes on one device).  MV_DIST_FUZZ=n runs n seeds (default 6: about a minute); MV_DIST_FUZZ_FIRST=k starts at seed k."""
import os
import sys

import numpy as np
import pytest

from tests.test_dist_frame_loop import ROOT, _free_port, run_group, single_gpu

pytestmark = pytest.mark.gpu

MODES = ["chains", "single", "peer", "peer-folded", "peer-four", "peer-root", "peer-tiles", "peer-x2", "peer-x3", "peer-x4", "peer-x2-volumes", "peer-x4-root", "peer-cells", "peer-x4-cells"]


def draw(seed):
    r = np.random.default_rng(1000 + seed)
    world = int(r.choice([2, 3, 4, 5, 6, 7, 8]))
    mode = MODES[int(r.integers(len(MODES)))]
    cfg = dict(N=int(r.choice([16, 24, 32, 36, 40, 48, 56])), L=int(r.integers(6, 45)), V=int(r.choice([1, 2, 4, 4, 9, 16, 16])),  # (N = 12 with one-texel mips: the corners)
               W=int(r.integers(64, 330)), H=int(r.integers(40, 250)), maxRay=int(r.choice([16, 32, 48, 64, 96])), maxLight=int(r.choice([8, 16, 24, 32])))
    if world >= 6:  # (many processes on the one GPU: keep each small)
        cfg["N"] = min(cfg["N"], 40)
    frames = int(r.integers(8, 15))
    every = int(r.choice([3, 5, 7, frames]))
    static_depth = bool(r.integers(2))
    cfg["setters"] = int(r.integers(1, 1 << 30)) if r.random() < 0.5 else 0  # (half of the seeds: setters between the frames, frames without a composite)
    cfg["special"] = int(r.integers(1, 1 << 30)) if r.random() < 0.25 else 0  # (a quarter: +-inf, NaN, denormal ... halfs in the volumes)
    return world, mode, cfg, frames, every, static_depth


# hand-picked corners the draw does not reach: N > 256 (no per-axis address tables in LDS: the LUT = false instantiations of the batched light and
# view marches), ragged N (faces of 33 and 17 texels: partial tiles), a viewport of ONE tile row on eight ranks, one volume on eight ranks
CORNERS = {
    "big-n-x2": (2, "peer-x2", dict(N=264, L=12, V=2, W=96, H=64, maxRay=24, maxLight=8), 6, 3, True),
    "big-n-chains": (2, "chains", dict(N=260, L=10, V=1, W=80, H=48, maxRay=16, maxLight=8), 5, 5, True),
    "ragged-n-x4": (3, "peer-x4", dict(N=33, L=9, V=4, W=130, H=70, maxRay=32, maxLight=16), 9, 9, True),
    "one-tile-row": (8, "peer-x3-root", dict(N=24, L=10, V=4, W=200, H=8, maxRay=32, maxLight=8), 7, 7, True),
    "one-volume": (8, "peer-folded", dict(N=32, L=33, V=1, W=96, H=96, maxRay=48, maxLight=16), 8, 4, False),
    "one-texel-mips": (4, "peer-x2", dict(N=12, L=7, V=4, W=24, H=16, maxRay=16, maxLight=8, special=77), 9, 3, True),
    "special-halfs-x4": (3, "peer-x4-root", dict(N=24, L=11, V=9, W=150, H=90, maxRay=32, maxLight=16, special=78), 9, 9, True),
}


def draw_or_corner(seed):
    return CORNERS[seed] if isinstance(seed, str) else draw(seed)


N_SEEDS = int(os.environ.get("MV_DIST_FUZZ", "6"))
FIRST = int(os.environ.get("MV_DIST_FUZZ_FIRST", "0"))


@pytest.mark.parametrize("seed", list(range(FIRST, FIRST + N_SEEDS)) + list(CORNERS))
def test_random_config_world_and_transport_match_single_gpu(tmp_path, seed, monkeypatch):
    world, mode, cfg, frames, every, static_depth = draw_or_corner(seed)
    label = f"seed {seed}: world {world} {mode} {cfg} frames {frames} every {every} static_depth {static_depth}"
    print(label)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key, name in (("N", "MV_TEST_N"), ("L", "MV_TEST_L"), ("V", "MV_TEST_V"), ("W", "MV_TEST_W"), ("H", "MV_TEST_H"), ("maxRay", "MV_TEST_MAXRAY"), ("maxLight", "MV_TEST_MAXLIGHT")):
        env[name] = str(cfg[key])
        monkeypatch.setenv(name, str(cfg[key]))  # (single_gpu() below reads the same config)
    if mode.endswith("-cells"):  # round 4's light push: whole cells instead of plain texels rebuilt by the receivers
        mode, env["MV_TEST_PUSH_CELLS"] = mode[:-len("-cells")], "1"
    if mode.endswith("-root"):  # ("peer-root", "peer-x3-root")
        mode, env["MV_TEST_ROWS_TO_ROOT"] = mode[:-len("-root")], "1"
    if mode == "peer-tiles":
        mode, env["MV_TEST_DEAL_VOLUMES"] = "peer", "0"
    if mode.endswith("-volumes"):
        mode, env["MV_TEST_DEAL_VOLUMES"] = mode[:-len("-volumes")], "1"
    if "-x" in mode:
        env["MV_TEST_FRAMES_PER_LAUNCH"] = mode[-1]
        mode, env["MV_TEST_FOUR_CHAINS"] = "peer-folded", "1"
    if mode.endswith("-four"):
        env["MV_TEST_FOUR_CHAINS"] = "1"
        mode = mode[:-len("-four")]
    if mode.startswith("peer"):
        env["MV_TEST_PEER_FOLD"] = "1" if mode == "peer-folded" else "0"
        mode, env["MV_TEST_TRANSPORT"] = "chains", "peer"
    if seed if isinstance(seed, str) else seed % 2:  # (every other seed, and the corners: the ranks' hosts stall at different frames for different times)
        env["MV_TEST_SKEW_SEED"] = str(sum(map(ord, seed)) if isinstance(seed, str) else seed)
    if cfg.get("special"):
        env["MV_TEST_SPECIAL_VOLUMES"] = str(cfg["special"])
        monkeypatch.setenv("MV_TEST_SPECIAL_VOLUMES", str(cfg["special"]))
    if cfg.get("setters"):  # (tests/dist_worker.py setter_script: light probe, sample caps, light, volume / light-map uploads, shadow map; DistFrame(composite=False))
        env["MV_TEST_SETTER_SEED"] = str(cfg["setters"])
        monkeypatch.setenv("MV_TEST_SETTER_SEED", str(cfg["setters"]))
    out = str(tmp_path / "dist.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(frames), str(every), mode, "1" if static_depth else "0"]
    r = run_group(cmd, 400, env)
    assert r.returncode == 0, label + "\n" + r.stderr[-3000:]
    d = np.load(out)
    imgs, infos, counts, lms = single_gpu(world, frames, every, static_depth)
    assert np.array_equal(d["infos"], infos), label
    for i in range(len(imgs)):
        assert np.array_equal(d["imgs"][i], imgs[i]), f"{label}: composite of checkpoint {i} differs ({(d['imgs'][i] != imgs[i]).sum()} halfs)"
    assert np.array_equal(d["lms"], lms), label + ": light maps"
    assert np.array_equal(d["counts"], counts), label + ": counters"
