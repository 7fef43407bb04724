"""world_size-2 gloo test of the multi-GPU protocol on CPU: volumes sharded v % 2, replicated cull,
each rank marches only its volumes (the ORACLE stands in for the kernels here), one all-gather of the packed
selected-mip cube faces, each rank composites its band of rows, bands gathered -> must equal the single-process frame."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multivolumes_amd import distributed as D


def test_partition_and_layout_pure():
    assert D.owned_volumes(16, 3, 8) == [3, 11]
    assert D.owned_volumes(5, 1, 2) == [1, 3]
    assert sorted(sum((D.owned_volumes(13, r, 4) for r in range(4)), [])) == list(range(13))
    assert D.slots_per_shard(13, 4) == 4
    info = np.zeros((4, 4), np.uint16)
    info[:, 0] = [2, 1, 3, 1]
    d, sb, tb = D.exchange_layout(256, info, [0, 1, 2, 3], 4, 2)
    assert (d, sb, tb) == (128, 6 * 128 * 128 * 12, 2 * 6 * 128 * 128 * 12)
    d, sb, tb = D.exchange_layout(256, info, [], 4, 2)
    assert d == 256
    # C4 figure of SURVEY 8(e): 16 volumes at mip 1 over 8 GPUs -> 2.36 MB per rank, 18.9 MB in total
    info16 = np.ones((16, 4), np.uint16)
    d, sb, tb = D.exchange_layout(256, info16, list(range(16)), 16, 8)
    assert tb == 2 * 6 * 128 * 128 * 12 == 2359296
    bands = [D.row_range(1024, r, 8) for r in range(8)]
    assert bands[0][0] == 0 and bands[-1][1] == 1024 and all(bands[i][1] == bands[i + 1][0] for i in range(7))
    assert all(a % 8 == 0 for a, _ in bands)
    bands = [D.row_range(100, r, 3) for r in range(3)]
    assert bands[0][0] == 0 and bands[-1][1] == 100 and all(bands[i][1] == bands[i + 1][0] for i in range(2))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multivolumes_amd import scene
        from multivolumes_amd import raycaster as rc
        from oracle import mv_oracle as o
        from tests.helpers import default_descs, host_frame
        o.load().mvo_set_num_threads(2)
        cfg = dict(scene.CONFIGS["T1"], W=96, H=80)
        V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
        M = 32
        vols = [rc.synth_volume(N, k) for k in range(V)]  # every rank holds every volume
        descs = default_descs(V, V, N)
        eye, vp, svp, cb, po = host_frame(cfg)
        depth, shadow = scene.depth_and_shadow(cfg, vp, svp, M)
        info, vis, cube = o.cull(V, po, descs, cb, cfg["maxRay"])  # replicated cull
        mine = [v for v in D.owned_volumes(V, rank, world)]
        # frame f relights visible[f % count] (CSRayMarchL.hlsl:29-33) on the rank that owns that volume; the others skip the light march
        lms = None
        for f in range(len(vis)):
            if (int(vis[f % len(vis)]) if len(vis) else f % V) not in mine:
                continue
            _, _, _, cbf, _ = host_frame(cfg, f)
            _, lms, _ = o.march_light(V, N, L, M, vols, descs, po, cbf, shadow, None, vis, cfg["maxLight"], lms)
        if lms is None:
            lms = np.zeros((V, L, L, L), np.uint32)
        own_cube = [int(v) for v in cube if int(v) in mine]
        cm, cd, st = o.march_view(V, N, L, W, H, vols, lms, depth, po, cb, info, own_cube)
        # exchange
        d, slot_bytes, shard_bytes = D.exchange_layout(N, info, vis, V, world)
        send = D.pack_shard_numpy(N, V, rank, world, info, vis, cm, cd, d, o.cube_chain_texels(N), o.cube_mip_offset)
        assert send.size == shard_bytes
        recv = torch.empty(shard_bytes * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, torch.from_numpy(send))
        D.unpack_gathered_numpy(N, V, rank, world, info, vis, recv.numpy(), cm, cd, d, o.cube_mip_offset)
        # light maps of the volumes on the screen-space scheme are needed by RayCast on every rank: gather them too
        lm_t = torch.from_numpy(lms.view(np.uint8).copy())
        gathered_lm = [torch.empty_like(lm_t) for _ in range(world)]
        dist.all_gather(gathered_lm, lm_t)
        for r in range(world):
            for v in D.owned_volumes(V, r, world):
                lms[v] = gathered_lm[r].numpy().view(np.uint32).reshape(lms.shape)[v]
        # composite own band, gather bands
        r0, r1 = D.row_range(H, rank, world)
        img, _ = o.composite(V, N, L, W, H, vols, lms, depth, po, cb, info, vis, cm, cd, r0, r1)
        band = torch.from_numpy(img.view(np.uint8).copy())
        bands = [torch.empty_like(band) for _ in range(world)]
        dist.all_gather(bands, band)
        full = np.zeros_like(img)
        for r in range(world):
            a, b = D.row_range(H, r, world)
            full[a:b] = bands[r].numpy().view(np.uint16).reshape(img.shape)[a:b]
        samples = torch.tensor([st["view_samples"], st["rays"]], dtype=torch.int64)
        dist.all_reduce(samples)
        if rank == 0:
            np.savez(out_path, full=full, cm=cm, cd=cd, lms=lms, samples=samples.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_pipeline_matches_single_process(oracle, tmp_path):
    from multivolumes_amd import scene
    from multivolumes_amd import raycaster as rc
    from tests.helpers import oracle_pipeline
    world = 2
    out_path = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    # single-process reference: the same frames, every one relit by the one process
    cfg = dict(scene.CONFIGS["T1"], W=96, H=80)
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, 32)
    assert np.array_equal(got["lms"], ref["lightMaps"]), "per-shard relighting must produce the same light maps"
    assert np.array_equal(got["cm"], ref["cubeMaps"]) and np.array_equal(got["cd"], ref["cubeDepths"])
    img, _ = oracle.composite(cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"],
                              ref["info"], ref["vis"], ref["cubeMaps"], ref["cubeDepths"])
    assert np.array_equal(got["full"], img)
    assert int(got["samples"][0]) == ref["stats"]["view_samples"] and int(got["samples"][1]) == ref["stats"]["rays"]
