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


def test_light_layers_and_cell_rebuild_pure():
    """The peer-store transport's light push, restated in numpy: the ranks' z slabs tile the map exactly once, and the cells every receiver
    rebuilds from the slabs' PLAIN texels (a cell holds texels of one z only) are the cells of the whole map."""
    rng = np.random.default_rng(5)
    for L, world in ((96, 8), (48, 3), (16, 5), (10, 8), (7, 2)):
        covered = np.zeros(L, np.int32)
        for r in range(world):
            z0, z1 = D.light_layers_of_rank(L, r, world)
            assert 0 <= z0 <= z1 <= L and z0 % 4 == 0
            covered[z0:z1] += 1
        assert (covered == 1).all(), (L, world)
        plain = rng.integers(0, 2 ** 32, size=(L, L, L), dtype=np.uint64).astype(np.uint32)
        whole = D.light_cells_from_texels(plain)
        assert whole.shape == (L, L, L, 4) and np.array_equal(whole[..., 0], plain)
        assert np.array_equal(whole[3, 2, L - 1], [plain[3, 2, L - 1]] * 2 + [plain[3, 3, L - 1]] * 2), "x + 1 clamps"
        rebuilt = np.zeros_like(whole)
        for r in range(world):  # what rank r pushes is all a receiver needs for rank r's layers
            z0, z1 = D.light_layers_of_rank(L, r, world)
            if z1 > z0:
                rebuilt[z0:z1] = D.light_cells_from_texels(plain[z0:z1])
        assert np.array_equal(rebuilt, whole)


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
    assert np.array_equal(got["lms"], ref["lightMaps"]), "the light march shared out by voxel blocks assembles the single-process light maps"
    assert np.array_equal(got["cm"], ref["cubeMaps"]) and np.array_equal(got["cd"], ref["cubeDepths"])
    img, _ = oracle.composite(cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"],
                              ref["info"], ref["vis"], ref["cubeMaps"], ref["cubeDepths"])
    assert np.array_equal(got["full"], img)
    assert int(got["samples"][0]) == ref["stats"]["view_samples"] and int(got["samples"][1]) == ref["stats"]["rays"]


def test_the_loops_work_deal_pure():
    """The rule by which the in-library frame loop deals a frame's work (distributed.py mirrors of the cull's last workgroup): every
    visible cube-map volume to exactly one rank, loads within one volume's cost of each other, the exchange layout dense per owner; every
    tile row to exactly one rank, and a bijective rank-major storage order."""
    rng = np.random.default_rng(5)
    for world in (2, 3, 4, 8):
        for V, N in ((16, 256), (64, 256), (7, 48)):
            info = np.zeros((V, 4), np.uint16)
            info[:, 0] = rng.integers(0, 3, V)                         # mip
            info[:, 1] = rng.integers(8, 257, V)                       # SmpCount
            info[:, 2] = rng.integers(1, 64, V) | np.where(rng.random(V) < 0.7, 0x8000, 0)   # face mask | cube-map scheme bit
            visible = [v for v in range(V) if rng.random() < 0.8]
            owner = D.deal_cube_volumes(info, visible, V, N, world)
            cube = [v for v in visible if info[v, 2] & 0x8000]
            cost = {v: bin(int(info[v, 2]) & 0x3F).count("1") * (((max(1, N >> int(info[v, 0]))) + 7) // 8) ** 2 * int(info[v, 1]) + 1 for v in cube}
            load = [sum(cost[v] for v in cube if owner[v] == r) for r in range(world)]
            assert all(0 <= owner[v] < world for v in range(V))
            assert all(owner[v] == v % world for v in range(V) if v not in cube)
            if cube:
                assert max(load) - min(load) <= max(cost.values()), (world, V, load)     # the greedy bound
            off, biggest = D.compact_exchange_layout(info, visible, owner, V, N, world)
            for r in range(world):
                mine = [v for v in cube if owner[v] == r]
                run = 0
                for v in mine:                                            # ascending, back to back
                    assert off[v] == run
                    run += bin(int(info[v, 2]) & 0x3F).count("1") * max(1, N >> int(info[v, 0])) ** 2
                assert run <= biggest
            assert biggest == max(sum(bin(int(info[v, 2]) & 0x3F).count("1") * max(1, N >> int(info[v, 0])) ** 2 for v in cube if owner[v] == r) for r in range(world))
        for H in (80, 168, 200, 1024, 1):
            T = (H + 7) // 8
            rows = [D.tile_rows_of_rank(H, r, world) for r in range(world)]
            assert sorted(t for rr in rows for t in rr) == list(range(T))
            stored = [D.stored_tile_row(t, H, world) for t in range(T)]
            per_rank = (T + world - 1) // world
            assert len(set(stored)) == T and max(stored) < world * per_rank
            for r in range(world):                                        # a rank's rows are one contiguous block, in order
                assert [D.stored_tile_row(t, H, world) for t in rows[r]] == list(range(r * per_rank, r * per_rank + len(rows[r])))


def _worker_dealt(rank, world, port, out_path):
    """One frame of the loop's protocol on the oracle: cube-map volumes dealt by cost, compact exchange, tile rows dealt round-robin,
    rank-major image, all over gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multivolumes_amd import scene
        from multivolumes_amd import raycaster as rc
        from oracle import mv_oracle as o
        from tests.helpers import default_descs, host_frame, oracle_pipeline
        o.load().mvo_set_num_threads(2)
        cfg = dict(scene.CONFIGS["T1"], W=96, H=88)
        V, N, L, W, H = cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"]
        vols = [rc.synth_volume(N, k) for k in range(V)]
        eye = (24.0, 16.0, -52.0)
        ref = oracle_pipeline(o, cfg, vols, 32, frames=V - 1, eye=eye)   # every frame but the last: all ranks hold every light map
        info, vis, cube, lms = ref["info"], ref["vis"], ref["cube"], ref["lightMaps"]
        # The LAST frame's light march, shared out like the loop's: every rank marches its 1 / world of the 4x4x4 voxel blocks of the relit
        # map (every rank holds every volume), the shares are all-gathered as plain words, 64 per block, and every rank assembles the map.
        _, _, _, cb_last, po_last = host_frame(cfg, V - 1, eye)
        B = (L + 3) // 4
        total = B * B * B
        per = (total + world - 1) // world
        lo, hi = min(rank * per, total), min((rank + 1) * per, total)
        o.load().mvo_set_light_block_range(lo, hi)
        mine_lm = np.zeros_like(lms)
        relit, mine_lm, _ = o.march_light(V, N, L, 32, vols, ref["descs"], po_last, cb_last, ref["shadow"], None, vis, cfg["maxLight"], mine_lm)
        o.load().mvo_set_light_block_range(0, 0xffffffff)

        def block_voxels(b):
            bx, by, bz = b % B, (b // B) % B, b // (B * B)
            z, y, x = np.meshgrid(np.arange(bz * 4, bz * 4 + 4), np.arange(by * 4, by * 4 + 4), np.arange(bx * 4, bx * 4 + 4), indexing="ij")
            ok = (z < L) & (y < L) & (x < L)
            return z, y, x, ok

        share = np.zeros((per, 64), np.uint32)
        for b in range(lo, hi):
            z, y, x, ok = block_voxels(b)
            share[b - lo].reshape(4, 4, 4)[ok] = mine_lm[relit][z[ok], y[ok], x[ok]]
        words = torch.empty(world * per * 64, dtype=torch.int32)
        dist.all_gather_into_tensor(words, torch.from_numpy(share.view(np.int32).ravel().copy()))
        words = words.numpy().view(np.uint32).reshape(world * per, 64)
        for b in range(total):
            z, y, x, ok = block_voxels(b)
            lms[relit][z[ok], y[ok], x[ok]] = words[b].reshape(4, 4, 4)[ok]
        ref = dict(ref, cb=cb_last, po=po_last)
        owner = D.deal_cube_volumes(info, vis, V, N, world)
        mine = [int(v) for v in cube if owner[int(v)] == rank]
        cm, cd, st = o.march_view(V, N, L, W, H, vols, lms, ref["depth"], ref["po"], ref["cb"], info, mine)
        # compact exchange: marched faces only, back to back per owner
        off, biggest = D.compact_exchange_layout(info, vis, owner, V, N, world)
        shard = np.zeros(max(16, (biggest * 12 + 15) & ~15), np.uint8)
        chain_off = o.cube_mip_offset

        def faces(v):
            d = max(1, N >> int(info[v, 0]))
            return [f for f in range(6) if int(info[v, 2]) & (1 << f)], d

        for v in mine:
            fs, d = faces(v)
            n = len(fs) * d * d
            base = int(off[v]) * 12
            col = np.concatenate([cm[v][chain_off(N, int(info[v, 0])) + f * d * d:chain_off(N, int(info[v, 0])) + (f + 1) * d * d] for f in fs])
            dep = np.concatenate([cd[v][chain_off(N, int(info[v, 0])) + f * d * d:chain_off(N, int(info[v, 0])) + (f + 1) * d * d] for f in fs])
            shard[base:base + n * 8] = np.ascontiguousarray(col).view(np.uint8).ravel()
            shard[base + n * 8:base + n * 12] = np.ascontiguousarray(dep).view(np.uint8).ravel()
        recv = torch.empty(shard.size * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, torch.from_numpy(shard))
        recv = recv.numpy()
        for v in (int(x) for x in cube):
            r = int(owner[v])
            if r == rank:
                continue
            fs, d = faces(v)
            n = len(fs) * d * d
            base = r * shard.size + int(off[v]) * 12
            col = recv[base:base + n * 8].view(np.uint16).reshape(n, 4)
            dep = recv[base + n * 8:base + n * 12].view(np.float32)
            for k, f in enumerate(fs):
                o0 = chain_off(N, int(info[v, 0])) + f * d * d
                cm[v][o0:o0 + d * d] = col[k * d * d:(k + 1) * d * d]
                cd[v][o0:o0 + d * d] = dep[k * d * d:(k + 1) * d * d]
        # composite of this rank's tile rows into its block of the rank-major image, in-place-style all-gather of equal blocks
        T = (H + 7) // 8
        per_rank = (T + world - 1) // world
        block = np.zeros((per_rank * 8, W, 4), np.uint16)
        for j, t in enumerate(D.tile_rows_of_rank(H, rank, world)):
            r0, r1 = t * 8, min(t * 8 + 8, H)
            img, _ = o.composite(V, N, L, W, H, vols, lms, ref["depth"], ref["po"], ref["cb"], info, vis, cm, cd, r0, r1)
            block[j * 8:j * 8 + (r1 - r0)] = img[r0:r1]
        stored = torch.empty(world * block.size * 2, dtype=torch.uint8)
        dist.all_gather_into_tensor(stored, torch.from_numpy(block.view(np.uint8).ravel().copy()))
        stored = stored.numpy().view(np.uint16).reshape(world * per_rank * 8, W, 4)
        full = np.zeros((H, W, 4), np.uint16)
        for t in range(T):
            r0, r1 = t * 8, min(t * 8 + 8, H)
            s0 = D.stored_tile_row(t, H, world) * 8
            full[r0:r1] = stored[s0:s0 + (r1 - r0)]
        samples = torch.tensor([st["view_samples"], st["rays"]], dtype=torch.int64)
        dist.all_reduce(samples)
        if rank == 0:
            np.savez(out_path, full=full, cm=cm, cd=cd, samples=samples.numpy(), owner=owner, lms=lms)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_protocol_with_the_loops_work_deal(oracle, tmp_path, world):
    """The partition mv_dist_frame uses, at protocol level on CPU: `world` gloo ranks march their dealt volumes on the oracle, exchange
    the marched faces in the compact layout, composite their tile rows and gather the rank-major image -- it equals one process."""
    from multivolumes_amd import scene
    from multivolumes_amd import raycaster as rc
    from tests.helpers import oracle_pipeline
    out_path = str(tmp_path / "rank0.npz")
    mp.spawn(_worker_dealt, args=(world, _free_port(), out_path), nprocs=world, join=True)
    got = np.load(out_path)
    cfg = dict(scene.CONFIGS["T1"], W=96, H=88)
    vols = [rc.synth_volume(cfg["N"], k) for k in range(cfg["V"])]
    ref = oracle_pipeline(oracle, cfg, vols, 32, eye=(24.0, 16.0, -52.0))
    assert len(ref["cube"]) >= 2 and len(set(int(got["owner"][int(v)]) for v in ref["cube"])) >= 2, "the frame's volumes must be spread over ranks"
    assert np.array_equal(got["lms"], ref["lightMaps"]), "the light march shared out by voxel blocks assembles the single-process light maps"
    assert np.array_equal(got["cm"], ref["cubeMaps"]) and np.array_equal(got["cd"], ref["cubeDepths"])
    img, _ = oracle.composite(cfg["V"], cfg["N"], cfg["L"], cfg["W"], cfg["H"], vols, ref["lightMaps"], ref["depth"], ref["po"], ref["cb"],
                              ref["info"], ref["vis"], ref["cubeMaps"], ref["cubeDepths"])
    assert np.array_equal(got["full"], img)
    assert int(got["samples"][0]) == ref["stats"]["view_samples"] and int(got["samples"][1]) == ref["stats"]["rays"]


def test_tile_deal_is_a_partition_of_every_face():
    """Peer-store transport, several frames per launch: the tiles of a marched cube face are dealt over the ranks in groups of whole tile rows
    (csrc/mv_kernels.hpp dealtGroup / dealtTile; include/multivolumes.h mv_dealt_tiles).  Whatever the face size, the world size and the volume:
    every tile belongs to exactly one rank, a rank's k-th tiles are distinct, the counts the cull sizes its work list with are the lists'
    lengths, and power-of-two worlds get equal shares of faces their size divides.  (Pure host arithmetic: runs without a GPU.)"""
    import ctypes as C
    from multivolumes_amd import _lib
    lib = _lib.load()
    buf = (C.c_uint32 * 4096)()
    for tiles_x in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 20, 32, 40):
        T = tiles_x * tiles_x
        for world in range(1, 9):
            for v in (0, 1, 2, 5, 7, 8, 13, 63):
                seen = []
                counts = []
                for rank in range(world):
                    n = lib.mv_dealt_tiles(tiles_x, rank, world, v, buf, 4096)
                    assert n == lib.mv_dealt_tiles(tiles_x, rank, world, v, None, 0), "the count does not depend on the buffer"
                    mine = list(buf[:n])
                    assert len(set(mine)) == n and all(t < T for t in mine), (tiles_x, world, v, rank)
                    assert mine == sorted(mine), "a rank marches its tiles in face order"
                    seen += mine
                    counts.append(n)
                assert sorted(seen) == list(range(T)), (tiles_x, world, v)
                if world in (1, 2, 4, 8) and tiles_x % (2 * world) == 0:
                    assert len(set(counts)) == 1, (tiles_x, world, v, counts)   # two-row groups, equally many per rank
                    # ... and they are whole rows: a rank's tiles come in runs of 2 * tiles_x
                    n = lib.mv_dealt_tiles(tiles_x, 0, world, v, buf, 4096)
                    runs = [buf[i] for i in range(0, n, 2 * tiles_x)]
                    assert all(r % (2 * tiles_x) == 0 for r in runs), (tiles_x, world, v)
    assert lib.mv_dealt_tiles(0, 0, 1, 0, buf, 16) == 0 and lib.mv_dealt_tiles(4, 3, 3, 0, buf, 16) == 0   # (nonsense in: nothing out)


def test_light_rebuild_range_for_every_light_grid_and_world():
    """Peer-store transport, plain light push: the range of cells a rank rebuilds from the other ranks' texels (csrc/mv_kernels.hpp
    lightRebuildRange -- the arithmetic peerRebuildLightRole runs on the device -- through include/multivolumes.h mv_light_rebuild_range) for EVERY
    light grid mv_create accepts and every world size up to 64: the ranks' slabs tile the map once and agree with the numpy restatement, a rank
    rebuilds exactly the cells outside its slab, and no index it forms leaves the map.  Round 5's form wrapped for a rank whose slab would start
    beyond the map (L = 48 on 8 ranks: 129 024 cells of a 110 592-cell map; T2 on 6 or 7 ranks, T1 on 5): those cases are named.
    (Pure host arithmetic: runs without a GPU.)"""
    import ctypes as C
    from multivolumes_amd import _lib
    lib = _lib.load()
    out = (C.c_uint32 * 5)()
    wrapped_in_round_5 = 0
    for L in list(range(2, 131)) + [191, 192, 255, 256, 257, 383, 512, 639, 640]:
        plane, B = L * L, (L + 3) // 4
        for world in range(1, 65):
            covered = 0
            per = (B + world - 1) // world
            for rank in range(world):
                assert lib.mv_light_rebuild_range(L, rank, world, out) == 0
                z0, z1, before, skip, total = (int(x) for x in out)
                a, b = D.light_layers_of_rank(L, rank, world)
                assert (z1 - z0) == (b - a) and (b == a or (z0, z1) == (a, b)), (L, world, rank)
                assert 0 <= z0 <= z1 <= L and before == z0 * plane and skip == (z1 - z0) * plane and skip + total == L * plane, (L, world, rank)
                # the last cell index the kernel forms: k = total - 1 -> k (if below `before`) or k + skip: inside the map, outside the slab
                if total:
                    last = total - 1 if total - 1 < before else total - 1 + skip
                    assert last < L * plane and not (z0 * plane <= last < z1 * plane), (L, world, rank)
                covered += z1 - z0
                if rank * per * 4 > L:
                    wrapped_in_round_5 += 1
                    assert z0 == z1 == L and total == L * plane
            assert covered == L, (L, world)
    assert wrapped_in_round_5 > 1000
    for L, world, rank in ((48, 8, 7), (32, 6, 5), (32, 7, 5), (32, 7, 6), (24, 5, 4), (64, 7, 6)):  # ADVICE r05's cases
        assert lib.mv_light_rebuild_range(L, rank, world, out) == 0 and tuple(out) == (L, L, L ** 3, 0, L ** 3), (L, world, rank, tuple(out))
    assert lib.mv_light_rebuild_range(1, 0, 1, out) != 0 and lib.mv_light_rebuild_range(96, 8, 8, out) != 0 and lib.mv_light_rebuild_range(96, 0, 65, out) != 0
