"""Multi-GPU plumbing for the path: one process per GPU, every rank holds every volume texture (marchL couples all
volumes), cull replicated, the frame's work shared out, one all-gather per frame of the marched cube faces (colour + depth), then
each rank composites its share of the screen rows and the rows are gathered.

The frame loop itself lives in the library (mv_dist_frame, include/multivolumes.h); init_frame_loop() below attaches it, with
RCCL opened by the library or -- for gloo tests -- torch.distributed collectives handed over as callbacks.  Inside the loop the
cull deals the work per frame (cube-map volumes by estimated cost, screen rows by 8-row tile row): deal_cube_volumes,
compact_exchange_layout, tile_rows_of_rank and stored_tile_row restate that rule in numpy.
The other pure functions (owned_volumes, exchange_layout, row_range, pack / unpack mirrors) state the static partition (v % world,
contiguous row bands) and the slot-based layout of the building blocks mv_exchange_pack / mv_exchange_unpack / mv_shard_rows, so
that CPU tests can check both protocols over gloo without a GPU.
"""
import os

import numpy as np

from . import _lib

NUM_CUBE_MIP = 5


def owned_volumes(V, rank, world):
    """Volumes rank `rank` marches: v % world == rank (matches DeviceState.shardIndex / shardCount)."""
    return [v for v in range(V) if v % world == rank]


def slots_per_shard(V, world):
    return (V + world - 1) // world


def exchange_layout(N, volume_info, visible, V, world):
    """(slot_dim, slot_bytes, shard_bytes) exactly as mv_exchange_layout computes them on the device:
    slot edge = N >> (min selected mip over the visible volumes); slot = 6 d^2 RGBA16F texels + 6 d^2 R32F depths."""
    mips = [int(volume_info[v][0]) for v in visible]
    min_mip = min(mips) if mips else 0
    if min_mip >= NUM_CUBE_MIP:
        min_mip = 0
    d = max(1, N >> min_mip)
    slot_bytes = 6 * d * d * (8 + 4)
    return d, slot_bytes, slot_bytes * slots_per_shard(V, world)


def row_range(H, rank, world):
    """Band of screen rows rank `rank` composites (8-row aligned so that 8x8 composite tiles never straddle ranks)."""
    tiles = (H + 7) // 8
    t0, t1 = (tiles * rank) // world, (tiles * (rank + 1)) // world
    return min(t0 * 8, H), min(t1 * 8, H)


def pack_shard_numpy(N, V, rank, world, volume_info, visible, cube_maps, cube_depths, slot_dim, chain_texels, mip_offset):
    """numpy mirror of exchangePackKernel: cube_maps [V, chain, 4] u16, cube_depths [V, chain] f32 -> uint8[shard_bytes]."""
    n_slot = 6 * slot_dim * slot_dim
    out = np.zeros(slots_per_shard(V, world) * n_slot * 12, np.uint8)
    vis = set(int(v) for v in visible)
    for j, v in enumerate(owned_volumes(V, rank, world)):
        if v not in vis:
            continue
        mip = int(volume_info[v][0])
        dv = max(1, N >> mip)
        n = 6 * dv * dv
        off = mip_offset(N, mip)
        base = j * n_slot * 12
        out[base:base + n * 8] = np.ascontiguousarray(cube_maps[v][off:off + n]).view(np.uint8).ravel()
        out[base + n_slot * 8:base + n_slot * 8 + n * 4] = np.ascontiguousarray(cube_depths[v][off:off + n]).view(np.uint8).ravel()
    return out


def unpack_gathered_numpy(N, V, rank, world, volume_info, visible, gathered, cube_maps, cube_depths, slot_dim, mip_offset):
    """numpy mirror of exchangeUnpackKernel: scatter the rank-major all-gather result into every non-owned volume."""
    n_slot = 6 * slot_dim * slot_dim
    per = slots_per_shard(V, world)
    vis = set(int(v) for v in visible)
    for r in range(world):
        if r == rank:
            continue
        for j, v in enumerate(owned_volumes(V, r, world)):
            if v not in vis:
                continue
            mip = int(volume_info[v][0])
            dv = max(1, N >> mip)
            n = 6 * dv * dv
            off = mip_offset(N, mip)
            base = (r * per + j) * n_slot * 12
            cube_maps[v][off:off + n] = gathered[base:base + n * 8].view(np.uint16).reshape(n, 4)
            cube_depths[v][off:off + n] = gathered[base + n_slot * 8:base + n_slot * 8 + n * 4].view(np.float32)


# ---- the work deal of the in-library frame loop (mv_dist_frame), restated on the host ----
# Inside the loop the replicated cull (csrc/mv_kernels.hip cullKernel, its last workgroup) decides per frame who marches which cube-map
# volume and who owns which screen rows; every rank derives the same answer from VolumeInfo.  These functions are that rule in numpy:
# the CPU tests run the protocol with them over gloo, a GPU test compares them with what the kernel wrote.
def deal_cube_volumes(volume_info, visible, V, N, world):
    """owner[v] for every volume.  Visible cube-map volumes are dealt by estimated march cost -- marched faces x 8x8 tiles per face x
    SmpCount (+ 1) -- longest first, each to the rank with the least so far (ties: lowest volume, lowest rank); every other volume
    keeps v % world.  (world <= 64: beyond that the kernel keeps v % world for all.)"""
    owner = np.arange(V, dtype=np.uint32) % world
    if world < 2 or world > 64:
        return owner
    vis = set(int(v) for v in visible)
    cost = np.zeros(V, np.uint64)
    for v in range(V):
        mask = int(volume_info[v, 2])
        if v not in vis or not (mask & 0x8000):
            continue
        dims = max(1, N >> int(volume_info[v, 0]))
        tiles = ((dims + 7) >> 3) ** 2
        c = bin(mask & 0x3F).count("1") * tiles * max(int(volume_info[v, 1]), 1)
        cost[v] = min(c + 1, 0xFFFFFFFF)
    load = np.zeros(world, np.uint64)
    left = cost.copy()
    for _ in range(int((cost > 0).sum())):
        v = int(np.argmax(left))          # first of the equals = lowest volume
        r = int(np.argmin(load))          # first of the equals = lowest rank
        owner[v] = r
        load[r] += left[v]
        left[v] = 0
    return owner


def compact_exchange_layout(volume_info, visible, owner, V, N, world):
    """(texel offset of every volume inside its owner's shard, texels of the largest shard): a shard holds the marched faces of the
    owner's visible cube-map volumes at their mip, ascending volume order, colour then depth per volume (exchangeCompactKernel)."""
    vis = set(int(v) for v in visible)
    off = np.zeros(V, np.uint32)
    run = np.zeros(world, np.uint64)
    for v in range(V):
        mask = int(volume_info[v, 2])
        if v not in vis or not (mask & 0x8000):
            continue
        d = max(1, N >> int(volume_info[v, 0]))
        off[v] = run[owner[v]]
        run[owner[v]] += bin(mask & 0x3F).count("1") * d * d
    return off, int(run.max()) if world else 0


def tile_rows_of_rank(H, rank, world):
    """8-row tile rows a rank marches (RayCast-scheme volumes) and composites inside the loop: every world-th one."""
    return list(range(rank, (H + 7) // 8, world))


def stored_tile_row(t, H, world):
    """Where tile row t sits in the loop's composite buffer (rank-major: every rank's rows are one contiguous block)."""
    per_rank = ((H + 7) // 8 + world - 1) // world
    return (t % world) * per_rank + t // world


def light_layers_of_rank(L, rank, world):
    """Peer-store transport (csrc/mv_kernels.hpp lightLayersOfRank): the z slices [z0, z1) of the relit light map a rank marches and pushes -- a
    slab of ceil(B / world) adjacent layers of 4 slices (B = ceil(L / 4) layers of voxel blocks), empty for ranks beyond the last layer."""
    B = (L + 3) // 4
    per = (B + world - 1) // world
    first = rank * per
    if first >= B:
        return 0, 0
    return first * 4, min((first + min(per, B - first)) * 4, L)


def light_cells_from_texels(plain):
    """The device layout of a light map from its plain R11G11B10F texels [z][y][x] (csrc/mv_kernels.hpp peerRebuildLightRole; mv_device.hpp
    "light maps"): cell (x, y, z) = {t(x, y), t(x + 1, y), t(x, y + 1), t(x + 1, y + 1)} of slice z, +1 clamped -- texels of one z only, which is
    why a rank's layers travel as plain texels (a quarter of the bytes) and every receiver rebuilds their cells from them alone."""
    import numpy as np
    ny, nx = plain.shape[1], plain.shape[2]   # (any number of z slices: a rank's slab rebuilds alone)
    y1, x1 = np.minimum(np.arange(ny) + 1, ny - 1), np.minimum(np.arange(nx) + 1, nx - 1)
    return np.stack([plain, plain[:, :, x1], plain[:, y1, :], plain[:, y1][:, :, x1]], axis=-1)


def init_peer_transport(caster, timeout_ms=None, fold=None, rows_to_root=False, deal_volumes=None, acquire=True, push_cells=False):
    """The peer-store transport (include/multivolumes.h): every rank exports IPC handles of its light maps, cube maps and frame buffers,
    torch.distributed only gathers the blobs (any backend), every rank maps the others and the kernels store into all of them.
    Every step is collective and unconditional; a vote after init + self-test decides for all ranks.  Returns True if it is attached."""
    import sys
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    err = None
    blob = bytes(_lib.MV_DIST_PEER_BYTES + 1)
    try:
        blob = caster.DistPeerExport() + b"\x01"
    except Exception as e:  # (e.g. a context without whole-frame chains)
        err = e
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    gathered = torch.empty(world * len(blob), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, mine)
    host = gathered.cpu().numpy().reshape(world, len(blob))
    ok = 1 if (err is None and bool((host[:, -1] == 1).all())) else 0
    if ok:
        try:
            caster.DistInitPeer(host[:, :-1].tobytes(), fold=fold, rows_to_root=rows_to_root, deal_volumes=deal_volumes, acquire=acquire, push_cells=push_cells)
            if timeout_ms is not None:
                caster.DistSetPeerTimeoutMs(timeout_ms)
            caster.DistSelfTest()  # patterns through every mapping and every flag, verified on every rank, twice
        except Exception as e:
            err, ok = e, 0
    vote = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN)
    if int(vote.item()) == 1:
        return True
    print(f"[rank {rank}] peer-store transport unavailable ({err if err is not None else 'another rank failed'})", file=sys.stderr)
    caster.DistShutdown()
    return False


def init_frame_loop(caster, group_main=None, group_light=None, prefer_rccl=True, prefer_peer=False, peer_timeout_ms=None, peer_fold=None, peer_rows_to_root=False, peer_deal_volumes=None, peer_acquire=True, peer_push_cells=False):
    """Attach the in-library multi-GPU frame loop (mv_dist_frame) to a sharded caster.

    prefer_peer: try the peer-store transport first (whole-frame-chain contexts on one node; returns "peer").

    With the NCCL (= RCCL) backend the library opens RCCL itself and creates its own communicators (three: one per frame chain) from ids that rank 0 makes and
    torch.distributed only broadcasts: no Python and no torch call inside the frame loop.  Otherwise (gloo: CPU-side tests, or RCCL
    could not be initialised) the collectives are torch.distributed calls on tensors that alias the library's buffers, handed to the
    library as callbacks.  Returns "rccl" or "torch-callbacks"."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    if prefer_peer and init_peer_transport(caster, peer_timeout_ms, peer_fold, peer_rows_to_root, peer_deal_volumes, peer_acquire, peer_push_cells):
        return "peer"
    if prefer_rccl and backend == "nccl":
        # Every step below is unconditional and collective, whatever failed locally: a rank never leaves the others waiting in a
        # collective it skips.  (1) rank 0 makes the ids -- zeros and a flag if it cannot -- and ALWAYS broadcasts them;
        # (2) the ranks vote (MIN) on "I can open RCCL and got valid ids" BEFORE anyone enters ncclCommInitRank, which is collective;
        # (3) init + self-test; (4) a second vote: one rank's failure moves them all to the callbacks.
        err = None
        msg = torch.zeros(_lib.MV_DIST_ID_BYTES + 1, dtype=torch.uint8, device="cuda")
        if rank == 0:
            try:
                msg[:_lib.MV_DIST_ID_BYTES].copy_(torch.frombuffer(bytearray(caster.DistUniqueId()), dtype=torch.uint8))
                msg[_lib.MV_DIST_ID_BYTES] = 1
            except Exception as e:
                err = e
        dist.broadcast(msg, src=0)
        host = msg.cpu().numpy()
        can = 1 if (int(host[_lib.MV_DIST_ID_BYTES]) == 1 and err is None and caster.DistRcclAvailable()) else 0
        vote = torch.tensor([can], dtype=torch.int32, device="cuda")
        dist.all_reduce(vote, op=dist.ReduceOp.MIN)
        if int(vote.item()) == 1:
            try:
                caster.DistInitRccl(bytes(host[:_lib.MV_DIST_ID_BYTES].tobytes()))
                caster.DistSelfTest()  # every collective the frame loop uses, verified once, before any frame
            except Exception as e:
                err = e
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                return "rccl"
        elif err is None and not can:
            err = "librccl.so could not be opened here" if int(host[_lib.MV_DIST_ID_BYTES]) == 1 else "rank 0 could not make the RCCL ids"
        import sys  # loud, and visible in the bench line ("transport: torch-callbacks")
        print(f"[rank {rank}] in-library RCCL transport unavailable ({err if err is not None else 'another rank failed'}); "
              "using torch.distributed callbacks", file=sys.stderr)
        caster.DistShutdown()
    # one group per communicator index the library uses (`which` = 0..2: include/multivolumes.h)
    # (as many as the context has frames in flight -- mv_config.frames_in_flight up to 5 -- and never fewer than the two the single-view-stream schedule uses)
    n_groups = max(3, int(getattr(caster, "framesInFlight", 3) or 3))
    groups = [group_main if group_main is not None else dist.new_group(list(range(world))),
              group_light if group_light is not None else dist.new_group(list(range(world)))]
    groups += [dist.new_group(list(range(world))) for _ in range(n_groups - 2)]

    def allgather(send, recv, nbytes, stream, which):
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                dist.all_gather_into_tensor(_device_view(recv, nbytes * world), _device_view(send, nbytes), group=groups[which])
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    def allreduce(buf, count, stream, which):
        try:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                t = _device_view(buf, count * 4).view(torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=groups[which])
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    caster.DistInitCallbacks(allgather, allreduce)
    caster.DistSelfTest()
    return "torch-callbacks"


def _device_view(ptr, nbytes):
    """torch uint8 tensor aliasing `nbytes` of device memory at `ptr` (context-owned buffer)."""
    import torch

    class _Iface:
        pass

    o = _Iface()
    o.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    return torch.as_tensor(o, device="cuda")
