"""Multi-GPU plumbing for the path: one process per GPU, volumes sharded v % world, every rank holds
every volume texture (marchL couples all volumes), cull replicated, marchL / marchV on owned volumes only, one RCCL
all-gather per frame of the selected-mip cube faces (colour + depth), then each rank composites its band of screen rows.

torch.distributed is plumbing here: it moves the bytes the C ABI packs (mv_exchange_pack / mv_exchange_unpack).
The pure functions (owned_volumes, exchange_layout, row_range) mirror the device-side layout so that CPU tests can
check the protocol over gloo without a GPU.
"""
import os

import numpy as np

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
    d = N >> min_mip
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
        dv = N >> mip
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
            dv = N >> mip
            n = 6 * dv * dv
            off = mip_offset(N, mip)
            base = (r * per + j) * n_slot * 12
            cube_maps[v][off:off + n] = gathered[base:base + n * 8].view(np.uint16).reshape(n, 4)
            cube_depths[v][off:off + n] = gathered[base + n_slot * 8:base + n_slot * 8 + n * 4].view(np.float32)


class CubeFaceExchange:
    """Per-rank driver of the per-frame exchange on the GPU: pack -> all_gather_into_tensor (RCCL) -> unpack.
    Buffers are allocated once per layout and live in HBM; nothing crosses the host."""

    def __init__(self, caster, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.caster, self.group = caster, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._bytes = None
        self.send = self.recv = None

    def run(self, volume_info, visible):
        """volume_info / visible: this frame's (replicated) cull results, e.g. from caster.ReadBackCull()."""
        if self.world == 1:
            return 0
        c = self.caster
        d, slot_bytes, shard_bytes = exchange_layout(c.gridSize, volume_info, visible, c.numVolumes, self.world)
        if shard_bytes != self._bytes:
            self.send = self.torch.empty(shard_bytes, dtype=self.torch.uint8, device="cuda")
            self.recv = self.torch.empty(shard_bytes * self.world, dtype=self.torch.uint8, device="cuda")
            self._bytes = shard_bytes
        c.ExchangePack(self.send.data_ptr(), d)
        self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        c.ExchangeUnpack(self.recv.data_ptr(), d)
        return shard_bytes * self.world


class LightMapExchange:
    """Screen-space (RayCast) layers read the light map of their volume (RayCast.hlsli:76), and every rank marches its own rows
    of every such volume inside the view march.  ONE volume is relit per frame -- visible[frameIdx % count], by the rank that
    owns it (CSRayMarchL.hlsl:29-33; the cull is replicated, so every rank knows which) -- so per frame at most that one light
    map has to travel, and only if its volume is on the screen-space scheme: a broadcast from its owner before the view march.
    Frames whose relit volume is on the cube-map scheme move nothing."""

    def __init__(self, caster, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.caster, self.group = caster, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.stage = None

    @staticmethod
    def relit_volume(visible, frame_idx, V):
        """The volume this frame relights (the same on every rank)."""
        return int(visible[frame_idx % len(visible)]) if len(visible) else frame_idx % V

    def run(self, volume_info, visible, frame_idx):
        if self.world == 1:
            return 0
        v = self.relit_volume(visible, frame_idx, self.caster.numVolumes)
        if not len(visible) or (int(volume_info[v][2]) & 0x8000):
            return 0  # on the cube-map scheme (or nothing visible): only its owner's view march reads that light map
        ptr, n = self.caster.GetBuffer(1, v)  # MV_BUF_LIGHT_MAP (device layout); the context's stream waits for the light march
        view = _device_view(ptr, n)
        # the collective runs on a torch-owned staging tensor (context buffers are only copied from / to)
        if self.stage is None or self.stage.numel() != n:
            self.stage = self.torch.empty(n, dtype=self.torch.uint8, device="cuda")
        owner = v % self.world
        if self.rank == owner:
            self.stage.copy_(view)
        self.dist.broadcast(self.stage, src=owner, group=self.group)
        if self.rank != owner:
            view.copy_(self.stage)
        return n


_IN_PLACE_REFUSED = []  # set once if torch rejects the in-place all-gather (argument checks raise before anything is enqueued)


def gather_composite_rows(caster, H, W, rank, world, group=None):
    """After every rank composited its row band: all-gather the bands so every rank's composite buffer holds the frame."""
    import torch
    import torch.distributed as dist
    ptr, nbytes = caster.GetBuffer(5)  # MV_BUF_COMPOSITE: W x H RGBA16F
    bands = [row_range(H, r, world) for r in range(world)]
    max_rows = max(b[1] - b[0] for b in bands)
    row_bytes = W * 8
    r0, r1 = bands[rank]
    frame = _device_view(ptr, nbytes)  # aliases the context's buffer, no copy
    if os.environ.get("MV_INPLACE_GATHER") == "1" and all(b[1] - b[0] == max_rows for b in bands) and dist.get_backend(group) == "nccl" and not _IN_PLACE_REFUSED:
        # equal bands: the frame buffer IS the all-gather output and this rank's band its input, in place (no staging copies);
        # NCCL / RCCL define that (sendbuff == recvbuff + rank * count) -- if this torch build objects to the overlap, stage instead
        try:
            dist.all_gather_into_tensor(frame[:world * max_rows * row_bytes], frame[r0 * row_bytes:r1 * row_bytes], group=group)
            return world * max_rows * row_bytes
        except (RuntimeError, ValueError):
            _IN_PLACE_REFUSED.append(True)
    send = torch.zeros(max_rows * row_bytes, dtype=torch.uint8, device="cuda")
    send[:(r1 - r0) * row_bytes].copy_(frame[r0 * row_bytes:r1 * row_bytes])
    recv = torch.empty(world * max_rows * row_bytes, dtype=torch.uint8, device="cuda")
    dist.all_gather_into_tensor(recv, send, group=group)
    for r, (a, b) in enumerate(bands):
        if r != rank and b > a:
            frame[a * row_bytes:b * row_bytes].copy_(recv[r * max_rows * row_bytes:r * max_rows * row_bytes + (b - a) * row_bytes])
    return world * max_rows * row_bytes


def _device_view(ptr, nbytes):
    """torch uint8 tensor aliasing `nbytes` of device memory at `ptr` (context-owned buffer)."""
    import torch

    class _Iface:
        pass

    o = _Iface()
    o.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
    return torch.as_tensor(o, device="cuda")
