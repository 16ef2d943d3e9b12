"""Multi-GPU tile split: one process per GPU, scene replicated, the 64x64 tiles t of the render
rectangle with t % world == rank rendered by `rank`, slabs gathered to rank 0 (RCCL over xGMI
when the tensors live on GPUs; any torch.distributed backend works) and scattered into the frame.

The reference shares one std::vector between its pool threads (main.cpp:1656, 1809-1825): the
gather is the only exchange step this path has.  The tile arithmetic here mirrors rt_owned_tiles /
k_unpack in csrc/rt_gpu.hip; `unpack_slab_numpy` is the host statement of that scatter used by
the CPU (gloo) tests.
"""
import numpy as np

from . import capi

TILE = capi.RT_TILE


def tile_grid(params):
    tx = (params.x1 - params.x0 + TILE - 1) // TILE
    ty = (params.y1 - params.y0 + TILE - 1) // TILE
    return tx, ty


def owned_tiles(params, rank, world):
    tx, ty = tile_grid(params)
    total = tx * ty
    return (total - rank + world - 1) // world if total > rank else 0


def tile_rect(params, rank, world, k):
    """Pixel rectangle (x0, y0, x1, y1) of the k-th tile owned by `rank`, clipped to the render rectangle."""
    tx, ty = tile_grid(params)
    t = rank + k * world
    x0 = params.x0 + (t % tx) * TILE
    y0 = params.y0 + (t // tx) * TILE
    return x0, y0, min(x0 + TILE, params.x1), min(y0 + TILE, params.y1)


def pack_slab_numpy(frame, params, rank, world, num_tiles=None):
    """frame[H,W,3] -> slab[num_tiles, TILE*TILE, 3] of the tiles owned by `rank` (zero padded)."""
    n = owned_tiles(params, rank, world)
    slab = np.zeros((num_tiles if num_tiles is not None else n, TILE * TILE, 3), np.float32)
    for k in range(n):
        x0, y0, x1, y1 = tile_rect(params, rank, world, k)
        blk = np.zeros((TILE, TILE, 3), np.float32)
        blk[: y1 - y0, : x1 - x0] = frame[y0:y1, x0:x1]
        slab[k] = blk.reshape(-1, 3)
    return slab


def unpack_slab_numpy(slab, params, rank, world, frame):
    """Scatter one rank's slab into frame[H,W,3] (host mirror of rt_unpack_tiles)."""
    for k in range(owned_tiles(params, rank, world)):
        x0, y0, x1, y1 = tile_rect(params, rank, world, k)
        blk = np.asarray(slab[k]).reshape(TILE, TILE, 3)
        frame[y0:y1, x0:x1] = blk[: y1 - y0, : x1 - x0]
    return frame


def gather_frame(slab, params, rank, world, unpack, group=None, gather_list=None):
    """Collective: every rank passes its slab (torch tensor [max_owned, TILE*TILE, 3]); rank 0 gets
    them all (dist.gather = grouped send/recv: 7 concurrent point-to-point transfers on xGMI) and
    calls unpack(src_rank, slab_tensor) for each.  Returns True on rank 0."""
    import torch
    import torch.distributed as dist
    if world == 1:
        unpack(0, slab)
        return True
    if gather_list is None and rank == 0:
        gather_list = [torch.empty_like(slab) for _ in range(world)]
    dist.gather(slab, gather_list, dst=0, group=group)
    if rank == 0:
        for src in range(world):
            unpack(src, gather_list[src])
        return True
    return False


class PendingGather:
    """A gather in flight (gather_frame_start): finish() waits for it and scatters the slabs on rank 0."""
    def __init__(self, work, gather_list, rank, world, unpack, keep):
        self.work, self.gather_list, self.rank, self.world, self.unpack, self.keep = work, gather_list, rank, world, unpack, keep

    def finish(self):
        if self.work is not None:
            self.work.wait()   # device tensors: the current stream waits for the collective, the host does not
        if self.rank == 0:
            for src in range(self.world):
                self.unpack(src, self.gather_list[src])
        self.keep = None
        return self.rank == 0


def gather_frame_start(slab, params, rank, world, unpack, group=None, gather_list=None, force_collective=False):
    """The same exchange as gather_frame, split in two so that a caller with two slabs can render frame k+1 while
    frame k's slabs travel: start the gather (async_op), return a PendingGather; call finish() on it later — after
    queueing the next render.  `slab` and `gather_list` must stay untouched until finish().  world == 1 runs no
    collective unless force_collective (how the RCCL code path is exercised on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    if world == 1 and not force_collective:
        return PendingGather(None, [slab], 0, 1, unpack, slab)
    if gather_list is None and rank == 0:
        gather_list = [torch.empty_like(slab) for _ in range(world)]
    work = dist.gather(slab, gather_list if rank == 0 else None, dst=0, group=group, async_op=True)
    return PendingGather(work, gather_list, rank, world, unpack, slab)
