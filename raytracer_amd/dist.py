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
