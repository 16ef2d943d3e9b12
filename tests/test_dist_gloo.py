"""The N > 1 path on CPU: two gloo ranks each own the tiles t % 2 == rank of a frame, rank 0
gathers the slabs (the one exchange step of the path) and scatters them into the frame."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest  # noqa: F401  (puts the repo root on sys.path)
from raytracer_amd import capi, dist as rdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = capi.default_params(200, 136)
        rng = np.random.default_rng(42)
        truth = rng.uniform(size=(136, 200, 3)).astype(np.float32)   # stands for "what the renderer produces"
        max_owned = rdist.owned_tiles(params, 0, world)
        slab = torch.from_numpy(rdist.pack_slab_numpy(truth, params, rank, world, num_tiles=max_owned))
        frame = np.zeros_like(truth)

        def unpack(src, t):
            rdist.unpack_slab_numpy(t.numpy(), params, src, world, frame)
        is_root = rdist.gather_frame(slab, params, rank, world, unpack)
        if is_root:
            np.save(out_path, np.array([np.array_equal(frame, truth)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_tile_gather(tmp_path):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert np.load(out)[0]


def test_three_rank_tile_gather_uneven(tmp_path):
    out = str(tmp_path / "ok3.npy")
    mp.spawn(_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert np.load(out)[0]


def test_eight_rank_tile_gather_uneven(tmp_path):
    """world 8 over 12 tiles: ranks 0-3 own two tiles, ranks 4-7 one (the slabs are padded to rank 0's count)."""
    out = str(tmp_path / "ok8.npy")
    mp.spawn(_worker, args=(8, _free_port(), out), nprocs=8, join=True)
    assert np.load(out)[0]


def _worker_pipelined(rank, world, port, out_path):
    """bench.py's N > 1 loop: two slabs, the gather of frame k in flight while frame k+1 is produced."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = capi.default_params(200, 136)
        rng = np.random.default_rng(7)
        truths = [rng.uniform(size=(136, 200, 3)).astype(np.float32) for _ in range(4)]
        max_owned = rdist.owned_tiles(params, 0, world)
        slabs = [torch.zeros((max_owned, rdist.TILE * rdist.TILE, 3)) for _ in range(2)]
        lists = [[torch.zeros_like(slabs[0]) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
        frame = np.zeros_like(truths[0])
        ok = []

        def unpack(src, t):
            rdist.unpack_slab_numpy(t.numpy(), params, src, world, frame)
        pending = None
        for k, truth in enumerate(truths):
            slabs[k % 2].copy_(torch.from_numpy(rdist.pack_slab_numpy(truth, params, rank, world, num_tiles=max_owned)))   # "render"
            nxt = rdist.gather_frame_start(slabs[k % 2], params, rank, world, unpack, gather_list=lists[k % 2])
            if pending is not None and pending.finish():
                ok.append(np.array_equal(frame, truths[k - 1]))
            pending = nxt
        if pending.finish():
            ok.append(np.array_equal(frame, truths[-1]))
        if rank == 0:
            np.save(out_path, np.array([len(ok) == len(truths) and all(ok)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_two_slabs(tmp_path):
    out = str(tmp_path / "okp.npy")
    mp.spawn(_worker_pipelined, args=(3, _free_port(), out), nprocs=3, join=True)
    assert np.load(out)[0]
