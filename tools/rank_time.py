#!/usr/bin/env python3
"""tools/rank_time.py — load balance of the tile split without a multi-GPU node: renders every rank's share
(tiles t % world == rank) of config 3 for world sizes 1, 2, 4, 8 on ONE GPU and prints the per-rank times.
Measured on MI355X (round 1 end): 187.4 | 94.2 | 47.5 | 25.3 ms max per rank = 99 / 99 / 93 % of ideal at 2 / 4 / 8
ranks (the shares are even to 2 %; a rank's 1/8 frame pays the fixed per-launch costs), then the gather + unpack step."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from raytracer_amd import capi, scenes
from raytracer_amd.render import Renderer
world = scenes.config3_scene(3840, 2160, 16)
r = Renderer(0); r.upload(world.flatten())
dev = torch.device("cuda:0")
for W in (1, 2, 4, 8):
    ts = []
    for rank in range(W):
        p = world.params(seed=1); p.tile_rank, p.tile_world = rank, W
        n = r.owned_tiles(p, rank, W)
        slab = torch.zeros((n, 64*64, 3), dtype=torch.float32, device=dev)
        r.render_device(world.camera, p, slab.data_ptr(), None, None, None, None)
        torch.cuda.synchronize(); t = time.perf_counter()
        r.render_device(world.camera, p, slab.data_ptr(), None, None, None, None)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    if W == 1: t1 = max(ts)
    print(W, "ranks: per-rank ms", [round(x, 1) for x in ts], "max", round(max(ts), 1), "ideal", round(t1 / W, 1), flush=True)
