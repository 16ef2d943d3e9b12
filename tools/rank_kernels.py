#!/usr/bin/env python3
"""tools/rank_kernels.py — per-kernel times of rank 0's share at world sizes 1 and 8 (times x world), on one GPU:
shows which kernels pay fixed per-launch tails when a rank renders an eighth of the frame.
Measured (round 1 end): k_bvh<AO> 113.4 -> 119.4, k_shadow 11.2 -> 14.1, k_ao 46.0 -> 48.4, k_bvh<closest> 3.9 -> 5.9 ms."""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from raytracer_amd import capi, scenes
from raytracer_amd.render import Renderer
world = scenes.config3_scene(3840, 2160, 16)
r = Renderer(0); r.upload(world.flatten())
dev = torch.device("cuda:0")
for W in (1, 8):
    p = world.params(seed=1); p.tile_rank, p.tile_world = 0, W
    n = r.owned_tiles(p, 0, W)
    slab = torch.zeros((n, 64*64, 3), dtype=torch.float32, device=dev)
    r.render_device(world.camera, p, slab.data_ptr(), None, None, None, None)
    st = capi.rt_stats(); p.flags = capi.RT_FLAG_TIMING
    torch.cuda.synchronize(); t=time.perf_counter()
    r.render_device(world.camera, p, slab.data_ptr(), None, None, st, None)
    torch.cuda.synchronize(); wall=(time.perf_counter()-t)*1e3
    k = {nme: round(float(st.ms_kernel[i]) * (W if W>1 else 1), 2) for i, nme in enumerate(capi.RT_K_NAMES)}
    print(W, "wall", round(wall,2), "ms_total", round(st.ms_total,2), "kernel ms x W:", k, "sum", round(sum(st.ms_kernel),2))
