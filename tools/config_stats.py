#!/usr/bin/env python3
"""tools/config_stats.py — traversal counters per ray class (closest / shadow / AO) of configs 3 and 5 at reduced size:
gate boxes tested, BVH nodes fetched, triangles tested per ray, and the share of rays that reach the triangle-BVH kernel."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from raytracer_amd import capi, scenes  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--instances", type=int, default=2500)
args = ap.parse_args()
r = Renderer(0)
for name, world in (("config3", scenes.config3_scene(1920, 1080, 4)),
                    ("config5", scenes.config5_scene(1920, 1080, 4, instances=args.instances))):
    r.upload(world.flatten())
    p = world.params(seed=1)
    p.flags = capi.RT_FLAG_STATS
    frame = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda:0")
    st = capi.rt_stats()
    r.render_device(world.camera, p, frame.data_ptr(), None, None, st, torch.cuda.current_stream().cuda_stream)
    rays = [st.rays_primary + st.rays_bounce, st.rays_shadow, st.rays_ao]
    out = {"config": name}
    for k, cls in enumerate(("closest", "shadow", "ao")):
        n = max(1, rays[k])
        out[cls] = {"rays": int(rays[k]), "gates_per_ray": round(st.gates[k] / n, 2), "bvh_ray_share": round(st.bvh_rays[k] / n, 3),
                    "nodes_per_bvh_ray": round(st.nodes[k] / max(1, st.bvh_rays[k] if k != 1 else rays[k]), 2),
                    "tris_per_bvh_ray": round(st.tris[k] / max(1, st.bvh_rays[k] if k != 1 else rays[k]), 2)}
    print(json.dumps(out), flush=True)
r.close()
