#!/usr/bin/env python3
"""tools/builder_compare.py — SURVEY 8f N1 measurement: host binned-SAH builder vs device LBVH builder.
For config 3 (one 0.5 M-triangle mesh) and config 5 (2500 x 4096-triangle meshes) prints the commit time, the
builder's own time, which builder was used (the device builder falls back when its tree is too deep), the
frame time, the nodes / triangles visited per BVH ray and a frame checksum (equal for both builders)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    r = Renderer(0)
    cases = [("config3 1920x1080 4spp", lambda: scenes.config3_scene(1920, 1080, 4)),
             ("config5 960x540 4spp", lambda: scenes.config5_scene(960, 540, 4, instances=2500))]
    for name, make in cases:
        world = make()
        flat = world.flatten()
        for builder, bname in ((capi.RT_BUILDER_HOST_SAH, "host_sah"), (capi.RT_BUILDER_GPU_LBVH, "gpu_lbvh")):
            t0 = time.perf_counter()
            r.upload(flat, builder=builder)
            commit = time.perf_counter() - t0
            used, secs = r.builder_used()
            params = world.params(seed=1)
            frame = torch.zeros((params.height, params.width, 3), dtype=torch.float32, device=dev)
            st = capi.rt_stats()
            params.flags = capi.RT_FLAG_STATS
            r.render_device(world.camera, params, frame.data_ptr(), None, None, st, None)
            params.flags = 0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r.render_device(world.camera, params, frame.data_ptr(), None, None, None, None)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) * 1e3
            print(json.dumps({
                "scene": name, "triangles": int(flat.arrays["triangles"].shape[0]), "requested": bname,
                "used": "gpu_lbvh" if used == capi.RT_BUILDER_GPU_LBVH else "host_sah",
                "commit_s": round(commit, 3), "bvh_build_s": round(secs, 4), "frame_ms": round(ms, 2),
                "mrays_per_s": round(st.rays / ms / 1e3, 1),
                "nodes_per_bvh_ray": [round(st.nodes[k] / max(1, st.bvh_rays[k]), 1) for k in (0, 2)],
                "tris_per_bvh_ray": [round(st.tris[k] / max(1, st.bvh_rays[k]), 1) for k in (0, 2)],
                "checksum": int(frame.view(torch.int32).to(torch.int64).sum().item()),
            }), flush=True)
    r.close()


if __name__ == "__main__":
    main()
