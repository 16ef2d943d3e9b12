#!/usr/bin/env python3
"""tools/time_configs.py — wall-clock sanity of the BASELINE.json configs other than the bench's (config 3),
at reduced image sizes so the whole sweep takes seconds on one MI355X.  Not a bench line: bench.py is.

    python tools/time_configs.py [--scale 4]
prints one JSON line per config: scene-commit seconds, frame ms, Mrays/s, per-kernel ms.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from raytracer_amd import capi, scenes  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=4, help="divide each config's image size by this")
    ap.add_argument("--instances", type=int, default=2500)
    ap.add_argument("--only", default=None, help="one of config1, config2, config4, config5")
    ap.add_argument("--full", action="store_true", help="BASELINE.json's stated sizes: config 4 at 3840x2160, 64 spp, depth 8; config 5 at 7680x4320 and --spp5 samples")
    ap.add_argument("--spp5", type=int, default=16, help="--full: samples per pixel of config 5 (stated: 256)")
    ap.add_argument("--no-stats-pass", action="store_true", help="skip the counting pass (ray counts come from the timed pass; no nodes / gates per ray)")
    args = ap.parse_args()
    s = args.scale
    cases = [
        ("config1", lambda: scenes.default_scene(1000 // s * 2, 1000 // s * 2)),
        ("config2", lambda: scenes.config2_scene(1920 // s * 2, 1080 // s * 2)),
        ("config4", lambda: scenes.config4_scene(3840 // s, 2160 // s, 16)),
        ("config5", lambda: scenes.config5_scene(7680 // (2 * s), 4320 // (2 * s), 4, instances=args.instances)),
    ]
    if args.full:
        cases = [
            ("config4", lambda: scenes.config4_scene(3840, 2160, 64)),
            ("config5", lambda: scenes.config5_scene(7680, 4320, args.spp5, instances=args.instances)),
        ]
    if args.only:
        cases = [c for c in cases if c[0] == args.only]
    dev = torch.device("cuda:0")
    r = Renderer(0)
    for name, make in cases:
        t0 = time.perf_counter()
        world = make()
        flat = world.flatten()
        t1 = time.perf_counter()
        r.upload(flat)
        t2 = time.perf_counter()
        params = world.params(seed=1, max_depth=8) if name == "config4" else world.params(seed=1)   # config 4 is stated at bounce depth 8
        frame = torch.zeros((params.height, params.width, 3), dtype=torch.float32, device=dev)
        st = capi.rt_stats()
        params.flags = capi.RT_FLAG_STATS
        if not args.no_stats_pass:
            r.render_device(world.camera, params, frame.data_ptr(), None, None, st, None)
        rays = st.rays
        st2 = capi.rt_stats()
        params.flags = capi.RT_FLAG_TIMING
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        r.render_device(world.camera, params, frame.data_ptr(), None, None, st2, None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t3) * 1e3
        if args.no_stats_pass:
            st, rays = st2, st2.rays
        print(json.dumps({
            "config": name, "size": [params.width, params.height], "spp": params.num_samples, "max_depth": params.max_depth, "batches": int(st2.batches),
            "triangles": int(flat.arrays["triangles"].shape[0]), "gates": None,
            "scene_build_s": round(t1 - t0, 2), "commit_s": round(t2 - t1, 2), "frame_ms": round(ms, 2),
            "rays": int(rays), "mrays_per_s": round(rays / ms / 1e3, 1),
            "kernel_ms": {n: round(float(st2.ms_kernel[i]), 2) for i, n in enumerate(capi.RT_K_NAMES)},
            "nodes_per_bvh_ray": [round(st.nodes[k] / max(1, st.bvh_rays[k]), 1) for k in (0, 2)],
            "gates_per_ray": {"closest": round(st.gates[0] / max(1, st.rays_primary + st.rays_bounce), 2), "shadow": round(st.gates[1] / max(1, st.rays_shadow), 2),
                              "ao": round(st.gates[2] / max(1, st.rays_ao), 2)},
            "ao_rays": int(st.rays_ao), "ao_bvh_share": round(st.bvh_rays[2] / max(1, st.rays_ao), 3),
        }), flush=True)
    r.close()


if __name__ == "__main__":
    main()
