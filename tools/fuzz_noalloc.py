#!/usr/bin/env python3
"""tools/fuzz_noalloc.py [--first N] [--count M] — random scenes (tests/util.py::random_scene, images scaled x3 so a
frame has several tiles) through rt_render_device_noalloc against rt_render_device: frames must be bit-identical."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import util  # noqa: E402
from raytracer_amd import capi  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=0)
ap.add_argument("--count", type=int, default=200)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
gpu = Renderer(0)
bad = 0
for seed in range(args.first, args.first + args.count):
    w, params = util.random_scene(seed)
    p = capi.rt_params.from_buffer_copy(params)
    p.width *= 3; p.height *= 3; p.x1 *= 3; p.y1 *= 3
    w.camera.frustum_dist *= 3
    gpu.upload(w.flatten())
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device=dev)
    b = torch.full((p.height, p.width, 3), -1.0, dtype=torch.float32, device=dev)
    gpu.render_device(w.camera, p, a.data_ptr(), None, None, None, st)
    gpu.render_device_noalloc(w.camera, p, b.data_ptr(), None, None, st)
    torch.cuda.synchronize()
    if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
        bad += 1
        print("seed", seed, "differs on", int((a.view(torch.int32) != b.view(torch.int32)).sum()), "values")
print(f"noalloc fuzz: {args.count} seeds, {bad} failures")
sys.exit(1 if bad else 0)
