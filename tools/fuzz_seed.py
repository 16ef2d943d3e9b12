#!/usr/bin/env python3
"""tools/fuzz_seed.py SEED [SEED...] — one random scene (tests/util.py::random_scene) under several RT_DEBUG settings
against the oracle: prints the largest 8-bit difference per setting (diagnosis of tools/fuzz_gpu.py failures)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import util  # noqa: E402
from oracle.bindings import PortOracle  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402

port = PortOracle()
gpu = Renderer(0)
for seed in map(int, sys.argv[1:]):
    w, params = util.random_scene(seed)
    flat = w.flatten()
    want = port.render(flat, w.camera, params, rng_mode=1, nthreads=8)
    line = [f"seed {seed} (lights {len(w.lights)}, ao {w.ambient.occlusion_samples if hasattr(w.ambient, 'occlusion_samples') else '?'}, gates ?)"]
    for dbg in ("0", "4", "8", "12", "16", "24", "28"):
        os.environ["RT_DEBUG"] = dbg
        gpu.upload(flat)
        rgb, prim, t, st = gpu.render(w.camera, params, ids=True)
        d = util.lsb_diff(rgb, want.rgb)
        line.append(f"dbg{dbg}: {int(d.max())} ({int((d > 1).sum())} px)")
    print("  ".join(line))
