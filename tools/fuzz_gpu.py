#!/usr/bin/env python3
"""tools/fuzz_gpu.py — fuzz campaign on the GPU box: random scenes (tests/util.py::random_scene) rendered by
the HIP path, with both triangle-BVH builders, against the C oracle; primitive ids and ray parameters must be
bit-equal, 8-bit RGB within 1 LSB.

    python tools/fuzz_gpu.py [--first 100] [--count 200] [--seconds 400] [--scale 4]
prints one line per failing seed and a summary; exit status 1 if any seed failed."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import util  # noqa: E402
from oracle.bindings import PortOracle  # noqa: E402
from raytracer_amd import capi  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--seconds", type=float, default=400.0)
    ap.add_argument("--scale", type=int, default=1, help="multiplies the 40x30 image (and the frustum distance)")
    args = ap.parse_args()
    port = PortOracle()
    gpu = Renderer(0)
    t0 = time.time()
    bad = done = 0
    for seed in range(args.first, args.first + args.count):
        if time.time() - t0 > args.seconds:
            break
        w, params = util.random_scene(seed)
        if args.scale > 1:
            w.width, w.height = w.width * args.scale, w.height * args.scale
            w.camera.frustum_dist *= args.scale
            params = w.params(seed=seed)
        flat = w.flatten()
        want = port.render(flat, w.camera, params, rng_mode=1, nthreads=16)
        for builder in (capi.RT_BUILDER_HOST_SAH, capi.RT_BUILDER_GPU_LBVH):
            gpu.upload(flat, builder=builder)
            rgb, prim, t, st = gpu.render(w.camera, params, ids=True)
            ok = np.array_equal(prim, want.primid) and util.bits_equal(t, want.t) and util.lsb_diff(rgb, want.rgb).max() <= 1
            if not ok:
                bad += 1
                print(f"seed {seed} builder {builder}: ids differ on {(prim != want.primid).sum()} px, "
                      f"max LSB diff {util.lsb_diff(rgb, want.rgb).max()}", flush=True)
        done += 1
        if done % 20 == 0:
            print(f"... {done} seeds, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz: {done} seeds x 2 builders, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
