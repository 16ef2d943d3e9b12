#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz by running the REAL reference (oracle/_ref/libref_oracle.so,
compiled from /root/reference by oracle/Makefile) on the small seeded scenes of tests/util.py.

Each fixture holds inputs-by-name (the scene builders are deterministic code in
raytracer_amd/scenes.py, the parameters are stored) and the reference's outputs: float32 RGB,
primary-hit primitive id / t / normal / uv of sample 0, and the closest-hit query count.
rng_mode 1 = the reference's rand() replaced by the counter hash of include/rt_rng.h and pixel
sample points drawn from it (see oracle/ref_harness.cpp), which makes jitter + AO reproducible.

Also writes known-answer vectors of single reference functions (kat.npz) and the reference's own
decode of models/teapot/default.png (teapot_default_rgba.npz).
Usage:  python tests/golden/make_golden.py        (needs /root/reference; run in the build container)
        python tests/golden/make_golden.py converged    only the stochastic fixture below (about 4 minutes)
        python tests/golden/make_golden.py converged_random   the same for the reference's Random sampler
        python tests/golden/make_golden.py bmp          only bmp_savetofile.npz (the reference's saveToFile bytes)
        python tests/golden/make_golden.py only NAME... only these scene fixtures
        python tests/golden/make_golden.py cpp_host     only cpp_default_160x120.npz (fixture of the C++ drop-in test)

converged_default_libc_rand.npz (SURVEY.md 8c "stochastic parity protocol"): the mean of 32 independent
runs of the unmodified reference (rng_mode 0: libc rand() and its own Sampler with std::random_device,
8 pool threads) on the default scene at 72x72, Jitter(16,83), 4x4 ambient occlusion = 512 samples per
pixel, plus the leave-one-out PSNR of the single runs against the mean of the others.  The file changes
from run to run by design; the tests that read it only use statistics with wide margins.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import util  # noqa: E402
from oracle.bindings import RefOracle, f3  # noqa: E402
from raytracer_amd import capi  # noqa: E402


def converged(ref, runs=32):
    from raytracer_amd import scenes, scene as S
    w = scenes.default_scene(72, 72, sampler=S.Jitter(16, 83), ao_samples=4)
    flat, p = w.flatten(), w.params()
    imgs = np.stack([ref.render(flat, w.camera, p, rng_mode=0, nthreads=8, ids=False).rgb.astype(np.float64)
                     for _ in range(runs)])
    total = imgs.sum(axis=0)
    loo = [util.psnr(imgs[k], (total - imgs[k]) / (runs - 1)) for k in range(runs)]
    np.savez_compressed(os.path.join(HERE, "converged_default_libc_rand.npz"), rgb=(total / runs).astype(np.float32),
                        runs=np.int32(runs), single_run_psnr=np.array(loo, np.float64),
                        params=np.frombuffer(bytes(p), dtype=np.uint8))
    print("converged: single-run PSNR against the mean of the others %.2f dB (min %.2f)" % (np.mean(loo), np.min(loo)))


def converged_random(ref, runs=32):
    """The same protocol for the reference's own Random sampler (main.cpp:1019-1035: std::mt19937 seeded by
    std::random_device, not reproducible): mean of `runs` runs of the unmodified reference, Random(16,83), 4x4 AO."""
    from raytracer_amd import scenes, scene as S
    w = scenes.default_scene(72, 72, sampler=S.Random(16, 83), ao_samples=4)
    flat, p = w.flatten(), w.params()
    imgs = np.stack([ref.render(flat, w.camera, p, rng_mode=0, nthreads=8, ids=False).rgb.astype(np.float64)
                     for _ in range(runs)])
    total = imgs.sum(axis=0)
    loo = [util.psnr(imgs[k], (total - imgs[k]) / (runs - 1)) for k in range(runs)]
    np.savez_compressed(os.path.join(HERE, "converged_default_random.npz"), rgb=(total / runs).astype(np.float32),
                        runs=np.int32(runs), single_run_psnr=np.array(loo, np.float64),
                        params=np.frombuffer(bytes(p), dtype=np.uint8))
    print("converged (Random sampler): single-run PSNR against the mean of the others %.2f dB (min %.2f)" % (np.mean(loo), np.min(loo)))


def regular4_scene():
    """The reference's main() scene with Regular(4, 1) — the one multi-sample mode in which the reference's OWN Sampler
    (Sampler::sampleUnitSquare over Regular::generateSamples, main.cpp:539-548, 971-991) is deterministic up to the
    order of the four samples — flat ambient (no AO, so no rand() in shading)."""
    from raytracer_amd import scene as S, scenes
    w = scenes.default_scene(160, 120, sampler=S.Regular(4, 1), ao_samples=0)
    return w, w.params(seed=1)


def regular4(ref):
    """default_regular4.npz: rendered by the real reference through its own sampler (harness rng_mode 2 routes every
    sample through sampler->sampleUnitSquare(); one thread, as SURVEY 8c prescribes for the shared Sampler)."""
    w, p = regular4_scene()
    r = ref.render(w.flatten(), w.camera, p, rng_mode=2, nthreads=1, ids=False)
    np.savez_compressed(os.path.join(HERE, "default_regular4.npz"), rgb=r.rgb, rays=np.uint64(r.rays),
                        params=np.frombuffer(bytes(p), dtype=np.uint8))
    print("default_regular4", r.rgb.shape, "rays", r.rays)


def bmp_inputs():
    """Float images for the saveToFile fixtures: widths 200 and 101 (row padding 0 and 1 bytes), a 4x4 image
    (fewer pixels than header bytes) and a 3x5 one (3 bytes of padding).  Values in [0,1] with the exact
    multiples of 1/255 and their neighbours mixed in (truncation edges)."""
    rng = np.random.default_rng(2024)
    out = {}
    for name, (w, h) in (("200x24", (200, 24)), ("101x47", (101, 47)), ("4x4", (4, 4)), ("3x5", (3, 5))):
        img = rng.random((h, w, 3), dtype=np.float32)
        k = rng.integers(0, 256, (h, w, 3)).astype(np.float32) / np.float32(255)
        pick = rng.integers(0, 4, (h, w, 3))
        img = np.where(pick == 0, k, img)
        img = np.where(pick == 1, np.nextafter(k, np.float32(0), dtype=np.float32), img)
        img = np.where(pick == 2, np.nextafter(k, np.float32(2), dtype=np.float32), img).astype(np.float32)
        img = np.clip(img, 0.0, 1.0)
        img.reshape(-1)[:3] = (0.0, 1.0, 0.5)
        out[name] = np.ascontiguousarray(img)
    return out


def bmp(ref):
    """saveToFile fixtures: the float inputs and the bytes the REFERENCE's saveToFile (main.cpp:1837-1857,
    stbi_write_bmp) wrote for them."""
    fx = {}
    for name, img in bmp_inputs().items():
        fx["rgb_" + name] = img
        fx["bmp_" + name] = np.frombuffer(ref.save_bmp_bytes(img), np.uint8)
        print("bmp", name, len(fx["bmp_" + name]), "bytes")
    np.savez_compressed(os.path.join(HERE, "bmp_savetofile.npz"), **fx)


def cpp_host(ref):
    """What examples/default_scene.cpp (the reference's main() scene script on the C++ drop-in host) must write
    for `default_scene <models> out.bmp 160 120`: the REAL reference's render of the same scene (rng_mode 1,
    seed 1 = World::seed's default) as 8-bit RGB, and the BMP its saveToFile writes for it."""
    from raytracer_amd import scenes, scene as S
    w = scenes.default_scene(160, 120, sampler=S.Jitter(4, 83), ao_samples=4)
    p = w.params(seed=1)
    r = ref.render(w.flatten(), w.camera, p, rng_mode=1, nthreads=8, ids=False)
    np.savez_compressed(os.path.join(HERE, "cpp_default_160x120.npz"), rgb8=util.quantize(r.rgb),
                        bmp=np.frombuffer(ref.save_bmp_bytes(r.rgb), np.uint8),
                        params=np.frombuffer(bytes(p), dtype=np.uint8))
    print("cpp_host fixture", r.rgb.shape, "rays", r.rays)


def bbox(ref):
    """getBoundingBoxPoints() of the shapes of the reference's main() scene (7 spheres, 2 free triangles, 3
    MultiMeshes of the stand-in mesh), as the REAL reference computes them: what the C++ drop-in's
    getBoundingBoxPoints() must return (examples/default_scene <models> - - - bbox)."""
    from raytracer_amd import scenes, scene as S, capi
    w = scenes.default_scene(160, 120, sampler=S.Jitter(4, 83), ao_samples=4)
    flat = w.flatten()
    p = w.params(seed=1)
    d = flat.desc
    out = {
        "spheres": np.stack([ref.bbox_points(flat, p, capi.RT_SHAPE_SPHERE, i) for i in range(d.num_spheres)]),
        "multimeshes": np.stack([ref.bbox_points(flat, p, capi.RT_SHAPE_MULTIMESH, i) for i in range(d.num_multimeshes)]),
    }
    in_mesh = np.zeros(d.num_triangles, bool)
    for m in flat.arrays["meshes"]:
        in_mesh[int(m["first_triangle"]): int(m["first_triangle"]) + int(m["num_triangles"])] = True
    tri_ids = [int(i) for i in np.flatnonzero(~in_mesh)]
    out["triangles"] = np.stack([ref.bbox_points(flat, p, capi.RT_SHAPE_TRIANGLE, i) for i in tri_ids])
    np.savez_compressed(os.path.join(HERE, "bbox_points.npz"), **out)
    print("bbox fixture", {k: v.shape for k, v in out.items()})


def main():
    ref = RefOracle()
    if sys.argv[1:] == ["bbox"]:
        bbox(ref)
        return
    if sys.argv[1:] == ["cpp_host"]:
        cpp_host(ref)
        return
    if sys.argv[1:] == ["converged"]:
        converged(ref)
        return
    if sys.argv[1:] == ["converged_random"]:
        converged_random(ref)
        return
    if sys.argv[1:] == ["bmp"]:
        bmp(ref)
        return
    if sys.argv[1:] == ["regular4"]:
        regular4(ref)
        return
    only = set(sys.argv[2:]) if sys.argv[1:2] == ["only"] else None   # `only name ...`: just these scene fixtures
    for name, world, params in util.small_scenes():
        if only is not None and name not in only:
            continue
        flat = world.flatten()
        r = ref.render(flat, world.camera, params, rng_mode=1, nthreads=8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rgb=r.rgb, primid=r.primid, t=r.t, normal=r.normal,
                            uv=r.uv, rays=np.uint64(r.rays),
                            params=np.frombuffer(bytes(params), dtype=np.uint8))
        print(name, r.rgb.shape, "rays", r.rays)
    if only is not None:
        return
    bmp(ref)
    cpp_host(ref)
    regular4(ref)

    # known-answer vectors of single functions
    rng = np.random.default_rng(1234)
    L = ref.lib
    kat = {}
    ab = rng.uniform(0, 1, (64, 2)).astype(np.float32)
    hemi = np.zeros((64, 3), np.float32)
    for i in range(64):
        out = (C.c_float * 3)()
        L.ref_sample_hemisphere(float(ab[i, 0]), float(ab[i, 1]), out)
        hemi[i] = list(out)
    kat["hemi_in"], kat["hemi_out"] = ab, hemi
    ups = rng.normal(size=(64, 3)).astype(np.float32)
    vs = rng.normal(size=(64, 3)).astype(np.float32)
    rot = np.zeros((64, 3), np.float32)
    for i in range(64):
        out = (C.c_float * 3)()
        L.ref_rotate_to_space(f3(ups[i]), f3(vs[i]), out)
        rot[i] = list(out)
    kat["rot_up"], kat["rot_v"], kat["rot_out"] = ups, vs, rot
    a = rng.uniform(-2, 2, 256).astype(np.float32)
    b = (a * (1 + rng.choice([0, 1e-8, 6e-8, 1.2e-7, -6e-8, 1e-3], 256))).astype(np.float32)
    kat["geq_a"], kat["geq_b"] = a, b
    kat["geq_out"] = np.array([L.ref_geq(float(x), float(y)) for x, y in zip(a, b)], np.int32)
    cam = capi.rt_camera()
    cam.eye[:] = [-1000, 300, 250]; cam.look_at[:] = [0, 0, -100]; cam.up[:] = [0, 1, 0]; cam.frustum_dist = 700.0
    xy = rng.uniform(-500, 500, (64, 2)).astype(np.float32)
    cd = np.zeros((64, 3), np.float32)
    for i in range(64):
        o3, d3 = (C.c_float * 3)(), (C.c_float * 3)()
        L.ref_camera_ray(C.byref(cam), float(xy[i, 0]), float(xy[i, 1]), o3, d3)
        cd[i] = list(d3)
    kat["cam_xy"], kat["cam_d"] = xy, cd
    # box gate: rays around / inside / grazing a box
    pts = np.array([[-10, -20, -30], [40, 50, 60], [0, 0, 0], [5, 5, 5], [-10, 50, 0], [40, -20, 60]], np.float32)
    ro = rng.uniform(-100, 100, (512, 3)).astype(np.float32)
    rd = rng.normal(size=(512, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[::7, 0] = 0.0   # axis-parallel components
    ro[::5] = rng.uniform(-9, 39, (len(ro[::5]), 3)).astype(np.float32)   # origins inside
    fp = C.POINTER(C.c_float)
    kat["box_pts"], kat["box_o"], kat["box_d"] = pts, ro, rd
    kat["box_out"] = np.array([L.ref_box_intersects(pts.ctypes.data_as(fp), 6, f3(ro[i]), f3(rd[i])) for i in range(512)], np.int32)
    # Phong term
    n = rng.normal(size=(64, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True)
    wo = rng.normal(size=(64, 3)).astype(np.float32); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rng.normal(size=(64, 3)).astype(np.float32); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    kat["spec_n"], kat["spec_wo"], kat["spec_wi"] = n, wo, wi
    kat["spec_out"] = np.array([L.ref_specular_fn(0.3, 16, f3(n[i]), f3(wo[i]), f3(wi[i])) for i in range(64)], np.float32)
    # refraction
    out = np.zeros((64, 4), np.float32)
    iors = rng.choice([1.0, 1.7], 64).astype(np.float32)
    for i in range(64):
        o3 = (C.c_float * 3)(); e = C.c_float()
        L.ref_refraction(1.7, float(iors[i]), f3(wo[i]), f3(n[i]), o3, C.byref(e))
        out[i] = list(o3) + [e.value]
    kat["refr_ior"], kat["refr_out"] = iors, out
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **kat)

    tex = ref.load_texture("/root/reference/Raytracer/models/teapot/default.png")
    np.savez_compressed(os.path.join(HERE, "teapot_default_rgba.npz"), rgba=tex)
    print("texture", tex.shape, np.unique(tex[..., 0]))


if __name__ == "__main__":
    main()
