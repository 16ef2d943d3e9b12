"""The C++ host mirror (raytracer_amd/host/Raytracer.hpp): the reference-style scene script
examples/default_scene.cpp — OBJ/MTL/texture loaded from a model directory, BVHAccel, ThreadPool with
40x40 raytrace() tile jobs, saveToFile — must produce the same BMP as the Python host for the same scene."""
import os
import subprocess

import numpy as np
import pytest

import conftest
import util
from raytracer_amd import scenes, scene as S

EXE = os.path.join(conftest.ROOT, "examples", "default_scene")


def write_model_dir(root):
    """models/teapot/{teapot.obj, default.mtl, default.ppm}: the procedural stand-in for the absent teapot.obj."""
    d = os.path.join(root, "teapot")
    os.makedirs(d, exist_ok=True)
    tex = scenes.checker_texture()
    obj = scenes.torus_objmesh(None)
    sh = obj.shapes[0]
    with open(os.path.join(d, "teapot.obj"), "w") as f:
        f.write("mtllib default.mtl\no torus\nusemtl default\n")
        for p in sh.positions:
            f.write("v %.9g %.9g %.9g\n" % tuple(float(x) for x in p))
        for t in sh.texcoords:
            f.write("vt %.9g %.9g\n" % tuple(float(x) for x in t))
        for a, b, c in sh.indices + 1:
            f.write(f"f {a}/{a} {b}/{b} {c}/{c}\n")
    with open(os.path.join(d, "default.mtl"), "w") as f:
        f.write("newmtl default\nKd 1 1 1\nmap_Kd default.ppm\n")
    with open(os.path.join(d, "default.ppm"), "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (tex.width, tex.height))
        f.write(tex.rgba[:, :, :3].tobytes())


def read_bmp(path):
    b = open(path, "rb").read()
    w = int.from_bytes(b[18:22], "little"); h = int.from_bytes(b[22:26], "little")
    row = (w * 3 + 3) // 4 * 4
    img = np.frombuffer(b[54:], np.uint8).reshape(h, row)[:, : w * 3].reshape(h, w, 3)
    return img[::-1, :, ::-1]


def test_example_is_built():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.dirname(EXE)], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_scene_script_matches_python_host(gpu, tmp_path):
    write_model_dir(str(tmp_path))
    out = str(tmp_path / "raytrace.bmp")
    r = subprocess.run([EXE, str(tmp_path), out, "200", "200"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Running time is" in r.stdout
    assert "tile jobs: 1600, GPU launches: 1" in r.stdout, r.stdout   # the 40x40 jobs coalesce at waitDone()
    got = read_bmp(out)
    out2 = str(tmp_path / "whole.bmp")
    r = subprocess.run([EXE, str(tmp_path), out2, "200", "200", "whole"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_bmp(out2), got), "tile jobs and a single render() call disagree"
    world = scenes.default_scene(200, 200, sampler=S.Jitter(4, 83), ao_samples=4)
    gpu.upload(world)
    rgb, _, _, _ = gpu.render(world.camera, world.params(seed=1))
    want = util.quantize(rgb)
    bad = np.argwhere((got != want).any(axis=2))
    assert len(bad) == 0, f"{len(bad)} pixels differ, first {bad[:8].tolist()}, got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}"
