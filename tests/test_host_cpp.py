"""The C++ host mirror (raytracer_amd/host/Raytracer.hpp): the reference-style scene script
examples/default_scene.cpp — OBJ/MTL/texture loaded from a model directory, BVHAccel, ThreadPool with
40x40 raytrace() tile jobs, saveToFile — must produce the BMP the real reference renders for the same scene
(golden fixture)."""
import os
import subprocess

import numpy as np
import pytest

import conftest
import util
from raytracer_amd import scenes, scene as S

EXE = os.path.join(conftest.ROOT, "examples", "default_scene")


def write_model_dir(root, png_backslash=False):
    """models/teapot/{teapot.obj, default.mtl, default.ppm}: the procedural stand-in for the absent teapot.obj.
    png_backslash: the texture as a PNG in a sub-directory, named in the MTL with a backslash path the way the
    reference's models/sponza/sponza.mtl names its textures (map_Kd textures\\sponza_thorn_diff.png)."""
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
    if png_backslash:
        os.makedirs(os.path.join(d, "textures"), exist_ok=True)
        with open(os.path.join(d, "default.mtl"), "w") as f:
            f.write("newmtl default\nKd 1 1 1\nmap_Kd textures\\default.png\n")
        rgb = np.ascontiguousarray(tex.rgba[:, :, :3])
        with open(os.path.join(d, "textures", "default.png"), "wb") as f:
            f.write(_png(tex.width, tex.height, 2, 8, [bytes(r.tobytes()) for r in rgb], filt=lambda y: y % 5, level=6, split=2))
        return
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


def test_bounding_box_points_equal_the_reference(tmp_path):
    """Sphere / Triangle / MultiMesh::getBoundingBoxPoints() of the drop-in header (the lists a scene script hands to
    BVHAccel::addShape) against tests/golden/bbox_points.npz, the REAL reference's values for the shapes of its
    main() scene (make_golden.py bbox): bit-identical, in the reference's order x-, x+, y-, y+, z-, z+.  No GPU."""
    test_example_is_built()
    g = np.load(os.path.join(util.GOLDEN, "bbox_points.npz"))
    write_model_dir(str(tmp_path))
    r = subprocess.run([EXE, str(tmp_path), "-", "160", "120", "bbox"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = {"sphere": [], "triangle": [], "multimesh": []}
    for line in r.stdout.splitlines():
        name, *vals = line.split()
        got[name].append(np.array([float.fromhex(v) for v in vals], np.float32).reshape(6, 3))
    for name, key in (("sphere", "spheres"), ("triangle", "triangles"), ("multimesh", "multimeshes")):
        mine = np.stack(got[name])
        assert mine.shape == g[key].shape
        assert np.array_equal(mine.view(np.uint32), g[key].view(np.uint32)), name


@pytest.mark.gpu
def test_cpp_scene_script_matches_the_reference_render(tmp_path):
    """SURVEY 8f N3 + N4 end to end: examples/default_scene.cpp = the reference's main() scene script compiled
    against the drop-in Raytracer.hpp (OBJ/MTL/texture loaded from a model directory, BVHAccel, ThreadPool with
    40x40 raytrace() tile jobs, saveToFile).  Its BMP is compared with tests/golden/cpp_default_160x120.npz: the
    REAL reference's render of the same scene (made by make_golden.py cpp_host) — 8-bit RGB within 1 LSB, >= 95 %
    of the bytes identical, and the file's header / row layout identical to the reference-written BMP."""
    g = np.load(os.path.join(util.GOLDEN, "cpp_default_160x120.npz"))
    write_model_dir(str(tmp_path))
    out = str(tmp_path / "raytrace.bmp")
    r = subprocess.run([EXE, str(tmp_path), out, "160", "120"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Running time is" in r.stdout
    assert "tile jobs: 1600, GPU launches: 1" in r.stdout, r.stdout   # the 40x40 jobs coalesce at waitDone()
    got = read_bmp(out)
    want = g["rgb8"]
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1, f"{(d > 1).sum()} bytes differ by more than 1 LSB from the reference render"
    assert (d == 0).mean() >= 0.95
    ref_file, my_file = g["bmp"].tobytes(), open(out, "rb").read()
    assert len(my_file) == len(ref_file) and my_file[:54] == ref_file[:54]
    # the same script on "three GPUs": three contexts (all on device 0 here) through rt_render_multi
    out3 = str(tmp_path / "multi.bmp")
    env = dict(os.environ, RT_GPU_DEVICES="0,0,0")
    r = subprocess.run([EXE, str(tmp_path), out3, "160", "120"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    assert open(out3, "rb").read() == my_file, "the multi-context render differs from the single-context one"
    # N3 once more through the PNG decoder and a backslash map_Kd path (the convention of the reference's sponza.mtl):
    # the texture as textures/default.png, named "textures\\default.png" in the MTL — the same BMP, byte for byte
    pdir = tmp_path / "png_models"
    write_model_dir(str(pdir), png_backslash=True)
    outp = str(tmp_path / "png.bmp")
    r = subprocess.run([EXE, str(pdir), outp, "160", "120"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert open(outp, "rb").read() == my_file, "PNG texture behind a backslash path renders differently from the PPM one"
    out2 = str(tmp_path / "whole.bmp")
    r = subprocess.run([EXE, str(tmp_path), out2, "160", "120", "whole"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_bmp(out2), got), "tile jobs and a single render() call disagree"


# ---- PNG decoder of the C++ host: no GPU needed ------------------------------------------
DUMP = os.path.join(conftest.ROOT, "examples", "texture_dump")


def _png(w, h, ctype, depth, rows, plte=None, trns=None, filt=None, level=6, split=1):
    """Minimal PNG writer: rows = h byte strings of packed samples; filt(y) -> filter type per row."""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, channels * depth // 8)
    raw = b""
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filt(y) if filt else 0
        out = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: pred = 0
            elif ft == 1: pred = a
            elif ft == 2: pred = b
            elif ft == 3: pred = (a + b) >> 1
            else:
                pp = a + b - c
                pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[i] = (v - pred) & 255
        raw += bytes([ft]) + bytes(out)
        prev = row
    z = zlib.compress(raw, level)
    parts = [z[i * len(z) // split:(i + 1) * len(z) // split] for i in range(split)]
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if plte is not None: png += chunk(b"PLTE", plte)
    if trns is not None: png += chunk(b"tRNS", trns)
    for part in parts: png += chunk(b"IDAT", part)
    return png + chunk(b"IEND", b"")


def _dump(path, tmp_path):
    out = str(tmp_path / "dump.rgba")
    r = subprocess.run([DUMP, path, out], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    b = open(out, "rb").read()
    nl = b.index(b"\n")
    w, h = (int(x) for x in b[:nl].split())
    return np.frombuffer(b[nl + 1:], np.uint8).reshape(h, w, 4)


def test_png_decoder_matches_stb_conventions(tmp_path):
    """Every colour type / bit depth / filter, stored + fixed + dynamic deflate blocks, split IDATs;
    expected RGBA8 follows stbi_load(.., STBI_rgb_alpha): grey -> (g,g,g,255), 16 bit -> high byte,
    sub-byte grey scaled to 0..255, palette + tRNS."""
    if not os.path.exists(DUMP):
        subprocess.run(["make", "-C", os.path.dirname(DUMP)], check=True, stdout=subprocess.DEVNULL)
    rng = np.random.default_rng(7)
    w, h = 37, 23
    cases = []
    # smooth + noisy content so that dynamic Huffman blocks with long matches and literals both occur
    base = (np.add.outer(np.arange(h) * 5, np.arange(w) * 3)[:, :, None] + np.arange(4)[None, None, :] * 40) % 256
    noisy = rng.integers(0, 256, (h, w, 4))
    for name, img in (("smooth", base), ("noisy", noisy)):
        img = img.astype(np.uint8)
        for level in (0, 1, 9):
            filt = (lambda y: y % 5)
            rgba = img
            cases.append((f"rgba8_{name}_{level}", _png(w, h, 6, 8, [bytes(r.tobytes()) for r in rgba], filt=filt, level=level, split=3), rgba))
            rgb = img[:, :, :3]
            exp = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], 2)
            cases.append((f"rgb8_{name}_{level}", _png(w, h, 2, 8, [bytes(r.tobytes()) for r in rgb], filt=filt, level=level), exp))
    g = base[:, :, 0].astype(np.uint8)
    cases.append(("grey8", _png(w, h, 0, 8, [bytes(r.tobytes()) for r in g], filt=lambda y: 4), np.stack([g, g, g, np.full_like(g, 255)], 2)))
    ga = base[:, :, :2].astype(np.uint8)
    cases.append(("greyalpha8", _png(w, h, 4, 8, [bytes(r.tobytes()) for r in ga], filt=lambda y: 3), np.stack([ga[:, :, 0]] * 3 + [ga[:, :, 1]], 2)))
    hi = noisy[:, :, :3].astype(np.uint8); lo = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    rows16 = [bytes(np.stack([hi[y], lo[y]], 2).tobytes()) for y in range(h)]
    cases.append(("rgb16", _png(w, h, 2, 16, rows16, filt=lambda y: 1), np.concatenate([hi, np.full((h, w, 1), 255, np.uint8)], 2)))
    idx = rng.integers(0, 16, (h, w)).astype(np.uint8)
    plte = rng.integers(0, 256, (16, 3)).astype(np.uint8); trns = rng.integers(0, 256, 9).astype(np.uint8)
    exp = np.concatenate([plte[idx], np.where(idx < 9, np.concatenate([trns, np.full(7, 255, np.uint8)])[idx], 255)[:, :, None].astype(np.uint8)], 2)
    cases.append(("pal8", _png(w, h, 3, 8, [bytes(r.tobytes()) for r in idx], plte=plte.tobytes(), trns=trns.tobytes()), exp))
    packed4 = [bytes(np.packbits(np.unpackbits(r[:, None], axis=1)[:, 4:].reshape(-1)).tobytes()) for r in idx]
    cases.append(("pal4", _png(w, h, 3, 4, packed4, plte=plte.tobytes(), trns=trns.tobytes()), exp))
    g2 = rng.integers(0, 4, (h, w)).astype(np.uint8)
    packed2 = [bytes(np.packbits(np.unpackbits(r[:, None], axis=1)[:, 6:].reshape(-1)).tobytes()) for r in g2]
    cases.append(("grey2", _png(w, h, 0, 2, packed2), np.stack([g2 * 85] * 3 + [np.full_like(g2, 255)], 2)))
    for name, data, want in cases:
        path = str(tmp_path / (name + ".png"))
        open(path, "wb").write(data)
        got = _dump(path, tmp_path)
        assert got.shape == want.shape and np.array_equal(got, want), name
    ref_png = "/root/reference/Raytracer/models/teapot/default.png"
    if os.path.exists(ref_png):   # the reference's own texture decodes to the checker the scenes generate
        tex = scenes.checker_texture()
        assert np.array_equal(_dump(ref_png, tmp_path), tex.rgba)


# ---- the drop-in claim: the reference's own main() compiles against the drop-in header -----------------
REF_MAIN = "/root/reference/Raytracer/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference sources are only present in the build container")
def test_reference_main_compiles_unchanged_against_the_drop_in_header(tmp_path):
    """north_star: 'the C++ host code keeps the existing Raytracer.hpp scene/camera/render entry points so it is a
    drop-in'.  The text of the reference's main() (main.cpp:1667-1835: World, Jitter, loadObjMesh, Textured / Specular
    / Mirror / Refraction setters, Triangle / Sphere / Plane / MultiMesh, BVHAccel::addShape(getBoundingBoxPoints())
    / generateBVH, lights, Camera, ThreadPool::startJob(raytrace(...)) / waitDone, saveToFile, atlas Timer) is read
    from the reference at test time into tmp_path (never committed) and compiled, unchanged, against
    raytracer_amd/host/Raytracer.hpp."""
    lines = open(REF_MAIN, encoding="utf-8", errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.strip() == "int main()")
    assert 1600 < start < 1700, "main() moved: re-check the cited range"
    depth, end = 0, None
    for i in range(start, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        if depth == 0 and i > start + 1:
            end = i
            break
    assert end is not None and end - start > 150
    src = tmp_path / "reference_main.cpp"
    src.write_text('#include "Raytracer.hpp"\n' + "\n".join(lines[start:end + 1]) + "\n")
    host = os.path.join(conftest.ROOT, "raytracer_amd", "host")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", host, "-I", os.path.join(conftest.ROOT, "include"), str(src)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
