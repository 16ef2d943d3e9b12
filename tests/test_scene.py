"""Host logic: the Python mirror of the reference's scene classes and the tile arithmetic."""
import numpy as np

from raytracer_amd import capi, dist, scenes, scene as S


def test_default_scene_flattening_counts():
    w = scenes.default_scene(100, 100)
    f = w.flatten()
    d = f.desc
    assert (d.num_planes, d.num_spheres, d.num_multimeshes, d.num_meshes, d.num_bvhs) == (6, 7, 3, 3, 1)
    assert d.num_triangles == 2 + 3 * 4096 and d.num_bvh_leaves == 12 and d.num_scene == 7
    assert d.num_lights == 1 and d.num_textures == 1
    kinds = [int(k) for k in f.arrays["bvh_leaves"]["kind"]]
    assert kinds == [capi.RT_SHAPE_TRIANGLE] * 2 + [capi.RT_SHAPE_MULTIMESH] * 3 + [capi.RT_SHAPE_SPHERE] * 7   # main.cpp:1763-1779
    assert [int(k) for k in f.arrays["scene"]["kind"]] == [capi.RT_SHAPE_PLANE] * 6 + [capi.RT_SHAPE_BVH]


def test_mesh_offset_is_added_in_float32():
    tex = S.Textured(scenes.checker_texture())
    obj = scenes.torus_objmesh(tex, 8, 4)
    off = np.array([40, 0, -200], np.float32)
    m = S.Mesh(obj.shapes[0], off)
    want = (obj.shapes[0].positions + off[None]).astype(np.float32)[obj.shapes[0].indices]
    assert np.array_equal(m.tri_v, want)


def test_meshes_face_outwards():
    """Triangles are single-sided in the reference: generated meshes must face outwards."""
    for obj, c in ((scenes.torus_objmesh(None, 16, 8, centre=(0, 0, 0)), None),
                   (scenes.bumpy_sphere_objmesh(None, 24, 12), np.zeros(3))):
        sh = obj.shapes[0]
        v = sh.positions[sh.indices].astype(np.float64)
        n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        cen = v.mean(axis=1)
        if c is None:   # torus: outward = away from the ring centre line
            ring = cen.copy(); ring[:, 1] = 0
            ring = 60.0 * ring / np.linalg.norm(ring, axis=1, keepdims=True)
            out = cen - ring
        else:
            out = cen - c
        assert (np.einsum("ij,ij->i", n, out) > 0).all()
        assert sh.texcoords.min() >= 0 and sh.texcoords.max() < 1


def test_tile_partition_covers_every_pixel_once():
    p = capi.default_params(200, 136)
    p.x0, p.y0, p.x1, p.y1 = 3, 5, 197, 130
    for world in (1, 2, 3, 8):
        cover = np.zeros((136, 200), np.int32)
        n = 0
        for r in range(world):
            k = dist.owned_tiles(p, r, world)
            n += k
            for i in range(k):
                x0, y0, x1, y1 = dist.tile_rect(p, r, world, i)
                cover[y0:y1, x0:x1] += 1
        tx, ty = dist.tile_grid(p)
        assert n == tx * ty
        assert (cover[5:130, 3:197] == 1).all() and cover.sum() == (130 - 5) * (197 - 3)


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(0)
    frame = rng.uniform(size=(100, 150, 3)).astype(np.float32)
    p = capi.default_params(150, 100)
    out = np.zeros_like(frame)
    for r in range(4):
        dist.unpack_slab_numpy(dist.pack_slab_numpy(frame, p, r, 4), p, r, 4, out)
    assert np.array_equal(out, frame)


def test_bmp_writer(tmp_path):
    from raytracer_amd.render import save_bmp
    img = np.zeros((3, 5, 3), np.float32)
    img[0, 0] = (1.0, 0.5, 0.25)
    f = tmp_path / "x.bmp"
    save_bmp(str(f), img)
    b = f.read_bytes()
    assert b[:2] == b"BM" and len(b) == 54 + 3 * 16   # rows padded to 4 bytes (stb_image_write.h:470-477)
    top_row = b[54 + 2 * 16: 54 + 3 * 16]              # bottom-up file order: image row 0 is last
    assert list(top_row[:3]) == [63, 127, 255]          # BGR, truncating conversion


def test_fast_pairing_equals_reference_scan(emul):
    """The O(n^2 log n) sorted-pair form of BVHAccel::generateBVH's greedy pairing (rt_scene.cpp) must
    build the same hierarchy as the reference's own O(n^3) scan, ties included: 21 x 19 spheres on an
    integer lattice (most centroid distances occur many times), then the same with jittered centres."""
    import os
    import numpy as np
    from raytracer_amd import scene as S
    rng = np.random.default_rng(3)
    for jitter in (0.0, 0.4):
        w = S.World()
        w.width, w.height = 16, 16
        w.sampler = S.Regular(1, 1)
        mat = S.Matte()
        bvh = S.BVHAccel()
        for y in range(19):
            for x in range(21):
                c = (float(x) + jitter * rng.random(), float(y) + jitter * rng.random(), float((x * y) % 3))
                bvh.addShape(S.Sphere(c, 0.3).setMaterial(mat))
        bvh.generateBVH()
        w.scene = [bvh]
        w.ambient = S.Ambient(0, 1000.0)
        w.lights = []
        w.camera = S.Camera((0, 0, 50), (10, 10, 0), (0, 1, 0), 20.0)
        flat = w.flatten()
        fast = emul.bvh_dump(flat, cap=2048)
        os.environ["EMUL_REFERENCE_SCAN"] = "1"
        try:
            slow = emul.bvh_dump(flat, cap=2048)
        finally:
            del os.environ["EMUL_REFERENCE_SCAN"]
        assert len(fast) >= 2 * 399 - 1
        assert fast.shape == slow.shape and np.array_equal(fast.view(np.uint32), slow.view(np.uint32))
