"""Host build of the device functions (tests/emul) against the C oracle: exercises the flattener,
the triangle BVH (conservative culling must never lose a hit the reference's linear scan finds),
the wavefront queues and the per-level shading stack without a GPU."""
import numpy as np
import pytest

import util
from raytracer_amd import capi, scenes, scene as S


def check(emul, port, world, params, min_exact=0.95):
    flat = world.flatten()
    want = port.render(flat, world.camera, params, rng_mode=1, nthreads=4)
    rgb, prim, t, cnt = emul.render(flat, world.camera, params)
    assert np.array_equal(prim, want.primid)
    assert util.bits_equal(t, want.t)
    assert util.lsb_diff(rgb, want.rgb).max() <= 1
    assert (rgb.view(np.uint32) == want.rgb.view(np.uint32)).mean() >= min_exact
    return cnt, want


def test_mesh_heavy_view(emul, port):
    """Camera close to a mesh: most rays run the triangle BVH, AO rays start inside the mesh box
    (negative-t hits)."""
    w = scenes.default_scene(72, 72, sampler=S.Regular(1, 1), ao_samples=4, mesh_nu=24, mesh_nv=12)
    w.camera = S.Camera((-120, 120, -60), (40, 25, -200), (0, 1, 0), 90.0)
    cnt, want = check(emul, port, w, w.params(seed=9))
    assert cnt[4 + 2] > 0 and cnt[7 + 2] > 0   # AO rays visited BVH nodes and tested triangles


def test_camera_inside_mesh_box(emul, port):
    w = scenes.config3_scene(48, 32, spp=1, mesh_nu=40, mesh_nv=20)
    w.camera = S.Camera((-250, 170, 0), (0, 0, -100), (0, 1, 0), 40.0)   # at the sphere's centre
    check(emul, port, w, w.params(seed=4))


def test_many_lights_and_matte(emul, port):
    w = scenes.default_scene(48, 48, sampler=S.Jitter(4, 3), ao_samples=2, mesh_nu=16, mesh_nv=8)
    l2 = S.PointLight((200, 500, 300)); l2.setColour((0.9, 0.8, 0.7)); l2.scaleRadiance(0.6)
    l3 = S.Directional((0.3, 1.0, 0.2)); l3.setColour((0.5, 0.5, 1.0)); l3.scaleRadiance(0.4)
    w.lights += [l2, l3]
    matte = S.Matte(); matte.setDiffuseColour((0.2, 0.7, 0.9)); matte.setDiffuseReflection(0.8); matte.setAmbientReflection(0.3)
    w.scene.insert(0, S.Sphere((-500, 100, 100), 60).setMaterial(matte))
    check(emul, port, w, w.params(seed=21), min_exact=0.90)


def test_shadow_through_transparent_sphere(emul, port):
    """Material::shadowed returns 0.8 when only Refraction surfaces block the light."""
    w = S.World(); w.width, w.height = 64, 48; w.sampler = S.Regular(1, 1)
    grey = S.Specular(); grey.setDiffuseColour((0.8, 0.8, 0.8))
    glass = S.Refraction()
    red = S.Specular(); red.setDiffuseColour((0.9, 0.2, 0.2))
    floor = S.Plane((0, -100, 0), (0, 1, 0)).setMaterial(grey)
    s1 = S.Sphere((0, 60, 0), 60).setMaterial(glass)
    s2 = S.Sphere((220, 0, 40), 50).setMaterial(red)
    bvh = S.BVHAccel(); bvh.addShape(s1); bvh.addShape(s2); bvh.generateBVH()
    w.scene = [floor, bvh]
    w.ambient = S.Ambient(0); w.ambient.setColour((1, 1, 1)); w.ambient.scaleRadiance(0.2)
    light = S.PointLight((0, 600, 0)); light.setColour((1, 1, 1)); light.scaleRadiance(1.0)
    w.lights = [light]
    w.camera = S.Camera((0, 200, 600), (0, 0, 0), (0, 1, 0), 60.0)
    cnt, want = check(emul, port, w, w.params())
    assert len(np.unique(want.rgb.round(3))) > 3


def test_single_leaf_bvh_and_lone_shapes(emul, port):
    """BVHAccel with one leaf calls shape->hit without any gate (main.cpp:430-438); Mesh and
    Triangle directly in world->scene."""
    w = S.World(); w.width, w.height = 40, 40; w.sampler = S.Regular(4, 1)
    tex = S.Textured(scenes.checker_texture())
    obj = scenes.torus_objmesh(tex, 12, 8, R=60, r=25)
    bvh = S.BVHAccel(); bvh.addShape(S.MultiMesh(obj, "", (0, 0, 0))); bvh.generateBVH()
    mesh = S.Mesh(scenes.torus_objmesh(tex, 10, 6, R=40, r=15).shapes[0], (150, 0, 0))
    tri = S.Triangle((-200, -50, -100), (200, -50, -100), (0, 250, -100), (0, 0), (1, 0), (0.5, 1)).setMaterial(tex)
    w.scene = [bvh, mesh, tri]
    w.ambient = S.Ambient(2, 500.0); w.ambient.setColour((1, 1, 1)); w.ambient.scaleRadiance(0.5)
    light = S.PointLight((100, 400, 500)); light.setColour((1, 1, 1)); light.scaleRadiance(1.0)
    w.lights = [light]
    w.camera = S.Camera((0, 80, 400), (0, 20, 0), (0, 1, 0), 40.0)
    check(emul, port, w, w.params(seed=2))


def test_empty_scene_and_misses(emul, port):
    w = S.World(); w.width, w.height = 16, 8; w.sampler = S.Regular(1, 1)
    w.scene = []
    w.ambient = S.Ambient(0)
    w.camera = S.Camera((0, 0, 10), (0, 0, 0), (0, 1, 0), 10.0)
    flat = w.flatten()
    rgb, prim, t, cnt = emul.render(flat, w.camera, w.params())
    assert (prim == capi.RT_MISS).all() and (rgb == 0).all() and (t == np.finfo(np.float32).max).all()


def test_tile_split_matches_full_frame(emul):
    """The multi-GPU partition on the host build: ranks 0..2 of 3 render disjoint tiles whose union is the frame."""
    name, world, params = util.small_scenes()[0]
    p = capi.rt_params.from_buffer_copy(params)
    flat = world.flatten()
    full, _, _, _ = emul.render(flat, world.camera, p)
    acc = np.zeros_like(full)
    for r in range(3):
        q = capi.rt_params.from_buffer_copy(p)
        q.tile_rank, q.tile_world = r, 3
        part, _, _, _ = emul.render(flat, world.camera, q)
        acc += part
    assert util.bits_equal(acc, full)


def test_more_meshes_on_a_ray_than_deferred_roots(emul, port):
    w = util.mesh_tunnel_scene()
    cnt, want = check(emul, port, w, w.params(seed=5), min_exact=0.90)
    assert (want.primid != capi.RT_MISS).mean() > 0.5


def test_gate_classification_is_sound(emul):
    """gate_maybe == False must imply the reference box test fails, gate_surely == True that it
    passes (rt_core.hpp); 4 M random + adversarial rays (edges, corners, planes, tiny components).
    Round 3: the fused gate_class (0 <=> gate_maybe false, 1 => the exact test passes) and gate_enter == the exact test."""
    import ctypes as C
    import numpy as np
    lib = emul.lib
    lib.emul_gate_check.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    out = np.zeros(6, np.float64)
    for seed in (1, 2):
        assert lib.emul_gate_check(2_000_000, seed, out.ctypes.data) == 0
        tested, no, yes, amb, bad, fused_yes = out
        assert bad == 0, out
        assert fused_yes > 0.8 * yes, out                 # the cheap accept decides most of what gate_surely accepts
        assert no > 0.1 * tested and yes > 0.1 * tested   # both shortcuts are exercised
        assert amb < 0.35 * tested                        # the adversarial modes are ~3/8 of the cases


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_scenes(emul, port, seed):
    """Fuzz (util.random_scene): the device code on the host must reproduce the oracle's primitive ids
    and ray parameters exactly (conservative culling / gate shortcuts may never change a result)."""
    w, params = util.random_scene(seed)
    flat = w.flatten()
    want = port.render(flat, w.camera, params, rng_mode=1, nthreads=4)
    rgb, prim, t, cnt = emul.render(flat, w.camera, params)
    assert np.array_equal(prim, want.primid)
    assert util.bits_equal(t, want.t)
    assert util.lsb_diff(rgb, want.rgb).max() <= 1


def test_queue_chunk_properties(emul):
    """queue_chunk(): a power of two in [64, largest], never more than a quarter of a wave's expected appends
    (above the floor of 64), monotone in the number of appends, and the largest chunk for the bench's launches."""
    import ctypes as C
    f = emul.lib.emul_queue_chunk
    f.argtypes = [C.c_ulonglong, C.c_uint32, C.c_uint32]
    f.restype = C.c_uint32
    for largest in (512, 1024):
        for waves in (1, 4, 6144, 8192, 131072):
            last = 0
            for appends in (0, 1, 63, 64, 1000, 10**5, 10**6, 10**7, 10**8, 534_773_760, 2**32):
                c = f(appends, waves, largest)
                assert 64 <= c <= largest and c & (c - 1) == 0
                assert c == 64 or c * 4 * waves <= appends
                assert c >= last
                last = c
    assert f(33_423_360, 6144, 512) == 512          # k_prep, config 3: 1536 blocks of 4 waves
    assert f(534_773_760, 131072, 1024) == 512      # k_ao, config 3: 32768 blocks
    assert f(2_000_000, 8192, 1024) == 64           # a small frame


def test_compact_gate_walk_equals_the_full_one(emul):
    """k_ao_walk_lds walks 16-byte copies of the gate nodes (boxes on a 16-bit grid, rounded outwards) out of LDS and
    fetches the exact box only when the decoded one may be entered (gate_step_q, rt_core.hpp).  On an instanced scene
    (config 5's structure: a BVHAccel over hundreds of MultiMeshes) and on the reference's own scene, random rays and
    rays aimed at the corners / edges / faces of the gate boxes take the same walk as with the full nodes: same
    sequence of nodes, same closest primitive, same mesh roots."""
    w = scenes.config5_scene(64, 48, 1, instances=300, mesh_nu=6, mesh_nv=4)
    rays, steps, from_q, bad, n = emul.qgate_compare(w.flatten(), 60000, seed=3)
    assert n > 600 and rays > 59000
    assert bad == 0
    assert from_q > 0.3 * steps   # the compact copy alone decides a good part of the steps (misses + same-box gates)
    w = scenes.default_scene(64, 48, mesh_nu=8, mesh_nv=4)
    rays, steps, from_q, bad, n = emul.qgate_compare(w.flatten(), 40000, seed=5)
    assert n > 0 and bad == 0
