"""Scene builders for the BASELINE.json configurations.

The reference's `teapot.obj` / `sponza.obj` are not in its repository (its .gitignore drops
*.obj), so every mesh here is generated procedurally, identically for the oracle, the reference
harness and the GPU.  The teapot texture (`models/teapot/default.png`, a 128x128 16x16-cell grey
checker, 199/231) is regenerated procedurally; tests/test_oracle_ref.py checks it against the
reference's own decode of that file when /root/reference is present.
"""
import numpy as np

from . import scene as S

f32 = np.float32


# ---- procedural assets ---------------------------------------------------------------------------
def checker_texture(size=128, cells=16, lo=199, hi=231):
    """RGBA8 grey checker equal to the decoded models/teapot/default.png."""
    c = size // cells
    yy, xx = np.mgrid[0:size, 0:size]
    v = np.where(((xx // c) + (yy // c)) % 2 == 0, lo, hi).astype(np.uint8)
    rgba = np.stack([v, v, v, np.full_like(v, 255)], axis=-1)
    return S.ImageTexture(rgba)


def _grid_mesh(P, nu, nv, wrap_u, wrap_v, tile_u, tile_v, flip=False):
    """Triangulate a (nu+1 or nu) x (nv+1 or nv) grid of points P[i,j] (float64 [.,.,3]) into an
    un-indexed triangle soup with per-quad tiled uv in [0,1)."""
    iu = np.arange(nu)
    jv = np.arange(nv)
    I, J = np.meshgrid(iu, jv, indexing="ij")
    I1 = (I + 1) % P.shape[0] if wrap_u else I + 1
    J1 = (J + 1) % P.shape[1] if wrap_v else J + 1
    p00, p10, p01, p11 = P[I, J], P[I1, J], P[I, J1], P[I1, J1]
    u0 = np.mod(I * tile_u / nu, 1.0)
    v0 = np.mod(J * tile_v / nv, 1.0)
    u1 = np.minimum(u0 + tile_u / nu, 0.999)
    v1 = np.minimum(v0 + tile_v / nv, 0.999)
    t00 = np.stack([u0, v0], -1); t10 = np.stack([u1, v0], -1)
    t01 = np.stack([u0, v1], -1); t11 = np.stack([u1, v1], -1)
    if flip:
        A = np.stack([p00, p10, p01], axis=2); TA = np.stack([t00, t10, t01], axis=2)
        B = np.stack([p10, p11, p01], axis=2); TB = np.stack([t10, t11, t01], axis=2)
    else:
        A = np.stack([p00, p01, p10], axis=2); TA = np.stack([t00, t01, t10], axis=2)
        B = np.stack([p10, p01, p11], axis=2); TB = np.stack([t10, t01, t11], axis=2)
    V = np.concatenate([A.reshape(-1, 3, 3), B.reshape(-1, 3, 3)]).astype(f32)
    T = np.concatenate([TA.reshape(-1, 3, 2), TB.reshape(-1, 3, 2)]).astype(f32)
    # drop degenerate triangles (poles)
    e1 = V[:, 1] - V[:, 0]; e2 = V[:, 2] - V[:, 0]
    keep = np.linalg.norm(np.cross(e1.astype(np.float64), e2.astype(np.float64)), axis=1) > 1e-9
    return V[keep], T[keep]


def _soup_to_objshape(V, T, material):
    n = V.shape[0]
    return S.ObjShape(V.reshape(-1, 3), T.reshape(-1, 2), np.arange(3 * n).reshape(n, 3), material)


def torus_objmesh(material, nu=64, nv=32, R=60.0, r=25.0, centre=(0.0, 25.0, 0.0), tile=(4, 2)):
    """Teapot stand-in: nu*nv*2 outward-facing triangles (4096 by default), one OBJ shape."""
    th = np.arange(nu) * (2 * np.pi / nu)
    ph = np.arange(nv) * (2 * np.pi / nv)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    P = np.stack([(R + r * np.cos(PH)) * np.cos(TH), r * np.sin(PH), (R + r * np.cos(PH)) * np.sin(TH)], -1)
    P = P + np.asarray(centre)[None, None, :]
    V, T = _grid_mesh(P, nu, nv, True, True, tile[0], tile[1])
    return S.ObjMesh([_soup_to_objshape(V, T, material)])


def bumpy_sphere_objmesh(material, nu=512, nv=256, radius=250.0, bump=0.08, centre=(0.0, 0.0, 0.0),
                         tile=(32, 16), waves=(9, 7)):
    """Large textured mesh for config 3/5: a sphere displaced by a product of sines, ~2*nu*nv tris."""
    th = np.arange(nu) * (2 * np.pi / nu)
    ph = np.linspace(0.0, np.pi, nv + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    rad = radius * (1.0 + bump * np.sin(waves[0] * TH) * np.sin(waves[1] * PH))
    P = np.stack([rad * np.sin(PH) * np.cos(TH), rad * np.cos(PH), rad * np.sin(PH) * np.sin(TH)], -1)
    P = P + np.asarray(centre)[None, None, :]
    V, T = _grid_mesh(P, nu, nv, True, False, tile[0], tile[1], flip=True)
    return S.ObjMesh([_soup_to_objshape(V, T, material)])


# ---- the reference's default scene (config 1; data from main.cpp:1678-1801) ---------------------------
def _room_materials():
    def spec(col, ka=None, kd=None):
        m = S.Specular()
        m.setDiffuseColour(col)
        if ka is not None: m.setAmbientReflection(ka)
        if kd is not None: m.setDiffuseReflection(kd)
        return m
    return dict(
        specular=spec((0.9, 0.1, 0.4), kd=2.0),
        ps1=spec((1.0, 0.1, 0.1), ka=1.0),
        ps2=spec((0.1, 0.1, 1.0), ka=1.0),
        ps3=spec((0.85, 0.85, 0.85), ka=1.0),
        ps4=spec((0.1, 1.0, 0.1), ka=1.0),
    )


def _room_planes(m):
    p1 = S.Plane((0, -100, 0), (0, 1, 0)).setMaterial(m["ps3"])
    p3 = S.Plane((0, 550, 0), (0, -1, 0)).setMaterial(m["ps3"])
    p4 = S.Plane((-1050, 0, 0), (1, 0, 0)).setMaterial(m["ps3"])
    p2 = S.Plane((350, 0, 0), (-1, 0, 0)).setMaterial(m["ps4"])
    p5 = S.Plane((0, 0, -450), (0, 0, 1)).setMaterial(m["ps1"])
    p6 = S.Plane((0, 0, 450), (0, 0, -1)).setMaterial(m["ps2"])
    return [p1, p2, p3, p4, p5, p6]


def default_scene(width=1000, height=1000, sampler=None, ao_samples=4, mesh_nu=64, mesh_nv=32,
                  mirrors=True, frustum_dist=None):
    """Config 1: the scene main() builds, with a torus in place of the absent teapot.obj.
    mirrors=False replaces Mirror/Refraction by Specular (config 2: primary + shadow rays only)."""
    w = S.World()
    w.width, w.height = width, height
    w.sampler = sampler or S.Jitter(4, 83)
    m = _room_materials()
    tex = checker_texture()
    mesh_tex = S.Textured(tex)                 # what MultiMesh builds from the .mtl (defaults)
    teapot_tex = S.Textured(tex); teapot_tex.setAmbientReflection(0.7)
    if mirrors:
        mirror = S.Mirror(); mirror.setAmbientColour((0.753, 0.753, 0.753)); mirror.setAmbientReflection(0.0)
        refract = S.Refraction()
    else:
        mirror = S.Specular(); mirror.setDiffuseColour((0.753, 0.753, 0.753))
        refract = S.Specular(); refract.setDiffuseColour((0.6, 0.8, 0.9))
    obj = torus_objmesh(mesh_tex, mesh_nu, mesh_nv)

    t1 = S.Triangle((-200, 0, -300), (200, 0, -300), (0, 400, -300), (0, 0), (1, 0), (0.5, 1)).setMaterial(mirror)
    t2 = S.Triangle((200, 0, 0), (200, 0, 400), (200, 400, 200), (0, 0), (1, 0), (0.5, 1)).setMaterial(teapot_tex)
    s1 = S.Sphere((-450, 0, -350), 100).setMaterial(m["specular"])
    s2 = S.Sphere((-350, 0, 300), 100).setMaterial(mirror)
    s3 = S.Sphere((-250, 0, 80), 100).setMaterial(mirror)
    s4 = S.Sphere((-250, 180, 150), 100).setMaterial(refract)
    s5 = S.Sphere((-600, -50, -50), 30).setMaterial(m["ps4"])
    s6 = S.Sphere((-620, -30, -90), 50).setMaterial(m["ps1"])
    s7 = S.Sphere((-585, -20, -55), 40).setMaterial(m["ps2"])
    mm1 = S.MultiMesh(obj, "teapot", (40, 0, -200))
    mm2 = S.MultiMesh(obj, "teapot", (-430, -100, 400))
    mm3 = S.MultiMesh(obj, "teapot", (-150, 500, 200))
    bvh = S.BVHAccel()
    for s in (t1, t2, mm1, mm2, mm3, s1, s2, s3, s4, s5, s6, s7):
        bvh.addShape(s)
    bvh.generateBVH()
    w.scene = _room_planes(m) + [bvh]
    w.ambient = S.Ambient(ao_samples, 1000.0)
    w.ambient.setColour((1, 1, 1)); w.ambient.scaleRadiance(0.5)
    light = S.PointLight((-1000, 300, -400)); light.setColour((1, 1, 1)); light.scaleRadiance(1.0)
    w.lights = [light]
    fd = frustum_dist if frustum_dist is not None else 700.0 * width / 1000.0
    w.camera = S.Camera((-1000, 300, 250), (0, 0, -100), (0, 1, 0), fd)
    return w


def config2_scene(width=1920, height=1080):
    """Config 2: default geometry, 1 spp, jitter off, flat ambient, primary + shadow rays only."""
    return default_scene(width, height, sampler=S.Regular(1, 1), ao_samples=0, mirrors=False)


def config3_scene(width=3840, height=2160, spp=16, mesh_nu=700, mesh_nv=360):
    """Config 3 (the headline metric's scene): room + one large textured mesh (~0.5 M triangles)
    with a triangle BVH, 16 jittered spp, 4x4 ambient occlusion, one point light."""
    w = S.World()
    w.width, w.height = width, height
    w.sampler = S.Jitter(spp, 83)
    m = _room_materials()
    tex = checker_texture()
    mesh_tex = S.Textured(tex); mesh_tex.setAmbientReflection(0.7)
    obj = bumpy_sphere_objmesh(mesh_tex, mesh_nu, mesh_nv, radius=260.0, centre=(-250.0, 170.0, 0.0))
    mm = S.MultiMesh(obj, "teapot", (0, 0, 0))
    s1 = S.Sphere((-450, 0, -330), 100).setMaterial(m["specular"])
    s2 = S.Sphere((100, -20, 250), 80).setMaterial(m["ps2"])
    bvh = S.BVHAccel()
    for s in (mm, s1, s2):
        bvh.addShape(s)
    bvh.generateBVH()
    w.scene = _room_planes(m) + [bvh]
    w.ambient = S.Ambient(4, 1000.0)
    w.ambient.setColour((1, 1, 1)); w.ambient.scaleRadiance(0.5)
    light = S.PointLight((-1000, 300, -400)); light.setColour((1, 1, 1)); light.scaleRadiance(1.0)
    w.lights = [light]
    w.camera = S.Camera((-1000, 300, 250), (0, 0, -100), (0, 1, 0), 700.0 * height / 1000.0)
    return w


def config4_scene(width=3840, height=2160, spp=64):
    """Config 4: default geometry plus extra Mirror / Refraction spheres and a mirror wall pair so
    chains reach depth 8 (render with max_depth=8)."""
    w = default_scene(width, height, sampler=S.Jitter(spp, 83), frustum_dist=700.0 * height / 1000.0)
    mirror = S.Mirror(); mirror.setAmbientColour((0.753, 0.753, 0.753)); mirror.setAmbientReflection(0.0)
    refract = S.Refraction()
    bvh = w.scene[-1]
    extra = [
        S.Sphere((-700, 50, 200), 90).setMaterial(refract),
        S.Sphere((-100, 300, -150), 80).setMaterial(mirror),
        S.Triangle((340, -100, -440), (340, -100, 440), (340, 540, 0)).setMaterial(mirror),
        S.Triangle((-1040, -100, 440), (-1040, -100, -440), (-1040, 540, 0)).setMaterial(mirror),
    ]
    for s in extra:
        bvh.addShape(s)
    return w


def config5_scene(width=7680, height=4320, spp=256, instances=2500, mesh_nu=64, mesh_nv=32, seed=12345):
    """Config 5: ~10 M triangles = `instances` pre-translated copies of a 4096-triangle mesh on a
    jittered grid (the reference's instancing = vertex copies, main.cpp:859-864)."""
    w = S.World()
    w.width, w.height = width, height
    w.sampler = S.Jitter(spp, 83)
    m = _room_materials()
    tex = checker_texture()
    mesh_tex = S.Textured(tex)
    obj = torus_objmesh(mesh_tex, mesh_nu, mesh_nv, R=9.0, r=4.0, centre=(0, 0, 0))
    g = int(np.ceil(np.sqrt(instances)))
    state = int(seed)
    bvh = S.BVHAccel()
    count = 0
    for gy in range(g):
        for gx in range(g):
            if count >= instances:
                break
            vals = []
            for _ in range(3):
                state = (state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
                vals.append(float((state >> 33) / float(1 << 31)))
            x = -1000.0 + (gx + 0.25 + 0.5 * vals[0]) * (1300.0 / g)
            z = -420.0 + (gy + 0.25 + 0.5 * vals[1]) * (840.0 / g)
            y = -80.0 + 500.0 * vals[2]
            bvh.addShape(S.MultiMesh(obj, "teapot", (x, y, z)))
            count += 1
    bvh.generateBVH()
    w.scene = _room_planes(m) + [bvh]
    w.ambient = S.Ambient(4, 1000.0)
    w.ambient.setColour((1, 1, 1)); w.ambient.scaleRadiance(0.5)
    light = S.PointLight((-1000, 300, -400)); light.setColour((1, 1, 1)); light.scaleRadiance(1.0)
    w.lights = [light]
    w.camera = S.Camera((-1000, 300, 250), (0, 0, -100), (0, 1, 0), 700.0 * height / 1000.0)
    return w
