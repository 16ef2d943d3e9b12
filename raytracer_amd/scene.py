"""Host-side mirror of the reference's scene classes (Raytracer.hpp) for Python callers.

Same names, argument meaning and defaults as the reference (`Sphere(centre, radius)`,
`Plane(point, normal)`, `Triangle(p0,p1,p2,uv0,uv1,uv2)`, `MultiMesh(objMesh, subdir, offset)`,
`BVHAccel.addShape(...)/generateBVH()`, material setters, `PointLight(p)`, `Camera(...)`, ...).
`World.flatten()` turns the object graph into the flat arrays of include/rt_amd.h.  Nothing
here computes pixels: that is the HIP library's job.
"""
import ctypes as C

import numpy as np

from . import capi

f32 = np.float32


def _v3(v):
    a = np.asarray(v, dtype=f32).reshape(3)
    return a


# ---- materials (Raytracer.hpp:516-600) -----------------------------------------------------
class Material:
    kind = None

    def __init__(self):
        self.colour = np.ones(3, f32)   # Lambertian default colour (main.cpp:1038)
        self.kd = f32(0.5)
        self.ka = f32(0.5)
        self.ks = f32(0.3)              # SpecularReflection defaults (main.cpp:1071-1073)
        self.spec_exp = 16
        self.ior = f32(1.7)
        self.texture = None

    @property
    def mTransparent(self):
        return self.kind == capi.RT_MAT_REFRACTION


class Matte(Material):
    kind = capi.RT_MAT_MATTE

    def setDiffuseReflection(self, k): self.kd = f32(k)
    def setAmbientReflection(self, k): self.ka = f32(k)
    def setDiffuseColour(self, c): self.colour = _v3(c)


class Specular(Material):
    kind = capi.RT_MAT_SPECULAR

    def setSpecularCoefficient(self, k): self.ks = f32(k)
    def setSpecularExp(self, k): self.spec_exp = int(k)
    def setDiffuseReflection(self, k): self.kd = f32(k)
    def setAmbientReflection(self, k): self.ka = f32(k)
    def setDiffuseColour(self, c): self.colour = _v3(c)


class Mirror(Material):
    kind = capi.RT_MAT_MIRROR

    def setAmbientColour(self, c): self.colour = _v3(c)
    def setAmbientReflection(self, k): self.ka = f32(k)


class Refraction(Material):
    kind = capi.RT_MAT_REFRACTION

    def setIndexOfRefraction(self, i): self.ior = f32(i)


class ImageTexture:
    """Decoded RGBA8 texels (what stbi_load(..., STBI_rgb_alpha) returns, main.cpp:147-165)."""

    def __init__(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        assert rgba.ndim == 3 and rgba.shape[2] == 4
        self.rgba = rgba
        self.height, self.width = rgba.shape[:2]


class Textured(Material):
    kind = capi.RT_MAT_TEXTURED

    def __init__(self, texture):
        super().__init__()
        self.texture = texture  # ImageTexture or None (mSet == false: black)

    def setAmbientReflection(self, k): self.ka = f32(k)
    def setSpecularReflection(self, k): self.ks = f32(k)
    def setSpecularExp(self, k): self.spec_exp = int(k)


# ---- shapes (Raytracer.hpp:185-413) ----------------------------------------------------------
class Shape:
    def __init__(self):
        self.material = None

    def setMaterial(self, m):
        self.material = m
        return self


class Plane(Shape):
    def __init__(self, point, normal):
        super().__init__()
        self.point, self.normal = _v3(point), _v3(normal)


class Sphere(Shape):
    def __init__(self, centre, radius):
        super().__init__()
        self.centre, self.radius = _v3(centre), f32(radius)


class Triangle(Shape):
    def __init__(self, p0, p1, p2, uv0=(0, 0), uv1=(0, 0), uv2=(0, 0)):
        super().__init__()
        self.v = np.stack([_v3(p0), _v3(p1), _v3(p2)])
        self.uv = np.asarray([uv0, uv1, uv2], dtype=f32)


class ObjShape:
    """atlas::utils::Shape as the reference reads it: indexed positions + texcoords, one material."""

    def __init__(self, positions, texcoords, indices, material=None):
        self.positions = np.asarray(positions, dtype=f32).reshape(-1, 3)
        self.texcoords = np.asarray(texcoords, dtype=f32).reshape(-1, 2)
        self.indices = np.asarray(indices, dtype=np.int64).reshape(-1, 3)
        self.material = material  # a Textured (the reference builds it from the .mtl's map_Kd)


class ObjMesh:
    """atlas::utils::ObjMesh: what loadObjMesh returns (include/atlas/utils/LoadObjFile.hpp:13-39)."""

    def __init__(self, shapes):
        self.shapes = list(shapes)


class Mesh(Shape):
    """One OBJ shape: AABB gate + linear triangle list (main.cpp:851-912).  The offset is added
    to every vertex in float32, as the reference does (main.cpp:859-864)."""

    def __init__(self, obj_shape, offset=(0, 0, 0)):
        super().__init__()
        off = _v3(offset)
        pos = (obj_shape.positions + off[None, :]).astype(f32)
        self.tri_v = pos[obj_shape.indices]                    # [n,3,3]
        self.tri_uv = obj_shape.texcoords[obj_shape.indices]   # [n,3,2]
        self.material = obj_shape.material


class MultiMesh(Shape):
    def __init__(self, obj_mesh, model_sub_dir="", offset=(0, 0, 0)):
        super().__init__()
        self.meshes = [Mesh(s, offset) for s in obj_mesh.shapes]


class BVHAccel(Shape):
    def __init__(self):
        super().__init__()
        self.leaves = []

    def addShape(self, shape, _shape2=None):
        # reference signature: addShape(shape->getBoundingBoxPoints(), shape); the box is always
        # the shape's own, so only the shape is needed here.
        self.leaves.append(_shape2 if _shape2 is not None else shape)

    def generateBVH(self):
        pass  # the hierarchy is rebuilt by rt_scene_commit (reference algorithm, main.cpp:341-423)


# ---- lights / camera / samplers ----------------------------------------------------------------
class Light:
    def __init__(self):
        self.colour = np.zeros(3, f32)
        self.radiance = f32(0)

    def setColour(self, c): self.colour = _v3(c)
    def scaleRadiance(self, b): self.radiance = f32(b)


class PointLight(Light):
    kind = capi.RT_LIGHT_POINT

    def __init__(self, p=(0, 0, 0)):
        super().__init__()
        self.v = _v3(p)


class Directional(Light):
    kind = capi.RT_LIGHT_DIRECTIONAL

    def __init__(self, d=(0, 0, 1)):
        super().__init__()
        self.v = _v3(d)


class Ambient(Light):
    """world->ambient.  occlusion_samples=0 gives the base-class Light::L (no AO rays)."""

    def __init__(self, occlusion_samples=4, occlusion_dist=1000.0):
        super().__init__()
        self.occlusion_samples = int(occlusion_samples)
        self.occlusion_dist = f32(occlusion_dist)


class Camera:
    def __init__(self, position, look_at, up, frustum_dist):
        self.eye, self.look_at, self.up = _v3(position), _v3(look_at), _v3(up)
        self.frustum_dist = f32(frustum_dist)

    def to_c(self):
        c = capi.rt_camera()
        c.eye[:] = [float(x) for x in self.eye]
        c.look_at[:] = [float(x) for x in self.look_at]
        c.up[:] = [float(x) for x in self.up]
        c.frustum_dist = float(self.frustum_dist)
        return c


class Sampler:
    kind = None

    def __init__(self, num_samples, num_sets):
        self.num_samples, self.num_sets = int(num_samples), int(num_sets)


class Regular(Sampler):
    kind = capi.RT_SAMPLER_REGULAR


class Jitter(Sampler):
    kind = capi.RT_SAMPLER_JITTER


class Random(Sampler):
    kind = capi.RT_SAMPLER_RANDOM


# ---- World + flattening --------------------------------------------------------------------------
class FlatScene:
    """The arrays of an rt_scene_desc plus the ctypes struct pointing at them (kept alive here)."""

    def __init__(self):
        self.desc = capi.rt_scene_desc()
        self.arrays = {}
        self.textures = []


class World:
    def __init__(self):
        self.width = 1000
        self.height = 1000
        self.background = np.zeros(3, f32)
        self.sampler = Jitter(4, 83)
        self.scene = []
        self.lights = []
        self.ambient = Ambient()
        self.camera = None
        self.image = None

    def params(self, **kw):
        return capi.default_params(width=self.width, height=self.height,
                                   num_samples=self.sampler.num_samples, num_sets=self.sampler.num_sets,
                                   sampler=self.sampler.kind, **kw)

    def flatten(self):
        mats, mat_ids, texs, tex_ids = [], {}, [], {}

        def tex_id(t):
            if t is None:
                return -1
            if id(t) not in tex_ids:
                tex_ids[id(t)] = len(texs)
                texs.append(t)
            return tex_ids[id(t)]

        def mat_id(m):
            if m is None:
                raise ValueError("shape without a material (the reference would dereference a null Material)")
            if id(m) not in mat_ids:
                mat_ids[id(m)] = len(mats)
                mats.append(m)
            return mat_ids[id(m)]

        planes, spheres, tri_blocks, meshes, multimeshes, bvhs, bvh_leaves, scene = [], [], [], [], [], [], [], []
        ntri = 0
        seen = {}

        def add_tris(v, uv, material):
            nonlocal ntri
            n = v.shape[0]
            blk = np.zeros(n, capi.triangle_dt)
            blk["v0"], blk["v1"], blk["v2"] = v[:, 0], v[:, 1], v[:, 2]
            blk["uv0"], blk["uv1"], blk["uv2"] = uv[:, 0], uv[:, 1], uv[:, 2]
            blk["material"] = mat_id(material)
            tri_blocks.append(blk)
            first = ntri
            ntri += n
            return first

        def add_mesh(m):
            first = add_tris(m.tri_v, m.tri_uv, m.material)
            meshes.append((first, m.tri_v.shape[0]))
            return len(meshes) - 1

        def ref_of(s):
            if id(s) in seen:
                return seen[id(s)]
            if isinstance(s, Plane):
                planes.append((s.point, s.normal, mat_id(s.material)))
                r = (capi.RT_SHAPE_PLANE, len(planes) - 1)
            elif isinstance(s, Sphere):
                spheres.append((s.centre, s.radius, mat_id(s.material)))
                r = (capi.RT_SHAPE_SPHERE, len(spheres) - 1)
            elif isinstance(s, Triangle):
                r = (capi.RT_SHAPE_TRIANGLE, add_tris(s.v[None], s.uv[None], s.material))
            elif isinstance(s, Mesh):
                r = (capi.RT_SHAPE_MESH, add_mesh(s))
            elif isinstance(s, MultiMesh):
                first = len(meshes)
                for m in s.meshes:
                    add_mesh(m)
                multimeshes.append((first, len(s.meshes)))
                r = (capi.RT_SHAPE_MULTIMESH, len(multimeshes) - 1)
            elif isinstance(s, BVHAccel):
                leaves = [ref_of(l) for l in s.leaves]
                bvhs.append((len(bvh_leaves), len(leaves)))
                bvh_leaves.extend(leaves)
                r = (capi.RT_SHAPE_BVH, len(bvhs) - 1)
            else:
                raise TypeError(f"unknown shape {type(s)}")
            seen[id(s)] = r
            return r

        for s in self.scene:
            scene.append(ref_of(s))

        fs = FlatScene()
        A = fs.arrays
        A["planes"] = np.array([(p, n, m) for p, n, m in planes], dtype=capi.plane_dt) if planes else np.zeros(0, capi.plane_dt)
        A["spheres"] = np.array([(c, r, m) for c, r, m in spheres], dtype=capi.sphere_dt) if spheres else np.zeros(0, capi.sphere_dt)
        A["triangles"] = np.concatenate(tri_blocks) if tri_blocks else np.zeros(0, capi.triangle_dt)
        A["meshes"] = np.array(meshes, dtype=capi.mesh_dt) if meshes else np.zeros(0, capi.mesh_dt)
        A["multimeshes"] = np.array(multimeshes, dtype=capi.multimesh_dt) if multimeshes else np.zeros(0, capi.multimesh_dt)
        A["bvhs"] = np.array(bvhs, dtype=capi.bvh_dt) if bvhs else np.zeros(0, capi.bvh_dt)
        A["bvh_leaves"] = np.array(bvh_leaves, dtype=capi.shape_ref_dt) if bvh_leaves else np.zeros(0, capi.shape_ref_dt)
        A["scene"] = np.array(scene, dtype=capi.shape_ref_dt) if scene else np.zeros(0, capi.shape_ref_dt)
        marr = np.zeros(len(mats), capi.material_dt)
        for i, m in enumerate(mats):
            marr[i] = (m.kind, m.colour, m.kd, m.ka, m.ks, m.spec_exp, m.ior, tex_id(m.texture))
        A["materials"] = marr
        larr = np.zeros(len(self.lights), capi.light_dt)
        for i, l in enumerate(self.lights):
            larr[i] = (l.kind, l.v, l.colour, l.radiance)
        A["lights"] = larr
        fs.textures = texs
        tarr = (capi.rt_texture * max(1, len(texs)))()
        for i, t in enumerate(texs):
            tarr[i].width, tarr[i].height = t.width, t.height
            tarr[i].rgba = t.rgba.ctypes.data
        A["textures_c"] = tarr

        d = fs.desc
        for name in ("planes", "spheres", "triangles", "meshes", "multimeshes", "bvhs", "bvh_leaves",
                     "scene", "materials", "lights"):
            arr = A[name] = np.ascontiguousarray(A[name])
            setattr(d, name, arr.ctypes.data if arr.size else None)
            setattr(d, "num_" + name, arr.shape[0])
        d.textures = C.cast(tarr, C.POINTER(capi.rt_texture))
        d.num_textures = len(texs)
        a = self.ambient
        d.ambient.colour[:] = [float(x) for x in a.colour]
        d.ambient.radiance = float(a.radiance)
        d.ambient.occlusion_samples = int(getattr(a, "occlusion_samples", 0))
        d.ambient.occlusion_dist = float(getattr(a, "occlusion_dist", 1000.0))
        return fs
