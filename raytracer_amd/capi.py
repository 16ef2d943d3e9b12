"""ctypes mirror of include/rt_amd.h and the loader of the HIP library.

The product path has no CPU fallback: `load_library()` raises if `librt_amd.so` (built by
`__graft_entry__.build()` / `make -C raytracer_amd/csrc`) is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_amd.so")

# ---- enums (include/rt_amd.h) -------------------------------------------------------------
RT_SHAPE_PLANE, RT_SHAPE_SPHERE, RT_SHAPE_TRIANGLE, RT_SHAPE_MESH, RT_SHAPE_MULTIMESH, RT_SHAPE_BVH = range(6)
RT_MAT_MATTE, RT_MAT_SPECULAR, RT_MAT_TEXTURED, RT_MAT_MIRROR, RT_MAT_REFRACTION = range(5)
RT_LIGHT_POINT, RT_LIGHT_DIRECTIONAL = range(2)
RT_SAMPLER_REGULAR, RT_SAMPLER_JITTER, RT_SAMPLER_RANDOM = range(3)
RT_FLAG_STATS, RT_FLAG_TIMING = 1, 2
RT_TILE = 64
RT_K_NAMES = ["raygen", "trace", "trace_bvh", "prep", "shadow", "ao", "ao_bvh", "shade", "resolve"]
RT_K_COUNT = len(RT_K_NAMES)
RT_MISS = 0xFFFFFFFF

# ---- array element layouts (all fields 4-byte, so numpy's packed layout == the C layout) ---
plane_dt = np.dtype([("point", "<f4", 3), ("normal", "<f4", 3), ("material", "<i4")])
sphere_dt = np.dtype([("centre", "<f4", 3), ("radius", "<f4"), ("material", "<i4")])
triangle_dt = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3),
                        ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2), ("material", "<i4")])
mesh_dt = np.dtype([("first_triangle", "<u4"), ("num_triangles", "<u4")])
multimesh_dt = np.dtype([("first_mesh", "<u4"), ("num_meshes", "<u4")])
shape_ref_dt = np.dtype([("kind", "<i4"), ("index", "<u4")])
bvh_dt = np.dtype([("first_leaf", "<u4"), ("num_leaves", "<u4")])
material_dt = np.dtype([("kind", "<i4"), ("colour", "<f4", 3), ("kd", "<f4"), ("ka", "<f4"), ("ks", "<f4"),
                        ("spec_exp", "<i4"), ("ior", "<f4"), ("texture", "<i4")])
light_dt = np.dtype([("kind", "<i4"), ("v", "<f4", 3), ("colour", "<f4", 3), ("radiance", "<f4")])

assert plane_dt.itemsize == 28 and sphere_dt.itemsize == 20 and triangle_dt.itemsize == 64
assert material_dt.itemsize == 40 and light_dt.itemsize == 32


class rt_texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.c_void_p)]


class rt_ambient(C.Structure):
    _fields_ = [("colour", C.c_float * 3), ("radiance", C.c_float),
                ("occlusion_samples", C.c_int32), ("occlusion_dist", C.c_float)]


class rt_camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("look_at", C.c_float * 3), ("up", C.c_float * 3),
                ("frustum_dist", C.c_float)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("planes", C.c_void_p), ("num_planes", C.c_uint32),
        ("spheres", C.c_void_p), ("num_spheres", C.c_uint32),
        ("triangles", C.c_void_p), ("num_triangles", C.c_uint32),
        ("meshes", C.c_void_p), ("num_meshes", C.c_uint32),
        ("multimeshes", C.c_void_p), ("num_multimeshes", C.c_uint32),
        ("bvhs", C.c_void_p), ("num_bvhs", C.c_uint32),
        ("bvh_leaves", C.c_void_p), ("num_bvh_leaves", C.c_uint32),
        ("scene", C.c_void_p), ("num_scene", C.c_uint32),
        ("materials", C.c_void_p), ("num_materials", C.c_uint32),
        ("textures", C.POINTER(rt_texture)), ("num_textures", C.c_uint32),
        ("lights", C.c_void_p), ("num_lights", C.c_uint32),
        ("ambient", rt_ambient),
    ]


class rt_params(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
        ("num_samples", C.c_int32), ("num_sets", C.c_int32), ("sampler", C.c_int32),
        ("max_depth", C.c_int32), ("light_attenuation", C.c_float), ("seed", C.c_uint32),
        ("tile_rank", C.c_int32), ("tile_world", C.c_int32),
        ("batch_paths", C.c_uint32), ("flags", C.c_uint32),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_ao", C.c_uint64), ("rays_bounce", C.c_uint64),
        ("nodes", C.c_uint64 * 3), ("tris", C.c_uint64 * 3), ("gates", C.c_uint64 * 3), ("bvh_rays", C.c_uint64 * 3),
        ("ms_total", C.c_double), ("ms_kernel", C.c_double * RT_K_COUNT),
        ("launches", C.c_uint32 * RT_K_COUNT), ("batches", C.c_uint32),
        ("rays_skipped", C.c_uint64),
    ]

    @property
    def rays(self):
        """closest-hit queries this path traced"""
        return self.rays_primary + self.rays_shadow + self.rays_ao + self.rays_bounce

    @property
    def rays_reference_unit(self):
        """the reference's own query count for the same frame: what was traced + what was proven unnecessary"""
        return self.rays + self.rays_skipped


def default_params(width=1000, height=1000, num_samples=4, num_sets=83, sampler=RT_SAMPLER_JITTER,
                   max_depth=3, seed=1, **kw):
    """rt_params with the reference constants (Raytracer.hpp:52-53, main.cpp:1680-1683)."""
    p = rt_params(width=width, height=height, x0=0, y0=0, x1=width, y1=height,
                  num_samples=num_samples, num_sets=num_sets, sampler=sampler, max_depth=max_depth,
                  light_attenuation=500000.0, seed=seed, tile_rank=0, tile_world=1, batch_paths=0, flags=0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# ---- the HIP library ------------------------------------------------------------------------
RT_BUILDER_HOST_SAH, RT_BUILDER_GPU_LBVH = 0, 1

_EXPORTS = [
    "rt_init", "rt_shutdown", "rt_last_error", "rt_version", "rt_params_default",
    "rt_scene_create", "rt_scene_set_planes", "rt_scene_set_spheres", "rt_scene_set_triangles",
    "rt_scene_set_meshes", "rt_scene_set_multimeshes", "rt_scene_set_bvhs", "rt_scene_set_scene",
    "rt_scene_set_materials", "rt_scene_set_textures", "rt_scene_set_lights", "rt_scene_set_ambient",
    "rt_scene_set_desc", "rt_scene_commit", "rt_scene_destroy",
    "rt_render", "rt_render_device", "rt_owned_tiles", "rt_unpack_tiles", "rt_quantize_rgb8",
    "rt_bmp24_bytes", "rt_encode_bmp24", "rt_scene_set_builder", "rt_scene_builder_used",
    "rt_render_device_noalloc", "rt_render_multi", "rt_selftest_device_math",
]
RT_NOALLOC_SCRATCH_BYTES = 384 << 20

_lib = None


def exported_symbols():
    return list(_EXPORTS)


def _share_the_process_hip_runtime():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7) and load it by file name; librt_amd.so needs `libamdhip64.so.7`.  If torch's copy is loaded
    first, ours binds to it by SONAME and all is well; if ours pulls in /opt/rocm's copy first, a later `import torch`
    loads a SECOND runtime stack whose initialisation fails ("No HIP GPUs are available").  So when torch is
    installed and no HIP runtime is mapped yet, map torch's copy first.  Nothing is imported or initialised here."""
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            cand = os.path.join(libdir, name)
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass


def load_library(path=None):
    """dlopen librt_amd.so.  Raises (no fallback) when the HIP extension has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the render path.")
    _share_the_process_hip_runtime()
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.rt_init.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_shutdown.argtypes = [vp]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_version.restype = C.c_int
    lib.rt_params_default.argtypes = [C.POINTER(rt_params)]
    lib.rt_params_default.restype = None
    lib.rt_scene_create.argtypes = [vp, C.POINTER(vp)]
    for n in ("planes", "spheres", "triangles", "meshes", "multimeshes", "scene", "materials", "lights"):
        getattr(lib, f"rt_scene_set_{n}").argtypes = [vp, vp, C.c_size_t]
    lib.rt_scene_set_bvhs.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    lib.rt_scene_set_textures.argtypes = [vp, C.POINTER(rt_texture), C.c_size_t]
    lib.rt_scene_set_ambient.argtypes = [vp, C.POINTER(rt_ambient)]
    lib.rt_scene_set_desc.argtypes = [vp, C.POINTER(rt_scene_desc)]
    lib.rt_scene_commit.argtypes = [vp]
    lib.rt_scene_destroy.argtypes = [vp]
    lib.rt_render.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, vp, vp, C.POINTER(rt_stats)]
    lib.rt_render_device.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, vp, vp,
                                     C.POINTER(rt_stats), vp]
    lib.rt_render_device_noalloc.argtypes = [vp, C.POINTER(rt_camera), C.POINTER(rt_params), vp, vp, vp, vp]
    lib.rt_render_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(rt_camera), C.POINTER(rt_params), vp, C.POINTER(rt_stats)]
    lib.rt_owned_tiles.argtypes = [C.POINTER(rt_params), C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    lib.rt_unpack_tiles.argtypes = [vp, vp, C.POINTER(rt_params), C.c_int, C.c_int, vp, vp]
    lib.rt_quantize_rgb8.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.rt_selftest_device_math.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.rt_scene_set_builder.argtypes = [vp, C.c_int]
    lib.rt_scene_builder_used.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.rt_bmp24_bytes.argtypes = [C.c_int, C.c_int]
    lib.rt_bmp24_bytes.restype = C.c_size_t
    lib.rt_encode_bmp24.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    if path == LIB_PATH:
        _lib = lib
    return lib
