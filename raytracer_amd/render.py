"""Thin Python binding of the C ABI (include/rt_amd.h): context, scene upload, render.

`Renderer` is what a Python caller uses in place of the reference's ThreadPool + raytrace()
job loop (main.cpp:1803-1827): build a `World` (raytracer_amd.scene), `upload()` it, `render()`.
All pixel work happens in librt_amd.so on the GPU; this module only marshals arguments.
"""
import ctypes as C

import numpy as np

from . import capi


class RtError(RuntimeError):
    pass


class Renderer:
    def __init__(self, device=0, lib_path=None):
        self.lib = capi.load_library(lib_path)
        self.ctx = C.c_void_p()
        rc = self.lib.rt_init(int(device), C.byref(self.ctx))
        if rc != 0:
            raise RtError(f"rt_init failed ({rc}): {self.lib.rt_last_error(None).decode()}")
        self.scene = None
        self.device = int(device)

    # -- errors ------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise RtError(f"{what} failed ({rc}): {self.lib.rt_last_error(self.ctx).decode()}")

    # -- scene -------------------------------------------------------------------------------
    def upload(self, world_or_flat, builder=None):
        """rt_scene_create + rt_scene_set_* + rt_scene_commit.  builder: capi.RT_BUILDER_HOST_SAH (default) or
        capi.RT_BUILDER_GPU_LBVH (triangle BVHs built on the device; see builder_used())."""
        flat = world_or_flat.flatten() if hasattr(world_or_flat, "flatten") else world_or_flat
        if self.scene is not None:
            self.lib.rt_scene_destroy(self.scene)
            self.scene = None
        sc = C.c_void_p()
        self._check(self.lib.rt_scene_create(self.ctx, C.byref(sc)), "rt_scene_create")
        self.scene = sc
        A, d = flat.arrays, flat.desc
        L = self.lib

        def ptr(a):
            return a.ctypes.data if a.size else None
        self._check(L.rt_scene_set_planes(sc, ptr(A["planes"]), len(A["planes"])), "rt_scene_set_planes")
        self._check(L.rt_scene_set_spheres(sc, ptr(A["spheres"]), len(A["spheres"])), "rt_scene_set_spheres")
        self._check(L.rt_scene_set_triangles(sc, ptr(A["triangles"]), len(A["triangles"])), "rt_scene_set_triangles")
        self._check(L.rt_scene_set_meshes(sc, ptr(A["meshes"]), len(A["meshes"])), "rt_scene_set_meshes")
        self._check(L.rt_scene_set_multimeshes(sc, ptr(A["multimeshes"]), len(A["multimeshes"])), "rt_scene_set_multimeshes")
        self._check(L.rt_scene_set_bvhs(sc, ptr(A["bvhs"]), len(A["bvhs"]), ptr(A["bvh_leaves"]), len(A["bvh_leaves"])), "rt_scene_set_bvhs")
        self._check(L.rt_scene_set_scene(sc, ptr(A["scene"]), len(A["scene"])), "rt_scene_set_scene")
        self._check(L.rt_scene_set_materials(sc, ptr(A["materials"]), len(A["materials"])), "rt_scene_set_materials")
        self._check(L.rt_scene_set_textures(sc, d.textures, d.num_textures), "rt_scene_set_textures")
        self._check(L.rt_scene_set_lights(sc, ptr(A["lights"]), len(A["lights"])), "rt_scene_set_lights")
        self._check(L.rt_scene_set_ambient(sc, C.byref(d.ambient)), "rt_scene_set_ambient")
        if builder is not None:
            self._check(L.rt_scene_set_builder(sc, int(builder)), "rt_scene_set_builder")
        self._check(L.rt_scene_commit(sc), "rt_scene_commit")
        self._flat = flat
        return self

    def builder_used(self):
        """(builder, seconds) of the last commit's triangle-BVH build."""
        b, t = C.c_int(), C.c_double()
        self._check(self.lib.rt_scene_builder_used(self.scene, C.byref(b), C.byref(t)), "rt_scene_builder_used")
        return b.value, t.value

    # -- render ------------------------------------------------------------------------------
    def render(self, camera, params, ids=False, stats=True, rgb=None):
        """Blocking render to host arrays.  Returns (rgb[H,W,3] f32, primid[H,W] u32|None, t|None, rt_stats)."""
        cam = camera.to_c() if hasattr(camera, "to_c") else camera
        H, W = params.height, params.width
        if rgb is None:
            rgb = np.zeros((H, W, 3), np.float32)
        prim = np.full((H, W), capi.RT_MISS, np.uint32) if ids else None
        t = np.zeros((H, W), np.float32) if ids else None
        st = capi.rt_stats()
        rc = self.lib.rt_render(self.scene, C.byref(cam), C.byref(params), rgb.ctypes.data,
                                prim.ctypes.data if ids else None, t.ctypes.data if ids else None,
                                C.byref(st) if stats else None)
        self._check(rc, "rt_render")
        return rgb, prim, t, st

    def render_device(self, camera, params, d_rgb, d_prim=None, d_t=None, stats=None, stream=None):
        """Queue a render into device memory (pointers as ints, e.g. torch tensor.data_ptr())."""
        cam = camera.to_c() if hasattr(camera, "to_c") else camera
        rc = self.lib.rt_render_device(self.scene, C.byref(cam), C.byref(params), d_rgb, d_prim, d_t,
                                       C.byref(stats) if stats is not None else None, stream)
        self._check(rc, "rt_render_device")

    def render_device_noalloc(self, camera, params, d_rgb, d_prim=None, d_t=None, stream=None):
        """rt_render_device_noalloc: the same render out of the library's fixed device scratch, no allocation."""
        cam = camera.to_c() if hasattr(camera, "to_c") else camera
        rc = self.lib.rt_render_device_noalloc(self.scene, C.byref(cam), C.byref(params), d_rgb, d_prim, d_t, stream)
        self._check(rc, "rt_render_device_noalloc")

    @staticmethod
    def render_multi(renderers, camera, params, rgb=None, stats=False):
        """rt_render_multi: one frame over several Renderers (one context each, the same scene uploaded to all):
        tiles t % N == i rendered by renderer i, gathered on renderers[0]'s GPU.  Returns (rgb, [rt_stats] | None)."""
        r0 = renderers[0]
        cam = camera.to_c() if hasattr(camera, "to_c") else camera
        H, W = params.height, params.width
        if rgb is None:
            rgb = np.zeros((H, W, 3), np.float32)
        arr = (C.c_void_p * len(renderers))(*[r.scene for r in renderers])
        st = (capi.rt_stats * len(renderers))() if stats else None
        rc = r0.lib.rt_render_multi(arr, len(renderers), C.byref(cam), C.byref(params), rgb.ctypes.data, st)
        r0._check(rc, "rt_render_multi")
        return rgb, (list(st) if stats else None)

    def owned_tiles(self, params, rank, world):
        n = C.c_uint32()
        self._check(self.lib.rt_owned_tiles(C.byref(params), rank, world, C.byref(n)), "rt_owned_tiles")
        return n.value

    def unpack_tiles(self, d_slab, params, rank, world, d_frame, stream=None):
        self._check(self.lib.rt_unpack_tiles(self.ctx, d_slab, C.byref(params), rank, world, d_frame, stream), "rt_unpack_tiles")

    def quantize_rgb8(self, d_rgb, num_pixels, d_rgb8, stream=None):
        self._check(self.lib.rt_quantize_rgb8(self.ctx, d_rgb, num_pixels, d_rgb8, stream), "rt_quantize_rgb8")

    def selftest_device_math(self, n, seed=1):
        """rt_selftest_device_math: mismatch counts [sqrt, sin, cos, r / RAND_MAX] over n pseudo-random inputs."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.rt_selftest_device_math(self.ctx, int(n), int(seed), out), "rt_selftest_device_math")
        return [int(v) for v in out]

    def bmp24_bytes(self, width, height):
        return int(self.lib.rt_bmp24_bytes(int(width), int(height)))

    def encode_bmp24(self, d_rgb, width, height, d_file, stream=None):
        """saveToFile on the device: f32 RGB frame -> complete 24-bit BMP file image (device pointers)."""
        self._check(self.lib.rt_encode_bmp24(self.ctx, d_rgb, int(width), int(height), d_file, stream), "rt_encode_bmp24")

    def close(self):
        if self.scene is not None:
            self.lib.rt_scene_destroy(self.scene)
            self.scene = None
        if self.ctx:
            self.lib.rt_shutdown(self.ctx)
            self.ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def save_bmp(filename, rgb):
    """saveToFile (reference main.cpp:1837-1857): truncating 8-bit conversion, 24-bit BMP, row 0 on top.
    File layout follows stb_image_write's stbi_write_bmp (BGR, bottom-up rows padded to 4 bytes)."""
    rgb = np.asarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    data = (rgb * np.float32(255)).astype(np.uint8)
    pad = (-w * 3) % 4
    rows = data[::-1, :, ::-1]
    body = b"".join(r.tobytes() + b"\0" * pad for r in rows)
    import struct
    hdr = b"BM" + struct.pack("<IHHI", 14 + 40 + len(body), 0, 0, 54)
    hdr += struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, 0, 0, 0, 0, 0)
    with open(filename, "wb") as f:
        f.write(hdr + body)
