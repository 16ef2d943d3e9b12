"""ctypes bindings of the two test oracles.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by raytracer_amd/.

  RefOracle   oracle/_ref/libref_oracle.so : the real reference compiled from /root/reference
  PortOracle  oracle/liboracle.so          : our plain-C restatement (oracle/rt_oracle.c)
Both render an rt_scene_desc (raytracer_amd.scene.World.flatten()) on the CPU.
"""
import ctypes as C
import os
import tempfile

import numpy as np

from raytracer_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(_HERE, "_ref", "libref_oracle.so")
PORT_LIB = os.path.join(_HERE, "liboracle.so")


class ref_options(C.Structure):
    _fields_ = [("rng_mode", C.c_int32), ("nthreads", C.c_int32), ("shade", C.c_int32), ("tmpdir", C.c_char_p)]


class ref_outputs(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("primid", C.c_void_p), ("t", C.c_void_p), ("normal", C.c_void_p),
                ("uv", C.c_void_p), ("rays", C.c_uint64), ("seconds", C.c_double)]


class Result:
    pass


def _alloc(params, ids):
    W, H = params.width, params.height
    r = Result()
    r.rgb = np.zeros((H, W, 3), np.float32)
    r.primid = np.full((H, W), capi.RT_MISS, np.uint32) if ids else None
    r.t = np.zeros((H, W), np.float32) if ids else None
    r.normal = np.zeros((H, W, 3), np.float32) if ids else None
    r.uv = np.zeros((H, W, 2), np.float32) if ids else None
    return r


def _fill(o, r):
    o.rgb = r.rgb.ctypes.data
    o.primid = r.primid.ctypes.data if r.primid is not None else None
    o.t = r.t.ctypes.data if r.t is not None else None
    o.normal = r.normal.ctypes.data if r.normal is not None else None
    o.uv = r.uv.ctypes.data if r.uv is not None else None


class _Base:
    prefix = None
    path = None

    def __init__(self):
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path)
        self.lib = C.CDLL(self.path)
        self._tmp = tempfile.mkdtemp(prefix="rt_oracle_")
        fn = getattr(self.lib, self.prefix + "_render")
        fn.argtypes = [C.POINTER(capi.rt_scene_desc), C.POINTER(capi.rt_camera), C.POINTER(capi.rt_params),
                       C.POINTER(ref_options), C.POINTER(ref_outputs)]
        self._render = fn
        le = getattr(self.lib, self.prefix + "_last_error")
        le.restype = C.c_char_p
        self._last_error = le

    @classmethod
    def available(cls):
        return os.path.exists(cls.path)

    def render(self, flat, camera, params, rng_mode=1, nthreads=1, shade=True, ids=True):
        """rng_mode 0: libc rand + the reference Sampler; 1: counter hash (include/rt_rng.h) for rand()
        and for pixel sample points; 2: rand() == RAND_MAX/2 + the reference Sampler."""
        r = _alloc(params, ids)
        opt = ref_options(rng_mode=rng_mode, nthreads=nthreads, shade=1 if shade else 0,
                          tmpdir=self._tmp.encode())
        out = ref_outputs()
        _fill(out, r)
        cam = camera.to_c() if hasattr(camera, "to_c") else camera
        rc = self._render(C.byref(flat.desc), C.byref(cam), C.byref(params), C.byref(opt), C.byref(out))
        if rc != 0:
            raise RuntimeError(f"{self.prefix}_render failed: {self._last_error().decode()}")
        r.rays = out.rays
        r.seconds = out.seconds
        return r


class RefOracle(_Base):
    prefix = "ref"
    path = REF_LIB

    def __init__(self):
        super().__init__()
        L = self.lib
        fp = C.POINTER(C.c_float)
        L.ref_geq.argtypes = [C.c_float, C.c_float]
        L.ref_sample_hemisphere.argtypes = [C.c_float, C.c_float, fp]
        L.ref_rotate_to_space.argtypes = [fp, fp, fp]
        L.ref_camera_ray.argtypes = [C.POINTER(capi.rt_camera), C.c_float, C.c_float, fp, fp]
        L.ref_box_intersects.argtypes = [fp, C.c_int, fp, fp]
        L.ref_specular_fn.argtypes = [C.c_float, C.c_int, fp, fp, fp]
        L.ref_specular_fn.restype = C.c_float
        L.ref_refraction.argtypes = [C.c_float, C.c_float, fp, fp, fp, fp]
        L.ref_load_texture.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
        L.ref_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.ref_bvh_dump.argtypes = [C.POINTER(capi.rt_scene_desc), C.POINTER(capi.rt_params), C.c_char_p,
                                   C.c_uint32, C.c_void_p, C.c_int]
        L.ref_save_bmp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
        L.ref_bbox_points.argtypes = [C.POINTER(capi.rt_scene_desc), C.POINTER(capi.rt_params), C.c_char_p, C.c_int, C.c_uint32, fp]

    def save_bmp_bytes(self, rgb):
        """The bytes the reference's saveToFile (main.cpp:1837-1857, stbi_write_bmp) writes for rgb[H,W,3]."""
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w = rgb.shape[:2]
        path = os.path.join(self._tmp, "ref_save.bmp")
        if self.lib.ref_save_bmp(rgb.ctypes.data, w, h, path.encode()) != 0:
            raise RuntimeError("ref_save_bmp failed")
        with open(path, "rb") as f:
            return f.read()

    def bbox_points(self, flat, params, kind, index):
        """The reference's getBoundingBoxPoints() of one shape: float32 [6, 3]."""
        out = np.zeros((6, 3), np.float32)
        n = self.lib.ref_bbox_points(C.byref(flat.desc), C.byref(params), self._tmp.encode(), kind, index,
                                     out.ctypes.data_as(C.POINTER(C.c_float)))
        if n != 6:
            raise RuntimeError("ref_bbox_points failed")
        return out

    def bvh_dump(self, flat, params, which=0, cap=4096):
        out = np.zeros((cap, 9), np.float32)
        n = self.lib.ref_bvh_dump(C.byref(flat.desc), C.byref(params), self._tmp.encode(), which,
                                  out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("ref_bvh_dump failed")
        return out[:n]

    def load_texture(self, path):
        w, h = C.c_int(), C.c_int()
        if self.lib.ref_load_texture(path.encode(), C.byref(w), C.byref(h), None, 0) != 0:
            raise RuntimeError("stbi_load failed: " + path)
        out = np.zeros((h.value, w.value, 4), np.uint8)
        self.lib.ref_load_texture(path.encode(), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes)
        return out


class PortOracle(_Base):
    prefix = "oracle"
    path = PORT_LIB

    def __init__(self):
        super().__init__()
        L = self.lib
        fp = C.POINTER(C.c_float)
        L.oracle_geq.argtypes = [C.c_float, C.c_float]
        L.oracle_sample_hemisphere.argtypes = [C.c_float, C.c_float, fp]
        L.oracle_rotate_to_space.argtypes = [fp, fp, fp]
        L.oracle_camera_ray.argtypes = [C.POINTER(capi.rt_camera), C.c_float, C.c_float, fp, fp]
        L.oracle_box_intersects.argtypes = [fp, C.c_int, fp, fp]
        L.oracle_specular_fn.argtypes = [C.c_float, C.c_int, fp, fp, fp]
        L.oracle_specular_fn.restype = C.c_float
        L.oracle_refraction.argtypes = [C.c_float, C.c_float, fp, fp, fp, fp]
        L.oracle_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_bvh_dump.argtypes = [C.POINTER(capi.rt_scene_desc), C.c_uint32, C.c_void_p, C.c_int]

    def bvh_dump(self, flat, params=None, which=0, cap=4096):
        out = np.zeros((cap, 9), np.float32)
        n = self.lib.oracle_bvh_dump(C.byref(flat.desc), which, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("oracle_bvh_dump failed")
        return out[:n]


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def quantize(rgb):
    """saveToFile's truncating float->8 bit conversion (main.cpp:1847-1849)."""
    return (np.asarray(rgb, np.float32) * np.float32(255)).astype(np.uint8)
