"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/rt_amd.h
declares, struct layouts agree with the ctypes mirror, and rt_init fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import conftest
from raytracer_amd import capi

HEADER = os.path.join(conftest.ROOT, "include", "rt_amd.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        subprocess.run(["make", "-C", os.path.dirname(capi.LIB_PATH), "librt_amd.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return capi.load_library()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rt_amd.h but not exported by librt_amd.so"
    assert sorted(capi.exported_symbols()) == names


def test_struct_layouts_match_the_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    src = tmp_path / "layout.c"
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "rt_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(rt_plane), sizeof(rt_sphere), sizeof(rt_triangle),
         sizeof(rt_material), sizeof(rt_light), sizeof(rt_texture), sizeof(rt_ambient), sizeof(rt_camera),
         sizeof(rt_scene_desc), sizeof(rt_params), sizeof(rt_stats), offsetof(rt_stats, ms_kernel));
  return 0; }
''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [capi.plane_dt.itemsize, capi.sphere_dt.itemsize, capi.triangle_dt.itemsize, capi.material_dt.itemsize,
            capi.light_dt.itemsize, C.sizeof(capi.rt_texture), C.sizeof(capi.rt_ambient), C.sizeof(capi.rt_camera),
            C.sizeof(capi.rt_scene_desc), C.sizeof(capi.rt_params), C.sizeof(capi.rt_stats), capi.rt_stats.ms_kernel.offset]
    assert got == want


def test_defaults_are_the_reference_constants(lib):
    p = capi.rt_params()
    lib.rt_params_default(C.byref(p))
    assert (p.width, p.height, p.num_samples, p.num_sets, p.sampler) == (1000, 1000, 4, 83, capi.RT_SAMPLER_JITTER)
    assert p.max_depth == 3 and p.light_attenuation == 500000.0
    assert lib.rt_version() == 1


def test_no_gpu_means_a_loud_error_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = lib.rt_init(0, C.byref(ctx))
    assert rc < 0 and not ctx.value
    assert b"no CPU fallback" in lib.rt_last_error(None) or b"HIP" in lib.rt_last_error(None)


def test_product_never_touches_the_oracle():
    """Nothing under raytracer_amd/ may import, link or name the oracle."""
    pkg = os.path.join(conftest.ROOT, "raytracer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle.bindings" not in text and "from oracle" not in text, f
                assert "libemul" not in text, f
