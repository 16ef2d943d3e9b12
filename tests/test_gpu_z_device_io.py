"""Device I/O plumbing of the C ABI (SURVEY 8f N2 and the multi-GPU slab path): rt_render_device into
torch-owned memory, tile slabs + rt_unpack_tiles, rt_quantize_rgb8, rt_encode_bmp24, and the stream contract of
include/rt_amd.h.  Sorts after tests/test_gpu_parity.py on purpose (see that file's docstring).

The BMP encoder is pinned to bytes the REFERENCE wrote: tests/golden/bmp_savetofile.npz holds float images and the
files the reference's own saveToFile (main.cpp:1837-1857 -> stbi_write_bmp) produced for them
(tests/golden/make_golden.py bmp)."""
import os

import numpy as np
import pytest

import util
from raytracer_amd import capi

pytestmark = pytest.mark.gpu

CASES = util.small_scenes()


@pytest.mark.parametrize("mode", ["current", "side", "null"])
def test_encode_bmp24_equals_the_reference_written_file(gpu, mode):
    import torch
    g = np.load(os.path.join(util.GOLDEN, "bmp_savetofile.npz"))
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream() if mode == "side" else None
    for name in ("200x24", "101x47", "4x4", "3x5"):
        rgb = g["rgb_" + name]
        want = g["bmp_" + name].tobytes()
        h, w = rgb.shape[:2]
        assert gpu.bmp24_bytes(w, h) == len(want)
        ctx = torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            # the fills and the upload are queued, not complete, when the encoder is called
            buf = torch.full((len(want),), 0xAA, dtype=torch.uint8, device=dev)
            src = torch.from_numpy(rgb).to(dev, non_blocking=False).contiguous()
            junk = torch.zeros((1 << 22,), dtype=torch.float32, device=dev).add_(1.0)   # keeps the stream busy
            stream = None if mode == "null" else torch.cuda.current_stream().cuda_stream
            gpu.encode_bmp24(src.data_ptr(), w, h, buf.data_ptr(), stream)
            out = buf.cpu()   # on the same torch stream: ordered behind the encoder (stream mode) / after its wait (NULL)
        torch.cuda.synchronize()
        assert out.numpy().tobytes() == want, f"{name} ({mode} stream)"
        del junk


def test_quantize_equals_the_reference_conversion(gpu):
    """rt_quantize_rgb8 against the pixel bytes of the reference-written BMPs (BGR, bottom-up -> RGB, top-down)."""
    import torch
    g = np.load(os.path.join(util.GOLDEN, "bmp_savetofile.npz"))
    dev = torch.device("cuda", 0)
    for name in ("200x24", "101x47"):
        rgb = g["rgb_" + name]
        h, w = rgb.shape[:2]
        row = (w * 3 + 3) // 4 * 4
        want = g["bmp_" + name][54:].reshape(h, row)[::-1, : w * 3].reshape(h, w, 3)[:, :, ::-1]
        src = torch.from_numpy(rgb).to(dev)
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        gpu.quantize_rgb8(src.data_ptr(), h * w, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(out.cpu().numpy(), want)
        assert np.array_equal(util.quantize(rgb), want)   # the helper the parity tests use is the same conversion


@pytest.mark.parametrize("mode", ["current", "null"])
def test_device_outputs_slabs_unpack(gpu, mode):
    """rt_render_device into torch-owned device memory, tile slabs of two ranks scattered with rt_unpack_tiles:
    the frame equals rt_render's, with the work queued on torch's stream and through the NULL-stream path."""
    import torch
    name, world, params = CASES[0]
    p = capi.rt_params.from_buffer_copy(params)
    p.width, p.height, p.x1, p.y1 = 200, 136, 200, 136
    gpu.upload(world)
    full, _, _, _ = gpu.render(world.camera, p)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream if mode == "current" else None
    frame = torch.full((136, 200, 3), -1.0, dtype=torch.float32, device=dev)
    for r in range(2):
        q = capi.rt_params.from_buffer_copy(p)
        q.tile_rank, q.tile_world = r, 2
        n = gpu.owned_tiles(q, r, 2)
        slab = torch.full((n, 64 * 64, 3), -2.0, dtype=torch.float32, device=dev)
        gpu.render_device(world.camera, q, slab.data_ptr(), None, None, None, stream)
        gpu.unpack_tiles(slab.data_ptr(), q, r, 2, frame.data_ptr(), stream)
    got = frame.cpu().numpy()
    torch.cuda.synchronize()
    assert util.bits_equal(got, full)


def test_renders_on_alternating_streams_do_not_interfere(gpu):
    """One context carries one render at a time (include/rt_amd.h): renders queued back to back on two different
    streams share the context's ray queues, so the second must wait for the first on the device."""
    import torch
    name, world, params = CASES[2]
    gpu.upload(world)
    want, _, _, _ = gpu.render(world.camera, params)
    dev = torch.device("cuda", 0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    H, W = params.height, params.width
    outs = []
    for k in range(6):
        s = s1 if k % 2 == 0 else s2
        with torch.cuda.stream(s):
            o = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
            gpu.render_device(world.camera, params, o.data_ptr(), None, None, None, s.cuda_stream)
            outs.append(o)
    torch.cuda.synchronize()
    for o in outs:
        assert util.bits_equal(o.cpu().numpy(), want)


def test_device_arithmetic_of_the_ao_direction_is_exact(gpu):
    """The hand-written device forms behind Ambient::L's sample directions (rt_core.hpp: sqrt_not_tiny, the scalar-addend fma
    steps of sincos_f64, div_rand_max) against the plain formulations the compiler expands itself, bit for bit, on the
    device, over 2^27 pseudo-random inputs each (two seeds)."""
    for seed in (1, 0x5DEECE66D):
        assert gpu.selftest_device_math(1 << 27, seed) == [0, 0, 0, 0]
