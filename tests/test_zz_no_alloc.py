"""rt_render_device_noalloc (include/rt_amd.h): the allocation-free variant of rt_render_device, whose pipeline
scratch is a fixed array inside the library's device image.  Through the C ABI, against rt_render_device's output,
bit for bit, on two calls in a row — the second with a smaller input — plus the size limit's error.

Collected last on purpose (`pytest -x` at round end): a failure here hides nothing behind it.  Passes alone in a fresh
process and after the other tests."""
import numpy as np
import pytest

import util
from raytracer_amd import capi

pytestmark = pytest.mark.gpu


def _both(gpu, world, params):
    """(frame, primid, t) of rt_render_device and of rt_render_device_noalloc for the same call."""
    import torch
    dev = torch.device("cuda", 0)
    H, W = params.height, params.width
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for noalloc in (False, True):
        rgb = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
        prim = torch.full((H, W), 7, dtype=torch.int32, device=dev)
        t = torch.full((H, W), -1.0, dtype=torch.float32, device=dev)
        if noalloc:
            torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info()
            gpu.render_device_noalloc(world.camera, params, rgb.data_ptr(), prim.data_ptr(), t.data_ptr(), stream)
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info()
            assert free1 == free0, "the allocation-free entry point changed the device's free memory"
        else:
            gpu.render_device(world.camera, params, rgb.data_ptr(), prim.data_ptr(), t.data_ptr(), None, stream)
        torch.cuda.synchronize()
        outs.append((rgb.cpu().numpy(), prim.cpu().numpy().view(np.uint32), t.cpu().numpy()))
    return outs


def test_noalloc_render_equals_render_device_twice_in_a_row(gpu):
    from raytracer_amd import scenes, scene as S
    # first call: several batches out of the fixed scratch (16 spp + 4x4 AO + bounce levels: 2 tiles per batch)
    big = scenes.default_scene(320, 200, sampler=S.Jitter(16, 83), ao_samples=4)
    # second call, smaller input: fewer pixels, fewer samples, no bounce materials, a textured mesh scene
    small = scenes.config3_scene(96, 64, spp=4, mesh_nu=48, mesh_nv=24)
    for world, seed in ((big, 3), (small, 9)):
        gpu.upload(world)
        params = world.params(seed=seed)
        (rgb0, prim0, t0), (rgb1, prim1, t1) = _both(gpu, world, params)
        assert (prim0 != capi.RT_MISS).any() and rgb0.max() > 0.0
        assert np.array_equal(prim1, prim0)
        assert util.bits_equal(t1, t0)
        assert util.bits_equal(rgb1, rgb0)


def test_noalloc_tile_slabs_and_null_stream(gpu):
    """tile_world > 1 writes compact slabs like rt_render_device; stream = NULL waits for completion."""
    import torch
    name, world, params = util.small_scenes()[2]
    gpu.upload(world)
    dev = torch.device("cuda", 0)
    q = capi.rt_params.from_buffer_copy(params)
    q.tile_rank, q.tile_world = 1, 2
    n = gpu.owned_tiles(q, 1, 2)
    a = torch.zeros((n, 64 * 64, 3), dtype=torch.float32, device=dev)
    b = torch.zeros_like(a)
    gpu.render_device(world.camera, q, a.data_ptr(), None, None, None, None)
    gpu.render_device_noalloc(world.camera, q, b.data_ptr(), None, None, None)
    assert util.bits_equal(b.cpu().numpy(), a.cpu().numpy())


def test_noalloc_rejects_what_does_not_fit(gpu):
    from raytracer_amd import scenes, scene as S
    from raytracer_amd.render import RtError
    import torch
    world = scenes.default_scene(64, 64, sampler=S.Jitter(256, 83), ao_samples=4)   # one tile = 1 Mi paths: ~1.7 GB of queues
    gpu.upload(world)
    out = torch.zeros((64, 64, 3), dtype=torch.float32, device=torch.device("cuda", 0))
    with pytest.raises(RtError, match="fixed scratch holds"):
        gpu.render_device_noalloc(world.camera, world.params(), out.data_ptr())
    p = world.params()
    p.num_samples = 4
    p.flags = capi.RT_FLAG_TIMING
    with pytest.raises(RtError, match="not available"):
        gpu.render_device_noalloc(world.camera, p, out.data_ptr())
    p.flags = 0
    gpu.render_device_noalloc(world.camera, p, out.data_ptr())   # and the context still works afterwards
    assert float(out.max()) > 0.0


def test_noalloc_under_graph_capture(gpu):
    """The use case the header names: the call captured into a HIP graph (kernel and memset nodes only: no event is
    waited on or recorded while the stream captures), replayed twice, against rt_render_device."""
    import torch
    name, world, params = util.small_scenes()[2]
    gpu.upload(world)
    dev = torch.device("cuda", 0)
    H, W = params.height, params.width
    want = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    gpu.render_device(world.camera, params, want.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
    out = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
    gpu.render_device_noalloc(world.camera, params, out.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)   # warm-up outside the capture
    torch.cuda.synchronize()
    assert util.bits_equal(out.cpu().numpy(), want.cpu().numpy())
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        gpu.render_device_noalloc(world.camera, params, out.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        out.fill_(-1.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert util.bits_equal(out.cpu().numpy(), want.cpu().numpy())
    # and the context still renders eagerly afterwards, on another stream
    out.fill_(-1.0)
    gpu.render_device_noalloc(world.camera, params, out.data_ptr(), None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert util.bits_equal(out.cpu().numpy(), want.cpu().numpy())


def test_noalloc_calls_through_two_contexts_of_one_gpu_serialise(gpu):
    """Every context of a GPU shares that GPU's scratch array: calls through different contexts on different streams
    are ordered by the library (one 'last user' event per device), so both frames come out right."""
    import torch
    from raytracer_amd.render import Renderer
    name, world, params = util.small_scenes()[2]
    flat = world.flatten()
    gpu.upload(flat)
    dev = torch.device("cuda", 0)
    H, W = params.height, params.width
    want = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    gpu.render_device(world.camera, params, want.data_ptr(), None, None, None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    other = Renderer(0)
    try:
        other.upload(flat)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        a = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
        b = torch.full((H, W, 3), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            gpu.render_device_noalloc(world.camera, params, a.data_ptr(), None, None, s1.cuda_stream)
            other.render_device_noalloc(world.camera, params, b.data_ptr(), None, None, s2.cuda_stream)
        torch.cuda.synchronize()
        assert util.bits_equal(a.cpu().numpy(), want.cpu().numpy())
        assert util.bits_equal(b.cpu().numpy(), want.cpu().numpy())
    finally:
        other.close()
