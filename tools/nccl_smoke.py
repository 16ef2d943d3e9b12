#!/usr/bin/env python3
"""tools/nccl_smoke.py — what a one-GPU box can show of bench.py's RCCL branch: torch.distributed with backend
"nccl" (= RCCL) at world size 1 — process-group creation on the device, an all-reduce, and dist.gather (the call
raytracer_amd.dist.gather_frame makes) of a frame rendered on the same stream just before.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port P tools/nccl_smoke.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import util  # noqa: E402
from raytracer_amd import capi  # noqa: E402
from raytracer_amd.render import Renderer  # noqa: E402

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
name, w, params = util.small_scenes()[0]
r = Renderer(0)
r.upload(w.flatten())
full, _, _, _ = r.render(w.camera, params)
p = capi.rt_params.from_buffer_copy(params)
st = torch.cuda.current_stream().cuda_stream
# at world size 1 the library writes a frame, not a slab (slabs need tile_world > 1): the gathered tensor is the frame
out = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device=dev)
r.render_device(w.camera, p, out.data_ptr(), None, None, None, st)
gl = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
dist.gather(out, gl, dst=0)   # raytracer_amd.dist.gather_frame skips the call at world size 1; here it runs
frame = gl[0]
dist.barrier()
torch.cuda.synchronize()
ok = np.array_equal(frame.cpu().numpy().view(np.uint32), full.view(np.uint32))
print("nccl smoke:", "OK" if ok else "MISMATCH", "backend", dist.get_backend(), "world", world)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
