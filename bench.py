#!/usr/bin/env python3
"""bench.py — Mrays/s of the render path on config 3 (4K, 16 jittered spp + 4x4 AO, textured BVH mesh).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one full frame.  With N > 1 every rank (one process per GPU) renders the 64x64 tiles
t with t % N == rank into a compact slab; slabs are gathered to rank 0 over RCCL and scattered
into the frame there.  Total work is fixed as N grows (strong scaling).  Rank 0 prints one JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# Vector-ALU peak in lane-instructions per second: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles and SIMD
# (the SQ counters of profiles/*_pmc.csv: SQ_ACTIVE_INST_VALU = one quad-cycle per instruction), 64 lanes, 2.4 GHz.
VALU_PEAK_GLANES = 256 * 4 * 64 / 4 * 2.4   # = 39 321.6 G lane-instructions / s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=(3, 4, 5),
                    help="BASELINE.json config: 3 (default, the headline: 4K, 16 spp + AO16, one textured BVH mesh), 4 (mirror + "
                         "transparency scene, bounce depth 8, 3840x2160, 64 spp: the 2/4/8-GPU config) or 5 (10.2 M triangles in 2500 "
                         "instanced meshes behind the reference's shape-level BVH; default size 1920x1080, 4 spp)")
    ap.add_argument("--force-collective", action="store_true", help="run the gather collective at world size 1 too (RCCL code path on a one-GPU box)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--instances", type=int, default=2500, help="config 5: instanced meshes of 4096 triangles")
    ap.add_argument("--mesh-nu", type=int, default=700)
    ap.add_argument("--mesh-nv", type=int, default=360)
    ap.add_argument("--lib", default=None, help="alternative librt_amd.so (kernel tuning builds)")
    ap.add_argument("--batch-paths", type=int, default=0, help="paths per wavefront batch (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL over xGMI) or gloo (rehearsal: ranks may share one GPU, slabs staged through the host)")
    ap.add_argument("--cpu-crop", type=int, default=None, help="edge of the centre crop timed on the CPU (default 96; config 5: 24)")
    ap.add_argument("--cpu-baseline", action="store_true", help="config 5 only: time the reference too (it builds 10 M Triangle objects: ~1 min, ~3 GB)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracer_amd import capi, scenes
    from raytracer_amd import dist as rdist
    from raytracer_amd.render import Renderer

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    distributed = world_size > 1 or args.force_collective
    gpu_index = local_rank % max(1, torch.cuda.device_count()) if args.backend == "gloo" else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    if distributed:
        if world_size == 1:   # --force-collective without a launcher
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world_size)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)

    if args.config == 3:
        args.width, args.height, args.spp = args.width or 3840, args.height or 2160, args.spp or 16
        world = scenes.config3_scene(args.width, args.height, args.spp, args.mesh_nu, args.mesh_nv)
    elif args.config == 4:
        args.width, args.height, args.spp = args.width or 3840, args.height or 2160, args.spp or 64
        world = scenes.config4_scene(args.width, args.height, args.spp)
    else:
        args.width, args.height, args.spp = args.width or 1920, args.height or 1080, args.spp or 4
        world = scenes.config5_scene(args.width, args.height, args.spp, instances=args.instances)
    flat = world.flatten()
    ntri = int(flat.arrays["triangles"].shape[0])
    r = Renderer(gpu_index, args.lib)
    r.upload(flat)
    params = world.params(seed=1, max_depth=8) if args.config == 4 else world.params(seed=1)   # BASELINE.json states config 4 at bounce depth 8
    params.tile_rank, params.tile_world = rank, world_size
    params.batch_paths = args.batch_paths
    W, H = args.width, args.height

    owned = r.owned_tiles(params, rank, world_size)
    max_owned = r.owned_tiles(params, 0, world_size)
    if distributed:
        # TWO slabs (and two receive lists on rank 0): frame k+1 renders into the other slab while frame k's slabs travel
        # to rank 0 and are scattered there; the collective runs on the backend's own stream (async_op), the render and
        # the unpack on torch's current stream, which waits for a gather only when it is about to unpack it.
        slabs = [torch.zeros((max_owned, capi.RT_TILE * capi.RT_TILE, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        on_gpu = args.backend == "nccl"
        gather_lists = [[torch.zeros_like(slabs[0]) if on_gpu else torch.zeros_like(slabs[0], device="cpu") for _ in range(world_size)] if rank == 0 else None
                        for _ in range(2)]
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if rank == 0 else None
    else:
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    state = {"k": 0, "pending": None}

    def unpack(src, t):
        # on torch's current stream: ordered behind the gather that filled `t` (PendingGather.finish waited for it)
        if not t.is_cuda:
            t = t.to(dev)
        if world_size == 1:   # --force-collective rehearsal: with tile_world == 1 the library writes a row-major frame, not a tile slab (include/rt_amd.h)
            frame.view(-1).copy_(t.view(-1)[:frame.numel()])
            return
        r.unpack_tiles(t.data_ptr(), params, src, world_size, frame.data_ptr(), torch.cuda.current_stream().cuda_stream)

    def flush():
        if state["pending"] is not None:
            state["pending"].finish()
            state["pending"] = None

    def step(flags=0, stats=None):
        params.flags = flags
        k = state["k"] % 2
        state["k"] += 1
        out = slabs[k] if distributed else frame
        # on torch's current stream: ordered behind the allocation fills above and behind the unpack of the frame before
        # last, whose gather read this slab (include/rt_amd.h, stream contract)
        r.render_device(world.camera, params, out.data_ptr(), None, None, stats, torch.cuda.current_stream().cuda_stream)
        if distributed:   # the one exchange step: slabs -> rank 0 (RCCL over xGMI), scattered into the frame there
            src = slabs[k] if on_gpu else slabs[k].cpu()   # gloo rehearsal: staged through the host
            nxt = rdist.gather_frame_start(src, params, rank, world_size, unpack, gather_list=gather_lists[k], force_collective=True)
            flush()   # the previous frame's gather has had this frame's render to hide behind
            state["pending"] = nxt

    def sync():
        flush()
        torch.cuda.synchronize()
        if distributed and world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed: ray counts and traversal counters (deterministic per frame), also the first warm-up
    st_count = capi.rt_stats()
    step(capi.RT_FLAG_STATS, st_count)
    for _ in range(max(0, args.warmup - 1)):
        step()
    rays_rank = st_count.rays

    kernel_ms = np.zeros(capi.RT_K_COUNT)
    launches = np.zeros(capi.RT_K_COUNT)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = capi.rt_stats()
        step(capi.RT_FLAG_TIMING, st)
        kernel_ms += np.array(list(st.ms_kernel))
        launches += np.array(list(st.launches))
    sync()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([elapsed, float(rays_rank), float(st_count.rays_skipped)], dtype=torch.float64, device=dev)
    if distributed and args.backend != "nccl":
        tot = tot.cpu()
    if distributed:
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0]); rays_total = float(tsum[1]); skipped_total = float(tsum[2])
    else:
        rays_total = float(rays_rank); skipped_total = float(st_count.rays_skipped)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / (ms_per_step * 1e-3) / 1e6
        traffic_all = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic_all = json.load(open(tpath)).get(f"config{args.config}", {})
            except Exception:
                traffic_all = {}
        ao_rays = float(st_count.rays_ao)
        if args.config in (3, 4):
            # roofline of the dominant kernel = the compacted triangle-BVH traversal of the ambient-occlusion rays
            # (k_bvh<AO>), rank 0's launches.  Algorithmic bytes per ray it processes: B = 96 + 64*V + 48*T (SURVEY.md
            # 8d: ray + hit records, V 64-byte four-wide nodes, T 48-byte triangles; V and T come from the counting pass
            # and count cache hits too).
            kname, k = "k_bvh<AO>", capi.RT_K_NAMES.index("ao_bvh")
            if args.config == 4 and kernel_ms[capi.RT_K_NAMES.index("ao")] > kernel_ms[k]:
                kname, k = "k_ao", capi.RT_K_NAMES.index("ao")   # the reference's own scene: the gate walk dominates
            units = float(st_count.bvh_rays[2]) if kname == "k_bvh<AO>" else ao_rays
            V = st_count.nodes[2] / max(1.0, units)
            T = st_count.tris[2] / max(1.0, units)
            G = st_count.gates[2] / max(1.0, ao_rays)
            b_ray = 96.0 + 64.0 * V + 48.0 * T if kname == "k_bvh<AO>" else 48.0 + 48.0 * G + 64.0 * st_count.bvh_rays[2] / max(1.0, ao_rays) + 4.0
            extra = {"nodes_per_ray": round(V, 2), "tris_per_ray": round(T, 2), "ao_rays_total": int(ao_rays), "ao_rays_needing_bvh": int(st_count.bvh_rays[2])}
        else:
            # config 5: the gate walk of the AO rays (k_ao_walk) dominates: per ray the 48-byte AO frame, G 48-byte gate
            # nodes of the reference's shape-level hierarchy (G from the counting pass, LDS-staged ones included),
            # one 64-byte queue item per ray that reaches a mesh, 4 bytes of result.
            kname, k = "k_ao_walk", capi.RT_K_NAMES.index("ao")
            units = ao_rays
            G = st_count.gates[2] / max(1.0, units)
            share = st_count.bvh_rays[2] / max(1.0, units)
            b_ray = 48.0 + 48.0 * G + 64.0 * share + 4.0
            extra = {"gates_per_ray": round(G, 2), "queued_share": round(share, 3), "ao_rays_total": int(ao_rays)}
        k_ms = kernel_ms[k] / args.steps
        k_launches = launches[k] / args.steps / (2.0 if k == capi.RT_K_NAMES.index("ao") else 1.0)   # the 'ao' class also times the (empty) overflow launch
        launch_s = (k_ms / max(1.0, k_launches)) * 1e-3
        units_launch = units / max(1.0, k_launches)
        algorithmic = (units * b_ray) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # What binds is read from the counters, not assumed (profiles/traffic.json, written by tools/summarize_profile.py
        # from separate rocprofv3 --pmc passes over this same command): per unit (ray) of the dominant kernel the HBM bytes
        # (FETCH_SIZE corrected by the factor calibrated on this access pattern, tools/fetch_calib.hip, + WRITE_SIZE) and
        # the vector-ALU lane-instructions (SQ_THREAD_CYCLES_VALU: active lanes summed over the VALU instructions).  Both
        # are properties of the code on this workload; the durations they are divided by are measured live here with
        # HIP events around every launch (RT_FLAG_TIMING).
        same_kernel = traffic_all.get("kernel", "").startswith(kname.split("<")[0])
        hbm_b = traffic_all.get("hbm_bytes_per_unit") if same_kernel else None
        lanes = traffic_all.get("valu_lane_insts_per_unit") if same_kernel else None
        bound = traffic_all.get("bound", "valu") if same_kernel else "valu"
        traffic = hbm_b * units_launch if hbm_b else None
        hbm_gbs = traffic / launch_s / 1e9 if traffic and launch_s > 0 else None
        valu_g = lanes * units_launch / launch_s / 1e9 if lanes and launch_s > 0 else None
        roofline = {"kernel": kname, "bound": bound}
        if bound == "hbm" and hbm_gbs is not None:
            roofline.update({"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4)})
        elif bound == "valu" and valu_g is not None:
            # useful vector-ALU lane-slots: lanes that did work / lane-slots the SIMDs could have issued in the launch time
            roofline.update({"achieved": round(valu_g, 1), "peak": round(VALU_PEAK_GLANES, 1), "unit": "Glane-instr/s", "frac": round(valu_g / VALU_PEAK_GLANES, 4)})
        else:   # another unit binds (e.g. "ta/l1"): its busy share as measured under rocprofv3
            roofline.update({"achieved": traffic_all.get("bound_busy") if same_kernel else None, "peak": 1.0, "unit": "busy share (profiled run)",
                             "frac": traffic_all.get("bound_busy") if same_kernel else None})
        roofline.update({
            "traffic": round(traffic, 1) if traffic else None,
            "hbm_gbs": round(hbm_gbs, 1) if hbm_gbs else None, "hbm_frac": round(hbm_gbs / HBM_PEAK_GBS, 4) if hbm_gbs else None,
            "valu_glane_insts_per_s": round(valu_g, 1) if valu_g else None, "valu_frac": round(valu_g / VALU_PEAK_GLANES, 4) if valu_g else None,
            "valu_issue_busy_profiled": traffic_all.get("valu_busy") if same_kernel else None,
            "valu_active_lanes_profiled": traffic_all.get("valu_active_lanes") if same_kernel else None,
            "fetch_size_factor": traffic_all.get("fetch_factor") if same_kernel else None,
            # SURVEY 8d's figure kept beside it: ALGORITHMIC bytes (cache hits count) over the launch time; caches serve most
            # of them, so it is not a bound and may exceed the HBM peak
            "algorithmic_gbs": round(algorithmic, 1), "algorithmic_frac": round(algorithmic / HBM_PEAK_GBS, 4), "bytes_per_ray": round(b_ray, 1),
            "rays_per_launch": round(units_launch, 1), "avg_launch_ms": round(k_ms / max(1.0, k_launches), 4),
            "mrays_per_s_in_kernel": round(units / (k_ms * 1e-3) / 1e6, 1) if k_ms > 0 else None,
            "counters_from": traffic_all.get("tag") if same_kernel else None,
        })
        roofline.update(extra)
        torch.cuda.synchronize()
        checksum = int(frame.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF   # same for every N
        workload = {
            3: f"config3: {W}x{H}, {args.spp} jittered spp + 4x4 ambient occlusion, room + textured {ntri}-triangle BVH mesh, 1 point light",
            4: f"config4: {W}x{H}, {args.spp} jittered spp + 4x4 ambient occlusion, mirror + transparency scene, bounce depth {params.max_depth}, {ntri} triangles, 1 point light",
            5: (f"config5: {W}x{H}, {args.spp} jittered spp + 4x4 ambient occlusion, room + {args.instances} instanced "
                f"meshes = {ntri} triangles behind the reference's shape-level BVH, 1 point light"),
        }[args.config]
        ref_unit_total = rays_total + skipped_total
        line = {
            "metric": "Mrays/s", "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world_size,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # `value` counts the closest-hit queries this path TRACED.  The reference traces more for the same frame (a shadow
            # ray before it tests n.wi > 0, AO rays it multiplies by ka = 0): rays_per_frame_reference_unit is its own count
            # (tests pin traced + skipped == the reference's), value_reference_unit the same frames per second in that unit —
            # the figure to hold against cpu_baseline.value, which is in the reference's unit by construction.
            "value_reference_unit": round(ref_unit_total / (ms_per_step * 1e-3) / 1e6, 2),
            "config": {"workload": workload,
                       "rays_per_frame": int(rays_total), "rays_per_frame_reference_unit": int(ref_unit_total), "frame_checksum": checksum,
                       "batches_per_frame": int(st_count.batches),
                       "parallelism": f"tiles64 x {world_size} gpu"},
            "kernel_ms_per_step": {n: round(float(kernel_ms[i]) / args.steps, 3) for i, n in enumerate(capi.RT_K_NAMES)},
            "roofline": roofline,
        }
        if world_size == 1 and not args.no_cpu_baseline and (args.config in (3, 4) and not args.force_collective or args.cpu_baseline):
            line["cpu_baseline"] = cpu_baseline(world, flat, args)
        print(json.dumps(line), flush=True)
    r.close()
    if distributed:
        dist.destroy_process_group()


def cpu_baseline(world, flat, args):
    """The reference itself (oracle/_ref, built from /root/reference in the build container) — or,
    when that library is absent, our C restatement — timed on this host on a centre crop of the
    same scene at 1 spp (the reference algorithm scans every triangle of a mesh per ray)."""
    from oracle import bindings
    from raytracer_amd import capi
    # all the cores this process may use: the box's CPU share (exported as OMP_NUM_THREADS there; os.cpu_count()
    # shows the whole machine), else the affinity mask
    host_cpus = os.cpu_count() or 1
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or len(os.sched_getaffinity(0))
    nthreads = max(1, min(share, host_cpus))
    if bindings.RefOracle.available():
        orc, kind = bindings.RefOracle(), "reference"
    else:
        orc, kind = bindings.PortOracle(), "port"
    c = args.cpu_crop or {3: 96, 4: 64}.get(args.config, 24)
    p = world.params(seed=1, max_depth=8) if args.config == 4 else world.params(seed=1)
    p.num_samples = 1
    p.x0, p.x1 = args.width // 2 - c // 2, args.width // 2 - c // 2 + c
    p.y0, p.y1 = args.height // 2 - c // 2, args.height // 2 - c // 2 + c
    res = orc.render(flat, world.camera, p, rng_mode=1, nthreads=nthreads, ids=False)
    return {"value": round(res.rays / res.seconds / 1e6, 4), "unit": "Mrays/s (the reference's own query count: compare with value_reference_unit)",
            "cores": nthreads, "threads": nthreads, "host_cpus": host_cpus, "kind": kind,
            "sample": (f"{c}x{c} px centre crop of the same frame at 1 spp (jitter + 4x4 AO on), {res.rays} rays in {res.seconds:.1f} s; "
                       f"the reference's Shape::hit / Material::shade called from the harness's own pixel loop on {nthreads} std::threads "
                       "(its raytrace() mis-indexes non-square images, SURVEY Q14, and its ThreadPool is bypassed)")}


if __name__ == "__main__":
    main()
