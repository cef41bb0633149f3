#!/usr/bin/env python3
"""Headline benchmark: Msamples/s (whole job) of the path-tracing hot path at 1920x1080, depth 50.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (N = 1): the bouncing-spheres scene (main.rs:161-228, seed 1, ~485 spheres with Lambertian / metal /
dielectric materials, moving spheres, checker ground, defocus 0.6) under the reference's camera for that scene at
1920x1080, 1000 samples per pixel, max depth 50.  One step = one whole frame (2073.6 Msamples), scene already
resident in HBM.  N > 1: the same frame, its 1000 samples split into N contiguous ranges, one per rank, and the
fp32 accumulation buffers reduced onto rank 0 (RCCL) before the sRGB resolve: strong scaling.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel (render_kernel) against the FP32 vector peak: algorithmic flops of the reference's
                own hit()/scatter() work per sample (counted by the instrumented CPU oracle on this fixture) x samples
                per launch / the kernel's HIP-event duration
  cpu_baseline  the CPU oracle (C++ restatement of the reference's CPU renderer, reference-faithful mode) timed on
                this box's host cores on a bounded sample of the same workload (N = 1, rank 0 only)
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic f32 operations of the reference per event (one each for add/sub/mul/div/sqrt/min/max/compare/
# transcendental), counted in the cited Rust lines
FLOPS = {
    "node_tests": 27,        # Aabb::hit, mesh.rs:515-535: per axis 1 div, 2 sub, 2 mul, order compare, max, min, exit compare
    "sphere_tests": 29,      # Sphere::hit to the discriminant, mesh.rs:114-120
    "sphere_hits": 38,       # sqrt, roots, range, point, normal, uv, Hit::new, mesh.rs:125-138, :40-45
    "quad_tests": 50,        # Quad::hit, mesh.rs:203-220
    "samples": 30,           # get_ray, camera.rs:147-163
    "shade_lambert": 25,     # material.rs:70-96
    "shade_metal": 35,       # material.rs:104-119
    "shade_dielectric": 60,  # material.rs:149-173
    "shade_iso": 25,
    "tex_checker": 8,        # texture.rs:63-74
    "tex_image": 10,         # texture.rs:100-114
    "tex_noise": 910,        # texture.rs:132-137 with 7 octaves of ~130
}
FP32_VECTOR_PEAK_TFLOPS = 157.3

HBM_PEAK_TBS = 8.0


def measured_hbm_traffic(tag):
    """(bytes, file) -- HBM-side bytes of one render_kernel launch of a profiled workload (tag "c2" = the default
    workload, "c4" = the million-triangle configuration) from the newest committed rocprofv3 PMC passes under profiles/
    (FETCH_SIZE and WRITE_SIZE are collected in separate runs; FETCH_SIZE counts 64 B per 128-B request on gfx950 and is
    doubled).  A PMC pass cannot run inside this process, so this is the one figure of the line that is not measured
    live; the file it came from is named beside it."""
    import csv
    import glob

    if tag is None:
        return None, None
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_hbm_counters.csv")), reverse=True)
    if tag == "c2":
        paths.append(os.path.join(ROOT, "profiles", "r1_final_hbm_counters.csv"))
    for path in paths:
        if not os.path.exists(path):
            continue
        kb = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if "render_kernel" in row["Kernel_Name"]:
                    kb[row["Counter_Name"]] = float(row["Counter_Value"])
        if "FETCH_SIZE" in kb and "WRITE_SIZE" in kb:
            return int((2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024), os.path.relpath(path, ROOT)
    return None, None


def memory_counters_per_ray(tag):
    """L2 requests and bytes past L2 per ray of a profiled workload, from the newest committed summary under profiles/
    (written by scratch/profile_round.sh from rocprofv3 PMC passes of this same command); None when there is none."""
    import glob

    if tag is None:
        return None
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{tag}_memory_per_ray.json")), reverse=True)
    for path in paths:
        with open(path) as f:
            d = json.load(f)
        d["source"] = os.path.relpath(path, ROOT)
        return d
    return None


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: start one process per GPU through torch.distributed.run from
    this process, which has not touched the GPU, and pass on rank 0's JSON line and the exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="bouncing-spheres")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--traversal", choices=["fast", "strict"], default="fast")
    ap.add_argument("--split", choices=["spp", "tile"], default="spp", help="multi-GPU partition: sample ranges or interleaved 8x8 tiles")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--rr", type=int, default=0, help="Russian roulette start depth (0 = off, as in the reference)")
    ap.add_argument("--cpu-spp", type=int, default=16, help="samples per pixel of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-scale", type=int, default=1, help="CPU legs (event counts and baseline) render width/N x height/N of the same camera: "
                                                             "per-sample figures do not depend on the resolution (large scenes)")
    ap.add_argument("--mesh-nlat", type=int, default=0, help="triangle-mesh / mixed scenes: latitude rings of the synthetic mesh (500 = 998k triangles)")
    ap.add_argument("--save", default=None, help="write the frame to this PNG (rank 0)")
    ap.add_argument("--work-counts", default=None, help="write the event counts the roofline is priced with (oracle events per sample, "
                                                        "the kernel's own tests of this run) to this JSON file")
    ap.add_argument("--no-secondary", action="store_true", help="default workload only: skip the million-triangle block (`secondary.c4`)")
    ap.add_argument("--rmse-scale", type=int, default=4, help="image_rmse: the frames compared are width/N x height/N of the same camera")
    ap.add_argument("--rmse-spp", type=int, default=64, help="image_rmse: samples per pixel of the frames compared")
    ap.add_argument("--in-library", action="store_true",
                    help="multi-GPU through the C ABI instead of one process per GPU: ONE process, rtx_render_multi drives --gpus devices")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.in_library:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))  # before anything touches the GPU

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.in_library:
        return in_library(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    render = importlib.import_module("ray-tracer_amd.render")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    parallel = importlib.import_module("ray-tracer_amd.parallel")

    world_args = {"n_lat": args.mesh_nlat} if args.mesh_nlat else None
    cfg, bvh = scenes.make(args.scene, seed=args.seed, world_args=world_args, width=args.width, height=args.height,
                           samples_per_pixel=args.spp, max_depth=args.depth)
    scene = render.Scene(bvh)  # flatten + upload: resident in HBM before the timed region
    cam = render.Camera(cfg)
    trav = render.TRAVERSAL_STRICT if args.traversal == "strict" else render.TRAVERSAL_FAST
    n3 = cfg.width * cfg.height * 3
    accum = torch.zeros(n3, device="cuda", dtype=torch.float32)
    rgb = torch.zeros(n3, device="cuda", dtype=torch.uint8)
    stream = torch.cuda.current_stream().cuda_stream
    tile_rank, tile_world = (rank, world) if args.split == "tile" else (0, 1)
    kernel_ms, rays, box_tests, prim_tests, flags = [], [], [], [], []
    timings = {}

    def step(record):
        accum.zero_()
        st = render.Stats()

        def render_fn(begin, end):  # this rank's share: a sample range of every pixel, or all samples of its tiles
            cam.render_accumulate(scene, accum.data_ptr(), begin, end, seed=args.seed, traversal=trav, rr_start_depth=args.rr,
                                  stream=stream, stats=st, tile_rank=tile_rank, tile_world=tile_world)
            return accum  # (asking for stats synchronises the stream: render_s is the share's device time plus launch)

        parallel.render_split(render_fn, args.spp, rank, world, dist=dist, split=args.split, timings=timings if record else None)
        if world > 1 and record:
            t_sync = time.perf_counter()
            torch.cuda.synchronize()  # the reduce is asynchronous: its device time ends here
            timings["reduce_s"] += time.perf_counter() - t_sync
        if rank == 0:
            cam.resolve(accum.data_ptr(), rgb.data_ptr(), stream=stream)
        if record:
            kernel_ms.append(st.kernel_ms)
            rays.append(st.rays)
            box_tests.append(st.box_tests)
            prim_tests.append(st.prim_tests)
            flags.append(st.flags)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed, float(sum(rays)), float(sum(box_tests)), float(sum(prim_tests))], device="cuda", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays, total_box, total_prim = float(t[1]), float(t[2]), float(t[3])
        # what every rank spent where, so that a scaling curve explains itself: kernel time of its share, host time in the
        # render call, time in the reduce (which includes waiting for the slowest rank)
        mine = torch.tensor([float(np.mean(kernel_ms)), timings.get("render_s", 0.0) / args.steps * 1e3, timings.get("reduce_s", 0.0) / args.steps * 1e3],
                            device="cuda", dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = {"kernel_ms": [round(float(g[0]), 3) for g in gathered], "render_call_ms": [round(float(g[1]), 3) for g in gathered],
                    "reduce_ms": [round(float(g[2]), 3) for g in gathered]}
    else:
        total_rays, total_box, total_prim = float(sum(rays)), float(sum(box_tests)), float(sum(prim_tests))

    if rank == 0:
        parallelism = f"{args.split}-split x{world}, one process per GPU + {'RCCL' if args.backend == 'nccl' else args.backend} reduce" if world > 1 else "single GPU"
        out = result_line(args, cfg, scene.n_prims, world, elapsed, total_rays, total_box, total_prim, float(np.mean(kernel_ms)), parallelism)
        if per_rank is not None:
            out["per_rank"] = per_rank
        if args.save:
            from PIL import Image

            Image.fromarray(rgb.cpu().numpy().reshape(cfg.height, cfg.width, 3)).save(args.save)
        if world == 1:
            # the same frame through rtx_render with host buffers: scene resident, image (and nothing else) over PCIe
            t1 = time.perf_counter()
            cam.render(scene, seed=args.seed, traversal=trav, rr_start_depth=args.rr)
            dt = time.perf_counter() - t1
            out["pcie_inclusive"] = {"value": round(cfg.width * cfg.height * args.spp / dt / 1e6, 2), "unit": "Msamples/s", "ms": round(dt * 1e3, 3),
                                     "note": "one rtx_render call with host buffers: device allocation, render, resolve, RGB8 copied to the host"}
        if world == 1 and not args.no_cpu:
            cpu_and_roofline(out, args, cfg, bvh, scene, float(np.mean(kernel_ms)), total_rays, total_box, total_prim, flags[-1])
            image_rmse(out, args, cfg, bvh, scene, render, trav)
            if workload_tag(args) == "c2" and not args.no_secondary:
                scene.close()  # (frees the device copy of the first scene)
                out["secondary"] = {"c4": secondary_c4(args, render, scenes, torch)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def result_line(args, cfg, n_prims, n_gpus, elapsed, total_rays, total_box, total_prim, kernel_ms, parallelism):
    samples_per_step = cfg.width * cfg.height * args.spp
    return {
        "metric": "Msamples/sec (whole node) at 1920x1080, depth 50",
        "value": round(samples_per_step * args.steps / elapsed / 1e6, 2),
        "unit": "Msamples/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.scene} seed {args.seed} ({n_prims} primitives), "
                               f"{cfg.width}x{cfg.height}, {args.spp} spp, depth {args.depth}",
                   "traversal": args.traversal, "russian_roulette_start": args.rr, "parallelism": parallelism},
        "mrays_per_s": round(total_rays / elapsed / 1e6, 1),
        "rays_per_sample": round(total_rays / (samples_per_step * args.steps), 3),
        # counted by the kernel itself in this run (lanes x steps of its box / primitive stages)
        "box_tests_per_ray": round(total_box / max(total_rays, 1.0), 3),
        "prim_tests_per_ray": round(total_prim / max(total_rays, 1.0), 3),
        "kernel_ms": round(kernel_ms, 3),  # render_kernel, HIP events on the launch stream (rank 0 / slowest device)
    }


def workload_tag(args):
    """"c2" for the default workload, "c4" for the million-triangle configuration: the workloads whose profiles are committed."""
    key = (args.scene, args.seed, args.width, args.height, args.spp, args.depth, args.traversal, args.rr, args.mesh_nlat)
    return {("bouncing-spheres", 1, 1920, 1080, 1000, 50, "fast", 0, 0): "c2", ("triangle-mesh", 1, 1920, 1080, 256, 50, "fast", 0, 500): "c4"}.get(key)


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))


def image_rmse(out, args, cfg, bvh, scene, render, trav):
    """The metric's second half: image RMSE against the CPU reference.  A reduced frame of the same scene and camera is
    rendered by the reference-faithful oracle with two seeds (A, B) and by the GPU with a third; all three are linear fp32
    radiance means.  RMSE(CPU A, CPU B) is the Monte-Carlo noise floor at this sample count; the GPU frame passes when its
    RMSE against A stays within 1.25 x that floor and its mean luminance (Rec. 709) within 0.5 % of A's -- or within
    1.5 x the CPU's own seed-to-seed luminance difference where the estimator is that noisy."""
    import numpy as np
    from oracle import binding as ob

    w, h, spp = max(16, cfg.width // args.rmse_scale), max(16, cfg.height // args.rmse_scale), args.rmse_spp
    small = cfg.replace(width=w, height=h, samples_per_pixel=spp)
    cores = host_cores()
    print(f"[bench] image RMSE: {w}x{h} at {spp} spp, two oracle frames on {cores} threads ...", file=sys.stderr, flush=True)
    ob.select("fast")
    om, oc = ob.OracleMesh(bvh), ob.OracleCamera(small)
    t0 = time.perf_counter()
    a = oc.render(om, seed=101, rng_kind=0, threads=cores, want_rgb=False)[0] / spp
    b = oc.render(om, seed=202, rng_kind=0, threads=cores, want_rgb=False)[0] / spp
    cpu_s = time.perf_counter() - t0
    ob.select("checker")
    g = render.Camera(small).render(scene, seed=303, traversal=trav, rr_start_depth=args.rr, want_accum=True)[1] / spp
    luma = np.array([0.2126, 0.7152, 0.0722])
    la, lb, lg = float((a @ luma).mean()), float((b @ luma).mean()), float((g @ luma).mean())
    r0 = float(np.sqrt(np.mean((a - b) ** 2)))
    r1 = float(np.sqrt(np.mean((g - a) ** 2)))
    lum_tol = max(0.005, 1.5 * abs(la - lb) / abs(la))
    out["image_rmse"] = {
        "frame": f"{w}x{h} at {spp} spp, linear fp32 radiance means",
        "gpu_vs_cpu_a": round(r1, 6), "cpu_a_vs_cpu_b": round(r0, 6), "ratio": round(r1 / r0, 4) if r0 > 0 else None,
        "mean_luminance": {"gpu": round(lg, 6), "cpu_a": round(la, 6), "cpu_b": round(lb, 6), "gpu_vs_cpu_a_rel": round(abs(lg - la) / abs(la), 6),
                           "tolerance_rel": round(lum_tol, 6)},
        "pass": bool(r1 <= 1.25 * r0 and abs(lg - la) <= lum_tol * abs(la)),
        "criterion": "RMSE(GPU, CPU seed A) <= 1.25 x RMSE(CPU seed A, CPU seed B); mean luminance within 0.5 % (or 1.5 x the CPU's own seed-to-seed difference)",
        "cpu": f"reference-faithful oracle (ChaCha12 stream, rejection samplers), seeds 101 / 202, {cpu_s:.1f} s; GPU seed 303",
    }


def secondary_c4(args, render, scenes, torch):
    """The million-triangle configuration (triangle-mesh, 998 000 triangles, 1920x1080, 256 spp) measured like the
    headline workload: one step = one frame, scene resident; with its own roofline (the walk reads its nodes and
    primitives from HBM / L2: bound "hbm"), CPU baseline and image RMSE, all on reduced CPU frames (the oracle's
    reference-topology tree tests ~16 000 nodes per ray on this scene)."""
    import argparse
    import numpy as np

    a2 = argparse.Namespace(**vars(args))
    a2.scene, a2.seed, a2.mesh_nlat, a2.spp, a2.width, a2.height = "triangle-mesh", 1, 500, 256, 1920, 1080
    a2.cpu_scale, a2.cpu_spp, a2.rmse_scale, a2.rmse_spp, a2.work_counts = 8, 4, 20, 16, None
    print("[bench] secondary workload: building the 998 000-triangle scene ...", file=sys.stderr, flush=True)
    cfg, bvh = scenes.make(a2.scene, seed=a2.seed, world_args={"n_lat": a2.mesh_nlat}, width=a2.width, height=a2.height,
                           samples_per_pixel=a2.spp, max_depth=a2.depth)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    trav = render.TRAVERSAL_STRICT if a2.traversal == "strict" else render.TRAVERSAL_FAST
    n3 = cfg.width * cfg.height * 3
    accum = torch.zeros(n3, device="cuda", dtype=torch.float32)
    rgb = torch.zeros(n3, device="cuda", dtype=torch.uint8)
    stream = torch.cuda.current_stream().cuda_stream
    stats = []

    def step(record):
        accum.zero_()
        st = render.Stats()
        cam.render_accumulate(scene, accum.data_ptr(), 0, a2.spp, seed=a2.seed, traversal=trav, rr_start_depth=a2.rr, stream=stream, stats=st)
        cam.resolve(accum.data_ptr(), rgb.data_ptr(), stream=stream)
        if record:
            stats.append(st)

    for _ in range(a2.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a2.steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rays, box, prim = (float(sum(getattr(s, k) for s in stats)) for k in ("rays", "box_tests", "prim_tests"))
    kernel_ms = float(np.mean([s.kernel_ms for s in stats]))
    out = result_line(a2, cfg, scene.n_prims, 1, elapsed, rays, box, prim, kernel_ms, "single GPU")
    for k in ("metric", "n_gpus", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        del out[k]
    cpu_and_roofline(out, a2, cfg, bvh, scene, kernel_ms, rays, box, prim, stats[-1].flags)
    image_rmse(out, a2, cfg, bvh, scene, render, trav)
    return out


def cpu_and_roofline(out, args, cfg, bvh, scene, kernel_ms, total_rays, total_box, total_prim, flags):
    """CPU baseline (the oracle timed on this box's cores on a bounded sample of the same frame) and the roofline of the
    dominant kernel from this run's own counts.  flags: rtx_stats_t.flags of the timed launches."""
    from oracle import binding as ob

    entries_in_lds, packed, wide = bool(flags & 1), bool(flags & 2), bool(flags & 4)

    samples_per_step = cfg.width * cfg.height * args.spp
    cores = host_cores()
    # event counts per sample (hits, shades, texture evaluations: the same events whoever traces the rays) from the
    # instrumented build on a 2-spp frame; the TIMED render uses the uninstrumented -O3 build
    cw, ch = max(1, cfg.width // args.cpu_scale), max(1, cfg.height // args.cpu_scale)
    print(f"[bench] CPU legs: {cw}x{ch}, counting events at 2 spp ...", file=sys.stderr, flush=True)
    ob.select("checker")
    ob.counters_reset()
    ob.OracleCamera(cfg.replace(width=cw, height=ch, samples_per_pixel=2)).render(ob.OracleMesh(bvh), seed=args.seed, rng_kind=0, threads=cores, want_rgb=False)
    cnt = ob.counters()
    print(f"[bench] timing the CPU baseline at {args.cpu_spp} spp on {cores} threads ...", file=sys.stderr, flush=True)
    ob.select("fast")
    om, oc = ob.OracleMesh(bvh), ob.OracleCamera(cfg.replace(width=cw, height=ch, samples_per_pixel=args.cpu_spp))
    t1 = time.perf_counter()
    oc.render(om, seed=args.seed, rng_kind=0, threads=cores, want_rgb=False)
    cpu_s = time.perf_counter() - t1
    ob.select("checker")
    cpu_samples = cw * ch * args.cpu_spp
    out["cpu_baseline"] = {
        "value": round(cpu_samples / cpu_s / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"same scene and camera, {cw}x{ch} at {args.cpu_spp} spp ({cpu_samples / 1e6:.1f} Msamples, "
                  f"{cpu_s:.1f} s): C++ restatement of the reference CPU renderer (pointer-tree BVH, virtual dispatch, "
                  f"ChaCha12 + rejection samplers), not the Rust binary",
        "build": "g++ " + ob.BUILD_FLAGS["fast"],
        "rays_per_sample": round(cnt["rays"] / cnt["samples"], 3),
        "node_tests_per_ray": round(cnt["node_tests"] / cnt["rays"], 2),
        "prim_tests_per_ray": round((cnt["sphere_tests"] + cnt["quad_tests"]) / cnt["rays"], 2),
    }
    out["speedup_vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1)

    # algorithmic f32 operations per sample at the reference's per-event costs (FLOPS):
    #   reference_work  every event as the reference performs it (its loose boxes make it test far more nodes)
    #   own work        box and primitive tests as THIS RUN's kernel counted them; hits, shading and textures per sample
    #                   as the oracle counted them (those events do not depend on who traces the ray)
    ref_fps = sum(FLOPS[k] * cnt[k] for k in FLOPS) / cnt["samples"]
    n_samples = samples_per_step * args.steps
    prim_events = cnt["sphere_tests"] + cnt["quad_tests"]
    prim_cost = (FLOPS["sphere_tests"] * cnt["sphere_tests"] + FLOPS["quad_tests"] * cnt["quad_tests"]) / max(prim_events, 1)
    other = sum(FLOPS[k] * cnt[k] for k in FLOPS if k not in ("node_tests", "sphere_tests", "quad_tests")) / cnt["samples"]
    own_fps = FLOPS["node_tests"] * total_box / n_samples + prim_cost * total_prim / n_samples + other
    secs = kernel_ms * 1e-3
    own_tf = own_fps * samples_per_step / secs / 1e12
    ref_tf = ref_fps * samples_per_step / secs / 1e12
    tag = workload_tag(args)
    traffic, traffic_file = measured_hbm_traffic(tag)
    valu = {"bound": "valu", "kernel": "render_kernel", "achieved": round(own_tf, 3), "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(own_tf / FP32_VECTOR_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_file, "kernel_ms": round(kernel_ms, 3),
            "algorithmic_flops_per_sample": round(own_fps, 1),
            "note": "f32 operations of the kernel's own walk -- box and primitive tests counted by the kernel in this run -- plus the "
                    "reference's hit completion, shading and texture work per sample, all at the reference's per-event costs, per "
                    "launch / the kernel's HIP-event time",
            "reference_work": {"flops_per_sample": round(ref_fps, 1), "achieved": round(ref_tf, 3), "frac": round(ref_tf / FP32_VECTOR_PEAK_TFLOPS, 4),
                               "note": "the same samples priced with the tests the REFERENCE performs (its origin-including boxes make its "
                                       "tree nearly useless): work delivered, which can exceed the peak because the kernel avoids it"}}
    # entries beyond the LDS budget are walked from HBM / L2.  Algorithmic bytes per test, as the kernel lays its data out:
    #   wide form (no transforms / media):     a node of 64 B holds up to four child boxes: 16 B per box test; a primitive record 80 B
    #   packed array (RTX_WIDE=0):             box 16 B (six halfs + link), sphere 32 B, quad / triangle 80 B, all inline
    #   32-byte entries otherwise:             box 32 B, sphere 32 B, quad / triangle 32 B entry + 80 B record
    box_b, sph_b, planar_b = (16.0, 80.0, 80.0) if wide else (16.0, 32.0, 80.0) if packed else (32.0, 32.0, 112.0)
    rec_bytes = (sph_b * cnt["sphere_tests"] + planar_b * cnt["quad_tests"]) / max(prim_events, 1)
    gbs = (box_b * total_box + rec_bytes * total_prim) / args.steps / secs / 1e9
    hbm = {"bound": "hbm", "kernel": "render_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
           "frac": round(gbs / (HBM_PEAK_TBS * 1e3), 4), "frac_is": "the walk's demand / HBM peak", "traffic": traffic, "traffic_source": traffic_file,
           "kernel_ms": round(kernel_ms, 3), "layout": "wide" if wide else "packed" if packed else "entries",
           "bytes_per_test": {"box": box_b, "primitive": round(rec_bytes, 1)},
           "bytes_per_ray": round((box_b * total_box + rec_bytes * total_prim) / max(total_rays, 1.0), 1),
           "measured_per_ray": memory_counters_per_ray(tag),
           "note": "algorithmic node + primitive bytes of the kernel's own walk (tests counted by the kernel in this run x the bytes "
                   "one test reads) per launch / kernel time; most of these reads hit L1, L2 or the Infinity Cache, so `frac` is the "
                   "walk's demand over the HBM peak, not an achieved DRAM fraction: the counters under measured_per_ray say where "
                   "the bytes actually came from"}
    if args.work_counts:
        with open(args.work_counts, "w") as f:
            json.dump({
                "workload": out["config"]["workload"],
                "flops_per_event": FLOPS,
                "oracle_events": {"note": "instrumented CPU restatement, reference-faithful mode, same fixture and seed, 2 spp", **{k: int(v) for k, v in cnt.items()}},
                "kernel_run": {"note": "counted by render_kernel in the timed steps of this run", "steps": args.steps, "samples": int(n_samples),
                               "rays": int(total_rays), "box_tests": int(total_box), "prim_tests": int(total_prim), "kernel_ms": round(kernel_ms, 3),
                               "entries_in_lds": bool(entries_in_lds), "packed_array": bool(packed)},
                "flops_per_sample": {"reference_work": round(ref_fps, 1), "own_walk": round(own_fps, 1)},
                "bytes_per_test": {"box": box_b, "primitive": round(rec_bytes, 1)},
            }, f, indent=1)
            f.write("\n")
    if entries_in_lds:
        out["roofline"] = valu
    else:
        out["roofline"] = hbm
        out["roofline_valu"] = valu


def in_library(args):
    """--in-library: the multi-GPU path behind the C ABI.  One process; rtx_render_multi replicates the scene onto
    --gpus devices, renders every share on its own stream and gathers on device 0."""
    import numpy as np

    render = importlib.import_module("ray-tracer_amd.render")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    world_args = {"n_lat": args.mesh_nlat} if args.mesh_nlat else None
    cfg, bvh = scenes.make(args.scene, seed=args.seed, world_args=world_args, width=args.width, height=args.height,
                           samples_per_pixel=args.spp, max_depth=args.depth)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    trav = render.TRAVERSAL_STRICT if args.traversal == "strict" else render.TRAVERSAL_FAST
    split = render.SPLIT_TILES if args.split == "tile" else render.SPLIT_SAMPLES
    devices = [0] * args.gpus if args.one_device else list(range(args.gpus))
    kernel_ms, rays, box, prim = [], 0.0, 0.0, 0.0
    rgb = None
    for _ in range(args.warmup):
        cam.render(scene, seed=args.seed, traversal=trav, rr_start_depth=args.rr, split=split, devices=devices)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgb, st = cam.render(scene, seed=args.seed, traversal=trav, rr_start_depth=args.rr, split=split, devices=devices, want_stats=True)
        kernel_ms.append(st.kernel_ms)
        rays, box, prim = rays + st.rays, box + st.box_tests, prim + st.prim_tests
    elapsed = time.perf_counter() - t0  # every call blocks until the frame is on the host
    out = result_line(args, cfg, scene.n_prims, args.gpus, elapsed, rays, box, prim, float(np.mean(kernel_ms)),
                      f"{args.split}-split x{args.gpus}, one process: rtx_render_multi, peer copies to device 0"
                      + (" (all shares on GPU 0: rehearsal)" if args.one_device else ""))
    if args.save:
        from PIL import Image

        Image.fromarray(rgb).save(args.save)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
