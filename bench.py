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

# What the kernel itself executes per ray on the default workload, from the build with stage counters
# (profiles/r1_stage_counters.txt): its SAH tree needs far fewer tests than the reference's tree.
OWN_TESTS_PER_RAY = {"node_tests": 24.04, "sphere_tests": 2.11}


def measured_hbm_traffic(workload_is_default):
    """HBM bytes of one render_kernel launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate runs; FETCH_SIZE counts 64 B per 128-B request on gfx950 and is doubled)."""
    path = os.path.join(ROOT, "profiles", "r1_final_hbm_counters.csv")
    if not workload_is_default or not os.path.exists(path):
        return None
    import csv

    kb = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if "render_kernel" in row["Kernel_Name"]:
                kb[row["Counter_Name"]] = float(row["Counter_Value"])
    if "FETCH_SIZE" not in kb or "WRITE_SIZE" not in kb:
        return None
    return int((2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024)


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
    ap.add_argument("--mesh-nlat", type=int, default=0, help="triangle-mesh / mixed scenes: latitude rings of the synthetic mesh (500 = 998k triangles)")
    ap.add_argument("--save", default=None, help="write the frame to this PNG (rank 0)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    if args.split == "tile":
        begin, end, tile_rank, tile_world = 0, args.spp, rank, world
    else:
        (begin, end), tile_rank, tile_world = parallel.sample_range(args.spp, rank, world), 0, 1
    kernel_ms, rays = [], []

    def step(record):
        accum.zero_()
        st = render.Stats()
        cam.render_accumulate(scene, accum.data_ptr(), begin, end, seed=args.seed, traversal=trav, rr_start_depth=args.rr,
                              stream=stream, stats=st, tile_rank=tile_rank, tile_world=tile_world)
        if world > 1:
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            cam.resolve(accum.data_ptr(), rgb.data_ptr(), stream=stream)
        if record:
            kernel_ms.append(st.kernel_ms)
            rays.append(st.rays)

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
    if world > 1:
        t = torch.tensor([elapsed, float(sum(rays))], device="cuda", dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rays = float(t[1])
    else:
        total_rays = float(sum(rays))

    if rank == 0:
        samples_per_step = cfg.width * cfg.height * args.spp
        value = samples_per_step * args.steps / elapsed / 1e6
        out = {
            "metric": "Msamples/sec (whole node) at 1920x1080, depth 50",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene} seed {args.seed} ({scene.n_prims} primitives), "
                                   f"{cfg.width}x{cfg.height}, {args.spp} spp, depth {args.depth}",
                       "traversal": args.traversal, "russian_roulette_start": args.rr,
                       "parallelism": f"{args.split}-split x{world} + {'RCCL' if args.backend == 'nccl' else args.backend} reduce" if world > 1 else "single GPU"},
            "mrays_per_s": round(total_rays / elapsed / 1e6, 1),
            "rays_per_sample": round(total_rays / (samples_per_step * args.steps), 3),
            "kernel_ms": round(float(np.mean(kernel_ms)), 3),  # rank 0's render_kernel, HIP events on the launch stream
        }
        if args.save:
            from PIL import Image

            Image.fromarray(rgb.cpu().numpy().reshape(cfg.height, cfg.width, 3)).save(args.save)

        if world == 1 and not args.no_cpu:
            # CPU baseline + work counts: the oracle renders a bounded sample of the same frame on this box's cores
            from oracle import binding as ob

            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
            ccfg = cfg.replace(samples_per_pixel=args.cpu_spp)
            om, oc = ob.OracleMesh(bvh), ob.OracleCamera(ccfg)
            ob.counters_reset()
            t1 = time.perf_counter()
            oc.render(om, seed=args.seed, rng_kind=0, threads=cores, want_rgb=False)
            cpu_s = time.perf_counter() - t1
            cnt = ob.counters()
            cpu_samples = cfg.width * cfg.height * args.cpu_spp
            flops_per_sample = sum(FLOPS[k] * cnt[k] for k in FLOPS) / cnt["samples"]
            out["cpu_baseline"] = {
                "value": round(cpu_samples / cpu_s / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
                "sample": f"same scene and camera, {cfg.width}x{cfg.height} at {args.cpu_spp} spp ({cpu_samples / 1e6:.1f} Msamples, "
                          f"{cpu_s:.1f} s): C++ restatement of the reference CPU renderer (pointer-tree BVH, virtual dispatch, "
                          f"ChaCha12 + rejection samplers), not the Rust binary",
                "rays_per_sample": round(cnt["rays"] / cnt["samples"], 3),
                "node_tests_per_ray": round(cnt["node_tests"] / cnt["rays"], 2),
                "sphere_tests_per_ray": round(cnt["sphere_tests"] / cnt["rays"], 2),
            }
            out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
            ms = float(np.mean(kernel_ms))
            flops_per_launch = flops_per_sample * samples_per_step
            achieved = flops_per_launch / (ms * 1e-3) / 1e12
            default_workload = (args.scene, args.seed, args.width, args.height, args.spp, args.depth, args.traversal, args.rr) == \
                ("bouncing-spheres", 1, 1920, 1080, 1000, 50, "fast", 0)
            ref_work = {
                "flops_per_sample": round(flops_per_sample, 1), "achieved": round(achieved, 3), "frac": round(achieved / FP32_VECTOR_PEAK_TFLOPS, 4),
                "note": "f32 operations the REFERENCE performs for these samples (instrumented oracle; its loose origin-including "
                        "boxes make it test ~110 nodes and ~75 spheres per ray) / kernel time: work delivered, which can exceed "
                        "the peak because the kernel reaches the same hits with far fewer tests",
            }
            out["roofline"] = {"bound": "valu", "kernel": "render_kernel", "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "traffic": measured_hbm_traffic(default_workload), "kernel_ms": round(ms, 3)}
            if default_workload:
                # the kernel's own algorithm, same per-event costs: the operations its walk needs for these samples
                rps = total_rays / (samples_per_step * args.steps)
                own = dict(cnt)
                own["node_tests"] = OWN_TESTS_PER_RAY["node_tests"] * rps * cnt["samples"]
                own["sphere_tests"] = OWN_TESTS_PER_RAY["sphere_tests"] * rps * cnt["samples"]
                own_fps = sum(FLOPS[k] * own[k] for k in FLOPS) / cnt["samples"]
                own_tf = own_fps * samples_per_step / (ms * 1e-3) / 1e12
                out["roofline"].update({
                    "achieved": round(own_tf, 3), "frac": round(own_tf / FP32_VECTOR_PEAK_TFLOPS, 4),
                    "algorithmic_flops_per_sample": round(own_fps, 1),
                    "note": "algorithmic work of the kernel's own walk: 24.0 box + 2.1 sphere tests per ray (stage-counter build, "
                            "profiles/r1_stage_counters.txt) and the reference's shading, at the reference's per-event costs. The "
                            "kernel issues 0.72 G vector instructions/s per SIMD (the measured issue ceiling of this instruction mix) "
                            "at 58% active lanes; most of a box test is min/max/compare, not FMA",
                    "reference_work": ref_work,
                })
            else:
                out["roofline"].update({"achieved": ref_work["achieved"], "frac": ref_work["frac"],
                                        "algorithmic_flops_per_sample": ref_work["flops_per_sample"], "note": ref_work["note"]})
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
