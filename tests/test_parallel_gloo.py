"""Sample splitting across ranks (one process per GPU in production) rehearsed with world size 2 on CPU / gloo.
The stand-in renderer is the CPU oracle in mirror mode, whose stream -- like the HIP kernels' -- is a pure function of
(seed, pixel, sample index), so the reduced buffer must equal the single-rank buffer up to fp32 addition order."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sample_range_partition():
    parallel = importlib.import_module("ray-tracer_amd.parallel")
    for spp in (1, 7, 16, 1000):
        for world in (1, 2, 3, 4, 8):
            ranges = [parallel.sample_range(spp, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))  # contiguous, disjoint, complete
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.sample_range(8, 2, 2)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parallel = importlib.import_module("ray-tracer_amd.parallel")
    scenes = importlib.import_module("ray-tracer_amd.scenes")
    from oracle import binding as ob

    cfg, bvh = scenes.make("default", seed=0, width=48, height=27, samples_per_pixel=8, max_depth=20)
    mesh, cam = ob.OracleMesh(bvh), ob.OracleCamera(cfg)

    def render_fn(begin, end):
        acc, _ = cam.render(mesh, seed=5, rng_kind=1, sample_begin=begin, sample_end=end, threads=1, want_rgb=False)
        return torch.from_numpy(acc.copy())

    total = parallel.render_split(render_fn, cfg.samples_per_pixel, rank, world)
    if rank == 0:
        np.save(out_path, total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_rank(tmp_path, oracle, scenes):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    cfg, bvh = scenes.make("default", seed=0, width=48, height=27, samples_per_pixel=8, max_depth=20)
    want, _ = oracle.OracleCamera(cfg).render(oracle.OracleMesh(bvh), seed=5, rng_kind=1, threads=1, want_rgb=False)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)
    assert not np.array_equal(got, np.zeros_like(got))
