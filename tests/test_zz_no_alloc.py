"""rtx_render_accumulate_static -- the entry point that allocates nothing (its partial images and counters live in a
fixed device array inside the library) -- against rtx_render_accumulate, through the C ABI, on two calls in a row,
the second with a smaller input."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def render():
    r = importlib.import_module("ray-tracer_amd.render")
    assert r.device_count() >= 1, "no gfx950 device visible"
    return r


def _both(render, torch, scene, cam, cfg, spp, seed, **kw):
    n = cfg.width * cfg.height * 3
    st = torch.cuda.current_stream().cuda_stream
    want = torch.zeros(n, device="cuda")
    got = torch.zeros(n, device="cuda")
    cam.render_accumulate(scene, want.data_ptr(), 0, spp, seed=seed, stream=st, **kw)  # also prepares the scene for this camera
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    stats = render.Stats()
    cam.render_accumulate(scene, got.data_ptr(), 0, spp, seed=seed, stream=st, static=True, stats=stats, **kw)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "the allocation-free entry point changed the device's free memory"
    assert stats.rays > 0 and stats.box_tests > 0
    return want, got


def test_static_scratch_entry_point_matches_the_allocating_one(render, scenes):
    import torch

    # call 1: ragged frame, one sample per batch -> the two entry points cut the work identically: bit for bit
    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=203, height=117, samples_per_pixel=16)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    want, got = _both(render, torch, scene, cam, cfg, 16, seed=9)
    assert torch.equal(want, got)
    assert float(want.abs().sum()) > 0
    # call 2, right after, smaller input (another scene, fewer pixels and samples): the array still holds call 1's data
    cfg2, bvh2 = scenes.make("cornell-box", seed=0, width=96, height=64, samples_per_pixel=8)
    scene2, cam2 = render.Scene(bvh2), render.Camera(cfg2)
    want2, got2 = _both(render, torch, scene2, cam2, cfg2, 8, seed=4)
    assert torch.equal(want2, got2)
    assert float(want2.abs().sum()) > 0
    # call 3: many samples per batch, a tile share and the strict walk: equal up to the order of fp32 additions
    cfg3, bvh3 = scenes.make("default", seed=0, width=160, height=90, samples_per_pixel=300)
    scene3, cam3 = render.Scene(bvh3), render.Camera(cfg3)
    want3, got3 = _both(render, torch, scene3, cam3, cfg3, 300, seed=2, tile_rank=1, tile_world=2, traversal=render.TRAVERSAL_STRICT)
    assert torch.allclose(want3, got3, rtol=2e-5, atol=1e-4)
    assert bool(((want3 == 0) == (got3 == 0)).all())  # the same tiles were rendered


def test_static_scratch_entry_point_limits(render, scenes):
    import torch

    st = torch.cuda.current_stream().cuda_stream
    # a frame whose single partial image (12 * W * H bytes) exceeds the fixed array is refused, not truncated
    cfg, bvh = scenes.make("default", seed=0, width=3840, height=2160, samples_per_pixel=1)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    with pytest.raises(render.RtxError) as e:
        cam.render_accumulate(scene, buf.data_ptr(), 0, 1, seed=1, stream=st, static=True)
    assert e.value.code == render.RTX_ERR_UNSUPPORTED and "RTX_STATIC_SCRATCH_BYTES" in str(e.value)
    # 1920x1080 fits (two partial images): fewer, larger batches than the allocating call, same frame up to addition order
    cfg, bvh = scenes.make("default", seed=0, width=1920, height=1080, samples_per_pixel=4)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    want, got = _both(render, torch, scene, cam, cfg, 4, seed=3)
    assert torch.allclose(want, got, rtol=2e-5, atol=1e-4)
    # a camera outside the range the fast boxes were padded for needs the scene prepared by an allocating call first
    cfg, bvh = scenes.make("quads", seed=0, width=64, height=64, samples_per_pixel=4, lookfrom=(0.0, 0.0, 900.0))
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    with pytest.raises(render.RtxError) as e:
        cam.render_accumulate(scene, buf.data_ptr(), 0, 4, seed=1, stream=st, static=True)
    assert e.value.code == render.RTX_ERR_UNSUPPORTED
    cam.render_accumulate(scene, buf.data_ptr(), 0, 4, seed=1, stream=st, static=True, traversal=render.TRAVERSAL_STRICT)  # needs no preparation
    want, got = _both(render, torch, scene, cam, cfg, 4, seed=1)
    assert torch.equal(want, got)
