"""GPU parity at the sizes and scene classes the benchmark configurations name: statistical image parity against the
reference-faithful oracle for the textured / emissive / participating-media scenes, the million-triangle mesh against
the oracle on a ray subset at 1920x1080, a ten-million-triangle build with a tile split at 3840x2160, repeat-run
behaviour at the benchmark's batch sizes, the fused box test on axis-parallel rays, and the multi-device entry point
(rehearsed with several shares on one GPU)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32
MISS = 0xFFFFFFFF
LUMA = np.array([0.2126, 0.7152, 0.0722])


@pytest.fixture(scope="module")
def render():
    r = importlib.import_module("ray-tracer_amd.render")
    assert r.device_count() >= 1, "no gfx950 device visible"
    return r


@pytest.fixture(scope="module")
def capi():
    return importlib.import_module("ray-tracer_amd.capi")


def _surface_rays(capi, hits, n, seed):
    """Rays starting on hit points with random un-normalised directions and times (where bounce rays start)."""
    rng = np.random.default_rng(seed)
    pts = hits["p"][hits["prim_id"] != MISS]
    rays = np.zeros(n, dtype=capi.RAY_DTYPE)
    rays["o"] = pts[rng.integers(0, len(pts), n)]
    rays["d"] = rng.normal(size=(n, 3)).astype(F) * rng.uniform(0.2, 12.0, size=(n, 1)).astype(F)
    rays["time"] = rng.uniform(0, 1, n).astype(F)
    return rays


# ------------------------------------------------------------------------------------------------ converged images
CONVERGED = [
    # name, seed, width, height, spp, mean-luminance tolerance
    ("earth", 0, 200, 150, 256, 0.005),          # image texture (texture.rs:83-115)
    ("perlin-spheres", 1, 160, 120, 256, 0.005),  # noise texture (texture.rs:117-244)
    ("quads", 0, 160, 160, 256, 0.005),          # Quad (mesh.rs:190-257)
    ("simple-light", 1, 160, 120, 512, 0.015),   # DiffuseLight on a black background (material.rs:181-212)
    ("cornell-smoke", 0, 100, 100, 512, 0.02),   # ConstantMedium + Isotropic (mesh.rs:627-696, material.rs:214-246)
    ("final-scene", 1, 80, 80, 384, 0.02),     # everything at once (main.rs:450-540)
]


@pytest.mark.parametrize("name,seed,w,h,spp,tol", CONVERGED)
def test_converged_image_matches_faithful_oracle(render, oracle, scenes, name, seed, w, h, spp, tol):
    """Against the reference-faithful oracle (ChaCha12 stream, rejection samplers, recursion) -- not the mirror stream:
    RMSE(GPU, CPU seed A) <= 1.25 * RMSE(CPU seed A, CPU seed B) on the linear fp32 frames, and the mean luminance within
    `tol` (or 1.5 x the CPU's own seed-to-seed difference where the estimator is that noisy)."""
    cfg, bvh = scenes.make(name, seed=seed, width=w, height=h, samples_per_pixel=spp)
    om, oc = oracle.OracleMesh(bvh), oracle.OracleCamera(cfg)
    a, _ = oc.render(om, seed=101, rng_kind=0, want_rgb=False)
    b, _ = oc.render(om, seed=202, rng_kind=0, want_rgb=False)
    _, g = render.Camera(cfg).render(render.Scene(bvh), seed=303, want_accum=True)
    a, b, g = a / spp, b / spp, g / spp
    r0 = np.sqrt(np.mean((a - b) ** 2))
    r1 = np.sqrt(np.mean((g - a) ** 2))
    assert r1 <= 1.25 * r0, (name, r1, r0)
    la, lb, lg = float((a @ LUMA).mean()), float((b @ LUMA).mean()), float((g @ LUMA).mean())
    assert abs(lg - la) <= max(tol * la, 1.5 * abs(la - lb)), (name, lg, la, lb)


# ------------------------------------------------------------------------------------------------ triangle configurations
def test_million_triangle_mesh_matches_oracle_at_1080p(render, oracle, scenes, capi):
    """998 000 triangles under the 1920x1080 camera: a subset of the primary rays and of surface-origin secondary rays
    traced through the oracle's pointer tree (built once, ~5 s) and through both walks: ids, t, normal and point
    bit-exact.  Every primary ray of the frame: fast walk == reference-order walk."""
    cfg, bvh = scenes.make("triangle-mesh", seed=1, world_args=dict(n_lat=500), width=1920, height=1080)
    scene = render.Scene(bvh)
    assert scene.desc.n_tris == 998000
    rays = oracle.OracleCamera(cfg).center_rays(0.5)
    strict = scene.trace(rays, traversal=render.TRAVERSAL_STRICT)
    fast = scene.trace(rays, traversal=render.TRAVERSAL_FAST)
    assert np.array_equal(strict["prim_id"], fast["prim_id"]) and np.array_equal(strict["t"], fast["t"])
    assert np.count_nonzero(strict["prim_id"] >= 1) > 200_000  # the mesh fills a good part of the frame
    om = oracle.OracleMesh(bvh)
    pick = np.arange(0, len(rays), 509)  # ~4 000 primary rays spread over the frame
    sub = np.concatenate([rays[pick], _surface_rays(capi, strict, 3000, 7)])
    want = om.trace(sub)
    for mode in (render.TRAVERSAL_STRICT, render.TRAVERSAL_FAST, render.TRAVERSAL_FAST_FUSED):
        got = scene.trace(sub, traversal=mode)
        bad = np.count_nonzero(got["prim_id"] != want["prim_id"])
        assert bad == 0, (mode, bad)
        hit = want["prim_id"] != MISS
        assert np.array_equal(got["t"][hit], want["t"][hit]) and np.array_equal(got["n"][hit], want["n"][hit])
        assert np.array_equal(got["p"][hit], want["p"][hit])
    assert np.count_nonzero(want["prim_id"][len(pick):] != MISS) > 1000


def test_ten_million_triangles_tile_split_at_4k(render, scenes, capi):
    """The mixed configuration at its stated size: 10 M triangles + textured / glass spheres, 3840x2160.  The fast walk
    and the reference-order walk agree on a ray subset; two interleaved tile shares add up to the single-share frame
    bit for bit (1 sample per pixel)."""
    import torch

    cfg, bvh = scenes.make("mixed", seed=1, world_args=dict(n_lat=1582), width=3840, height=2160, samples_per_pixel=1)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    assert scene.desc.n_tris > 10_000_000
    k = cam.constants()
    rng = np.random.default_rng(3)
    xs, ys = rng.integers(0, cfg.width, 200_000), rng.integers(0, cfg.height, 200_000)
    rays = np.zeros(len(xs), dtype=capi.RAY_DTYPE)
    rays["o"] = k["center"]
    rays["d"] = k["pixel00_loc"] + xs[:, None].astype(F) * k["pixel_delta_u"] + ys[:, None].astype(F) * k["pixel_delta_v"] - k["center"]
    rays["time"] = 0.5
    a = scene.trace(rays, traversal=render.TRAVERSAL_STRICT)
    b = scene.trace(rays, traversal=render.TRAVERSAL_FAST)
    assert a.tobytes() == b.tobytes()
    assert np.count_nonzero(a["prim_id"] >= 5) > 2000  # triangles are in view
    sec = _surface_rays(capi, a, 100_000, 5)
    assert scene.trace(sec, traversal=render.TRAVERSAL_STRICT).tobytes() == scene.trace(sec, traversal=render.TRAVERSAL_FAST).tobytes()
    n = cfg.width * cfg.height * 3
    st = torch.cuda.current_stream().cuda_stream
    full = torch.zeros(n, device="cuda")
    cam.render_accumulate(scene, full.data_ptr(), 0, 1, seed=2, stream=st)
    parts = torch.zeros(n, device="cuda")
    for r in range(2):
        cam.render_accumulate(scene, parts.data_ptr(), 0, 1, seed=2, stream=st, tile_rank=r, tile_world=2)
    torch.cuda.synchronize()
    assert torch.equal(parts, full)
    assert bool(torch.isfinite(full).all()) and float(full.mean()) > 0.05


# ------------------------------------------------------------------------------------------------ repeat runs
def test_repeat_runs_at_benchmark_batch_sizes(render, scenes):
    """With more samples than batches, several samples of a pixel share one accumulator and are added in the order
    their paths finish, which depends on how the waves were scheduled: two runs agree to fp32 addition order (a few
    ulps of the per-batch sums), not bit for bit.  With one sample per batch they are bit-identical."""
    import torch

    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=128, height=128, samples_per_pixel=640)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    n = cfg.width * cfg.height * 3
    st = torch.cuda.current_stream().cuda_stream
    runs = []
    for _ in range(2):
        buf = torch.zeros(n, device="cuda")
        cam.render_accumulate(scene, buf.data_ptr(), 0, 640, seed=11, stream=st)
        runs.append(buf)
    torch.cuda.synchronize()
    a, b = runs
    scale = a.abs().clamp_min(1.0)
    assert float(((a - b).abs() / scale).max()) < 2e-5, "repeat runs differ by more than fp32 addition order"
    assert abs(float(a.double().sum()) - float(b.double().sum())) < 1e-6 * float(a.double().abs().sum())
    one = []
    for _ in range(2):
        buf = torch.zeros(n, device="cuda")
        cam.render_accumulate(scene, buf.data_ptr(), 0, 8, seed=11, stream=st)
        one.append(buf)
    torch.cuda.synchronize()
    assert torch.equal(one[0], one[1])
    # RTX_RENDER_REPRODUCIBLE: the same bits at any sample count (a wave finishes one item before it starts the next), and
    # the same frame as the default schedule up to the order of additions
    rep = []
    for _ in range(3):
        buf = torch.zeros(n, device="cuda")
        cam.render_accumulate(scene, buf.data_ptr(), 0, 640, seed=11, stream=st, reproducible=True)
        rep.append(buf)
    torch.cuda.synchronize()
    assert torch.equal(rep[0], rep[1]) and torch.equal(rep[0], rep[2])
    assert float(((rep[0] - a).abs() / scale).max()) < 2e-5


def test_sample_range_zero_zero_renders_nothing(render, scenes):
    """rtx_render_accumulate renders exactly the range it is given: (0, 0) is an empty range like any other, so a rank
    whose share of the samples is empty (spp < ranks) adds nothing."""
    import torch

    parallel = importlib.import_module("ray-tracer_amd.parallel")
    cfg, bvh = scenes.make("default", seed=0, width=64, height=40, samples_per_pixel=1)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    n = cfg.width * cfg.height * 3
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.full((n,), 3.0, device="cuda")
    cam.render_accumulate(scene, buf.data_ptr(), 0, 0, seed=1, stream=st)
    torch.cuda.synchronize()
    assert bool((buf == 3.0).all())
    single = torch.zeros(n, device="cuda")
    cam.render_accumulate(scene, single.data_ptr(), 0, 1, seed=1, stream=st)
    summed = torch.zeros(n, device="cuda")
    ranges = [parallel.sample_range(1, r, 2) for r in range(2)]
    assert ranges == [(0, 0), (0, 1)]
    for lo, hi in ranges:
        cam.render_accumulate(scene, summed.data_ptr(), lo, hi, seed=1, stream=st)
    torch.cuda.synchronize()
    assert torch.equal(single, summed)


# ------------------------------------------------------------------------------------------------ fused box test
def test_fused_box_test_on_axis_parallel_rays(render, oracle, scenes, capi):
    """The render kernels' box test (one fused multiply-add per plane) on rays with exactly-zero direction components,
    from origins away from the coordinate origin: same hits as the conservative test and as the oracle.  A zero
    component makes 1/d infinite; the boxes that straddle such a ray must not be culled."""
    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=64, height=36)
    scene, om = render.Scene(bvh), oracle.OracleMesh(bvh)
    rng = np.random.default_rng(9)
    n = 60_000
    rays = np.zeros(n, dtype=capi.RAY_DTYPE)
    rays["o"] = np.stack([rng.uniform(-11, 11, n), rng.uniform(0.05, 3.0, n), rng.uniform(-11, 11, n)], axis=-1).astype(F)
    axes = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (1, 0, 0), (-1, 0, 0), (0, -1, -1), (1, -1, 0), (-2, 0, 3), (0, -0.0, -1),
                     (-0.0, -1, 0.0)], dtype=F)
    rays["d"] = axes[rng.integers(0, len(axes), n)] * rng.uniform(0.5, 9.0, size=(n, 1)).astype(F)
    rays["time"] = rng.uniform(0, 1, n).astype(F)
    want = om.trace(rays)
    robust = scene.trace(rays, traversal=render.TRAVERSAL_FAST)
    fused = scene.trace(rays, traversal=render.TRAVERSAL_FAST_FUSED)
    assert fused.tobytes() == robust.tobytes()
    # the reference itself can lose a hit where its box test meets 0 * inf (mesh.rs:522-526); everywhere else all three agree
    same = fused["prim_id"] == want["prim_id"]
    assert np.mean(same) > 0.999, float(np.mean(same))
    assert np.array_equal(fused["t"][same & (want["prim_id"] != MISS)], want["t"][same & (want["prim_id"] != MISS)])
    assert np.count_nonzero(want["prim_id"] != MISS) > n // 4
    # general rays as well: primary and surface-origin secondary rays of two more scenes
    for name, seed in (("quads", 0), ("triangle-mesh", 1)):
        cfg, bvh = scenes.make(name, seed=seed, width=160, height=90)
        scene, om = render.Scene(bvh), oracle.OracleMesh(bvh)
        prim = oracle.OracleCamera(cfg).center_rays(0.5)
        hits = om.trace(prim)
        both = np.concatenate([prim, _surface_rays(capi, hits, 20_000, 4)])
        assert scene.trace(both, traversal=render.TRAVERSAL_FAST_FUSED).tobytes() == scene.trace(both, traversal=render.TRAVERSAL_FAST).tobytes()


# ------------------------------------------------------------------------------------------------ several devices, one call
@pytest.mark.parametrize("name,seed", [("default", 0), ("cornell-box", 0)])
def test_render_multi_equals_single_device(render, scenes, name, seed):
    """rtx_render_multi with three shares (all on GPU 0: the multi-GPU path on a one-GPU machine).  Sample split: the
    frame equals the single-device frame up to fp32 addition order; tile split: bit for bit, through the fp32 gather
    (accum requested) and through the RGB8 gather (bytes only)."""
    cfg, bvh = scenes.make(name, seed=seed, width=203, height=117, samples_per_pixel=12)  # ragged tiles, one sample per batch
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    rgb1, acc1, st1 = cam.render(scene, seed=5, want_accum=True, want_stats=True)
    rgb_s, acc_s, st_s = cam.render(scene, seed=5, want_accum=True, want_stats=True, devices=[0, 0, 0], split=render.SPLIT_SAMPLES)
    assert np.allclose(acc_s, acc1, rtol=1e-5, atol=1e-5)
    d = np.abs(rgb_s.astype(np.int32) - rgb1.astype(np.int32))
    assert d.max() <= 1 and np.mean(d != 0) < 0.01
    assert st_s.samples == st1.samples and st_s.rays == st1.rays and st_s.n_launches == 3
    rgb_t, acc_t = cam.render(scene, seed=5, want_accum=True, devices=[0, 0, 0], split=render.SPLIT_TILES)
    assert np.array_equal(acc_t, acc1) and np.array_equal(rgb_t, rgb1)
    rgb_b = cam.render(scene, seed=5, devices=[0, 0, 0], split=render.SPLIT_TILES)  # owners resolve, only bytes travel
    assert np.array_equal(rgb_b, rgb1)
    # more shares than samples: the empty shares add nothing
    cam1 = render.Camera(cfg.replace(samples_per_pixel=2))
    a = cam1.render(scene, seed=5, want_accum=True)[1]
    b = cam1.render(scene, seed=5, want_accum=True, devices=[0, 0, 0, 0], split=render.SPLIT_SAMPLES)[1]
    assert np.allclose(a, b, rtol=1e-5, atol=1e-5)
    # n_gpus in the options of the plain call; a device that does not exist is an error, not a crash
    if render.device_count() >= 2:
        assert np.allclose(cam.render(scene, seed=5, want_accum=True, n_gpus=2)[1], acc1, rtol=1e-5, atol=1e-5)
    with pytest.raises(render.RtxError) as e:
        cam.render(scene, seed=5, devices=[0, 63])
    assert e.value.code == render.RTX_ERR_NO_DEVICE


def test_render_multi_follows_the_camera(render, scenes):
    """The fast data is ordered for a camera once, on the scene's own device, and the other shares' copies are brought up
    to date device to device: a second camera on the far side of the scene (and farther out: its boxes are enlarged), then
    the first again, each time the single-device frame -- and a trace afterwards still finds the same hits."""
    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=160, height=96, samples_per_pixel=6)
    scene = render.Scene(bvh)
    cams = [render.Camera(cfg), render.Camera(cfg.replace(lookfrom=tuple(-3.0 * float(c) for c in cfg.lookfrom))), render.Camera(cfg)]
    for k, cam in enumerate(cams):
        single = cam.render(scene, seed=8, want_accum=True)[1]
        rgb_t, acc_t, st = cam.render(scene, seed=8, want_accum=True, want_stats=True, devices=[0, 0, 0], split=render.SPLIT_TILES)
        assert np.array_equal(acc_t, single), k  # one sample per batch: bit for bit
        assert st.flags & 8 == 0  # every share sits on the gathering device
        acc_s = cam.render(scene, seed=8, want_accum=True, devices=[0, 0], split=render.SPLIT_SAMPLES)[1]
        assert np.allclose(acc_s, single, rtol=1e-5, atol=1e-5), k


def test_work_counters_and_stats(render, scenes):
    """The kernel's own test counts come back with the stats: plausible, repeatable, and they respond to the traversal."""
    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=160, height=90, samples_per_pixel=8)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    _, f1 = cam.render(scene, seed=3, want_stats=True)
    _, f2 = cam.render(scene, seed=3, want_stats=True)
    _, s = cam.render(scene, seed=3, traversal=render.TRAVERSAL_STRICT, want_stats=True)
    assert f1.samples == 160 * 90 * 8 and f1.rays > f1.samples
    assert (f1.rays, f1.box_tests, f1.prim_tests) == (f2.rays, f2.box_tests, f2.prim_tests)
    assert 5 < f1.box_tests / f1.rays < 60 and 0.5 < f1.prim_tests / f1.rays < 10
    assert abs(s.rays - f1.rays) < 0.01 * f1.rays and s.box_tests > 2 * f1.box_tests  # the reference's origin-including boxes cull far less
    assert f1.flags & 1  # a few hundred spheres: entries resident in LDS


# ------------------------------------------------------------------------------------------------ deep instancing
def _nested_scene(scene):
    """Five nested Translate / RotateY levels around a cube and four around a sphere (the reference nests them freely,
    mesh.rs:324-432), beside plain primitives."""
    white = scene.Lambertian.rgb(0.73, 0.73, 0.73)
    cube = scene.Cube((0, 0, 0), (1.2, 1.6, 1.2), white)
    for k, (angle, off) in enumerate([(15.0, (0.3, 0.0, -0.2)), (-28.0, (-0.4, 0.1, 0.5)), (40.0, (0.6, 0.0, 0.1))]):
        cube = scene.Translate(scene.RotateY(cube, angle), off) if k < 2 else scene.RotateY(cube, angle)  # T(R(T(R(R(cube))))) = 5 levels
    ball = scene.Sphere.stationary((0.0, 0.5, 0.0), 0.5, scene.Metal.rgb(0.8, 0.7, 0.6).with_roughness(0.05))
    ball = scene.Translate(scene.RotateY(scene.Translate(scene.RotateY(ball, 33.0), (1.0, 0.0, 0.0)), -70.0), (-2.0, 0.0, 1.0))
    w = scene.World()
    w.push(scene.Sphere.stationary((0.0, -1000.0, 0.0), 1000.0, scene.Lambertian.rgb(0.5, 0.6, 0.5)))
    w.push(cube)
    w.push(ball)
    w.push(scene.Sphere.stationary((2.2, 0.6, 0.4), 0.6, scene.Dielectric(1.5)))
    cfg = scene.CameraConfig(width=160, height=120, samples_per_pixel=8, max_depth=20, vfov=35.0, lookfrom=(6.0, 3.0, 7.0),
                             lookat=(0.0, 0.7, 0.0), vup=(0, 1, 0))
    return cfg, scene.Bvh(w)


def test_deeply_nested_transforms_match_oracle(render, oracle, scene, capi):
    cfg, bvh = _nested_scene(scene)
    om, gs = oracle.OracleMesh(bvh), render.Scene(bvh)
    prim = oracle.OracleCamera(cfg).center_rays(0.5)
    want = om.trace(prim)
    assert len(set(want["prim_id"][want["prim_id"] != MISS].tolist())) >= 5  # ground, cube faces, both wrapped objects
    rays = np.concatenate([prim, _surface_rays(capi, want, 30_000, 2)])
    want = om.trace(rays)
    got = gs.trace(rays, traversal=render.TRAVERSAL_STRICT)
    assert np.array_equal(got["prim_id"], want["prim_id"])
    hit = want["prim_id"] != MISS
    for f in ("t", "n", "p"):  # the wrong back-rotation of RotateY (mesh.rs:378-382) is applied once per level, as the reference does
        assert np.array_equal(got[f][hit], want[f][hit]), f
    for mode in (render.TRAVERSAL_FAST, render.TRAVERSAL_FAST_FUSED):
        fast = gs.trace(rays, traversal=mode)
        assert np.count_nonzero(fast["prim_id"] != want["prim_id"]) <= 2
    ref, _ = oracle.OracleCamera(cfg).render(om, seed=5, rng_kind=1)
    _, acc = render.Camera(cfg).render(gs, seed=5, traversal=render.TRAVERSAL_STRICT, want_accum=True)
    rel = np.abs(acc - ref) / (np.abs(ref) + 1e-2 * cfg.samples_per_pixel)
    assert np.median(rel) < 1e-3 and np.mean(rel < 2e-2) > 0.95


def _placeable_instances_scene(scene):
    """Instances one and two levels deep over primitives at rest (a cube, a sphere, a list of two quads): the fast array
    holds their primitives in the scene's own tree instead of transform brackets."""
    white = scene.Lambertian.rgb(0.73, 0.73, 0.73)
    w = scene.World()
    w.push(scene.Sphere.stationary((0.0, -1000.0, 0.0), 1000.0, scene.Lambertian.rgb(0.5, 0.6, 0.5)))
    w.push(scene.RotateY(scene.Cube((0, 0, 0), (1.0, 1.4, 1.0), white), 25.0))
    w.push(scene.Translate(scene.RotateY(scene.Sphere.stationary((0.0, 0.5, 0.0), 0.5, scene.Metal.rgb(0.8, 0.7, 0.6).with_roughness(0.05)), 33.0), (-2.0, 0.0, 1.0)))
    w.push(scene.Translate(scene.Cube((0, 0, 0), (0.7, 0.7, 0.7), scene.Lambertian.rgb(0.7, 0.2, 0.2)), (2.0, 0.0, -1.0)))
    w.push(scene.Sphere.stationary((2.2, 0.6, 1.4), 0.6, scene.Dielectric(1.5)))
    cfg = scene.CameraConfig(width=160, height=120, samples_per_pixel=8, max_depth=20, vfov=35.0, lookfrom=(6.0, 3.0, 7.0),
                             lookat=(0.0, 0.7, 0.0), vup=(0, 1, 0))
    return cfg, scene.Bvh(w)


def test_placed_instance_primitives_match_oracle(render, oracle, scene, capi):
    """The walks over placed instance primitives (exact, and the render kernels' own) return the oracle's records -- ids,
    t, normal and point bit for bit, the reference's back-rotation of RotateY included -- and a fast render of the
    mirror stream is the oracle's frame."""
    cfg, bvh = _placeable_instances_scene(scene)
    om, gs = oracle.OracleMesh(bvh), render.Scene(bvh)
    prim = oracle.OracleCamera(cfg).center_rays(0.5)
    want = om.trace(prim)
    assert len(set(want["prim_id"][want["prim_id"] != MISS].tolist())) >= 8
    rays = np.concatenate([prim, _surface_rays(capi, want, 30_000, 4)])
    want = om.trace(rays)
    hit = want["prim_id"] != MISS
    for mode in (render.TRAVERSAL_STRICT, render.TRAVERSAL_FAST, render.TRAVERSAL_RENDER):
        got = gs.trace(rays, traversal=mode)
        assert np.count_nonzero(got["prim_id"] != want["prim_id"]) <= (0 if mode == render.TRAVERSAL_STRICT else 2), mode
        same = hit & (got["prim_id"] == want["prim_id"])
        for f in ("t", "n", "p"):
            if mode == render.TRAVERSAL_RENDER:
                assert np.mean(np.all(np.atleast_2d(got[f][same].T == want[f][same].T), axis=0)) > 0.9999, (mode, f)
            else:
                assert np.array_equal(got[f][same], want[f][same]), (mode, f)
    ref, _ = oracle.OracleCamera(cfg).render(om, seed=5, rng_kind=1)
    _, acc = render.Camera(cfg).render(gs, seed=5, traversal=render.TRAVERSAL_FAST, want_accum=True)
    rel = np.abs(acc - ref) / (np.abs(ref) + 1e-2 * cfg.samples_per_pixel)
    assert np.median(rel) < 1e-3 and np.mean(rel < 2e-2) > 0.95
