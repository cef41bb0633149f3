"""The arithmetic the render kernels run, gated ray by ray.

`rtx_trace` with RTX_TRAVERSAL_RENDER walks the data a fast render walks through the very device functions
`render_kernel` calls (`pk_render_box`, `pk_render_prim`, `pk_render_hit`; `sphere_test` / `planar_test` with the hardware
square root and reciprocal and `make_hit_x` with a reciprocal radius for scenes with transforms or media).  It is compared
with the oracle -- the restatement of Sphere::hit / Quad::hit / Hit::new (mesh.rs:113-139, :202-221, :33-51) -- or, where
the oracle is too slow or draws differently (media), with the library's exact walks, which are bit-equal to the oracle
wherever the two were compared.

Measured (MI355X, every case below, ~9 M rays): hit ids, t, point and normal are BIT-EQUAL to the oracle's on every ray.
The render walk takes its square roots and quotients through the hardware's reciprocal square root / reciprocal plus one
Newton step on an exactly computed residual (`quick_sqrt`, `quick_div`), which lands on the correctly rounded value for
all but a vanishing share of operands -- but is not guaranteed to.  `check_render_walk` therefore asserts two things:
at least 99.99 % of the hit distances and normals bit-equal, and for the rest the bounds below, which say what one unit
in the last place of a root may do.  What the reference's own expression magnifies, the bound has to allow for: a
sphere's root is (h - sqrt(disc)) / a, a difference of two numbers of the size |origin - centre| * |d|, so one ulp of the
square root moves the hit POINT by an ulp of |origin - centre| whatever t is (a camera two units above a sphere of radius
1000; without the Newton step 1-2 % of surface-origin rays were beyond 1e-5 * t for this reason).  With
    E = 1e-5 * |d| * t  +  4 ulp * (largest |coordinate| of the primitive + of the ray origin)
* |t - t_ref| * |d| <= E and |p - p_ref| <= E (+ 2 ulp of p);
* normals within 1e-5 + 8 ulp * (coordinates + segment length) / radius -- a sphere's normal is (p - c) / r, which magnifies
  the last place of p by |p| / r (a sphere of radius 0.2 at x = 11) -- the sign aside where the ray grazes the surface
  (Hit::new's flip, mesh.rs:40-45, is decided by the sign of a rounded zero there; counted);
* sphere uv within 1e-5 + (the normal's bound beyond 1e-5, at least 1e-6) / sqrt(1 - n.y^2): acos and atan2
  (mesh.rs:182-187) magnify the last place of the normal near the poles, where the reference returns NaN as soon as rounding pushes |n.y| past 1 (u compared across its 0 / 1 seam);
* quad / triangle uv, which is the hit point in units of the primitive's edges (mesh.rs:216-217): within 1e-5 + E / edge;
* ids equal, except for NEAR TIES (both sides hit, within E of each other) and EDGE HITS (the hit one side reports and the
  other does not lies within the bounds above of the edge of its primitive's domain -- alpha, beta at 0 / 1, a triangle's
  alpha + beta at 1, a root at the interval start, which can also leave one side with the far root of the same sphere):
  every differing ray must be one of the two, and they are counted.  Media draw a random free-flight distance per visit
  (mesh.rs:659), so one differing decision shifts the draws of the ray's later visits: scenes with media are compared with
  the library's exact fast walk (same visiting order) and a differing ray that involves a medium counts as an edge hit.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32
MISS = 0xFFFFFFFF
ULP = 2.0 ** -23


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


class PrimInfo:
    """Per prim id, from the flattened scene: kind (0 sphere, 1 quad, 2 triangle, 3 medium), the largest |coordinate| the
    primitive reaches, and its size (radius / shortest edge)."""

    def __init__(self, scene):
        d = scene.desc

        def arr(ptr, n, dtype=np.float32):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float if dtype == np.float32 else C.c_uint32)), shape=(n,)).copy() if n else np.zeros(0, dtype)

        ids, kind, cmag, size = [], [], [], []
        if d.n_spheres:
            n = d.n_spheres
            c0 = np.stack([arr(d.sph_cx, n), arr(d.sph_cy, n), arr(d.sph_cz, n)], axis=1).astype(np.float64)
            dc = np.stack([arr(d.sph_dx, n), arr(d.sph_dy, n), arr(d.sph_dz, n)], axis=1).astype(np.float64)
            r = arr(d.sph_radius, n).astype(np.float64)
            ids.append(arr(d.sph_prim_id, n, np.uint32))
            kind.append(np.zeros(n, np.int8))
            cmag.append(np.maximum(np.abs(c0), np.abs(c0 + dc)).max(axis=1) + r)
            size.append(r)
        if d.n_quads:
            n = d.n_quads
            q = np.stack([arr(d.quad_f[k], n) for k in range(9)], axis=1).astype(np.float64)
            Q, u, v = q[:, 0:3], q[:, 3:6], q[:, 6:9]
            ids.append(arr(d.quad_prim_id, n, np.uint32))
            kind.append(np.ones(n, np.int8))
            cmag.append(np.max([np.abs(Q), np.abs(Q + u), np.abs(Q + v), np.abs(Q + u + v)], axis=0).max(axis=1))
            size.append(np.minimum(np.linalg.norm(u, axis=1), np.linalg.norm(v, axis=1)))
        if d.n_tris:
            n = d.n_tris
            t = np.stack([arr(d.tri_f[k], n) for k in range(9)], axis=1).astype(np.float64)
            ids.append(arr(d.tri_prim_id, n, np.uint32))
            kind.append(np.full(n, 2, np.int8))
            cmag.append(np.abs(t).max(axis=1))
            size.append(np.minimum(np.linalg.norm(t[:, 3:6] - t[:, 0:3], axis=1), np.linalg.norm(t[:, 6:9] - t[:, 0:3], axis=1)))
        ids, kind, cmag, size = (np.concatenate(x) for x in (ids, kind, cmag, size))
        top = int(ids.max()) + 1
        for i in range(d.n_objects):
            if d.objects[i].kind == 3:
                top = max(top, d.objects[i].prim_id + 1)
        self.kind = np.full(top, 3, np.int8)  # whatever is not a primitive is a medium
        self.cmag = np.full(top, float(cmag.max()))
        self.size = np.full(top, 1.0)
        self.kind[ids], self.cmag[ids], self.size[ids] = kind, cmag, size


def check_render_walk(got, want, rays, info, label, max_id_frac=5e-5, tmin=0.001, tmax=np.inf):
    """Asserts the bounds of the module docstring; returns the counts for the log."""
    n = len(want)
    o, d = rays["o"].astype(np.float64), rays["d"].astype(np.float64)
    dlen = np.linalg.norm(d, axis=1)
    omag = np.abs(o).max(axis=1)
    hit_w, hit_g = want["prim_id"] != MISS, got["prim_id"] != MISS

    def err_bound(h, sel):  # E of the docstring for the hits h[sel]
        pid = h["prim_id"][sel]
        return 1e-5 * dlen[sel] * np.abs(h["t"][sel].astype(np.float64)) + 4 * ULP * (info.cmag[pid] + omag[sel])

    def at_edge(h, sel):  # the hits h[sel] lie on the edge of their primitive's domain, or at an end of the interval
        pid = h["prim_id"][sel]
        k, E = info.kind[pid], err_bound(h, sel)
        t = h["t"][sel].astype(np.float64)
        a, b = h["uv"][sel][:, 0].astype(np.float64), h["uv"][sel][:, 1].astype(np.float64)
        tol = 1e-5 + E / info.size[pid]
        quad = np.minimum.reduce([np.abs(a), np.abs(b), np.abs(1 - a), np.abs(1 - b)]) <= tol
        tri = np.minimum.reduce([np.abs(a), np.abs(b), np.abs(1 - a - b)]) <= 2 * tol
        ends = (np.abs(t - tmin) * dlen[sel] <= E) | (np.abs(t - tmax) * dlen[sel] <= E if np.isfinite(tmax) else False)
        return ends | ((k == 1) & quad) | ((k == 2) & tri) | (k == 3)

    differ = got["prim_id"] != want["prim_id"]
    n_differ = int(np.count_nonzero(differ))
    both = np.flatnonzero(differ & hit_w & hit_g)
    tg, tw = got["t"][both].astype(np.float64), want["t"][both].astype(np.float64)
    tie = np.abs(tg - tw) * dlen[both] <= np.maximum(err_bound(got, both), err_bound(want, both))
    # not a tie: the nearer of the two hits is the one the other side did not accept
    g_near = tg < tw
    edge_both = np.where(g_near, at_edge(got, both), at_edge(want, both))
    assert np.all(tie | edge_both), f"{label}: {np.count_nonzero(~(tie | edge_both))} rays hit different primitives, neither a near tie nor an edge hit"
    lone_w, lone_g = np.flatnonzero(differ & hit_w & ~hit_g), np.flatnonzero(differ & ~hit_w & hit_g)
    e_w, e_g = at_edge(want, lone_w), at_edge(got, lone_g)
    assert np.all(e_w) and np.all(e_g), f"{label}: {np.count_nonzero(~e_w) + np.count_nonzero(~e_g)} rays hit on one side only, away from any edge"
    assert n_differ <= max(3, max_id_frac * n), f"{label}: {n_differ} of {n} hit ids differ"

    same = np.flatnonzero(~differ & hit_w)
    E = err_bound(want, same)
    t_g, t_w = got["t"][same].astype(np.float64), want["t"][same].astype(np.float64)
    dt = np.abs(t_g - t_w) * dlen[same]
    # the same sphere, the other root: the near root sits at the interval start, accepted on one side only (an edge hit)
    other_root = (dt > E) & (info.kind[want["prim_id"][same]] == 0) & (np.abs(np.minimum(t_g, t_w) - tmin) * dlen[same] <= E)
    n_other = int(np.count_nonzero(other_root))
    assert n_other <= max(3, max_id_frac * n), f"{label}: {n_other} rays take the other root of their sphere"
    same = same[~other_root]
    pid = want["prim_id"][same]
    E, dt, t_w = E[~other_root], dt[~other_root], t_w[~other_root]
    assert np.all(dt <= E), f"{label}: t beyond the bound on {np.count_nonzero(dt > E)} rays (worst {np.max(dt / E):.3g} x)"
    rel_t = np.abs(got["t"][same] - t_w) / np.abs(t_w)
    dp = np.abs(got["p"][same].astype(np.float64) - want["p"][same]).max(axis=1)
    Ep = E + 2 * ULP * np.abs(want["p"][same]).max(axis=1)
    assert np.all(dp <= Ep), f"{label}: point beyond the bound on {np.count_nonzero(dp > Ep)} rays (worst {np.max(dp / Ep):.3g} x)"
    nw, ng = want["n"][same].astype(np.float64), got["n"][same].astype(np.float64)
    dn = np.abs(ng - nw).max(axis=1)
    # a sphere's normal is (p - c) / r: the last place of p (and of t, through p) is magnified by |p| / r
    tol_n = 1e-5 + np.where(info.kind[pid] == 0, 8 * ULP * (info.cmag[pid] + omag[same] + dlen[same] * t_w) / info.size[pid], 0.0)
    grazing = np.abs(np.einsum("ij,ij->i", d[same], nw)) <= 2 * tol_n * dlen[same]
    flipped = grazing & (np.abs(ng + nw).max(axis=1) <= tol_n) & (dn > tol_n)
    dn = np.where(flipped, 0.0, dn)
    assert np.all(dn <= tol_n), f"{label}: normal beyond the bound on {np.count_nonzero(dn > tol_n)} rays (worst {np.max(dn):.3g})"
    n_flip = int(np.count_nonzero(flipped))
    assert n_flip <= max(2, 3e-5 * n), f"{label}: {n_flip} normals flipped at grazing incidence"
    gu, wu = got["uv"][same].astype(np.float64), want["uv"][same].astype(np.float64)
    k = info.kind[pid]
    sph = k == 0
    du = np.abs(gu[:, 0] - wu[:, 0])
    du = np.where(sph, np.minimum(du, np.abs(1.0 - du)), du)  # u = 0 and u = 1 are the same meridian (mesh.rs:185-186)
    duv = np.maximum(du, np.abs(gu[:, 1] - wu[:, 1]))
    ny2 = np.minimum(nw[:, 1] ** 2, 1.0)
    tol = np.where(sph, 1e-5 + np.maximum(1e-6, tol_n - 1e-5) / np.sqrt(np.maximum(1.0 - ny2, 1e-14)), 1e-5 + Ep / info.size[pid])
    nan_g, nan_w = np.isnan(gu).any(axis=1), np.isnan(wu).any(axis=1)
    # a NaN (|n.y| rounded past 1) on one side only: legitimate within 4 ulp of the pole
    nan_ok = (nan_g | nan_w) & ((nan_g & nan_w) | (1.0 - np.abs(nw[:, 1]) <= 4 * ULP))
    with np.errstate(invalid="ignore"):
        bad_uv = ~(duv <= tol) & ~nan_ok & (k != 3)
    assert not bad_uv.any(), f"{label}: uv beyond the bound on {np.count_nonzero(bad_uv)} rays"
    assert np.all(got["front_face"][hit_g] == 1)
    stats = dict(rays=n, hits=int(np.count_nonzero(hit_w)), ids_differ=n_differ, near_ties=int(np.count_nonzero(tie)),
                 edge_hits=n_differ - int(np.count_nonzero(tie)) + n_other, grazing_flips=n_flip,
                 t_bits_equal=round(float(np.mean(got["t"][same] == want["t"][same])), 6) if len(same) else 1.0,
                 n_bits_equal=round(float(np.mean(np.all(got["n"][same] == want["n"][same], axis=1))), 6) if len(same) else 1.0,
                 t_rel_median=float(np.median(rel_t)) if len(same) else 0.0, t_rel_p9999=float(np.quantile(rel_t, 0.9999)) if len(same) else 0.0,
                 t_rel_max=float(rel_t.max(initial=0.0)), within_1e5=round(float(np.mean(rel_t <= 1e-5)), 6) if len(same) else 1.0,
                 n_err_max=float(dn.max(initial=0.0)), n_within_1e5=round(float(np.mean(dn <= 1e-5)), 6) if len(same) else 1.0,
                 uv_within_1e5=round(float(np.mean((duv <= 1e-5) | (nan_g & nan_w) | (k == 3))), 6) if len(same) else 1.0)
    print(f"[render walk] {label}: {stats}")
    assert stats["t_bits_equal"] >= 0.9999, f"{label}: only {stats['t_bits_equal']:.6f} of the hit distances are bit-equal"
    assert stats["n_bits_equal"] >= 0.9999, f"{label}: only {stats['n_bits_equal']:.6f} of the normals are bit-equal"
    return stats


# ------------------------------------------------------------------------------------------------ C2: array in LDS, float boxes
def test_render_walk_benchmark_frame_matches_oracle(render, oracle, scenes, capi):
    """Every pixel-centre ray of the 1920x1080 bouncing-spheres frame and 100 000 surface-origin rays, with the array in
    its build order and again after a render ordered it for the camera (what the benchmark's kernel walks)."""
    import torch

    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=1920, height=1080, samples_per_pixel=1)
    rays = oracle.OracleCamera(cfg).center_rays(0.5)
    om = oracle.OracleMesh(bvh)
    want = om.trace(rays)
    sec = _surface_rays(capi, want, 100_000, 21)
    want_sec = om.trace(sec)
    scene = render.Scene(bvh)
    info = PrimInfo(scene)
    check_render_walk(scene.trace(rays, traversal=render.TRAVERSAL_RENDER), want, rays, info, "C2 primary, build order")
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    render.Camera(cfg).render_accumulate(scene, buf.data_ptr(), 0, 1, seed=1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = scene.trace(rays, traversal=render.TRAVERSAL_RENDER)
    st = check_render_walk(got, want, rays, info, "C2 primary, camera order")
    assert st["hits"] > 1_000_000
    check_render_walk(scene.trace(sec, traversal=render.TRAVERSAL_RENDER), want_sec, sec, info, "C2 secondary")
    # and against the exact fast walk on the same device data: same ids wherever the oracle agreed
    fast = scene.trace(rays, traversal=render.TRAVERSAL_FAST)
    assert np.count_nonzero(fast["prim_id"] != got["prim_id"]) <= 3


@pytest.mark.parametrize("name,seed,w,h", [("default", 0, 400, 225), ("earth", 0, 400, 225), ("quads", 0, 300, 300), ("checkered-spheres", 0, 320, 240),
                                           ("perlin-spheres", 1, 200, 150), ("simple-light", 1, 320, 240), ("mixed", 2, 320, 180)])
@pytest.mark.parametrize("half", [False, True])
def test_render_walk_small_scenes(render, oracle, scenes, capi, name, seed, w, h, half, monkeypatch):
    """Primary and surface-origin rays of the other bracket-free scenes, after a render from the scene's camera (quads,
    earth and simple-light look at the scene from outside its coordinate range: the boxes are enlarged for that), in both
    encodings of the packed array: float boxes (as rendered from LDS) and half-precision boxes (as large scenes are
    rendered from HBM; forced here through RTX_PACKED_HALF so that it meets the oracle on many rays)."""
    import torch

    if half:
        monkeypatch.setenv("RTX_PACKED_HALF", "1")
    cfg, bvh = scenes.make(name, seed=seed, width=w, height=h, samples_per_pixel=1)
    om = oracle.OracleMesh(bvh)
    rays = oracle.OracleCamera(cfg).center_rays(0.5)
    want = om.trace(rays)
    sec = _surface_rays(capi, want, 50_000, 6)
    scene = render.Scene(bvh)
    info = PrimInfo(scene)
    buf = torch.zeros(w * h * 3, device="cuda")
    st = render.Stats()
    render.Camera(cfg).render_accumulate(scene, buf.data_ptr(), 0, 1, seed=1, stream=torch.cuda.current_stream().cuda_stream, stats=st)
    assert bool(st.flags & 2) and (not half or not bool(st.flags & 1))  # packed array; walked from HBM when half boxes are forced
    check_render_walk(scene.trace(rays, traversal=render.TRAVERSAL_RENDER), want, rays, info, f"{name} primary half={half}")
    check_render_walk(scene.trace(sec, traversal=render.TRAVERSAL_RENDER), om.trace(sec), sec, info, f"{name} secondary half={half}")


def test_render_walk_array_near_the_lds_limit(render, oracle, scene, capi):
    """A packed array of ~66 KB (700 spheres): still rendered from LDS, above the 48 KB a kernel gets without asking --
    the render walk's LDS copy (trace) and the render itself both have to request it."""
    import torch

    rng = np.random.default_rng(4)
    w = scene.World()
    w.push(scene.Sphere.stationary((0.0, -1000.0, 0.0), 1000.0, scene.Lambertian.rgb(0.5, 0.5, 0.5)))
    for k in range(699):
        c = (float(rng.uniform(-12, 12)), float(rng.uniform(0.15, 2.5)), float(rng.uniform(-12, 12)))
        w.push(scene.Sphere.stationary(c, float(rng.uniform(0.1, 0.3)), scene.Lambertian.rgb(0.6, 0.4, 0.3)))
    cfg = scene.CameraConfig(width=320, height=200, samples_per_pixel=1, max_depth=8, vfov=30.0, lookfrom=(13.0, 3.0, 4.0), lookat=(0.0, 0.5, 0.0),
                             vup=(0, 1, 0))
    bvh = scene.Bvh(w)
    om, gs = oracle.OracleMesh(bvh), render.Scene(bvh)
    info = PrimInfo(gs)
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    st = render.Stats()
    render.Camera(cfg).render_accumulate(gs, buf.data_ptr(), 0, 1, seed=1, stream=torch.cuda.current_stream().cuda_stream, stats=st)
    torch.cuda.synchronize()
    assert (st.flags & 3) == 3  # packed array, walked from LDS
    slots, _ = gs.traversal_dump(render.TRAVERSAL_FAST_FUSED)
    assert 60 * 1024 < slots.shape[0] * 16 <= 72 * 1024
    rays = oracle.OracleCamera(cfg).center_rays(0.5)
    want = om.trace(rays)
    check_render_walk(gs.trace(rays, traversal=render.TRAVERSAL_RENDER), want, rays, info, "700 spheres primary")
    sec = _surface_rays(capi, want, 40_000, 9)
    check_render_walk(gs.trace(sec, traversal=render.TRAVERSAL_RENDER), om.trace(sec), sec, info, "700 spheres secondary")


@pytest.mark.parametrize("half", [False, True])
def test_render_walk_axis_parallel_and_interval(render, oracle, scenes, capi, half, monkeypatch):
    """Rays with exactly-zero direction components (1/d infinite: the scaled, clamped box test must not cull a box that
    straddles the ray), from inside the scene; and explicit intervals, whose start the scaled box test replaces by 0."""
    if half:
        monkeypatch.setenv("RTX_PACKED_HALF", "1")
    cfg, bvh = scenes.make("bouncing-spheres", seed=1, width=64, height=36)
    scene, om = render.Scene(bvh), oracle.OracleMesh(bvh)
    info = PrimInfo(scene)
    rng = np.random.default_rng(9)
    n = 60_000
    rays = np.zeros(n, dtype=capi.RAY_DTYPE)
    rays["o"] = np.stack([rng.uniform(-11, 11, n), rng.uniform(0.05, 3.0, n), rng.uniform(-11, 11, n)], axis=-1).astype(F)
    axes = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (1, 0, 0), (-1, 0, 0), (0, -1, -1), (1, -1, 0), (-2, 0, 3), (0, -0.0, -1),
                     (-0.0, -1, 0.0)], dtype=F)
    rays["d"] = axes[rng.integers(0, len(axes), n)] * rng.uniform(0.5, 9.0, size=(n, 1)).astype(F)
    rays["time"] = rng.uniform(0, 1, n).astype(F)
    want = om.trace(rays)
    got = scene.trace(rays, traversal=render.TRAVERSAL_RENDER)
    fast = scene.trace(rays, traversal=render.TRAVERSAL_FAST)
    # the reference loses a hit where its own box test meets 0 * inf (mesh.rs:522-526): compare with the exact fast walk
    # everywhere, and with the oracle wherever that one agrees with it
    check_render_walk(got, fast, rays, info, f"axis-parallel vs fast walk half={half}")
    agree = fast["prim_id"] == want["prim_id"]
    assert np.mean(agree) > 0.999
    check_render_walk(got[agree], want[agree], rays[agree], info, f"axis-parallel vs oracle half={half}")
    general = _surface_rays(capi, want, 20_000, 12)
    for tmin, tmax in [(0.001, np.inf), (0.5, 30.0), (2.0, 4.0), (0.01, 1.0)]:
        w2 = om.trace(general, tmin=tmin, tmax=tmax)
        g2 = scene.trace(general, tmin=tmin, tmax=tmax, traversal=render.TRAVERSAL_RENDER)
        check_render_walk(g2, w2, general, info, f"interval [{tmin}, {tmax}) half={half}", tmin=tmin, tmax=tmax)


# ------------------------------------------------------------------------------------------------ C4 / C5: array in HBM, half boxes
def test_render_walk_million_triangles_at_1080p(render, oracle, scenes, capi):
    """The million-triangle configuration under its 1920x1080 camera, after a render: the render walk against the exact
    fast walk on every primary ray of the frame, and against the oracle on ~4 000 primary and 3 000 surface-origin rays."""
    import torch

    cfg, bvh = scenes.make("triangle-mesh", seed=1, world_args=dict(n_lat=500), width=1920, height=1080, samples_per_pixel=1)
    scene = render.Scene(bvh)
    info = PrimInfo(scene)
    assert scene.desc.n_tris == 998000
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    st = render.Stats()
    render.Camera(cfg).render_accumulate(scene, buf.data_ptr(), 0, 1, seed=1, stream=torch.cuda.current_stream().cuda_stream, stats=st)
    assert bool(st.flags & 2) and not bool(st.flags & 1)  # packed array walked from HBM
    rays = oracle.OracleCamera(cfg).center_rays(0.5)
    fast = scene.trace(rays, traversal=render.TRAVERSAL_FAST)
    got = scene.trace(rays, traversal=render.TRAVERSAL_RENDER)
    s = check_render_walk(got, fast, rays, info, "C4 frame vs exact fast walk")
    assert s["hits"] > 1_000_000
    sec_all = _surface_rays(capi, fast, 300_000, 8)
    check_render_walk(scene.trace(sec_all, traversal=render.TRAVERSAL_RENDER), scene.trace(sec_all, traversal=render.TRAVERSAL_FAST), sec_all, info, "C4 secondary vs exact fast walk")
    om = oracle.OracleMesh(bvh)
    sub = np.concatenate([rays[np.arange(0, len(rays), 509)], sec_all[:3000]])
    check_render_walk(scene.trace(sub, traversal=render.TRAVERSAL_RENDER), om.trace(sub), sub, info, "C4 subset vs oracle")


def test_render_walk_ten_million_triangles(render, scenes, capi):
    """The mixed configuration at its stated size (10 M triangles + spheres under the 3840x2160 camera), where
    half-precision boxes are coarsest relative to the triangles: 200 000 primary and 100 000 surface-origin rays against
    the reference-order walk of the same library (bit-exact with the oracle wherever it was compared)."""
    import torch

    cfg, bvh = scenes.make("mixed", seed=1, world_args=dict(n_lat=1582), width=3840, height=2160, samples_per_pixel=1)
    scene, cam = render.Scene(bvh), render.Camera(cfg)
    info = PrimInfo(scene)
    assert scene.desc.n_tris > 10_000_000
    k = cam.constants()
    rng = np.random.default_rng(3)
    xs, ys = rng.integers(0, cfg.width, 200_000), rng.integers(0, cfg.height, 200_000)
    rays = np.zeros(len(xs), dtype=capi.RAY_DTYPE)
    rays["o"] = k["center"]
    rays["d"] = k["pixel00_loc"] + xs[:, None].astype(F) * k["pixel_delta_u"] + ys[:, None].astype(F) * k["pixel_delta_v"] - k["center"]
    rays["time"] = 0.5
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    cam.render_accumulate(scene, buf.data_ptr(), 0, 1, seed=2, stream=torch.cuda.current_stream().cuda_stream,
                          tile_rank=0, tile_world=64)  # orders the array for the camera (a 64th of the frame is enough)
    torch.cuda.synchronize()
    strict = scene.trace(rays, traversal=render.TRAVERSAL_STRICT)
    check_render_walk(scene.trace(rays, traversal=render.TRAVERSAL_RENDER), strict, rays, info, "C5 primary vs reference-order walk")
    assert np.count_nonzero(strict["prim_id"] >= 5) > 2000
    sec = _surface_rays(capi, strict, 100_000, 5)
    check_render_walk(scene.trace(sec, traversal=render.TRAVERSAL_RENDER), scene.trace(sec, traversal=render.TRAVERSAL_STRICT), sec, info, "C5 secondary vs reference-order walk")


# ------------------------------------------------------------------------------------------------ scenes with transforms / media
@pytest.mark.parametrize("name,seed", [("cornell-box", 0), ("final-scene", 1), ("cornell-smoke", 0)])
def test_render_walk_instanced_scenes(render, oracle, scenes, capi, name, seed):
    """Cube / Translate / RotateY / ConstantMedium: the bracket variants' walk (32-byte entries, fused box test, hardware
    roots, reciprocal-radius normals).  Without media: against the oracle.  With media the free-flight draws depend on
    the visiting order, so the comparison is with the library's exact fast walk, which visits in the same order."""
    import torch

    cfg, bvh = scenes.make(name, seed=seed, width=200, height=200, samples_per_pixel=1)
    om, scene = oracle.OracleMesh(bvh), render.Scene(bvh)
    info = PrimInfo(scene)
    buf = torch.zeros(cfg.width * cfg.height * 3, device="cuda")
    render.Camera(cfg).render_accumulate(scene, buf.data_ptr(), 0, 1, seed=1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    prim = oracle.OracleCamera(cfg).center_rays(0.5)
    want = om.trace(prim, seed=77, mirror=True)
    sec = _surface_rays(capi, want, 30_000, 3)
    for label, rays in (("primary", prim), ("secondary", sec)):
        got = scene.trace(rays, traversal=render.TRAVERSAL_RENDER, seed=77)
        fast = scene.trace(rays, traversal=render.TRAVERSAL_FAST, seed=77)
        if name == "cornell-box":
            check_render_walk(got, om.trace(rays, seed=77, mirror=True), rays, info, f"{name} {label} vs oracle")
        check_render_walk(got, fast, rays, info, f"{name} {label} vs exact fast walk", max_id_frac=2e-4 if name == "cornell-box" else 3e-3)
