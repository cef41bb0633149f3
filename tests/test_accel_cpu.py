"""The traversal data the library builds on the host (threaded arrays: strict, fast, fast ordered for a viewpoint),
walked on the CPU by a small float32 interpreter of the kernels' walk and compared with the oracle's closest hits.
No GPU needed: rtx_traversal_dump touches no device."""
import importlib
import os

import numpy as np
import pytest

F = np.float32
ENT_END = 0xFFFFFFFE
MISS = 0xFFFFFFFF
INF = F(np.inf)


@pytest.fixture(scope="module")
def render():
    return importlib.import_module("ray-tracer_amd.render")


def _arr(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype) if n else np.zeros(0, dtype=dtype)


class SceneArrays:
    def __init__(self, desc):
        n = desc.n_spheres
        self.sph_id = _arr(desc.sph_prim_id, n, np.uint32)
        nq = desc.n_quads
        self.quad = np.stack([_arr(desc.quad_f[k], nq, F) for k in range(16)], axis=1) if nq else np.zeros((0, 16), dtype=F)
        self.quad_id = _arr(desc.quad_prim_id, nq, np.uint32)


def dot3(ax, ay, az, bx, by, bz):
    return ax * bx + ay * by + az * bz  # glam's left-to-right order, every operation rounded to f32


def slab(bounds, o, d, tmin, tmax, strict):
    for a in range(3):
        inv = F(1.0) / d[a]
        t0 = (bounds[2 * a] - o[a]) * inv
        t1 = (bounds[2 * a + 1] - o[a]) * inv
        if strict:
            lo, hi = (t0, t1) if t0 < t1 else (t1, t0)  # mesh.rs:522-526
        else:
            lo, hi = np.fmin(t0, t1), np.fmax(t0, t1)
        tmin = np.fmax(tmin, lo)  # f32::max / min ignore a NaN operand, as fmax / fmin do
        tmax = np.fmin(tmax, hi)
    return not (tmax <= tmin)


def walk(ents, rank, sa, ray, strict, tmin=F(0.001)):
    """Closest hit by walking the threaded array front to back; returns (prim_id, t, steps)."""
    o = [F(x) for x in ray["o"]]
    d = [F(x) for x in ray["d"]]
    time = F(ray["time"])
    a = dot3(d[0], d[1], d[2], d[0], d[1], d[2])
    fl = ents.view(F)
    off, best_t, best_ent, steps = 0, INF, None, 0

    def accept(t, idx):
        if strict:
            return tmin <= t and t < best_t
        if not (tmin <= t):
            return False
        if t < best_t:
            return True
        return t == best_t and best_ent is not None and rank[idx] < rank[best_ent]

    while True:
        steps += 1
        assert steps < 100000, "walk does not terminate"
        i = off // 32
        word = int(ents[i, 7])
        if word == ENT_END:
            break
        if word == 0:
            off = off + 32 if slab(fl[i, :6], o, d, tmin, best_t, strict) else int(ents[i, 6])
            continue
        kind, index = word & 15, word >> 4
        if kind & 8:
            # primitive of a Translate / RotateY instance placed in this frame's tree: the ray goes into the primitive's own
            # frame through the entry's chain record (mesh.rs:332, :365-372), outermost level first
            saved = (o, d, a)
            rec = int(ents[i, 3 if (kind & 3) == 1 else 1]) // 32
            for level in range(2):
                what, p = int(ents[rec, 4 * level]), fl[rec, 4 * level + 1:4 * level + 4]
                if what == 1:
                    o = [o[k] - p[k] for k in range(3)]
                elif what == 2:
                    sn, cs = p[0], p[1]
                    o = [cs * o[0] - sn * o[2], o[1], sn * o[0] + cs * o[2]]
                    d = [cs * d[0] - sn * d[2], d[1], sn * d[0] + cs * d[2]]
            a = dot3(d[0], d[1], d[2], d[0], d[1], d[2])
            kind &= 3
        else:
            saved = None
        if kind == 1:
            c = [fl[i, k] + (fl[i, 3 + k] * time if saved is None else F(0)) for k in range(3)]  # (a placed sphere is at rest)
            oc = [c[k] - o[k] for k in range(3)]
            h = dot3(d[0], d[1], d[2], oc[0], oc[1], oc[2])
            cc = dot3(oc[0], oc[1], oc[2], oc[0], oc[1], oc[2]) - fl[i, 6] * fl[i, 6]
            disc = h * h - a * cc
            if not (disc < 0):
                sq = np.sqrt(disc)
                root = (h - sq) / a
                ok = accept(root, i)
                if not ok and (strict or not (tmin <= root)):
                    root = (h + sq) / a
                    ok = accept(root, i)
                if ok:
                    best_t, best_ent = root, i
        elif kind == 2:
            q = sa.quad[index]
            denom = dot3(q[9], q[10], q[11], d[0], d[1], d[2])
            if not (abs(denom) < F(1e-8)):
                t = (q[15] - dot3(q[9], q[10], q[11], o[0], o[1], o[2])) / denom
                if accept(t, i):
                    p = [(o[k] + d[k] * t) - q[k] for k in range(3)]
                    u, v, w = q[3:6], q[6:9], q[12:15]
                    alpha = dot3(w[0], w[1], w[2], p[1] * v[2] - v[1] * p[2], p[2] * v[0] - v[2] * p[0], p[0] * v[1] - v[0] * p[1])
                    beta = dot3(w[0], w[1], w[2], u[1] * p[2] - p[1] * u[2], u[2] * p[0] - p[2] * u[0], u[0] * p[1] - p[0] * u[1])
                    if 0 <= alpha < 1 and 0 <= beta < 1:
                        best_t, best_ent = t, i
        else:
            raise AssertionError(f"entry kind {kind} is not interpreted here")
        if saved is not None:
            o, d, a = saved
        off += 32
    if best_ent is None:
        return MISS, INF, steps
    word = int(ents[best_ent, 7])
    pid = sa.sph_id[word >> 4] if (word & 3) == 1 else sa.quad_id[word >> 4]
    return int(pid), best_t, steps


PK_PRIM = 0x10000000


def packed_box_hit(c, e, o, d, tmin, tmax):
    """Box of the packed array, centre c and half extent e: near / far = centre distance -+ half extent * |1 / d| (real
    arithmetic is enough here: the point is that the rounded box still contains every primitive under it, which a wrong
    rounding direction would break)."""
    enter, leave = float(tmin), float(tmax)
    for a in range(3):
        da = float(d[a])
        if abs(da) < 1e-22:
            da = np.copysign(1e-22, da)
        tc = (c[a] - float(o[a])) / da
        half = e[a] / abs(da)
        enter, leave = max(enter, tc - half), min(leave, tc + half)
    return not (leave <= enter)


def walk_packed(slots, rank, ray, half_boxes, tmin=F(0.001)):
    """Closest hit by walking the packed fast array (16-byte slots); returns (prim id or sphere index, t, is_sphere, steps)."""
    o = [F(x) for x in ray["o"]]
    d = [F(x) for x in ray["d"]]
    time = F(ray["time"])
    a = dot3(d[0], d[1], d[2], d[0], d[1], d[2])
    fl = slots.view(F)
    i, best_t, best, steps = 0, INF, None, 0

    def accept(t, at):
        if not (tmin <= t):
            return False
        if t < best_t:
            return True
        return t == best_t and best is not None and rank[at] < rank[best]

    while True:
        steps += 1
        assert steps < 100000, "walk does not terminate"
        tag = int(slots[i, 3])
        if tag == ENT_END:
            break
        if tag < PK_PRIM:
            if not half_boxes:
                assert tag % 16 == 0
                tag //= 16  # float boxes carry their miss link as a byte offset
            assert tag > i, "miss links point forward"
            if half_boxes:  # one slot: six halfs
                h = np.array(slots[i, :3], dtype=np.uint32).view(np.float16).astype(np.float64)
                hit = packed_box_hit(h[:3], h[3:], o, d, tmin, best_t)
            else:  # two slots: (centre, link) (half extent, 0)
                hit = packed_box_hit(fl[i, :3].astype(np.float64), fl[i + 1, :3].astype(np.float64), o, d, tmin, best_t)
            i = i + (1 if half_boxes else 2) if hit else tag
            continue
        kind = tag >> 28
        if kind == 1:
            c = [fl[i, k] + fl[i + 1, k] * time for k in range(3)]
            oc = [c[k] - o[k] for k in range(3)]
            h = dot3(d[0], d[1], d[2], oc[0], oc[1], oc[2])
            cc = dot3(oc[0], oc[1], oc[2], oc[0], oc[1], oc[2]) - fl[i + 1, 3] * fl[i + 1, 3]
            disc = h * h - a * cc
            if not (disc < 0):
                sq = np.sqrt(disc)
                root = (h - sq) / a
                if not (tmin <= root):
                    root = (h + sq) / a
                if accept(root, i):
                    best_t, best = root, i
            i += 2 + ((tag >> 27) & 1)  # (a pad slot may follow a sphere: its tag says so)
        else:
            n, q, dd, u, v, w = fl[i, :3], fl[i + 1, :3], fl[i + 1, 3], fl[i + 2, :3], fl[i + 3, :3], fl[i + 4, :3]
            denom = dot3(n[0], n[1], n[2], d[0], d[1], d[2])
            if not (abs(denom) < F(1e-8)):
                t = (dd - dot3(n[0], n[1], n[2], o[0], o[1], o[2])) / denom
                if accept(t, i):
                    p = [(o[k] + d[k] * t) - q[k] for k in range(3)]
                    alpha = dot3(w[0], w[1], w[2], p[1] * v[2] - v[1] * p[2], p[2] * v[0] - v[2] * p[0], p[0] * v[1] - v[0] * p[1])
                    beta = dot3(w[0], w[1], w[2], u[1] * p[2] - p[1] * u[2], u[2] * p[0] - p[2] * u[0], u[0] * p[1] - p[0] * u[1])
                    inside = (alpha >= 0 and beta >= 0 and alpha + beta <= 1) if kind == 3 else (0 <= alpha < 1 and 0 <= beta < 1)
                    if inside:
                        best_t, best = t, i
            i += 5
    if best is None:
        return MISS, INF, False, steps
    tag = int(slots[best, 3])
    if (tag >> 28) == 1:
        return tag & 0x07FFFFFF, best_t, True, steps
    return int(slots[best + 3, 3]), best_t, False, steps


def prim_test_packed(fl, slots, i, o, d, a, time, tmin, accept):
    """Primitive record starting at slot i of a packed / wide primitive array: (t, True) if it is accepted."""
    kind = int(slots[i, 3]) >> 28
    if kind == 1:
        c = [fl[i, k] + fl[i + 1, k] * time for k in range(3)]
        oc = [c[k] - o[k] for k in range(3)]
        h = dot3(d[0], d[1], d[2], oc[0], oc[1], oc[2])
        cc = dot3(oc[0], oc[1], oc[2], oc[0], oc[1], oc[2]) - fl[i + 1, 3] * fl[i + 1, 3]
        disc = h * h - a * cc
        if disc < 0:
            return None
        sq = np.sqrt(disc)
        root = (h - sq) / a
        if not (tmin <= root):
            root = (h + sq) / a
        return root if accept(root, i) else None
    n, q, dd, u, v, w = fl[i, :3], fl[i + 1, :3], fl[i + 1, 3], fl[i + 2, :3], fl[i + 3, :3], fl[i + 4, :3]
    denom = dot3(n[0], n[1], n[2], d[0], d[1], d[2])
    if abs(denom) < F(1e-8):
        return None
    t = (dd - dot3(n[0], n[1], n[2], o[0], o[1], o[2])) / denom
    if not accept(t, i):
        return None
    p = [(o[k] + d[k] * t) - q[k] for k in range(3)]
    alpha = dot3(w[0], w[1], w[2], p[1] * v[2] - v[1] * p[2], p[2] * v[0] - v[2] * p[0], p[0] * v[1] - v[0] * p[1])
    beta = dot3(w[0], w[1], w[2], u[1] * p[2] - p[1] * u[2], u[2] * p[0] - p[2] * u[0], u[0] * p[1] - p[0] * u[1])
    inside = (alpha >= 0 and beta >= 0 and alpha + beta <= 1) if kind == 3 else (0 <= alpha < 1 and 0 <= beta < 1)
    return t if inside else None


def walk_wide(nodes, prims, rank, levels, ray, tmin=F(0.001)):
    """Closest hit by walking the wide form: four-child nodes, nearest child node first, pending groups on a stack that
    never holds more than `levels` groups; returns (prim id or sphere index, t, is_sphere, node steps, primitive tests)."""
    o = [F(x) for x in ray["o"]]
    d = [F(x) for x in ray["d"]]
    time = F(ray["time"])
    a = dot3(d[0], d[1], d[2], d[0], d[1], d[2])
    nf, pf = nodes.view(F), prims.view(F)
    best_t, best = INF, None

    def accept(t, at):
        if not (tmin <= t):
            return False
        if t < best_t:
            return True
        return t == best_t and best is not None and rank[at] < rank[best]

    def enter(lo, hi):  # distance at which the ray enters the box [lo, hi], None on a miss (real arithmetic, as packed_box_hit)
        t0, t1 = float(tmin), float(best_t)
        for ax in range(3):
            da = float(d[ax])
            if abs(da) < 1e-22:
                da = np.copysign(1e-22, da)
            ta, tb = (lo[ax] - float(o[ax])) / da, (hi[ax] - float(o[ax])) / da
            if da < 0:  # the near plane is chosen by the direction's sign (an unused child's inverted box is then never hit)
                ta, tb = tb, ta
            t0, t1 = max(t0, ta), min(t1, tb)
        return None if t1 <= t0 else t0

    stack, node, n_nodes, n_tests = [], 0, 0, 0
    while node is not None:
        n_nodes += 1
        assert n_nodes < 100000, "walk does not terminate"
        w = nodes[node]
        origin, scale = nf[node, 0:3].astype(np.float64), nf[node, 4:7].astype(np.float64)
        child_base, prim_base, leaves, n_int = int(w[3]), int(w[7]), int(w[11]), int(w[15]) & 255
        leaf = [(leaves >> (8 * j)) & 255 for j in range(4)]
        assert all(leaf[j] == 0 for j in range(n_int)), "a child node has no primitives of its own"
        hits = []
        for j in range(4):
            qlo = np.array([(int(w[8 + a]) >> (8 * j)) & 255 for a in range(3)], dtype=np.float64)
            qhi = np.array([(int(w[12 + a]) >> (8 * j)) & 255 for a in range(3)], dtype=np.float64)
            e = enter(origin + scale * qlo, origin + scale * qhi)
            if j < n_int:
                if e is not None:
                    hits.append((e, child_base + j))
            elif leaf[j]:
                if e is not None:
                    for k in range(leaf[j] & 31, (leaf[j] & 31) + (leaf[j] >> 5)):
                        n_tests += 1
                        at = prim_base + 5 * k
                        t = prim_test_packed(pf, prims, at, o, d, a, time, tmin, accept)
                        if t is not None:
                            best_t, best = t, at
            else:
                assert e is None and np.all(qlo == 255) and np.all(qhi == 0), "an unused child is never hit"
        hits.sort()
        if len(hits) > 1:
            stack.append([n for _, n in hits[1:]])
            assert len(stack) <= levels, "more pending groups than the tree has levels"
        if hits:
            node = hits[0][1]
        else:
            node = None
            while stack and node is None:
                node = stack[-1].pop(0)
                if not stack[-1]:
                    stack.pop()
    if best is None:
        return MISS, INF, False, n_nodes, n_tests
    tag = int(prims[best, 3])
    if (tag >> 28) == 1:
        return tag & (PK_PRIM - 1), best_t, True, n_nodes, n_tests
    return int(prims[best + 3, 3]), best_t, False, n_nodes, n_tests


def sample_rays(oracle, cfg, bvh, n_primary, n_secondary, seed):
    capi = importlib.import_module("ray-tracer_amd.capi")
    rng = np.random.default_rng(seed)
    prim_rays = oracle.OracleCamera(cfg).center_rays(0.5)
    pick = prim_rays[rng.choice(len(prim_rays), size=min(n_primary, len(prim_rays)), replace=False)]
    hits = oracle.OracleMesh(bvh).trace(prim_rays)
    pts = hits["p"][hits["prim_id"] != MISS]
    sec = np.zeros(n_secondary, dtype=capi.RAY_DTYPE)
    sec["o"] = pts[rng.integers(0, len(pts), n_secondary)]
    sec["d"] = rng.normal(size=(n_secondary, 3)).astype(F) * rng.uniform(0.2, 8.0, size=(n_secondary, 1)).astype(F)
    sec["time"] = rng.uniform(0, 1, n_secondary).astype(F)
    return np.concatenate([pick, sec])


CASES = [("default", 0, 120, 80), ("quads", 0, 120, 60), ("bouncing-spheres", 1, 150, 100)]


@pytest.mark.parametrize("name,seed,n_primary,n_secondary", CASES)
def test_threaded_arrays_walk_to_the_oracle_hits(render, oracle, scenes, name, seed, n_primary, n_secondary):
    cfg, bvh = scenes.make(name, seed=seed, width=64, height=36)
    rays = sample_rays(oracle, cfg, bvh, n_primary, n_secondary, 7)
    want = oracle.OracleMesh(bvh).trace(rays)
    hs = render.Scene(bvh, upload=False)
    sa = SceneArrays(hs.desc)
    view = np.asarray(cfg.lookfrom, dtype=F)
    arrays = {
        "strict": (hs.traversal_dump(render.TRAVERSAL_STRICT), True),
        "fast": (hs.traversal_dump(render.TRAVERSAL_FAST), False),
        "fast ordered for the camera": (hs.traversal_dump(render.TRAVERSAL_FAST, view=view), False),
        "fast ordered from the far side, enlarged": (hs.traversal_dump(render.TRAVERSAL_FAST, view=-view, grow=1e-4), False),
    }
    steps = {}
    with np.errstate(all="ignore"):
        for label, ((ents, rank), strict) in arrays.items():
            total = 0
            for r, w in zip(rays, want):
                pid, t, n = walk(ents, rank, sa, r, strict)
                total += n
                assert pid == int(w["prim_id"]), f"{name}/{label}: prim {pid} != {int(w['prim_id'])}"
                if pid != MISS:
                    assert t == w["t"], f"{name}/{label}: t differs"  # same un-fused f32 arithmetic: bit-equal
            steps[label] = total / len(rays)
    # the packed form of the fast array (what the render kernels walk): same hits from half-precision boxes
    packed_steps = 0
    for view_arg, grow, half in ((None, 0.0, False), (view, 0.0, True), (-view, 1e-4, False)):
        os.environ["RTX_PACKED_HALF"] = "1" if half else "0"  # small arrays get float boxes unless asked otherwise
        slots, prank = hs.traversal_dump(render.TRAVERSAL_FAST_FUSED, view=view_arg, grow=grow)
        del os.environ["RTX_PACKED_HALF"]
        assert slots.shape[1] == 4 and slots[-1, 3] == ENT_END
        with np.errstate(all="ignore"):
            for r, w in zip(rays, want):
                pid, t, is_sphere, n = walk_packed(slots, prank, r, half)
                packed_steps += n
                if is_sphere:
                    pid = int(sa.sph_id[pid])
                assert pid == int(w["prim_id"]), f"{name}/packed: prim {pid} != {int(w['prim_id'])}"
                if pid != MISS:
                    assert t == w["t"], f"{name}/packed: t differs"
    if name == "bouncing-spheres":
        assert packed_steps / (3 * len(rays)) < 1.15 * steps["fast"]  # half-precision boxes are only slightly looser
    # all four arrays hold the same primitives; ordering changes neither size nor content
    sizes = {k: len(v[0][0]) for k, v in arrays.items()}
    assert sizes["fast"] == sizes["fast ordered for the camera"] == sizes["fast ordered from the far side, enlarged"]
    if name == "bouncing-spheres":
        # the point of the fast tree and of the camera ordering: far fewer entries visited per ray
        assert steps["fast"] < 0.5 * steps["strict"]
        assert steps["fast ordered for the camera"] < steps["fast"]


def test_instances_are_placed_in_the_enclosing_tree(render, oracle, scenes):
    """cornell-box: the two Translate(RotateY(Cube)) instances leave no transform brackets in the fast array -- their
    twelve quads stand in the scene's own tree as instance primitives with a chain record behind the terminator -- and
    the walk through them reaches the oracle's hits with bit-identical distances.  A scene with an instance that cannot
    be placed (a chain deeper than a record holds) keeps brackets for all of them."""
    cfg, bvh = scenes.make("cornell-box", seed=0, width=48, height=48)
    hs = render.Scene(bvh, upload=False)
    sa = SceneArrays(hs.desc)
    rays = sample_rays(oracle, cfg, bvh, 150, 120, 5)
    want = oracle.OracleMesh(bvh).trace(rays)
    for view in (None, np.asarray(cfg.lookfrom, dtype=F)):
        ents, rank = hs.traversal_dump(render.TRAVERSAL_FAST, view=view)
        end = int(np.nonzero(ents[:, 7] == ENT_END)[0][0])
        kinds = ents[:end, 7] & 15
        boxes = ents[:end, 7] == 0
        assert not np.any((kinds[~boxes] >= 4) & (kinds[~boxes] < 8)), "transform brackets left in the array"
        placed = ~boxes & ((kinds & 8) != 0)
        assert np.count_nonzero(placed) == 12 and np.count_nonzero(~boxes) == 18
        chains = np.unique(ents[:end][placed][:, 1])
        assert len(ents) - end - 1 == len(chains) == 2 and np.all(chains > end * 32) and np.all(chains % 32 == 0)
        rec = ents[chains[0] // 32]
        assert rec[0] == 1 and rec[4] == 2  # Translate outside, RotateY inside
        with np.errstate(all="ignore"):
            for r, w in zip(rays, want):
                pid, t, _ = walk(ents, rank, sa, r, False)
                assert pid == int(w["prim_id"])
                if pid != MISS:
                    assert t == w["t"]
    # the strict array keeps the reference's structure
    strict, _ = hs.traversal_dump(render.TRAVERSAL_STRICT)
    assert np.count_nonzero((strict[:, 7] != 0) & ((strict[:, 7] & 15) == 4)) == 4
    os.environ["RTX_PLACE_INSTANCES"] = "0"
    try:
        ents, _ = render.Scene(bvh, upload=False).traversal_dump(render.TRAVERSAL_FAST)
    finally:
        del os.environ["RTX_PLACE_INSTANCES"]
    assert np.count_nonzero((ents[:, 7] != 0) & ((ents[:, 7] & 15) == 4)) == 4 and not np.any((ents[:, 7] != 0) & (ents[:, 7] != ENT_END) & ((ents[:, 7] & 8) != 0))


def _instance_world(scene, deep):
    """A ground sphere, a cube one level deep, a sphere two levels deep and, if `deep`, a cube three levels deep."""
    white = scene.Lambertian.rgb(0.73, 0.73, 0.73)
    w = scene.World()
    w.push(scene.Sphere.stationary((0.0, -1000.0, 0.0), 1000.0, scene.Lambertian.rgb(0.5, 0.6, 0.5)))
    w.push(scene.RotateY(scene.Cube((0, 0, 0), (1.0, 1.4, 1.0), white), 25.0))
    w.push(scene.Translate(scene.RotateY(scene.Sphere.stationary((0.0, 0.5, 0.0), 0.5, white), 33.0), (-2.0, 0.0, 1.0)))
    if deep:
        w.push(scene.Translate(scene.RotateY(scene.Translate(scene.Cube((0, 0, 0), (0.8, 0.8, 0.8), white), (0.2, 0.0, 0.1)), -40.0), (2.0, 0.0, -1.0)))
    cfg = scene.CameraConfig(width=40, height=30, samples_per_pixel=1, max_depth=5, vfov=35.0, lookfrom=(6.0, 3.0, 7.0), lookat=(0.0, 0.7, 0.0),
                             vup=(0, 1, 0))
    return cfg, scene.Bvh(w)


@pytest.mark.parametrize("deep", [False, True])
def test_instances_are_placed_all_or_none(render, oracle, scene, deep):
    """One- and two-level instances over primitives at rest are placed (chain records of one and two levels); an instance
    nested deeper than a record holds keeps its brackets, and then every instance of the scene does."""
    cfg, bvh = _instance_world(scene, deep)
    hs = render.Scene(bvh, upload=False)
    ents, rank = hs.traversal_dump(render.TRAVERSAL_FAST, view=np.asarray(cfg.lookfrom, dtype=F))
    end = int(np.nonzero(ents[:, 7] == ENT_END)[0][0])
    prim = ents[:end][ents[:end, 7] != 0]
    pushes = np.count_nonzero((prim[:, 7] & 15) == 4)
    placed = np.count_nonzero((prim[:, 7] & 8) != 0)
    if deep:
        assert pushes == 6 and placed == 0 and len(ents) == end + 1
        return
    assert pushes == 0 and placed == 7 and len(ents) == end + 3
    sa = SceneArrays(hs.desc)
    rays = sample_rays(oracle, cfg, bvh, 200, 150, 11)
    want = oracle.OracleMesh(bvh).trace(rays)
    assert len(set(want["prim_id"].tolist())) >= 5
    with np.errstate(all="ignore"):
        for r, w in zip(rays, want):
            pid, t, _ = walk(ents, rank, sa, r, False)
            assert pid == int(w["prim_id"])
            if pid != MISS:
                assert t == w["t"]


def test_traversal_dump_structure(render, scenes):
    """Links stay inside the array and point forward, the terminator is last, ranks are those of the strict order."""
    cfg, bvh = scenes.make("bouncing-spheres", seed=1)
    hs = render.Scene(bvh, upload=False)
    for mode in (render.TRAVERSAL_STRICT, render.TRAVERSAL_FAST):
        ents, rank = hs.traversal_dump(mode, view=(13, 2, 3) if mode == render.TRAVERSAL_FAST else None)
        n = len(ents)
        assert ents[-1, 7] == ENT_END and np.count_nonzero(ents[:, 7] == ENT_END) == 1
        box = ents[:, 7] == 0
        links = ents[box, 6]
        at = np.nonzero(box)[0] * 32
        assert np.all(links % 32 == 0) and np.all(links > at) and np.all(links <= (n - 1) * 32)
        prims = ~box
        prims[-1] = False
        assert np.count_nonzero(prims) >= hs.n_prims
        assert len(np.unique(rank[prims])) == hs.n_prims  # one rank per primitive, shared by repeated entries
    with pytest.raises(render.RtxError):
        hs.traversal_dump(7)


@pytest.mark.parametrize("name,seed,n_primary,n_secondary", CASES + [("mixed", 2, 100, 80)])
def test_wide_array_walks_to_the_oracle_hits(render, oracle, scenes, name, seed, n_primary, n_secondary):
    """The wide form (what the render kernels walk for scenes too large for LDS), built here for small scenes: every
    primitive is in it exactly once, links stay inside the arrays, and walking it reaches the oracle's hit with a
    bit-equal t -- also with boxes enlarged for a distant camera -- in fewer steps than the packed array takes."""
    cfg, bvh = scenes.make(name, seed=seed, width=64, height=36)
    rays = sample_rays(oracle, cfg, bvh, n_primary, n_secondary, 7)
    want = oracle.OracleMesh(bvh).trace(rays)
    hs = render.Scene(bvh, upload=False)
    sa = SceneArrays(hs.desc)
    tri_id = _arr(hs.desc.tri_prim_id, hs.desc.n_tris, np.uint32)
    nodes, prims, rank, levels = hs.wide_dump()
    n = len(nodes)
    assert nodes.shape[1] == 16 and prims.shape[0] == 5 * hs.n_prims and 1 <= levels <= 32
    n_int = nodes[:, 15] & 255
    assert np.all((nodes[:, 15] >> 8) == n_int + np.count_nonzero(np.stack([(nodes[:, 11] >> (8 * j)) & 255 for j in range(4)], axis=1), axis=1))
    leaf = np.stack([(nodes[:, 11] >> (8 * j)) & 255 for j in range(4)], axis=1)
    counts, starts = leaf >> 5, leaf & 31
    assert np.all(n_int <= 4) and np.all((nodes[:, 3] + n_int <= n) | (n_int == 0))
    assert counts.sum() == hs.n_prims and np.all(nodes[:, 7] + 5 * counts.sum(axis=1) <= len(prims))
    # the leaves of a node share the node's primitives without gaps or overlap, in child order
    for c, st in zip(counts, starts):
        run = 0
        for cj, sj in zip(c.tolist(), st.tolist()):
            if cj:
                assert sj == run
                run += cj
    children = np.concatenate([nodes[i, 3] + np.arange(n_int[i]) for i in range(n)]) if n else np.zeros(0)
    assert len(children) == n - 1 and len(np.unique(children)) == n - 1 and (n == 1 or children.min() == 1)  # every node but the root has one parent
    tags = prims[::5, 3]
    assert np.all((tags >> 28) >= 1) and np.all((tags >> 28) <= 3)
    ids = [int(sa.sph_id[t & (PK_PRIM - 1)]) if (t >> 28) == 1 else int(prims[5 * k + 3, 3]) for k, t in enumerate(tags)]
    assert sorted(ids) == sorted(set(ids)) and len(ids) == hs.n_prims
    if len(tri_id):
        assert set(tri_id.tolist()) <= set(ids)
    os.environ["RTX_PACKED_HALF"] = "1"
    slots, prank = hs.traversal_dump(render.TRAVERSAL_FAST_FUSED, view=np.asarray(cfg.lookfrom, dtype=F))
    del os.environ["RTX_PACKED_HALF"]
    wide_steps = packed_steps = 0
    for grow in (0.0, 1e-4):
        if grow:
            nodes, prims, rank, levels = hs.wide_dump(grow=grow)
        with np.errstate(all="ignore"):
            for r, w in zip(rays, want):
                pid, t, is_sphere, n_nodes, n_tests = walk_wide(nodes, prims, rank, levels, r)
                if is_sphere:
                    pid = int(sa.sph_id[pid])
                assert pid == int(w["prim_id"]), f"{name}/wide: prim {pid} != {int(w['prim_id'])}"
                if pid != MISS:
                    assert t == w["t"], f"{name}/wide: t differs"
                if not grow:
                    wide_steps += n_nodes + n_tests
                    packed_steps += walk_packed(slots, prank, r, True)[3]
    # a step of either walk is one dependent fetch: the wide form needs well under half as many
    assert wide_steps < 0.6 * packed_steps, (wide_steps / len(rays), packed_steps / len(rays))
    print(f"[wide] {name}: {n} nodes, {levels} levels, {wide_steps / len(rays):.1f} steps per ray against {packed_steps / len(rays):.1f} over the packed array")
    # a scene with transforms has no wide form
    cfg2, bvh2 = scenes.make("cornell-box", seed=0, width=16, height=16)
    with pytest.raises(render.RtxError) as e:
        render.Scene(bvh2, upload=False).wide_dump()
    assert e.value.code == render.RTX_ERR_UNSUPPORTED


def test_fast_builder_structure_choices(render, scenes):
    """Three choices of the fast builder that only change how many tests a walk makes (the walks above reach the oracle's
    hits on the arrays as built): a unit no ray of the scene can miss -- the ground sphere whose padded box reaches over
    everything, a medium that fills the scene -- stands right behind the root without a box of its own; a quad or
    triangle test is priced at several box steps, so the scenes of quads get one primitive per leaf; and the packed
    array of an LDS-sized scene holds entries of both 16-byte parities (a pad slot behind every eighth sphere)."""
    def kinds(name, **kw):
        cfg, bvh = scenes.make(name, seed=1, **kw)
        ents, _ = render.Scene(bvh, upload=False).traversal_dump(render.TRAVERSAL_FAST, view=np.asarray(cfg.lookfrom, dtype=F))
        return ents, [("box" if w == 0 else int(w) & 15) for w in ents[:, 7]]

    ents, k = kinds("bouncing-spheres")
    assert k[0] == "box" and k[1] == 1 and k[2] == "box"  # root, the ground sphere bare, the tree of the others
    assert ents[2, 6] == ents[0, 6]  # ... which ends where the root ends
    assert ents.view(F)[1, 6] == 1000.0
    ents, k = kinds("final-scene")
    assert k[:5] == ["box", 6, 1, 7, "box"] and ents[1, 4] == 1  # root, the fog: begin (boundary = one sphere), sphere, end
    for name in ("cornell-box", "cornell-smoke", "quads"):
        ents, k = kinds(name)
        prim = np.array([x != "box" and (x & 7) in (2, 3) for x in k])
        runs = np.diff(np.flatnonzero(np.diff(np.concatenate([[0], prim.astype(int), [0]])))).tolist()[::2]  # lengths of runs of planar entries
        assert max(runs) == 1, (name, runs)
    cfg, bvh = scenes.make("bouncing-spheres", seed=1)
    slots, _ = render.Scene(bvh, upload=False).traversal_dump(render.TRAVERSAL_FAST_FUSED, view=np.asarray(cfg.lookfrom, dtype=F))
    starts, i = [], 0
    while int(slots[i, 3]) != ENT_END:
        tag = int(slots[i, 3])
        starts.append(i)
        i += 2 if tag < PK_PRIM else (2 + ((tag >> 27) & 1) if (tag >> 28) == 1 else 5)
    odd = np.mean(np.array(starts) % 2 == 1)
    assert 0.3 < odd < 0.7, odd  # entries start at even and at odd slots alike
