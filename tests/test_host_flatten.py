"""Host layer (C++ constructors + flattener) against the oracle, and the C-ABI surface.  No GPU needed."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def render():
    return importlib.import_module("ray-tracer_amd.render")


def test_c_abi_exports_every_declared_symbol(render):
    """Every function include/rtx.h declares is exported by librtx.so (and nothing is called here)."""
    text = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtx_[a-z_]+)\s*\(", text))
    declared -= {"rtx_status"}
    lib = render.rtx()
    assert declared == set(render.RTX_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header(render):
    assert C.sizeof(render.BvhNode) == 32 and C.sizeof(render.ObjectRec) == 48
    assert C.sizeof(render.MaterialRec) == 16 and C.sizeof(render.TextureRec) == 32
    assert C.sizeof(render.CameraFFI) == 4 * 4 + 4 + 7 * 12 + 4


@pytest.mark.parametrize("name", ["bouncing-spheres", "quads", "default", "simple-light", "cornell-box", "final-scene"])
def test_bvh_topology_and_boxes_match_oracle(render, oracle, scenes, name):
    """Bvh::from restated twice (host layer, oracle) gives the same leaf order and the same node boxes."""
    cfg, bvh = scenes.make(name, seed=3)
    hs = render.Scene(bvh, upload=False)
    om = oracle.OracleMesh(bvh)
    assert hs.n_prims == om.n_prims
    assert np.array_equal(hs.leaf_order(), om.leaf_order())
    assert np.array_equal(hs.bbox(), om.bbox())
    if name != "final-scene":  # the oracle dump follows only top-level Bvh chains
        assert np.array_equal(hs.node_boxes(), om.node_boxes())


@pytest.mark.parametrize("name", ["bouncing-spheres", "earth", "quads", "cornell-box", "default"])
def test_camera_constants_match_oracle(render, oracle, scenes, name):
    cfg = scenes.CAMERAS[name]()
    a = render.Camera(cfg).constants()
    b = oracle.OracleCamera(cfg).constants()
    for k in b:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_flatten_shares_arc_clones(render, scenes):
    """checkered-spheres pushes one Arc<Lambertian<Checker>> twice (main.rs:233-240): one material, three textures."""
    cfg, bvh = scenes.make("checkered-spheres")
    s = render.Scene(bvh, upload=False)  # keep the scene alive: desc borrows from it
    d = s.desc
    assert d.n_spheres == 2 and d.n_materials == 1 and d.n_textures == 3
    assert d.sph_material[0] == d.sph_material[1] == 0
    assert d.n_bvh_nodes == 1 and d.root == 0


def test_flatten_quad_derived_fields(render, scene):
    """Quad::new derives normal, d, w (mesh.rs:233-244): Q=(-2,-2,0) u=(4,0,0) v=(0,4,0) -> n=(0,0,1) d=0 w=(0,0,1/16)."""
    w = scene.World().push(scene.Quad((-2, -2, 0), (4, 0, 0), (0, 4, 0), scene.Lambertian.rgb(.5, .5, .5)))
    s = render.Scene(scene.Bvh(w), upload=False)
    d = s.desc
    assert d.n_quads == 1
    f = [d.quad_f[k][0] for k in range(16)]
    assert f[9:12] == [0, 0, 1] and f[12:15] == [0, 0, F(1) / F(16)] and f[15] == 0


def test_cube_is_a_list_of_six_quads(render, scene):
    w = scene.World().push(scene.Cube((0, 0, 0), (1, 2, 3), scene.Lambertian.rgb(.5, .5, .5)))
    s = render.Scene(scene.Bvh(w), upload=False)
    d = s.desc
    assert d.n_quads == 6 and d.n_objects == 1  # the cube's inner World; Bvh::from consumes the outer one
    assert list(s.leaf_order()) == [0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5]  # one-element Bvh: left == right


def test_errors(render, scene):
    with pytest.raises(ValueError):
        scene.Sphere.stationary((0, 0, 0), -1.0, scene.Lambertian.rgb(1, 1, 1))
    with pytest.raises(ValueError):
        scene.Bvh(scene.World())
    # the native constructor enforces the same assert (mesh.rs:148)
    be = importlib.import_module("ray-tracer_amd.capi").Backend(render.host(), "rth")
    m = be.mat_dielectric(1.5)
    with pytest.raises(ValueError):
        be.sphere_stationary((0, 0, 0), -1.0, m, 0)


def test_no_gpu_means_loud_failure(render, scenes):
    """Without a gfx950 device the product path refuses to run (no CPU fallback)."""
    if render.device_count() > 0:
        pytest.skip("a GPU is present")
    cfg, bvh = scenes.make("quads")
    with pytest.raises(render.RtxError) as e:
        render.Scene(bvh, upload=True)
    assert e.value.code == render.RTX_ERR_NO_DEVICE


def test_triangle_mesh_flattens_like_the_oracle(render, oracle, scenes):
    cfg, bvh = scenes.make("triangle-mesh", seed=2, world_args=dict(n_lat=12))
    hs, om = render.Scene(bvh, upload=False), oracle.OracleMesh(bvh)
    assert hs.desc.n_tris == 4 * 12 * 11 and hs.n_prims == om.n_prims == 1 + 4 * 12 * 11
    assert np.array_equal(hs.leaf_order(), om.leaf_order())
    assert np.array_equal(hs.node_boxes(), om.node_boxes())


def test_unsupported_scene_graphs_are_refused_not_mis_rendered(render, scene):
    """What the traversal data cannot express is refused when it is built (RTX_ERR_UNSUPPORTED), host side, no GPU needed:
    ConstantMedium boundaries nested more than four deep (mesh.rs:627-631 takes any `B: Mesh`; a walk keeps one saved
    boundary-query state per enclosing medium), and transforms nested deeper than the kernels' limit of 64."""
    white = scene.Lambertian.rgb(0.7, 0.7, 0.7)
    ball = scene.Sphere.stationary((0, 0, 0), 1.0, white)

    def nested(levels):
        m = ball
        for k in range(levels):
            m = scene.ConstantMedium.from_color(m, 0.1 * (k + 1), (1, 1, 1))
        w = scene.World()
        w.push(m)
        w.push(scene.Sphere.stationary((3, 0, 0), 1.0, white))
        return scene.Bvh(w)

    for mode in (render.TRAVERSAL_STRICT, render.TRAVERSAL_FAST):
        with pytest.raises(render.RtxError) as e:
            render.Scene(nested(5), upload=False).traversal_dump(mode)
        assert e.value.code == render.RTX_ERR_UNSUPPORTED and "ConstantMedium" in str(e.value)
        # four levels are accepted; every begin / end entry carries its nesting level (word 3 / word 0), 0 = outermost
        ents, _ = render.Scene(nested(4), upload=False).traversal_dump(mode)
        kind = ents[:, 7] & 15
        assert sorted(ents[kind == 6, 3].tolist()) == [0, 1, 2, 3] and sorted(ents[kind == 7, 0].tolist()) == [0, 1, 2, 3]
        for b in np.flatnonzero(kind == 6):
            end = ents[b, 6] // 32  # (links are byte offsets of 32-byte entries)
            assert (ents[end, 7] & 15) == 7 and ents[end, 0] == ents[b, 3] and ents[end, 6] == 32 * b
    obj = ball
    for k in range(65):
        obj = scene.Translate(obj, (0.01, 0.0, 0.0)) if k % 2 else scene.RotateY(obj, 1.0)
    w = scene.World()
    w.push(obj)
    w.push(scene.Sphere.stationary((3, 0, 0), 1.0, white))
    with pytest.raises(render.RtxError) as e:
        render.Scene(scene.Bvh(w), upload=False).traversal_dump(render.TRAVERSAL_STRICT)
    assert e.value.code == render.RTX_ERR_UNSUPPORTED and "64" in str(e.value)
    # 64 levels are accepted
    obj = ball
    for k in range(64):
        obj = scene.Translate(obj, (0.01, 0.0, 0.0)) if k % 2 else scene.RotateY(obj, 1.0)
    w = scene.World()
    w.push(obj)
    w.push(scene.Sphere.stationary((3, 0, 0), 1.0, white))
    ents, _ = render.Scene(scene.Bvh(w), upload=False).traversal_dump(render.TRAVERSAL_STRICT)
    assert np.count_nonzero((ents[:, 7] & 15) == 4) == 64
