"""Front end over the native libraries: librtx_host.so (C++ host layer: constructors + flattener) and librtx.so
(HIP kernels behind the C ABI of include/rtx.h).

    scene  = Scene(bvh)                      # lower the description, flatten it (host), upload it (GPU)
    camera = Camera(config)                  # Camera::new
    image  = camera.render(scene)            # Camera::render -> H x W x 3 uint8

There is no CPU fallback: loading fails loudly when the libraries are missing, and every compute call fails
with RTX_ERR_NO_DEVICE when no gfx950 GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi
from .scene import lower

F = np.float32
_DIR = os.path.dirname(os.path.abspath(__file__))

TRAVERSAL_FAST, TRAVERSAL_STRICT, TRAVERSAL_FAST_FUSED, TRAVERSAL_RENDER = 0, 1, 2, 3
SPLIT_SAMPLES, SPLIT_TILES = 0, 1
RENDER_REPRODUCIBLE = 1
RTX_OK, RTX_ERR_INVALID, RTX_ERR_UNSUPPORTED, RTX_ERR_DEVICE, RTX_ERR_NO_DEVICE = 0, -1, -2, -3, -4


class RtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtx error {code}: {msg}")
        self.code = code


class CameraFFI(C.Structure):
    """rtx_camera_t"""

    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
                ("defocus_angle", C.c_float), ("center", C.c_float * 3), ("pixel00_loc", C.c_float * 3),
                ("pixel_delta_u", C.c_float * 3), ("pixel_delta_v", C.c_float * 3), ("defocus_disk_u", C.c_float * 3),
                ("defocus_disk_v", C.c_float * 3), ("background", C.c_float * 3), ("pixel_samples_scale", C.c_float)]


class TraceOpts(C.Structure):
    """rtx_trace_opts_t"""

    _fields_ = [("tmin", C.c_float), ("tmax", C.c_float), ("traversal", C.c_uint32), ("seed", C.c_uint64)]


class RenderOpts(C.Structure):
    """rtx_render_opts_t"""

    _fields_ = [("seed", C.c_uint64), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32), ("traversal", C.c_uint32),
                ("rr_start_depth", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32), ("n_gpus", C.c_uint32), ("split", C.c_uint32),
                ("flags", C.c_uint32)]


class Stats(C.Structure):
    """rtx_stats_t"""

    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("n_launches", C.c_uint32), ("flags", C.c_uint32), ("box_tests", C.c_uint64), ("prim_tests", C.c_uint64)]


class BvhNode(C.Structure):
    _fields_ = [("bounds", C.c_float * 6), ("left", C.c_uint32), ("right", C.c_uint32)]


class ObjectRec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("child", C.c_uint32), ("first", C.c_uint32), ("count", C.c_uint32), ("f", C.c_float * 3),
                ("material", C.c_uint32), ("prim_id", C.c_uint32), ("pad", C.c_uint32 * 3)]


class MaterialRec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("param", C.c_float), ("pad", C.c_uint32)]


class TextureRec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("scale", C.c_float), ("a", C.c_uint32), ("b", C.c_uint32), ("rgb", C.c_float * 3), ("pad", C.c_uint32)]


class ImageRec(C.Structure):
    _fields_ = [("rgb", C.POINTER(C.c_uint8)), ("width", C.c_uint32), ("height", C.c_uint32)]


class PerlinRec(C.Structure):
    _fields_ = [("values", capi.c_float_p), ("perm_x", capi.c_u32_p), ("perm_y", capi.c_u32_p), ("perm_z", capi.c_u32_p)]


class SceneDesc(C.Structure):
    """rtx_scene_desc_t"""

    _fields_ = [
        ("n_spheres", C.c_uint32),
        ("sph_cx", capi.c_float_p), ("sph_cy", capi.c_float_p), ("sph_cz", capi.c_float_p),
        ("sph_dx", capi.c_float_p), ("sph_dy", capi.c_float_p), ("sph_dz", capi.c_float_p),
        ("sph_radius", capi.c_float_p), ("sph_material", capi.c_u32_p), ("sph_prim_id", capi.c_u32_p),
        ("n_quads", C.c_uint32), ("quad_f", capi.c_float_p * 16), ("quad_material", capi.c_u32_p), ("quad_prim_id", capi.c_u32_p),
        ("n_tris", C.c_uint32), ("tri_f", capi.c_float_p * 9), ("tri_material", capi.c_u32_p), ("tri_prim_id", capi.c_u32_p),
        ("n_objects", C.c_uint32), ("objects", C.POINTER(ObjectRec)),
        ("n_refs", C.c_uint32), ("refs", capi.c_u32_p),
        ("n_bvh_nodes", C.c_uint32), ("bvh_nodes", C.POINTER(BvhNode)),
        ("root", C.c_uint32),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(MaterialRec)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(TextureRec)),
        ("n_images", C.c_uint32), ("images", C.POINTER(ImageRec)),
        ("n_perlins", C.c_uint32), ("perlins", C.POINTER(PerlinRec)),
    ]


_host = None
_rtx = None

RTX_SYMBOLS = ["rtx_device_count", "rtx_last_error", "rtx_scene_create", "rtx_scene_destroy", "rtx_trace", "rtx_trace_device",
               "rtx_render", "rtx_render_multi", "rtx_render_accumulate", "rtx_render_accumulate_static", "rtx_resolve", "rtx_traversal_dump", "rtx_wide_dump"]


def rtx():
    """librtx.so (the C ABI)."""
    global _rtx
    if _rtx is not None:
        return _rtx
    path = os.environ.get("RTX_LIB") or os.path.join(_DIR, "librtx.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {_DIR}` (hipcc, gfx950); there is no fallback path")
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    V = C.c_void_p
    L.rtx_device_count.restype = C.c_int
    L.rtx_last_error.restype = C.c_char_p
    L.rtx_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(V)]
    L.rtx_scene_destroy.argtypes = [V]
    L.rtx_trace.argtypes = [V, V, C.c_size_t, C.POINTER(TraceOpts), V]
    L.rtx_trace_device.argtypes = [V, V, C.c_size_t, C.POINTER(TraceOpts), V, V]
    L.rtx_render.argtypes = [V, C.POINTER(CameraFFI), C.POINTER(RenderOpts), V, V, C.POINTER(Stats)]
    L.rtx_render_multi.argtypes = [V, C.POINTER(CameraFFI), C.POINTER(RenderOpts), C.POINTER(C.c_int), C.c_uint32, C.c_uint32, V, V, C.POINTER(Stats)]
    L.rtx_render_accumulate.argtypes = [V, C.POINTER(CameraFFI), C.POINTER(RenderOpts), V, V, C.POINTER(Stats)]
    L.rtx_render_accumulate_static.argtypes = [V, C.POINTER(CameraFFI), C.POINTER(RenderOpts), V, V, C.POINTER(Stats)]
    L.rtx_resolve.argtypes = [V, C.c_uint32, C.c_uint32, C.c_float, V, V]
    L.rtx_traversal_dump.argtypes = [C.POINTER(SceneDesc), C.c_uint32, capi.c_float_p, C.c_float, capi.c_u32_p, capi.c_u32_p, C.c_size_t,
                                     C.POINTER(C.c_size_t)]
    L.rtx_wide_dump.argtypes = [C.POINTER(SceneDesc), C.c_float, capi.c_u32_p, C.c_size_t, capi.c_u32_p, capi.c_u32_p, C.c_size_t,
                                C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    _rtx = L
    return L


def host():
    """librtx_host.so (C++ host layer)."""
    global _host
    if _host is not None:
        return _host
    rtx()
    path = os.path.join(_DIR, "librtx_host.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {_DIR}`")
    L = C.CDLL(path)
    capi.declare_constructors(L, "rth")
    V = C.c_void_p
    L.rth_last_error.restype = C.c_char_p
    L.rth_mesh_bbox.argtypes = [V, C.c_float * 6]
    L.rth_camera_ffi.argtypes = [V, C.POINTER(CameraFFI)]
    L.rth_scene_flatten.restype = V
    L.rth_scene_flatten.argtypes = [V]
    L.rth_scene_desc.restype = C.POINTER(SceneDesc)
    L.rth_scene_desc.argtypes = [V]
    L.rth_scene_upload.restype = C.c_int
    L.rth_scene_upload.argtypes = [V]
    L.rth_scene_device.restype = V
    L.rth_scene_device.argtypes = [V]
    L.rth_scene_free.argtypes = [V]
    L.rth_scene_leaf_order.restype = C.c_size_t
    L.rth_scene_leaf_order.argtypes = [V, capi.c_u32_p, C.c_size_t]
    _host = L
    return L


def _check(rc):
    if rc != RTX_OK:
        raise RtxError(rc, rtx().rtx_last_error().decode())


def device_count():
    return rtx().rtx_device_count()


class Scene:
    """A scene description lowered into the host layer, flattened, and (lazily) resident on the current GPU."""

    def __init__(self, root, upload=True):
        self.be = capi.Backend(host(), "rth")
        self.root_handle, self.n_prims = lower(root, self.be)
        self.handle = host().rth_scene_flatten(self.root_handle)
        if not self.handle:
            raise ValueError(host().rth_last_error().decode())
        self.desc = host().rth_scene_desc(self.handle).contents
        if upload:
            self.upload()

    def upload(self):
        rc = host().rth_scene_upload(self.handle)
        if rc != RTX_OK:
            raise RtxError(rc, host().rth_last_error().decode())
        return self

    @property
    def device(self):
        d = host().rth_scene_device(self.handle)
        if not d:
            raise RuntimeError("scene is not uploaded")
        return d

    def traversal_dump(self, traversal=TRAVERSAL_FAST, view=None, grow=0.0):
        """(entries [n, 8] uint32, rank [n] uint32): the threaded array the kernels would walk (host only, no GPU)."""
        n = C.c_size_t(0)
        vp = None
        if view is not None:
            varr = np.ascontiguousarray(view, dtype=F)
            vp = varr.ctypes.data_as(capi.c_float_p)
        _check(rtx().rtx_traversal_dump(C.byref(self.desc), traversal, vp, grow, None, None, 0, C.byref(n)))
        ents = np.zeros((n.value, 4 if traversal == TRAVERSAL_FAST_FUSED else 8), dtype=np.uint32)  # packed form: 4 words per slot
        rank = np.zeros(n.value, dtype=np.uint32)
        _check(rtx().rtx_traversal_dump(C.byref(self.desc), traversal, vp, grow, ents.ctypes.data_as(capi.c_u32_p),
                                        rank.ctypes.data_as(capi.c_u32_p), n.value, C.byref(n)))
        return ents, rank

    def wide_dump(self, grow=0.0):
        """(nodes [n, 16] uint32, prims [slots, 4] uint32, rank [slots] uint32, levels): the wide form of the fast data
        (host only, no GPU); RtxError(RTX_ERR_UNSUPPORTED) when the scene has none."""
        nn, ns, lv = C.c_size_t(0), C.c_size_t(0), C.c_uint32(0)
        _check(rtx().rtx_wide_dump(C.byref(self.desc), grow, None, 0, None, None, 0, C.byref(nn), C.byref(ns), C.byref(lv)))
        nodes = np.zeros((nn.value, 16), dtype=np.uint32)
        prims = np.zeros((ns.value, 4), dtype=np.uint32)
        rank = np.zeros(ns.value, dtype=np.uint32)
        _check(rtx().rtx_wide_dump(C.byref(self.desc), grow, nodes.ctypes.data_as(capi.c_u32_p), nn.value, prims.ctypes.data_as(capi.c_u32_p),
                                   rank.ctypes.data_as(capi.c_u32_p), ns.value, C.byref(nn), C.byref(ns), C.byref(lv)))
        return nodes, prims, rank, lv.value

    def bbox(self):
        out = (C.c_float * 6)()
        host().rth_mesh_bbox(self.root_handle, out)
        return np.array(out, dtype=F)

    def leaf_order(self):
        n = host().rth_scene_leaf_order(self.handle, None, 0)
        out = np.zeros(n, dtype=np.uint32)
        host().rth_scene_leaf_order(self.handle, out.ctypes.data_as(capi.c_u32_p), n)
        return out

    def node_boxes(self):
        """Boxes of the flattened reference BVH in node-index (pre-order) order."""
        d = self.desc
        return np.array([list(d.bvh_nodes[i].bounds) for i in range(d.n_bvh_nodes)], dtype=F).reshape(-1, 6)

    def trace(self, rays, tmin=0.001, tmax=float("inf"), traversal=TRAVERSAL_FAST, seed=0):
        """Closest hits (Mesh::hit of the top-level object) through rtx_trace."""
        rays = np.ascontiguousarray(rays, dtype=capi.RAY_DTYPE)
        hits = np.zeros(len(rays), dtype=capi.HIT_DTYPE)
        opts = TraceOpts(tmin, tmax, traversal, seed)
        _check(rtx().rtx_trace(self.device, rays.ctypes.data, len(rays), C.byref(opts), hits.ctypes.data))
        return hits

    def close(self):
        if self.handle:
            host().rth_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Camera:
    """Camera::new (camera.rs:57-96) + Camera::render (camera.rs:98)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.handle = capi.Backend(host(), "rth").camera_new(cfg)
        self.ffi = CameraFFI()
        host().rth_camera_ffi(self.handle, C.byref(self.ffi))

    def constants(self):
        f = self.ffi
        d = {n: np.array(getattr(f, n), dtype=F) for n in
             ["center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "defocus_disk_u", "defocus_disk_v"]}
        d["pixel_samples_scale"] = F(f.pixel_samples_scale)
        return d

    def render(self, scene, seed=0, traversal=TRAVERSAL_FAST, rr_start_depth=0, want_accum=False, want_stats=False, n_gpus=1,
               split=SPLIT_SAMPLES, devices=None, reproducible=False):
        """Whole-frame render through rtx_render (host buffers). Returns rgb8 [H,W,3] (+ accum [H,W,3] f32 sums, + Stats).
        n_gpus > 1: the one call drives GPUs 0 .. n_gpus-1; devices=[...] names them explicitly (rtx_render_multi; an
        ordinal may repeat to rehearse the multi-GPU path on one GPU)."""
        h, w = self.cfg.height, self.cfg.width
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        accum = np.zeros((h, w, 3), dtype=F) if want_accum else None
        st = Stats()
        if devices is not None:
            opts = RenderOpts(seed, 0, 0, traversal, rr_start_depth, 0, 0, 0, 0, RENDER_REPRODUCIBLE if reproducible else 0)
            ids = (C.c_int * len(devices))(*devices)
            _check(rtx().rtx_render_multi(scene.device, C.byref(self.ffi), C.byref(opts), ids, len(devices), split, rgb.ctypes.data,
                                          accum.ctypes.data if want_accum else None, C.byref(st)))
        else:
            opts = RenderOpts(seed, 0, 0, traversal, rr_start_depth, 0, 0, n_gpus, split, RENDER_REPRODUCIBLE if reproducible else 0)
            _check(rtx().rtx_render(scene.device, C.byref(self.ffi), C.byref(opts), rgb.ctypes.data,
                                    accum.ctypes.data if want_accum else None, C.byref(st)))
        out = [rgb]
        if want_accum:
            out.append(accum)
        if want_stats:
            out.append(st)
        return out[0] if len(out) == 1 else tuple(out)

    def render_accumulate(self, scene, accum_ptr, sample_begin, sample_end, seed=0, traversal=TRAVERSAL_FAST, rr_start_depth=0,
                          stream=None, stats=None, tile_rank=0, tile_world=1, static=False, reproducible=False):
        """rtx_render_accumulate on a device buffer (3*W*H f32) for the sample range [sample_begin, sample_end) of the
        tiles t with t % tile_world == tile_rank.  static=True: rtx_render_accumulate_static, the variant that keeps its
        scratch in a fixed device array and allocates nothing."""
        opts = RenderOpts(seed, sample_begin, sample_end, traversal, rr_start_depth, tile_rank, tile_world, 0, 0,
                          RENDER_REPRODUCIBLE if reproducible else 0)
        fn = rtx().rtx_render_accumulate_static if static else rtx().rtx_render_accumulate
        _check(fn(scene.device, C.byref(self.ffi), C.byref(opts), accum_ptr, stream,
                                           C.byref(stats) if stats is not None else None))

    def resolve(self, accum_ptr, rgb8_ptr, scale=None, stream=None):
        s = float(self.ffi.pixel_samples_scale) if scale is None else float(scale)
        _check(rtx().rtx_resolve(accum_ptr, self.cfg.width, self.cfg.height, s, rgb8_ptr, stream))
