"""ctypes plumbing for a native library that exposes the constructor-level C API

    <p>_tex_solid / _tex_checker / _tex_image / _tex_noise
    <p>_mat_lambertian / _mat_metal / _mat_dielectric / _mat_diffuse_light / _mat_isotropic
    <p>_sphere_stationary / _sphere_moving / _quad / _triangle / _cube / _translate / _rotate_y /
    <p>_constant_medium / _world_new / _world_push / _world_push_trimesh / _bvh_from
    <p>_camera_new

(`<p>` is the library's symbol prefix).  `Backend` adapts such a library to the calls `scene.lower()` makes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

F = np.float32
c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)


class CameraCfg(C.Structure):
    """CameraConfig (camera.rs:13-25) in declaration order, colours as linear RGB triples."""

    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32),
        ("vfov", C.c_float), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
        ("defocus_angle", C.c_float), ("focus_dist", C.c_float), ("background", C.c_float * 3),
    ]

    @classmethod
    def from_config(cls, cfg):
        return cls(cfg.width, cfg.height, cfg.samples_per_pixel, cfg.max_depth, float(cfg.vfov),
                   (C.c_float * 3)(*map(float, cfg.lookfrom)), (C.c_float * 3)(*map(float, cfg.lookat)),
                   (C.c_float * 3)(*map(float, cfg.vup)), float(cfg.defocus_angle), float(cfg.focus_dist),
                   (C.c_float * 3)(*map(float, cfg.background)))


class Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("time", C.c_float)]


class Hit(C.Structure):
    _fields_ = [("prim_id", C.c_uint32), ("t", C.c_float), ("n", C.c_float * 3), ("uv", C.c_float * 2),
                ("front_face", C.c_uint32), ("p", C.c_float * 3)]


RAY_DTYPE = np.dtype([("o", F, 3), ("d", F, 3), ("time", F)])
HIT_DTYPE = np.dtype([("prim_id", np.uint32), ("t", F), ("n", F, 3), ("uv", F, 2), ("front_face", np.uint32), ("p", F, 3)])
assert RAY_DTYPE.itemsize == C.sizeof(Ray) == 28
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 44

MISS = 0xFFFFFFFF


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in np.asarray(v, dtype=F).reshape(3)])


def declare_constructors(lib, p):
    """Set argtypes/restype of the constructor-level API on `lib` (symbol prefix `p`)."""
    V = C.c_void_p
    f3p = C.c_float * 3

    def d(name, res, *args):
        fn = getattr(lib, p + name)
        fn.restype = res
        fn.argtypes = list(args)
        return fn

    d("_tex_solid", V, C.c_float, C.c_float, C.c_float)
    d("_tex_checker", V, C.c_float, V, V)
    d("_tex_image", V, c_u8_p, C.c_uint32, C.c_uint32)
    d("_tex_noise", V, C.c_float, c_float_p, c_u32_p, c_u32_p, c_u32_p)
    d("_mat_lambertian", V, V)
    d("_mat_metal", V, V, C.c_float)
    d("_mat_dielectric", V, C.c_float)
    d("_mat_diffuse_light", V, V)
    d("_mat_isotropic", V, V)
    d("_sphere_stationary", V, f3p, C.c_float, V, C.c_uint32)
    d("_sphere_moving", V, f3p, f3p, C.c_float, V, C.c_uint32)
    d("_quad", V, f3p, f3p, f3p, V, C.c_uint32)
    d("_cube", V, f3p, f3p, V, C.c_uint32)
    d("_translate", V, V, f3p)
    d("_rotate_y", V, V, C.c_float)
    d("_constant_medium", V, V, C.c_float, V, C.c_uint32)
    d("_world_new", V)
    d("_world_push", C.c_int, V, V)
    d("_bvh_from", V, V)
    d("_camera_new", V, C.POINTER(CameraCfg))
    if hasattr(lib, p + "_triangle"):
        d("_triangle", V, f3p, f3p, f3p, V, C.c_uint32)
        d("_world_push_trimesh", C.c_int, V, c_float_p, C.c_uint32, c_u32_p, C.c_uint32, V, C.c_uint32)


class Backend:
    """Adapter between scene.lower() and a constructor-level native library."""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix
        self._keep = []  # numpy buffers the native side may still borrow during construction

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    def _chk(self, h, what):
        if not h:
            raise ValueError(f"{self.p}{what} rejected its arguments")
        return h

    def tex_solid(self, albedo):
        return self._chk(self._f("_tex_solid")(*map(float, albedo)), "_tex_solid")

    def tex_checker(self, scale, even, odd):
        return self._chk(self._f("_tex_checker")(float(scale), even, odd), "_tex_checker")

    def tex_image(self, rgb):
        h, w = rgb.shape[:2]
        return self._chk(self._f("_tex_image")(rgb.ctypes.data_as(c_u8_p), w, h), "_tex_image")

    def tex_noise(self, scale, values, px, py, pz):
        return self._chk(self._f("_tex_noise")(float(scale), fptr(values), px.ctypes.data_as(c_u32_p),
                                                py.ctypes.data_as(c_u32_p), pz.ctypes.data_as(c_u32_p)), "_tex_noise")

    def mat_lambertian(self, t):
        return self._chk(self._f("_mat_lambertian")(t), "_mat_lambertian")

    def mat_metal(self, t, roughness):
        return self._chk(self._f("_mat_metal")(t, float(roughness)), "_mat_metal")

    def mat_dielectric(self, ri):
        return self._chk(self._f("_mat_dielectric")(float(ri)), "_mat_dielectric")

    def mat_diffuse_light(self, t):
        return self._chk(self._f("_mat_diffuse_light")(t), "_mat_diffuse_light")

    def mat_isotropic(self, t):
        return self._chk(self._f("_mat_isotropic")(t), "_mat_isotropic")

    def sphere_stationary(self, c, r, m, pid):
        return self._chk(self._f("_sphere_stationary")(f3(c), float(r), m, pid), "_sphere_stationary")

    def sphere_moving(self, a, b, r, m, pid):
        return self._chk(self._f("_sphere_moving")(f3(a), f3(b), float(r), m, pid), "_sphere_moving")

    def quad(self, q, u, v, m, pid):
        return self._chk(self._f("_quad")(f3(q), f3(u), f3(v), m, pid), "_quad")

    def triangle(self, v0, v1, v2, m, pid):
        return self._chk(self._f("_triangle")(f3(v0), f3(v1), f3(v2), m, pid), "_triangle")

    def cube(self, a, b, m, pid):
        return self._chk(self._f("_cube")(f3(a), f3(b), m, pid), "_cube")

    def translate(self, o, off):
        return self._chk(self._f("_translate")(o, f3(off)), "_translate")

    def rotate_y(self, o, angle):
        return self._chk(self._f("_rotate_y")(o, float(angle)), "_rotate_y")

    def constant_medium(self, b, density, t, pid):
        return self._chk(self._f("_constant_medium")(b, float(density), t, pid), "_constant_medium")

    def world_new(self):
        return self._chk(self._f("_world_new")(), "_world_new")

    def world_push(self, w, m):
        if self._f("_world_push")(w, m) != 0:
            raise ValueError("world_push: not a world")

    def world_push_trimesh(self, w, vertices, indices, m, first_pid):
        self._keep += [vertices, indices]
        rc = self._f("_world_push_trimesh")(w, fptr(vertices), len(vertices), indices.ctypes.data_as(c_u32_p),
                                            len(indices), m, first_pid)
        if rc != 0:
            raise ValueError("world_push_trimesh rejected its arguments")

    def bvh_from(self, w):
        return self._chk(self._f("_bvh_from")(w), "_bvh_from")

    def camera_new(self, cfg):
        c = CameraCfg.from_config(cfg)
        return self._chk(self._f("_camera_new")(C.byref(c)), "_camera_new")
