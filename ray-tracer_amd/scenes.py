"""Seeded restatements of the reference's hard-coded scenes (/root/reference/src/main.rs:31-540).

Scene construction belongs to the host program in the reference (main.rs) and draws from an OS-seeded
generator, so no two runs of the reference render the same scene.  These generators follow the same
construction order and the same arithmetic (in float32) from an explicit seed, and are what the tests and
the benchmark render.  `SCENES` maps the reference's CLI names (kebab-case, main.rs:17-28) plus `default`
(the five-sphere scene the 400x225 configuration is quoted on) to (camera_config, world) factories.
"""
from __future__ import annotations

import os

import numpy as np

from .scene import (Bvh, CameraConfig, CheckerTexture, ConstantMedium, Cube, Dielectric, DiffuseLight, ImageTexture,
                    Lambertian, Metal, NoiseTexture, Quad, RotateY, SolidTexture, Sphere, Translate, TriangleMesh, World, vec3)

F = np.float32
# decoded raster of the reference's images/earthmap.jpg (1024x512 RGB8), shipped with the package
_EARTH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "earthmap_rgb8.png")

X, Y, Z = vec3(1, 0, 0), vec3(0, 1, 0), vec3(0, 0, 1)


class SceneRng:
    """Deterministic stand-in for rand::thread_rng() in scene construction: splitmix64 words with
    rand 0.8.5's float conversions (gen::<f32>() = 24 high bits * 2^-24; gen_range = 23 bits in [1,2) - 1,
    * scale + low)."""

    def __init__(self, seed):
        self.s = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) & 0xFFFFFFFFFFFFFFFF

    def next_u32(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        return z >> 32

    def gen(self):
        return F(self.next_u32() >> 8) * F(1.0 / 16777216.0)

    def gen_range(self, low, high):
        low, high = F(low), F(high)
        scale = high - low
        while True:
            bits = np.uint32((self.next_u32() >> 9) | 0x3F800000)
            v01 = bits.view(F) - F(1.0)
            res = v01 * scale + low
            if res < high:
                return res

    def gen_index(self, n):
        return self.next_u32() % n

    def random_vec(self):
        """utils.rs:6-12"""
        return np.array([self.gen_range(-1.0, 1.0), self.gen_range(-1.0, 1.0), self.gen_range(-1.0, 1.0)], dtype=F)


def earth_texture():
    return ImageTexture.open(_EARTH)


# ------------------------------------------------------------------------------------------- cameras
def _cam(**kw):
    return CameraConfig(**kw)


CAMERAS = {
    # main.rs:33-45
    "bouncing-spheres": lambda: _cam(width=800, height=600, samples_per_pixel=100, max_depth=50, lookfrom=(13, 2, 3),
                                     lookat=(0, 0, 0), vup=(0, 1, 0), vfov=20.0, defocus_angle=0.6, focus_dist=10.0),
    # main.rs:46-57
    "checkered-spheres": lambda: _cam(width=800, height=600, samples_per_pixel=100, max_depth=50, vfov=20.0,
                                      lookfrom=(13, 2, 3), lookat=(0, 0, 0), vup=(0, 1, 0), defocus_angle=0.0),
    # main.rs:58-68 (vfov left at the default 90)
    "earth": lambda: _cam(width=800, height=600, samples_per_pixel=100, max_depth=50, lookfrom=(0, 0, 12),
                          lookat=(0, 0, 0), vup=(0, 1, 0), defocus_angle=0.0),
    # main.rs:69-80
    "perlin-spheres": lambda: _cam(width=800, height=600, samples_per_pixel=100, max_depth=50, vfov=20.0,
                                   lookfrom=(13, 2, 3), lookat=(0, 0, 0), vup=(0, 1, 0), defocus_angle=0.0),
    # main.rs:81-92
    "quads": lambda: _cam(width=800, height=800, samples_per_pixel=100, max_depth=50, vfov=80.0, lookfrom=Z * F(9.0),
                          lookat=(0, 0, 0), vup=(0, 1, 0), defocus_angle=0.0),
    # main.rs:93-104 (vfov left at the default 90)
    "simple-light": lambda: _cam(width=800, height=600, samples_per_pixel=500, max_depth=50, background=(0, 0, 0),
                                 lookfrom=(26, 3, 6), lookat=Y * F(2.0), vup=(0, 1, 0), defocus_angle=0.0),
    # main.rs:105-117
    "cornell-box": lambda: _cam(width=600, height=600, samples_per_pixel=1000, max_depth=50, background=(0, 0, 0),
                                lookfrom=(278, 278, -800), lookat=(278, 278, 0), vup=(0, 1, 0), defocus_angle=0.0,
                                vfov=40.0),
    # main.rs:118-130
    "cornell-smoke": lambda: _cam(width=600, height=600, samples_per_pixel=200, max_depth=50, background=(0, 0, 0),
                                  lookfrom=(278, 278, -800), lookat=(278, 278, 0), vup=(0, 1, 0), defocus_angle=0.0,
                                  vfov=40.0),
    # main.rs:131-143
    "final-scene": lambda: _cam(width=800, height=800, samples_per_pixel=10000, max_depth=40, background=(0, 0, 0),
                                vfov=40.0, lookfrom=(478, 278, -600), lookat=(278, 278, 0), vup=(0, 1, 0),
                                defocus_angle=0.0),
}
# the reference has no default scene (the scene argument is required, main.rs:560); `default` is the
# five-sphere scene below under the bouncing-spheres camera at 400x225, 16 spp, depth 50
CAMERAS["triangle-mesh"] = lambda: CAMERAS["bouncing-spheres"]().replace(width=1920, height=1080, samples_per_pixel=256, defocus_angle=0.0)
CAMERAS["mixed"] = lambda: CAMERAS["bouncing-spheres"]().replace(width=3840, height=2160, samples_per_pixel=4096)
CAMERAS["default"] = lambda: CAMERAS["bouncing-spheres"]().replace(width=400, height=225, samples_per_pixel=16)


# ------------------------------------------------------------------------------------------- worlds
def _checker_ground():
    return Lambertian(CheckerTexture(0.32, SolidTexture.rgb(0.2, 0.3, 0.1), SolidTexture.rgb(0.9, 0.9, 0.9)))


def bouncing_spheres(seed=1):
    """main.rs:161-228"""
    rng = SceneRng(seed)
    world = World()
    world.push(Sphere.stationary(vec3(0.0, -1000.0, 0.0), 1000.0, _checker_ground()))
    for xi in range(-11, 11):
        for yi in range(-11, 11):
            x, y = F(xi), F(yi)
            cx = x + F(0.9) * rng.gen()
            cz = y + F(0.9) * rng.gen()
            center = vec3(cx, F(0.2), cz)
            d = center - vec3(4.0, 0.2, 0.0)
            if np.sqrt(F(d[0] * d[0]) + F(d[1] * d[1]) + F(d[2] * d[2])) <= F(0.9):
                continue
            choose_mat = rng.gen()
            if choose_mat < F(0.8):
                albedo = rng.random_vec() * rng.random_vec()  # components in (-1,1): negative albedos are legal
                to_center = center + Y * rng.gen_range(0.0, 0.5)
                world.push(Sphere.moving(center, to_center, 0.2, Lambertian.rgb(*albedo)))
            elif choose_mat < F(0.95):
                albedo = rng.random_vec()
                world.push(Sphere.stationary(center, 0.2, Metal.rgb(*albedo).with_roughness(rng.gen())))
            else:
                world.push(Sphere.stationary(center, 0.2, Dielectric()))
    world.push(Sphere.stationary(vec3(0.0, 1.0, 0.0), 1.0, Dielectric()))
    world.push(Sphere.stationary(vec3(-4.0, 1.0, 0.0), 1.0, Lambertian.rgb(0.4, 0.2, 0.1)))
    world.push(Sphere.stationary(vec3(4.0, 1.0, 0.0), 1.0, Metal.rgb(0.7, 0.6, 0.5).with_roughness(0.0)))
    return world


def checkered_spheres(seed=0):
    """main.rs:230-243"""
    world = World()
    checker = _checker_ground()
    world.push(Sphere.stationary(Y * F(-10.0), 10.0, checker))
    world.push(Sphere.stationary(Y * F(10.0), 10.0, checker))
    return world


def earth(seed=0):
    """main.rs:245-253"""
    world = World()
    world.push(Sphere.stationary(vec3(0, 0, 0), 8.0, Lambertian(earth_texture())))
    return world


def perlin_spheres(seed=1):
    """main.rs:255-269"""
    world = World()
    surface = Lambertian(NoiseTexture.new(SceneRng(seed), 4.0))
    world.push(Sphere.stationary(Y * F(-1000.0), 1000.0, surface))
    world.push(Sphere.stationary(Y * F(2.0), 2.0, surface))
    return world


def quads(seed=0):
    """main.rs:271-305"""
    world = World()
    world.push(Quad(vec3(-3.0, -2.0, 5.0), Z * F(-4.0), Y * F(4.0), Lambertian.rgb(1.0, 0.2, 0.2)))
    world.push(Quad(vec3(-2.0, -2.0, 0.0), X * F(4.0), Y * F(4.0), Lambertian.rgb(0.2, 1.0, 0.2)))
    world.push(Quad(vec3(3.0, -2.0, 1.0), Z * F(4.0), Y * F(4.0), Lambertian.rgb(0.2, 0.2, 1.0)))
    world.push(Quad(vec3(-2.0, 3.0, 1.0), X * F(4.0), Z * F(4.0), Lambertian.rgb(1.0, 0.5, 0.0)))
    world.push(Quad(vec3(-2.0, -3.0, 5.0), X * F(4.0), Z * F(-4.0), Lambertian.rgb(0.2, 0.8, 0.8)))
    return world


def simple_light(seed=1):
    """main.rs:307-326"""
    pertext = Lambertian(NoiseTexture.new(SceneRng(seed), 4.0))
    difflight = DiffuseLight.rgb(4.0, 4.0, 4.0)
    world = World()
    world.push(Sphere.stationary(Y * F(-1000.0), 1000.0, pertext))
    world.push(Sphere.stationary(Y * F(2.0), 2.0, pertext))
    world.push(Quad(vec3(3.0, 1.0, -2.0), X * F(2.0), Y * F(2.0), difflight))
    world.push(Sphere.stationary(Y * F(7.0), 2.0, difflight))
    return world


def _cornell_walls(world, light):
    white = Lambertian.rgb(0.73, 0.73, 0.73)
    world.push(Quad(X * F(555.0), Y * F(555.0), Z * F(555.0), Lambertian.rgb(0.12, 0.45, 0.15)))
    world.push(Quad(vec3(0, 0, 0), Y * F(555.0), Z * F(555.0), Lambertian.rgb(0.65, 0.05, 0.05)))
    world.push(light)
    world.push(Quad(vec3(0, 0, 0), X * F(555.0), Z * F(555.0), white))
    world.push(Quad(vec3(1, 1, 1) * F(555.0), X * F(-555.0), Z * F(-555.0), white))
    world.push(Quad(Z * F(555.0), X * F(555.0), Y * F(555.0), white))
    return white


def cornell_box(seed=0):
    """main.rs:328-385"""
    world = World()
    white = _cornell_walls(world, Quad(vec3(343.0, 554.0, 332.0), X * F(-130.0), Z * F(-105.0),
                                       DiffuseLight.rgb(15.0, 15.0, 15.0)))
    world.push(Translate(RotateY(Cube(vec3(0, 0, 0), vec3(165.0, 330.0, 165.0), white), 15.0), vec3(265.0, 0.0, 295.0)))
    world.push(Translate(RotateY(Cube(vec3(0, 0, 0), vec3(165.0, 165.0, 165.0), white), -18.0), vec3(130.0, 0.0, 65.0)))
    return world


def cornell_smoke(seed=0):
    """main.rs:387-447"""
    world = World()
    white = _cornell_walls(world, Quad(vec3(113.0, 554.0, 127.0), X * F(330.0), Z * F(305.0),
                                       DiffuseLight.rgb(7.0, 7.0, 7.0)))
    box1 = Translate(RotateY(Cube(vec3(0, 0, 0), vec3(165.0, 330.0, 165.0), white), 15.0), vec3(265.0, 0.0, 295.0))
    world.push(ConstantMedium.from_color(box1, 0.01, (0, 0, 0)))
    box2 = Translate(RotateY(Cube(vec3(0, 0, 0), vec3(165.0, 165.0, 165.0), white), -18.0), vec3(130.0, 0.0, 65.0))
    world.push(ConstantMedium.from_color(box2, 0.01, (1, 1, 1)))
    return world


def final_scene(seed=1):
    """main.rs:450-540"""
    rng = SceneRng(seed)
    boxes1 = World()
    ground = Lambertian.rgb(0.48, 0.83, 0.53)
    for i in range(20):
        for j in range(20):
            w = F(100.0)
            x0 = F(-1000.0) + F(i) * w
            z0 = F(-1000.0) + F(j) * w
            y0 = F(0.0)
            x1 = x0 + w
            y1 = rng.gen_range(0.0, 100.0)  # 0.0..=100. in the reference: the closed end has measure zero
            z1 = z0 + w
            boxes1.push(Cube(vec3(x0, y0, z0), vec3(x1, y1, z1), ground))
    world = World()
    world.push(Bvh(boxes1))
    world.push(Quad(vec3(123.0, 554.0, 147.0), X * F(300.0), Z * F(265.0), DiffuseLight.rgb(7.0, 7.0, 7.0)))
    center1 = vec3(400.0, 400.0, 200.0)
    center2 = center1 + X * F(30.0)
    world.push(Sphere.moving(center1, center2, 50.0, Lambertian.rgb(0.7, 0.3, 0.1)))
    world.push(Sphere.stationary(vec3(260.0, 150.0, 45.0), 50.0, Dielectric(1.5)))
    world.push(Sphere.stationary(vec3(0.0, 150.0, 145.0), 50.0, Metal.rgb(0.8, 0.8, 0.9).with_roughness(1.0)))
    boundary = Sphere.stationary(vec3(360.0, 150.0, 145.0), 70.0, Dielectric(1.5))
    world.push(ConstantMedium.from_color(boundary, 0.2, (0.2, 0.4, 0.9)))
    boundary = Sphere.stationary(vec3(0, 0, 0), 5000.0, Dielectric(1.5))
    world.push(ConstantMedium.from_color(boundary, 0.0001, (1, 1, 1)))
    world.push(Sphere.stationary(vec3(400.0, 200.0, 400.0), 100.0, Lambertian(earth_texture())))
    pertext = NoiseTexture.new(rng, 0.2)
    world.push(Sphere.stationary(vec3(220.0, 280.0, 300.0), 80.0, Lambertian(pertext)))
    boxes2 = World()
    for _ in range(1000):
        c = vec3(rng.gen_range(0.0, 165.0), rng.gen_range(0.0, 165.0), rng.gen_range(0.0, 165.0))
        boxes2.push(Sphere.stationary(c, 10.0, Lambertian.rgb(0.73, 0.73, 0.73)))
    world.push(Translate(RotateY(Bvh(boxes2), 15.0), vec3(-100.0, 270.0, 395.0)))
    return world


def default_scene(seed=0):
    """The five-sphere scene the 400x225 / 16 spp configuration is quoted on: checker ground, glass, earth-textured
    Lambertian, polished metal and a small brown Lambertian, all through Sphere::stationary."""
    world = World()
    world.push(Sphere.stationary(vec3(0.0, -1000.0, 0.0), 1000.0, _checker_ground()))
    world.push(Sphere.stationary(vec3(0.0, 1.0, 0.0), 1.0, Dielectric(1.5)))
    world.push(Sphere.stationary(vec3(-4.0, 1.0, 0.0), 1.0, Lambertian(earth_texture())))
    world.push(Sphere.stationary(vec3(4.0, 1.0, 0.0), 1.0, Metal.rgb(0.7, 0.6, 0.5).with_roughness(0.0)))
    world.push(Sphere.stationary(vec3(1.5, 0.5, 2.5), 0.5, Lambertian.rgb(0.4, 0.2, 0.1)))
    return world


def displaced_sphere_mesh(n_lat, seed=1, center=(0.0, 1.2, 0.0), radius=1.2):
    """Synthetic triangle mesh (the reference has no triangle type or mesh loader, and no mesh asset is available
    offline): a latitude / longitude tessellation of a sphere whose radius is displaced by a few seeded sinusoids.
    n_lat rings x 2*n_lat segments -> 4 * n_lat * (n_lat - 1) triangles (n_lat = 500 gives 998 000)."""
    rng = np.random.default_rng(seed)
    n_lon = 2 * n_lat
    k = rng.integers(2, 9, size=(4, 2)).astype(np.float64)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    amp = np.array([0.10, 0.06, 0.04, 0.03])
    theta = np.linspace(0.0, np.pi, n_lat + 1)[:, None]
    phi = np.linspace(0.0, 2 * np.pi, n_lon, endpoint=False)[None, :]
    disp = sum(a * np.sin(kk[0] * theta + p) * np.cos(kk[1] * phi) for a, kk, p in zip(amp, k, ph))
    r = radius * (1.0 + disp * np.sin(theta) ** 2)  # undisplaced at the poles so that they stay single points
    x = r * np.sin(theta) * np.cos(phi) + center[0]
    y = r * np.cos(theta) + center[1] + 0 * phi
    z = r * np.sin(theta) * np.sin(phi) + center[2]
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(F)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = i * n_lon + j
    b = i * n_lon + (j + 1) % n_lon
    c = (i + 1) * n_lon + j
    d = (i + 1) * n_lon + (j + 1) % n_lon
    t1 = np.stack([a, c, b], axis=-1)[1:]      # the first ring's upper triangles are degenerate at the pole
    t2 = np.stack([b, c, d], axis=-1)[:-1]     # likewise the last ring's lower triangles
    tris = np.concatenate([t1.reshape(-1, 3), t2.reshape(-1, 3)]).astype(np.uint32)
    return verts, tris


def triangle_mesh(seed=1, n_lat=40):
    """Triangle-mesh scene: the displaced-sphere mesh (Lambertian) on the checker ground under the default background."""
    world = World()
    world.push(Sphere.stationary(vec3(0.0, -1000.0, 0.0), 1000.0, _checker_ground()))
    verts, tris = displaced_sphere_mesh(n_lat, seed)
    world.push(TriangleMesh(verts, tris, Lambertian.rgb(0.7, 0.45, 0.2)))
    return world


def mixed_scene(seed=1, n_lat=40):
    """Mixed scene: the five default spheres (earth texture, metal, glass) + a triangle mesh + three more glass spheres."""
    world = default_scene(seed)
    verts, tris = displaced_sphere_mesh(n_lat, seed, center=(-1.0, 0.8, 3.0), radius=0.8)
    world.push(TriangleMesh(verts, tris, Lambertian.rgb(0.25, 0.5, 0.7)))
    for c in [(2.5, 0.4, 4.0), (-2.5, 0.4, 1.5), (0.5, 0.3, 4.5)]:
        world.push(Sphere.stationary(vec3(*c), c[1], Dielectric(1.5)))
    return world


WORLDS = {
    "bouncing-spheres": bouncing_spheres,
    "checkered-spheres": checkered_spheres,
    "earth": earth,
    "perlin-spheres": perlin_spheres,
    "quads": quads,
    "simple-light": simple_light,
    "cornell-box": cornell_box,
    "cornell-smoke": cornell_smoke,
    "final-scene": final_scene,
    "default": default_scene,
    "triangle-mesh": triangle_mesh,
    "mixed": mixed_scene,
}


def make(name, seed=1, world_args=None, **camera_overrides):
    """(CameraConfig, Bvh) for a scene name; camera_overrides mirror the CLI's -w/--height/-s/-d/-v (main.rs:567-574)."""
    cfg = CAMERAS[name]().replace(**camera_overrides)
    return cfg, Bvh.from_world(WORLDS[name](seed, **(world_args or {})))
