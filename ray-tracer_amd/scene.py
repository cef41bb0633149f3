"""Scene description types named after the reference's constructors.

The reference builds a scene out of Rust objects (`Sphere::stationary`, `Quad::new`, `Lambertian::rgb`,
`World::push`, `Bvh::from`, `Camera::new`; /root/reference/src/{mesh,material,texture,camera}.rs).  The
classes here carry exactly the constructor arguments of those types and nothing else: they hold no
arithmetic.  `lower()` walks a description and replays it, constructor by constructor, against a native
library that owns the real objects (the C++ host layer of this package, which flattens them for the GPU).
All floats are kept as numpy float32 so that nothing is silently computed in double precision.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def vec3(x, y=None, z=None):
    """Vec3::new(x, y, z) as a float32 triple."""
    if y is None:
        a = np.asarray(x, dtype=F).reshape(3)
        return a.copy()
    return np.array([x, y, z], dtype=F)


# --------------------------------------------------------------------------- textures (texture.rs)
class SolidTexture:
    """texture.rs:14-44"""

    def __init__(self, albedo):
        self.albedo = vec3(albedo)

    @classmethod
    def rgb(cls, r, g, b):
        return cls(vec3(r, g, b))


class CheckerTexture:
    """texture.rs:46-81; default scale 1.0, even (0.2,0.3,0.1), odd (0.9,0.9,0.9)"""

    def __init__(self, scale=1.0, even=None, odd=None):
        self.scale = F(scale)
        self.even = even if even is not None else SolidTexture.rgb(0.2, 0.3, 0.1)
        self.odd = odd if odd is not None else SolidTexture.rgb(0.9, 0.9, 0.9)

    def with_scale(self, scale):
        return CheckerTexture(scale, self.even, self.odd)


class ImageTexture:
    """texture.rs:83-115.  Holds the decoded RGB8 raster (H x W x 3, top row first)."""

    def __init__(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self.rgb = rgb

    @classmethod
    def open(cls, path):
        from PIL import Image

        return cls(np.asarray(Image.open(path).convert("RGB")))


class NoiseTexture:
    """texture.rs:117-138.  The Perlin tables (texture.rs:148-185) are drawn from thread_rng in the
    reference, i.e. they differ per run; here they are explicit inputs generated from a seed."""

    def __init__(self, scale, values, perm_x, perm_y, perm_z):
        self.scale = F(scale)
        self.values = np.ascontiguousarray(values, dtype=F).reshape(256, 3)
        self.perm_x = np.ascontiguousarray(perm_x, dtype=np.uint32).reshape(256)
        self.perm_y = np.ascontiguousarray(perm_y, dtype=np.uint32).reshape(256)
        self.perm_z = np.ascontiguousarray(perm_z, dtype=np.uint32).reshape(256)

    @classmethod
    def new(cls, rng, scale):
        """NoiseTexture::new(rng, scale): values in [-1,1)^3 un-normalised; perm via swaps with
        gen_range(0..i) (exclusive upper bound, texture.rs:177-182)."""
        values = np.array([[rng.gen_range(-1.0, 1.0) for _ in range(3)] for _ in range(256)], dtype=F)

        def perm():
            p = list(range(256))
            for i in range(255, 0, -1):
                t = rng.gen_index(i)
                p[i], p[t] = p[t], p[i]
            return np.array(p, dtype=np.uint32)

        return cls(scale, values, perm(), perm(), perm())


# --------------------------------------------------------------------------- materials (material.rs)
class Lambertian:
    """material.rs:53-96"""

    def __init__(self, texture):
        self.texture = texture

    @classmethod
    def rgb(cls, r, g, b):
        return cls(SolidTexture.rgb(r, g, b))


class Metal:
    """material.rs:98-135"""

    def __init__(self, texture, roughness=0.0):
        self.texture = texture
        self.roughness = F(roughness)

    @classmethod
    def rgb(cls, r, g, b):
        return cls(SolidTexture.rgb(r, g, b), 0.0)

    def with_roughness(self, roughness):
        return Metal(self.texture, roughness)


class Dielectric:
    """material.rs:137-179; default refraction_index 1.5"""

    def __init__(self, refraction_index=1.5):
        self.refraction_index = F(refraction_index)


class DiffuseLight:
    """material.rs:181-212"""

    def __init__(self, texture):
        self.texture = texture

    @classmethod
    def rgb(cls, r, g, b):
        return cls(SolidTexture.rgb(r, g, b))


class Isotropic:
    """material.rs:214-246"""

    def __init__(self, texture):
        self.texture = texture


# --------------------------------------------------------------------------- hittables (mesh.rs)
class Sphere:
    """mesh.rs:103-188"""

    def __init__(self, center, to, radius, material, moving):
        if not (radius >= 0):
            raise ValueError("Radius cannot be less than 0.")  # assert! mesh.rs:148,162
        self.center, self.to, self.radius, self.material, self.is_moving = vec3(center), vec3(to), F(radius), material, moving

    @classmethod
    def stationary(cls, center, radius, material):
        return cls(center, center, radius, material, False)

    @classmethod
    def moving(cls, frm, to, radius, material):
        return cls(frm, to, radius, material, True)


class Quad:
    """mesh.rs:190-257  Quad::new(translation, u, v, material)"""

    def __init__(self, translation, u, v, material):
        self.translation, self.u, self.v, self.material = vec3(translation), vec3(u), vec3(v), material


class Triangle:
    """Build-defined primitive (the reference has none): vertices v0, v1, v2."""

    def __init__(self, v0, v1, v2, material):
        self.v0, self.v1, self.v2, self.material = vec3(v0), vec3(v1), vec3(v2), material


class TriangleMesh:
    """Build-defined: an indexed triangle list pushed into a World as individual triangles."""

    def __init__(self, vertices, indices, material):
        self.vertices = np.ascontiguousarray(vertices, dtype=F).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self.material = material


class Cube:
    """mesh.rs:259-322  Cube::new(a, b, material)"""

    def __init__(self, a, b, material):
        self.a, self.b, self.material = vec3(a), vec3(b), material


class Translate:
    """mesh.rs:324-354"""

    def __init__(self, obj, offset):
        self.object, self.offset = obj, vec3(offset)


class RotateY:
    """mesh.rs:356-432 (angle in degrees)"""

    def __init__(self, obj, angle):
        self.object, self.angle = obj, F(angle)


class ConstantMedium:
    """mesh.rs:627-696"""

    def __init__(self, boundary, density, texture):
        self.boundary, self.density, self.texture = boundary, F(density), texture

    @classmethod
    def from_color(cls, boundary, density, rgb):
        return cls(boundary, density, SolidTexture(vec3(rgb)))


class World:
    """mesh.rs:54-101"""

    def __init__(self):
        self.meshes = []

    def push(self, mesh):
        self.meshes.append(mesh)
        return self


class Bvh:
    """mesh.rs:554-625  Bvh::from(&world)"""

    def __init__(self, world):
        if not world.meshes:
            raise ValueError("Bvh::from of an empty world never terminates in the reference")
        self.world = world

    @classmethod
    def from_world(cls, world):
        return cls(world)


class CameraConfig:
    """camera.rs:13-43 with the reference's defaults."""

    def __init__(self, width=100, height=100, samples_per_pixel=10, max_depth=10, vfov=90.0,
                 lookfrom=(0, 0, 0), lookat=(0, 0, -1), vup=(0, 1, 0), defocus_angle=0.0, focus_dist=10.0,
                 background=(0.7, 0.8, 1.0)):
        self.width, self.height = int(width), int(height)
        self.samples_per_pixel, self.max_depth = int(samples_per_pixel), int(max_depth)
        self.vfov = F(vfov)
        self.lookfrom, self.lookat, self.vup = vec3(lookfrom), vec3(lookat), vec3(vup)
        self.defocus_angle, self.focus_dist = F(defocus_angle), F(focus_dist)
        self.background = vec3(background)

    def replace(self, **kw):
        c = CameraConfig.__new__(CameraConfig)
        c.__dict__.update(self.__dict__)
        for k, v in kw.items():
            if v is None:
                continue
            cur = getattr(c, k)
            if isinstance(cur, np.ndarray):
                v = vec3(v)
            elif isinstance(cur, np.floating):
                v = F(v)
            else:
                v = int(v)
            setattr(c, k, v)
        return c


# --------------------------------------------------------------------------- lowering
def lower(root, be):
    """Replay the description `root` (a Bvh, World or any hittable) against backend `be`.

    Primitive ids are handed out here, in construction (depth-first, push) order: a sphere, quad or triangle takes
    one id, a cube six (one per face, in the order mesh.rs:283-318 pushes them), a constant medium one after its
    boundary.  Shared objects (Arc clones in the reference) are lowered once and shared.
    Returns (handle, n_prims).
    """
    memo = {}
    next_id = [0]

    def take(n=1):
        v = next_id[0]
        next_id[0] += n
        return v

    def tex(t):
        k = id(t)
        if k in memo:
            return memo[k]
        if isinstance(t, SolidTexture):
            h = be.tex_solid(t.albedo)
        elif isinstance(t, CheckerTexture):
            h = be.tex_checker(t.scale, tex(t.even), tex(t.odd))
        elif isinstance(t, ImageTexture):
            h = be.tex_image(t.rgb)
        elif isinstance(t, NoiseTexture):
            h = be.tex_noise(t.scale, t.values, t.perm_x, t.perm_y, t.perm_z)
        else:
            raise TypeError(f"not a texture: {t!r}")
        memo[k] = h
        return h

    def mat(m):
        k = id(m)
        if k in memo:
            return memo[k]
        if isinstance(m, Lambertian):
            h = be.mat_lambertian(tex(m.texture))
        elif isinstance(m, Metal):
            h = be.mat_metal(tex(m.texture), m.roughness)
        elif isinstance(m, Dielectric):
            h = be.mat_dielectric(m.refraction_index)
        elif isinstance(m, DiffuseLight):
            h = be.mat_diffuse_light(tex(m.texture))
        elif isinstance(m, Isotropic):
            h = be.mat_isotropic(tex(m.texture))
        else:
            raise TypeError(f"not a material: {m!r}")
        memo[k] = h
        return h

    def mesh(o):
        k = id(o)
        if k in memo:
            return memo[k]
        if isinstance(o, Sphere):
            if o.is_moving:
                h = be.sphere_moving(o.center, o.to, o.radius, mat(o.material), take())
            else:
                h = be.sphere_stationary(o.center, o.radius, mat(o.material), take())
        elif isinstance(o, Quad):
            h = be.quad(o.translation, o.u, o.v, mat(o.material), take())
        elif isinstance(o, Triangle):
            h = be.triangle(o.v0, o.v1, o.v2, mat(o.material), take())
        elif isinstance(o, Cube):
            h = be.cube(o.a, o.b, mat(o.material), take(6))
        elif isinstance(o, Translate):
            h = be.translate(mesh(o.object), o.offset)
        elif isinstance(o, RotateY):
            h = be.rotate_y(mesh(o.object), o.angle)
        elif isinstance(o, ConstantMedium):
            b = mesh(o.boundary)
            h = be.constant_medium(b, o.density, tex(o.texture), take())
        elif isinstance(o, World):
            h = be.world_new()
            for c in o.meshes:
                if isinstance(c, TriangleMesh):
                    be.world_push_trimesh(h, c.vertices, c.indices, mat(c.material), take(len(c.indices)))
                else:
                    be.world_push(h, mesh(c))
        elif isinstance(o, Bvh):
            h = be.bvh_from(mesh(o.world))
        else:
            raise TypeError(f"not a hittable: {o!r}")
        memo[k] = h
        return h

    h = mesh(root)
    return h, next_id[0]
