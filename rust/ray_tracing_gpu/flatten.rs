//! Additive code for the `ray_tracing` crate (add as `src/flatten.rs`, `pub mod flatten;` in lib.rs, and the trait
//! methods below next to each type).  It turns the crate's scene objects into the flat arrays `librtx.so` takes and
//! adds `Camera::render_gpu`, the GPU counterpart of `Camera::render` (src/camera.rs:98).
//!
//! Nothing in the reference changes behaviour: `hit` / `scatter` / `value` stay, scene construction in main.rs stays,
//! and `main` swaps `camera.render(&bvh_world)` (src/main.rs:579) for `camera.render_gpu(&bvh_world)?`.
//!
//! Written without a Rust toolchain at hand (none in the authoring environment): treat it as reviewed source, not as
//! compiled code.  The C++ host layer of this repository (`ray-tracer_amd/host/`) implements the same flattening and
//! is what the parity tests run.
//!
//! Why trait methods: `Mesh`, `Material` and `Texture` are used as opaque trait objects (`Arc<dyn Mesh + Sync + Send>`,
//! src/mesh.rs:55) and the concrete structs keep their fields private (src/mesh.rs:104-110, :190-199), so only code
//! inside the defining modules can read what has to be exported.

use std::collections::HashMap;

use rtx_sys::*;

/// Collects the flat scene.  Indices handed out here are the ones `rtx_scene_desc_t` refers to.
#[derive(Default)]
pub struct SceneBuilder {
    // spheres (src/mesh.rs:104-110)
    sph: [Vec<f32>; 7], // cx cy cz dx dy dz radius
    sph_material: Vec<u32>,
    sph_prim_id: Vec<u32>,
    // quads (src/mesh.rs:190-199): Q u v normal w d
    quad: [Vec<f32>; 16],
    quad_material: Vec<u32>,
    quad_prim_id: Vec<u32>,
    objects: Vec<rtx_object_t>,
    refs: Vec<rtx_ref_t>,
    nodes: Vec<rtx_bvh_node_t>,
    materials: Vec<rtx_material_t>,
    textures: Vec<rtx_texture_t>,
    image_bytes: Vec<Vec<u8>>,
    image_dims: Vec<(u32, u32)>,
    perlin_values: Vec<Vec<f32>>,
    perlin_perm: Vec<[Vec<u32>; 3]>,
    next_prim_id: u32,
    /// `Arc`-shared meshes (the same object under two parents) are flattened once.
    seen: HashMap<*const (), rtx_ref_t>,
}

impl SceneBuilder {
    pub fn shared(&mut self, key: *const (), make: impl FnOnce(&mut Self) -> rtx_ref_t) -> rtx_ref_t {
        if let Some(r) = self.seen.get(&key) {
            return *r;
        }
        let r = make(self);
        self.seen.insert(key, r);
        r
    }

    fn prim_id(&mut self) -> u32 {
        let id = self.next_prim_id;
        self.next_prim_id += 1;
        id
    }

    pub fn push_sphere(&mut self, c0: [f32; 3], dc: [f32; 3], radius: f32, material: u32) -> rtx_ref_t {
        let i = self.sph_material.len() as u32;
        for k in 0..3 {
            self.sph[k].push(c0[k]);
            self.sph[3 + k].push(dc[k]);
        }
        self.sph[6].push(radius);
        self.sph_material.push(material);
        let id = self.prim_id();
        self.sph_prim_id.push(id);
        rtx_ref(RTX_REF_SPHERE, i)
    }

    /// `q`, `u`, `v` as given to `Quad::new`, `normal`, `w`, `d` as it derived them (src/mesh.rs:233-244): the
    /// derived values are exported, not recomputed, so the device tests the very same plane.
    pub fn push_quad(&mut self, q: [f32; 3], u: [f32; 3], v: [f32; 3], normal: [f32; 3], w: [f32; 3], d: f32, material: u32) -> rtx_ref_t {
        let i = self.quad_material.len() as u32;
        for k in 0..3 {
            self.quad[k].push(q[k]);
            self.quad[3 + k].push(u[k]);
            self.quad[6 + k].push(v[k]);
            self.quad[9 + k].push(normal[k]);
            self.quad[12 + k].push(w[k]);
        }
        self.quad[15].push(d);
        self.quad_material.push(material);
        let id = self.prim_id();
        self.quad_prim_id.push(id);
        rtx_ref(RTX_REF_QUAD, i)
    }

    /// `World` / `Cube`: a linear closest-hit scan over the children in push order (src/mesh.rs:66-82).
    pub fn push_list(&mut self, children: &[rtx_ref_t]) -> rtx_ref_t {
        let first = self.refs.len() as u32;
        self.refs.extend_from_slice(children);
        self.push_object(rtx_object_t { kind: RTX_OBJ_LIST, first, count: children.len() as u32, ..Default::default() })
    }

    pub fn push_translate(&mut self, child: rtx_ref_t, offset: [f32; 3]) -> rtx_ref_t {
        self.push_object(rtx_object_t { kind: RTX_OBJ_TRANSLATE, child, f: offset, ..Default::default() })
    }

    /// `sin_theta`, `cos_theta` as `RotateY::new` stored them (src/mesh.rs:396-400).
    pub fn push_rotate_y(&mut self, child: rtx_ref_t, sin_theta: f32, cos_theta: f32) -> rtx_ref_t {
        self.push_object(rtx_object_t { kind: RTX_OBJ_ROTATE_Y, child, f: [sin_theta, cos_theta, 0.0], ..Default::default() })
    }

    /// `density` is the field of `ConstantMedium` (src/mesh.rs:630); `material` its `Isotropic`.
    pub fn push_medium(&mut self, boundary: rtx_ref_t, density: f32, material: u32) -> rtx_ref_t {
        let prim_id = self.prim_id();
        self.push_object(rtx_object_t { kind: RTX_OBJ_MEDIUM, child: boundary, f: [density, 0.0, 0.0], material, prim_id, ..Default::default() })
    }

    fn push_object(&mut self, o: rtx_object_t) -> rtx_ref_t {
        self.objects.push(o);
        rtx_ref(RTX_REF_OBJECT, self.objects.len() as u32 - 1)
    }

    /// Pre-order node allocation keeps index 0 for the top-level `Bvh`.
    pub fn reserve_node(&mut self) -> u32 {
        self.nodes.push(rtx_bvh_node_t::default());
        self.nodes.len() as u32 - 1
    }

    /// `bounds` = x.start, x.end, y.start, y.end, z.start, z.end of the box the `Bvh` stores (src/mesh.rs:557), i.e.
    /// after `pad_to_minimus` and including the origin as `Aabb::default().merge(..)` makes it (src/mesh.rs:567-569).
    pub fn set_node(&mut self, index: u32, bounds: [f32; 6], left: rtx_ref_t, right: rtx_ref_t) {
        self.nodes[index as usize] = rtx_bvh_node_t { bounds, left, right };
    }

    pub fn push_material(&mut self, kind: u32, texture: u32, param: f32) -> u32 {
        self.materials.push(rtx_material_t { kind, texture, param, pad: 0 });
        self.materials.len() as u32 - 1
    }

    pub fn push_solid(&mut self, rgb: [f32; 3]) -> u32 {
        self.push_texture(rtx_texture_t { kind: RTX_TEX_SOLID, rgb, ..Default::default() })
    }

    pub fn push_checker(&mut self, scale: f32, even: u32, odd: u32) -> u32 {
        self.push_texture(rtx_texture_t { kind: RTX_TEX_CHECKER, scale, a: even, b: odd, ..Default::default() })
    }

    /// `rgb8` = `image.to_rgb8().into_raw()` of the `DynamicImage` (src/texture.rs:84): the texel bytes
    /// `get_pixel` reads (src/texture.rs:104-106), row-major, top row first.
    pub fn push_image(&mut self, rgb8: Vec<u8>, width: u32, height: u32) -> u32 {
        self.image_bytes.push(rgb8);
        self.image_dims.push((width, height));
        let a = self.image_bytes.len() as u32 - 1;
        self.push_texture(rtx_texture_t { kind: RTX_TEX_IMAGE, a, ..Default::default() })
    }

    /// The four tables of `Perlin<256>` (src/texture.rs:140-145) as drawn at construction.
    pub fn push_noise(&mut self, scale: f32, values: &[[f32; 3]; 256], perm: [&[usize; 256]; 3]) -> u32 {
        self.perlin_values.push(values.iter().flatten().copied().collect());
        self.perlin_perm.push([0, 1, 2].map(|k| perm[k].iter().map(|&p| p as u32).collect()));
        let a = self.perlin_values.len() as u32 - 1;
        self.push_texture(rtx_texture_t { kind: RTX_TEX_NOISE, scale, a, ..Default::default() })
    }

    fn push_texture(&mut self, t: rtx_texture_t) -> u32 {
        self.textures.push(t);
        self.textures.len() as u32 - 1
    }

    /// Runs `f` with a description that borrows this builder's storage (valid only inside the call).
    pub fn with_desc<R>(&self, root: rtx_ref_t, f: impl FnOnce(&rtx_scene_desc_t) -> R) -> R {
        let images: Vec<rtx_image_t> = self
            .image_bytes
            .iter()
            .zip(&self.image_dims)
            .map(|(b, &(width, height))| rtx_image_t { rgb: b.as_ptr(), width, height })
            .collect();
        let perlins: Vec<rtx_perlin_t> = self
            .perlin_values
            .iter()
            .zip(&self.perlin_perm)
            .map(|(v, p)| rtx_perlin_t { values: v.as_ptr(), perm_x: p[0].as_ptr(), perm_y: p[1].as_ptr(), perm_z: p[2].as_ptr() })
            .collect();
        let mut quad_f = [std::ptr::null(); 16];
        for k in 0..16 {
            quad_f[k] = self.quad[k].as_ptr();
        }
        let desc = rtx_scene_desc_t {
            n_spheres: self.sph_material.len() as u32,
            sph_cx: self.sph[0].as_ptr(),
            sph_cy: self.sph[1].as_ptr(),
            sph_cz: self.sph[2].as_ptr(),
            sph_dx: self.sph[3].as_ptr(),
            sph_dy: self.sph[4].as_ptr(),
            sph_dz: self.sph[5].as_ptr(),
            sph_radius: self.sph[6].as_ptr(),
            sph_material: self.sph_material.as_ptr(),
            sph_prim_id: self.sph_prim_id.as_ptr(),
            n_quads: self.quad_material.len() as u32,
            quad_f,
            quad_material: self.quad_material.as_ptr(),
            quad_prim_id: self.quad_prim_id.as_ptr(),
            n_tris: 0, // the reference has no triangle type
            tri_f: [std::ptr::null(); 9],
            tri_material: std::ptr::null(),
            tri_prim_id: std::ptr::null(),
            n_objects: self.objects.len() as u32,
            objects: self.objects.as_ptr(),
            n_refs: self.refs.len() as u32,
            refs: self.refs.as_ptr(),
            n_bvh_nodes: self.nodes.len() as u32,
            bvh_nodes: self.nodes.as_ptr(),
            root,
            n_materials: self.materials.len() as u32,
            materials: self.materials.as_ptr(),
            n_textures: self.textures.len() as u32,
            textures: self.textures.as_ptr(),
            n_images: images.len() as u32,
            images: images.as_ptr(),
            n_perlins: perlins.len() as u32,
            perlins: perlins.as_ptr(),
        };
        f(&desc)
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Trait methods to add (each next to the type it reads; shown here in one place).
//
// src/texture.rs
//     pub trait Texture { fn value(..) -> LinearRgba;  fn flatten(&self, b: &mut SceneBuilder) -> u32; }
//     impl Texture for SolidTexture      { fn flatten(&self, b) -> u32 { b.push_solid([self.albedo.red, self.albedo.green, self.albedo.blue]) } }
//     impl<E, O> Texture for CheckerTexture<E, O> {
//         fn flatten(&self, b) -> u32 { let e = self.even.flatten(b); let o = self.odd.flatten(b); b.push_checker(self.scale, e, o) } }
//         // `scale` is the field as stored; value() multiplies the point by it (src/texture.rs:63-67)
//     impl Texture for ImageTexture      { fn flatten(&self, b) -> u32 { let i = self.image.to_rgb8(); let (w, h) = i.dimensions(); b.push_image(i.into_raw(), w, h) } }
//     impl Texture for NoiseTexture<256> { fn flatten(&self, b) -> u32 {
//         let v = self.perlin.values.map(|v| [v.x, v.y, v.z]);
//         b.push_noise(self.scale, &v, [&self.perlin.perm_x, &self.perlin.perm_y, &self.perlin.perm_z]) } }
//
// src/material.rs
//     pub trait Material { fn scatter(..); fn emitted(..);  fn flatten(&self, b: &mut SceneBuilder) -> u32; }
//     Lambertian<T>   -> b.push_material(RTX_MAT_LAMBERTIAN,    self.texture.flatten(b), 0.0)
//     Metal<T>        -> b.push_material(RTX_MAT_METAL,         self.texture.flatten(b), self.roughness)
//     Dielectric      -> b.push_material(RTX_MAT_DIELECTRIC,    0,                       self.refraction_index)
//     DiffuseLight<T> -> b.push_material(RTX_MAT_DIFFUSE_LIGHT, self.texture.flatten(b), 0.0)
//     Isotropic<T>    -> b.push_material(RTX_MAT_ISOTROPIC,     self.texture.flatten(b), 0.0)
//
// src/mesh.rs
//     pub trait Mesh { fn hit(..); fn bounding_box(&self) -> &Aabb;  fn flatten(&self, b: &mut SceneBuilder) -> rtx_ref_t; }
//     fn v3(v: Vec3) -> [f32; 3] { [v.x, v.y, v.z] }
//     fn box6(a: &Aabb) -> [f32; 6] { [a.x.start(), a.x.end(), a.y.start(), a.y.end(), a.z.start(), a.z.end()] }
//     Sphere<M>  -> { let m = self.material.flatten(b); b.push_sphere(v3(self.initial_center), v3(self.center_delta), self.radius, m) }
//     Quad<M>    -> { let m = self.material.flatten(b);
//                     b.push_quad(v3(self.translation), v3(self.u), v3(self.v), v3(self.normal), v3(self.w), self.d, m) }
//     World      -> { let c: Vec<_> = self.meshes.iter().map(|m| b.shared(Arc::as_ptr(m) as *const (), |b| m.flatten(b))).collect(); b.push_list(&c) }
//     Cube       -> self.0.flatten(b)
//     Translate<T> -> { let c = self.object.flatten(b); b.push_translate(c, v3(self.offset)) }
//     RotateY<T>   -> { let c = self.object.flatten(b); b.push_rotate_y(c, self.sin_theta, self.cos_theta) }
//     ConstantMedium<B, T> -> { let c = self.boundary.flatten(b); let m = self.material.flatten(b); b.push_medium(c, self.density, m) }
//         // `density` is stored as given; hit() negates its reciprocal per call (src/mesh.rs:659)
//     Bvh        -> { let i = b.reserve_node();
//                     let l = b.shared(Arc::as_ptr(&self.left) as *const (), |b| self.left.flatten(b));
//                     let r = b.shared(Arc::as_ptr(&self.right) as *const (), |b| self.right.flatten(b));   // left == right for one element
//                     b.set_node(i, box6(&self.bbox), l, r); rtx_ref(RTX_REF_NODE, i) }
//
// src/camera.rs
//     impl Camera {
//         fn to_ffi(&self) -> rtx_camera_t {
//             let bg = self.config.background.to_linear();
//             rtx_camera_t {
//                 width: self.config.width, height: self.config.height,
//                 samples_per_pixel: self.config.samples_per_pixel as u32, max_depth: self.config.max_depth as u32,
//                 defocus_angle: self.config.defocus_angle,
//                 center: v3(self.center), pixel00_loc: v3(self.pixel00_loc),
//                 pixel_delta_u: v3(self.pixel_delta_u), pixel_delta_v: v3(self.pixel_delta_v),
//                 defocus_disk_u: v3(self.defocus_disk_u), defocus_disk_v: v3(self.defocus_disk_v),
//                 background: [bg.red, bg.green, bg.blue], pixel_samples_scale: self.pixel_samples_scale,
//             }
//         }
//         pub fn render_gpu(&self, mesh: &impl Mesh) -> Result<RgbImage, String> { crate::flatten::render_gpu(self.to_ffi(), mesh) }
//     }
// ---------------------------------------------------------------------------------------------------------------

/// Owns a device scene for the duration of a render.
struct DeviceScene(*mut rtx_scene_t);
impl Drop for DeviceScene {
    fn drop(&mut self) {
        unsafe { rtx_scene_destroy(self.0) }
    }
}

fn check(rc: i32) -> Result<(), String> {
    if rc == RTX_OK { Ok(()) } else { Err(format!("librtx error {rc}: {}", last_error())) }
}

/// `Camera::render` on the GPU: flatten, upload, render, wrap the bytes.  Blocks like the reference call; errors come
/// back as `Err` instead of panicking across the boundary.  `RTX_ERR_NO_DEVICE` means no gfx950 GPU is visible: the
/// caller may then fall back to `Camera::render`, the library itself never renders on the CPU.
pub fn render_gpu(camera: rtx_camera_t, mesh: &impl crate::mesh::Mesh) -> Result<image::RgbImage, String> {
    let mut b = SceneBuilder::default();
    let root = mesh.flatten(&mut b);
    let scene = b.with_desc(root, |desc| {
        let mut s = std::ptr::null_mut();
        check(unsafe { rtx_scene_create(desc, &mut s) }).map(|_| DeviceScene(s))
    })?;
    let mut rgb = vec![0u8; camera.width as usize * camera.height as usize * 3];
    // the reference seeds from the OS on every run (thread_rng); so does this call
    let opts = rtx_render_opts_t { seed: rand::random(), ..Default::default() };
    check(unsafe { rtx_render(scene.0, &camera, &opts, rgb.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()) })?;
    image::RgbImage::from_raw(camera.width, camera.height, rgb).ok_or_else(|| "image size mismatch".to_string())
}
