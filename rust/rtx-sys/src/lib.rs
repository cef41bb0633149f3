//! Raw declarations of `include/rtx.h`, field for field.  Not compiled in the authoring environment (no Rust
//! toolchain there); layouts are pinned on the C side by static_asserts (32 / 48 / 16 / 32 / 28 / 44 bytes).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RTX_OK: c_int = 0;
pub const RTX_ERR_INVALID: c_int = -1;
pub const RTX_ERR_UNSUPPORTED: c_int = -2;
pub const RTX_ERR_DEVICE: c_int = -3;
pub const RTX_ERR_NO_DEVICE: c_int = -4;

pub type rtx_ref_t = u32;
pub const RTX_REF_NODE: u32 = 0;
pub const RTX_REF_SPHERE: u32 = 1;
pub const RTX_REF_QUAD: u32 = 2;
pub const RTX_REF_TRI: u32 = 3;
pub const RTX_REF_OBJECT: u32 = 4;
pub const RTX_MISS: u32 = 0xFFFF_FFFF;
#[inline]
pub const fn rtx_ref(kind: u32, index: u32) -> rtx_ref_t {
    (kind << 28) | (index & 0x0FFF_FFFF)
}

pub const RTX_OBJ_LIST: u32 = 0;
pub const RTX_OBJ_TRANSLATE: u32 = 1;
pub const RTX_OBJ_ROTATE_Y: u32 = 2;
pub const RTX_OBJ_MEDIUM: u32 = 3;

pub const RTX_MAT_LAMBERTIAN: u32 = 0;
pub const RTX_MAT_METAL: u32 = 1;
pub const RTX_MAT_DIELECTRIC: u32 = 2;
pub const RTX_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RTX_MAT_ISOTROPIC: u32 = 4;

pub const RTX_TEX_SOLID: u32 = 0;
pub const RTX_TEX_CHECKER: u32 = 1;
pub const RTX_TEX_IMAGE: u32 = 2;
pub const RTX_TEX_NOISE: u32 = 3;

pub const RTX_TRAVERSAL_FAST: u32 = 0;
pub const RTX_TRAVERSAL_STRICT: u32 = 1;
/// rtx_trace / rtx_traversal_dump only: the fast data as the render kernels walk it, exact primitive arithmetic.
pub const RTX_TRAVERSAL_FAST_FUSED: u32 = 2;
/// rtx_trace only: every step through the functions the render kernels call.
pub const RTX_TRAVERSAL_RENDER: u32 = 3;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_bvh_node_t {
    pub bounds: [f32; 6],
    pub left: rtx_ref_t,
    pub right: rtx_ref_t,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_object_t {
    pub kind: u32,
    pub child: rtx_ref_t,
    pub first: u32,
    pub count: u32,
    pub f: [f32; 3],
    pub material: u32,
    pub prim_id: u32,
    pub pad: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_material_t {
    pub kind: u32,
    pub texture: u32,
    pub param: f32,
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_texture_t {
    pub kind: u32,
    pub scale: f32,
    pub a: u32,
    pub b: u32,
    pub rgb: [f32; 3],
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtx_image_t {
    pub rgb: *const u8,
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtx_perlin_t {
    pub values: *const f32,
    pub perm_x: *const u32,
    pub perm_y: *const u32,
    pub perm_z: *const u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtx_scene_desc_t {
    pub n_spheres: u32,
    pub sph_cx: *const f32,
    pub sph_cy: *const f32,
    pub sph_cz: *const f32,
    pub sph_dx: *const f32,
    pub sph_dy: *const f32,
    pub sph_dz: *const f32,
    pub sph_radius: *const f32,
    pub sph_material: *const u32,
    pub sph_prim_id: *const u32,

    pub n_quads: u32,
    pub quad_f: [*const f32; 16],
    pub quad_material: *const u32,
    pub quad_prim_id: *const u32,

    pub n_tris: u32,
    pub tri_f: [*const f32; 9],
    pub tri_material: *const u32,
    pub tri_prim_id: *const u32,

    pub n_objects: u32,
    pub objects: *const rtx_object_t,
    pub n_refs: u32,
    pub refs: *const rtx_ref_t,
    pub n_bvh_nodes: u32,
    pub bvh_nodes: *const rtx_bvh_node_t,
    pub root: rtx_ref_t,

    pub n_materials: u32,
    pub materials: *const rtx_material_t,
    pub n_textures: u32,
    pub textures: *const rtx_texture_t,
    pub n_images: u32,
    pub images: *const rtx_image_t,
    pub n_perlins: u32,
    pub perlins: *const rtx_perlin_t,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_camera_t {
    pub width: u32,
    pub height: u32,
    pub samples_per_pixel: u32,
    pub max_depth: u32,
    pub defocus_angle: f32,
    pub center: [f32; 3],
    pub pixel00_loc: [f32; 3],
    pub pixel_delta_u: [f32; 3],
    pub pixel_delta_v: [f32; 3],
    pub defocus_disk_u: [f32; 3],
    pub defocus_disk_v: [f32; 3],
    pub background: [f32; 3],
    pub pixel_samples_scale: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_ray_t {
    pub o: [f32; 3],
    pub d: [f32; 3],
    pub time: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_hit_t {
    pub prim_id: u32,
    pub t: f32,
    pub n: [f32; 3],
    pub uv: [f32; 2],
    pub front_face: u32,
    pub p: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_trace_opts_t {
    pub tmin: f32,
    pub tmax: f32,
    pub traversal: u32,
    pub seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_render_opts_t {
    pub seed: u64,
    pub sample_begin: u32,
    pub sample_end: u32,
    pub traversal: u32,
    pub rr_start_depth: u32,
    pub tile_rank: u32,
    pub tile_world: u32,
    /// rtx_render only: 0 or 1 = the scene's device; N > 1 = devices 0..N-1, driven from the one call.
    pub n_gpus: u32,
    /// RTX_SPLIT_SAMPLES (0) or RTX_SPLIT_TILES (1) when n_gpus > 1.
    pub split: u32,
    /// RTX_RENDER_* bits.
    pub flags: u32,
}

/// rtx_render_opts_t.flags: two calls with the same arguments return the same bits (a wave finishes one work item before
/// it starts the next; about 6 % slower on the benchmark scene).
pub const RTX_RENDER_REPRODUCIBLE: u32 = 1;

pub const RTX_SPLIT_SAMPLES: u32 = 0;
pub const RTX_SPLIT_TILES: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_stats_t {
    pub samples: u64,
    pub rays: u64,
    pub kernel_ms: f64,
    pub total_ms: f64,
    pub n_launches: u32,
    /// bit 0: the traversal entries were resident in LDS; bit 1: packed array; bit 2: wide form; bit 3 (rtx_render_multi): a
    /// share's copies were staged through the host (no peer access), see rtx_last_error()
    pub flags: u32,
    pub box_tests: u64,
    pub prim_tests: u64,
}

#[repr(C)]
pub struct rtx_scene_t {
    _opaque: [u8; 0],
}

extern "C" {
    pub fn rtx_device_count() -> c_int;
    pub fn rtx_last_error() -> *const c_char;
    pub fn rtx_scene_create(desc: *const rtx_scene_desc_t, out: *mut *mut rtx_scene_t) -> c_int;
    pub fn rtx_scene_destroy(scene: *mut rtx_scene_t);
    pub fn rtx_trace(scene: *const rtx_scene_t, rays: *const rtx_ray_t, n: usize, opts: *const rtx_trace_opts_t, hits_out: *mut rtx_hit_t) -> c_int;
    pub fn rtx_trace_device(scene: *const rtx_scene_t, rays_dev: *const rtx_ray_t, n: usize, opts: *const rtx_trace_opts_t, hits_dev: *mut rtx_hit_t,
                            stream: *mut c_void) -> c_int;
    pub fn rtx_render(scene: *mut rtx_scene_t, camera: *const rtx_camera_t, opts: *const rtx_render_opts_t, rgb8_out: *mut u8, accum_out: *mut f32,
                      stats: *mut rtx_stats_t) -> c_int;
    /// One blocking call that drives several GPUs (devices == null: 0..n_devices-1).
    pub fn rtx_render_multi(scene: *mut rtx_scene_t, camera: *const rtx_camera_t, opts: *const rtx_render_opts_t, devices: *const c_int,
                            n_devices: u32, split: u32, rgb8_out: *mut u8, accum_out: *mut f32, stats: *mut rtx_stats_t) -> c_int;
    pub fn rtx_render_accumulate(scene: *mut rtx_scene_t, camera: *const rtx_camera_t, opts: *const rtx_render_opts_t, accum_dev: *mut f32,
                                 stream: *mut c_void, stats: *mut rtx_stats_t) -> c_int;
    /// rtx_render_accumulate for callers that cannot allocate device memory: its scratch lives in a fixed array of
    /// RTX_STATIC_SCRATCH_BYTES inside the library (one per GPU); RTX_ERR_UNSUPPORTED for frames beyond it and for cameras
    /// the fast data was not prepared for.  Calls that overlap on one GPU race.
    pub fn rtx_render_accumulate_static(scene: *mut rtx_scene_t, camera: *const rtx_camera_t, opts: *const rtx_render_opts_t, accum_dev: *mut f32,
                                        stream: *mut c_void, stats: *mut rtx_stats_t) -> c_int;
    pub fn rtx_resolve(accum_dev: *const f32, width: u32, height: u32, scale: f32, rgb8_dev: *mut u8, stream: *mut c_void) -> c_int;
    /// Host-only inspection of the traversal data (no device is touched).
    pub fn rtx_traversal_dump(desc: *const rtx_scene_desc_t, traversal: u32, view: *const f32, grow: f32, entries_out: *mut u32, rank_out: *mut u32,
                              capacity: usize, n_entries: *mut usize) -> c_int;
    /// Host-only inspection of the wide form (four-child nodes of 64 bytes, five-slot primitive records).
    pub fn rtx_wide_dump(desc: *const rtx_scene_desc_t, grow: f32, nodes_out: *mut u32, node_capacity: usize, prims_out: *mut u32, rank_out: *mut u32,
                         slot_capacity: usize, n_nodes: *mut usize, n_slots: *mut usize, levels: *mut u32) -> c_int;
}

/// Size of the fixed scratch array of rtx_render_accumulate_static.
pub const RTX_STATIC_SCRATCH_BYTES: u32 = 64 << 20;

/// Text of the last failure on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = rtx_last_error();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
