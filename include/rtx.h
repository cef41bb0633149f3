/*
 * rtx.h -- C ABI of the MI355X (gfx950) path-tracing backend for magmast/ray-tracer.
 *
 * The reference has no FFI of its own: its only seam is the Rust method
 *     pub fn render(&self, mesh: &(impl Mesh + Sync)) -> RgbImage          (src/camera.rs:98)
 * called once from main (src/main.rs:579) with the top-level Bvh (src/main.rs:577).  This header is
 * what a `-sys` crate for that seam binds: the Rust host walks its World / Bvh / Material / Texture
 * objects once, writes them into the flat arrays below (structure-of-arrays primitives, tagged
 * references, a reference-topology BVH) and calls rtx_render(); the bytes that come back are the
 * RgbImage (row-major, top row first, 3 bytes per pixel).
 *
 * Conventions
 *  - plain C, no torch / HIP types in any signature; `void* stream` is a hipStream_t (NULL = default).
 *  - every pointer inside a *_desc_t is borrowed for the duration of the call that receives it.
 *  - functions return 0 (RTX_OK) or a negative rtx_status; rtx_last_error() gives the thread-local text.
 *  - all floating point is IEEE binary32; hit decisions are computed without FMA contraction so that
 *    they are bit-identical to the reference's un-fused Rust arithmetic.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_INVALID = -1,     /* bad argument: null pointer, zero image, empty scene, NaN bounds, bad reference */
    RTX_ERR_UNSUPPORTED = -2, /* well-formed input that this build does not implement */
    RTX_ERR_DEVICE = -3,      /* HIP runtime failure (message carries hipGetErrorString) */
    RTX_ERR_NO_DEVICE = -4    /* no gfx950 device visible: there is no CPU fallback */
} rtx_status;

/* ---- tagged reference to a hittable: kind in bits [31:28], index in bits [27:0] --------------------- */
typedef uint32_t rtx_ref_t;
enum {
    RTX_REF_NODE = 0,   /* index into bvh_nodes        (mesh.rs:554-625  Bvh)                       */
    RTX_REF_SPHERE = 1, /* index into the sphere SoA   (mesh.rs:103-188  Sphere)                    */
    RTX_REF_QUAD = 2,   /* index into the quad SoA     (mesh.rs:190-257  Quad)                      */
    RTX_REF_TRI = 3,    /* index into the triangle SoA (no counterpart in the reference)            */
    RTX_REF_OBJECT = 4  /* index into objects          (World / Cube / Translate / RotateY / ConstantMedium) */
};
#define RTX_REF(kind, index) ((rtx_ref_t)(((uint32_t)(kind) << 28) | ((uint32_t)(index) & 0x0FFFFFFFu)))
#define RTX_REF_KIND(r) ((uint32_t)(r) >> 28)
#define RTX_REF_INDEX(r) ((uint32_t)(r) & 0x0FFFFFFFu)
#define RTX_MISS 0xFFFFFFFFu

/* One Bvh object of the reference (mesh.rs:554-558): its box and its two children, left first.
 * bounds = x.start, x.end, y.start, y.end, z.start, z.end -- the reference's padded, origin-including
 * boxes (mesh.rs:567-569, lib.rs:35-39) so that the strict traversal culls exactly as Bvh::hit does. */
typedef struct rtx_bvh_node {
    float bounds[6];
    rtx_ref_t left, right;
} rtx_bvh_node_t; /* 32 bytes */

/* Composite hittables. */
enum {
    RTX_OBJ_LIST = 0,      /* World / Cube (mesh.rs:54-101, :259-322): linear closest-hit scan of refs[first .. first+count) */
    RTX_OBJ_TRANSLATE = 1, /* mesh.rs:324-354: child, f[0..3] = offset                               */
    RTX_OBJ_ROTATE_Y = 2,  /* mesh.rs:356-432: child, f[0] = sin_theta, f[1] = cos_theta              */
    RTX_OBJ_MEDIUM = 3     /* mesh.rs:627-696: child = boundary, f[0] = density, material = Isotropic, prim_id.
                            * The boundary may be any hittable, one that contains further RTX_OBJ_MEDIUM objects included
                            * (the reference's `B: Mesh`; none of its scenes builds that): the inner media's hits then
                            * bound the outer medium as mesh.rs:635-638 has it.  Media may nest four deep -- a walk parks
                            * one boundary-query state per enclosing medium, in device memory the scene allocates when it
                            * nests any (60 bytes per resident lane); rtx_scene_create refuses deeper nesting with
                            * RTX_ERR_UNSUPPORTED.  Calls that walk such a scene (trace or render) must not overlap in
                            * time on one device copy: they would share that memory. */
};
typedef struct rtx_object {
    uint32_t kind;
    rtx_ref_t child; /* TRANSLATE / ROTATE_Y / MEDIUM */
    uint32_t first;  /* LIST: first index into refs */
    uint32_t count;  /* LIST: number of refs */
    float f[3];
    uint32_t material; /* MEDIUM */
    uint32_t prim_id;  /* MEDIUM: id reported by hits */
    uint32_t pad[3];
} rtx_object_t; /* 48 bytes */

/* ---- materials (material.rs) and textures (texture.rs) ---------------------------------------------- */
enum {
    RTX_MAT_LAMBERTIAN = 0,    /* material.rs:53-96                                   */
    RTX_MAT_METAL = 1,         /* material.rs:98-135    param = roughness             */
    RTX_MAT_DIELECTRIC = 2,    /* material.rs:137-179   param = refraction_index      */
    RTX_MAT_DIFFUSE_LIGHT = 3, /* material.rs:181-212                                 */
    RTX_MAT_ISOTROPIC = 4      /* material.rs:214-246                                 */
};
typedef struct rtx_material {
    uint32_t kind;
    uint32_t texture; /* index into textures (unused by DIELECTRIC) */
    float param;
    uint32_t pad;
} rtx_material_t; /* 16 bytes */

enum {
    RTX_TEX_SOLID = 0,   /* texture.rs:14-44    rgb                                                 */
    RTX_TEX_CHECKER = 1, /* texture.rs:46-81    scale, a = even texture, b = odd texture            */
    RTX_TEX_IMAGE = 2,   /* texture.rs:83-115   a = index into images                               */
    RTX_TEX_NOISE = 3    /* texture.rs:117-138  scale, a = index into perlins                       */
};
typedef struct rtx_texture {
    uint32_t kind;
    float scale;
    uint32_t a, b;
    float rgb[3];
    uint32_t pad;
} rtx_texture_t; /* 32 bytes */

/* Decoded raster of an ImageTexture (image::DynamicImage, texture.rs:84): RGB8, row-major, top row first. */
typedef struct rtx_image {
    const uint8_t* rgb;
    uint32_t width, height;
} rtx_image_t;

/* Perlin tables (texture.rs:140-145, N = 256).  The reference draws them from thread_rng at construction
 * (texture.rs:148-185), so they are inputs here: values = 256 xyz triples, perm_* = 256 entries each. */
typedef struct rtx_perlin {
    const float* values;
    const uint32_t* perm_x;
    const uint32_t* perm_y;
    const uint32_t* perm_z;
} rtx_perlin_t;

/* ---- the flattened scene ------------------------------------------------------------------------------ */
typedef struct rtx_scene_desc {
    /* spheres, structure of arrays (mesh.rs:104-110): initial_center, center_delta, radius */
    uint32_t n_spheres;
    const float *sph_cx, *sph_cy, *sph_cz;
    const float *sph_dx, *sph_dy, *sph_dz;
    const float* sph_radius;
    const uint32_t* sph_material;
    const uint32_t* sph_prim_id; /* id reported in rtx_hit_t.prim_id */

    /* quads, structure of arrays (mesh.rs:190-199): translation Q, u, v and the derived normal, d, w that
     * Quad::new computes (mesh.rs:233-244); quad_f[k] is component k of
     * { Qx,Qy,Qz, ux,uy,uz, vx,vy,vz, nx,ny,nz, wx,wy,wz, d } */
    uint32_t n_quads;
    const float* quad_f[16];
    const uint32_t* quad_material;
    const uint32_t* quad_prim_id;

    /* triangles, structure of arrays (build-defined primitive, absent from the reference):
     * tri_f[k] is component k of { v0x,v0y,v0z, v1x,v1y,v1z, v2x,v2y,v2z } */
    uint32_t n_tris;
    const float* tri_f[9];
    const uint32_t* tri_material;
    const uint32_t* tri_prim_id;

    /* composites and the reference-topology BVH (Bvh::from, mesh.rs:560-605) */
    uint32_t n_objects;
    const rtx_object_t* objects;
    uint32_t n_refs;
    const rtx_ref_t* refs;
    uint32_t n_bvh_nodes;
    const rtx_bvh_node_t* bvh_nodes;
    rtx_ref_t root; /* what Camera::render is handed: normally RTX_REF(RTX_REF_NODE, 0) */

    uint32_t n_materials;
    const rtx_material_t* materials;
    uint32_t n_textures;
    const rtx_texture_t* textures;
    uint32_t n_images;
    const rtx_image_t* images;
    uint32_t n_perlins;
    const rtx_perlin_t* perlins;
} rtx_scene_desc_t;

/* Camera constants exactly as Camera::new derives them (camera.rs:57-96) plus the config fields the
 * render loop reads (camera.rs:109-168). */
typedef struct rtx_camera {
    uint32_t width, height;
    uint32_t samples_per_pixel;
    uint32_t max_depth;
    float defocus_angle;
    float center[3];
    float pixel00_loc[3];
    float pixel_delta_u[3];
    float pixel_delta_v[3];
    float defocus_disk_u[3];
    float defocus_disk_v[3];
    float background[3];       /* linear RGB (camera.rs:141) */
    float pixel_samples_scale; /* 1 / samples_per_pixel (camera.rs:91) */
} rtx_camera_t;

typedef struct rtx_ray {
    float o[3];
    float d[3]; /* not normalised (camera.rs:158) */
    float time;
} rtx_ray_t; /* 28 bytes */

/* Hit record = the reference's Hit (mesh.rs:23-30) with the material replaced by ids. */
typedef struct rtx_hit {
    uint32_t prim_id; /* RTX_MISS when nothing was hit */
    float t;          /* Hit::distance, in ray-parameter units */
    float n[3];       /* Hit::normal, already flipped against the ray (mesh.rs:45) */
    float uv[2];
    uint32_t front_face; /* Hit::front_face as the reference stores it: always 1 (mesh.rs:47) */
    float p[3];          /* Hit::point */
} rtx_hit_t;             /* 44 bytes */

enum {
    RTX_TRAVERSAL_FAST = 0,  /* library-built SAH BVH over tight boxes, closest hit with the reference's tie-break */
    RTX_TRAVERSAL_STRICT = 1, /* walks bvh_nodes exactly as Bvh::hit does (mesh.rs:608-620), same boxes, same order */
    RTX_TRAVERSAL_FAST_FUSED = 2, /* rtx_trace / rtx_traversal_dump only: the fast data exactly as the render kernels walk
                                    it -- the packed form (16-byte slots, half-precision boxes in centre / half-extent
                                    form) when the scene has no Translate / RotateY / ConstantMedium, else the 32-byte
                                    entries -- with their box test (fused multiply-adds, hardware reciprocal).  Same
                                    hits as FAST for ray origins inside the scene's coordinate range or, after a
                                    render, within the reach of that render's camera: the boxes are padded for that. */
    RTX_TRAVERSAL_RENDER = 3 /* rtx_trace only: every step of the walk through the very functions the kernel of a
                                    RTX_TRAVERSAL_FAST render of this scene calls -- the data that kernel walks (the packed
                                    form, or the wide form for scenes too large for on-chip memory, or 32-byte entries with
                                    transforms / media), ray parameters scaled by 2^-60 with the interval start clamped
                                    to 0 where that kernel does so, and its square roots and quotients: the hardware's
                                    reciprocal square root / reciprocal plus one Newton step on an exactly computed
                                    residual.  That lands on the correctly rounded value for all but a vanishing share
                                    of operands, so hit ids, distances, points and normals equal FAST's bit for bit on
                                    practically every ray (on every one of ~9 million in the test-suite) -- but it is
                                    not guaranteed to, which is why the mode exists: the arithmetic the benchmark times
                                    is checked ray by ray against the reference's Sphere::hit / Quad::hit
                                    (mesh.rs:113-139, :202-221). */
};

typedef struct rtx_trace_opts {
    float tmin, tmax;   /* the Interval handed to Mesh::hit; Camera::ray_color uses [0.001, +inf) (camera.rs:127) */
    uint32_t traversal; /* RTX_TRAVERSAL_* */
    uint64_t seed;      /* feeds ConstantMedium's free-flight draw (mesh.rs:659) */
} rtx_trace_opts_t;

enum {
    RTX_SPLIT_SAMPLES = 0, /* every device renders a contiguous range of the sample indices of every pixel */
    RTX_SPLIT_TILES = 1    /* every device renders all samples of its interleaved share of the 8x8 pixel tiles */
};

typedef struct rtx_render_opts {
    uint64_t seed;
    uint32_t sample_begin, sample_end; /* sample index range to render.  rtx_render / rtx_render_multi read 0,0 as
                                          "all of 0..samples_per_pixel"; rtx_render_accumulate renders exactly the
                                          range it is given (begin == end: nothing). */
    uint32_t traversal;                /* RTX_TRAVERSAL_* */
    uint32_t rr_start_depth;           /* Russian roulette from this bounce on; 0 = off (the reference has none) */
    uint32_t tile_rank, tile_world;    /* tile split: render only the 8x8 pixel tiles t with t % tile_world == tile_rank
                                          (row-major tile index); tile_world 0 or 1 = all tiles.  Other pixels of
                                          accum_dev are left untouched, so per-rank buffers add up to the frame. */
    uint32_t n_gpus;                   /* rtx_render only: 0 or 1 = the scene's device; N > 1 = devices 0..N-1, driven
                                          from this one call (see rtx_render_multi) */
    uint32_t split;                    /* rtx_render with n_gpus > 1: RTX_SPLIT_* */
    uint32_t flags;                    /* RTX_RENDER_* */
} rtx_render_opts_t;

/* rtx_render_opts_t.flags */
enum {
    RTX_RENDER_REPRODUCIBLE = 1 /* Two calls with the same arguments return the same bits, whatever the sample count.
                                   By default a wave works on several (tile, sample batch) items at once, so the order
                                   in which the samples of one pixel are added depends on which items the wave happened
                                   to get before: frames agree to the order of fp32 additions only.  With this flag a
                                   wave finishes one item before it starts the next; what it then does is a function of
                                   the item alone.  Costs about 6 % on the benchmark scene (lanes idle while an item
                                   drains).  The reference itself is not reproducible at all (thread_rng), so this has
                                   no counterpart there. */
};

typedef struct rtx_stats {
    uint64_t samples;     /* pixel samples rendered by the call */
    uint64_t rays;        /* ray segments handed to the closest-hit search */
    double kernel_ms;     /* device time of the path-tracing kernel(s), HIP events on the launch stream */
    double total_ms;      /* wall time of the call */
    uint32_t n_launches;  /* launches of the dominant kernel */
    uint32_t flags;       /* bit 3 (rtx_render_multi): a share's device cannot reach the gathering device directly, so its copies
                           * were staged through host memory; rtx_last_error() then holds a note naming the devices.
                           * bit 2: the wide form was walked (four-child nodes of 64 B, primitives of 80 B; see rtx_wide_dump).
                           * bit 0: the traversal entries were resident in LDS (small scene); else walked from HBM / L2.
                           * bit 1: the packed array was walked (16-byte slots: a box is 16 B in HBM, 32 B in LDS; a sphere 32 B;
                           * a quad / triangle 80 B) instead of 32-byte entries with separate 80-byte planar records */
    uint64_t box_tests;   /* Aabb::hit evaluations the kernel performed (lanes x box steps), counted in the kernel */
    uint64_t prim_tests;  /* Sphere / Quad / triangle / bracket entries the kernel evaluated */
} rtx_stats_t;

typedef struct rtx_scene rtx_scene_t; /* device-resident scene */

/* Number of usable gfx950 devices; 0 when none (every compute entry point then fails with RTX_ERR_NO_DEVICE). */
int rtx_device_count(void);
/* Thread-local text of the last failure on this thread. */
const char* rtx_last_error(void);

/* Validate, upload to the current HIP device and build the traversal data.  Replaces the borrow of the
 * top-level Bvh that Camera::render holds for the duration of a render (camera.rs:98). */
int rtx_scene_create(const rtx_scene_desc_t* desc, rtx_scene_t** out);
void rtx_scene_destroy(rtx_scene_t* scene);

/* Closest hit of n rays -- the Mesh::hit contract (mesh.rs:17-21) of the top-level object, used by the
 * parity gate.  Host buffers. */
int rtx_trace(const rtx_scene_t* scene, const rtx_ray_t* rays, size_t n, const rtx_trace_opts_t* opts,
              rtx_hit_t* hits_out);
/* Same with device-resident rays / hits on `stream`; asynchronous. */
int rtx_trace_device(const rtx_scene_t* scene, const rtx_ray_t* rays_dev, size_t n, const rtx_trace_opts_t* opts,
                     rtx_hit_t* hits_dev, void* stream);

/* Camera::render (camera.rs:98-107): renders the whole frame and returns the RgbImage bytes.
 * rgb8_out: W*H*3 bytes, caller owned.  accum_out (nullable): 3*W*H f32, the per-pixel SUM of radiance
 * samples before the 1/spp scale (camera.rs:112-119).  stats nullable.  Host buffers; blocks.
 * opts->n_gpus > 1 renders on devices 0 .. n_gpus-1 (rtx_render_multi with devices == NULL). */
int rtx_render(rtx_scene_t* scene, const rtx_camera_t* camera, const rtx_render_opts_t* opts,
               uint8_t* rgb8_out, float* accum_out, rtx_stats_t* stats);

/* Camera::render on several GPUs from ONE blocking call of one host thread -- the reference's render is a single
 * blocking call (camera.rs:98, called at main.rs:579), so the library drives the devices itself: the scene's device
 * data is replicated onto every listed device (device-to-device copies, kept with the scene for later calls), every
 * device renders its share on its own stream -- split RTX_SPLIT_SAMPLES: the sample indices
 * [k*spp/N, (k+1)*spp/N) of every pixel; RTX_SPLIT_TILES: all samples of the 8x8 tiles t % N == k -- and pushes its
 * result into a staging slot on the first listed device over the GPU fabric (hipMemcpyPeerAsync): the fp32 sums,
 * or, for a tile split without accum_out, only its resolved RGB8 bytes.  The first device adds the slots in device
 * order and resolves.  Because the random stream is a function of (seed, pixel, sample index), the frame equals the
 * single-device frame up to the order of fp32 additions.  (Bit for bit only with a tile split AND one sample per batch
 * -- a call cuts its sample range into batches of roughly 20 or more samples, so that means a few samples per pixel:
 * the samples a batch gives one pixel meet in one accumulator in the order their paths finish, which varies from run
 * to run; which samples a pixel gets, and every sample's value, never do.)
 * The traversal data is ordered for the camera once, on the scene's own device (one host pass and upload, as in
 * rtx_render_accumulate); the other devices' copies are brought up to date device to device before any share is
 * launched, and no share waits for another device at launch.  If a device has no direct (peer) access to the first
 * one, its copies go through host memory: the call still succeeds, sets bit 3 of stats->flags and leaves a note that
 * names the devices in rtx_last_error().  A failure after the first share was launched waits for the shares already in
 * flight before it returns.
 * devices: n_devices HIP device ordinals, or NULL for 0 .. n_devices-1; an ordinal may repeat (several shares on
 * one GPU: a rehearsal of the multi-GPU path on a single-GPU machine).  stats (nullable): samples, rays and test
 * counts are summed over the devices, kernel_ms is the longest device's. */
int rtx_render_multi(rtx_scene_t* scene, const rtx_camera_t* camera, const rtx_render_opts_t* opts,
                     const int* devices, uint32_t n_devices, uint32_t split, uint8_t* rgb8_out, float* accum_out,
                     rtx_stats_t* stats);

/* Camera::render_pixel's sample loop (camera.rs:109-122) for the samples [sample_begin, sample_end) of every
 * pixel: ADDS their radiance sum into accum_dev (3*W*H f32, device memory, caller zeroes it).  This is the
 * unit that multi-GPU sample splitting shards; the buffers of all ranks are summed before rtx_resolve.
 * Asynchronous on `stream` unless stats is non-null (then it synchronises the stream to read the timers).
 * With RTX_TRAVERSAL_FAST the first call, and any call whose camera centre differs from the previous one's, first
 * waits for the device and re-orders the scene's fast traversal data for that viewpoint (nearer subtrees are
 * visited first; hits are unchanged).  Calls on one scene must therefore not overlap. */
int rtx_render_accumulate(rtx_scene_t* scene, const rtx_camera_t* camera, const rtx_render_opts_t* opts,
                          float* accum_dev, void* stream, rtx_stats_t* stats);

/* The same call for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc;
 * whether the call can be recorded into a HIP graph has not been tested: it records two timing events on the stream).  It has no counterpart in the reference.  Everything rtx_render_accumulate keeps
 * in library-owned device memory -- the per-batch partial images and the work counters -- lives in one fixed array
 * of RTX_STATIC_SCRATCH_BYTES inside the library's code object (one copy per GPU, held until the process ends), and
 * the call performs no hipMalloc / hipFree and no upload: it fails with RTX_ERR_UNSUPPORTED when
 * 12 * width * height bytes (one partial image) exceed the array, and when the scene's fast traversal data has not
 * been prepared for this camera position by an earlier rtx_render_accumulate / rtx_render with the same camera
 * centre (RTX_TRAVERSAL_STRICT needs no preparation).  Frames are cut into as many sample batches as fit the array,
 * so results equal rtx_render_accumulate's up to the order of fp32 additions.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams or threads) race.  With stats non-null the call
 * synchronises the stream. */
#define RTX_STATIC_SCRATCH_BYTES (64u << 20)
int rtx_render_accumulate_static(rtx_scene_t* scene, const rtx_camera_t* camera, const rtx_render_opts_t* opts,
                                 float* accum_dev, void* stream, rtx_stats_t* stats);

/* camera.rs:101, :119: rgb8 = round(clamp(srgb_oetf(accum * scale), 0, 1) * 255).  Device buffers. */
int rtx_resolve(const float* accum_dev, uint32_t width, uint32_t height, float scale, uint8_t* rgb8_dev,
                void* stream);

/* Inspection, host only (no device is touched; no counterpart in the reference): writes the traversal data that
 * rtx_scene_create builds for `desc` -- 8 words per 32-byte entry as the kernels read them, links as byte offsets,
 * terminator last -- and the tie-break rank per entry.  traversal = RTX_TRAVERSAL_*.  view (nullable, 3 floats): order
 * the fast data for that viewpoint as a render from there would; grow: enlarge every fast box by this much.
 * With entries_out == NULL only *n_entries is set.  Used by the CPU test-suite to walk the data without a GPU.
 * RTX_TRAVERSAL_FAST_FUSED dumps the packed form instead: 4 words per 16-byte slot, *n_entries and capacity count
 * slots, rank_out gets one word per slot (RTX_ERR_UNSUPPORTED for a scene with transforms or media). */
int rtx_traversal_dump(const rtx_scene_desc_t* desc, uint32_t traversal, const float* view, float grow,
                       uint32_t* entries_out, uint32_t* rank_out, size_t capacity, size_t* n_entries);

/* Inspection, host only, like rtx_traversal_dump: the WIDE form of the fast traversal data, which the render kernels walk
 * for scenes without Translate / RotateY / ConstantMedium that are too large for on-chip memory (no counterpart in the
 * reference; replaces Bvh::hit's recursion, mesh.rs:607-625, like the other forms).  nodes_out: 16 words per node --
 * four 16-byte slots (origin.xyz, child_base) (scale.xyz, prim_base) (lo_x, lo_y, lo_z, leaves) (hi_x, hi_y, hi_z,
 * counts = n_internal | n_children << 8): a node has up to four children; byte j of lo_a / hi_a bounds child j on axis a as
 * [origin_a + scale_a * lo, origin_a + scale_a * hi] (an unused child: lo 255, hi 0).  Children j < n_internal are the
 * nodes child_base + j; the others are leaves whose primitives lie one after the other from slot prim_base in child
 * order: byte j of `leaves` = count << 5 | index of the leaf's first primitive record among the node's.
 * prims_out: 4 words per slot, 5 slots per primitive, laid out as the packed form's entries (a sphere uses two of its
 * five).  rank_out: one word per slot.  levels: nodes on the longest path from the root.  With both output pointers
 * NULL only the counts are set.  RTX_ERR_UNSUPPORTED when the scene has no wide form. */
int rtx_wide_dump(const rtx_scene_desc_t* desc, float grow, uint32_t* nodes_out, size_t node_capacity, uint32_t* prims_out,
                  uint32_t* rank_out, size_t slot_capacity, size_t* n_nodes, size_t* n_slots, uint32_t* levels);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
