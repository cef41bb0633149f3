// librtx.so -- HIP kernels for gfx950 and the C ABI of include/rtx.h.
//
// Kernels
//   trace_kernel, trace_packed_kernel   closest hit of arbitrary rays (the Mesh::hit contract); one lane per ray.
//   render_kernel   Camera::render_pixel / ray_color as a persistent wavefront path tracer (see below).
//   reduce_kernel   sums the per-batch partial images into the caller's accumulation buffer, in batch order.
//   resolve_kernel  1/spp scale, sRGB transfer and 8-bit quantisation.
//   add_slots_kernel, merge_tiles_kernel   combine the shares of several devices (rtx_render_multi).
//
// render_kernel.  Work is cut into items = (8x8 pixel tile, batch of sample indices).  Each 64-lane wave claims
// items from a global counter and owns an item's pool of (pixel, sample) pairs: a lane that needs work takes the
// next pair from the pool (ray generation), walks the threaded BVH (extend; held in LDS when it fits), and when enough
// lanes of the wave have finished their walk the wave shades them together (material evaluation, scatter), after
// which finished paths are replaced from the pool in the same pass, so the wave's lanes stay compacted without any
// queue in memory.  Radiance is added into the item's tile accumulator in LDS and written once per item into a
// per-batch partial image: no global atomics.  Which samples a pixel gets and what each of them contributes does not
// depend on how items were scheduled; the order in which the samples of one batch are added does (fp32 rounding only).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "accel.hpp"
#include "rtx.h"
#include "rtx_device.hpp"

using namespace rtxd;
static_assert(PK_PRIM == rtx_accel::PK_PRIM && PK_PAD_FLAG == rtx_accel::PK_PAD_FLAG && PK_INDEX_MASK == rtx_accel::PK_INDEX_MASK, "one tag layout on both sides");
static_assert(MED_MAX_DEPTH == rtx_accel::MAX_MED_DEPTH && XF_MAX_DEPTH == rtx_accel::MAX_XF_DEPTH, "the walks hold what the host-side builder accepts");

// ================================================================================================ errors
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr uint32_t MAX_ITEM_SLOTS_HOST = 8;  // items in flight per wave, upper limit (render_kernel keeps one tile accumulator per slot in LDS)
// Tuning knobs, read from the environment once per process.
struct Knobs {
    uint32_t view_order, items_per_wave, max_batches, sched_k, w_prim, w_serv, sched_kp, item_slots, lds_ents_max, planar_lds_max, perlin_lds, waves_per_block,
        blocks_per_cu, debug, wide, w_mix, sched_k_wide, xcd_items, skip_root, drop_unculling, w_prim_xf;
};
static const Knobs& knobs() {
    static const Knobs k = [] {
        auto u = [](const char* name, uint32_t dflt) {
            const char* v = std::getenv(name);
            return v ? (uint32_t)std::strtoul(v, nullptr, 10) : dflt;
        };
        Knobs x;
        x.view_order = u("RTX_VIEW_ORDER", 1);
        x.items_per_wave = u("RTX_ITEMS_PER_WAVE", 0);
        x.max_batches = u("RTX_MAX_BATCHES", 64);
        x.sched_k = std::max(1u, u("RTX_SCHED_K", 6));
        x.w_prim = std::max(1u, u("RTX_W_PRIM", 8));
        // the bracket variants' primitive step holds the quad test and the bracket code (five box steps and more): it waits
        // for fuller lanes than a sphere test does (measured 3 to 12: cornell-smoke +10 % at 4 against 8, the others +1 %)
        x.w_prim_xf = std::max(1u, u("RTX_W_PRIM_XF", std::getenv("RTX_W_PRIM") ? x.w_prim : 4));
        x.w_serv = std::max(1u, u("RTX_W_SERV", 3));
        x.sched_kp = u("RTX_SCHED_KP", 32);
        x.item_slots = std::min(MAX_ITEM_SLOTS_HOST, std::max(1u, u("RTX_ITEM_SLOTS", 4)));
        // (an array in LDS shares the 160 KB with at least one block's accumulators, and has a per-slot shading table)
        x.lds_ents_max = std::min<uint32_t>(u("RTX_LDS_ENTS_MAX", 72 * 1024), 120 * 1024);
        x.perlin_lds = u("RTX_PERLIN_LDS", 1);
        x.planar_lds_max = u("RTX_PLANAR_LDS_MAX", 16 * 1024);
        x.waves_per_block = u("RTX_WAVES_PER_BLOCK", 0);
        x.blocks_per_cu = u("RTX_BLOCKS_PER_CU", 0);
        x.debug = u("RTX_DEBUG", 0);
        x.wide = u("RTX_WIDE", 1);
        x.w_mix = u("RTX_W_MIX", 3);
        x.sched_k_wide = std::max(1u, u("RTX_SCHED_K_WIDE", 3));
        x.xcd_items = u("RTX_XCD_ITEMS", 1);
        x.skip_root = u("RTX_SKIP_ROOT", 1);
        x.drop_unculling = u("RTX_DROP_UNCULLING", 1);
        return x;
    }();
    return k;
}

// ================================================================================================ scene
constexpr size_t COUNTER_BYTES = 256;
constexpr uint32_t XCD_COUNTER_AT = 48;  // dword index of the eight per-XCD item counters inside the counter block (behind the tallies)
constexpr size_t PK_SHADE_TABLE_MAX_BYTES = 160 * 1024;  // packed arrays up to this size (any that can live in LDS) get a per-slot shading table

static bool env_flag(const char* name, bool dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) != 0 : dflt;
}  // [0] item counter (u32); u64 words from byte 8: rays, box tests, primitive tests, then diagnostics

// The scene's device data on one GPU, with the render scratch that goes with it.
constexpr uint32_t MED_LANES_PER_CU = 2048;  // what a CU holds at most: 32 waves

struct DeviceCopy {
    int device = 0;
    int n_cus = 0;
    DevScene dev{};
    std::vector<std::pair<void*, size_t>> allocs;  // everything `dev` points into, in upload order
    // render scratch (grown on demand)
    float* partial = nullptr;
    size_t partial_bytes = 0;
    uint32_t* counters = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // the viewpoint this copy's fast array is currently ordered for
    float view[3] = {0, 0, 0};
    float view_grow = 0.0f;
    bool view_valid = false;
    float wide_grow = 0.0f;  // how far the boxes of this copy's wide nodes are enlarged
    bool peer_direct = true;  // replica on another device: it can reach the primary's device directly (GPU fabric)
    std::string peer_note;    // why not, when it cannot
    // rtx_render_multi only: this share's stream, sums and bytes; staging slots on the gathering device
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    float* accum = nullptr;
    uint8_t* rgb = nullptr;
    void* staging = nullptr;
    size_t accum_bytes = 0, rgb_bytes = 0, staging_bytes = 0;
};

struct rtx_scene {
    DeviceCopy primary;
    std::vector<DeviceCopy*> replicas;  // rtx_render_multi: share k > 0 (or share 0 on another device than the primary's)
    uint32_t n_ents[2] = {0, 0};
    uint32_t n_boxes[2] = {0, 0};
    float coord_range = 0.0f;    // largest |coordinate| the fast boxes were padded for
    uint32_t n_perlins = 0;
    uint32_t n_images = 0;
    bool has_tex = false;       // scene uses image or noise textures (uv evaluation, Perlin tables)
    bool has_brackets = false;  // scene uses Translate / RotateY / ConstantMedium
    bool media_extras = false;  // media inside media, or media bounded by one sphere: rendered and traced by the NEST variants
    bool fast_has_frames = false;  // ... and its fast array still holds transform-enter / leave entries (instances whose
                                   // primitives could not be placed in the enclosing frame's tree)
    // the fast array in entry-index form: re-ordered for a viewpoint when a render asks for it
    std::vector<uint4> mat_recs;  // 2 per material: the shading record a hit entry gets (see DevScene::shade)
    rtx_accel::Threaded fast_indexed;
    // packed form of the fast array (bracket-free scenes): what re-packing for a viewpoint needs on the host
    bool packed = false;
    bool pk_half = false;  // boxes of the packed array in half precision (arrays too large for LDS)
    uint32_t pk_slots = 0;
    uint32_t n_quads = 0;
    uint32_t n_planar = 0;  // quads + triangles (80-byte records)
    // wide form of the fast array (bracket-free scenes too large for LDS): what the render kernels walk instead of the packed one
    bool wide = false;
    uint32_t w_nodes = 0, w_slots = 0, w_levels = 0;
    std::vector<float4> planar_host;      // 5 per quad / triangle, as uploaded
    std::vector<uint32_t> planar_id_host;
};

template <class T>
static int upload(DeviceCopy* c, const std::vector<T>& host, const T** dev_out) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(host.size() * sizeof(T), 16);
    HIP_TRY(hipMalloc(&p, bytes));
    c->allocs.emplace_back(p, bytes);
    if (!host.empty()) HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev_out = reinterpret_cast<const T*>(p);
    return RTX_OK;
}

static bool tex_reads_uv(const rtx_scene_desc_t& d, uint32_t t, int depth = 0) {
    if (t >= d.n_textures || depth > 16) return false;
    const rtx_texture_t& x = d.textures[t];
    if (x.kind == RTX_TEX_IMAGE) return true;
    if (x.kind == RTX_TEX_CHECKER) return tex_reads_uv(d, x.a, depth + 1) || tex_reads_uv(d, x.b, depth + 1);
    return false;
}

extern "C" int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

extern "C" const char* rtx_last_error(void) { return g_err.c_str(); }

static int validate(const rtx_scene_desc_t* d) {
    if (!d) return fail(RTX_ERR_INVALID, "null scene description");
    if (d->n_spheres && !(d->sph_cx && d->sph_cy && d->sph_cz && d->sph_dx && d->sph_dy && d->sph_dz && d->sph_radius && d->sph_material && d->sph_prim_id))
        return fail(RTX_ERR_INVALID, "sphere arrays missing");
    if (d->n_quads) {
        for (int k = 0; k < 16; k++)
            if (!d->quad_f[k]) return fail(RTX_ERR_INVALID, "quad arrays missing");
        if (!d->quad_material || !d->quad_prim_id) return fail(RTX_ERR_INVALID, "quad arrays missing");
    }
    if (d->n_tris) {
        for (int k = 0; k < 9; k++)
            if (!d->tri_f[k]) return fail(RTX_ERR_INVALID, "triangle arrays missing");
        if (!d->tri_material || !d->tri_prim_id) return fail(RTX_ERR_INVALID, "triangle arrays missing");
    }
    if (d->n_spheres + d->n_quads + d->n_tris == 0) return fail(RTX_ERR_INVALID, "empty scene: no primitives");
    if (d->n_spheres >= (1u << 28) || d->n_quads >= (1u << 27) || d->n_tris >= (1u << 27)) return fail(RTX_ERR_INVALID, "too many primitives");
    if (d->n_materials == 0 || !d->materials) return fail(RTX_ERR_INVALID, "no materials");
    if ((d->n_bvh_nodes && !d->bvh_nodes) || (d->n_objects && !d->objects) || (d->n_refs && !d->refs) || (d->n_textures && !d->textures) ||
        (d->n_images && !d->images) || (d->n_perlins && !d->perlins))
        return fail(RTX_ERR_INVALID, "array pointer missing for a non-zero count");
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        if (d->sph_material[i] >= d->n_materials) return fail(RTX_ERR_INVALID, "sphere material out of range");
        if (!(d->sph_radius[i] >= 0.0f)) return fail(RTX_ERR_INVALID, "Radius cannot be less than 0.");
    }
    for (uint32_t i = 0; i < d->n_quads; i++)
        if (d->quad_material[i] >= d->n_materials) return fail(RTX_ERR_INVALID, "quad material out of range");
    for (uint32_t i = 0; i < d->n_tris; i++)
        if (d->tri_material[i] >= d->n_materials) return fail(RTX_ERR_INVALID, "triangle material out of range");
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rtx_material_t& m = d->materials[i];
        if (m.kind > RTX_MAT_ISOTROPIC) return fail(RTX_ERR_INVALID, "unknown material kind");
        if (m.kind != RTX_MAT_DIELECTRIC && m.texture >= d->n_textures) return fail(RTX_ERR_INVALID, "material texture out of range");
    }
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rtx_texture_t& t = d->textures[i];
        if (t.kind > RTX_TEX_NOISE) return fail(RTX_ERR_INVALID, "unknown texture kind");
        if (t.kind == RTX_TEX_CHECKER && (t.a >= d->n_textures || t.b >= d->n_textures)) return fail(RTX_ERR_INVALID, "checker child out of range");
        if (t.kind == RTX_TEX_IMAGE && t.a >= d->n_images) return fail(RTX_ERR_INVALID, "image index out of range");
        if (t.kind == RTX_TEX_NOISE && t.a >= d->n_perlins) return fail(RTX_ERR_INVALID, "perlin index out of range");
    }
    for (uint32_t i = 0; i < d->n_images; i++)
        if (!d->images[i].rgb || !d->images[i].width || !d->images[i].height) return fail(RTX_ERR_INVALID, "empty image");
    for (uint32_t i = 0; i < d->n_perlins; i++) {
        const rtx_perlin_t& p = d->perlins[i];
        if (!p.values || !p.perm_x || !p.perm_y || !p.perm_z) return fail(RTX_ERR_INVALID, "perlin tables missing");
        for (int k = 0; k < 256; k++)
            if ((p.perm_x[k] | p.perm_y[k] | p.perm_z[k]) > 255u) return fail(RTX_ERR_INVALID, "perlin permutation entry out of range");
    }
    return RTX_OK;
}

// Frees what a copy holds on its device (the current device is changed and not restored).
static void release_copy(DeviceCopy* c) {
    hipSetDevice(c->device);
    for (auto& a : c->allocs) hipFree(a.first);
    if (c->partial) hipFree(c->partial);
    if (c->counters) hipFree(c->counters);
    if (c->accum) hipFree(c->accum);
    if (c->rgb) hipFree(c->rgb);
    if (c->staging) hipFree(c->staging);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->done) hipEventDestroy(c->done);
    if (c->stream) hipStreamDestroy(c->stream);
    *c = DeviceCopy();
}

extern "C" void rtx_scene_destroy(rtx_scene_t* s) {
    if (!s) return;
    int prev = 0;
    hipGetDevice(&prev);
    for (DeviceCopy* r : s->replicas) {
        if (!r) continue;
        release_copy(r);
        delete r;
    }
    release_copy(&s->primary);
    hipSetDevice(prev);
    delete s;
}

// Planar records as the device keeps them, 5 x float4 per primitive -- (Q.xyz, d) (u.xyz, material bits) (v.xyz, 0)
// (n.xyz, 0) (w.xyz, 0) -- quads first, then triangles, and their prim ids in the same order.
static void build_planar(const rtx_scene_desc_t& desc, std::vector<float4>& planar, std::vector<uint32_t>& planar_id) {
    const rtx_scene_desc_t* d = &desc;
    auto bf = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    // planar records: quads as given; triangles through the same derivation Quad::new applies to (Q, u, v)
    // (mesh.rs:233-244) with Q = v0, u = v1 - v0, v = v2 - v0.
    size_t n_planar = (size_t)d->n_quads + d->n_tris;
    planar.assign(5 * n_planar, make_float4(0, 0, 0, 0));
    planar_id.assign(n_planar, 0u);
    for (uint32_t i = 0; i < d->n_quads; i++) {
        const float* const* q = d->quad_f;
        planar[5 * i + 0] = make_float4(q[0][i], q[1][i], q[2][i], q[15][i]);
        planar[5 * i + 1] = make_float4(q[3][i], q[4][i], q[5][i], bf(d->quad_material[i]));
        planar[5 * i + 2] = make_float4(q[6][i], q[7][i], q[8][i], 0.0f);
        planar[5 * i + 3] = make_float4(q[9][i], q[10][i], q[11][i], 0.0f);
        planar[5 * i + 4] = make_float4(q[12][i], q[13][i], q[14][i], 0.0f);
        planar_id[i] = d->quad_prim_id[i];
    }
    for (uint32_t i = 0; i < d->n_tris; i++) {
        const float* const* t = d->tri_f;
        float q[3] = {t[0][i], t[1][i], t[2][i]};
        float u[3] = {t[3][i] - q[0], t[4][i] - q[1], t[5][i] - q[2]};
        float v[3] = {t[6][i] - q[0], t[7][i] - q[1], t[8][i] - q[2]};
        float n[3] = {u[1] * v[2] - v[1] * u[2], u[2] * v[0] - v[2] * u[0], u[0] * v[1] - v[0] * u[1]};
        float nn = (n[0] * n[0]) + (n[1] * n[1]) + (n[2] * n[2]);
        float inv = 1.0f / std::sqrt(nn);
        float nrm[3] = {n[0] * inv, n[1] * inv, n[2] * inv};
        float dd = (nrm[0] * q[0]) + (nrm[1] * q[1]) + (nrm[2] * q[2]);
        size_t p = (size_t)d->n_quads + i;
        planar[5 * p + 0] = make_float4(q[0], q[1], q[2], dd);
        planar[5 * p + 1] = make_float4(u[0], u[1], u[2], bf(d->tri_material[i]));
        planar[5 * p + 2] = make_float4(v[0], v[1], v[2], 0.0f);
        planar[5 * p + 3] = make_float4(nrm[0], nrm[1], nrm[2], 0.0f);
        planar[5 * p + 4] = make_float4(n[0] / nn, n[1] / nn, n[2] / nn, 0.0f);
        planar_id[p] = d->tri_prim_id[i];
    }
}

// Builds the traversal arrays of a scene: the choices that depend on the form the fast array will be walked in.
static int build_arrays(const rtx_scene_desc_t& d, rtx_accel::Threaded& strict, rtx_accel::Threaded& fast, rtx_accel::Threaded& indexed, float& range,
                        std::string& err) {
    const Knobs& K = knobs();
    const size_t estimate = 64 * (size_t)d.n_spheres + 112 * ((size_t)d.n_quads + d.n_tris);  // a box and an entry per primitive
    bool brackets = false;  // (a scene with Translate / RotateY / ConstantMedium keeps the threaded form whatever its size)
    for (uint32_t i = 0; i < d.n_objects; i++) brackets = brackets || d.objects[i].kind != RTX_OBJ_LIST;
    const bool threaded = estimate <= K.lds_ents_max || brackets;  // (what is walked in the threaded form; else the wide form, most likely)
    const bool drop = K.drop_unculling != 0u;
    // In a threaded array a quad or triangle test costs about five box steps, and the builder prices it so (leaves of
    // several of them rarely pay); the wide form fetches the primitives of a leaf in one go and keeps the plain price.
    const uint32_t max_leaf = 4u;
    const float planar_cost = threaded ? 5.0f : 1.0f;
    int rc = rtx_accel::build(d, strict, fast, indexed, range, err, drop, max_leaf, planar_cost);
    return rc;
}

extern "C" int rtx_traversal_dump(const rtx_scene_desc_t* d, uint32_t traversal, const float* view, float grow, uint32_t* entries_out,
                                  uint32_t* rank_out, size_t capacity, size_t* n_entries) {
    if (!n_entries) return fail(RTX_ERR_INVALID, "null argument");
    if (traversal > RTX_TRAVERSAL_FAST_FUSED) return fail(RTX_ERR_INVALID, "unknown traversal mode");
    int rc = validate(d);
    if (rc != RTX_OK) return rc;
    rtx_accel::Threaded strict, fast, indexed;
    std::string err;
    float range = 0.0f;
    rc = build_arrays(*d, strict, fast, indexed, range, err);
    if (rc != RTX_OK) return fail(rc, err);
    if (traversal == RTX_TRAVERSAL_FAST_FUSED) {
        // the packed form: 4 words per 16-byte slot, rank per slot
        const float origin[3] = {0.0f, 0.0f, 0.0f};
        rtx_accel::Threaded ordered;
        rtx_accel::Reorder(indexed, ordered).run(view ? view : origin, grow, view != nullptr);
        std::vector<float4> planar;
        std::vector<uint32_t> planar_id;
        build_planar(*d, planar, planar_id);
        rtx_accel::Packed pk;
        const bool half = rtx_accel::packed_slots(ordered, false) * 16 > knobs().lds_ents_max || env_flag("RTX_PACKED_HALF", false);
        if (!rtx_accel::pack(ordered, reinterpret_cast<const float*>(planar.data()), planar_id.data(), d->n_quads, half, pk))
            return fail(RTX_ERR_UNSUPPORTED, "scenes with Translate / RotateY / ConstantMedium have no packed fast array");
        *n_entries = pk.n_slots();
        if (!entries_out) return RTX_OK;
        if (capacity < pk.n_slots()) return fail(RTX_ERR_INVALID, "capacity too small");
        std::memcpy(entries_out, pk.words.data(), pk.words.size() * 4);
        if (rank_out) std::memcpy(rank_out, pk.rank.data(), pk.rank.size() * 4);
        return RTX_OK;
    }
    if (traversal == RTX_TRAVERSAL_FAST && (view || grow > 0.0f)) {
        const float origin[3] = {0.0f, 0.0f, 0.0f};
        rtx_accel::Threaded ordered;
        rtx_accel::Reorder(indexed, ordered).run(view ? view : origin, grow, view != nullptr);
        rtx_accel::finalize(ordered);
        fast = ordered;
    }
    const rtx_accel::Threaded& t = traversal == RTX_TRAVERSAL_STRICT ? strict : fast;
    *n_entries = t.ents.size();
    if (!entries_out) return RTX_OK;
    if (capacity < t.ents.size()) return fail(RTX_ERR_INVALID, "capacity too small");
    std::memcpy(entries_out, t.ents.data(), t.ents.size() * sizeof(rtx_accel::Ent));
    if (rank_out) std::memcpy(rank_out, t.rank.data(), t.rank.size() * sizeof(uint32_t));
    return RTX_OK;
}

extern "C" int rtx_wide_dump(const rtx_scene_desc_t* d, float grow, uint32_t* nodes_out, size_t node_capacity, uint32_t* prims_out, uint32_t* rank_out,
                             size_t slot_capacity, size_t* n_nodes, size_t* n_slots, uint32_t* levels) {
    if (!n_nodes || !n_slots) return fail(RTX_ERR_INVALID, "null argument");
    int rc = validate(d);
    if (rc != RTX_OK) return rc;
    rtx_accel::Threaded strict, fast, indexed;
    std::string err;
    float range = 0.0f;
    rc = rtx_accel::build(*d, strict, fast, indexed, range, err);
    if (rc != RTX_OK) return fail(rc, err);
    std::vector<float4> planar;
    std::vector<uint32_t> planar_id;
    build_planar(*d, planar, planar_id);
    rtx_accel::Wide w;
    if (!rtx_accel::WideBuilder(indexed, reinterpret_cast<const float*>(planar.data()), planar_id.data(), d->n_quads, grow, w).run())
        return fail(RTX_ERR_UNSUPPORTED, "no wide array for this scene (Translate / RotateY / ConstantMedium, or a tree deeper than the walk's stack)");
    *n_nodes = w.n_nodes();
    *n_slots = w.n_slots();
    if (levels) *levels = w.levels;
    if (!nodes_out && !prims_out) return RTX_OK;
    if (node_capacity < w.n_nodes() || slot_capacity < w.n_slots()) return fail(RTX_ERR_INVALID, "capacity too small");
    if (nodes_out) std::memcpy(nodes_out, w.nodes.data(), w.nodes.size() * 4);
    if (prims_out) std::memcpy(prims_out, w.prims.data(), w.prims.size() * 4);
    if (rank_out) std::memcpy(rank_out, w.rank.data(), w.rank.size() * 4);
    return RTX_OK;
}

// Per-entry shading records of a threaded array: the record of the entry's material, zeros where nothing can be hit.
static std::vector<uint4> expand_shade(const rtx_accel::Threaded& t, const std::vector<uint4>& mat_recs) {
    std::vector<uint4> out(2 * t.ents.size(), make_uint4(0, 0, 0, 0));
    for (size_t i = 0; i < t.ents.size(); i++) {
        const uint32_t m = t.mat[i];
        if (m != rtx_accel::NONE && 2 * (size_t)m + 1 < mat_recs.size()) {
            out[2 * i] = mat_recs[2 * m];
            out[2 * i + 1] = mat_recs[2 * m + 1];
        }
    }
    return out;
}

// The same per slot of a packed array.
static std::vector<uint4> expand_shade_packed(const rtx_accel::Packed& pk, const std::vector<uint4>& mat_recs) {
    std::vector<uint4> out(2 * pk.n_slots(), make_uint4(0, 0, 0, 0));
    for (size_t i = 0; i < pk.n_slots(); i++) {
        const uint32_t m = pk.mat[i];
        if (m != rtx_accel::NONE && 2 * (size_t)m + 1 < mat_recs.size()) {
            out[2 * i] = mat_recs[2 * m];
            out[2 * i + 1] = mat_recs[2 * m + 1];
        }
    }
    return out;
}

static int scene_create_impl(const rtx_scene_desc_t* d, rtx_scene* s) {
    DeviceCopy* dc = &s->primary;
    rtx_accel::Threaded strict, fast;
    std::string err;
    int rc = build_arrays(*d, strict, fast, s->fast_indexed, s->coord_range, err);
    if (rc != RTX_OK) return fail(rc, err);
    if (strict.has_brackets) s->has_brackets = true;  // transforms / media need the bracket-aware kernel variant
    s->fast_has_frames = fast.has_frames;
    s->media_extras = strict.med_levels > 1 || fast.sphere_media;

    HIP_TRY(hipGetDevice(&dc->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dc->device));
    dc->n_cus = prop.multiProcessorCount;

    // shading record per material: kind, flags, parameter, and the colour when the texture is a plain solid one
    // (the dielectric attenuates with white, material.rs:175)
    auto fbits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    s->mat_recs.assign(2 * (size_t)d->n_materials, make_uint4(0, 0, 0, 0));
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rtx_material_t& m = d->materials[i];
        const bool glass = m.kind == RTX_MAT_DIELECTRIC;
        const bool solid = glass || d->textures[m.texture].kind == RTX_TEX_SOLID;
        const bool uv = !glass && tex_reads_uv(*d, m.texture);
        const float one[3] = {1.0f, 1.0f, 1.0f};
        const float* rgb = glass ? one : d->textures[m.texture].rgb;
        s->mat_recs[2 * i] = make_uint4(m.kind | (solid ? 256u : 0u) | (uv ? 512u : 0u), fbits(m.param), glass ? 0u : m.texture, 0u);
        s->mat_recs[2 * i + 1] = solid ? make_uint4(fbits(rgb[0]), fbits(rgb[1]), fbits(rgb[2]), 0u) : make_uint4(0, 0, 0, 0);
        // a checker of two plain colours on a material without a parameter: scale and both colours fit the record
        // (1024: word 1 = scale, words 2-4 = even colour, words 5-7 = odd colour)
        if (!glass && m.kind != RTX_MAT_METAL && d->textures[m.texture].kind == RTX_TEX_CHECKER) {
            const rtx_texture_t& c = d->textures[m.texture];
            if (d->textures[c.a].kind == RTX_TEX_SOLID && d->textures[c.b].kind == RTX_TEX_SOLID) {
                const float* e = d->textures[c.a].rgb;
                const float* o = d->textures[c.b].rgb;
                s->mat_recs[2 * i] = make_uint4(m.kind | 1024u, fbits(c.scale), fbits(e[0]), fbits(e[1]));
                s->mat_recs[2 * i + 1] = make_uint4(fbits(e[2]), fbits(o[0]), fbits(o[1]), fbits(o[2]));
            }
        }
    }

    const rtx_accel::Threaded* th[2] = {&fast, &strict};
    for (int m = 0; m < 2; m++) {
        const rtx_accel::Ent* ep = nullptr;
        if ((rc = upload(dc, th[m]->ents, &ep)) != RTX_OK) return rc;
        dc->dev.ents[m] = reinterpret_cast<const uint4*>(ep);
        if ((rc = upload(dc, th[m]->rank, &dc->dev.rank[m])) != RTX_OK) return rc;
        if ((rc = upload(dc, expand_shade(*th[m], s->mat_recs), &dc->dev.shade[m])) != RTX_OK) return rc;
        dc->dev.n_ents[m] = s->n_ents[m] = (uint32_t)th[m]->ents.size();  // terminator included
        s->n_boxes[m] = th[m]->n_boxes;
    }
    auto fb = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto bf = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };

    std::vector<float4> sph(2 * (size_t)d->n_spheres);
    std::vector<uint32_t> sph_id(d->n_spheres);
    for (uint32_t i = 0; i < d->n_spheres; i++) {
        sph[2 * i] = make_float4(d->sph_cx[i], d->sph_cy[i], d->sph_cz[i], d->sph_radius[i]);
        sph[2 * i + 1] = make_float4(d->sph_dx[i], d->sph_dy[i], d->sph_dz[i], bf(d->sph_material[i]));
        sph_id[i] = d->sph_prim_id[i];
    }
    if ((rc = upload(dc, sph, &dc->dev.sph)) != RTX_OK) return rc;
    if ((rc = upload(dc, sph_id, &dc->dev.sph_id)) != RTX_OK) return rc;

    std::vector<float4> planar;
    std::vector<uint32_t> planar_id;
    build_planar(*d, planar, planar_id);
    if ((rc = upload(dc, planar, &dc->dev.planar)) != RTX_OK) return rc;
    if ((rc = upload(dc, planar_id, &dc->dev.planar_id)) != RTX_OK) return rc;
    dc->dev.n_quads = d->n_quads;
    s->n_quads = d->n_quads;
    s->n_planar = d->n_quads + d->n_tris;
    if ((rc = upload(dc, s->mat_recs, &dc->dev.mat_recs)) != RTX_OK) return rc;
    {
        // the packed form of the fast array, which the render kernels walk when the scene has no brackets
        rtx_accel::Packed pk;
        s->pk_half = rtx_accel::packed_slots(s->fast_indexed, false) * 16 > knobs().lds_ents_max || env_flag("RTX_PACKED_HALF", false);
        s->packed = env_flag("RTX_PACKED", true) &&
                    rtx_accel::pack(s->fast_indexed, reinterpret_cast<const float*>(planar.data()), planar_id.data(), d->n_quads, s->pk_half, pk);
        if (s->packed) {
            const uint4* words = nullptr;
            std::vector<uint4> w4(pk.n_slots());
            std::memcpy(w4.data(), pk.words.data(), pk.words.size() * 4);
            if ((rc = upload(dc, w4, &words)) != RTX_OK) return rc;
            dc->dev.pk = words;
            dc->dev.pk_slots = s->pk_slots = (uint32_t)pk.n_slots();
            dc->dev.pk_half = s->pk_half ? 1u : 0u;
            if ((rc = upload(dc, pk.rank, &dc->dev.pk_rank)) != RTX_OK) return rc;
            if (pk.n_slots() * 16 <= PK_SHADE_TABLE_MAX_BYTES) {
                if ((rc = upload(dc, expand_shade_packed(pk, s->mat_recs), &dc->dev.pk_shade)) != RTX_OK) return rc;
            }
            s->planar_host = planar;
            s->planar_id_host = planar_id;
        }
        if (s->packed && s->pk_half && knobs().wide != 0u) {
            rtx_accel::Wide w;
            if (rtx_accel::WideBuilder(s->fast_indexed, reinterpret_cast<const float*>(planar.data()), planar_id.data(), d->n_quads, 0.0f, w).run()) {
                std::vector<uint4> n4(w.nodes.size() / 4), p4(w.prims.size() / 4);
                std::memcpy(n4.data(), w.nodes.data(), w.nodes.size() * 4);
                std::memcpy(p4.data(), w.prims.data(), w.prims.size() * 4);
                if ((rc = upload(dc, n4, &dc->dev.wnodes)) != RTX_OK) return rc;
                if ((rc = upload(dc, p4, &dc->dev.wprims)) != RTX_OK) return rc;
                if ((rc = upload(dc, w.rank, &dc->dev.wrank)) != RTX_OK) return rc;
                dc->dev.w_levels = s->w_levels = w.levels;
                dc->dev.w_first = w.n_first;
                s->w_nodes = (uint32_t)w.n_nodes();
                s->w_slots = (uint32_t)w.n_slots();
                s->wide = true;
            }
        }
    }

    std::vector<uint4> mats(d->n_materials);
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rtx_material_t& m = d->materials[i];
        uint32_t flags = (m.kind != RTX_MAT_DIELECTRIC && tex_reads_uv(*d, m.texture)) ? 1u : 0u;
        mats[i] = make_uint4(m.kind, m.texture, fb(m.param), flags);
    }
    if ((rc = upload(dc, mats, &dc->dev.mats)) != RTX_OK) return rc;

    std::vector<uint4> texs(2 * (size_t)d->n_textures);
    for (uint32_t i = 0; i < d->n_textures; i++) {
        const rtx_texture_t& t = d->textures[i];
        texs[2 * i] = make_uint4(t.kind, fb(t.scale), t.a, t.b);
        texs[2 * i + 1] = make_uint4(fb(t.rgb[0]), fb(t.rgb[1]), fb(t.rgb[2]), 0);
    }
    if ((rc = upload(dc, texs, &dc->dev.texs)) != RTX_OK) return rc;
    for (uint32_t i = 0; i < d->n_textures; i++)
        if (d->textures[i].kind == RTX_TEX_IMAGE || d->textures[i].kind == RTX_TEX_NOISE) s->has_tex = true;

    std::vector<DevImage> images(d->n_images);
    for (uint32_t i = 0; i < d->n_images; i++) {
        const rtx_image_t& im = d->images[i];
        std::vector<uint8_t> px(im.rgb, im.rgb + (size_t)im.width * im.height * 3);
        const uint8_t* dp = nullptr;
        if ((rc = upload(dc, px, &dp)) != RTX_OK) return rc;
        images[i] = DevImage{dp, im.width, im.height};
    }
    if ((rc = upload(dc, images, &dc->dev.images)) != RTX_OK) return rc;

    std::vector<float> perlin(1536 * (size_t)d->n_perlins);
    for (uint32_t i = 0; i < d->n_perlins; i++) {
        const rtx_perlin_t& p = d->perlins[i];
        float* dst = &perlin[1536 * (size_t)i];
        std::memcpy(dst, p.values, 768 * sizeof(float));
        std::memcpy(dst + 768, p.perm_x, 256 * 4);
        std::memcpy(dst + 1024, p.perm_y, 256 * 4);
        std::memcpy(dst + 1280, p.perm_z, 256 * 4);
    }
    if ((rc = upload(dc, perlin, &dc->dev.perlin)) != RTX_OK) return rc;
    s->n_perlins = d->n_perlins;

    s->n_images = d->n_images;
    if (strict.med_levels > 1) {
        // media inside a medium's boundary: room for every lane a launch can hold to park the state of the enclosing
        // media (MED_MAX_DEPTH - 1 records of five words each)
        const std::vector<uint32_t> room((size_t)(MED_MAX_DEPTH - 1) * 5 * MED_LANES_PER_CU * (size_t)dc->n_cus);
        const uint32_t* p = nullptr;
        if ((rc = upload(dc, room, &p)) != RTX_OK) return rc;
        dc->dev.med_stack = const_cast<uint32_t*>(p);
        dc->dev.med_lanes = MED_LANES_PER_CU * (uint32_t)dc->n_cus;
    }
    HIP_TRY(hipMalloc((void**)&dc->counters, COUNTER_BYTES));
    HIP_TRY(hipEventCreate(&dc->ev0));
    HIP_TRY(hipEventCreate(&dc->ev1));
    return RTX_OK;
}

extern "C" int rtx_scene_create(const rtx_scene_desc_t* desc, rtx_scene_t** out) {
    if (!out) return fail(RTX_ERR_INVALID, "null output pointer");
    *out = nullptr;
    int rc = validate(desc);
    if (rc != RTX_OK) return rc;
    if (rtx_device_count() <= 0) return fail(RTX_ERR_NO_DEVICE, "no gfx950 device visible (there is no CPU fallback)");
    rtx_scene* s = new rtx_scene();
    rc = scene_create_impl(desc, s);
    if (rc != RTX_OK) {
        std::string keep = g_err;
        rtx_scene_destroy(s);
        g_err = keep;
        return rc;
    }
    *out = s;
    return RTX_OK;
}

static uint32_t seed_word(uint64_t seed, int which) {  // the two 32-bit words the kernels key their streams on (rng_init)
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    auto h = [](uint32_t v) {
        uint32_t s = v * 747796405u + 2891336453u;
        uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        return (w >> 22u) ^ w;
    };
    return which == 0 ? h(lo ^ h(hi)) : h((hi + 0x85EBCA6Bu) ^ h(lo + 0x9E3779B9u));
}

// ================================================================================================ trace
// QUICK: primitive tests and hit record as the render kernels' fast walk evaluates them (RTX_TRAVERSAL_RENDER).
// NEST: the scene holds media inside other media's boundaries.
template <bool STRICT, int SLAB, bool QUICK = false, bool NEST = false>
__global__ __launch_bounds__(256) void trace_kernel(DevScene S, const rtx_ray_t* __restrict__ rays, size_t n, float tmin, float tmax, uint32_t seed0,
                                                    uint32_t seed1, rtx_hit_t* __restrict__ hits, uint32_t start_off) {
    constexpr int M = STRICT ? RTX_TRAVERSAL_STRICT : RTX_TRAVERSAL_FAST;
    const uint4* __restrict__ ents = S.ents[M];
    const uint32_t* __restrict__ rank = S.rank[M];
    const EntReader reader{ents};
    // One lane per ray -- or, when the scene nests media and the launch is therefore no larger than the scene's parking
    // space for their state, a lane takes every (lanes of the launch)-th ray.
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    auto med_stack = [&]() { return MedStack{S.med_stack, S.med_lanes, g}; };
    for (size_t i = g; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Ray wray;
        wray.ox = rays[i].o[0]; wray.oy = rays[i].o[1]; wray.oz = rays[i].o[2];
        wray.dx = rays[i].d[0]; wray.dy = rays[i].d[1]; wray.dz = rays[i].d[2];
        wray.time = rays[i].time;
        Ray r = wray;
        RayConst rc = ray_const<SLAB>(r);
        Rng rng = rng_init(seed0, seed1, (uint32_t)i, 0u);  // only a ConstantMedium draws from it (mesh.rs:659)
        WalkX x;
        walkx_init(x, tmin);
        Best best;
        best.t = tmax;
        best.ent = NO_HIT;
        uint32_t off = start_off;  // (0, or behind the root box where a render of the scene starts its walks)
        for (;;) {
            uint4 ea, eb;
            reader.load(off, ea, eb);
            if (eb.w == ENT_END) break;
            if (eb.w != ENT_BOX && is_bracket_kind(eb.w & 15u)) {
                bracket_step<STRICT, SLAB, true, NEST, QUICK>(reader, ea, eb, off, r, rc, wray, best, x, med_stack, tmin, rng, rank);
            } else {
                const uint32_t at = off;
                off = walk_step<STRICT, SLAB, QUICK>(S, reader, ea, eb, off, r, rc, x.tmin, best, rank);
                if (best.ent == at) x.best_xf = x.cur_xf;
            }
        }
        rtx_hit_t out;
        if (best.ent == NO_HIT) {
            out.prim_id = RTX_MISS;
            out.t = 0.0f;
            out.n[0] = out.n[1] = out.n[2] = 0.0f;
            out.uv[0] = out.uv[1] = 0.0f;
            out.front_face = 0;
            out.p[0] = out.p[1] = out.p[2] = 0.0f;
        } else {
            HitRec h = make_hit_x<true, !QUICK>(S, reader, wray, best, x.best_xf);
            out.prim_id = h.prim_id;
            out.t = best.t;
            out.n[0] = h.nx; out.n[1] = h.ny; out.n[2] = h.nz;
            out.uv[0] = h.u; out.uv[1] = h.v;
            out.front_face = 1;  // Hit::new stores `front_face: true` unconditionally (mesh.rs:47)
            out.p[0] = h.px; out.p[1] = h.py; out.p[2] = h.pz;
        }
        hits[i] = out;
    }
}

// The same contract over the packed form of the fast array, with the box test the render kernels use.
// RENDER = false: exact roots and hit record on top of the fused box test (RTX_TRAVERSAL_FAST_FUSED).
// RENDER = true : every step through the functions render_kernel<..., PK> itself calls -- the scaled, clamped box test on
//                 float boxes (SCALED) or the half-precision box test, hardware square root / reciprocal in the primitive
//                 tests, reciprocal-multiplied normals (RTX_TRAVERSAL_RENDER).
template <bool RENDER, bool SCALED>
__global__ __launch_bounds__(256) void trace_packed_kernel(DevScene S, const rtx_ray_t* __restrict__ rays, size_t n, float tmin, float tmax,
                                                            rtx_hit_t* __restrict__ hits, uint32_t start_off) {
    static_assert(RENDER || !SCALED, "only the render walk runs in scaled ray parameters");
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SlotReader slots{S.pk};
    const uint32_t* __restrict__ rank = S.pk_rank;
    Ray r;
    r.ox = rays[i].o[0]; r.oy = rays[i].o[1]; r.oz = rays[i].o[2];
    r.dx = rays[i].d[0]; r.dy = rays[i].d[1]; r.dz = rays[i].d[2];
    r.time = rays[i].time;
    RayConst rc = ray_const<SLAB_FMA>(r);
    if (SCALED) scale_ray_const(rc);
    Best best;
    best.t = SCALED ? tmax * PK_T_SCALE : tmax;
    best.ent = NO_HIT;
    uint32_t off = start_off;
    for (;;) {
        const uint4 e = slots.at(off);
        if (e.w == ENT_END) break;
        if (e.w < PK_PRIM) {
            bool hit;
            if (RENDER)
                hit = pk_render_box<SCALED>(e, SCALED ? slots.at(off + PK_SLOT) : e, rc, tmin, best.t);
            else
                hit = S.pk_half ? pk_slab_hit(e, rc, tmin, best.t) : pk_slab_hit(e, slots.at(off + PK_SLOT), rc, tmin, best.t);
            off = hit ? off + (S.pk_half ? PK_SLOT : 2u * PK_SLOT) : (S.pk_half ? e.w * PK_SLOT : e.w);  // float boxes carry byte links
        } else {
            const uint32_t at = off;
            const auto slot_at = [&](int k) { return slots.at(at + (uint32_t)k * PK_SLOT); };
            if (RENDER)
                off += pk_render_prim<SCALED>(e, slots.at(off + PK_SLOT), slot_at, off, r, rc, tmin, best, rank);
            else
                off += pk_prim_test<false>(e, slots.at(off + PK_SLOT), slot_at, off, r, rc, tmin, best, rank);
        }
    }
    rtx_hit_t out;
    std::memset(&out, 0, sizeof(out));
    out.prim_id = RTX_MISS;
    if (best.ent != NO_HIT) {
        const uint32_t at = best.ent;
        const auto slot_at = [&](int k) { return slots.at(at + (uint32_t)k * PK_SLOT); };
        const HitRec h = RENDER ? pk_render_hit<SCALED, true, true>(S, slot_at, r, best.t) : pk_make_hit<true, true, true>(S, slot_at, r, best.t);
        if (SCALED) best.t *= PK_T_UNSCALE;
        out.prim_id = h.prim_id;
        out.t = best.t;
        out.n[0] = h.nx; out.n[1] = h.ny; out.n[2] = h.nz;
        out.uv[0] = h.u; out.uv[1] = h.v;
        out.front_face = 1;  // Hit::new stores `front_face: true` unconditionally (mesh.rs:47)
        out.p[0] = h.px; out.p[1] = h.py; out.p[2] = h.pz;
    }
    hits[i] = out;
}

// RTX_TRAVERSAL_RENDER for arrays a render keeps in LDS (float boxes): the block copies the packed array to LDS address 0
// as render_kernel<false, true, false, *, true> does, and every wave walks its 64 rays with that kernel's own pieces --
// runs of box steps through pk_box_run_lds (here: until no lane of the wave stands at a box), primitive tests and hit
// record through pk_render_prim / pk_render_hit -- reading entries from LDS by byte offset.
extern __shared__ uint4 trace_smem[];
__global__ __launch_bounds__(256) void trace_packed_lds_kernel(DevScene S, const rtx_ray_t* __restrict__ rays, size_t n, float tmin, float tmax,
                                                                rtx_hit_t* __restrict__ hits, uint32_t start_off) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    for (uint32_t k = threadIdx.x; k < S.pk_slots; k += blockDim.x) trace_smem[k] = S.pk[k];
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)trace_smem != 0u) __builtin_trap();  // byte offsets are LDS addresses
    __syncthreads();
    const auto slot = [](uint32_t o) {
        const u32x4 v = *(lds_u4*)(uintptr_t)o;
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint32_t* __restrict__ rank = S.pk_rank;
    Ray r{};
    if (live) {
        r.ox = rays[i].o[0]; r.oy = rays[i].o[1]; r.oz = rays[i].o[2];
        r.dx = rays[i].d[0]; r.dy = rays[i].d[1]; r.dz = rays[i].d[2];
        r.time = rays[i].time;
    }
    RayConst rc = ray_const<SLAB_FMA>(r);
    scale_ray_const(rc);
    Best best;
    best.t = tmax * PK_T_SCALE;
    best.ent = NO_HIT;
    uint32_t off = start_off, st = live ? slot(start_off).w : ENT_END;
    for (;;) {
        const bool at_prim = st >= PK_PRIM && st < ENT_END;
        if (__ballot(st < ENT_END) == 0ull) break;
        const unsigned long long boxes = __ballot(st < PK_PRIM);
        uint32_t steps = 0;
        if (boxes != 0ull) pk_box_run_lds(boxes, off, st, steps, rc, best.t, 1u);
        if (at_prim) {
            const uint32_t at = off;
            off += pk_render_prim<true>(slot(at), slot(at + PK_SLOT), [&](int k) { return slot(at + (uint32_t)k * PK_SLOT); }, at, r, rc, tmin, best, rank);
            st = slot(off).w;
        }
    }
    if (!live) return;
    rtx_hit_t out;
    std::memset(&out, 0, sizeof(out));
    out.prim_id = RTX_MISS;
    if (best.ent != NO_HIT) {
        const uint32_t at = best.ent;
        const HitRec h = pk_render_hit<true, true, true>(S, [&](int k) { return slot(at + (uint32_t)k * PK_SLOT); }, r, best.t);
        out.prim_id = h.prim_id;
        out.t = best.t * PK_T_UNSCALE;
        out.n[0] = h.nx; out.n[1] = h.ny; out.n[2] = h.nz;
        out.uv[0] = h.u; out.uv[1] = h.v;
        out.front_face = 1;  // Hit::new stores `front_face: true` unconditionally (mesh.rs:47)
        out.p[0] = h.px; out.p[1] = h.py; out.p[2] = h.pz;
    }
    hits[i] = out;
}

// The same contract over the wide form of the fast array: the node step and the primitive tests of
// render_kernel<..., PK, WIDE>, with the stack of pending groups in private memory.
// RENDER = false: exact roots and hit record (RTX_TRAVERSAL_FAST_FUSED); true: the render kernels' (RTX_TRAVERSAL_RENDER).
template <bool RENDER>
__global__ __launch_bounds__(256) void trace_wide_kernel(DevScene S, const rtx_ray_t* __restrict__ rays, size_t n, float tmin, float tmax,
                                                          rtx_hit_t* __restrict__ hits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SlotReader prims{S.wprims};
    const uint32_t* __restrict__ rank = S.wrank;
    Ray r;
    r.ox = rays[i].o[0]; r.oy = rays[i].o[1]; r.oz = rays[i].o[2];
    r.dx = rays[i].d[0]; r.dy = rays[i].d[1]; r.dz = rays[i].d[2];
    r.time = rays[i].time;
    const RayConst rc = ray_const<SLAB_FMA>(r);
    Best best;
    best.t = tmax;
    best.ent = NO_HIT;
    uint32_t stack[rtx_accel::WIDE_MAX_LEVELS];
    uint32_t sp = 0, group = 0, node = 0;
    for (uint32_t k = 0; k < S.w_first; k++) {  // the primitives that stand outside the tree: every walk tests them first
        const uint32_t o = k * 5u * PK_SLOT;
        const auto slot_at = [&](int q) { return prims.at(o + (uint32_t)q * PK_SLOT); };
        if (RENDER)
            pk_render_prim<false>(slot_at(0), slot_at(1), slot_at, o, r, rc, tmin, best, rank);
        else
            pk_prim_test<false>(slot_at(0), slot_at(1), slot_at, o, r, rc, tmin, best, rank);
    }
    for (;;) {
        const uint4* __restrict__ nd = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(S.wnodes) + (size_t)node * WIDE_NODE_BYTES);
        uint32_t next, g, tbase, tmask, n_children;
        wide_node_step(nd[0], nd[1], nd[2], nd[3], rc, tmin, best.t, next, g, tbase, tmask, n_children);
        if (g != 0u) {
            if (group != 0u && sp < rtx_accel::WIDE_MAX_LEVELS) stack[sp++] = group;
            group = g;
        }
        while (tmask != 0u) {
            const uint32_t k = (uint32_t)__builtin_ctz(tmask);
            tmask &= tmask - 1u;
            const uint32_t o = (tbase + k * 5u) * PK_SLOT;
            const auto slot_at = [&](int q) { return prims.at(o + (uint32_t)q * PK_SLOT); };
            if (RENDER)
                pk_render_prim<false>(slot_at(0), slot_at(1), slot_at, o, r, rc, tmin, best, rank);
            else
                pk_prim_test<false>(slot_at(0), slot_at(1), slot_at, o, r, rc, tmin, best, rank);
        }
        if (next != WIDE_NONE) {
            node = next;
            continue;
        }
        if (group == 0u && sp != 0u) group = stack[--sp];
        if (group == 0u) break;
        node = wide_group_pop(group);
    }
    rtx_hit_t out;
    std::memset(&out, 0, sizeof(out));
    out.prim_id = RTX_MISS;
    if (best.ent != NO_HIT) {
        const uint32_t at = best.ent;
        const auto slot_at = [&](int k) { return prims.at(at + (uint32_t)k * PK_SLOT); };
        const HitRec h = RENDER ? pk_render_hit<false, true, true>(S, slot_at, r, best.t) : pk_make_hit<true, true, true>(S, slot_at, r, best.t);
        out.prim_id = h.prim_id;
        out.t = best.t;
        out.n[0] = h.nx; out.n[1] = h.ny; out.n[2] = h.nz;
        out.uv[0] = h.u; out.uv[1] = h.v;
        out.front_face = 1;  // Hit::new stores `front_face: true` unconditionally (mesh.rs:47)
        out.p[0] = h.px; out.p[1] = h.py; out.p[2] = h.pz;
    }
    hits[i] = out;
}

// Every ray of a render starts at the camera or on a surface.  When the camera stands inside the fast array's root box no
// ray can miss that box (the surfaces are inside it by construction), so the walks start at the entry behind it: the
// root's first child, whose miss link leads to the second and on to the end as if the root had been entered.  Only cost
// changes, never a hit: a box test can only cull.  Returns the byte offset a fast walk of this scene starts at.
static uint32_t walk_start_off(const rtx_scene* scene, const float center[3], bool packed_form) {
    if (knobs().skip_root == 0u || scene->fast_indexed.ents.size() < 2 || scene->fast_indexed.ents[0].w[7] != rtx_accel::ENT_BOX) return 0u;
    const rtx_accel::Ent& root = scene->fast_indexed.ents[0];
    for (int a = 0; a < 3; a++) {
        float lo, hi;
        std::memcpy(&lo, &root.w[2 * a], 4);
        std::memcpy(&hi, &root.w[2 * a + 1], 4);
        if (!(center[a] > lo && center[a] < hi)) return 0u;
    }
    return packed_form ? (scene->pk_half ? 16u : 32u) : 32u;
}

static int kernel_allow_lds(const void* kern, int device, size_t lds);

extern "C" int rtx_trace_device(const rtx_scene_t* scene, const rtx_ray_t* rays_dev, size_t n, const rtx_trace_opts_t* opts, rtx_hit_t* hits_dev,
                                void* stream) {
    if (!scene || !opts || (n && (!rays_dev || !hits_dev))) return fail(RTX_ERR_INVALID, "null argument");
    if (opts->traversal > RTX_TRAVERSAL_RENDER) return fail(RTX_ERR_INVALID, "unknown traversal mode");
    if (n == 0) return RTX_OK;
    if (n > ((size_t)1 << 39)) return fail(RTX_ERR_INVALID, "too many rays for one call");
    hipStream_t st = (hipStream_t)stream;
    dim3 block(256), grid((unsigned)((n + 255) / 256));
    const DevScene& S = scene->primary.dev;
    // (a scene that nests media parks walk state per lane: trace_kernel then strides over the rays with the lanes it has room for)
    dim3 xgrid(S.med_stack ? std::min<unsigned>(grid.x, std::max(1u, S.med_lanes / 256u)) : grid.x);
    const bool nest = scene->media_extras;
    // RTX_TRAVERSAL_RENDER starts its walks where the last render of the scene started them (behind the root box when that
    // render's camera stood inside it)
    const uint32_t start = opts->traversal == RTX_TRAVERSAL_RENDER && scene->primary.view_valid ? walk_start_off(scene, scene->primary.view, scene->packed) : 0u;
#define RTX_TRACE(STRICT_, SLAB_, QUICK_, START_)                                                                                                                    \
    hipLaunchKernelGGL((nest ? trace_kernel<STRICT_, SLAB_, QUICK_, true> : trace_kernel<STRICT_, SLAB_, QUICK_, false>), xgrid, block, 0, st, S, rays_dev, n, \
                       opts->tmin, opts->tmax, seed_word(opts->seed, 0), seed_word(opts->seed, 1), hits_dev, START_)
    if (opts->traversal == RTX_TRAVERSAL_STRICT)
        RTX_TRACE(true, SLAB_STRICT, false, 0u);
    else if (opts->traversal == RTX_TRAVERSAL_FAST)
        RTX_TRACE(false, SLAB_ROBUST, false, 0u);
    else if (opts->traversal == RTX_TRAVERSAL_FAST_FUSED && scene->wide)  // what the render kernels walk for this scene
        hipLaunchKernelGGL((trace_wide_kernel<false>), grid, block, 0, st, S, rays_dev, n, opts->tmin, opts->tmax, hits_dev);
    else if (opts->traversal == RTX_TRAVERSAL_FAST_FUSED && scene->packed)
        hipLaunchKernelGGL((trace_packed_kernel<false, false>), grid, block, 0, st, S, rays_dev, n, opts->tmin, opts->tmax, hits_dev, 0u);
    else if (opts->traversal == RTX_TRAVERSAL_FAST_FUSED)
        RTX_TRACE(false, SLAB_FMA, false, 0u);
    // RTX_TRAVERSAL_RENDER: the walk of the render kernel variant a fast render of this scene launches
    else if (scene->wide)  // render_kernel<false, false, false, *, true, true>: four-child nodes
        hipLaunchKernelGGL((trace_wide_kernel<true>), grid, block, 0, st, S, rays_dev, n, opts->tmin, opts->tmax, hits_dev);
    else if (scene->packed && !scene->pk_half) {  // render_kernel<false, true, false, *, true>: float boxes in LDS, scaled ray parameters
        const size_t lds = (size_t)S.pk_slots * 16;
        int rc = kernel_allow_lds((const void*)trace_packed_lds_kernel, scene->primary.device, lds);
        if (rc != RTX_OK) return rc;
        hipLaunchKernelGGL(trace_packed_lds_kernel, grid, block, lds, st, S, rays_dev, n, opts->tmin, opts->tmax, hits_dev, start);
    }
    else if (scene->packed)  // render_kernel<false, false, false, *, true>: half-precision boxes
        hipLaunchKernelGGL((trace_packed_kernel<true, false>), grid, block, 0, st, S, rays_dev, n, opts->tmin, opts->tmax, hits_dev, start);
    else  // render_kernel<false, *, *, *, false>: 32-byte entries (scenes with Translate / RotateY / ConstantMedium)
        RTX_TRACE(false, SLAB_FMA, true, start);
#undef RTX_TRACE
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

extern "C" int rtx_trace(const rtx_scene_t* scene, const rtx_ray_t* rays, size_t n, const rtx_trace_opts_t* opts, rtx_hit_t* hits_out) {
    if (!scene || !opts || (n && (!rays || !hits_out))) return fail(RTX_ERR_INVALID, "null argument");
    if (n == 0) return RTX_OK;
    HIP_TRY(hipSetDevice(scene->primary.device));
    rtx_ray_t* dr = nullptr;
    rtx_hit_t* dh = nullptr;
    HIP_TRY(hipMalloc((void**)&dr, n * sizeof(rtx_ray_t)));
    hipError_t e = hipMalloc((void**)&dh, n * sizeof(rtx_hit_t));
    if (e != hipSuccess) {
        hipFree(dr);
        return fail(RTX_ERR_DEVICE, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    int rc = RTX_OK;
    e = hipMemcpy(dr, rays, n * sizeof(rtx_ray_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = rtx_trace_device(scene, dr, n, opts, dh, nullptr);
        if (rc == RTX_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(hits_out, dh, n * sizeof(rtx_hit_t), hipMemcpyDeviceToHost);
        }
    }
    hipFree(dr);
    hipFree(dh);
    if (rc != RTX_OK) return rc;
    if (e != hipSuccess) return fail(RTX_ERR_DEVICE, std::string("trace: ") + hipGetErrorString(e));
    return RTX_OK;
}

// ================================================================================================ render
struct RenderParams {
    DevScene S;
    rtx_camera_t cam;
    uint32_t seed0, seed1;  // the two stream keys derived from the 64-bit seed (seed_word)
    uint32_t sample_begin, sample_end;
    uint32_t batch;       // samples per item
    uint32_t n_batches;
    uint32_t tiles_x, tiles_y;
    uint32_t tile_rank, tile_world, n_owned;  // this call renders tiles tile_rank, tile_rank + tile_world, ... (n_owned of them)
    uint32_t n_items;
    uint32_t rr_start;
    uint32_t sched_k;     // box stage keeps running while n_box * sched_k >= lanes queued elsewhere
    uint32_t w_prim, w_serv;  // stage weights (eighths) of the primitive and service queues against the box queue
    uint32_t w_mix;       // wide walk: see the walk stage
    uint32_t box_run_min; // ceil(64 / (sched_k + 1)): the lanes a box run of the packed LDS walk needs to go on when all 64 have work
    uint32_t sched_kp;    // primitive stage keeps running while n_prim * sched_kp >= 8 * lanes queued elsewhere (0 = one step)
    uint32_t perlin_in_lds;  // 1: the launch reserved 6 KB of LDS per block for Perlin table set 0
    uint32_t planar_in_lds;  // quad / triangle records (80 bytes each) the launch reserved LDS for, behind the Perlin tables: all of them or 0
    uint32_t n_slots;     // items in flight per wave (tile accumulators per wave): 1 = no overlap of items
    uint32_t xcd_parts;   // scenes in HBM: 0 = one item counter, 1 = a counter per band of tile columns, 2 = per range of owned tiles
    uint32_t wide_stack_u4;  // WIDE variants: where the block's walk stacks start in LDS, in 16-byte units (behind accumulators and tables)
    uint32_t start_off;   // byte offset of the entry a walk starts at: 0, or the entry behind the root box when no ray of the call can miss that box
    uint32_t* item_counter;
    unsigned long long* ray_counter;  // [0] rays, [1] box tests, [2] primitive tests (then diagnostics)
    float* partial;       // [n_batches][H][W][3]
};

#ifndef RTX_WIDE_WAVES
#define RTX_WIDE_WAVES 4  // waves per SIMD the wide variants leave registers for
#endif
#ifndef RTX_FRAMELESS_WAVES
#define RTX_FRAMELESS_WAVES 6  // waves per SIMD the frame-less bracket variants leave registers for (80: three spilled, +6 % over 5)
#endif
constexpr int TILE = 8;            // 8x8 pixels = one pixel per lane of a wave
constexpr int MAX_WAVES_PER_BLOCK = 16;  // all waves of a block share one LDS copy of the entries; the launch picks 4, 8, 12 or 16
constexpr int BLOCK_THREADS = 64 * MAX_WAVES_PER_BLOCK;
constexpr int ACC_FLOATS = 64 * 3;   // one tile accumulator
constexpr int SLOT_FLOATS = ACC_FLOATS + 4;  // LDS per item slot of a wave: the accumulator and a 16-byte header (all accumulators first)

extern __shared__ uint4 smem[];

// Persistent wavefront path tracer.  Every lane of a wave carries one path and stands in one of four queues:
//   IDLE  (needs a new pixel sample)   BOX  (its next entry is a box)   PRIM  (its next entry is a primitive)
//   DONE  (its walk is complete and waits for shading).
// The queue a lane is in is word 7 of the entry it stands at (0 = box, ENT_END = done, ST_IDLE, else primitive), so
// arriving at an entry also files the lane.  Each pass the wave counts the queues and runs the stage whose queue is
// fullest -- box tests, primitive tests, or service (shade + scatter, then ray generation for the lanes that became
// free) -- for exactly the lanes queued for it.  Lanes are never moved: compaction happens in time, by always running
// the stage most lanes wait for.
// PK: walk the packed form of the fast array (16-byte slots, half-precision boxes) instead of the 32-byte entries; only
// for the fast walk of scenes without brackets (!STRICT && !XF).
// WIDE (with PK, array in HBM): walk the wide form instead -- four-child nodes of one 128-byte line each, primitives in
// five-slot records, a per-lane stack of pending child groups in LDS.  The box stage becomes the node stage: one step
// tests the four children of the lane's node, hands the primitives of the leaves it hit to the primitive stage and
// descends into the nearest child node.
// NOFRAMES (with XF, fast walk): the array has media brackets and / or instance primitives placed in the caller's frame but
// no transform-enter / leave entries, so a walk never leaves the caller's frame: no second copy of the ray, no frame ids.
template <bool STRICT, bool ENTS_IN_LDS, bool XF, bool TEX, bool PK = false, bool WIDE = false, bool NOFRAMES = false, bool NEST = false>
// Registers per variant, through the waves per SIMD the compiler must leave room for: the lean variants 64 (8 waves), the
// bracket variants 96 (5 waves: +10 % over 4 on the Cornell scenes, three spilled registers; at 6 they spill 47 and lose
// 20 %), the texture variants end up at 95 on their own.
__global__ __launch_bounds__(BLOCK_THREADS, XF ? (TEX ? 4 : (NOFRAMES ? RTX_FRAMELESS_WAVES : 5)) : (TEX ? 4 : (WIDE ? RTX_WIDE_WAVES : 8))) void render_kernel(RenderParams P) {
    static_assert(!PK || (!STRICT && !XF), "the packed array exists for the fast walk of bracket-free scenes only");
    static_assert(!WIDE || (PK && !ENTS_IN_LDS), "the wide array is walked from HBM in place of the packed one");
    static_assert(!NOFRAMES || (XF && !STRICT), "only the fast array of a scene with brackets can be free of frames");
    static_assert(!NEST || XF, "media nest in arrays with brackets only");
    constexpr bool FRAMES = XF && !NOFRAMES;
    constexpr int M = STRICT ? RTX_TRAVERSAL_STRICT : RTX_TRAVERSAL_FAST;
    constexpr int SLAB = STRICT ? SLAB_STRICT : SLAB_FMA;
    constexpr bool SCALED = PK && ENTS_IN_LDS;  // the walk runs in scaled ray parameters (pk_slab_hit_scaled); best.t is scaled too
    const DevScene& S = P.S;
    const uint4* __restrict__ gents = WIDE ? S.wprims : PK ? S.pk : S.ents[M];  // (WIDE: entry offsets are offsets into the primitive records)
    const uint32_t* __restrict__ rank = WIDE ? S.wrank : PK ? S.pk_rank : S.rank[M];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // what word 3 / word 7 of the entry a lane stands at says about the lane
    // (WIDE: one walk stage serves the lanes at a node and the lanes at a primitive alike, so every walking lane counts as "box")
    auto at_box = [](uint32_t w) { return WIDE ? w < ENT_END : PK ? w < PK_PRIM : w == ENT_BOX; };
    auto at_prim = [](uint32_t w) { return WIDE ? false : PK ? w - PK_PRIM < ENT_END - PK_PRIM : w - 1u < ENT_END - 1u; };

    // LDS: [entries incl. terminator (optional)] [per-wave tile accumulators]
    const uint32_t ent_u4 = ENTS_IN_LDS ? (PK ? S.pk_slots : 2u * S.n_ents[M]) : 0u;
    float* acc = reinterpret_cast<float*>(smem + ent_u4) + wave * (P.n_slots * SLOT_FLOATS);
    if (ENTS_IN_LDS) {
        for (uint32_t i = threadIdx.x; i < ent_u4; i += blockDim.x) smem[i] = gents[i];
    }
    // TEX: Perlin table set 0 (6 KB) behind the accumulators when the launch reserved room for it
    uint32_t lds_perlin = 0xFFFFFFFFu;
    if (TEX && P.perlin_in_lds != 0u) {
        const uint32_t at_u4 = ent_u4 + (blockDim.x >> 6) * (P.n_slots * SLOT_FLOATS / 4);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(S.perlin);
        for (uint32_t i = threadIdx.x; i < 1536u / 4u; i += blockDim.x) smem[at_u4 + i] = src[i];
        lds_perlin = at_u4 * 16u;
    }
    // Unpacked variants: the quad / triangle records go to LDS as well when they are few (a Cornell box: 18), so that a
    // primitive step does not wait twice for global memory (plane first, edges after the range check).
    const float4* planar_recs = S.planar;
    if (!PK && P.planar_in_lds != 0u) {
        const uint32_t at_u4 = ent_u4 + (blockDim.x >> 6) * (P.n_slots * SLOT_FLOATS / 4) + (TEX && P.perlin_in_lds != 0u ? 1536u / 4u : 0u);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(S.planar);
        for (uint32_t i = threadIdx.x; i < 5u * P.planar_in_lds; i += blockDim.x) smem[at_u4 + i] = src[i];
        planar_recs = reinterpret_cast<const float4*>(smem + at_u4);
    }
    __syncthreads();

    // The kernel arguments once more, as they lie in the kernarg segment.  Values the walk loops need (scheduler
    // constants, a pointer or two) are read from P and stay in scalar registers; everything a stage needs only when it
    // runs -- camera, item geometry, table pointers -- is read through RTX_ARGS() where it is used, so that it is
    // re-loaded there (scalar loads from a hot line) instead of occupying scalar registers across the whole loop: the
    // lean variants have 80 of them per wave, and what does not fit is spilled into vector registers.
    typedef __attribute__((address_space(4))) const RenderParams KArgs;
    KArgs* const kargs = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#define RTX_ARGS()                        \
    [&] {                                 \
        KArgs* p_ = kargs;                \
        asm volatile("" : "+s"(p_));      \
        return p_;                        \
    }()
    const float tmin = 0.001f;  // Camera::ray_color's interval start (camera.rs:127)
    uint32_t rays_traced = 0;  // per lane; summed in 64 bits at the end
    uint32_t wave_box = 0, wave_prim = 0;  // wave-uniform: lanes x steps of the box / primitive stage = tests performed
    uint32_t lane_box = 0;                 // box tests of this lane where they are counted per lane: the wide walk (child boxes, two to four
                                           // per node) and the hand-scheduled box runs over arrays in LDS
#ifdef RTX_STAGE_COUNTERS
    unsigned long long sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // box steps, box lanes, prim steps, prim lanes, service passes, shade lanes, gen lanes, sched passes
    unsigned long long tc[4] = {0, 0, 0, 0};              // wave clocks inside the box / primitive / service stage, and in total
    const unsigned long long t_begin = __builtin_readcyclecounter();
#define RTX_TIC() const unsigned long long tic_ = __builtin_readcyclecounter()
#define RTX_TOC(i) tc[i] += __builtin_readcyclecounter() - tic_
#else
#define RTX_TIC()
#define RTX_TOC(i)
#endif

    // per-lane path state
    uint32_t off = 0;       // byte offset of the entry the lane stands at
    uint32_t st = ST_IDLE;  // word 7 of that entry = the lane's queue.  Only these two cross from stage to stage: an
                            // entry itself is loaded by the stage that consumes it, so its eight registers are free
                            // everywhere else and no copies are needed where the stages' control flow merges
    Ray ray{};
    RayConst rc{};
    Best best{};
    float br = 0, bg = 0, bb = 0;  // path throughput
    uint32_t slot = 0, depth_left = 0;
    Rng rng{};
    // XF only: the caller's (world) ray and the bracket state of the walk (transforms, media)
    Ray wray{};
    WalkX wx{};
    // (XF only) where this lane parks the boundary-query state of enclosing media: asked for at a medium inside a medium
    auto med_stack = [&]() {
        KArgs* q_ = RTX_ARGS();
        return MedStack{q_->S.med_stack, q_->S.med_lanes, blockIdx.x * blockDim.x + threadIdx.x};
    };
    // WIDE only: the node to visit next, the pending group of the deepest level that has one (older groups are on the
    // lane's stack in LDS, w_sp of them), and the primitives handed out by the last node
    uint32_t w_node = WIDE_NONE, w_group = 0, w_sp = 0, w_tbase = 0, w_tmask = 0;
    // stack element k of this lane: dword (k * blockDim.x + threadIdx.x) from the stack base: lanes never share a bank
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t w_stack0 = WIDE ? P.wide_stack_u4 * 16u + threadIdx.x * 4u : 0u;
    const uint32_t w_stride = blockDim.x * 4u;
    // the lane's walk moves on to its next node: out of its pending group, else out of the newest group on its stack;
    // when nothing is pending the walk is complete
    auto wide_advance = [&]() {
        if (w_group == 0u && w_sp != 0u) {
            w_sp--;
            w_group = *(lds_u32*)(uintptr_t)(w_stack0 + w_sp * w_stride);
        }
        if (w_group == 0u) {
            st = ENT_END;
        } else {
            w_node = wide_group_pop(w_group);
            st = 0u;
        }
    };
    const typename std::conditional<ENTS_IN_LDS, EntReaderLds, EntReader>::type reader = [&] {
        if constexpr (ENTS_IN_LDS)
            return EntReaderLds{};
        else
            return EntReader{gents};
    }();

    // The entries start at LDS address 0 (this kernel has no static LDS, checked below), so a byte offset is already
    // the LDS address of an entry and no base needs adding per load.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const u32x4 lds_u4;
    if (ENTS_IN_LDS && (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem != 0u) __builtin_trap();
#define RTX_LOAD_ENTRY(o, a, b)                                                \
    do {                                                                       \
        if (ENTS_IN_LDS) {                                                     \
            lds_u4* base_ = (lds_u4*)(uintptr_t)(uint32_t)(o);                 \
            const u32x4 a_ = base_[0], b_ = base_[1];                          \
            a = make_uint4(a_.x, a_.y, a_.z, a_.w);                            \
            b = make_uint4(b_.x, b_.y, b_.z, b_.w);                            \
        } else {                                                               \
            const char* base_ = reinterpret_cast<const char*>(gents) + (o);    \
            a = *reinterpret_cast<const uint4*>(base_);                        \
            b = *reinterpret_cast<const uint4*>(base_ + 16);                   \
        }                                                                      \
    } while (0)
#define RTX_LOAD_SLOT(o, a)                                                                    \
    do {                                                                                       \
        if (ENTS_IN_LDS) {                                                                     \
            const u32x4 a_ = *(lds_u4*)(uintptr_t)(uint32_t)(o);                               \
            a = make_uint4(a_.x, a_.y, a_.z, a_.w);                                            \
        } else {                                                                               \
            a = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(gents) + (o));   \
        }                                                                                      \
    } while (0)
    constexpr uint32_t STATE_AT = PK ? 12u : 28u;  // byte offset of the state word in an entry
#define RTX_LOAD_STATE(o, w)                                                                              \
    do {                                                                                                 \
        if (ENTS_IN_LDS) {                                                                               \
            w = *(__attribute__((address_space(3))) const uint32_t*)(uintptr_t)(uint32_t)((o) + STATE_AT); \
        } else {                                                                                         \
            w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(gents) + (o) + STATE_AT); \
        }                                                                                                \
    } while (0)
    // reads slot k of the packed entry at byte offset o (shading, rare paths)
    auto slot_of = [&](uint32_t o) {
        return [=](int k) {
            uint4 v;
            RTX_LOAD_SLOT(o + (uint32_t)k * PK_SLOT, v);
            return v;
        };
    };
    uint32_t root_st = 0u;  // queue of the first entry (wave-uniform); a wide walk starts at the root node
    const uint32_t start_off = WIDE ? 0u : P.start_off;
    if (!WIDE) {
        RTX_LOAD_STATE(start_off, root_st);
        root_st = __builtin_amdgcn_readfirstlane(root_st);
    } else if (P.S.w_first != 0u) {
        root_st = PK_PRIM;  // a wide walk begins at the primitives that stand outside the tree, as if a node had handed them out
    }
#define RTX_BEGIN_WALK()                 \
    do {                                 \
        rc = ray_const<SLAB>(ray);       \
        if (SCALED) scale_ray_const(rc); \
        best.t = __builtin_inff();       \
        best.ent = NO_HIT;               \
        if (XF) {                        \
            if (FRAMES) wray = ray;      \
            walkx_init(wx, tmin);         \
        }                                \
        rays_traced++;                   \
        off = start_off;                 \
        st = root_st;                    \
        if (WIDE) {                      \
            w_node = 0u;                 \
            w_group = 0u;                \
            w_sp = 0u;                   \
            w_tbase = 0u;                \
            w_tmask = (1u << P.S.w_first) - 1u; \
        }                                \
    } while (0)

    // Several items are in flight per wave (P.n_slots of them), each with a tile accumulator of its own.  One, `cur`, still
    // has (pixel, sample) pairs in its pool: free lanes draw from it.  The others are draining: their pools are used up and
    // their last paths are under way.  When cur's pool runs out, every item none of whose paths is still alive is written
    // out and the next item is claimed into a free slot.  Without this an item would end with a drain phase in which most
    // lanes sit idle; and with two slots only, one long path (fifty bounces, each a chain of dependent memory reads in a
    // large scene) holds up every item behind it.
    uint32_t cur = 0;                                           // slot of the item whose pool is being drawn from
    uint32_t cur_txy = 0, cur_batch = 0;                        // its tile origin (x | y << 16) and batch of sample indices
    uint32_t pool_next = 0, pool_total = 0;                     // pool entry q = (sample q / 64, tile slot q % 64)
    uint32_t live = 0;                                          // bit k: slot k holds an item that has not been written out
    bool more = true;                                           // the global item counter may still have items
    const uint32_t n_slots = P.n_slots, all_slots = (1u << n_slots) - 1u;
    uint32_t* const item_hdr = reinterpret_cast<uint32_t*>(acc + n_slots * ACC_FLOATS);  // [slot][4]: tile origin, batch

    // Packed or wide arrays walked from HBM (the large scenes): each XCD has an L2 of its own, so items are handed out per XCD from eight parts of the image
    // (bands of tile columns; contiguous ranges of the owned tiles when the call renders a strided share of them), each
    // with a counter of its own: the waves of an XCD work next to each other in the image, and what their rays read stays
    // in that XCD's L2 instead of being fetched into all eight (million-triangle mesh: 17 % fewer L2 misses; the frame
    // time does not change, since that walk is bound by instruction issue).  A part that runs out sends its waves on to
    // the next part (they join its owners at the front of its counter), so the parts need not be equal in work.
    auto claim_near = [&](KArgs* q, uint32_t& tx, uint32_t& ty, uint32_t& batch) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
#pragma unroll 1
        for (uint32_t i = 0; i < 8u; i++) {
            const uint32_t part = (xcc + i) & 7u;
            const bool cols = q->xcd_parts == 1u;
            const uint32_t span = cols ? q->tiles_x : q->n_owned;            // what is cut into eight
            const uint32_t lo = part * span / 8u, hi = (part + 1u) * span / 8u;
            const uint32_t part_tiles = cols ? (hi - lo) * q->tiles_y : hi - lo;
            if (part_tiles == 0u) continue;
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd(q->item_counter + XCD_COUNTER_AT + part, 1u);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= part_tiles * q->n_batches) continue;
            batch = k / part_tiles;
            const uint32_t r = k % part_tiles;
            if (cols) {
                tx = lo + r % (hi - lo);
                ty = r / (hi - lo);
            } else {
                const uint32_t tile = q->tile_rank + (lo + r) * q->tile_world;
                tx = tile % q->tiles_x;
                ty = tile / q->tiles_x;
            }
            return true;
        }
        return false;
    };

#define RTX_CLAIM(k)                                                                                             \
    do {                                                                                                         \
        KArgs* q_ = RTX_ARGS();                                                                                  \
        uint32_t item_ = 0, tx_ = 0, ty_ = 0, batch_ = 0;                                                        \
        bool got_ = false;                                                                                       \
        if (PK && !ENTS_IN_LDS && q_->xcd_parts != 0u) {                                                         \
            got_ = claim_near(q_, tx_, ty_, batch_);                                                             \
        } else {                                                                                                 \
            if (lane == 0) item_ = atomicAdd(q_->item_counter, 1u);                                              \
            item_ = __builtin_amdgcn_readfirstlane(item_);                                                       \
            if (item_ < q_->n_items) {                                                                           \
                const uint32_t tile_ = q_->tile_rank + (item_ % q_->n_owned) * q_->tile_world;                   \
                got_ = true;                                                                                     \
                batch_ = item_ / q_->n_owned;                                                                    \
                tx_ = tile_ % q_->tiles_x;                                                                       \
                ty_ = tile_ / q_->tiles_x;                                                                       \
            }                                                                                                    \
        }                                                                                                        \
        if (!got_) {                                                                                             \
            more = false;                                                                                        \
        } else {                                                                                                 \
            cur = (k);                                                                                           \
            live |= 1u << cur;                                                                                   \
            cur_batch = __builtin_amdgcn_readfirstlane(batch_);                                                  \
            cur_txy = __builtin_amdgcn_readfirstlane((tx_ * TILE) | ((ty_ * TILE) << 16));                        \
            pool_total = __builtin_amdgcn_readfirstlane(64u * min(q_->batch, q_->sample_end - (q_->sample_begin + cur_batch * q_->batch))); \
            pool_next = 0;                                                                                       \
            acc[cur * ACC_FLOATS + lane] = 0.0f;                                                                 \
            acc[cur * ACC_FLOATS + lane + 64] = 0.0f;                                                            \
            acc[cur * ACC_FLOATS + lane + 128] = 0.0f;                                                           \
            if (lane == 0) {                                                                                     \
                item_hdr[4u * cur] = cur_txy;                                                                    \
                item_hdr[4u * cur + 1u] = cur_batch;                                                             \
            }                                                                                                    \
        }                                                                                                        \
    } while (0)

    RTX_CLAIM(0u);
    if (live != 0u) {
        for (;;) {
            const unsigned long long box_lanes = __ballot(at_box(st));
            const uint32_t n_box = (uint32_t)__popcll(box_lanes);
            // finished walks and free lanes: the two largest state words, one compare for both
            static_assert(ENT_END == 0xFFFFFFFEu && ST_IDLE == 0xFFFFFFFFu, "done and idle are the two largest state words");
            const uint32_t n_rest = (uint32_t)__popcll(__ballot(st >= ENT_END));
            const uint32_t n_prim = 64u - n_box - n_rest;
            // Fast path: the pool holds a pair for every lane, so every finished or free lane can be served and all 64
            // lanes have or can get work.  Everything else -- the pool running low or out, items to write out and to
            // claim -- sits behind one test of what the pool has left.
            uint32_t n_serv = n_rest, n_fixed = 64u;
            if (__builtin_expect(pool_total - pool_next < 64u, 0)) {  // (pool_next never passes pool_total)
                if (pool_next >= pool_total) {
                    // write out the items no lane carries a path of: the tile's sums go into the batch's partial image (each
                    // pixel of each batch has exactly one writer)
                    const uint32_t my_item = st != ST_IDLE ? slot >> 6 : 0xFFFFFFFFu;
                    for (uint32_t k = 0; k < n_slots; k++) {
                        if (((live >> k) & 1u) == 0u || __ballot(my_item == k) != 0ull) continue;
                        KArgs* Q = RTX_ARGS();
                        const uint32_t txy = __builtin_amdgcn_readfirstlane(item_hdr[4u * k]), batch = __builtin_amdgcn_readfirstlane(item_hdr[4u * k + 1u]);
                        const uint32_t px = (txy & 0xFFFFu) + (lane & 7u), py = (txy >> 16) + (lane >> 3);
                        if (px < Q->cam.width && py < Q->cam.height) {
                            float* dst = Q->partial + ((size_t)batch * Q->cam.height * Q->cam.width + (size_t)py * Q->cam.width + px) * 3;
                            dst[0] = acc[k * ACC_FLOATS + lane * 3 + 0];
                            dst[1] = acc[k * ACC_FLOATS + lane * 3 + 1];
                            dst[2] = acc[k * ACC_FLOATS + lane * 3 + 2];
                        }
                        live &= ~(1u << k);
                    }
                    if (more && live != all_slots) RTX_CLAIM((uint32_t)__builtin_ctz(~live));  // free lanes move on to the next item
                    if (live == 0u) break;  // nothing left anywhere
                }
                const uint32_t pool_left = pool_total - pool_next;
                // lanes the service stage has work for: every finished walk, and as many free lanes as the pool has pairs for
                if (pool_left < 64u) {
                    const uint32_t n_idle = (uint32_t)__popcll(__ballot(st == ST_IDLE));
                    n_serv = n_rest - n_idle + min(n_idle, pool_left);
                }
                n_fixed = n_box + n_prim + n_serv;  // lanes that have or can get work
                if (n_fixed == 0u) continue;  // only possible while an item with a zero-depth pool drains
            }
#ifdef RTX_STAGE_COUNTERS
            sc[7]++;
#endif

            // run the stage with the best score: queue length weighted by a per-stage constant (eighths)
            const uint32_t sc_box = n_box * 8u, sc_prim = n_prim * P.w_prim, sc_serv = n_serv * P.w_serv;
            // (some lane has work and the weights are at least 1, so the largest score is not 0: a stage without lanes never wins)
            if (sc_box >= max(sc_prim, sc_serv)) {
                // ---- extend, box stage: Aabb::hit for the lanes standing at a box; keep going while boxes stay the
                //      largest queue (one ballot per step: everything that is not BOX is "the rest")
                uint32_t nb = n_box;
                RTX_TIC();
                if (WIDE) {
                    // walk stage: a lane stands at a node (tests its four child boxes) or at a primitive the last node
                    // handed out (tests it).  Both kinds fetch first -- a node's four slots, a primitive's five, from the
                    // lane's own address -- and compute after, so one pass pays one trip to memory for every walking lane.
                    const uint32_t k1 = P.sched_k + 1u;
                    bool here = at_box(st);
                    uint32_t tests = 0;
                    do {
#ifdef RTX_STAGE_COUNTERS
                        sc[0]++;
                        sc[1] += nb;
#endif
                        const bool at_node = st == 0u;
                        // the kind fewer lanes stand at sits a pass out while the other has more than w_mix times as many:
                        // a wave executes both kinds' code one after the other, whatever the number of lanes in each
                        const uint32_t n_at_prim = (uint32_t)__popcll(__ballot(here && !at_node)), n_at_node = nb - n_at_prim;
                        const bool go_node = n_at_node * P.w_mix >= n_at_prim, go_prim = n_at_prim * P.w_mix >= n_at_node;
                        tests += go_prim ? n_at_prim : 0u;
                        if (here && (at_node ? go_node : go_prim)) {
                            const uint32_t prim_i = (uint32_t)__builtin_ctz(w_tmask | 0x80000000u);
                            const uint32_t prim_o = (w_tbase + prim_i * 5u) * PK_SLOT;
                            const char* addr = at_node ? reinterpret_cast<const char*>(P.S.wnodes) + (size_t)w_node * WIDE_NODE_BYTES
                                                       : reinterpret_cast<const char*>(gents) + prim_o;
                            const uint4* __restrict__ rec = reinterpret_cast<const uint4*>(addr);
                            const uint4 s0 = rec[0], s1 = rec[1];
                            uint4 s2 = rec[2], s3 = rec[3], s4;
                            asm("" : "=v"(s4.x), "=v"(s4.y), "=v"(s4.z), "=v"(s4.w));  // (only lanes at a primitive read it)
                            if (!at_node) s4 = rec[4];
                            // (whole slots in one load each: left to itself the compiler fetches the words only node lanes
                            // read with loads of their own, and the walk is bound by the number of vector memory instructions)
                            asm("" : "+v"(s2.x), "+v"(s2.y), "+v"(s2.z), "+v"(s2.w), "+v"(s3.x), "+v"(s3.y), "+v"(s3.z), "+v"(s3.w));
                            if (at_node) {
                                uint32_t next, group, n_children;
                                wide_node_step(s0, s1, s2, s3, rc, tmin, best.t, next, group, w_tbase, w_tmask, n_children);
                                lane_box += n_children;
                                if (group != 0u) {
                                    if (w_group != 0u) {
                                        *(lds_u32*)(uintptr_t)(w_stack0 + w_sp * w_stride) = w_group;
                                        w_sp++;
                                    }
                                    w_group = group;
                                }
                                w_node = next;
                                if (w_tmask != 0u) {
                                    st = PK_PRIM;  // the leaves' primitives first: a hit shortens the interval for the nodes to come
                                } else if (next == WIDE_NONE) {
                                    wide_advance();
                                }
                            } else {
                                w_tmask &= w_tmask - 1u;
                                pk_render_prim<false>(s0, s1, [&](int k) { return k == 2 ? s2 : k == 3 ? s3 : s4; }, prim_o, ray, rc, tmin, best, rank);
                                if (w_tmask == 0u) {
                                    if (w_node != WIDE_NONE)
                                        st = 0u;
                                    else
                                        wide_advance();
                                }
                            }
                        }
                        here = at_box(st);
                        nb = (uint32_t)__popcll(__ballot(here));
                    } while (nb * k1 >= n_fixed && nb > 0);
                    wave_prim += tests;
                } else if (PK) {
                    // The lane's state word is the box's miss link (in slots).  Arrays in LDS keep float boxes (two slots),
                    // arrays in HBM half-precision ones (one slot).
                    constexpr uint32_t BOX_BYTES = ENTS_IN_LDS ? 2u * PK_SLOT : PK_SLOT;
                    const uint32_t k1 = P.sched_k + 1u;
                    bool here = at_box(st);
                    // lanes that are not at a box never read e / e1: leave their registers as they are instead of clearing them
                    uint4 e, e1;
                    asm("" : "=v"(e.x), "=v"(e.y), "=v"(e.z), "=v"(e.w), "=v"(e1.x), "=v"(e1.y), "=v"(e1.z), "=v"(e1.w));
                    uint32_t tests = 0;  // summed in the loop's own scalar and added to the wave's tally once per pass
#if defined(RTX_STAGE_COUNTERS) || defined(RTX_BOX_LOOP_COMPILED)
                    constexpr bool BOX_RUN = false;  // (the stage counters count steps in the loop below)
#else
                    constexpr bool BOX_RUN = ENTS_IN_LDS;
#endif
                    if (BOX_RUN) {
                        // the loop below, hand-scheduled; it goes on while lanes x k1 >= n_fixed
                        const uint32_t min_lanes = n_fixed == 64u ? P.box_run_min : (n_fixed + k1 - 1u) / k1;
                        pk_box_run_lds(box_lanes, off, st, lane_box, rc, best.t, min_lanes);
                    } else {
                        if (here) {
                            RTX_LOAD_SLOT(off, e);
                            if (ENTS_IN_LDS) RTX_LOAD_SLOT(off + PK_SLOT, e1);
                        }
                        do {
                            tests += nb;
#ifdef RTX_STAGE_COUNTERS
                            sc[0]++;
                            sc[1] += nb;
#endif
                            if (here) {
                                const bool hit = pk_render_box<SCALED>(e, e1, rc, tmin, best.t);
                                off = hit ? off + BOX_BYTES : (ENTS_IN_LDS ? st : st * PK_SLOT);  // float boxes carry byte links
                                RTX_LOAD_SLOT(off, e);
                                if (ENTS_IN_LDS) RTX_LOAD_SLOT(off + PK_SLOT, e1);
                                st = e.w;
                            }
                            here = at_box(st);
                            nb = (uint32_t)__popcll(__ballot(here));
                        } while (nb * k1 >= n_fixed && nb > 0);
                    }
                    wave_box += tests;
#if !defined(RTX_STAGE_COUNTERS) && !defined(RTX_BOX_LOOP_COMPILED)
                } else if (!STRICT && ENTS_IN_LDS) {
                    // fast walk over 32-byte entries in LDS: the loop below, hand-scheduled like the packed one
                    const uint32_t k1 = P.sched_k + 1u;
                    const uint32_t min_lanes = n_fixed == 64u ? P.box_run_min : (n_fixed + k1 - 1u) / k1;
                    box_run_lds(box_lanes, off, st, lane_box, rc, XF ? wx.tmin : tmin, best.t, min_lanes);
#endif
                } else {
                    uint4 ea, eb;  // (as above: only lanes at a box read them)
                    asm("" : "=v"(ea.x), "=v"(ea.y), "=v"(ea.z), "=v"(ea.w), "=v"(eb.x), "=v"(eb.y), "=v"(eb.z), "=v"(eb.w));
                    if (st == ENT_BOX) RTX_LOAD_ENTRY(off, ea, eb);
                    do {
                        wave_box += nb;
#ifdef RTX_STAGE_COUNTERS
                        sc[0]++;
                        sc[1] += __popcll(__ballot(st == ENT_BOX));
#endif
                        if (st == ENT_BOX) {
                            const bool hit = slab_hit<SLAB>(ea, eb, ray, rc, XF ? wx.tmin : tmin, best.t);
                            off = hit ? off + ENT_BYTES : eb.z;
                            RTX_LOAD_ENTRY(off, ea, eb);
                            st = eb.w;
                        }
                        nb = (uint32_t)__popcll(__ballot(st == ENT_BOX));
                    } while (nb * P.sched_k >= (n_fixed - nb) && nb > 0);
                }
                RTX_TOC(0);
            } else if (sc_prim >= sc_serv) {
                // ---- extend, primitive stage: Sphere::hit / Quad::hit for the lanes standing at a primitive; repeated
                //      while primitives stay the bulk of the work (lists, leaves and instances put them in a row)
                uint32_t np = n_prim;
                RTX_TIC();
                do {
                    wave_prim += np;
#ifdef RTX_STAGE_COUNTERS
                    sc[2]++;
                    sc[3] += np;
#endif
                    if (WIDE) {
                        // (the walk stage tests primitives as well)
                    } else if (PK) {
                        if (at_prim(st)) {
                            uint4 s0, s1;
                            RTX_LOAD_SLOT(off, s0);
                            RTX_LOAD_SLOT(off + PK_SLOT, s1);
                            off += pk_render_prim<SCALED>(s0, s1, slot_of(off), off, ray, rc, tmin, best, rank);
                            RTX_LOAD_STATE(off, st);
                        }
                    } else if (at_prim(st)) {  // neither box (0) nor done / idle
                        uint4 ea, eb;
                        RTX_LOAD_ENTRY(off, ea, eb);
                        const float lane_tmin = XF ? wx.tmin : tmin;
                        const uint32_t kind15 = eb.w & 15u;
                        if (XF && (NOFRAMES ? is_bracket_kind(kind15) : kind15 >= ENT_XF_PUSH)) {
                            // transform enter / leave, medium begin / end (mesh.rs:324-432, :627-696)
                            if constexpr (FRAMES)
                                bracket_step<STRICT, SLAB, true, NEST, !STRICT>(reader, ea, eb, off, ray, rc, wray, best, wx, med_stack, tmin, rng, rank);
                            else
                                bracket_step<STRICT, SLAB, false, NEST, !STRICT>(reader, ea, eb, off, ray, rc, ray, best, wx, med_stack, tmin, rng, rank);
                        } else {
                            // A primitive of an instance placed in this frame's tree (frame-less arrays only: the host places
                            // either every instance of a scene or none) is tested in its own frame, with a moved copy of the ray.
                            const uint32_t kind = NOFRAMES ? kind15 & 3u : kind15;  // (not a bracket: sphere, quad or triangle, placed or not)
                            auto test = [&](const Ray& r_, const RayConst& rc_) {
                                if (kind == ENT_SPHERE) {
                                    sphere_test<STRICT, 5, !STRICT>(ea, eb, r_, rc_, lane_tmin, best, off, rank);  // (fast walk: roots through the hardware square root / reciprocal)
                                } else {
                                    const uint32_t pi = eb.w >> 4;
                                    const bool tri = kind == ENT_TRI;
                                    planar_test<STRICT, !STRICT>(planar_recs, tri ? pi + S.n_quads : pi, tri, r_, lane_tmin, best, off, rank);
                                }
                            };
                            if (NOFRAMES) {
                                Ray tr = ray;
                                RayConst trc = rc;
                                if ((eb.w & ENT_INST) != 0u) inst_enter(reader, ea, eb, tr, trc.a);
                                test(tr, trc);
                            } else {
                                test(ray, rc);
                            }
                            if (FRAMES && best.ent == off) wx.best_xf = wx.cur_xf;
                            off += ENT_BYTES;
                        }
                        RTX_LOAD_STATE(off, st);
                    }
                    np = (uint32_t)__popcll(__ballot(at_prim(st)));
                } while ((XF || (PK && !ENTS_IN_LDS)) && np * (XF ? P.sched_kp : 8u * P.sched_kp) >= 8u * (n_fixed - np) && np > 0);  // (small scenes: one step, the loop is not worth its code;
                  // leaves of a large mesh put up to four triangles in a row, and every pass through the scheduler waits for a state word from HBM: +2.5 %)
                RTX_TOC(1);
            } else {
                // ---- service stage 1, shade: camera.rs:127-142 for every lane whose walk is complete
                RTX_TIC();
#ifdef RTX_STAGE_COUNTERS
                sc[4]++;
                sc[5] += (uint32_t)__popcll(__ballot(st == ENT_END));
                sc[6] += n_serv - (uint32_t)__popcll(__ballot(st == ENT_END));
#endif
                KArgs* Q = RTX_ARGS();
                __attribute__((address_space(4))) const rtx_camera_t& C = Q->cam;
                // Keep the 32-bit tallies from wrapping, however much work one item holds: checked once per service pass (a
                // check per box pass costs 3 % on small scenes), between two of which a wave adds at most 64 lanes x one walk
                // each -- far below the 3e9 of headroom.
                if ((wave_box | wave_prim) >= 0x40000000u) {
                    if (lane == 0) {
                        atomicAdd(Q->ray_counter + 1, (unsigned long long)wave_box);
                        atomicAdd(Q->ray_counter + 2, (unsigned long long)wave_prim);
                    }
                    wave_box = 0;
                    wave_prim = 0;
                }
                bool begin_walk = false;  // scattered and newly generated rays start their walk together, below
                // what a hit is shaded with (layout: DevScene::shade), indexed like the entries / slots
                auto shade_rec = [&](uint4& s0, uint4& s1) {
                    if (PK && !ENTS_IN_LDS) {
                        // large scenes keep no per-slot table: the entry names its material (a sphere through its record)
                        const auto slots = slot_of(best.ent);
                        const uint4 first = slots(0);
                        const uint32_t m = (first.w >> 28) == ENT_SPHERE ? __float_as_uint(Q->S.sph[2u * (first.w & PK_INDEX_MASK) + 1u].w) : slots(2).w;
                        s0 = Q->S.mat_recs[2u * m];
                        s1 = Q->S.mat_recs[2u * m + 1u];
                    } else {
                        const uint4* srec = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(PK ? Q->S.pk_shade : Q->S.shade[M]) + (PK ? 2u * best.ent : best.ent));
                        s0 = srec[0];
                        s1 = srec[1];
                    }
                };
                // Material bins (builds with -DRTX_MAT_BINS only; measured 28% slower on the benchmark scene): of the lanes
                // waiting with a hit, only those of the most frequent material kind are shaded in this pass, so that the pass
                // runs one material's code; the others keep waiting.
                bool defer = false;
#ifdef RTX_MAT_BINS
                {
                    uint32_t kind = 0xFFu;
                    if (st == ENT_END && best.ent != NO_HIT) {
                        uint4 s0, s1;
                        shade_rec(s0, s1);
                        kind = s0.x & 255u;
                    }
                    uint32_t top_kind = 0u, top_n = 0u;
                    for (uint32_t k = 0; k <= RTX_MAT_ISOTROPIC; k++) {
                        const uint32_t n = (uint32_t)__popcll(__ballot(kind == k));
                        if (n > top_n) {
                            top_n = n;
                            top_kind = k;
                        }
                    }
                    defer = kind != 0xFFu && kind != top_kind;
                }
#endif
                if (st == ENT_END && !defer) {
                    st = ST_IDLE;
                    if (best.ent == NO_HIT) {
                        atomicAdd(&acc[slot * 3 + 0], br * C.background[0]);
                        atomicAdd(&acc[slot * 3 + 1], bg * C.background[1]);
                        atomicAdd(&acc[slot * 3 + 2], bb * C.background[2]);
                    } else {
                        HitRec h;
                        uint4 s0, s1;  // what the hit is shaded with (layout: DevScene::shade)
                        shade_rec(s0, s1);  // issued first: it is the long read
                        if (PK) {
                            const auto slots = slot_of(best.ent);
                            // uv costs an acos and an atan2: evaluate it only for materials whose texture reads it
                            const bool want_uv = TEX && (s0.x & 512u) != 0u;
                            h = want_uv ? pk_render_hit<SCALED, true, false>(S, slots, ray, best.t) : pk_render_hit<SCALED, false, false>(S, slots, ray, best.t);
                        } else {
                            uint4 ha, hb;
                            RTX_LOAD_ENTRY(best.ent, ha, hb);
                            const bool want_uv = TEX && (s0.x & 512u) != 0u;
                            if (XF) {
                                const uint32_t hit_xf = FRAMES ? wx.best_xf : XF_NONE;
                                h = want_uv ? make_hit_x<true, false, NOFRAMES>(S, reader, ray, best, hit_xf) : make_hit_x<false, false, NOFRAMES>(S, reader, ray, best, hit_xf);
                            } else {
                                h = want_uv ? make_hit<true, false>(S, ha, hb, ray, best.t) : make_hit<false, false>(S, ha, hb, ray, best.t);
                            }
                        }
                        uint4 mrec = make_uint4(s0.x & 255u, s0.z, s0.y, 0u);
                        bool solid = (s0.x & 256u) != 0u;
                        float cr = as_f(s1.x), cg = as_f(s1.y), cb = as_f(s1.z);
                        if ((s0.x & 1024u) != 0u) {
                            // CheckerTexture::value (texture.rs:63-74) of two solid colours, all of it in the record
                            const float scale = as_f(s0.y);
                            const int32_t cx = as_i32_sat(floorf(h.px * scale)), cy = as_i32_sat(floorf(h.py * scale)), cz = as_i32_sat(floorf(h.pz * scale));
                            const bool even = ((int32_t)((uint32_t)cx + (uint32_t)cy + (uint32_t)cz)) % 2 == 0;
                            cr = even ? as_f(s0.z) : as_f(s1.y);
                            cg = even ? as_f(s0.w) : as_f(s1.z);
                            cb = even ? as_f(s1.x) : as_f(s1.w);
                            mrec.y = 0u;
                            mrec.z = 0u;
                            solid = true;
                        }
                        ScatterOut so = shade<TEX>(S, ray, h, rng, mrec, solid, cr, cg, cb, lds_perlin);
                        if (so.er != 0.0f || so.eg != 0.0f || so.eb != 0.0f) {
                            atomicAdd(&acc[slot * 3 + 0], br * so.er);
                            atomicAdd(&acc[slot * 3 + 1], bg * so.eg);
                            atomicAdd(&acc[slot * 3 + 2], bb * so.eb);
                        }
                        if (so.scattered) {
                            br *= so.ar;
                            bg *= so.ag;
                            bb *= so.ab;
                            depth_left--;
                            bool alive = depth_left > 0;
                            if (Q->rr_start != 0u) {  // wave-uniform: the reference has no Russian roulette
                                if (alive && (C.max_depth - depth_left) >= Q->rr_start) {
                                    // on the largest |throughput| component (albedos may be negative)
                                    const float q = fminf(1.0f, fmaxf(fmaxf(fabsf(br), fabsf(bg)), fabsf(bb)));
                                    if (rng_f32(rng) >= q) {
                                        alive = false;
                                    } else {
                                        const float iq = __builtin_amdgcn_rcpf(q);
                                        br *= iq;
                                        bg *= iq;
                                        bb *= iq;
                                    }
                                }
                            }
                            if (alive) {
                                ray = so.ray;
                                st = root_st;  // no longer free: ray generation below must not take this lane
                                begin_walk = true;
                            }
                        }
                    }
                }
                // ---- service stage 2, ray generation: free lanes take the next (pixel, sample) pairs of the pool
                //      (camera.rs:147-163)
                const unsigned long long idle = __ballot(st == ST_IDLE);
                if (idle && pool_next < pool_total) {
                    const uint32_t g_tx = cur_txy & 0xFFFFu, g_ty = cur_txy >> 16;
                    const uint32_t g_s0 = Q->sample_begin + cur_batch * Q->batch;
                    const uint32_t my = pool_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    const uint32_t my_slot = my & 63u;
                    const uint32_t x = g_tx + (my_slot & 7u), y = g_ty + (my_slot >> 3);
                    if (st == ST_IDLE && my < pool_total && x < C.width && y < C.height) {
                        slot = my_slot | (cur << 6);
                        const uint32_t sample = g_s0 + (my >> 6);
                        rng = rng_init(Q->seed0, Q->seed1, y * C.width + x, sample);
                        const float ox = rng_f32(rng) - 0.5f;
                        const float oy = rng_f32(rng) - 0.5f;
                        const float fx = (float)x + ox, fy = (float)y + oy;
                        const float sx = (C.pixel00_loc[0] + (fx * C.pixel_delta_u[0])) + (fy * C.pixel_delta_v[0]);
                        const float sy = (C.pixel00_loc[1] + (fx * C.pixel_delta_u[1])) + (fy * C.pixel_delta_v[1]);
                        const float sz = (C.pixel00_loc[2] + (fx * C.pixel_delta_u[2])) + (fy * C.pixel_delta_v[2]);
                        ray.ox = C.center[0];
                        ray.oy = C.center[1];
                        ray.oz = C.center[2];
                        if (!(C.defocus_angle <= 0.0f)) {
                            float dxk, dyk;
                            random_in_unit_disk(rng, dxk, dyk);
                            ray.ox = (C.center[0] + (dxk * C.defocus_disk_u[0])) + (dyk * C.defocus_disk_v[0]);
                            ray.oy = (C.center[1] + (dxk * C.defocus_disk_u[1])) + (dyk * C.defocus_disk_v[1]);
                            ray.oz = (C.center[2] + (dxk * C.defocus_disk_u[2])) + (dyk * C.defocus_disk_v[2]);
                        }
                        ray.dx = sx - ray.ox;
                        ray.dy = sy - ray.oy;
                        ray.dz = sz - ray.oz;
                        ray.time = rng_f32(rng);
                        br = bg = bb = 1.0f;
                        depth_left = C.max_depth;
                        // ray_color(depth 0) is black before any hit test (camera.rs:125): the lane stays free
                        begin_walk = depth_left != 0;
                    }
                    pool_next = __builtin_amdgcn_readfirstlane(min(pool_total, pool_next + (uint32_t)__popcll(idle)));
                }
                if (begin_walk) RTX_BEGIN_WALK();
                RTX_TOC(2);
            }
        }
    }
#undef RTX_CLAIM
#undef RTX_LOAD_ENTRY
#undef RTX_LOAD_SLOT
#undef RTX_LOAD_STATE
#undef RTX_BEGIN_WALK
    // One update of the launch's tallies per block: thousands of waves adding to the same three words at the end of a
    // launch take longer than a short launch itself.  Every wave leaves its sums in its own accumulator space.
    unsigned long long wave_rays = rays_traced;
    for (int o = 32; o > 0; o >>= 1) wave_rays += __shfl_down(wave_rays, o);
    unsigned long long wave_lane_box = lane_box;
    for (int o = 32; o > 0; o >>= 1) wave_lane_box += __shfl_down(wave_lane_box, o);
    unsigned long long* const tally = reinterpret_cast<unsigned long long*>(acc);
    if (lane == 0) {
        tally[0] = wave_rays;
        tally[1] = wave_lane_box + (unsigned long long)wave_box;  // (the wide walk and the box runs in LDS count per lane, the others per wave)
        tally[2] = wave_prim;
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        unsigned long long sum = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); w++) sum += reinterpret_cast<const unsigned long long*>(reinterpret_cast<float*>(smem + ent_u4) + w * (P.n_slots * SLOT_FLOATS))[threadIdx.x];
        if (sum) atomicAdd(RTX_ARGS()->ray_counter + threadIdx.x, sum);
    }
#ifdef RTX_STAGE_COUNTERS
    tc[3] = __builtin_readcyclecounter() - t_begin;
    if (lane == 0) {
        for (int i = 0; i < 8; i++) atomicAdd(P.ray_counter + 6 + i, sc[i]);
        for (int i = 0; i < 4; i++) atomicAdd(P.ray_counter + 14 + i, tc[i]);
    }
#endif
#undef RTX_TIC
#undef RTX_TOC
#undef RTX_ARGS
}

__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ partial, uint32_t n_batches, size_t n, float* __restrict__ accum, uint32_t width,
                                                     uint32_t tiles_x, uint32_t tile_rank, uint32_t tile_world) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (tile_world > 1u) {  // pixels of tiles another rank owns were not rendered by this call
        const uint32_t pix = (uint32_t)(i / 3), px = pix % width, py = pix / width;
        const uint32_t tile = (py / TILE) * tiles_x + px / TILE;
        if (tile % tile_world != tile_rank) return;
    }
    float s = accum[i];
    for (uint32_t b = 0; b < n_batches; b++) s += partial[(size_t)b * n + i];
    accum[i] = s;
}

__global__ __launch_bounds__(256) void resolve_kernel(const float* __restrict__ accum, size_t n, float scale, uint8_t* __restrict__ rgb8) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rgb8[i] = srgb_u8(scale * accum[i]);  // pixel_samples_scale * color (camera.rs:119), then camera.rs:101
}

static int check_camera(const rtx_camera_t* c) {
    if (!c) return fail(RTX_ERR_INVALID, "null camera");
    if (c->width == 0 || c->height == 0) return fail(RTX_ERR_INVALID, "zero image size");
    // a tile origin is kept as x | y << 16
    if (c->width > 65535u || c->height > 65535u) return fail(RTX_ERR_INVALID, "image wider or taller than 65535 pixels");
    if ((uint64_t)c->width * c->height > (1ull << 30)) return fail(RTX_ERR_INVALID, "image too large");
    if (c->samples_per_pixel == 0) return fail(RTX_ERR_INVALID, "samples_per_pixel is zero");
    return RTX_OK;
}

// What a camera needs from the fast array: the viewpoint it should be ordered for and how far its boxes must be grown.
// `reach` is the largest |coordinate| a camera ray can start from; beyond the range the boxes were padded for, they are
// enlarged by the rounding error of the fused box tests at that distance (~2.4e-7 * |origin| per plane, taken 3x).
struct ViewNeed {
    bool strict_only;  // no finite reach: only the strict walk is safe
    float grow;
};
static ViewNeed view_need(const rtx_scene* scene, const rtx_camera_t* cam) {
    float cmax = std::fmax(std::fabs(cam->center[0]), std::fmax(std::fabs(cam->center[1]), std::fabs(cam->center[2])));
    float dmax = 0.0f;
    for (int k = 0; k < 3; k++) dmax = std::fmax(dmax, std::fmax(std::fabs(cam->defocus_disk_u[k]), std::fabs(cam->defocus_disk_v[k])));
    const float reach = cmax + 2.0f * dmax;
    ViewNeed n;
    n.strict_only = !std::isfinite(reach) || reach > 1e12f;
    n.grow = (!n.strict_only && reach > scene->coord_range) ? 7e-7f * reach : 0.0f;
    return n;
}

// Re-orders the fast array for rays leaving `center` (children nearer to it are visited first) and replaces the device
// copy.  Only the visiting order changes: every trace returns the same hit as before.
static int order_fast_for_view(rtx_scene* scene, DeviceCopy* c, const float center[3], float grow) {
    if (c->view_valid && std::memcmp(c->view, center, sizeof(c->view)) == 0 && grow == c->view_grow) return RTX_OK;
    if (scene->wide) {
        // A wide walk orders the children of a node per ray, so there is nothing to order for a viewpoint; only boxes that
        // must be enlarged for a distant camera are rebuilt (same tree, same sizes) and uploaded.
        if (grow != c->wide_grow) {
            rtx_accel::Wide w;
            if (!rtx_accel::WideBuilder(scene->fast_indexed, reinterpret_cast<const float*>(scene->planar_host.data()), scene->planar_id_host.data(), scene->n_quads, grow, w).run() ||
                w.n_nodes() != scene->w_nodes || w.n_slots() != scene->w_slots)
                return fail(RTX_ERR_DEVICE, "internal: rebuilt wide array changed size");
            HIP_TRY(hipDeviceSynchronize());  // no walk may be in flight while the nodes are replaced
            HIP_TRY(hipMemcpy(const_cast<uint4*>(c->dev.wnodes), w.nodes.data(), w.nodes.size() * 4, hipMemcpyHostToDevice));
            c->wide_grow = grow;
        }
        std::memcpy(c->view, center, sizeof(c->view));
        c->view_grow = grow;
        c->view_valid = true;
        return RTX_OK;
    }
    rtx_accel::Threaded ordered;
    rtx_accel::Reorder(scene->fast_indexed, ordered).run(center, grow);
    rtx_accel::Packed pk;
    if (scene->packed) {
        if (!rtx_accel::pack(ordered, reinterpret_cast<const float*>(scene->planar_host.data()), scene->planar_id_host.data(), scene->n_quads, scene->pk_half, pk) ||
            pk.n_slots() != scene->pk_slots)
            return fail(RTX_ERR_DEVICE, "internal: re-ordered packed array changed size");
    }
    rtx_accel::finalize(ordered);
    if (ordered.ents.size() != scene->n_ents[RTX_TRAVERSAL_FAST]) return fail(RTX_ERR_DEVICE, "internal: re-ordered array changed size");
    HIP_TRY(hipDeviceSynchronize());  // no walk may be in flight while the array is replaced
    if (scene->packed) {
        HIP_TRY(hipMemcpy(const_cast<uint4*>(c->dev.pk), pk.words.data(), pk.words.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(const_cast<uint32_t*>(c->dev.pk_rank), pk.rank.data(), pk.rank.size() * 4, hipMemcpyHostToDevice));
        if (c->dev.pk_shade) {
            const std::vector<uint4> sh = expand_shade_packed(pk, scene->mat_recs);
            HIP_TRY(hipMemcpy(const_cast<uint4*>(c->dev.pk_shade), sh.data(), sh.size() * sizeof(uint4), hipMemcpyHostToDevice));
        }
    }
    HIP_TRY(hipMemcpy(const_cast<uint4*>(c->dev.ents[RTX_TRAVERSAL_FAST]), ordered.ents.data(), ordered.ents.size() * sizeof(rtx_accel::Ent), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<uint32_t*>(c->dev.rank[RTX_TRAVERSAL_FAST]), ordered.rank.data(), ordered.rank.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const std::vector<uint4> shade = expand_shade(ordered, scene->mat_recs);
    HIP_TRY(hipMemcpy(const_cast<uint4*>(c->dev.shade[RTX_TRAVERSAL_FAST]), shade.data(), shade.size() * sizeof(uint4), hipMemcpyHostToDevice));
    std::memcpy(c->view, center, sizeof(c->view));
    c->view_grow = grow;
    c->view_valid = true;
    return RTX_OK;
}

// Register count of a kernel variant, and the largest dynamic LDS size its launches were enabled for: asked of the runtime
// once per variant (and device), so that a call in a tight loop makes no attribute queries.
static std::mutex g_kernel_info_mutex;
static int kernel_regs(const void* kern, uint32_t* regs_out) {
    static std::vector<std::pair<const void*, uint32_t>> cache;
    std::lock_guard<std::mutex> lock(g_kernel_info_mutex);
    for (const auto& e : cache)
        if (e.first == kern) {
            *regs_out = e.second;
            return RTX_OK;
        }
    hipFuncAttributes fa;
    HIP_TRY(hipFuncGetAttributes(&fa, kern));
    cache.emplace_back(kern, (uint32_t)std::max(fa.numRegs, 8));
    *regs_out = cache.back().second;
    return RTX_OK;
}
static int kernel_allow_lds(const void* kern, int device, size_t lds) {
    struct Entry {
        const void* kern;
        int device;
        size_t lds;
    };
    static std::vector<Entry> cache;
    if (lds <= 48 * 1024) return RTX_OK;
    std::lock_guard<std::mutex> lock(g_kernel_info_mutex);
    for (Entry& e : cache)
        if (e.kern == kern && e.device == device) {
            if (e.lds >= lds) return RTX_OK;
            HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            e.lds = lds;
            return RTX_OK;
        }
    HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cache.push_back(Entry{kern, device, lds});
    return RTX_OK;
}

// Where a render keeps its library-owned device memory.
struct Scratch {
    float* partial;        // per-batch partial images
    size_t partial_bytes;  // capacity
    uint32_t* counters;    // COUNTER_BYTES
    bool may_grow;         // partial may be re-allocated to the size the frame wants
};

// One render_kernel launch + reduce on copy `c` (which lives on the current device).
static int render_accumulate_on(rtx_scene* scene, DeviceCopy* c, const rtx_camera_t* cam, const rtx_render_opts_t* opts, float* accum_dev,
                                hipStream_t st, rtx_stats_t* stats, Scratch* scratch, bool may_prepare) {
    if (!scene || !opts || !accum_dev) return fail(RTX_ERR_INVALID, "null argument");
    int rc = check_camera(cam);
    if (rc != RTX_OK) return rc;
    if (opts->traversal > RTX_TRAVERSAL_STRICT) return fail(RTX_ERR_INVALID, "unknown traversal mode");
    const uint32_t sb = opts->sample_begin, se = opts->sample_end;
    if (se < sb) return fail(RTX_ERR_INVALID, "sample_end < sample_begin");
    auto t_start = std::chrono::steady_clock::now();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (se == sb) return RTX_OK;
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != c->device) return fail(RTX_ERR_INVALID, "scene lives on another device than the current one");
    const Knobs& K = knobs();

    uint32_t M = opts->traversal;
    if (M == RTX_TRAVERSAL_FAST) {
        // The fused box test of the fast walk needs boxes padded for the distance rays start from: the scene's own
        // coordinate range covers bounce rays, the camera's position is covered when the array is ordered for it.
        const ViewNeed need = view_need(scene, cam);
        if (need.strict_only) {
            M = RTX_TRAVERSAL_STRICT;
        } else if (may_prepare) {
            if (K.view_order != 0u || need.grow > 0.0f) {
                if ((rc = order_fast_for_view(scene, c, cam->center, need.grow)) != RTX_OK) return rc;
            }
        } else if ((c->view_valid ? c->view_grow : 0.0f) < need.grow) {
            return fail(RTX_ERR_UNSUPPORTED, "the scene's fast traversal data is not prepared for this camera position: render once with "
                                             "rtx_render_accumulate from this camera centre first, or use RTX_TRAVERSAL_STRICT");
        }
    }
    RenderParams P;
    std::memset(&P, 0, sizeof(P));
    P.S = c->dev;
    P.cam = *cam;
    P.seed0 = seed_word(opts->seed, 0);
    P.seed1 = seed_word(opts->seed, 1);
    P.sample_begin = sb;
    P.sample_end = se;
    P.tiles_x = (cam->width + TILE - 1) / TILE;
    P.tiles_y = (cam->height + TILE - 1) / TILE;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    const uint32_t spp = se - sb;
    const size_t npix3 = (size_t)cam->width * cam->height * 3;
    // The sample range is cut into batches so that the chip's waves (CUs x 32) get several items each: many items
    // balance the tail of the launch, but an item shorter than ~20 samples per pixel spends too much of its life
    // filling and draining.  Between 4 and 64 items per wave, as many as keep a batch at 20 samples or more.
    const uint32_t waves = (uint32_t)c->n_cus * 32u;
    const uint32_t tile_world = std::max<uint32_t>(1u, opts->tile_world);
    const uint32_t my_tiles = std::max<uint32_t>(1u, (n_tiles + tile_world - 1) / tile_world);  // (about) the tiles this call owns
    uint32_t n_batches;
    if (K.items_per_wave != 0u) {
        n_batches = (waves * K.items_per_wave + my_tiles - 1) / my_tiles;
    } else {
        const uint32_t lo = (waves * 4u + my_tiles - 1) / my_tiles, hi = (waves * 64u + my_tiles - 1) / my_tiles;
        n_batches = std::min(hi, std::max(lo, spp / 20u));
    }
    n_batches = std::min<uint32_t>(spp, std::max<uint32_t>(1u, n_batches));
    n_batches = std::min<uint32_t>(n_batches, std::max(1u, K.max_batches));
    if (!scratch->may_grow) {
        // fixed scratch: as many batches as fit
        const size_t fit = scratch->partial_bytes / (npix3 * sizeof(float));
        if (fit == 0) return fail(RTX_ERR_UNSUPPORTED, "frame too large for the fixed scratch array: 12 * width * height bytes exceed RTX_STATIC_SCRATCH_BYTES");
        n_batches = (uint32_t)std::min<size_t>(n_batches, fit);
    }
    P.batch = (spp + n_batches - 1) / n_batches;
    P.n_batches = (spp + P.batch - 1) / P.batch;
    P.tile_world = std::max<uint32_t>(1u, opts->tile_world);
    P.tile_rank = opts->tile_rank;
    if (P.tile_rank >= P.tile_world) return fail(RTX_ERR_INVALID, "tile_rank >= tile_world");
    P.n_owned = n_tiles > P.tile_rank ? (n_tiles - P.tile_rank + P.tile_world - 1) / P.tile_world : 0u;
    P.n_items = P.n_owned * P.n_batches;
    if (P.n_items == 0) return RTX_OK;
    P.rr_start = opts->rr_start_depth;
    P.sched_k = K.sched_k;
    P.w_prim = scene->has_brackets ? K.w_prim_xf : K.w_prim;
    P.w_serv = K.w_serv;
    P.sched_kp = K.sched_kp;
    P.w_mix = std::max(1u, K.w_mix);

    const size_t need = npix3 * sizeof(float) * P.n_batches;
    if (scratch->partial_bytes < need) {  // only with may_grow
        HIP_TRY(hipStreamSynchronize(st));
        if (c->partial) hipFree(c->partial);
        c->partial = nullptr;
        c->partial_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->partial, need));
        c->partial_bytes = need;
        scratch->partial = c->partial;
        scratch->partial_bytes = need;
    }
    P.partial = scratch->partial;
    P.item_counter = scratch->counters;
    P.ray_counter = reinterpret_cast<unsigned long long*>(scratch->counters + 2);
    HIP_TRY(hipMemsetAsync(scratch->counters, 0, COUNTER_BYTES, st));

    const bool pk = M == RTX_TRAVERSAL_FAST && scene->packed;  // (a scene with brackets has no packed array)
    const bool wide = pk && scene->wide && c->dev.wnodes != nullptr;
    const size_t ent_bytes = pk ? (size_t)scene->pk_slots * 16 : (size_t)scene->n_ents[M] * 32;
    // (a packed array has float boxes exactly when it was sized for LDS, and the kernel variants read it accordingly)
    const bool in_lds = pk ? !scene->pk_half : ent_bytes <= K.lds_ents_max;
    // one instantiation per (traversal, entries in LDS, brackets, uv / noise textures): features a scene does not use
    // cost registers, and registers cost resident waves
    static void (*const kernels[16])(RenderParams) = {
        render_kernel<false, false, false, false>, render_kernel<false, false, false, true>, render_kernel<false, false, true, false>,
        render_kernel<false, false, true, true>,   render_kernel<false, true, false, false>, render_kernel<false, true, false, true>,
        render_kernel<false, true, true, false>,   render_kernel<false, true, true, true>,   render_kernel<true, false, false, false>,
        render_kernel<true, false, false, true>,   render_kernel<true, false, true, false>,  render_kernel<true, false, true, true>,
        render_kernel<true, true, false, false>,   render_kernel<true, true, false, true>,   render_kernel<true, true, true, false>,
        render_kernel<true, true, true, true>};
    static void (*const packed_kernels[4])(RenderParams) = {
        render_kernel<false, false, false, false, true>, render_kernel<false, false, false, true, true>,
        render_kernel<false, true, false, false, true>,  render_kernel<false, true, false, true, true>};
    static void (*const wide_kernels[2])(RenderParams) = {render_kernel<false, false, false, false, true, true>, render_kernel<false, false, false, true, true, true>};
    P.box_run_min = (64u + K.sched_k) / (K.sched_k + 1u);
    P.xcd_parts = K.xcd_items == 0u ? 0u : (P.tile_world == 1u ? 1u : 2u);
    P.start_off = M == RTX_TRAVERSAL_FAST && !wide ? walk_start_off(scene, cam->center, pk) : 0u;
    if (wide) P.sched_k = K.sched_k_wide;  // (a wide step serves nodes and primitives alike: it pays to leave it for other stages sooner)
    static void (*const frameless_kernels[4])(RenderParams) = {
        render_kernel<false, false, true, false, false, false, true>, render_kernel<false, false, true, true, false, false, true>,
        render_kernel<false, true, true, false, false, false, true>,  render_kernel<false, true, true, true, false, false, true>};
    // scenes with media inside other media's boundaries: the bracket variants that park the enclosing media's state
    static void (*const nest_kernels[8])(RenderParams) = {
        render_kernel<false, false, true, false, false, false, false, true>, render_kernel<false, false, true, true, false, false, false, true>,
        render_kernel<false, true, true, false, false, false, false, true>,  render_kernel<false, true, true, true, false, false, false, true>,
        render_kernel<true, false, true, false, false, false, false, true>,  render_kernel<true, false, true, true, false, false, false, true>,
        render_kernel<true, true, true, false, false, false, false, true>,   render_kernel<true, true, true, true, false, false, false, true>};
    static void (*const nest_frameless_kernels[4])(RenderParams) = {
        render_kernel<false, false, true, false, false, false, true, true>, render_kernel<false, false, true, true, false, false, true, true>,
        render_kernel<false, true, true, false, false, false, true, true>,  render_kernel<false, true, true, true, false, false, true, true>};
    const bool nest = scene->media_extras;
    const bool frameless = M == RTX_TRAVERSAL_FAST && scene->has_brackets && !scene->fast_has_frames;
    void (*kern)(RenderParams) = nest && frameless ? nest_frameless_kernels[(in_lds ? 2 : 0) | (scene->has_tex ? 1 : 0)]
                                 : nest ? nest_kernels[(M == RTX_TRAVERSAL_STRICT ? 4 : 0) | (in_lds ? 2 : 0) | (scene->has_tex ? 1 : 0)]
                                 : wide ? wide_kernels[scene->has_tex ? 1 : 0]
                                 : pk ? packed_kernels[(in_lds ? 2 : 0) | (scene->has_tex ? 1 : 0)]
                                 : frameless ? frameless_kernels[(in_lds ? 2 : 0) | (scene->has_tex ? 1 : 0)]
                                    : kernels[(M == RTX_TRAVERSAL_STRICT ? 8 : 0) | (in_lds ? 4 : 0) | (scene->has_brackets ? 2 : 0) | (scene->has_tex ? 1 : 0)];

    // noise textures read their tables 336 times per value: table set 0 goes to LDS (6 KB per block)
    const size_t perlin_lds = scene->has_tex && scene->n_perlins != 0 && K.perlin_lds != 0u ? 1536 * sizeof(float) : 0;
    P.perlin_in_lds = perlin_lds != 0 ? 1u : 0u;
    // unpacked arrays in LDS: up to 16 KB of quad / triangle records follow them
    const size_t planar_lds = (!pk && in_lds && scene->n_planar != 0 && (size_t)scene->n_planar * 80 <= K.planar_lds_max) ? (size_t)scene->n_planar * 80 : 0;
    P.planar_in_lds = planar_lds != 0 ? scene->n_planar : 0u;
    // Block shape: a block is 4, 8, 12 or 16 waves (1 to 4 per SIMD).  The variant's register count says how many waves a
    // SIMD can hold (512 VGPRs each, allocated in eights); take the shape that makes the most waves resident within
    // the 160 KB of LDS, and on a tie the larger block (fewer LDS copies of the entries).
    uint32_t vgprs = 0;
    if ((rc = kernel_regs((const void*)kern, &vgprs)) != RTX_OK) return rc;
    const uint32_t waves_per_simd = std::max(1u, std::min(8u, 512u / ((vgprs + 7u) & ~7u)));
    uint32_t wpb = 4, blocks_per_cu = 1, best_resident = 0;
    const size_t slot_bytes = SLOT_FLOATS * sizeof(float);
    const size_t stack_bytes = wide ? (size_t)std::max(1u, scene->w_levels) * 64 * 4 : 0;  // per wave: one dword per lane and level
    for (uint32_t b = 1; b <= (uint32_t)MAX_WAVES_PER_BLOCK / 4; b++) {
        const size_t l = (in_lds ? ent_bytes : 0) + (size_t)(4 * b) * (2 * slot_bytes + stack_bytes) + perlin_lds + planar_lds;  // with two item slots per wave
        const uint32_t fit = (uint32_t)std::min<size_t>(waves_per_simd / b, (160 * 1024) / (l + 2048));
        if (fit * b >= best_resident && fit > 0) {
            best_resident = fit * b;
            wpb = 4 * b;
            blocks_per_cu = fit;
        }
    }
    if (best_resident == 0) return fail(RTX_ERR_DEVICE, "internal: no block shape fits");
    if (K.waves_per_block != 0u) wpb = std::min<uint32_t>(MAX_WAVES_PER_BLOCK, std::max<uint32_t>(4u, K.waves_per_block & ~3u));
    if (K.blocks_per_cu != 0u) blocks_per_cu = K.blocks_per_cu;
    // items in flight per wave: as many as the LDS left over by that shape holds, two at least
    uint32_t n_slots = 2;
    while (n_slots < K.item_slots &&
           ((in_lds ? ent_bytes : 0) + (size_t)wpb * ((n_slots + 1) * slot_bytes + stack_bytes) + perlin_lds + planar_lds + 2048) * blocks_per_cu <= 160 * 1024)
        n_slots++;
    if (K.item_slots < 2u) n_slots = K.item_slots;
    if ((opts->flags & RTX_RENDER_REPRODUCIBLE) != 0u) n_slots = 1;  // one item at a time: a wave's adds depend on the item alone
    P.n_slots = n_slots;
    // (accumulators and tables end on a 16-byte boundary: 784 bytes per slot, 6144 of Perlin tables, 80 per planar record)
    P.wide_stack_u4 = (uint32_t)(((in_lds ? ent_bytes : 0) + (size_t)wpb * n_slots * slot_bytes + perlin_lds + planar_lds) / 16);
    const size_t lds = (in_lds ? ent_bytes : 0) + (size_t)wpb * (n_slots * slot_bytes + stack_bytes) + perlin_lds + planar_lds;
    if (lds > 160 * 1024) return fail(RTX_ERR_DEVICE, "internal: the launch's LDS does not fit");
    dim3 block(64 * wpb), grid((unsigned)std::min<uint64_t>((uint64_t)c->n_cus * blocks_per_cu, ((uint64_t)P.n_items + wpb - 1) / wpb));
    if (K.debug)
        std::fprintf(stderr, "[rtx] render_kernel: %u VGPRs, %u waves/block, %u blocks/CU, %zu B LDS/block, %u item slots/wave, %u batches of %u spp, %u items\n", vgprs, wpb,
                     blocks_per_cu, lds, n_slots, P.n_batches, P.batch, P.n_items);

    if ((rc = kernel_allow_lds((const void*)kern, c->device, lds)) != RTX_OK) return rc;
    if (c->dev.med_stack && (uint64_t)grid.x * block.x > c->dev.med_lanes) return fail(RTX_ERR_DEVICE, "internal: more lanes than the scene's nested media have parking room for");

    HIP_TRY(hipEventRecord(c->ev0, st));
    hipLaunchKernelGGL(kern, grid, block, lds, st, P);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev1, st));
    {
        dim3 rb(256), rg((unsigned)((npix3 + 255) / 256));
        hipLaunchKernelGGL(reduce_kernel, rg, rb, 0, st, (const float*)P.partial, P.n_batches, npix3, accum_dev, cam->width, P.tiles_x, P.tile_rank,
                           P.tile_world);
        HIP_TRY(hipGetLastError());
    }
    if (stats) {
        HIP_TRY(hipStreamSynchronize(st));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        unsigned long long cnt[3] = {0, 0, 0};  // rays, box tests, primitive tests
        HIP_TRY(hipMemcpy(cnt, scratch->counters + 2, sizeof(cnt), hipMemcpyDeviceToHost));
        uint64_t pixels = 0;  // of the tiles this call owns (edge tiles are ragged)
        for (uint32_t t = P.tile_rank; t < n_tiles; t += P.tile_world) {
            const uint32_t tx = (t % P.tiles_x) * TILE, ty = (t / P.tiles_x) * TILE;
            pixels += (uint64_t)std::min<uint32_t>(TILE, cam->width - tx) * std::min<uint32_t>(TILE, cam->height - ty);
        }
        stats->samples = pixels * spp;
        stats->rays = cnt[0];
        stats->box_tests = cnt[1];
        stats->prim_tests = cnt[2];
        stats->kernel_ms = ms;
        stats->n_launches = 1;
        stats->flags = (in_lds ? 1u : 0u) | (pk ? 2u : 0u) | (wide ? 4u : 0u);
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
#ifdef RTX_STAGE_COUNTERS
        unsigned long long sc[12];
        HIP_TRY(hipMemcpy(sc, reinterpret_cast<unsigned long long*>(scratch->counters + 2) + 6, 96, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[stage clocks] box %.1f%% primitive %.1f%% service %.1f%% scheduling and item turnover %.1f%% of the waves' time\n", 100.0 * sc[8] / sc[11],
                     100.0 * sc[9] / sc[11], 100.0 * sc[10] / sc[11], 100.0 * (sc[11] - sc[8] - sc[9] - sc[10]) / sc[11]);
        double r = (double)cnt[0];
        std::fprintf(stderr, "[stages] rays %.3g | box steps %.3g lanes/step %.1f tests/ray %.2f | prim steps %.3g lanes/step %.1f tests/ray %.2f | service %.3g shade lanes/pass %.1f gen lanes/pass %.1f | sched passes %.3g\n", r,
                     (double)sc[0], (double)sc[1] / sc[0], sc[1] / r, (double)sc[2], (double)sc[3] / sc[2], sc[3] / r, (double)sc[4], (double)sc[5] / sc[4], (double)sc[6] / sc[4], (double)sc[7]);
#endif
    }
    return RTX_OK;
}

extern "C" int rtx_render_accumulate(rtx_scene_t* scene, const rtx_camera_t* cam, const rtx_render_opts_t* opts, float* accum_dev, void* stream,
                                     rtx_stats_t* stats) {
    if (!scene) return fail(RTX_ERR_INVALID, "null argument");
    DeviceCopy* c = &scene->primary;
    Scratch scratch{c->partial, c->partial_bytes, c->counters, true};
    return render_accumulate_on(scene, c, cam, opts, accum_dev, (hipStream_t)stream, stats, &scratch, true);
}

// ------------------------------------------------------------------------------------------------ fixed scratch
// The library-owned device memory of rtx_render_accumulate_static: work counters, then the per-batch partial images.
// One copy of the array exists per GPU the code object is loaded on.  In host code the array's NAME is only a handle for
// hipGetSymbolAddress: the device address it returns (per device) is what kernels and memsets are given.
static __device__ __align__(256) unsigned char scratch[COUNTER_BYTES + (size_t)RTX_STATIC_SCRATCH_BYTES];
constexpr int MAX_SCRATCH_DEVICES = 64;
static void* g_scratch_addr[MAX_SCRATCH_DEVICES];  // device address of `scratch` on device i, resolved on first use there
static std::mutex g_scratch_mutex;

extern "C" int rtx_render_accumulate_static(rtx_scene_t* scene, const rtx_camera_t* cam, const rtx_render_opts_t* opts, float* accum_dev, void* stream,
                                            rtx_stats_t* stats) {
    if (!scene) return fail(RTX_ERR_INVALID, "null argument");
    DeviceCopy* c = &scene->primary;
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != c->device) return fail(RTX_ERR_INVALID, "scene lives on another device than the current one");
    if (cur < 0 || cur >= MAX_SCRATCH_DEVICES) return fail(RTX_ERR_UNSUPPORTED, "device ordinal beyond the fixed scratch table");
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_addr[cur]) {
            void* p = nullptr;
            HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));  // the current device's copy
            if (!p) return fail(RTX_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the scratch array");
            g_scratch_addr[cur] = p;
        }
        base = g_scratch_addr[cur];
    }
    Scratch fixed{reinterpret_cast<float*>(static_cast<unsigned char*>(base) + COUNTER_BYTES), (size_t)RTX_STATIC_SCRATCH_BYTES,
                  static_cast<uint32_t*>(base), false};
    return render_accumulate_on(scene, c, cam, opts, accum_dev, (hipStream_t)stream, stats, &fixed, false);
}

extern "C" int rtx_resolve(const float* accum_dev, uint32_t width, uint32_t height, float scale, uint8_t* rgb8_dev, void* stream) {
    if (!accum_dev || !rgb8_dev) return fail(RTX_ERR_INVALID, "null argument");
    if (!width || !height) return fail(RTX_ERR_INVALID, "zero image size");
    size_t n = (size_t)width * height * 3;
    dim3 b(256), g((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(resolve_kernel, g, b, 0, (hipStream_t)stream, accum_dev, n, scale, rgb8_dev);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

template <class T>
static int ensure_buffer(T** buf, size_t* have, size_t need) {
    if (*have >= need) return RTX_OK;
    if (*buf) hipFree(*buf);
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc((void**)buf, need));
    *have = need;
    return RTX_OK;
}

extern "C" int rtx_render(rtx_scene_t* scene, const rtx_camera_t* cam, const rtx_render_opts_t* opts, uint8_t* rgb8_out, float* accum_out,
                          rtx_stats_t* stats) {
    if (!scene || !opts || !rgb8_out) return fail(RTX_ERR_INVALID, "null argument");
    if (opts->n_gpus > 1u) return rtx_render_multi(scene, cam, opts, nullptr, opts->n_gpus, opts->split, rgb8_out, accum_out, stats);
    int rc = check_camera(cam);
    if (rc != RTX_OK) return rc;
    auto t_start = std::chrono::steady_clock::now();
    DeviceCopy* c = &scene->primary;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)cam->width * cam->height * 3;
    // the frame's device buffers stay with the scene: a small frame costs less to render than two allocations
    if ((rc = ensure_buffer(&c->accum, &c->accum_bytes, n * sizeof(float))) != RTX_OK) return rc;
    if ((rc = ensure_buffer(&c->rgb, &c->rgb_bytes, n)) != RTX_OK) return rc;
    rtx_render_opts_t o = *opts;
    if (o.sample_begin == 0 && o.sample_end == 0) o.sample_end = cam->samples_per_pixel;
    rtx_stats_t local;
    std::memset(&local, 0, sizeof(local));
    HIP_TRY(hipMemsetAsync(c->accum, 0, n * sizeof(float), nullptr));
    rc = rtx_render_accumulate(scene, cam, &o, c->accum, nullptr, stats ? &local : nullptr);  // (reading the timers costs a synchronise)
    if (rc == RTX_OK) rc = rtx_resolve(c->accum, cam->width, cam->height, cam->pixel_samples_scale, c->rgb, nullptr);
    if (rc != RTX_OK) return rc;
    HIP_TRY(hipMemcpy(rgb8_out, c->rgb, n, hipMemcpyDeviceToHost));  // (waits for the null stream)
    if (accum_out) HIP_TRY(hipMemcpy(accum_out, c->accum, n * sizeof(float), hipMemcpyDeviceToHost));
    if (stats) {
        *stats = local;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
    return RTX_OK;
}

// ================================================================================================ several devices
// A second copy of the primary's device data on the current device (device-to-device copies), pointers re-based.
static int replicate(const rtx_scene* scene, DeviceCopy* r) {
    const DeviceCopy& p = scene->primary;
    HIP_TRY(hipGetDevice(&r->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, r->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(RTX_ERR_NO_DEVICE, "device is not gfx950");
    r->n_cus = prop.multiProcessorCount;
    if (r->device != p.device) {
        // direct access to the primary's device makes the copies below (and the share's result on its way back) go over the
        // GPU fabric; without it the runtime stages them through host memory: slower, still correct -- and reported
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, r->device, p.device);
        if (e == hipSuccess && can) {
            e = hipDeviceEnablePeerAccess(p.device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
        }
        r->peer_direct = e == hipSuccess && can;
        if (!r->peer_direct)
            r->peer_note = "device " + std::to_string(r->device) + " has no direct access to device " + std::to_string(p.device) +
                           (e != hipSuccess ? std::string(" (") + hipGetErrorString(e) + ")" : std::string()) + ": copies are staged through the host";
        (void)hipGetLastError();
    }
    for (const auto& a : p.allocs) {
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, a.second));
        r->allocs.emplace_back(q, a.second);
        HIP_TRY(hipMemcpyPeer(q, r->device, a.first, p.device, a.second));
    }
    auto rebase = [&](const void* old) -> void* {
        for (size_t i = 0; i < p.allocs.size(); i++)
            if (p.allocs[i].first == old) return r->allocs[i].first;
        return nullptr;
    };
    r->dev = p.dev;
#define RTX_REBASE(field)                                                                        \
    do {                                                                                         \
        void* q_ = rebase((const void*)p.dev.field);                                             \
        if (!q_) return fail(RTX_ERR_DEVICE, "internal: device pointer outside the scene's allocations"); \
        r->dev.field = static_cast<std::remove_reference_t<decltype(r->dev.field)>>(q_);         \
    } while (0)
    for (int m = 0; m < 2; m++) {
        RTX_REBASE(ents[m]);
        RTX_REBASE(rank[m]);
        RTX_REBASE(shade[m]);
    }
    RTX_REBASE(sph);
    RTX_REBASE(sph_id);
    RTX_REBASE(planar);
    RTX_REBASE(planar_id);
    RTX_REBASE(mats);
    RTX_REBASE(texs);
    RTX_REBASE(images);
    RTX_REBASE(perlin);
    RTX_REBASE(mat_recs);
    if (p.dev.pk) {
        RTX_REBASE(pk);
        RTX_REBASE(pk_rank);
        if (p.dev.pk_shade) RTX_REBASE(pk_shade);
    }
    if (p.dev.med_stack) RTX_REBASE(med_stack);
    if (p.dev.wnodes) {
        RTX_REBASE(wnodes);
        RTX_REBASE(wprims);
        RTX_REBASE(wrank);
    }
#undef RTX_REBASE
    if (scene->n_images != 0) {  // the image table itself holds device pointers
        std::vector<DevImage> im(scene->n_images);
        HIP_TRY(hipMemcpy(im.data(), r->dev.images, im.size() * sizeof(DevImage), hipMemcpyDeviceToHost));
        for (DevImage& x : im) {
            void* q = rebase(x.rgb);
            if (!q) return fail(RTX_ERR_DEVICE, "internal: image raster outside the scene's allocations");
            x.rgb = reinterpret_cast<const uint8_t*>(q);
        }
        HIP_TRY(hipMemcpy(const_cast<DevImage*>(r->dev.images), im.data(), im.size() * sizeof(DevImage), hipMemcpyHostToDevice));
    }
    // the copied fast array carries the order the primary's had at this moment
    std::memcpy(r->view, p.view, sizeof(r->view));
    r->view_grow = p.view_grow;
    r->view_valid = p.view_valid;
    r->wide_grow = p.wide_grow;
    HIP_TRY(hipMalloc((void**)&r->counters, COUNTER_BYTES));
    HIP_TRY(hipEventCreate(&r->ev0));
    HIP_TRY(hipEventCreate(&r->ev1));
    return RTX_OK;
}

// Brings a replica's fast traversal data to the order / box sizes the primary's have now: device-to-device copies of the
// arrays order_fast_for_view rewrites, instead of a second host pass and upload per device.
static int refresh_replica_order(const rtx_scene* scene, DeviceCopy* r) {
    const DeviceCopy& p = scene->primary;
    if (r == &p) return RTX_OK;
    if (r->view_valid == p.view_valid && std::memcmp(r->view, p.view, sizeof(r->view)) == 0 && r->view_grow == p.view_grow && r->wide_grow == p.wide_grow)
        return RTX_OK;
    auto copy = [&](const void* dst, const void* src, size_t bytes) {
        return bytes == 0 ? hipSuccess : hipMemcpyPeer(const_cast<void*>(dst), r->device, src, p.device, bytes);
    };
    const int F = RTX_TRAVERSAL_FAST;
    if (scene->wide) {
        HIP_TRY(copy(r->dev.wnodes, p.dev.wnodes, (size_t)scene->w_nodes * rtx_accel::WIDE_NODE_WORDS * 4));
    } else {
        if (scene->packed) {
            HIP_TRY(copy(r->dev.pk, p.dev.pk, (size_t)scene->pk_slots * 16));
            HIP_TRY(copy(r->dev.pk_rank, p.dev.pk_rank, (size_t)scene->pk_slots * 4));
            if (p.dev.pk_shade) HIP_TRY(copy(r->dev.pk_shade, p.dev.pk_shade, (size_t)scene->pk_slots * 32));
        }
        HIP_TRY(copy(r->dev.ents[F], p.dev.ents[F], (size_t)scene->n_ents[F] * 32));
        HIP_TRY(copy(r->dev.rank[F], p.dev.rank[F], (size_t)scene->n_ents[F] * 4));
        HIP_TRY(copy(r->dev.shade[F], p.dev.shade[F], (size_t)scene->n_ents[F] * 32));
    }
    std::memcpy(r->view, p.view, sizeof(r->view));
    r->view_grow = p.view_grow;
    r->view_valid = p.view_valid;
    r->wide_grow = p.wide_grow;
    return RTX_OK;
}

__global__ __launch_bounds__(256) void add_slots_kernel(float* __restrict__ dst, const float* __restrict__ slots, uint32_t n_slots, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = dst[i];
    for (uint32_t k = 0; k < n_slots; k++) s += slots[(size_t)k * n + i];  // device order: the sum does not depend on arrival order
    dst[i] = s;
}

// Keeps, per pixel, the bytes of the one share that owns the pixel's tile.
__global__ __launch_bounds__(256) void merge_tiles_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ slots, uint32_t n_shares, size_t n,
                                                          uint32_t width, uint32_t tiles_x) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pix = (uint32_t)(i / 3), px = pix % width, py = pix / width;
    const uint32_t owner = ((py / TILE) * tiles_x + px / TILE) % n_shares;
    if (owner != 0u) dst[i] = slots[(size_t)(owner - 1u) * n + i];
}

extern "C" int rtx_render_multi(rtx_scene_t* scene, const rtx_camera_t* cam, const rtx_render_opts_t* opts, const int* devices, uint32_t n_devices,
                                uint32_t split, uint8_t* rgb8_out, float* accum_out, rtx_stats_t* stats) {
    if (!scene || !opts || !rgb8_out) return fail(RTX_ERR_INVALID, "null argument");
    int rc = check_camera(cam);
    if (rc != RTX_OK) return rc;
    if (n_devices == 0 || n_devices > 64) return fail(RTX_ERR_INVALID, "n_devices must be 1..64");
    if (split > RTX_SPLIT_TILES) return fail(RTX_ERR_INVALID, "unknown split");
    if (opts->tile_world > 1u) return fail(RTX_ERR_INVALID, "tile_rank / tile_world belong to rtx_render_accumulate; rtx_render_multi assigns the shares itself");
    auto t_start = std::chrono::steady_clock::now();
    int n_visible = 0;
    HIP_TRY(hipGetDeviceCount(&n_visible));
    std::vector<int> dev(n_devices);
    for (uint32_t k = 0; k < n_devices; k++) {
        dev[k] = devices ? devices[k] : (int)k;
        if (dev[k] < 0 || dev[k] >= n_visible) return fail(RTX_ERR_NO_DEVICE, "device ordinal out of range: fewer GPUs are visible than the call names");
    }
    uint32_t sb = opts->sample_begin, se = opts->sample_end;
    if (sb == 0 && se == 0) se = cam->samples_per_pixel;
    if (se < sb) return fail(RTX_ERR_INVALID, "sample_end < sample_begin");
    const uint32_t spp = se - sb;
    const size_t n = (size_t)cam->width * cam->height * 3;
    const bool bytes_only = split == RTX_SPLIT_TILES && accum_out == nullptr;  // a tile's owner resolves it; only RGB8 travels

    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    struct Restore {
        int d;
        ~Restore() { hipSetDevice(d); }
    } restore{prev};

    // The fast traversal data is prepared for this camera ONCE, on the primary (host pass + upload), before any replica is
    // made or refreshed: replicas then take it over device to device, and no share prepares anything at launch.
    uint32_t M = opts->traversal;
    if (M > RTX_TRAVERSAL_STRICT) return fail(RTX_ERR_INVALID, "unknown traversal mode");
    if (M == RTX_TRAVERSAL_FAST) {
        const ViewNeed need = view_need(scene, cam);
        if (!need.strict_only && (knobs().view_order != 0u || need.grow > 0.0f)) {
            HIP_TRY(hipSetDevice(scene->primary.device));
            if ((rc = order_fast_for_view(scene, &scene->primary, cam->center, need.grow)) != RTX_OK) return rc;
        }
    }
    // share k runs on copy k: the primary when it lives on the share's device and no earlier share took it
    if (scene->replicas.size() < n_devices) scene->replicas.resize(n_devices, nullptr);
    std::vector<DeviceCopy*> copy(n_devices, nullptr);
    bool primary_taken = false;
    std::string notes;
    for (uint32_t k = 0; k < n_devices; k++) {
        HIP_TRY(hipSetDevice(dev[k]));
        if (!primary_taken && dev[k] == scene->primary.device) {
            copy[k] = &scene->primary;
            primary_taken = true;
        } else {
            DeviceCopy*& r = scene->replicas[k];
            if (r && r->device != dev[k]) {
                release_copy(r);
                delete r;
                r = nullptr;
                HIP_TRY(hipSetDevice(dev[k]));
            }
            if (!r) {
                r = new DeviceCopy();
                if ((rc = replicate(scene, r)) != RTX_OK) {
                    std::string keep = g_err;
                    release_copy(r);
                    delete r;
                    r = nullptr;
                    g_err = keep;
                    return rc;
                }
            }
            if ((rc = refresh_replica_order(scene, r)) != RTX_OK) return rc;
            if (!r->peer_direct && notes.find(r->peer_note) == std::string::npos) notes += (notes.empty() ? "" : "; ") + r->peer_note;
            copy[k] = r;
        }
        DeviceCopy* c = copy[k];
        if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        if (!c->done) HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
        if ((rc = ensure_buffer(&c->accum, &c->accum_bytes, n * sizeof(float))) != RTX_OK) return rc;
        if ((rc = ensure_buffer(&c->rgb, &c->rgb_bytes, n)) != RTX_OK) return rc;
    }
    DeviceCopy* root = copy[0];
    HIP_TRY(hipSetDevice(dev[0]));
    const size_t slot_bytes = bytes_only ? n : n * sizeof(float);
    if (n_devices > 1 && (rc = ensure_buffer((char**)&root->staging, &root->staging_bytes, slot_bytes * (n_devices - 1))) != RTX_OK) return rc;

    // From the first launch on, a failure must not return while earlier shares' kernels and copies are still in flight: they
    // write into buffers the scene owns and the next call reuses.
    uint32_t launched = 0;
    auto bail = [&](int code) {
        const std::string keep = g_err;
        for (uint32_t k = 0; k < launched && k < n_devices; k++) {
            if (hipSetDevice(dev[k]) == hipSuccess) (void)hipStreamSynchronize(copy[k]->stream);
        }
        (void)hipGetLastError();
        g_err = keep;
        return code;
    };
#define MULTI_TRY(expr)                                                                                              \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess) return bail(fail(RTX_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)

    // launch every share on its own device and stream (nothing here waits for a device), then push its result into its
    // slot on the first device
    for (uint32_t k = 0; k < n_devices; k++) {
        DeviceCopy* c = copy[k];
        MULTI_TRY(hipSetDevice(dev[k]));
        launched = k + 1;
        rtx_render_opts_t o = *opts;
        o.n_gpus = 0;
        if (split == RTX_SPLIT_SAMPLES) {
            o.sample_begin = sb + (uint32_t)(((uint64_t)k * spp) / n_devices);
            o.sample_end = sb + (uint32_t)(((uint64_t)(k + 1) * spp) / n_devices);
            o.tile_rank = 0;
            o.tile_world = 1;
        } else {
            o.sample_begin = sb;
            o.sample_end = se;
            o.tile_rank = k;
            o.tile_world = n_devices;
        }
        MULTI_TRY(hipMemsetAsync(c->accum, 0, n * sizeof(float), c->stream));
        MULTI_TRY(hipMemsetAsync(c->counters, 0, COUNTER_BYTES, c->stream));  // a share with nothing to render launches nothing
        Scratch scratch{c->partial, c->partial_bytes, c->counters, true};
        if ((rc = render_accumulate_on(scene, c, cam, &o, c->accum, c->stream, nullptr, &scratch, false)) != RTX_OK) return bail(rc);
        if (bytes_only) {
            if ((rc = rtx_resolve(c->accum, cam->width, cam->height, cam->pixel_samples_scale, c->rgb, c->stream)) != RTX_OK) return bail(rc);
            if (k > 0) MULTI_TRY(hipMemcpyPeerAsync((char*)root->staging + (size_t)(k - 1) * slot_bytes, dev[0], c->rgb, dev[k], slot_bytes, c->stream));
        } else if (k > 0) {
            MULTI_TRY(hipMemcpyPeerAsync((char*)root->staging + (size_t)(k - 1) * slot_bytes, dev[0], c->accum, dev[k], slot_bytes, c->stream));
        }
        MULTI_TRY(hipEventRecord(c->done, c->stream));
    }
    // the first device waits for every slot, combines them in device order and resolves
    MULTI_TRY(hipSetDevice(dev[0]));
    for (uint32_t k = 1; k < n_devices; k++) MULTI_TRY(hipStreamWaitEvent(root->stream, copy[k]->done, 0));
    {
        dim3 b(256), g((unsigned)((n + 255) / 256));
        if (bytes_only) {
            if (n_devices > 1) {
                hipLaunchKernelGGL(merge_tiles_kernel, g, b, 0, root->stream, root->rgb, (const uint8_t*)root->staging, n_devices, n, cam->width,
                                   (cam->width + TILE - 1) / TILE);
                MULTI_TRY(hipGetLastError());
            }
        } else {
            if (n_devices > 1) {
                hipLaunchKernelGGL(add_slots_kernel, g, b, 0, root->stream, root->accum, (const float*)root->staging, n_devices - 1, n);
                MULTI_TRY(hipGetLastError());
            }
            if ((rc = rtx_resolve(root->accum, cam->width, cam->height, cam->pixel_samples_scale, root->rgb, root->stream)) != RTX_OK) return bail(rc);
        }
    }
    MULTI_TRY(hipMemcpyAsync(rgb8_out, root->rgb, n, hipMemcpyDeviceToHost, root->stream));
    if (accum_out) MULTI_TRY(hipMemcpyAsync(accum_out, root->accum, n * sizeof(float), hipMemcpyDeviceToHost, root->stream));
    MULTI_TRY(hipStreamSynchronize(root->stream));
#undef MULTI_TRY
    // (a share that cannot reach the first device directly still delivered its result, through the host: say so)
    if (!notes.empty()) g_err = "note: " + notes;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (uint32_t k = 0; k < n_devices; k++) {
            DeviceCopy* c = copy[k];
            HIP_TRY(hipSetDevice(dev[k]));
            HIP_TRY(hipStreamSynchronize(c->stream));
            unsigned long long cnt[3] = {0, 0, 0};
            HIP_TRY(hipMemcpy(cnt, c->counters + 2, sizeof(cnt), hipMemcpyDeviceToHost));
            float ms = 0;
            if (cnt[0] != 0 && hipEventElapsedTime(&ms, c->ev0, c->ev1) != hipSuccess) {
                (void)hipGetLastError();
                ms = 0;
            }
            stats->rays += cnt[0];
            stats->box_tests += cnt[1];
            stats->prim_tests += cnt[2];
            stats->kernel_ms = std::max(stats->kernel_ms, (double)ms);
            stats->n_launches += 1;
        }
        stats->samples = (uint64_t)cam->width * cam->height * spp;
        stats->flags = notes.empty() ? 0u : 8u;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
    return RTX_OK;
}
