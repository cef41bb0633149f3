// C handle API over the C++ host layer, for callers without a C++ compiler (the Python tests and bench).
// One function per constructor of the reference (names after the Rust items they stand for), plus
// scene flattening / upload and Camera::render.  Errors come back as null handles or negative codes;
// rth_last_error() holds the text.
#include <cstdio>

#include "ray_tracing.hpp"

using namespace ray_tracing;

struct rth_tex { TextureP p; };
struct rth_mat { MaterialP p; };
struct rth_mesh { std::shared_ptr<Mesh> p; };
struct rth_camera { std::shared_ptr<Camera> p; };
struct rth_scene {
    SceneBuilder b;
    rtx_scene_desc_t desc;
    rtx_scene_t* dev = nullptr;
    std::shared_ptr<Mesh> keep;  // the flat arrays borrow image / perlin storage from the objects
};
struct rth_camera_cfg_t {
    uint32_t width, height, samples_per_pixel, max_depth;
    float vfov;
    float lookfrom[3], lookat[3], vup[3];
    float defocus_angle, focus_dist;
    float background[3];
};

static thread_local std::string g_err;
static inline Vec3 v3(const float* f) { return Vec3(f[0], f[1], f[2]); }

template <class F>
static auto guard(F f) -> decltype(f()) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
        return decltype(f())();
    }
}

extern "C" {

const char* rth_last_error() { return g_err.c_str(); }

rth_tex* rth_tex_solid(float r, float g, float b) { return new rth_tex{SolidTexture::rgb(r, g, b)}; }
rth_tex* rth_tex_checker(float scale, rth_tex* even, rth_tex* odd) { return new rth_tex{std::make_shared<CheckerTexture>(scale, even->p, odd->p)}; }
rth_tex* rth_tex_image(const uint8_t* rgb, uint32_t w, uint32_t h) {
    if (!rgb || !w || !h) { g_err = "ImageTexture: empty raster"; return nullptr; }
    return new rth_tex{std::make_shared<ImageTexture>(rgb, w, h)};
}
rth_tex* rth_tex_noise(float scale, const float* values, const uint32_t* px, const uint32_t* py, const uint32_t* pz) {
    return new rth_tex{std::make_shared<NoiseTexture>(scale, values, px, py, pz)};
}
void rth_tex_free(rth_tex* t) { delete t; }

rth_mat* rth_mat_lambertian(rth_tex* t) { return new rth_mat{std::make_shared<Lambertian>(t->p)}; }
rth_mat* rth_mat_metal(rth_tex* t, float roughness) { return new rth_mat{std::make_shared<Metal>(t->p, roughness)}; }
rth_mat* rth_mat_dielectric(float ri) { return new rth_mat{std::make_shared<Dielectric>(ri)}; }
rth_mat* rth_mat_diffuse_light(rth_tex* t) { return new rth_mat{std::make_shared<DiffuseLight>(t->p)}; }
rth_mat* rth_mat_isotropic(rth_tex* t) { return new rth_mat{std::make_shared<Isotropic>(t->p)}; }
void rth_mat_free(rth_mat* m) { delete m; }

rth_mesh* rth_sphere_stationary(const float* c, float r, rth_mat* m, uint32_t prim_id) {
    return guard([&]() -> rth_mesh* { auto s = Sphere::stationary(v3(c), r, m->p); s->prim_id = prim_id; return new rth_mesh{s}; });
}
rth_mesh* rth_sphere_moving(const float* from, const float* to, float r, rth_mat* m, uint32_t prim_id) {
    return guard([&]() -> rth_mesh* { auto s = Sphere::moving(v3(from), v3(to), r, m->p); s->prim_id = prim_id; return new rth_mesh{s}; });
}
rth_mesh* rth_quad(const float* q, const float* u, const float* v, rth_mat* m, uint32_t prim_id) {
    auto s = std::make_shared<Quad>(v3(q), v3(u), v3(v), m->p);
    s->prim_id = prim_id;
    return new rth_mesh{s};
}
rth_mesh* rth_triangle(const float* a, const float* b, const float* c, rth_mat* m, uint32_t prim_id) {
    auto s = std::make_shared<Triangle>(v3(a), v3(b), v3(c), m->p);
    s->prim_id = prim_id;
    return new rth_mesh{s};
}
rth_mesh* rth_cube(const float* a, const float* b, rth_mat* m, uint32_t first_prim_id) {
    return new rth_mesh{std::make_shared<Cube>(v3(a), v3(b), m->p, first_prim_id)};
}
rth_mesh* rth_translate(rth_mesh* o, const float* off) { return new rth_mesh{std::make_shared<Translate>(o->p, v3(off))}; }
rth_mesh* rth_rotate_y(rth_mesh* o, float angle) { return new rth_mesh{std::make_shared<RotateY>(o->p, angle)}; }
rth_mesh* rth_constant_medium(rth_mesh* boundary, float density, rth_tex* tex, uint32_t prim_id) {
    auto cm = std::make_shared<ConstantMedium>(boundary->p, density, tex->p);
    cm->prim_id = prim_id;
    return new rth_mesh{cm};
}
rth_mesh* rth_world_new() { return new rth_mesh{std::make_shared<World>()}; }
int rth_world_push(rth_mesh* world, rth_mesh* m) {
    World* w = dynamic_cast<World*>(world->p.get());
    if (!w) { g_err = "world_push: not a World"; return -1; }
    w->push(m->p);
    return 0;
}
int rth_world_push_trimesh(rth_mesh* world, const float* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_tris,
                           rth_mat* m, uint32_t first_prim_id) {
    World* w = dynamic_cast<World*>(world->p.get());
    if (!w) { g_err = "world_push_trimesh: not a World"; return -1; }
    for (uint32_t i = 0; i < n_tris; i++) {
        uint32_t a = indices[3 * i], b = indices[3 * i + 1], c = indices[3 * i + 2];
        if (a >= n_vertices || b >= n_vertices || c >= n_vertices) { g_err = "world_push_trimesh: index out of range"; return -1; }
        auto t = std::make_shared<Triangle>(v3(vertices + 3 * a), v3(vertices + 3 * b), v3(vertices + 3 * c), m->p);
        t->prim_id = first_prim_id + i;
        w->push(t);
    }
    return 0;
}
rth_mesh* rth_bvh_from(rth_mesh* world) {
    World* w = dynamic_cast<World*>(world->p.get());
    if (!w) { g_err = "Bvh::from: not a World"; return nullptr; }
    return guard([&]() -> rth_mesh* { return new rth_mesh{Bvh::from(*w)}; });
}
void rth_mesh_free(rth_mesh* m) { delete m; }
void rth_mesh_bbox(rth_mesh* m, float* out6) {
    const Aabb& b = m->p->bounding_box();
    out6[0] = b.x.start; out6[1] = b.x.end; out6[2] = b.y.start; out6[3] = b.y.end; out6[4] = b.z.start; out6[5] = b.z.end;
}

rth_camera* rth_camera_new(const rth_camera_cfg_t* c) {
    CameraConfig cfg;
    cfg.width = c->width; cfg.height = c->height;
    cfg.samples_per_pixel = c->samples_per_pixel; cfg.max_depth = c->max_depth;
    cfg.vfov = c->vfov;
    cfg.lookfrom = v3(c->lookfrom); cfg.lookat = v3(c->lookat); cfg.vup = v3(c->vup);
    cfg.defocus_angle = c->defocus_angle; cfg.focus_dist = c->focus_dist;
    cfg.background = v3(c->background);
    return new rth_camera{std::make_shared<Camera>(cfg)};
}
void rth_camera_free(rth_camera* c) { delete c; }
void rth_camera_ffi(rth_camera* c, rtx_camera_t* out) { *out = c->p->to_ffi(); }

// Flatten (host only, no GPU needed).
rth_scene* rth_scene_flatten(rth_mesh* root) {
    return guard([&]() -> rth_scene* {
        auto s = new rth_scene();
        s->keep = root->p;
        flatten_scene(*root->p, s->b);
        s->desc = s->b.desc();
        return s;
    });
}
const rtx_scene_desc_t* rth_scene_desc(rth_scene* s) { return &s->desc; }
// Upload to the current GPU (rtx_scene_create).  Returns an rtx_status.
int rth_scene_upload(rth_scene* s) {
    if (s->dev) return RTX_OK;
    int rc = rtx_scene_create(&s->desc, &s->dev);
    if (rc != RTX_OK) g_err = rtx_last_error();
    return rc;
}
rtx_scene_t* rth_scene_device(rth_scene* s) { return s->dev; }
void rth_scene_free(rth_scene* s) {
    if (!s) return;
    if (s->dev) rtx_scene_destroy(s->dev);
    delete s;
}

// DFS (left, right) leaf prim ids of the flattened reference BVH, duplicates of len==1 nodes included:
// the order in which Bvh::hit tests primitives, i.e. the tie-break order.
static void leaf_walk(const rtx_scene_desc_t& d, rtx_ref_t r, std::vector<uint32_t>& out) {
    uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
    switch (k) {
        case RTX_REF_NODE: leaf_walk(d, d.bvh_nodes[i].left, out); leaf_walk(d, d.bvh_nodes[i].right, out); break;
        case RTX_REF_SPHERE: out.push_back(d.sph_prim_id[i]); break;
        case RTX_REF_QUAD: out.push_back(d.quad_prim_id[i]); break;
        case RTX_REF_TRI: out.push_back(d.tri_prim_id[i]); break;
        case RTX_REF_OBJECT: {
            const rtx_object_t& o = d.objects[i];
            if (o.kind == RTX_OBJ_LIST) {
                for (uint32_t c = 0; c < o.count; c++) leaf_walk(d, d.refs[o.first + c], out);
            } else if (o.kind == RTX_OBJ_MEDIUM) {
                out.push_back(o.prim_id);  // a medium reports its own id, never its boundary's
            } else {
                leaf_walk(d, o.child, out);
            }
            break;
        }
    }
}
size_t rth_scene_leaf_order(rth_scene* s, uint32_t* out, size_t cap) {
    std::vector<uint32_t> v;
    leaf_walk(s->desc, s->desc.root, v);
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}

}  // extern "C"
