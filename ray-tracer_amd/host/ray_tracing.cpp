// Host layer implementation: construction-time arithmetic of the reference's types + flattening to rtx_scene_desc_t.
// Build with -ffp-contract=off.
#include "ray_tracing.hpp"

#include <algorithm>
#include <future>

namespace ray_tracing {

static const float INF = std::numeric_limits<float>::infinity();

// ------------------------------------------------------------------------------------------- SceneBuilder
rtx_scene_desc_t SceneBuilder::desc() const {
    rtx_scene_desc_t d;
    std::memset(&d, 0, sizeof(d));
    d.n_spheres = (uint32_t)sph_mat.size();
    d.sph_cx = sph[0].data(); d.sph_cy = sph[1].data(); d.sph_cz = sph[2].data();
    d.sph_dx = sph[3].data(); d.sph_dy = sph[4].data(); d.sph_dz = sph[5].data();
    d.sph_radius = sph[6].data();
    d.sph_material = sph_mat.data();
    d.sph_prim_id = sph_id.data();
    d.n_quads = (uint32_t)quad_mat.size();
    for (int k = 0; k < 16; k++) d.quad_f[k] = quad[k].data();
    d.quad_material = quad_mat.data();
    d.quad_prim_id = quad_id.data();
    d.n_tris = (uint32_t)tri_mat.size();
    for (int k = 0; k < 9; k++) d.tri_f[k] = tri[k].data();
    d.tri_material = tri_mat.data();
    d.tri_prim_id = tri_id.data();
    d.n_objects = (uint32_t)objects.size();
    d.objects = objects.data();
    d.n_refs = (uint32_t)refs.size();
    d.refs = refs.data();
    d.n_bvh_nodes = (uint32_t)nodes.size();
    d.bvh_nodes = nodes.data();
    d.root = root;
    d.n_materials = (uint32_t)materials.size();
    d.materials = materials.data();
    d.n_textures = (uint32_t)textures.size();
    d.textures = textures.data();
    d.n_images = (uint32_t)images.size();
    d.images = images.data();
    d.n_perlins = (uint32_t)perlins.size();
    d.perlins = perlins.data();
    return d;
}

template <class F>
static uint32_t memoized(SceneBuilder& b, const void* key, F make) {
    auto it = b.memo.find(key);
    if (it != b.memo.end()) return it->second;
    uint32_t v = make();
    b.memo[key] = v;
    return v;
}

// ------------------------------------------------------------------------------------------- textures
static rtx_texture_t blank_texture(uint32_t kind) {
    rtx_texture_t t;
    std::memset(&t, 0, sizeof(t));
    t.kind = kind;
    return t;
}
uint32_t SolidTexture::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        rtx_texture_t t = blank_texture(RTX_TEX_SOLID);
        t.rgb[0] = albedo.x; t.rgb[1] = albedo.y; t.rgb[2] = albedo.z;
        b.textures.push_back(t);
        return (uint32_t)b.textures.size() - 1;
    });
}
uint32_t CheckerTexture::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        uint32_t e = even->flatten(b), o = odd->flatten(b);
        rtx_texture_t t = blank_texture(RTX_TEX_CHECKER);
        t.scale = scale; t.a = e; t.b = o;
        b.textures.push_back(t);
        return (uint32_t)b.textures.size() - 1;
    });
}
uint32_t ImageTexture::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        b.images.push_back(rtx_image_t{rgb.data(), width, height});
        rtx_texture_t t = blank_texture(RTX_TEX_IMAGE);
        t.a = (uint32_t)b.images.size() - 1;
        b.textures.push_back(t);
        return (uint32_t)b.textures.size() - 1;
    });
}
uint32_t NoiseTexture::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        b.perlins.push_back(rtx_perlin_t{values.data(), perm_x.data(), perm_y.data(), perm_z.data()});
        rtx_texture_t t = blank_texture(RTX_TEX_NOISE);
        t.scale = scale; t.a = (uint32_t)b.perlins.size() - 1;
        b.textures.push_back(t);
        return (uint32_t)b.textures.size() - 1;
    });
}

// ------------------------------------------------------------------------------------------- materials
static uint32_t push_material(SceneBuilder& b, uint32_t kind, uint32_t tex, float param) {
    b.materials.push_back(rtx_material_t{kind, tex, param, 0});
    return (uint32_t)b.materials.size() - 1;
}
uint32_t Lambertian::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] { return push_material(b, RTX_MAT_LAMBERTIAN, texture->flatten(b), 0.0f); });
}
uint32_t Metal::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] { return push_material(b, RTX_MAT_METAL, texture->flatten(b), roughness); });
}
uint32_t Dielectric::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] { return push_material(b, RTX_MAT_DIELECTRIC, 0, refraction_index); });
}
uint32_t DiffuseLight::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] { return push_material(b, RTX_MAT_DIFFUSE_LIGHT, texture->flatten(b), 0.0f); });
}
uint32_t Isotropic::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] { return push_material(b, RTX_MAT_ISOTROPIC, texture->flatten(b), 0.0f); });
}

// ------------------------------------------------------------------------------------------- primitives
std::shared_ptr<Sphere> Sphere::stationary(Vec3 center, float radius, MaterialP m) {
    if (!(radius >= 0.0f)) throw std::invalid_argument("Radius cannot be less than 0.");  // mesh.rs:148
    auto s = std::make_shared<Sphere>();
    Vec3 rvec(radius, radius, radius);
    s->initial_center = center;
    s->center_delta = Vec3(0, 0, 0);
    s->radius = radius;
    s->material = m;
    s->bbox = Aabb::from_extremes(center - rvec, center + rvec);
    return s;
}
std::shared_ptr<Sphere> Sphere::moving(Vec3 from, Vec3 to, float radius, MaterialP m) {
    if (!(radius >= 0.0f)) throw std::invalid_argument("Radius cannot be less than 0.");  // mesh.rs:162
    auto s = std::make_shared<Sphere>();
    Vec3 rvec(radius, radius, radius);
    Aabb box1 = Aabb::from_extremes(from - rvec, from + rvec);
    Aabb box2 = Aabb::from_extremes(to - rvec, to + rvec);
    s->initial_center = from;
    s->center_delta = to - from;
    s->radius = radius;
    s->material = m;
    s->bbox = box1.merge(box2);
    return s;
}
rtx_ref_t Sphere::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        uint32_t mat = material->flatten(b);
        const float f[7] = {initial_center.x, initial_center.y, initial_center.z, center_delta.x, center_delta.y, center_delta.z, radius};
        for (int k = 0; k < 7; k++) b.sph[k].push_back(f[k]);
        b.sph_mat.push_back(mat);
        b.sph_id.push_back(prim_id);
        return RTX_REF(RTX_REF_SPHERE, b.sph_mat.size() - 1);
    });
}

Quad::Quad(Vec3 translation_, Vec3 u_, Vec3 v_, MaterialP m) : translation(translation_), u(u_), v(v_), material(m) {
    Aabb d1 = Aabb::from_extremes(translation, translation + u + v);
    Aabb d2 = Aabb::from_extremes(translation + u, translation + v);
    Vec3 n = cross(u, v);
    normal = normalize(n);
    bbox = d1.merge(d2);
    d = dot(normal, translation);
    w = n / dot(n, n);
}
rtx_ref_t Quad::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        uint32_t mat = material->flatten(b);
        const float f[16] = {translation.x, translation.y, translation.z, u.x, u.y, u.z, v.x, v.y, v.z,
                             normal.x, normal.y, normal.z, w.x, w.y, w.z, d};
        for (int k = 0; k < 16; k++) b.quad[k].push_back(f[k]);
        b.quad_mat.push_back(mat);
        b.quad_id.push_back(prim_id);
        return RTX_REF(RTX_REF_QUAD, b.quad_mat.size() - 1);
    });
}

Triangle::Triangle(Vec3 a, Vec3 b_, Vec3 c, MaterialP m) : v0(a), v1(b_), v2(c), material(m) {
    Vec3 lo(std::fmin(a.x, std::fmin(b_.x, c.x)), std::fmin(a.y, std::fmin(b_.y, c.y)), std::fmin(a.z, std::fmin(b_.z, c.z)));
    Vec3 hi(std::fmax(a.x, std::fmax(b_.x, c.x)), std::fmax(a.y, std::fmax(b_.y, c.y)), std::fmax(a.z, std::fmax(b_.z, c.z)));
    bbox = Aabb::from_extremes(lo, hi);
}
rtx_ref_t Triangle::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        uint32_t mat = material->flatten(b);
        const float f[9] = {v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z};
        for (int k = 0; k < 9; k++) b.tri[k].push_back(f[k]);
        b.tri_mat.push_back(mat);
        b.tri_id.push_back(prim_id);
        return RTX_REF(RTX_REF_TRI, b.tri_mat.size() - 1);
    });
}

// ------------------------------------------------------------------------------------------- composites
static rtx_object_t blank_object(uint32_t kind) {
    rtx_object_t o;
    std::memset(&o, 0, sizeof(o));
    o.kind = kind;
    return o;
}
rtx_ref_t World::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        std::vector<rtx_ref_t> kids;
        kids.reserve(meshes.size());
        for (const MeshP& m : meshes) kids.push_back(m->flatten(b));
        rtx_object_t o = blank_object(RTX_OBJ_LIST);
        o.first = (uint32_t)b.refs.size();
        o.count = (uint32_t)kids.size();
        b.refs.insert(b.refs.end(), kids.begin(), kids.end());
        b.objects.push_back(o);
        return RTX_REF(RTX_REF_OBJECT, b.objects.size() - 1);
    });
}

Cube::Cube(Vec3 a, Vec3 b, MaterialP m, uint32_t first_prim_id) {  // mesh.rs:272-321
    Vec3 mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
    Vec3 mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
    Vec3 dx = Vec3(1, 0, 0) * (mx.x - mn.x);
    Vec3 dy = Vec3(0, 1, 0) * (mx.y - mn.y);
    Vec3 dz = Vec3(0, 0, 1) * (mx.z - mn.z);
    std::shared_ptr<Quad> q[6] = {
        std::make_shared<Quad>(Vec3(mn.x, mn.y, mx.z), dx, dy, m),   std::make_shared<Quad>(Vec3(mx.x, mn.y, mx.z), -dz, dy, m),
        std::make_shared<Quad>(Vec3(mx.x, mx.y, mn.z), -dx, -dy, m), std::make_shared<Quad>(Vec3(mn.x, mn.y, mn.z), dz, dy, m),
        std::make_shared<Quad>(Vec3(mn.x, mx.y, mx.z), dx, -dz, m),  std::make_shared<Quad>(Vec3(mn.x, mn.y, mn.z), dx, dz, m)};
    for (int i = 0; i < 6; i++) {
        q[i]->prim_id = first_prim_id + (uint32_t)i;
        world.push(q[i]);
    }
}

rtx_ref_t Translate::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        rtx_object_t o = blank_object(RTX_OBJ_TRANSLATE);
        o.child = object->flatten(b);
        o.f[0] = offset.x; o.f[1] = offset.y; o.f[2] = offset.z;
        b.objects.push_back(o);
        return RTX_REF(RTX_REF_OBJECT, b.objects.size() - 1);
    });
}

RotateY::RotateY(MeshP obj, float angle) : object(obj) {  // mesh.rs:396-431
    float radians = degrees_to_radians(angle);
    sin_theta = std::sin(radians);
    cos_theta = std::cos(radians);
    Aabb bb = obj->bounding_box();
    Vec3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                float x = (float)i * bb.x.end + (1.0f - (float)i) * bb.x.start;
                float y = (float)j * bb.y.end + (1.0f - (float)j) * bb.y.start;
                float z = (float)k * bb.z.end + (1.0f - (float)k) * bb.z.start;
                float newx = cos_theta * x + sin_theta * z;
                float newz = -sin_theta * x + cos_theta * z;
                Vec3 tester(newx, y, newz);
                for (int c = 0; c < 3; c++) {
                    mn[c] = std::fmin(mn[c], tester[c]);
                    mx[c] = std::fmax(mx[c], tester[c]);
                }
            }
    bbox = Aabb::from_extremes(mn, mx);
}
rtx_ref_t RotateY::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        rtx_object_t o = blank_object(RTX_OBJ_ROTATE_Y);
        o.child = object->flatten(b);
        o.f[0] = sin_theta; o.f[1] = cos_theta;
        b.objects.push_back(o);
        return RTX_REF(RTX_REF_OBJECT, b.objects.size() - 1);
    });
}
rtx_ref_t ConstantMedium::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        rtx_object_t o = blank_object(RTX_OBJ_MEDIUM);
        o.child = boundary->flatten(b);
        o.f[0] = density;
        o.material = material->flatten(b);
        o.prim_id = prim_id;
        b.objects.push_back(o);
        return RTX_REF(RTX_REF_OBJECT, b.objects.size() - 1);
    });
}

// f32::total_cmp key (mesh.rs:587-591)
static inline int32_t total_key(float f) {
    int32_t v;
    std::memcpy(&v, &f, 4);
    v ^= (int32_t)(((uint32_t)(v >> 31)) >> 1);
    return v;
}
// Bvh::from (mesh.rs:564-604).  The reference clones the slice, sorts the clone and recurses into clones of its halves;
// sorting a sub-range of one working array in place gives the same order (each recursion only ever reads its own
// range, and the sort is stable over the order its parent left), without copying reference-counted pointers at every
// level.  Large ranges build their two halves on separate threads: the halves are independent, so the tree is the same.
namespace {
struct BvhItem {
    const MeshP* mesh;
    const Aabb* box;
    int32_t key;
};
std::shared_ptr<Bvh> bvh_build(BvhItem* begin, BvhItem* end, int depth) {
    const size_t n = (size_t)(end - begin);
    auto node = std::make_shared<Bvh>();
    Aabb box;  // Aabb::default() = 0..0 on every axis: node boxes always contain the origin
    for (BvhItem* it = begin; it != end; ++it) box = box.merge(*it->box);
    node->bbox = box;
    const int axis = box.longest_axis();
    if (n == 1) {
        node->left = *begin->mesh;
        node->right = *begin->mesh;
    } else if (n == 2) {
        node->left = *begin[0].mesh;
        node->right = *begin[1].mesh;
    } else {
        for (BvhItem* it = begin; it != end; ++it) it->key = total_key((*it->box)[axis].start);
        std::stable_sort(begin, end, [](const BvhItem& a, const BvhItem& b) { return a.key < b.key; });
        BvhItem* mid = begin + n / 2;
        if (n >= (1u << 16) && depth < 5) {
            auto left = std::async(std::launch::async, bvh_build, begin, mid, depth + 1);
            node->right = bvh_build(mid, end, depth + 1);
            node->left = left.get();
        } else {
            node->left = bvh_build(begin, mid, depth + 1);
            node->right = bvh_build(mid, end, depth + 1);
        }
    }
    return node;
}
}  // namespace

std::shared_ptr<Bvh> Bvh::from(const std::vector<MeshP>& value) {
    if (value.empty()) throw std::invalid_argument("Bvh::from: empty slice (the reference recurses without bound)");
    std::vector<BvhItem> items(value.size());
    for (size_t i = 0; i < value.size(); i++) items[i] = BvhItem{&value[i], &value[i]->bounding_box(), 0};
    return bvh_build(items.data(), items.data() + items.size(), 0);
}
rtx_ref_t Bvh::flatten(SceneBuilder& b) const {
    return memoized(b, this, [&] {
        uint32_t idx = (uint32_t)b.nodes.size();
        b.nodes.push_back(rtx_bvh_node_t{});
        rtx_ref_t l = left->flatten(b);
        rtx_ref_t r = right->flatten(b);
        rtx_bvh_node_t n;
        n.bounds[0] = bbox.x.start; n.bounds[1] = bbox.x.end;
        n.bounds[2] = bbox.y.start; n.bounds[3] = bbox.y.end;
        n.bounds[4] = bbox.z.start; n.bounds[5] = bbox.z.end;
        n.left = l; n.right = r;
        b.nodes[idx] = n;
        return RTX_REF(RTX_REF_NODE, idx);
    });
}

void flatten_scene(const Mesh& root, SceneBuilder& b) { b.root = root.flatten(b); }

// ------------------------------------------------------------------------------------------- camera
Camera::Camera(const CameraConfig& cfg) : config(cfg) {  // camera.rs:57-96
    float aspect_ratio = (float)cfg.width / (float)cfg.height;
    center = cfg.lookfrom;
    float theta = degrees_to_radians(cfg.vfov);
    float h = std::tan(theta / 2.0f);
    float viewport_height = 2.0f * h * cfg.focus_dist;
    float viewport_width = viewport_height * aspect_ratio;
    Vec3 w = normalize(cfg.lookfrom - cfg.lookat);
    Vec3 u = cross(cfg.vup, w);  // deliberately not normalised (camera.rs:68)
    Vec3 v = cross(w, u);
    Vec3 viewport_u = viewport_width * u;
    Vec3 viewport_v = viewport_height * (-v);
    pixel_delta_u = viewport_u / (float)cfg.width;
    pixel_delta_v = viewport_v / (float)cfg.height;
    Vec3 viewport_upper_left = center - (cfg.focus_dist * w) - viewport_u / 2.0f - viewport_v / 2.0f;
    pixel00_loc = viewport_upper_left + 0.5f * (pixel_delta_u + pixel_delta_v);
    float defocus_radius = cfg.focus_dist * std::tan(degrees_to_radians(cfg.defocus_angle / 2.0f));
    defocus_disk_u = u * defocus_radius;
    defocus_disk_v = v * defocus_radius;
    pixel_samples_scale = 1.0f / (float)cfg.samples_per_pixel;
}

rtx_camera_t Camera::to_ffi() const {
    rtx_camera_t c;
    std::memset(&c, 0, sizeof(c));
    c.width = config.width; c.height = config.height;
    c.samples_per_pixel = config.samples_per_pixel; c.max_depth = config.max_depth;
    c.defocus_angle = config.defocus_angle;
    auto put = [](float* d, Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    put(c.center, center); put(c.pixel00_loc, pixel00_loc);
    put(c.pixel_delta_u, pixel_delta_u); put(c.pixel_delta_v, pixel_delta_v);
    put(c.defocus_disk_u, defocus_disk_u); put(c.defocus_disk_v, defocus_disk_v);
    put(c.background, config.background);
    c.pixel_samples_scale = pixel_samples_scale;
    return c;
}

GpuScene::GpuScene(const Mesh& root) {
    flatten_scene(root, builder_);
    rtx_scene_desc_t d = builder_.desc();
    int rc = rtx_scene_create(&d, &scene_);
    if (rc != RTX_OK) throw std::runtime_error(std::string("rtx_scene_create: ") + rtx_last_error());
}
GpuScene::~GpuScene() {
    if (scene_) rtx_scene_destroy(scene_);
}

RgbImage Camera::render(const GpuScene& scene, uint64_t seed, std::vector<float>* accum, rtx_stats_t* stats, uint32_t n_gpus, uint32_t split) const {
    RgbImage img;
    img.width = config.width; img.height = config.height;
    img.data.resize((size_t)config.width * config.height * 3);
    if (accum) accum->resize((size_t)config.width * config.height * 3);
    rtx_camera_t cam = to_ffi();
    rtx_render_opts_t opts;
    std::memset(&opts, 0, sizeof(opts));
    opts.seed = seed;
    opts.n_gpus = n_gpus;
    opts.split = split;
    int rc = rtx_render(scene.handle(), &cam, &opts, img.data.data(), accum ? accum->data() : nullptr, stats);
    if (rc != RTX_OK) throw std::runtime_error(std::string("rtx_render: ") + rtx_last_error());
    return img;
}
RgbImage Camera::render(const Mesh& mesh, uint64_t seed) const {
    GpuScene scene(mesh);
    return render(scene, seed);
}

}  // namespace ray_tracing
