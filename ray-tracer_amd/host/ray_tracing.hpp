// Host layer above the C ABI (include/rtx.h): the construction side of the reference's scene types.
//
// In the reference a scene is a graph of Rust objects (Sphere, Quad, Cube, Translate, RotateY, ConstantMedium,
// World, Bvh; Lambertian, Metal, Dielectric, DiffuseLight, Isotropic; Solid/Checker/Image/Noise textures) whose
// hit()/scatter()/value() methods the CPU renderer calls.  Here the same constructors exist with the same names
// and arguments, they derive the same construction-time quantities (bounding boxes, quad normal/d/w, BVH
// topology, camera constants) with the same un-fused f32 arithmetic, and instead of hit() each type has
// flatten(): it appends itself to the flat arrays of an rtx_scene_desc_t.  Camera::render() then hands the
// flattened scene to the HIP library, exactly where Camera::render (camera.rs:98) would start its pixel loop.
//
// Build with -ffp-contract=off (rustc never contracts a*b+c).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "rtx.h"

namespace ray_tracing {

// ------------------------------------------------------------------------------------------- math (glam Vec3 subset)
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 normalize(Vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

constexpr float PI = 3.1415926535897932385f;                         // utils.rs:4
inline float degrees_to_radians(float d) { return d * PI / 180.0f; }  // utils.rs:46-48

// ------------------------------------------------------------------------------------------- Interval / Aabb
struct Interval {  // lib.rs:32-104
    float start = 0, end = 0;
    Interval() {}
    Interval(float s, float e) : start(s), end(e) {}
    float size() const { return end - start; }
    Interval expand(float delta) const { float p = delta / 2.0f; return {start - p, end + p}; }
    Interval join(const Interval& o) const { return {std::fmin(start, o.start), std::fmax(end, o.end)}; }
};

struct Aabb {  // mesh.rs:434-552, construction side
    Interval x, y, z;
    const Interval& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    static Interval pad_to_minimus(Interval iv) { return iv.size() < 0.0001f ? iv.expand(0.0001f) : iv; }
    static Aabb make(Interval a, Interval b, Interval c) {
        Aabb r; r.x = pad_to_minimus(a); r.y = pad_to_minimus(b); r.z = pad_to_minimus(c); return r;
    }
    static Aabb from_extremes(Vec3 a, Vec3 b) {
        return make(a.x <= b.x ? Interval(a.x, b.x) : Interval(b.x, a.x), a.y <= b.y ? Interval(a.y, b.y) : Interval(b.y, a.y),
                    a.z <= b.z ? Interval(a.z, b.z) : Interval(b.z, a.z));
    }
    Aabb merge(const Aabb& o) const { return make(x.join(o.x), y.join(o.y), z.join(o.z)); }
    Aabb operator+(Vec3 r) const { Aabb b; b.x = {x.start + r.x, x.end + r.x}; b.y = {y.start + r.y, y.end + r.y}; b.z = {z.start + r.z, z.end + r.z}; return b; }
    int longest_axis() const {
        float dx = x.end - x.start, dy = y.end - y.start, dz = z.end - z.start;
        if (dx > dy && dx > dz) return 0;
        return dy > dz ? 1 : 2;
    }
};

// ------------------------------------------------------------------------------------------- flat scene under construction
class SceneBuilder {
  public:
    // SoA primitive storage
    std::vector<float> sph[7];  // cx cy cz dx dy dz r
    std::vector<uint32_t> sph_mat, sph_id;
    std::vector<float> quad[16];
    std::vector<uint32_t> quad_mat, quad_id;
    std::vector<float> tri[9];
    std::vector<uint32_t> tri_mat, tri_id;
    std::vector<rtx_object_t> objects;
    std::vector<rtx_ref_t> refs;
    std::vector<rtx_bvh_node_t> nodes;
    std::vector<rtx_material_t> materials;
    std::vector<rtx_texture_t> textures;
    std::vector<rtx_image_t> images;
    std::vector<rtx_perlin_t> perlins;
    rtx_ref_t root = RTX_MISS;
    // shared sub-objects (Arc clones in the reference) are flattened once
    std::unordered_map<const void*, uint32_t> memo;

    rtx_scene_desc_t desc() const;
};

// ------------------------------------------------------------------------------------------- textures (texture.rs)
struct Texture {
    virtual ~Texture() {}
    virtual uint32_t flatten(SceneBuilder& b) const = 0;
};
using TextureP = std::shared_ptr<const Texture>;

struct SolidTexture : Texture {  // texture.rs:14-44
    Vec3 albedo{1, 1, 1};
    SolidTexture() {}
    explicit SolidTexture(Vec3 a) : albedo(a) {}
    static std::shared_ptr<SolidTexture> rgb(float r, float g, float b) { return std::make_shared<SolidTexture>(Vec3(r, g, b)); }
    uint32_t flatten(SceneBuilder& b) const override;
};
struct CheckerTexture : Texture {  // texture.rs:46-81
    float scale = 1.0f;
    TextureP even, odd;
    CheckerTexture() : even(SolidTexture::rgb(0.2f, 0.3f, 0.1f)), odd(SolidTexture::rgb(0.9f, 0.9f, 0.9f)) {}
    CheckerTexture(float s, TextureP e, TextureP o) : scale(s), even(e), odd(o) {}
    uint32_t flatten(SceneBuilder& b) const override;
};
struct ImageTexture : Texture {  // texture.rs:83-115; owns the decoded RGB8 raster
    std::vector<uint8_t> rgb;
    uint32_t width = 0, height = 0;
    ImageTexture(const uint8_t* data, uint32_t w, uint32_t h) : rgb(data, data + (size_t)w * h * 3), width(w), height(h) {}
    uint32_t flatten(SceneBuilder& b) const override;
};
struct NoiseTexture : Texture {  // texture.rs:117-138 with the Perlin tables (texture.rs:140-145) as inputs
    float scale;
    std::vector<float> values;  // 256 * 3
    std::vector<uint32_t> perm_x, perm_y, perm_z;
    NoiseTexture(float s, const float* v, const uint32_t* px, const uint32_t* py, const uint32_t* pz)
        : scale(s), values(v, v + 768), perm_x(px, px + 256), perm_y(py, py + 256), perm_z(pz, pz + 256) {}
    uint32_t flatten(SceneBuilder& b) const override;
};

// ------------------------------------------------------------------------------------------- materials (material.rs)
struct Material {
    virtual ~Material() {}
    virtual uint32_t flatten(SceneBuilder& b) const = 0;
};
using MaterialP = std::shared_ptr<const Material>;

struct Lambertian : Material {  // material.rs:53-96
    TextureP texture;
    explicit Lambertian(TextureP t) : texture(t) {}
    static std::shared_ptr<Lambertian> rgb(float r, float g, float b) { return std::make_shared<Lambertian>(SolidTexture::rgb(r, g, b)); }
    uint32_t flatten(SceneBuilder& b) const override;
};
struct Metal : Material {  // material.rs:98-135
    TextureP texture;
    float roughness = 0;
    Metal(TextureP t, float r) : texture(t), roughness(r) {}
    static std::shared_ptr<Metal> rgb(float r, float g, float b) { return std::make_shared<Metal>(SolidTexture::rgb(r, g, b), 0.0f); }
    std::shared_ptr<Metal> with_roughness(float r) const { return std::make_shared<Metal>(texture, r); }
    uint32_t flatten(SceneBuilder& b) const override;
};
struct Dielectric : Material {  // material.rs:137-179
    float refraction_index = 1.5f;
    Dielectric() {}
    explicit Dielectric(float ri) : refraction_index(ri) {}
    uint32_t flatten(SceneBuilder& b) const override;
};
struct DiffuseLight : Material {  // material.rs:181-212
    TextureP texture;
    explicit DiffuseLight(TextureP t) : texture(t) {}
    static std::shared_ptr<DiffuseLight> rgb(float r, float g, float b) { return std::make_shared<DiffuseLight>(SolidTexture::rgb(r, g, b)); }
    uint32_t flatten(SceneBuilder& b) const override;
};
struct Isotropic : Material {  // material.rs:214-246
    TextureP texture;
    explicit Isotropic(TextureP t) : texture(t) {}
    uint32_t flatten(SceneBuilder& b) const override;
};

// ------------------------------------------------------------------------------------------- hittables (mesh.rs)
struct Mesh {  // mesh.rs:17-21 with hit() replaced by flatten()
    virtual ~Mesh() {}
    virtual const Aabb& bounding_box() const = 0;
    virtual rtx_ref_t flatten(SceneBuilder& b) const = 0;
};
using MeshP = std::shared_ptr<const Mesh>;

struct Sphere : Mesh {  // mesh.rs:103-188
    Vec3 initial_center, center_delta;
    float radius = 0;
    MaterialP material;
    Aabb bbox;
    uint32_t prim_id = 0;
    static std::shared_ptr<Sphere> stationary(Vec3 center, float radius, MaterialP m);  // mesh.rs:147-159
    static std::shared_ptr<Sphere> moving(Vec3 from, Vec3 to, float radius, MaterialP m);  // mesh.rs:161-176
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct Quad : Mesh {  // mesh.rs:190-257
    Vec3 translation, u, v, normal, w;
    float d = 0;
    MaterialP material;
    Aabb bbox;
    uint32_t prim_id = 0;
    Quad(Vec3 translation, Vec3 u, Vec3 v, MaterialP m);  // mesh.rs:229-246
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct Triangle : Mesh {  // build-defined primitive
    Vec3 v0, v1, v2;
    MaterialP material;
    Aabb bbox;
    uint32_t prim_id = 0;
    Triangle(Vec3 a, Vec3 b, Vec3 c, MaterialP m);
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct World : Mesh {  // mesh.rs:54-101
    std::vector<MeshP> meshes;
    Aabb bbox;
    void push(MeshP m) { bbox = bbox.merge(m->bounding_box()); meshes.push_back(m); }
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct Cube : Mesh {  // mesh.rs:259-322
    World world;
    Cube(Vec3 a, Vec3 b, MaterialP m, uint32_t first_prim_id = 0);
    const Aabb& bounding_box() const override { return world.bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override { return world.flatten(b); }
};
struct Translate : Mesh {  // mesh.rs:324-354
    Aabb bbox;
    Vec3 offset;
    MeshP object;
    Translate(MeshP obj, Vec3 off) : bbox(obj->bounding_box() + off), offset(off), object(obj) {}
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct RotateY : Mesh {  // mesh.rs:356-432
    MeshP object;
    float sin_theta, cos_theta;
    Aabb bbox;
    RotateY(MeshP obj, float angle);
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct ConstantMedium : Mesh {  // mesh.rs:627-696
    MeshP boundary;
    std::shared_ptr<const Isotropic> material;
    float density;
    uint32_t prim_id = 0;
    ConstantMedium(MeshP boundary_, float density_, TextureP tex) : boundary(boundary_), material(std::make_shared<Isotropic>(tex)), density(density_) {}
    const Aabb& bounding_box() const override { return boundary->bounding_box(); }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};
struct Bvh : Mesh {  // mesh.rs:554-625
    MeshP left, right;
    Aabb bbox;
    static std::shared_ptr<Bvh> from(const std::vector<MeshP>& value);  // mesh.rs:564-604
    static std::shared_ptr<Bvh> from(const World& w) { return from(w.meshes); }
    const Aabb& bounding_box() const override { return bbox; }
    rtx_ref_t flatten(SceneBuilder& b) const override;
};

// ------------------------------------------------------------------------------------------- camera (camera.rs)
struct CameraConfig {  // camera.rs:13-43
    uint32_t width = 100, height = 100;
    uint32_t samples_per_pixel = 10, max_depth = 10;
    float vfov = 90.0f;
    Vec3 lookfrom{0, 0, 0}, lookat{0, 0, -1}, vup{0, 1, 0};
    float defocus_angle = 0.0f, focus_dist = 10.0f;
    Vec3 background{0.7f, 0.8f, 1.0f};
};

struct RgbImage {  // image::RgbImage: row-major, top row first
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
};

// A flattened scene resident on the current GPU (what Camera::render borrows for the length of a render).
class GpuScene {
  public:
    explicit GpuScene(const Mesh& root);
    ~GpuScene();
    GpuScene(const GpuScene&) = delete;
    GpuScene& operator=(const GpuScene&) = delete;
    rtx_scene_t* handle() const { return scene_; }  // renders re-order the device data for their camera: not const
    const SceneBuilder& flat() const { return builder_; }

  private:
    SceneBuilder builder_;
    rtx_scene_t* scene_ = nullptr;
};

struct Camera {  // camera.rs:45-169
    CameraConfig config;
    Vec3 center, pixel00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v;
    float pixel_samples_scale;
    explicit Camera(const CameraConfig& cfg);  // camera.rs:57-96
    rtx_camera_t to_ffi() const;
    // Camera::render (camera.rs:98): flatten, upload, path-trace on the GPU, return the image.  Throws on failure.
    RgbImage render(const Mesh& mesh, uint64_t seed = 0) const;
    // n_gpus > 1: the one blocking call drives GPUs 0 .. n_gpus-1 itself (split = RTX_SPLIT_SAMPLES or RTX_SPLIT_TILES).
    RgbImage render(const GpuScene& scene, uint64_t seed = 0, std::vector<float>* accum = nullptr, rtx_stats_t* stats = nullptr,
                    uint32_t n_gpus = 1, uint32_t split = RTX_SPLIT_SAMPLES) const;
};

// Flatten `root` into `b` (sets b.root).  Throws std::invalid_argument on malformed input.
void flatten_scene(const Mesh& root, SceneBuilder& b);

}  // namespace ray_tracing
