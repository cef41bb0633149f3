// TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the magmast/ray-tracer render path.
//
// This file restates, line by line and quirk by quirk, the Rust reference's
//   src/lib.rs, src/utils.rs, src/camera.rs, src/mesh.rs, src/material.rs, src/texture.rs
// as scalar IEEE f32 C++ (build with -ffp-contract=off: rustc never fuses a*b+c).
// It is the checker the HIP path is compared against and the CPU baseline that is timed
// beside it. Nothing in the shipped product may link, load or call it.
//
// Parity pin status: the reference has no tests, golden vectors or fixtures of its own, and
// no Rust toolchain exists in the build container, so the pins are the known-answer values
// derived by hand from the reference source (tests/test_oracle_kat.py). Arithmetic that lives in
// third-party crates (glam 0.27.0 Vec3, bevy_color 0.14.2 sRGB transfer, rand 0.8.5 float
// conversions) is restated from those crates' published algorithms: "parity unpinned" for them.
//
// Every function cites the reference file:line it follows.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace orc {

static const float INF = std::numeric_limits<float>::infinity();
// utils.rs:4  (3.1415926535897932385 rounded to f32)
static const float PI = 3.1415926535897932385f;

// ----------------------------------------------------------------------------------------------
// glam 0.27.0 scalar Vec3 (Cargo.lock:577): plain 3 x f32, no SIMD, no FMA.
// ----------------------------------------------------------------------------------------------
struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
static inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
static inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
static inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
// glam: dot = (x*x') + (y*y') + (z*z'), evaluated left to right
static inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline float length_squared(Vec3 a) { return dot(a, a); }
static inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
// glam: normalize = self * (1.0 / length)   (reciprocal then multiply, not a divide)
static inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }
// glam: cross
static inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// Rust f32::max / f32::min: if one operand is NaN the other is returned (IEEE maxNum/minNum).
static inline float rmax(float a, float b) { return std::fmax(a, b); }
static inline float rmin(float a, float b) { return std::fmin(a, b); }
// Rust f32::clamp(min,max): NaN stays NaN
static inline float rclamp(float v, float lo, float hi) {
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}
// Rust `as i32` / `as u32` / `as u8` from f32: saturating, NaN -> 0
static inline int32_t as_i32(float v) {
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
static inline uint32_t as_u32(float v) {
    if (!(v == v)) return 0;
    if (v >= 4294967296.0f) return UINT32_MAX;
    if (v <= 0.0f) return 0;
    return (uint32_t)v;
}

// ----------------------------------------------------------------------------------------------
// Random numbers.  The reference draws everything from rand::thread_rng() (ChaCha12, OS seeded;
// camera.rs:110, material.rs:88,106,162,233, mesh.rs:659), so it is not reproducible.  The oracle
// keeps the ChaCha12 cost and the rand 0.8.5 float conversions but keys the generator on
// (seed, pixel) so a render is a pure function of its inputs.
//   kind 0 "faithful": ChaCha12 + the reference's rejection samplers (utils.rs:14-39).
//   kind 1 "mirror"  : the 32-bit PCG stream and the direct (non-rejection) samplers that the HIP
//                      kernels use, so that a GPU image can be compared sample by sample.
// ----------------------------------------------------------------------------------------------
struct ChaCha12 {
    uint32_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    void seed(uint64_t s0, uint64_t s1) {
        // splitmix64 expansion of the two seed words into a 256-bit key
        uint64_t z = s0 * 0x9E3779B97F4A7C15ull + s1;
        for (int i = 0; i < 4; i++) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t v = z;
            v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
            v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
            v ^= v >> 31;
            key[2 * i] = (uint32_t)v;
            key[2 * i + 1] = (uint32_t)(v >> 32);
        }
        counter = 0;
        idx = 16;
    }
    static inline uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
    void refill() {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                          key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                          (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
        uint32_t w[16];
        std::memcpy(w, s, sizeof(w));
#define ORC_QR(a, b, c, d)                                                                       \
    w[a] += w[b]; w[d] ^= w[a]; w[d] = rotl(w[d], 16);                                           \
    w[c] += w[d]; w[b] ^= w[c]; w[b] = rotl(w[b], 12);                                           \
    w[a] += w[b]; w[d] ^= w[a]; w[d] = rotl(w[d], 8);                                            \
    w[c] += w[d]; w[b] ^= w[c]; w[b] = rotl(w[b], 7);
        for (int r = 0; r < 6; r++) {  // 12 rounds = 6 double rounds
            ORC_QR(0, 4, 8, 12) ORC_QR(1, 5, 9, 13) ORC_QR(2, 6, 10, 14) ORC_QR(3, 7, 11, 15)
            ORC_QR(0, 5, 10, 15) ORC_QR(1, 6, 11, 12) ORC_QR(2, 7, 8, 13) ORC_QR(3, 4, 9, 14)
        }
#undef ORC_QR
        for (int i = 0; i < 16; i++) buf[i] = w[i] + s[i];
        counter++;
        idx = 0;
    }
    inline uint32_t next() {
        if (idx >= 16) refill();
        return buf[idx++];
    }
};

static inline uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}

struct Rng {
    int kind = 0;
    ChaCha12 cc;
    uint32_t pcg_state = 0, pcg_inc = 1;
    void seed_faithful(uint64_t seed, uint64_t stream) {
        kind = 0;
        cc.seed(seed, stream);
    }
    // identical to seed_word() + rng_init() of the HIP library: two keys hashed from both halves of the seed, then
    // separate hash chains over (key, pixel, sample) for the state and for the increment
    void seed_mirror(uint64_t seed, uint32_t pixel, uint32_t sample) {
        kind = 1;
        const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
        const uint32_t k0 = pcg_hash(lo ^ pcg_hash(hi));
        const uint32_t k1 = pcg_hash((hi + 0x85EBCA6Bu) ^ pcg_hash(lo + 0x9E3779B9u));
        pcg_state = pcg_hash(pcg_hash(k0 + pixel) + sample * 0x9E3779B9u);
        pcg_inc = (pcg_hash(pcg_hash(k1 ^ pixel) + sample * 0x85EBCA6Bu) << 1) | 1u;
    }
    inline uint32_t next_u32() {
        if (kind == 0) return cc.next();
        pcg_state = pcg_state * 747796405u + pcg_inc;
        uint32_t s = pcg_state;
        uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        return (w >> 22u) ^ w;
    }
    // rand 0.8.5 Standard for f32: 24 high bits * 2^-24  -> [0,1)
    inline float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
    // rand 0.8.5 UniformFloat::<f32>::sample_single(low, high): 23 bits into [1,2), -1, *scale + low
    inline float gen_range(float low, float high) {
        float scale = high - low;
        for (;;) {
            uint32_t bits = (next_u32() >> 9) | 0x3F800000u;
            float v12;
            std::memcpy(&v12, &bits, 4);
            float v01 = v12 - 1.0f;
            float res = v01 * scale + low;
            if (res < high) return res;
        }
    }
};

static thread_local Rng* tl_rng = nullptr;  // stands in for rand::thread_rng()

// utils.rs:6-12
static inline Vec3 random_vec(Rng& r) {
    float x = r.gen_range(-1.0f, 1.0f);
    float y = r.gen_range(-1.0f, 1.0f);
    float z = r.gen_range(-1.0f, 1.0f);
    return Vec3(x, y, z);
}
// utils.rs:14-26
static inline Vec3 random_vec_in_unit_sphere(Rng& r) {
    for (;;) {
        Vec3 p = random_vec(r);
        if (length_squared(p) < 1.0f) return p;
    }
}
// mirror-mode direct sampler: uniform on the unit sphere (same distribution as
// utils.rs:37-39, normalize(rejection sample)); formulas shared with the HIP kernels.
static inline Vec3 mirror_unit_vec(Rng& r) {
    float u1 = r.gen_f32();
    float u2 = r.gen_f32();
    float z = 1.0f - 2.0f * u1;
    float rr = std::sqrt(rmax(0.0f, 1.0f - z * z));
    float phi = 6.2831853071795864769f * u2;
    return Vec3(rr * std::cos(phi), rr * std::sin(phi), z);
}
// utils.rs:37-39
static inline Vec3 random_unit_vec(Rng& r) {
    if (r.kind == 1) return mirror_unit_vec(r);
    return normalize(random_vec_in_unit_sphere(r));
}
// utils.rs:28-35
static inline Vec3 random_vec_in_unit_disk(Rng& r) {
    if (r.kind == 1) {
        float u1 = r.gen_f32();
        float u2 = r.gen_f32();
        float rr = std::sqrt(u1);
        float phi = 6.2831853071795864769f * u2;
        return Vec3(rr * std::cos(phi), rr * std::sin(phi), 0.0f);
    }
    for (;;) {
        float x = r.gen_range(-1.0f, 1.0f);
        float y = r.gen_range(-1.0f, 1.0f);
        Vec3 p(x, y, 0.0f);
        if (length_squared(p) < 1.0f) return p;
    }
}
// utils.rs:41-44
static inline bool near_zero(Vec3 v) {
    float s = 1e-8f;
    return std::fabs(v.x) < s && std::fabs(v.y) < s && std::fabs(v.z) < s;
}
// utils.rs:46-48
static inline float degrees_to_radians(float d) { return d * PI / 180.0f; }

// ----------------------------------------------------------------------------------------------
// lib.rs
// ----------------------------------------------------------------------------------------------
struct Ray {  // lib.rs:11-30
    Vec3 origin, direction;
    float time = 0;
    Ray() {}
    Ray(Vec3 o, Vec3 d, float t) : origin(o), direction(d), time(t) {}
    Vec3 get_point(float distance) const { return origin + direction * distance; }  // lib.rs:27-29
};

struct Interval {  // lib.rs:32-104 (wraps Range<f32>)
    float start = 0, end = 0;  // Default = 0.0..0.  (lib.rs:35-39)
    Interval() {}
    Interval(float s, float e) : start(s), end(e) {}
    bool contains(float v) const { return start <= v && v < end; }  // Range::contains, half open (lib.rs:82-84)
    Interval expand(float delta) const {  // lib.rs:86-89
        float padding = delta / 2.0f;
        return Interval(start - padding, end + padding);
    }
    Interval join(const Interval& o) const {  // lib.rs:91-95
        return Interval(rmin(start, o.start), rmax(end, o.end));
    }
    float clamp(float v) const { return rclamp(v, start, end); }  // lib.rs:97-99
    float size() const { return end - start; }                    // lib.rs:101-103
    Interval add(float rhs) const { return Interval(start + rhs, end + rhs); }  // lib.rs:47-53
};
static const Interval UNIVERSE(-INF, INF);  // lib.rs:64

// ----------------------------------------------------------------------------------------------
// Work counters (instrumentation only; used to derive algorithmic flops per sample).
// ----------------------------------------------------------------------------------------------
struct Counters {
    uint64_t node_tests = 0, sphere_tests = 0, sphere_hits = 0, quad_tests = 0, quad_hits = 0;
    uint64_t rays = 0, samples = 0, misses = 0, depth_exhausted = 0;
    uint64_t shade_lambert = 0, shade_metal = 0, shade_dielectric = 0, shade_light = 0, shade_iso = 0;
    uint64_t tex_solid = 0, tex_checker = 0, tex_image = 0, tex_noise = 0, medium_tests = 0;
    void add(const Counters& o) {
        const uint64_t* s = reinterpret_cast<const uint64_t*>(&o);
        uint64_t* d = reinterpret_cast<uint64_t*>(this);
        for (size_t i = 0; i < sizeof(Counters) / 8; i++) d[i] += s[i];
    }
};
static thread_local Counters tl_cnt;
// -DORC_NO_COUNTERS compiles the tallies out: the build that is TIMED as the CPU baseline carries no instrumentation.
#ifdef ORC_NO_COUNTERS
#define ORC_COUNT(field) ((void)0)
#else
#define ORC_COUNT(field) (tl_cnt.field++)
#endif

// ----------------------------------------------------------------------------------------------
// texture.rs
// ----------------------------------------------------------------------------------------------
struct Texture {  // texture.rs:10-12
    virtual ~Texture() {}
    virtual Vec3 value(float u, float v, Vec3 point) const = 0;
};
struct SolidTexture : Texture {  // texture.rs:14-44
    Vec3 albedo;
    explicit SolidTexture(Vec3 a) : albedo(a) {}
    Vec3 value(float, float, Vec3) const override {
        ORC_COUNT(tex_solid);
        return albedo;
    }
};
struct CheckerTexture : Texture {  // texture.rs:46-81
    float scale;
    std::shared_ptr<Texture> even, odd;
    Vec3 value(float u, float v, Vec3 point) const override {  // texture.rs:63-74
        ORC_COUNT(tex_checker);
        int32_t x = as_i32(std::floor(point.x * scale));
        int32_t y = as_i32(std::floor(point.y * scale));
        int32_t z = as_i32(std::floor(point.z * scale));
        int32_t sum = (int32_t)((uint32_t)x + (uint32_t)y + (uint32_t)z);
        if (sum % 2 == 0) return even->value(u, v, point);
        return odd->value(u, v, point);
    }
};
struct ImageTexture : Texture {  // texture.rs:83-115; raster is the decoded RGB8 image
    std::vector<uint8_t> rgb;
    uint32_t width = 0, height = 0;
    Vec3 value(float uu, float vv, Vec3) const override {  // texture.rs:100-114
        ORC_COUNT(tex_image);
        float u = Interval(0.0f, 1.0f).clamp(uu);
        float v = 1.0f - Interval(0.0f, 1.0f).clamp(vv);
        uint32_t i = as_u32(u * ((float)width - 1.0f));
        uint32_t j = as_u32(v * ((float)height - 1.0f));
        const uint8_t* px = &rgb[((size_t)j * width + i) * 3];
        float color_scale = 1.0f / 255.0f;
        return Vec3((float)px[0] * color_scale, (float)px[1] * color_scale, (float)px[2] * color_scale);
    }
};
struct Perlin {  // texture.rs:140-244, N = 256; tables are inputs (the reference draws them from thread_rng)
    Vec3 values[256];
    uint32_t perm_x[256], perm_y[256], perm_z[256];
    float perlin_interp(const Vec3 c[2][2][2], float u, float v, float w) const {  // texture.rs:210-229
        float uu = u * u * (3.0f - 2.0f * u);
        float vv = v * v * (3.0f - 2.0f * v);
        float ww = w * w * (3.0f - 2.0f * w);
        float accum = 0.0f;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    Vec3 weight_v(u - (float)i, v - (float)j, w - (float)k);
                    accum += ((float)i * uu + (1.0f - (float)i) * (1.0f - uu)) *
                             ((float)j * vv + (1.0f - (float)j) * (1.0f - vv)) *
                             ((float)k * ww + (1.0f - (float)k) * (1.0f - ww)) * dot(c[i][j][k], weight_v);
                }
        return accum;
    }
    float noise(Vec3 p) const {  // texture.rs:187-208
        float u = p.x - std::floor(p.x);
        float v = p.y - std::floor(p.y);
        float w = p.z - std::floor(p.z);
        int32_t i = as_i32(std::floor(p.x));
        int32_t j = as_i32(std::floor(p.y));
        int32_t k = as_i32(std::floor(p.z));
        Vec3 c[2][2][2];
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++)
                    c[di][dj][dk] = values[perm_x[(uint32_t)(i + di) & 255u] ^ perm_y[(uint32_t)(j + dj) & 255u] ^
                                           perm_z[(uint32_t)(k + dk) & 255u]];
        return perlin_interp(c, u, v, w);
    }
    float turb(Vec3 p, int depth) const {  // texture.rs:231-243
        float accum = 0.0f;
        Vec3 temp_p = p;
        float weight = 1.0f;
        for (int n = 0; n < depth; n++) {
            accum += weight * noise(temp_p);
            temp_p = temp_p * 2.0f;
            weight *= 0.5f;
        }
        return std::fabs(accum);
    }
};
struct NoiseTexture : Texture {  // texture.rs:117-138
    Perlin perlin;
    float scale;
    Vec3 value(float, float, Vec3 point) const override {  // texture.rs:132-137
        ORC_COUNT(tex_noise);
        float s = 1.0f + std::sin(scale * point.z + 10.0f * perlin.turb(point, 7));
        // Vec3::ONE * 0.5 * (...)  evaluates (ONE*0.5) then * s
        Vec3 color = Vec3(1.0f, 1.0f, 1.0f) * 0.5f * s;
        return color;
    }
};

// ----------------------------------------------------------------------------------------------
// material.rs
// ----------------------------------------------------------------------------------------------
struct Material;
struct Hit {  // mesh.rs:23-30 (+ prim_id: which primitive object produced the hit, for ID parity)
    Vec3 point, normal;
    float distance = 0;
    bool front_face = true;
    const Material* material = nullptr;
    float u = 0, v = 0;
    uint32_t prim_id = 0xFFFFFFFFu;
    bool valid = false;
};
// mesh.rs:33-51  Hit::new -- front_face is computed, used for the flip, then stored as `true`
static inline Hit hit_new(const Ray& ray, float distance, Vec3 normal, const Material* m, float u, float v,
                          uint32_t prim_id) {
    Hit h;
    bool front_face = dot(ray.direction, normal) < 0.0f;
    h.point = ray.get_point(distance);
    h.normal = front_face ? normal : -normal;
    h.distance = distance;
    h.front_face = true;
    h.material = m;
    h.u = u;
    h.v = v;
    h.prim_id = prim_id;
    h.valid = true;
    return h;
}

struct Scatter {  // material.rs:48-51
    bool some = false;
    Vec3 attenuation;
    Ray scattered;
};
struct Material {  // material.rs:30-36
    virtual ~Material() {}
    virtual Scatter scatter(const Ray& ray, const Hit& hit) const = 0;
    virtual Vec3 emitted(float, float, Vec3) const { return Vec3(0, 0, 0); }
};
// material.rs:13-17 ; powi(5) = x * ((x*x)*(x*x)) (binary exponentiation of llvm.powi)
static inline float reflectance(float cosine, float refraction_index) {
    float r0 = (1.0f - refraction_index) / (1.0f + refraction_index);
    r0 *= r0;
    float b = 1.0f - cosine;
    float b2 = b * b;
    float b4 = b2 * b2;
    return r0 + (1.0f - r0) * (b * b4);
}
static inline Vec3 reflect(Vec3 v, Vec3 n) { return v - 2.0f * dot(v, n) * n; }  // material.rs:19-21
static inline Vec3 refract(Vec3 uv, Vec3 n, float etai_over_etat) {             // material.rs:23-28
    float cos_theta = -rmin(dot(uv, n), 1.0f);  // `-uv.dot(n).min(1.)` = -(dot.min(1))
    Vec3 r_out_prep = etai_over_etat * (uv + cos_theta * n);
    Vec3 r_out_parallel = -std::sqrt(std::fabs(1.0f - length_squared(r_out_prep))) * n;
    return r_out_prep + r_out_parallel;
}
struct Lambertian : Material {  // material.rs:53-96
    std::shared_ptr<Texture> texture;
    Scatter scatter(const Ray& ray, const Hit& hit) const override {  // material.rs:71-77, 87-95
        ORC_COUNT(shade_lambert);
        Vec3 dir = hit.normal + random_unit_vec(*tl_rng);
        Vec3 scatter_dir = near_zero(dir) ? hit.normal : dir;
        Scatter s;
        s.some = true;
        s.attenuation = texture->value(hit.u, hit.v, hit.point);
        s.scattered = Ray(hit.point, scatter_dir, ray.time);
        return s;
    }
};
struct Metal : Material {  // material.rs:98-135
    std::shared_ptr<Texture> texture;
    float roughness = 0;
    Scatter scatter(const Ray& ray, const Hit& hit) const override {  // material.rs:105-118
        ORC_COUNT(shade_metal);
        Vec3 reflected = reflect(ray.direction, hit.normal) + (roughness * random_unit_vec(*tl_rng));
        Scatter s;
        s.scattered = Ray(hit.point, reflected, ray.time);
        if (dot(s.scattered.direction, hit.normal) > 0.0f) {
            s.some = true;
            s.attenuation = texture->value(hit.u, hit.v, hit.point);
        }
        return s;
    }
};
struct Dielectric : Material {  // material.rs:137-179
    float refraction_index = 1.5f;
    Scatter scatter(const Ray& r_in, const Hit& hit) const override {  // material.rs:150-172
        ORC_COUNT(shade_dielectric);
        float ri = hit.front_face ? 1.0f / refraction_index : refraction_index;
        Vec3 unit_dir = normalize(r_in.direction);
        float cos_theta = -rmin(dot(unit_dir, hit.normal), 1.0f);
        float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0f;
        Vec3 dir;
        // `||` short-circuits: the random number is drawn only when refraction is possible.
        // (mirror stream: every scatter consumes two draws, as the HIP kernels do; the first is this one)
        float xi = 0.0f;
        bool drawn = false;
        if (tl_rng->kind == 1) {
            xi = tl_rng->gen_f32();
            (void)tl_rng->gen_f32();
            drawn = true;
        }
        if (cannot_refract || reflectance(cos_theta, ri) > (drawn ? xi : tl_rng->gen_f32()))
            dir = reflect(unit_dir, hit.normal);
        else
            dir = refract(unit_dir, hit.normal, ri);
        Scatter s;
        s.some = true;
        s.attenuation = Vec3(1, 1, 1);
        s.scattered = Ray(hit.point, dir, r_in.time);
        return s;
    }
};
struct DiffuseLight : Material {  // material.rs:181-212
    std::shared_ptr<Texture> texture;
    Scatter scatter(const Ray&, const Hit&) const override {
        ORC_COUNT(shade_light);
        return Scatter();
    }
    Vec3 emitted(float u, float v, Vec3 p) const override { return texture->value(u, v, p); }
};
struct Isotropic : Material {  // material.rs:214-246
    std::shared_ptr<Texture> texture;
    Scatter scatter(const Ray& ray, const Hit& hit) const override {  // material.rs:232-239
        ORC_COUNT(shade_iso);
        Scatter s;
        s.some = true;
        s.scattered = Ray(hit.point, random_unit_vec(*tl_rng), ray.time);  // field order: scattered first
        s.attenuation = texture->value(hit.u, hit.v, hit.point);
        return s;
    }
};

// ----------------------------------------------------------------------------------------------
// mesh.rs
// ----------------------------------------------------------------------------------------------
struct Aabb {  // mesh.rs:434-552
    Interval x, y, z;  // Default: 0..0 on every axis (lib.rs:35-39)
    const Interval& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    static Interval pad_to_minimus(Interval iv) {  // mesh.rs:475-482
        float delta = 0.0001f;
        if (iv.size() < delta) return iv.expand(delta);
        return iv;
    }
    static Aabb make(Interval x, Interval y, Interval z) {  // mesh.rs:467-473
        Aabb b;
        b.x = pad_to_minimus(x);
        b.y = pad_to_minimus(y);
        b.z = pad_to_minimus(z);
        return b;
    }
    static Aabb from_extremes(Vec3 a, Vec3 b) {  // mesh.rs:484-502
        return make(a.x <= b.x ? Interval(a.x, b.x) : Interval(b.x, a.x),
                    a.y <= b.y ? Interval(a.y, b.y) : Interval(b.y, a.y),
                    a.z <= b.z ? Interval(a.z, b.z) : Interval(b.z, a.z));
    }
    Aabb merge(const Aabb& o) const { return make(x.join(o.x), y.join(o.y), z.join(o.z)); }  // mesh.rs:504-510
    Aabb add(Vec3 rhs) const {  // mesh.rs:454-464 (no re-padding)
        Aabb b;
        b.x = x.add(rhs.x);
        b.y = y.add(rhs.y);
        b.z = z.add(rhs.z);
        return b;
    }
    bool hit(const Ray& ray, const Interval& ray_t_in) const {  // mesh.rs:512-538
        ORC_COUNT(node_tests);
        Interval ray_t = ray_t_in;
        for (int n = 0; n < 3; n++) {
            const Interval& iv = (*this)[n];
            float adinv = 1.0f / ray.direction[n];
            float t0 = (iv.start - ray.origin[n]) * adinv;
            float t1 = (iv.end - ray.origin[n]) * adinv;
            float lo, hi;
            if (t0 < t1) {
                lo = t0;
                hi = t1;
            } else {
                lo = t1;
                hi = t0;
            }
            ray_t.start = rmax(ray_t.start, lo);
            ray_t.end = rmin(ray_t.end, hi);
            if (ray_t.end <= ray_t.start) return false;
        }
        return true;
    }
    int longest_axis() const {  // mesh.rs:540-551
        float dx = x.end - x.start;
        float dy = y.end - y.start;
        float dz = z.end - z.start;
        if (dx > dy && dx > dz) return 0;
        if (dy > dz) return 1;
        return 2;
    }
};

struct Mesh {  // mesh.rs:17-21
    virtual ~Mesh() {}
    virtual Hit hit(const Ray& ray, const Interval& ray_t) const = 0;
    virtual const Aabb& bounding_box() const = 0;
};
typedef std::shared_ptr<Mesh> MeshP;

struct World : Mesh {  // mesh.rs:54-101
    std::vector<MeshP> meshes;
    Aabb bbox;  // Aabb::default()
    void push(MeshP m) {  // mesh.rs:97-100
        bbox = bbox.merge(m->bounding_box());
        meshes.push_back(m);
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:66-82
        Hit current;
        for (const MeshP& m : meshes) {
            float tmax = current.valid ? current.distance : ray_t.end;
            Interval t(ray_t.start, tmax);
            Hit h = m->hit(ray, t);
            if (h.valid) current = h;
        }
        return current;
    }
    const Aabb& bounding_box() const override { return bbox; }
};

struct Sphere : Mesh {  // mesh.rs:103-188
    Vec3 initial_center, center_delta;
    float radius = 0;
    std::shared_ptr<Material> material;
    Aabb bbox;
    uint32_t prim_id = 0;
    static std::shared_ptr<Sphere> stationary(Vec3 center, float radius, std::shared_ptr<Material> m) {  // mesh.rs:147-159
        auto s = std::make_shared<Sphere>();
        Vec3 rvec(radius, radius, radius);
        s->initial_center = center;
        s->center_delta = Vec3(0, 0, 0);
        s->radius = radius;
        s->material = m;
        s->bbox = Aabb::from_extremes(center - rvec, center + rvec);
        return s;
    }
    static std::shared_ptr<Sphere> moving(Vec3 from, Vec3 to, float radius, std::shared_ptr<Material> m) {  // mesh.rs:161-176
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
    Vec3 center(float time) const { return initial_center + center_delta * time; }  // mesh.rs:178-180
    static void uv(Vec3 p, float& u, float& v) {  // mesh.rs:182-187
        float theta = std::acos(-p.y);
        float phi = std::atan2(-p.z, p.x) + PI;
        u = phi / (2.0f * PI);
        v = theta / PI;
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:113-139
        ORC_COUNT(sphere_tests);
        Vec3 c = center(ray.time);
        Vec3 oc = c - ray.origin;
        float a = length_squared(ray.direction);
        float h = dot(ray.direction, oc);
        float cc = length_squared(oc) - radius * radius;
        float discriminant = h * h - a * cc;
        if (discriminant < 0.0f) return Hit();
        float sqrtd = std::sqrt(discriminant);
        float root = (h - sqrtd) / a;
        if (!ray_t.contains(root)) {
            root = (h + sqrtd) / a;
            if (!ray_t.contains(root)) return Hit();
        }
        ORC_COUNT(sphere_hits);
        Vec3 point = ray.get_point(root);
        Vec3 normal = (point - c) / radius;
        float u, v;
        uv(normal, u, v);
        return hit_new(ray, root, normal, material.get(), u, v, prim_id);
    }
    const Aabb& bounding_box() const override { return bbox; }
};

struct Quad : Mesh {  // mesh.rs:190-257
    Vec3 translation, u, v;
    std::shared_ptr<Material> material;
    Aabb bbox;
    Vec3 normal;
    float d = 0;
    Vec3 w;
    uint32_t prim_id = 0;
    static std::shared_ptr<Quad> make(Vec3 translation, Vec3 u, Vec3 v, std::shared_ptr<Material> m) {  // mesh.rs:229-246
        auto q = std::make_shared<Quad>();
        Aabb d1 = Aabb::from_extremes(translation, translation + u + v);
        Aabb d2 = Aabb::from_extremes(translation + u, translation + v);
        Vec3 n = cross(u, v);
        Vec3 normal = normalize(n);
        q->translation = translation;
        q->u = u;
        q->v = v;
        q->material = m;
        q->bbox = d1.merge(d2);
        q->normal = normal;
        q->d = dot(normal, translation);
        q->w = n / dot(n, n);
        return q;
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:202-221
        ORC_COUNT(quad_tests);
        float denom = dot(normal, ray.direction);
        if (std::fabs(denom) < 1e-8f) return Hit();
        float t = (d - dot(normal, ray.origin)) / denom;
        if (!ray_t.contains(t)) return Hit();
        Vec3 intersection = ray.get_point(t);
        Vec3 planar = intersection - translation;
        float alpha = dot(w, cross(planar, v));
        float beta = dot(w, cross(u, planar));
        Interval unit(0.0f, 1.0f);  // mesh.rs:248-256
        if (!(unit.contains(alpha) && unit.contains(beta))) return Hit();
        ORC_COUNT(quad_hits);
        return hit_new(ray, t, normal, material.get(), alpha, beta, prim_id);
    }
    const Aabb& bounding_box() const override { return bbox; }
};

// Triangle: NOT in the reference (it has no triangle type and no mesh loader).  Build-defined restatement used as the
// parity target for the triangle configurations: exactly Quad's plane / alpha / beta arithmetic (mesh.rs:202-221,
// :229-246) with Q = v0, u = v1 - v0, v = v2 - v0 and the triangle's domain alpha >= 0, beta >= 0, alpha + beta <= 1.
struct Triangle : Mesh {
    Vec3 q, u, v, normal, w;
    float d = 0;
    std::shared_ptr<Material> material;
    Aabb bbox;
    uint32_t prim_id = 0;
    static std::shared_ptr<Triangle> make(Vec3 v0, Vec3 v1, Vec3 v2, std::shared_ptr<Material> m) {
        auto t = std::make_shared<Triangle>();
        t->q = v0;
        t->u = v1 - v0;
        t->v = v2 - v0;
        Vec3 n = cross(t->u, t->v);
        t->normal = normalize(n);
        t->d = dot(t->normal, t->q);
        t->w = n / dot(n, n);
        t->material = m;
        Vec3 lo(rmin(v0.x, rmin(v1.x, v2.x)), rmin(v0.y, rmin(v1.y, v2.y)), rmin(v0.z, rmin(v1.z, v2.z)));
        Vec3 hi(rmax(v0.x, rmax(v1.x, v2.x)), rmax(v0.y, rmax(v1.y, v2.y)), rmax(v0.z, rmax(v1.z, v2.z)));
        t->bbox = Aabb::from_extremes(lo, hi);
        return t;
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override {
        ORC_COUNT(quad_tests);
        float denom = dot(normal, ray.direction);
        if (std::fabs(denom) < 1e-8f) return Hit();
        float t = (d - dot(normal, ray.origin)) / denom;
        if (!ray_t.contains(t)) return Hit();
        Vec3 intersection = ray.get_point(t);
        Vec3 planar = intersection - q;
        float alpha = dot(w, cross(planar, v));
        float beta = dot(w, cross(u, planar));
        if (!(alpha >= 0.0f && beta >= 0.0f && alpha + beta <= 1.0f)) return Hit();
        ORC_COUNT(quad_hits);
        return hit_new(ray, t, normal, material.get(), alpha, beta, prim_id);
    }
    const Aabb& bounding_box() const override { return bbox; }
};

struct Cube : Mesh {  // mesh.rs:259-322: a World of six quads sharing one material
    World world;
    static std::shared_ptr<Cube> make(Vec3 a, Vec3 b, std::shared_ptr<Material> m, uint32_t first_prim_id) {  // mesh.rs:272-321
        auto c = std::make_shared<Cube>();
        Vec3 mn(rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z));
        Vec3 mx(rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z));
        Vec3 dx = Vec3(1, 0, 0) * (mx.x - mn.x);
        Vec3 dy = Vec3(0, 1, 0) * (mx.y - mn.y);
        Vec3 dz = Vec3(0, 0, 1) * (mx.z - mn.z);
        std::shared_ptr<Quad> q[6];
        q[0] = Quad::make(Vec3(mn.x, mn.y, mx.z), dx, dy, m);
        q[1] = Quad::make(Vec3(mx.x, mn.y, mx.z), -dz, dy, m);
        q[2] = Quad::make(Vec3(mx.x, mx.y, mn.z), -dx, -dy, m);
        q[3] = Quad::make(Vec3(mn.x, mn.y, mn.z), dz, dy, m);
        q[4] = Quad::make(Vec3(mn.x, mx.y, mx.z), dx, -dz, m);
        q[5] = Quad::make(Vec3(mn.x, mn.y, mn.z), dx, dz, m);
        for (int i = 0; i < 6; i++) {
            q[i]->prim_id = first_prim_id + (uint32_t)i;
            c->world.push(q[i]);
        }
        return c;
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override { return world.hit(ray, ray_t); }
    const Aabb& bounding_box() const override { return world.bbox; }
};

struct Translate : Mesh {  // mesh.rs:324-354
    Aabb bbox;
    Vec3 offset;
    MeshP object;
    Translate(MeshP obj, Vec3 off) : bbox(obj->bounding_box().add(off)), offset(off), object(obj) {}
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:331-339
        Ray r(ray.origin - offset, ray.direction, ray.time);
        Hit h = object->hit(r, ray_t);
        if (h.valid) h.point = h.point + offset;
        return h;
    }
    const Aabb& bounding_box() const override { return bbox; }
};

struct RotateY : Mesh {  // mesh.rs:356-432
    MeshP object;
    float sin_theta, cos_theta;
    Aabb bbox;
    RotateY(MeshP obj, float angle) : object(obj) {  // mesh.rs:396-431
        float radians = degrees_to_radians(angle);
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        Aabb b = obj->bounding_box();
        Vec3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    float x = (float)i * b.x.end + (1.0f - (float)i) * b.x.start;
                    float y = (float)j * b.y.end + (1.0f - (float)j) * b.y.start;
                    float z = (float)k * b.z.end + (1.0f - (float)k) * b.z.start;
                    float newx = cos_theta * x + sin_theta * z;
                    float newz = -sin_theta * x + cos_theta * z;
                    Vec3 tester(newx, y, newz);
                    for (int c = 0; c < 3; c++) {
                        mn[c] = rmin(mn[c], tester[c]);
                        mx[c] = rmax(mx[c], tester[c]);
                    }
                }
        bbox = Aabb::from_extremes(mn, mx);
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:364-388
        Vec3 origin = ray.origin;
        Vec3 direction = ray.direction;
        origin[0] = cos_theta * ray.origin[0] - sin_theta * ray.origin[2];
        origin[2] = sin_theta * ray.origin[0] + cos_theta * ray.origin[2];
        direction[0] = cos_theta * ray.direction[0] - sin_theta * ray.direction[2];
        direction[2] = sin_theta * ray.direction[0] + cos_theta * ray.direction[2];
        Hit h = object->hit(Ray(origin, direction, ray.time), ray_t);
        if (h.valid) {
            // the reference reuses the forward rotation and reads the already-overwritten [0]
            h.point[0] = cos_theta * h.point[0] - sin_theta * h.point[2];
            h.point[2] = sin_theta * h.point[0] + cos_theta * h.point[2];
            h.normal[0] = cos_theta * h.normal[0] - sin_theta * h.normal[2];
            h.normal[2] = sin_theta * h.normal[0] + cos_theta * h.normal[2];
        }
        return h;
    }
    const Aabb& bounding_box() const override { return bbox; }
};

// f32::total_cmp ordering key
static inline int32_t total_key(float f) {
    int32_t b;
    std::memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

struct Bvh : Mesh {  // mesh.rs:554-625
    MeshP left, right;
    Aabb bbox;
    static std::shared_ptr<Bvh> from(const std::vector<MeshP>& value) {  // mesh.rs:564-604
        auto node = std::make_shared<Bvh>();
        Aabb bbox;  // Aabb::default(): 0..0, so every node box also contains the origin
        for (const MeshP& m : value) bbox = bbox.merge(m->bounding_box());
        int axis = bbox.longest_axis();
        node->bbox = bbox;
        if (value.size() == 1) {
            node->left = value[0];
            node->right = value[0];
        } else if (value.size() == 2) {
            node->left = value[0];
            node->right = value[1];
        } else {
            std::vector<MeshP> v(value);
            std::stable_sort(v.begin(), v.end(), [axis](const MeshP& a, const MeshP& b) {
                return total_key(a->bounding_box()[axis].start) < total_key(b->bounding_box()[axis].start);
            });
            size_t mid = v.size() / 2;
            node->left = from(std::vector<MeshP>(v.begin(), v.begin() + mid));
            node->right = from(std::vector<MeshP>(v.begin() + mid, v.end()));
        }
        return node;
    }
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:608-620
        if (!bbox.hit(ray, ray_t)) return Hit();
        Hit l = left->hit(ray, ray_t);
        if (l.valid) {
            Hit r = right->hit(ray, Interval(ray_t.start, l.distance));
            return r.valid ? r : l;
        }
        return right->hit(ray, ray_t);
    }
    const Aabb& bounding_box() const override { return bbox; }
};

struct ConstantMedium : Mesh {  // mesh.rs:627-696
    MeshP boundary;
    std::shared_ptr<Isotropic> material;
    float density;
    uint32_t prim_id = 0;
    Hit hit(const Ray& ray, const Interval& ray_t) const override {  // mesh.rs:634-675
        ORC_COUNT(medium_tests);
        Hit hit1 = boundary->hit(ray, UNIVERSE);
        if (!hit1.valid) return Hit();
        Hit hit2 = boundary->hit(ray, Interval(hit1.distance + 0.0001f, INF));
        if (!hit2.valid) return Hit();
        if (hit1.distance < ray_t.start) hit1.distance = ray_t.start;
        if (hit2.distance > ray_t.end) hit2.distance = ray_t.end;
        if (hit1.distance >= hit2.distance) return Hit();
        if (hit1.distance < 0.0f) hit1.distance = 0.0f;
        float ray_length = length(ray.direction);
        float distance_inside_boundary = (hit2.distance - hit1.distance) * ray_length;
        float dens = -1.0f / density;
        float hit_distance = dens * std::log(tl_rng->gen_f32());
        if (hit_distance > distance_inside_boundary) return Hit();
        float distance = hit1.distance + hit_distance / ray_length;
        Hit h;
        h.point = ray.get_point(distance);
        h.distance = distance;
        h.normal = Vec3(1, 0, 0);
        h.front_face = true;
        h.material = material.get();
        h.u = 0;
        h.v = 0;
        h.prim_id = prim_id;
        h.valid = true;
        return h;
    }
    const Aabb& bounding_box() const override { return boundary->bounding_box(); }
};

// ----------------------------------------------------------------------------------------------
// camera.rs
// ----------------------------------------------------------------------------------------------
struct CameraConfig {  // camera.rs:13-43
    uint32_t width = 100, height = 100;
    uint32_t samples_per_pixel = 10, max_depth = 10;
    float vfov = 90.0f;
    Vec3 lookfrom = Vec3(0, 0, 0), lookat = Vec3(0, 0, -1), vup = Vec3(0, 1, 0);
    float defocus_angle = 0.0f, focus_dist = 10.0f;
    Vec3 background = Vec3(0.7f, 0.8f, 1.0f);
};

// bevy_color 0.14.2 Srgba::gamma_function_inverse + to_u8_array_no_alpha (camera.rs:101)
static inline uint8_t srgb_u8(float value) {
    float s;
    if (value <= 0.0f)
        s = value;
    else if (value <= 0.0031308f)
        s = value * 12.92f;
    else
        s = (1.055f * std::pow(value, 1.0f / 2.4f)) - 0.055f;
    float q = std::round(rclamp(s, 0.0f, 1.0f) * 255.0f);
    if (!(q == q)) return 0;
    return (uint8_t)q;
}

struct Camera {  // camera.rs:45-169
    CameraConfig config;
    Vec3 center, pixel00_loc, pixel_delta_u, pixel_delta_v;
    float pixel_samples_scale;
    Vec3 defocus_disk_u, defocus_disk_v;
    explicit Camera(const CameraConfig& cfg) : config(cfg) {  // camera.rs:57-96
        float aspect_ratio = (float)cfg.width / (float)cfg.height;
        center = cfg.lookfrom;
        float theta = degrees_to_radians(cfg.vfov);
        float h = std::tan(theta / 2.0f);
        float viewport_height = 2.0f * h * cfg.focus_dist;
        float viewport_width = viewport_height * aspect_ratio;
        Vec3 w = normalize(cfg.lookfrom - cfg.lookat);
        Vec3 u = cross(cfg.vup, w);  // not normalised
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
    Vec3 defocus_disk_sample(Rng& rng) const {  // camera.rs:165-168
        Vec3 p = random_vec_in_unit_disk(rng);
        return center + (p.x * defocus_disk_u) + (p.y * defocus_disk_v);
    }
    Ray get_ray(Rng& rng, uint32_t x, uint32_t y) const {  // camera.rs:147-163
        float ox = rng.gen_f32() - 0.5f;
        float oy = rng.gen_f32() - 0.5f;
        Vec3 pixel_sample = pixel00_loc + (((float)x + ox) * pixel_delta_u) + (((float)y + oy) * pixel_delta_v);
        Vec3 ray_origin = (config.defocus_angle <= 0.0f) ? center : defocus_disk_sample(rng);
        Vec3 dir = pixel_sample - ray_origin;  // evaluated before the time draw (argument order)
        float time = rng.gen_f32();
        return Ray(ray_origin, dir, time);
    }
    // pixel-centre ray, no jitter, no lens, fixed time: the deterministic ray set of the hit-parity tests
    Ray center_ray(uint32_t x, uint32_t y, float time) const {
        Vec3 pixel_sample = pixel00_loc + (((float)x + 0.0f) * pixel_delta_u) + (((float)y + 0.0f) * pixel_delta_v);
        return Ray(center, pixel_sample - center, time);
    }
    Vec3 ray_color(const Ray& ray, const Mesh& world, uint32_t depth) const {  // camera.rs:124-145
        if (depth <= 0) {
            ORC_COUNT(depth_exhausted);
            return Vec3(0, 0, 0);
        }
        ORC_COUNT(rays);
        Hit hit = world.hit(ray, Interval(0.001f, INF));
        if (hit.valid) {
            Vec3 color_from_emission = hit.material->emitted(hit.u, hit.v, hit.point);
            Scatter sc = hit.material->scatter(ray, hit);
            if (sc.some) {
                Vec3 color_from_scatter = sc.attenuation * ray_color(sc.scattered, world, depth - 1);
                return color_from_scatter + color_from_emission;
            }
            return color_from_emission;
        }
        ORC_COUNT(misses);
        return config.background;
    }
};

}  // namespace orc

// ================================================================================================
// C API (ctypes-friendly).  Handles are heap boxes around shared_ptr.
// ================================================================================================
using namespace orc;

struct orc_tex { std::shared_ptr<Texture> p; };
struct orc_mat { std::shared_ptr<Material> p; };
struct orc_mesh { MeshP p; };
struct orc_camera { std::shared_ptr<Camera> p; };

struct orc_ray_t { float o[3]; float d[3]; float time; };
struct orc_hit_t {
    uint32_t prim_id;  // 0xFFFFFFFF = miss
    float t;
    float n[3];
    float uv[2];
    uint32_t front_face;
    float p[3];
};
struct orc_camera_cfg_t {
    uint32_t width, height, samples_per_pixel, max_depth;
    float vfov;
    float lookfrom[3], lookat[3], vup[3];
    float defocus_angle, focus_dist;
    float background[3];
};

static Counters g_counters;
static std::atomic_flag g_counters_lock = ATOMIC_FLAG_INIT;
static void flush_counters() {
    while (g_counters_lock.test_and_set(std::memory_order_acquire)) {}
    g_counters.add(tl_cnt);
    g_counters_lock.clear(std::memory_order_release);
    tl_cnt = Counters();
}

static inline Vec3 v3(const float* f) { return Vec3(f[0], f[1], f[2]); }

extern "C" {

orc_tex* orc_tex_solid(float r, float g, float b) { return new orc_tex{std::make_shared<SolidTexture>(Vec3(r, g, b))}; }
orc_tex* orc_tex_checker(float scale, orc_tex* even, orc_tex* odd) {
    auto t = std::make_shared<CheckerTexture>();
    t->scale = scale;
    t->even = even->p;
    t->odd = odd->p;
    return new orc_tex{t};
}
orc_tex* orc_tex_image(const uint8_t* rgb, uint32_t w, uint32_t h) {
    auto t = std::make_shared<ImageTexture>();
    t->rgb.assign(rgb, rgb + (size_t)w * h * 3);
    t->width = w;
    t->height = h;
    return new orc_tex{t};
}
orc_tex* orc_tex_noise(float scale, const float* values, const uint32_t* px, const uint32_t* py, const uint32_t* pz) {
    auto t = std::make_shared<NoiseTexture>();
    t->scale = scale;
    for (int i = 0; i < 256; i++) {
        t->perlin.values[i] = Vec3(values[3 * i], values[3 * i + 1], values[3 * i + 2]);
        t->perlin.perm_x[i] = px[i];
        t->perlin.perm_y[i] = py[i];
        t->perlin.perm_z[i] = pz[i];
    }
    return new orc_tex{t};
}
void orc_tex_value(orc_tex* t, float u, float v, const float* p, float* out) {
    Vec3 c = t->p->value(u, v, v3(p));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void orc_tex_free(orc_tex* t) { delete t; }

orc_mat* orc_mat_lambertian(orc_tex* t) { auto m = std::make_shared<Lambertian>(); m->texture = t->p; return new orc_mat{m}; }
orc_mat* orc_mat_metal(orc_tex* t, float roughness) { auto m = std::make_shared<Metal>(); m->texture = t->p; m->roughness = roughness; return new orc_mat{m}; }
orc_mat* orc_mat_dielectric(float ri) { auto m = std::make_shared<Dielectric>(); m->refraction_index = ri; return new orc_mat{m}; }
orc_mat* orc_mat_diffuse_light(orc_tex* t) { auto m = std::make_shared<DiffuseLight>(); m->texture = t->p; return new orc_mat{m}; }
orc_mat* orc_mat_isotropic(orc_tex* t) { auto m = std::make_shared<Isotropic>(); m->texture = t->p; return new orc_mat{m}; }
void orc_mat_free(orc_mat* m) { delete m; }

orc_mesh* orc_sphere_stationary(const float* c, float r, orc_mat* m, uint32_t prim_id) {
    if (!(r >= 0.0f)) return nullptr;  // assert!(radius >= 0.) mesh.rs:148
    auto s = Sphere::stationary(v3(c), r, m->p);
    s->prim_id = prim_id;
    return new orc_mesh{s};
}
orc_mesh* orc_sphere_moving(const float* from, const float* to, float r, orc_mat* m, uint32_t prim_id) {
    if (!(r >= 0.0f)) return nullptr;  // mesh.rs:162
    auto s = Sphere::moving(v3(from), v3(to), r, m->p);
    s->prim_id = prim_id;
    return new orc_mesh{s};
}
orc_mesh* orc_quad(const float* q, const float* u, const float* v, orc_mat* m, uint32_t prim_id) {
    auto s = Quad::make(v3(q), v3(u), v3(v), m->p);
    s->prim_id = prim_id;
    return new orc_mesh{s};
}
orc_mesh* orc_triangle(const float* a, const float* b, const float* c, orc_mat* m, uint32_t prim_id) {
    auto t = Triangle::make(v3(a), v3(b), v3(c), m->p);
    t->prim_id = prim_id;
    return new orc_mesh{t};
}
orc_mesh* orc_cube(const float* a, const float* b, orc_mat* m, uint32_t first_prim_id) {
    return new orc_mesh{Cube::make(v3(a), v3(b), m->p, first_prim_id)};
}
orc_mesh* orc_translate(orc_mesh* obj, const float* off) { return new orc_mesh{std::make_shared<Translate>(obj->p, v3(off))}; }
orc_mesh* orc_rotate_y(orc_mesh* obj, float angle) { return new orc_mesh{std::make_shared<RotateY>(obj->p, angle)}; }
orc_mesh* orc_constant_medium(orc_mesh* boundary, float density, orc_tex* tex, uint32_t prim_id) {
    auto cm = std::make_shared<ConstantMedium>();
    cm->boundary = boundary->p;
    cm->density = density;
    cm->material = std::make_shared<Isotropic>();
    cm->material->texture = tex->p;
    cm->prim_id = prim_id;
    return new orc_mesh{cm};
}
orc_mesh* orc_world_new() { return new orc_mesh{std::make_shared<World>()}; }
int orc_world_push(orc_mesh* world, orc_mesh* m) {
    World* w = dynamic_cast<World*>(world->p.get());
    if (!w) return -1;
    w->push(m->p);
    return 0;
}
int orc_world_push_trimesh(orc_mesh* world, const float* vertices, uint32_t n_vertices, const uint32_t* indices, uint32_t n_tris,
                           orc_mat* m, uint32_t first_prim_id) {
    World* w = dynamic_cast<World*>(world->p.get());
    if (!w) return -1;
    for (uint32_t i = 0; i < n_tris; i++) {
        uint32_t a = indices[3 * i], b = indices[3 * i + 1], c = indices[3 * i + 2];
        if (a >= n_vertices || b >= n_vertices || c >= n_vertices) return -1;
        auto t = Triangle::make(v3(vertices + 3 * a), v3(vertices + 3 * b), v3(vertices + 3 * c), m->p);
        t->prim_id = first_prim_id + i;
        w->push(t);
    }
    return 0;
}
// Bvh::from(&world)   (main.rs:577).  An empty world recurses without bound in the reference; refuse it.
orc_mesh* orc_bvh_from(orc_mesh* world) {
    World* w = dynamic_cast<World*>(world->p.get());
    if (!w || w->meshes.empty()) return nullptr;
    return new orc_mesh{Bvh::from(w->meshes)};
}
void orc_mesh_free(orc_mesh* m) { delete m; }
void orc_mesh_bbox(orc_mesh* m, float* out6) {
    const Aabb& b = m->p->bounding_box();
    out6[0] = b.x.start; out6[1] = b.x.end; out6[2] = b.y.start; out6[3] = b.y.end; out6[4] = b.z.start; out6[5] = b.z.end;
}

// DFS (left then right) list of leaf prim ids under a Bvh tree, duplicates of the len==1 case included.
// Cubes contribute their six quads in push order.  Returns the count (fills at most cap).
static void dfs_leaves(const Mesh* m, std::vector<uint32_t>& out) {
    if (auto b = dynamic_cast<const Bvh*>(m)) {
        dfs_leaves(b->left.get(), out);
        dfs_leaves(b->right.get(), out);
    } else if (auto w = dynamic_cast<const World*>(m)) {
        for (auto& c : w->meshes) dfs_leaves(c.get(), out);
    } else if (auto c = dynamic_cast<const Cube*>(m)) {
        dfs_leaves(&c->world, out);
    } else if (auto t = dynamic_cast<const Translate*>(m)) {
        dfs_leaves(t->object.get(), out);
    } else if (auto r = dynamic_cast<const RotateY*>(m)) {
        dfs_leaves(r->object.get(), out);
    } else if (auto s = dynamic_cast<const Sphere*>(m)) {
        out.push_back(s->prim_id);
    } else if (auto q = dynamic_cast<const Quad*>(m)) {
        out.push_back(q->prim_id);
    } else if (auto t3 = dynamic_cast<const Triangle*>(m)) {
        out.push_back(t3->prim_id);
    } else if (auto cm = dynamic_cast<const ConstantMedium*>(m)) {
        out.push_back(cm->prim_id);
    }
}
size_t orc_bvh_leaf_order(orc_mesh* m, uint32_t* out, size_t cap) {
    std::vector<uint32_t> v;
    dfs_leaves(m->p.get(), v);
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}
// Pre-order dump of the Bvh node boxes (x0,x1,y0,y1,z0,z1 per node); returns node count.
static void dfs_nodes(const Mesh* m, std::vector<float>& out) {
    if (auto b = dynamic_cast<const Bvh*>(m)) {
        const Aabb& a = b->bbox;
        float f[6] = {a.x.start, a.x.end, a.y.start, a.y.end, a.z.start, a.z.end};
        out.insert(out.end(), f, f + 6);
        dfs_nodes(b->left.get(), out);
        if (b->right.get() != b->left.get()) dfs_nodes(b->right.get(), out);
    }
}
size_t orc_bvh_node_boxes(orc_mesh* m, float* out, size_t cap_nodes) {
    std::vector<float> v;
    dfs_nodes(m->p.get(), v);
    size_t n = v.size() / 6;
    for (size_t i = 0; i < n && i < cap_nodes; i++) std::memcpy(out + 6 * i, &v[6 * i], 24);
    return n;
}

orc_camera* orc_camera_new(const orc_camera_cfg_t* c) {
    CameraConfig cfg;
    cfg.width = c->width; cfg.height = c->height;
    cfg.samples_per_pixel = c->samples_per_pixel; cfg.max_depth = c->max_depth;
    cfg.vfov = c->vfov;
    cfg.lookfrom = v3(c->lookfrom); cfg.lookat = v3(c->lookat); cfg.vup = v3(c->vup);
    cfg.defocus_angle = c->defocus_angle; cfg.focus_dist = c->focus_dist;
    cfg.background = v3(c->background);
    return new orc_camera{std::make_shared<Camera>(cfg)};
}
void orc_camera_free(orc_camera* c) { delete c; }
// center, pixel00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v (6 x vec3) + pixel_samples_scale
void orc_camera_constants(orc_camera* c, float* out19) {
    const Camera& cam = *c->p;
    const Vec3* vs[6] = {&cam.center, &cam.pixel00_loc, &cam.pixel_delta_u, &cam.pixel_delta_v, &cam.defocus_disk_u, &cam.defocus_disk_v};
    for (int i = 0; i < 6; i++) { out19[3 * i] = vs[i]->x; out19[3 * i + 1] = vs[i]->y; out19[3 * i + 2] = vs[i]->z; }
    out19[18] = cam.pixel_samples_scale;
}
// Pixel-centre primary rays (row-major, y outer), fixed time.
void orc_camera_center_rays(orc_camera* c, float time, orc_ray_t* out) {
    const Camera& cam = *c->p;
    for (uint32_t y = 0; y < cam.config.height; y++)
        for (uint32_t x = 0; x < cam.config.width; x++) {
            Ray r = cam.center_ray(x, y, time);
            orc_ray_t& o = out[(size_t)y * cam.config.width + x];
            o.o[0] = r.origin.x; o.o[1] = r.origin.y; o.o[2] = r.origin.z;
            o.d[0] = r.direction.x; o.d[1] = r.direction.y; o.d[2] = r.direction.z;
            o.time = r.time;
        }
}

// Closest hit of each ray in [tmin, tmax)  (the Mesh::hit contract, mesh.rs:17-21).
// `seed` feeds the generator ConstantMedium::hit draws from (mesh.rs:659); other meshes ignore it.
//   rng_kind 0: one ChaCha12 stream for the whole call;  1: the mirror stream keyed (seed, ray index, 0) per ray, which is
//   what rtx_trace uses, so that media can be compared hit for hit.
static void trace_impl(orc_mesh* m, const orc_ray_t* rays, size_t n, float tmin, float tmax, uint64_t seed, int rng_kind, orc_hit_t* out) {
    Rng rng;
    rng.seed_faithful(seed, 0x7261795fu);
    tl_rng = &rng;
    for (size_t i = 0; i < n; i++) {
        if (rng_kind == 1) rng.seed_mirror(seed, (uint32_t)i, 0);
        Ray r(Vec3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), Vec3(rays[i].d[0], rays[i].d[1], rays[i].d[2]), rays[i].time);
        Hit h = m->p->hit(r, Interval(tmin, tmax));
        orc_hit_t& o = out[i];
        if (!h.valid) {
            std::memset(&o, 0, sizeof(o));
            o.prim_id = 0xFFFFFFFFu;
            continue;
        }
        o.prim_id = h.prim_id;
        o.t = h.distance;
        o.n[0] = h.normal.x; o.n[1] = h.normal.y; o.n[2] = h.normal.z;
        o.uv[0] = h.u; o.uv[1] = h.v;
        o.front_face = h.front_face ? 1u : 0u;
        o.p[0] = h.point.x; o.p[1] = h.point.y; o.p[2] = h.point.z;
    }
    tl_rng = nullptr;
    flush_counters();
}
void orc_trace(orc_mesh* m, const orc_ray_t* rays, size_t n, float tmin, float tmax, uint64_t seed, orc_hit_t* out) {
    trace_impl(m, rays, n, tmin, tmax, seed, 0, out);
}
void orc_trace_mirror(orc_mesh* m, const orc_ray_t* rays, size_t n, float tmin, float tmax, uint64_t seed, orc_hit_t* out) {
    trace_impl(m, rays, n, tmin, tmax, seed, 1, out);
}

// Camera::render (camera.rs:98-122) for the sample index range [sample_begin, sample_end).
//   accum_out (nullable): 3*W*H f32, per pixel the SUM of the range's radiance samples in sample order
//   rgb8_out  (nullable): 3*W*H u8, sRGB quantisation of (sum * 1/samples_per_pixel)  -- meaningful when the
//                         range is the whole 0..samples_per_pixel
//   rng_kind 0 = faithful (ChaCha12 keyed (seed,pixel); rejection samplers), 1 = mirror (PCG keyed (seed,pixel,sample))
//   threads <= 0: hardware concurrency.  Pixels are handed out dynamically in row chunks (rayon stand-in).
void orc_render(orc_camera* c, orc_mesh* m, uint64_t seed, int rng_kind, uint32_t sample_begin, uint32_t sample_end,
                int threads, float* accum_out, uint8_t* rgb8_out) {
    const Camera& cam = *c->p;
    const Mesh& world = *m->p;
    const uint32_t W = cam.config.width, H = cam.config.height;
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    std::atomic<uint32_t> next_chunk(0);
    const uint32_t chunk = 16;  // pixels per grab
    const uint64_t total = (uint64_t)W * H;
    const uint32_t nchunks = (uint32_t)((total + chunk - 1) / chunk);
    auto worker = [&]() {
        Rng rng;
        tl_rng = &rng;
        for (;;) {
            uint32_t ci = next_chunk.fetch_add(1);
            if (ci >= nchunks) break;
            uint64_t p0 = (uint64_t)ci * chunk, p1 = std::min<uint64_t>(p0 + chunk, total);
            for (uint64_t p = p0; p < p1; p++) {
                uint32_t x = (uint32_t)(p % W), y = (uint32_t)(p / W);
                if (rng_kind == 0) rng.seed_faithful(seed, p * 0x100000000ull + sample_begin);
                // camera.rs:112-117: Vec3 sum over samples, in order, starting from ZERO
                Vec3 color(0, 0, 0);
                for (uint32_t s = sample_begin; s < sample_end; s++) {
                    if (rng_kind == 1) rng.seed_mirror(seed, (uint32_t)p, s);
                    Ray ray = cam.get_ray(rng, x, y);
                    ORC_COUNT(samples);
                    color = color + cam.ray_color(ray, world, cam.config.max_depth);
                }
                if (accum_out) {
                    accum_out[3 * p] = color.x; accum_out[3 * p + 1] = color.y; accum_out[3 * p + 2] = color.z;
                }
                if (rgb8_out) {
                    Vec3 px = cam.pixel_samples_scale * color;  // camera.rs:119
                    rgb8_out[3 * p] = srgb_u8(px.x); rgb8_out[3 * p + 1] = srgb_u8(px.y); rgb8_out[3 * p + 2] = srgb_u8(px.z);
                }
            }
        }
        tl_rng = nullptr;
        flush_counters();
    };
    if (threads == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; i++) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
}

// sRGB quantisation of a linear value (camera.rs:101)
uint8_t orc_srgb_u8(float linear) { return srgb_u8(linear); }
void orc_resolve(const float* accum, size_t npix, float scale, uint8_t* rgb8) {
    for (size_t i = 0; i < npix * 3; i++) rgb8[i] = srgb_u8(scale * accum[i]);
}

// Material known-answer helpers
float orc_reflectance(float cosine, float ri) { return reflectance(cosine, ri); }
void orc_reflect(const float* v, const float* n, float* out) { Vec3 r = reflect(v3(v), v3(n)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_refract(const float* uv, const float* n, float ratio, float* out) { Vec3 r = refract(v3(uv), v3(n), ratio); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
// One scatter call on a material for a given ray/hit (normal as stored, i.e. already flipped); returns 1 if Some.
int orc_scatter(orc_mat* m, const orc_ray_t* ray, const float* point, const float* normal, float u, float v, uint64_t seed,
                int rng_kind, float* atten_out, orc_ray_t* scattered_out, float* emitted_out) {
    Rng rng;
    if (rng_kind == 0) rng.seed_faithful(seed, 1); else rng.seed_mirror(seed, 0, 0);
    tl_rng = &rng;
    Ray r(Vec3(ray->o[0], ray->o[1], ray->o[2]), Vec3(ray->d[0], ray->d[1], ray->d[2]), ray->time);
    Hit h;
    h.point = v3(point); h.normal = v3(normal); h.u = u; h.v = v; h.front_face = true; h.valid = true; h.material = m->p.get();
    Vec3 e = m->p->emitted(u, v, h.point);
    Scatter s = m->p->scatter(r, h);
    tl_rng = nullptr;
    emitted_out[0] = e.x; emitted_out[1] = e.y; emitted_out[2] = e.z;
    if (!s.some) return 0;
    atten_out[0] = s.attenuation.x; atten_out[1] = s.attenuation.y; atten_out[2] = s.attenuation.z;
    scattered_out->o[0] = s.scattered.origin.x; scattered_out->o[1] = s.scattered.origin.y; scattered_out->o[2] = s.scattered.origin.z;
    scattered_out->d[0] = s.scattered.direction.x; scattered_out->d[1] = s.scattered.direction.y; scattered_out->d[2] = s.scattered.direction.z;
    scattered_out->time = s.scattered.time;
    return 1;
}
// Raw generator access (tests of the float conversions and of the mirror stream)
void orc_rng_mirror_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t* out, size_t n) {
    Rng r; r.seed_mirror(seed, pixel, sample);
    for (size_t i = 0; i < n; i++) out[i] = r.next_u32();
}
void orc_rng_faithful_f32(uint64_t seed, float* out01, float* out_m11, size_t n) {
    Rng r; r.seed_faithful(seed, 0);
    for (size_t i = 0; i < n; i++) { out01[i] = r.gen_f32(); out_m11[i] = r.gen_range(-1.0f, 1.0f); }
}

size_t orc_counters_size() { return sizeof(Counters) / 8; }
void orc_counters_get(uint64_t* out) { flush_counters(); std::memcpy(out, &g_counters, sizeof(Counters)); }
void orc_counters_reset() { flush_counters(); g_counters = Counters(); }

}  // extern "C"
