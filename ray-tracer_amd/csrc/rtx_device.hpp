// Device-side arithmetic of the render path: ray / primitive tests, the threaded closest-hit walk, hit records,
// textures, materials and the random stream.  Everything that decides WHICH primitive is hit and WHERE follows
// the reference's expressions operation for operation (this translation unit is compiled with
// -ffp-contract=off and correctly rounded divide / sqrt), so those results are bit-identical to the Rust code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtx.h"

namespace rtxd {

enum : uint32_t { ENT_BOX = 0, ENT_SPHERE = 1, ENT_QUAD = 2, ENT_TRI = 3, ENT_XF_PUSH = 4, ENT_XF_POP = 5, ENT_MED_BEGIN = 6, ENT_MED_END = 7 };
constexpr uint32_t ENT_INST = 8;  // | ENT_SPHERE / ENT_QUAD / ENT_TRI: primitive of an instance placed in the enclosing frame's tree (accel.hpp)
__device__ __forceinline__ bool is_bracket_kind(uint32_t kind) { return (kind & 12u) == 4u; }  // kind = word 7 & 15 of a non-box entry
constexpr uint32_t XF_NONE = 0xFFFFFFFFu;
// Word 7 of the entry a lane stands at is also that lane's walk state (see accel.hpp): 0 = box, ENT_END = walk
// complete, anything else = primitive.  ST_IDLE marks a lane without a path.
constexpr uint32_t ENT_END = 0xFFFFFFFEu;
constexpr uint32_t ST_IDLE = 0xFFFFFFFFu;
constexpr uint32_t ENT_BYTES = 32u;
constexpr uint32_t NO_HIT = 0xFFFFFFFFu;
constexpr float PI_F = 3.1415926535897932385f;  // utils.rs:4

struct DevImage {
    const uint8_t* rgb;
    uint32_t width, height;
};

struct DevScene {
    const uint4* ents[2];      // [RTX_TRAVERSAL_FAST], [RTX_TRAVERSAL_STRICT]: 2 x uint4 per entry
    const uint32_t* rank[2];   // per entry tie-break rank
    const uint4* shade[2];     // per entry, 2 x uint4: what a hit at that entry is shaded with, resolved on the host:
                               //   (material kind | 256 if the colour below is final | 512 if the texture reads uv,
                               //    param bits, texture, 0) (r, g, b bits, 0)
                               //   or, with 1024 set (checker of two plain colours, material without a parameter):
                               //   (kind | 1024, scale bits, even r, even g) (even b, odd r, odd g, odd b)
    uint32_t n_ents[2];
    const float4* sph;         // 2 per sphere: (c0.xyz, r) (dc.xyz, material bits)
    const uint32_t* sph_id;
    const float4* planar;      // 5 per quad or triangle: (Q.xyz, d) (u.xyz, material bits) (v.xyz, 0) (n.xyz, 0) (w.xyz, 0)
    const uint32_t* planar_id; // quads first, then triangles
    uint32_t n_quads;
    const uint4* mats;         // (kind, texture, param bits, flags)   flags bit0: texture reads uv
    const uint4* texs;         // 2 per texture: (kind, scale bits, a, b) (r, g, b bits, 0)
    const DevImage* images;
    const float* perlin;       // per table set: 768 floats (values) + 3 * 256 uint32 (perm x, y, z)
    // The packed form of the fast array (16-byte slots, half-precision boxes; see accel.hpp), null when the scene has
    // transform / medium brackets.  The render kernels walk it instead of ents[RTX_TRAVERSAL_FAST].
    const uint4* pk;
    uint32_t pk_slots;         // terminator included
    uint32_t pk_half;          // 1: boxes are one slot of halfs; 0: two slots of floats
    const uint32_t* pk_rank;   // per slot: tie-break rank of the primitive whose entry starts there
    const uint4* pk_shade;     // per slot, 2 x uint4 in the layout of `shade`; null when the array is too large for LDS
    const uint4* mat_recs;     // 2 x uint4 per material: the record `shade` holds for an entry of that material
    // The wide form of the fast array (four-child nodes of 128 bytes + five-slot primitive records; see accel.hpp), null
    // unless the scene is bracket-free and too large for LDS.  The render kernels then walk it instead of `pk`.
    const uint4* wnodes;       // 8 x uint4 per node, root first
    const uint4* wprims;       // 5 x uint4 per primitive
    const uint32_t* wrank;     // per slot of wprims: tie-break rank of the primitive whose record starts there
    uint32_t w_levels;         // nodes on the longest path from the root = groups a walk can have pending + 1
    uint32_t w_first;          // the first w_first records of wprims stand outside the tree: every walk tests them, then visits the root
    // Scenes with a ConstantMedium inside another one's boundary: where a walk parks the boundary-query state of the
    // enclosing media (layout: MedStack), null otherwise.  One record set per lane of a launch: launches over one
    // scene copy that overlap in time would share it.
    uint32_t* med_stack;
    uint32_t med_lanes;        // lanes it has room for
};

struct Ray {
    float ox, oy, oz;
    float dx, dy, dz;
    float time;
};

__device__ __forceinline__ float as_f(uint32_t u) { return __uint_as_float(u); }

// ---------------------------------------------------------------------------------------------- random stream
// 32-bit PCG (RXS-M-XS output) with a per-stream increment.  The stream is a pure function of
// (seed, pixel index, sample index); draw k of a stream is the k-th step from that state.
// The 64-bit seed enters as two words k0, k1, each a hash of both of its halves (seed_word on the host).  State and
// increment come from two separate hash chains over (key, pixel, sample), so that a seed does not merely permute the
// stream set over the pixels and state and increment are not functions of one 32-bit value.
struct Rng {
    uint32_t state, inc;
};
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__device__ __forceinline__ Rng rng_init(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample) {
    Rng r;
    r.state = pcg_hash(pcg_hash(k0 + pixel) + sample * 0x9E3779B9u);
    r.inc = (pcg_hash(pcg_hash(k1 ^ pixel) + sample * 0x85EBCA6Bu) << 1) | 1u;
    return r;
}
__device__ __forceinline__ uint32_t rng_u32(Rng& r) {
    r.state = r.state * 747796405u + r.inc;
    uint32_t s = r.state;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
// rand 0.8.5 `gen::<f32>()`: the 24 high bits scaled by 2^-24, in [0,1)
__device__ __forceinline__ float rng_f32(Rng& r) { return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f); }

// Uniform direction on the unit sphere from two draws: the distribution of utils.rs:37-39
// (normalised rejection sample) without the rejection loop.
__device__ __forceinline__ void random_unit_vec(Rng& r, float& x, float& y, float& z) {
    float u1 = rng_f32(r);
    float u2 = rng_f32(r);
    z = 1.0f - 2.0f * u1;
    float rr = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - z * z));  // samplers need no correctly rounded root
    x = rr * __builtin_amdgcn_cosf(u2);  // hardware cos takes revolutions: cos(2*pi*u2)
    y = rr * __builtin_amdgcn_sinf(u2);
}
// Uniform point in the unit disk (utils.rs:28-35 without the rejection loop).
__device__ __forceinline__ void random_in_unit_disk(Rng& r, float& x, float& y) {
    float u1 = rng_f32(r);
    float u2 = rng_f32(r);
    float rr = __builtin_amdgcn_sqrtf(u1);
    x = rr * __builtin_amdgcn_cosf(u2);
    y = rr * __builtin_amdgcn_sinf(u2);
}

// ---------------------------------------------------------------------------------------------- glam-order helpers
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx) + (ay * by) + (az * bz);
}

// ---------------------------------------------------------------------------------------------- closest-hit walk
// How a walk evaluates its box tests.
//   SLAB_STRICT  Aabb::hit exactly as the reference evaluates it (ordered select, (bound - origin) * (1 / d)).
//   SLAB_ROBUST  same products, min / max instead of the ordered select: conservative for any ray (fast rtx_trace).
//   SLAB_FMA     bound * inv + (-origin * inv) in one fused operation per plane, inv from the hardware reciprocal.
//                Its absolute error is at most ~1.2e-7 * (|origin| + |hit point|) along an axis, which the fast boxes'
//                padding covers as long as ray origins stay within the scene's coordinate range (render kernels).
enum : int { SLAB_STRICT = 0, SLAB_ROBUST = 1, SLAB_FMA = 2 };

struct RayConst {  // per-ray values the tests reuse (same expressions the reference re-evaluates per call)
    float ix, iy, iz;  // 1.0 / direction  (mesh.rs:517)
    float a;           // direction.length_squared()  (mesh.rs:116)
    float nx, ny, nz;  // SLAB_FMA only: -(origin * inv)
};
template <int SLAB>
__device__ __forceinline__ RayConst ray_const(const Ray& r) {
    RayConst c;
    if (SLAB == SLAB_FMA) {
        // A component of exactly 0 would give inv = inf and -(origin * inv) = -+inf, and bound * inf + -inf is NaN for
        // one plane only: the axis would cull boxes that straddle the ray.  Components below 1e-22 in magnitude are
        // therefore taken as +-1e-22: the axis then yields plane distances of +-1e22 x (bound - origin), far beyond any
        // interval end when the origin is inside the slab and culling when it is outside, which is the exact answer,
        // and bound * inv stays finite for coordinates up to 1e12.
        const float tiny = 1e-22f;
        const float dx = fabsf(r.dx) < tiny ? copysignf(tiny, r.dx) : r.dx;
        const float dy = fabsf(r.dy) < tiny ? copysignf(tiny, r.dy) : r.dy;
        const float dz = fabsf(r.dz) < tiny ? copysignf(tiny, r.dz) : r.dz;
        c.ix = __builtin_amdgcn_rcpf(dx);
        c.iy = __builtin_amdgcn_rcpf(dy);
        c.iz = __builtin_amdgcn_rcpf(dz);
        c.nx = -(r.ox * c.ix);
        c.ny = -(r.oy * c.iy);
        c.nz = -(r.oz * c.iz);
    } else {
        c.ix = 1.0f / r.dx;
        c.iy = 1.0f / r.dy;
        c.iz = 1.0f / r.dz;
        c.nx = c.ny = c.nz = 0.0f;
    }
    c.a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
    return c;
}

// Aabb::hit (mesh.rs:512-538).  Checking end <= start once after the three axes is equivalent to the reference's
// per-axis early return: start only grows and end only shrinks (NaN operands are ignored by max/min).
//  strict: the reference's ordered select `if t0 < t1 {(t0,t1)} else {(t1,t0)}`, which matters when 0 * inf = NaN.
//  others: min / max of the two plane distances.  Identical for ordinary values; a NaN (ray inside a bounding plane and
//          parallel to it, or inf - inf in the fused form) drops out of min / max, i.e. that axis does not cull.
template <int SLAB>
__device__ __forceinline__ bool slab_hit(const uint4& ea, const uint4& eb, const Ray& r, const RayConst& rc, float tmin, float tmax) {
    float x0, x1, y0, y1, z0, z1;
    if (SLAB == SLAB_FMA) {
        x0 = __builtin_fmaf(as_f(ea.x), rc.ix, rc.nx);
        x1 = __builtin_fmaf(as_f(ea.y), rc.ix, rc.nx);
        y0 = __builtin_fmaf(as_f(ea.z), rc.iy, rc.ny);
        y1 = __builtin_fmaf(as_f(ea.w), rc.iy, rc.ny);
        z0 = __builtin_fmaf(as_f(eb.x), rc.iz, rc.nz);
        z1 = __builtin_fmaf(as_f(eb.y), rc.iz, rc.nz);
    } else {
        x0 = (as_f(ea.x) - r.ox) * rc.ix;
        x1 = (as_f(ea.y) - r.ox) * rc.ix;
        y0 = (as_f(ea.z) - r.oy) * rc.iy;
        y1 = (as_f(ea.w) - r.oy) * rc.iy;
        z0 = (as_f(eb.x) - r.oz) * rc.iz;
        z1 = (as_f(eb.y) - r.oz) * rc.iz;
    }
    if (SLAB == SLAB_STRICT) {
        float lo = x0 < x1 ? x0 : x1, hi = x0 < x1 ? x1 : x0;
        tmin = fmaxf(tmin, lo);
        tmax = fminf(tmax, hi);
        lo = y0 < y1 ? y0 : y1;
        hi = y0 < y1 ? y1 : y0;
        tmin = fmaxf(tmin, lo);
        tmax = fminf(tmax, hi);
        lo = z0 < z1 ? z0 : z1;
        hi = z0 < z1 ? z1 : z0;
        tmin = fmaxf(tmin, lo);
        tmax = fminf(tmax, hi);
    } else if (SLAB == SLAB_FMA) {
        tmin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
        // The interval end comes from memory, so fminf would first re-quiet it on every call; the instruction itself
        // needs no such step (a NaN operand drops out, as in fminf).
        const float exit3 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
        asm("v_min_f32 %0, %1, %2" : "=v"(tmax) : "v"(exit3), "v"(tmax));
    } else {
        tmin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
        tmax = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    }
    return !(tmax <= tmin);
}

struct Best {
    float t;       // running interval end = distance of the closest accepted hit so far
    uint32_t ent;  // byte offset of that hit's entry, NO_HIT while none
};

// Accept a candidate distance under the mode's rule.
//  strict: Interval::contains (lib.rs:82-84) against the shrinking interval: tmin <= t < best.
//  fast  : same closest hit, visited in another order: on an exact tie the primitive the reference tests
//          first (lower rank) wins, as it does there because the later one needs t < tmax.
//  SHIFT : log2 of the bytes one rank-table element stands for (32-byte entries: 5; 16-byte slots of the packed array: 4).
template <bool STRICT, int SHIFT = 5>
__device__ __forceinline__ bool in_range(float t, float tmin, const Best& b, uint32_t off, const uint32_t* __restrict__ rank) {
    if (STRICT) return tmin <= t && t < b.t;
    if (!(tmin <= t)) return false;
    if (t < b.t) return true;
    return t == b.t && b.ent != NO_HIT && rank[off >> SHIFT] < rank[b.ent >> SHIFT];
}

// Square root and quotient for the render kernels' fast walk: the hardware's reciprocal square root / reciprocal (one
// unit in the last place) followed by one Newton step on the residual, which a fused multiply-add yields exactly.  The
// result is the correctly rounded one -- the reference's -- for all but a vanishing share of operands, at 5 + 2 plain
// multiply-adds instead of the ~12 + ~10 instructions of the guaranteed sequences.  A sphere's root is a difference of
// two numbers much larger than itself when the origin is near the surface of a large sphere; with the square root equal
// bit for bit that difference is too, and the hit distance agrees with the reference's to the last place or two.
//   quick_sqrt: x >= 0; x below 1e-37 (but for 0, which gives 0) is not supported.
//   quick_div : r = rcp(d) is passed in so that two quotients by one divisor share it.
__device__ __forceinline__ float quick_sqrt(float x) {
#ifdef RTX_QUICK_PLAIN
    return __builtin_amdgcn_sqrtf(x);
#else
    const float y = __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 1e-37f));  // (finite for x = 0, so that s = e = 0 below)
    const float s = x * y;
    const float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, 0.5f * y, s);
#endif
}
__device__ __forceinline__ float quick_div(float n, float d, float r) {
#ifdef RTX_QUICK_PLAIN
    return n * r;
#else
    const float q = n * r;
    const float e = __builtin_fmaf(-q, d, n);
    return __builtin_fmaf(e, r, q);
#endif
}

// Sphere::hit up to the accepted root (mesh.rs:113-133).
//  QUICK : the two roots through quick_sqrt / quick_div (hardware reciprocal square root and reciprocal, one Newton step
//          each) instead of the guaranteed correctly rounded sqrt and two divisions, which expand to ~35 instructions.
//          Only the render kernels' fast walk uses it (and the trace mode that gates it, RTX_TRAVERSAL_RENDER: hit
//          distances bit-equal to the oracle's on every ray measured); the other trace modes compute the roots with the
//          guaranteed sequences.  Equal inputs give equal roots, so coincident primitives tie as before.
template <bool STRICT, int SHIFT = 5, bool QUICK = false>
__device__ __forceinline__ void sphere_test(const uint4& ea, const uint4& eb, const Ray& r, const RayConst& rc, float tmin, Best& best,
                                            uint32_t idx, const uint32_t* __restrict__ rank) {
    static_assert(!(STRICT && QUICK), "the strict walk reproduces the reference's arithmetic");
    float cx = as_f(ea.x) + as_f(ea.w) * r.time;  // center(time) = initial_center + center_delta * time (mesh.rs:178-180)
    float cy = as_f(ea.y) + as_f(eb.x) * r.time;
    float cz = as_f(ea.z) + as_f(eb.y) * r.time;
    float radius = as_f(eb.z);
    float ocx = cx - r.ox, ocy = cy - r.oy, ocz = cz - r.oz;
    float h = dot3(r.dx, r.dy, r.dz, ocx, ocy, ocz);
    float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - radius * radius;
    float disc = h * h - rc.a * c;
    if (disc < 0.0f) return;
    float root;
    if (QUICK) {
        const float sqrtd = quick_sqrt(disc);
        const float inv_a = __builtin_amdgcn_rcpf(rc.a);
        root = quick_div(h - sqrtd, rc.a, inv_a);
        if (!(tmin <= root)) root = quick_div(h + sqrtd, rc.a, inv_a);
        if (!in_range<false, SHIFT>(root, tmin, best, idx, rank)) return;
    } else {
        const float sqrtd = __builtin_sqrtf(disc);
        root = (h - sqrtd) / rc.a;
        if (STRICT) {
            if (!in_range<true, SHIFT>(root, tmin, best, idx, rank)) {
                root = (h + sqrtd) / rc.a;
                if (!in_range<true, SHIFT>(root, tmin, best, idx, rank)) return;
            }
        } else {
            // the first root inside [tmin, end) is the near one unless it lies before tmin
            if (!(tmin <= root)) root = (h + sqrtd) / rc.a;
            if (!in_range<false, SHIFT>(root, tmin, best, idx, rank)) return;
        }
    }
    best.t = root;
    best.ent = idx;
}

// Quad::hit (mesh.rs:202-221, :248-256); TRI uses the same plane / alpha / beta with the triangle's domain.
// LD(k) returns record k of the primitive: 0 (Q.xyz, d)  1 (u.xyz, ..)  2 (v.xyz, ..)  3 (n.xyz, ..)  4 (w.xyz, ..).
// QUICK (render kernels' fast walk only): the division through quick_div.
template <bool STRICT, int SHIFT, bool QUICK = false, class LD>
__device__ __forceinline__ void planar_test_ld(LD ld, bool is_tri, const Ray& r, float tmin, Best& best, uint32_t idx,
                                               const uint32_t* __restrict__ rank) {
    static_assert(!(STRICT && QUICK), "the strict walk reproduces the reference's arithmetic");
    const float4 q = ld(0), n = ld(3);
    float denom = dot3(n.x, n.y, n.z, r.dx, r.dy, r.dz);
    if (fabsf(denom) < 1e-8f) return;
    const float num = q.w - dot3(n.x, n.y, n.z, r.ox, r.oy, r.oz);
    float t = QUICK ? quick_div(num, denom, __builtin_amdgcn_rcpf(denom)) : num / denom;
    if (!in_range<STRICT, SHIFT>(t, tmin, best, idx, rank)) return;
    const float4 u = ld(1), v = ld(2), w = ld(4);
    float px = (r.ox + r.dx * t) - q.x, py = (r.oy + r.dy * t) - q.y, pz = (r.oz + r.dz * t) - q.z;
    // alpha = w . (p x v), beta = w . (u x p)   with glam's cross
    float alpha = dot3(w.x, w.y, w.z, py * v.z - v.y * pz, pz * v.x - v.z * px, px * v.y - v.x * py);
    float beta = dot3(w.x, w.y, w.z, u.y * pz - py * u.z, u.z * px - pz * u.x, u.x * py - px * u.y);
    bool inside = is_tri ? (alpha >= 0.0f && beta >= 0.0f && alpha + beta <= 1.0f)
                         : (0.0f <= alpha && alpha < 1.0f && 0.0f <= beta && beta < 1.0f);
    if (!inside) return;
    best.t = t;
    best.ent = idx;
}
template <bool STRICT, bool QUICK = false>
__device__ __forceinline__ void planar_test(const float4* __restrict__ planar, uint32_t pidx, bool is_tri, const Ray& r, float tmin, Best& best,
                                            uint32_t idx, const uint32_t* __restrict__ rank) {
    planar_test_ld<STRICT, 5, QUICK>([&](int k) { return planar[5 * pidx + k]; }, is_tri, r, tmin, best, idx, rank);
}

// ---------------------------------------------------------------------------------------------- hit record
struct HitRec {
    float px, py, pz;  // Hit::point
    float nx, ny, nz;  // Hit::normal (flipped against the ray, mesh.rs:45)
    float u, v;
    uint32_t mat;
    uint32_t prim_id;
};

// Sphere::uv (mesh.rs:182-187) of the outward unit normal
__device__ __forceinline__ void sphere_uv(float nx, float ny, float nz, float& u, float& v) {
    float theta = acosf(-ny);
    float phi = atan2f(-nz, nx) + PI_F;
    u = phi / (2.0f * PI_F);
    v = theta / PI_F;
}

// The rest of Sphere::hit / Quad::hit after the root is accepted, plus Hit::new (mesh.rs:33-51, :135-138, :214-220).
// EXACT = true divides by the radius as the reference does (bit-identical normals, rtx_trace); false takes the quotient
// through the hardware reciprocal and one Newton step (quick_div: the same bits for all but a vanishing share of operands).
template <bool WANT_UV, bool EXACT = true>
__device__ __forceinline__ HitRec make_hit(const DevScene& S, const uint4& ea, const uint4& eb, const Ray& r, float t) {
    HitRec h;
    uint32_t kind = eb.w & 15u, pi = eb.w >> 4;
    h.px = r.ox + r.dx * t;  // Ray::get_point (lib.rs:27-29)
    h.py = r.oy + r.dy * t;
    h.pz = r.oz + r.dz * t;
    float nx, ny, nz;
    if (kind == ENT_SPHERE) {
        float cx = as_f(ea.x) + as_f(ea.w) * r.time;
        float cy = as_f(ea.y) + as_f(eb.x) * r.time;
        float cz = as_f(ea.z) + as_f(eb.y) * r.time;
        float radius = as_f(eb.z);
        if (EXACT) {
            nx = (h.px - cx) / radius;
            ny = (h.py - cy) / radius;
            nz = (h.pz - cz) / radius;
        } else {
            const float inv_r = __builtin_amdgcn_rcpf(radius);
            nx = quick_div(h.px - cx, radius, inv_r);
            ny = quick_div(h.py - cy, radius, inv_r);
            nz = quick_div(h.pz - cz, radius, inv_r);
        }
        h.u = 0.0f;
        h.v = 0.0f;
        if (WANT_UV) sphere_uv(nx, ny, nz, h.u, h.v);
        h.mat = __float_as_uint(S.sph[2 * pi + 1].w);
        h.prim_id = S.sph_id[pi];
    } else {
        uint32_t p = kind == ENT_TRI ? pi + S.n_quads : pi;
        const float4 q = S.planar[5 * p], u = S.planar[5 * p + 1], v = S.planar[5 * p + 2], n = S.planar[5 * p + 3], w = S.planar[5 * p + 4];
        float dx = h.px - q.x, dy = h.py - q.y, dz = h.pz - q.z;
        h.u = dot3(w.x, w.y, w.z, dy * v.z - v.y * dz, dz * v.x - v.z * dx, dx * v.y - v.x * dy);
        h.v = dot3(w.x, w.y, w.z, u.y * dz - dy * u.z, u.z * dx - dz * u.x, u.x * dy - dx * u.y);
        nx = n.x;
        ny = n.y;
        nz = n.z;
        h.mat = __float_as_uint(u.w);
        h.prim_id = S.planar_id[p];
    }
    bool front = dot3(r.dx, r.dy, r.dz, nx, ny, nz) < 0.0f;  // mesh.rs:40
    h.nx = front ? nx : -nx;
    h.ny = front ? ny : -ny;
    h.nz = front ? nz : -nz;
    return h;
}

// ---------------------------------------------------------------------------------------------- packed fast array
// 16-byte slots (layout: accel.hpp).  Word 3 of an entry's first slot is the walk state of a lane standing there:
// below PK_PRIM = box (and its miss link, in slots), kind << 28 | index = primitive, ENT_END = walk complete.
constexpr uint32_t PK_PRIM = 0x10000000u;
constexpr uint32_t PK_PAD_FLAG = 0x08000000u;   // in a sphere's tag: a one-slot pad follows the entry (layout: accel.hpp)
constexpr uint32_t PK_INDEX_MASK = 0x07FFFFFFu;  // the index part of a tag
constexpr uint32_t PK_SLOT = 16u;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// Box test in centre / half-extent form: per axis the distance to the centre plane, minus / plus half_extent * |1 / d|
// (three fused operations that read the halfs in place, no min / max).  Same conservative answer as slab_hit<SLAB_FMA>
// on the float box the halfs were rounded outward from; the boxes' padding covers the extra rounding step.
__device__ __forceinline__ bool pk_slab_hit(const uint4& e, const RayConst& rc, float tmin, float tmax) {
    const half2_t a = __builtin_bit_cast(half2_t, e.x), b = __builtin_bit_cast(half2_t, e.y), c = __builtin_bit_cast(half2_t, e.z);
    const float tcx = __builtin_fmaf((float)a.x, rc.ix, rc.nx);
    const float tcy = __builtin_fmaf((float)a.y, rc.iy, rc.ny);
    const float tcz = __builtin_fmaf((float)b.x, rc.iz, rc.nz);
    const float hx = (float)b.y, hy = (float)c.x, hz = (float)c.y;
    const float ax = __builtin_fabsf(rc.ix), ay = __builtin_fabsf(rc.iy), az = __builtin_fabsf(rc.iz);
    const float enter = fmaxf(fmaxf(__builtin_fmaf(-hx, ax, tcx), __builtin_fmaf(-hy, ay, tcy)), fmaxf(__builtin_fmaf(-hz, az, tcz), tmin));
    const float exit3 = fminf(fminf(__builtin_fmaf(hx, ax, tcx), __builtin_fmaf(hy, ay, tcy)), __builtin_fmaf(hz, az, tcz));
    float exit;
    asm("v_min_f32 %0, %1, %2" : "=v"(exit) : "v"(exit3), "v"(tmax));  // as in slab_hit: no re-quieting of the interval end
    return !(exit <= enter);
}

// The same with float centre / half extent: s0 = (centre.xyz, link), s1 = (half extent.xyz, 0).
__device__ __forceinline__ bool pk_slab_hit(const uint4& s0, const uint4& s1, const RayConst& rc, float tmin, float tmax) {
    const float tcx = __builtin_fmaf(as_f(s0.x), rc.ix, rc.nx);
    const float tcy = __builtin_fmaf(as_f(s0.y), rc.iy, rc.ny);
    const float tcz = __builtin_fmaf(as_f(s0.z), rc.iz, rc.nz);
    const float hx = as_f(s1.x), hy = as_f(s1.y), hz = as_f(s1.z);
    const float ax = __builtin_fabsf(rc.ix), ay = __builtin_fabsf(rc.iy), az = __builtin_fabsf(rc.iz);
    const float enter = fmaxf(fmaxf(__builtin_fmaf(-hx, ax, tcx), __builtin_fmaf(-hy, ay, tcy)), fmaxf(__builtin_fmaf(-hz, az, tcz), tmin));
    const float exit3 = fminf(fminf(__builtin_fmaf(hx, ax, tcx), __builtin_fmaf(hy, ay, tcy)), __builtin_fmaf(hz, az, tcz));
    float exit;
    asm("v_min_f32 %0, %1, %2" : "=v"(exit) : "v"(exit3), "v"(tmax));
    return !(exit <= enter);
}

// The render kernels with float boxes walk in scaled ray parameters t' = t * PK_T_SCALE: 1 / direction and -(origin / direction) carry the
// factor, so every plane distance comes out scaled, and so is the interval end the caller passes.  The factor is a power
// of two: all values are the unscaled ones with another exponent, and every comparison between them decides as before.
// What the scaling buys is the interval start for free: the fused operation that yields the last near-plane distance
// clamps its result to [0, 1] (an output modifier), which stands in for max(., tmin).  The start becomes 0 instead of
// tmin, and near distances beyond t' = 1 -- t = 2^60, farther than any box of a scene a float can describe unless the
// direction is shorter than ~1e-14 -- read as 1: both only ever accept a box the exact test would reject.
constexpr float PK_T_SCALE = 0x1p-60f, PK_T_UNSCALE = 0x1p60f;
__device__ __forceinline__ float fma_neg_abs_clamp01(float h, float inv, float c) {  // clamp(-h * |inv| + c, 0, 1)
    float d;
    asm("v_fma_f32 %0, -%1, |%2|, %3 clamp" : "=v"(d) : "v"(h), "v"(inv), "v"(c));
    return d;
}
__device__ __forceinline__ void scale_ray_const(RayConst& rc) {
    rc.ix *= PK_T_SCALE;
    rc.iy *= PK_T_SCALE;
    rc.iz *= PK_T_SCALE;
    rc.nx *= PK_T_SCALE;
    rc.ny *= PK_T_SCALE;
    rc.nz *= PK_T_SCALE;
}
__device__ __forceinline__ bool pk_slab_hit_scaled(const uint4& s0, const uint4& s1, const RayConst& rc, float tmax_scaled) {
    const float tcx = __builtin_fmaf(as_f(s0.x), rc.ix, rc.nx);
    const float tcy = __builtin_fmaf(as_f(s0.y), rc.iy, rc.ny);
    const float tcz = __builtin_fmaf(as_f(s0.z), rc.iz, rc.nz);
    const float hx = as_f(s1.x), hy = as_f(s1.y), hz = as_f(s1.z);
    const float ax = __builtin_fabsf(rc.ix), ay = __builtin_fabsf(rc.iy), az = __builtin_fabsf(rc.iz);
    const float enter = fmaxf(fmaxf(__builtin_fmaf(-hx, ax, tcx), __builtin_fmaf(-hy, ay, tcy)), fma_neg_abs_clamp01(hz, rc.iz, tcz));
    const float exit3 = fminf(fminf(__builtin_fmaf(hx, ax, tcx), __builtin_fmaf(hy, ay, tcy)), __builtin_fmaf(hz, az, tcz));
    float exit;
    asm("v_min_f32 %0, %1, %2" : "=v"(exit) : "v"(exit3), "v"(tmax_scaled));
    return !(exit <= enter);
}

// A run of box steps over the copy of the packed array (float boxes) that starts at LDS address 0, for the lanes of the
// wave whose state word says "box": each step is pk_slab_hit_scaled, then the lane moves to the next entry (hit) or to
// the box's miss link (its state word, a byte offset) and reads that entry's first two slots.  A lane leaves the run
// when it arrives at an entry that is not a box; the run ends for all when fewer than min_lanes (>= 1) are left.
// lane_tests counts the lane's own steps (the wave's box tests are the sum over its lanes).
// Written out instruction by instruction because of what the loop costs around the test itself: the wave is bound by
// instruction issue, scalar instructions included (six more of them per step cost 14 %), and the compiler's form of
// this loop -- a mask saved, tested and restored around the step, the wave's condition kept as a lane mask -- spends
// thirteen scalar instructions per step where three do: lanes that leave are taken out of EXEC by the compare that
// finds them (v_cmpx), the lanes left are counted from EXEC and compared with a threshold the caller derives once per run.  The two vector instructions
// between the compare that decides hit / miss and the select that consumes it (the step counter, the address of the
// next entry) are the wait states that pair needs.
// at_box: the wave's lanes whose state word says "box" (the caller has that mask from its own count), not empty.
// The first two slots of the current entry live in v[56:59] (centre.xyz, state) and v[60:63] (half extent.xyz, -).
__device__ __forceinline__ void pk_box_run_lds(unsigned long long at_box, uint32_t& off, uint32_t& st, uint32_t& lane_tests, const RayConst& rc,
                                               float tmax_scaled, uint32_t min_lanes) {
    uint32_t nb;
    unsigned long long saved;
    float t0, t1, t2;
    asm volatile(
        "s_mov_b64 %[saved], exec\n"
        "s_mov_b64 exec, %[joined]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b128 v[56:59], %[off]\n"
        "ds_read_b128 v[60:63], %[off] offset:16\n"
        "s_waitcnt lgkmcnt(1)\n"
        "1:\n"
        "v_fma_f32 %[t0], v56, %[ix], %[nx]\n"
        "v_fma_f32 %[t1], v57, %[iy], %[ny]\n"
        "v_fma_f32 %[t2], v58, %[iz], %[nz]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_fma_f32 v56, -v60, |%[ix]|, %[t0]\n"
        "v_fma_f32 v57, -v61, |%[iy]|, %[t1]\n"
        "v_fma_f32 v58, -v62, |%[iz]|, %[t2] clamp\n"
        "v_fma_f32 %[t0], v60, |%[ix]|, %[t0]\n"
        "v_fma_f32 %[t1], v61, |%[iy]|, %[t1]\n"
        "v_fma_f32 %[t2], v62, |%[iz]|, %[t2]\n"
        "v_max3_f32 v56, v56, v57, v58\n"
        "v_min3_f32 %[t0], %[t0], %[t1], %[t2]\n"
        "v_min_f32 %[t0], %[t0], %[tmax]\n"
        "v_cmp_nle_f32 vcc, %[t0], v56\n"
        "v_add_u32 %[cnt], 1, %[cnt]\n"
        "v_add_u32 %[t1], 32, %[off]\n"
        "v_cndmask_b32 %[off], v59, %[t1], vcc\n"
        "ds_read_b128 v[56:59], %[off]\n"
        "ds_read_b128 v[60:63], %[off] offset:16\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_cmpx_gt_u32 vcc, %[prim], v59\n"
        "s_bcnt1_i32_b64 %[nb], exec\n"
        "s_cmp_ge_u32 %[nb], %[thr]\n"
        "s_cbranch_scc1 1b\n"
        "s_mov_b64 exec, %[joined]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_mov_b32 %[st], v59\n"
        "s_mov_b64 exec, %[saved]\n"
        : [off] "+v"(off), [st] "+v"(st), [cnt] "+v"(lane_tests), [nb] "=&s"(nb), [saved] "=&s"(saved), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [ix] "v"(rc.ix), [iy] "v"(rc.iy), [iz] "v"(rc.iz), [nx] "v"(rc.nx), [ny] "v"(rc.ny), [nz] "v"(rc.nz), [tmax] "v"(tmax_scaled),
          [thr] "s"(min_lanes), [prim] "s"(PK_PRIM), [joined] "s"(at_box)
        : "vcc", "scc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// The same run over a copy of a fast array of 32-byte entries that starts at LDS address 0 (scenes with media or placed
// instance primitives, whose arrays have no packed form): each step is slab_hit<SLAB_FMA> -- six fused plane distances,
// their per-axis minima and maxima, the lane's own interval start (inside a medium's boundary query it is not the
// caller's) -- then the next entry (hit) or the box's miss link.  A box entry is (x0 x1 y0 y1) (z0 z1 link 0): its state
// word, 0 for a box, is the last word of the second read, so one wait per step covers both reads.
// v[56:59] and v[60:63] hold the current entry.
__device__ __forceinline__ void box_run_lds(unsigned long long at_box, uint32_t& off, uint32_t& st, uint32_t& lane_tests, const RayConst& rc, float tmin,
                                            float tmax, uint32_t min_lanes) {
    uint32_t nb;
    unsigned long long saved;
    float t0, t1, t2, t3, t4, t5;
    asm volatile(
        "s_mov_b64 %[saved], exec\n"
        "s_mov_b64 exec, %[joined]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b128 v[56:59], %[off]\n"
        "ds_read_b128 v[60:63], %[off] offset:16\n"
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "v_fma_f32 %[t0], v56, %[ix], %[nx]\n"
        "v_fma_f32 %[t1], v57, %[ix], %[nx]\n"
        "v_fma_f32 %[t2], v58, %[iy], %[ny]\n"
        "v_fma_f32 %[t3], v59, %[iy], %[ny]\n"
        "v_fma_f32 %[t4], v60, %[iz], %[nz]\n"
        "v_fma_f32 %[t5], v61, %[iz], %[nz]\n"
        "v_min_f32 v56, %[t0], %[t1]\n"
        "v_max_f32 %[t0], %[t0], %[t1]\n"
        "v_min_f32 v57, %[t2], %[t3]\n"
        "v_max_f32 %[t2], %[t2], %[t3]\n"
        "v_min_f32 v58, %[t4], %[t5]\n"
        "v_max_f32 %[t4], %[t4], %[t5]\n"
        "v_max_f32 v56, v56, %[tmin]\n"
        "v_max3_f32 v56, v57, v58, v56\n"
        "v_min3_f32 %[t0], %[t2], %[t4], %[t0]\n"
        "v_min_f32 %[t0], %[t0], %[tmax]\n"
        "v_cmp_nle_f32 vcc, %[t0], v56\n"
        "v_add_u32 %[cnt], 1, %[cnt]\n"
        "v_add_u32 %[t1], 32, %[off]\n"
        "v_cndmask_b32 %[off], v62, %[t1], vcc\n"
        "ds_read_b128 v[56:59], %[off]\n"
        "ds_read_b128 v[60:63], %[off] offset:16\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_cmpx_eq_u32 vcc, 0, v63\n"
        "s_bcnt1_i32_b64 %[nb], exec\n"
        "s_cmp_ge_u32 %[nb], %[thr]\n"
        "s_cbranch_scc1 1b\n"
        "s_mov_b64 exec, %[joined]\n"
        "v_mov_b32 %[st], v63\n"
        "s_mov_b64 exec, %[saved]\n"
        : [off] "+v"(off), [st] "+v"(st), [cnt] "+v"(lane_tests), [nb] "=&s"(nb), [saved] "=&s"(saved), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5)
        : [ix] "v"(rc.ix), [iy] "v"(rc.iy), [iz] "v"(rc.iz), [nx] "v"(rc.nx), [ny] "v"(rc.ny), [nz] "v"(rc.nz), [tmin] "v"(tmin), [tmax] "v"(tmax),
          [thr] "s"(min_lanes), [joined] "s"(at_box)
        : "vcc", "scc", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// Reads slots of the packed array through a generic pointer (trace kernel, rare paths).
struct SlotReader {
    const uint4* base;
    __device__ __forceinline__ uint4 at(uint32_t off) const { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + off); }
};
__device__ __forceinline__ float4 as_f4(const uint4& u) { return make_float4(as_f(u.x), as_f(u.y), as_f(u.z), as_f(u.w)); }

// Primitive test at the packed entry whose first two slots are s0, s1 (byte offset off); LD(k) reads slot k of it.
// Returns the entry's size in bytes.
template <bool QUICK, class LD>
__device__ __forceinline__ uint32_t pk_prim_test(const uint4& s0, const uint4& s1, LD ld, uint32_t off, const Ray& r, const RayConst& rc, float tmin,
                                                 Best& best, const uint32_t* __restrict__ rank) {
    const uint32_t kind = s0.w >> 28;
    if (kind == ENT_SPHERE) {
        // slots (c0.xyz, tag) (dc.xyz, r) in the word order sphere_test reads an unpacked entry
        sphere_test<false, 4, QUICK>(make_uint4(s0.x, s0.y, s0.z, s1.x), make_uint4(s1.y, s1.z, s1.w, 0u), r, rc, tmin, best, off, rank);
        return (2u + ((s0.w / PK_PAD_FLAG) & 1u)) * PK_SLOT;  // (a pad slot may follow: accel.hpp, pk_pad_every)
    }
    // slots (n, tag) (Q, d) (u, material) (v, prim id) (w, 0) -> records 0 Q  1 u  2 v  3 n  4 w
    planar_test_ld<false, 4, QUICK>(
        [&](int k) { return k == 0 ? as_f4(s1) : k == 3 ? as_f4(s0) : as_f4(ld(k == 4 ? 4 : k + 1)); }, kind == ENT_TRI, r, tmin, best, off, rank);
    return 5u * PK_SLOT;
}

// Hit record of the packed entry at the accepted offset.  IDS: also fill material and prim id (one more read for a sphere).
template <bool WANT_UV, bool EXACT, bool IDS, class LD>
__device__ __forceinline__ HitRec pk_make_hit(const DevScene& S, LD ld, const Ray& r, float t) {
    const uint4 s0 = ld(0), s1 = ld(1);
    const uint32_t kind = s0.w >> 28, pi = s0.w & PK_INDEX_MASK;
    HitRec h;
    h.px = r.ox + r.dx * t;  // Ray::get_point (lib.rs:27-29)
    h.py = r.oy + r.dy * t;
    h.pz = r.oz + r.dz * t;
    h.u = 0.0f;
    h.v = 0.0f;
    h.mat = 0u;
    h.prim_id = 0u;
    float nx, ny, nz;
    if (kind == ENT_SPHERE) {
        const float cx = as_f(s0.x) + as_f(s1.x) * r.time;
        const float cy = as_f(s0.y) + as_f(s1.y) * r.time;
        const float cz = as_f(s0.z) + as_f(s1.z) * r.time;
        const float radius = as_f(s1.w);
        if (EXACT) {
            nx = (h.px - cx) / radius;
            ny = (h.py - cy) / radius;
            nz = (h.pz - cz) / radius;
        } else {
            const float inv_r = __builtin_amdgcn_rcpf(radius);
            nx = quick_div(h.px - cx, radius, inv_r);
            ny = quick_div(h.py - cy, radius, inv_r);
            nz = quick_div(h.pz - cz, radius, inv_r);
        }
        if (WANT_UV) sphere_uv(nx, ny, nz, h.u, h.v);
        if (IDS) {
            h.mat = __float_as_uint(S.sph[2 * pi + 1].w);
            h.prim_id = S.sph_id[pi];
        }
    } else {
        nx = as_f(s0.x);
        ny = as_f(s0.y);
        nz = as_f(s0.z);
        if (WANT_UV || IDS) {
            const uint4 su = ld(2), sv = ld(3);
            if (WANT_UV) {
                const uint4 sw = ld(4);
                const float dx = h.px - as_f(s1.x), dy = h.py - as_f(s1.y), dz = h.pz - as_f(s1.z);
                const float ux = as_f(su.x), uy = as_f(su.y), uz = as_f(su.z), vx = as_f(sv.x), vy = as_f(sv.y), vz = as_f(sv.z);
                h.u = dot3(as_f(sw.x), as_f(sw.y), as_f(sw.z), dy * vz - vy * dz, dz * vx - vz * dx, dx * vy - vx * dy);
                h.v = dot3(as_f(sw.x), as_f(sw.y), as_f(sw.z), uy * dz - dy * uz, uz * dx - dz * ux, ux * dy - dx * uy);
            }
            h.mat = su.w;
            h.prim_id = sv.w;
        }
    }
    const bool front = dot3(r.dx, r.dy, r.dz, nx, ny, nz) < 0.0f;  // mesh.rs:40
    h.nx = front ? nx : -nx;
    h.ny = front ? ny : -ny;
    h.nz = front ? nz : -nz;
    return h;
}

// The steps of the render kernels' walk over the packed array.  render_kernel and the trace kernel that gates its
// arithmetic ray by ray (RTX_TRAVERSAL_RENDER) both call these, so what is checked is what is timed.
// SCALED: float boxes (arrays sized for LDS), walk in scaled ray parameters; else half-precision boxes, plain parameters.
template <bool SCALED>
__device__ __forceinline__ bool pk_render_box(const uint4& e, const uint4& e1, const RayConst& rc, float tmin, float best_t) {
    return SCALED ? pk_slab_hit_scaled(e, e1, rc, best_t) : pk_slab_hit(e, rc, tmin, best_t);
}
template <bool SCALED, class LD>
__device__ __forceinline__ uint32_t pk_render_prim(const uint4& s0, const uint4& s1, LD ld, uint32_t off, const Ray& ray, const RayConst& rc, float tmin,
                                                   Best& best, const uint32_t* __restrict__ rank) {
    Best real = best;  // the primitive tests work in the ray's own parameter
    if (SCALED) real.t = best.t * PK_T_UNSCALE;
    const uint32_t size = pk_prim_test<true>(s0, s1, ld, off, ray, rc, tmin, real, rank);
    best.ent = real.ent;
    best.t = SCALED ? real.t * PK_T_SCALE : real.t;
    return size;
}
template <bool SCALED, bool WANT_UV, bool IDS, class LD>
__device__ __forceinline__ HitRec pk_render_hit(const DevScene& S, LD ld, const Ray& ray, float best_t) {
    const float t_hit = SCALED ? best_t * PK_T_UNSCALE : best_t;  // exact: the factor is a power of two
    return pk_make_hit<WANT_UV, false, IDS>(S, ld, ray, t_hit);
}

// ---------------------------------------------------------------------------------------------- wide fast array
// One walk step at a four-child node (layout: accel.hpp).  Tests the four child boxes against [tmin, tmax] and reports
//   next   the nearest child node that was hit (WIDE_NONE if none),
//   group  the other child nodes that were hit, nearest first, as one word: node index of child 0 << 9 | three 3-bit
//          entries from bit 0, nearest first -- a child number 0 .. 3, or 7 = no further child; 0 if there is none,
//   tbase, tmask   the primitives of the leaves that were hit: bit i = the i-th primitive record from slot tbase.
// A child's box is given on the 256-step grid of the node's own box, one byte per bound.  In ray parameters the planes
// of axis a lie at q * A_a + B_a with A_a = scale_a / d_a and B_a = (origin_a - o_a) / d_a, one fused multiply-add per
// plane; which of a child's two bytes is the near plane depends only on the sign of d_a, so it is chosen once per axis
// for all four children (their bytes share a word).  The rounding of these operations is covered by the padding of the
// boxes the bytes were rounded outward from, as for the other fused box tests.
// Children are sorted by the distance at which the ray enters their boxes (ties and the order among leaves do not matter:
// the closest hit does not depend on the visiting order).  The child number rides in the three low bits of the distance's
// bit pattern, so sorting four words sorts the children; distances are >= tmin >= 0 in the render kernels, where bit
// patterns order like the values (a negative tmin in rtx_trace only costs ordering quality, never a hit).
constexpr uint32_t WIDE_NONE = 0xFFFFFFFFu;
constexpr uint32_t WIDE_NODE_BYTES = 64u;
constexpr uint32_t WIDE_PRIM_BYTES = 80u;

template <int J>
__device__ __forceinline__ float wide_byte(uint32_t w) {  // byte J of w as a float (v_cvt_f32_ubyteJ)
    return (float)((w >> (8 * J)) & 255u);
}
template <int J>
__device__ __forceinline__ uint32_t wide_child_key(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, float ax, float ay, float az,
                                                   float bx, float by, float bz, float tmin, float tmax) {
    const float enter = fmaxf(fmaxf(__builtin_fmaf(wide_byte<J>(nx), ax, bx), __builtin_fmaf(wide_byte<J>(ny), ay, by)),
                              fmaxf(__builtin_fmaf(wide_byte<J>(nz), az, bz), tmin));
    const float exit3 = fminf(fminf(__builtin_fmaf(wide_byte<J>(fx), ax, bx), __builtin_fmaf(wide_byte<J>(fy), ay, by)), __builtin_fmaf(wide_byte<J>(fz), az, bz));
    float exit;
    asm("v_min_f32 %0, %1, %2" : "=v"(exit) : "v"(exit3), "v"(tmax));  // (no re-quieting of the interval end, as in slab_hit)
    return (exit <= enter) ? 0xFFFFFFFFu : ((__float_as_uint(enter) & ~7u) | (uint32_t)J);
}
template <int J>
__device__ __forceinline__ uint32_t wide_leaf_bits(uint32_t leaves, uint32_t& is_leaf) {  // primitives of leaf J; is_leaf: all ones for a leaf, else 0
    const uint32_t b = (leaves >> (8 * J)) & 255u;
    is_leaf = b != 0u ? 0xFFFFFFFFu : 0u;
    // count << 5 | first  ->  `count` ones from bit `first`: two field extracts and v_bfm_b32 (which the compiler does
    // not form from the shifts)
    const uint32_t count = __builtin_amdgcn_ubfe(leaves, 8 * J + 5, 3), first = __builtin_amdgcn_ubfe(leaves, 8 * J, 5);
    uint32_t m;
    asm("v_bfm_b32 %0, %1, %2" : "=v"(m) : "v"(count), "v"(first));
    return m;
}

// s0 .. s3: the node's four slots
__device__ __forceinline__ void wide_node_step(const uint4& s0, const uint4& s1, const uint4& s2, const uint4& s3, const RayConst& rc, float tmin, float tmax,
                                               uint32_t& next, uint32_t& group, uint32_t& tbase, uint32_t& tmask, uint32_t& n_children) {
    const uint32_t child_base = s0.w, leaves = s2.w;
    const float ax = as_f(s1.x) * rc.ix, ay = as_f(s1.y) * rc.iy, az = as_f(s1.z) * rc.iz;
    const float bx = __builtin_fmaf(as_f(s0.x), rc.ix, rc.nx), by = __builtin_fmaf(as_f(s0.y), rc.iy, rc.ny), bz = __builtin_fmaf(as_f(s0.z), rc.iz, rc.nz);
    // near / far planes per axis for all four children at once
    const bool px = rc.ix >= 0.0f, py = rc.iy >= 0.0f, pz = rc.iz >= 0.0f;
    const uint32_t nx = px ? s2.x : s3.x, fx = px ? s3.x : s2.x;
    const uint32_t ny = py ? s2.y : s3.y, fy = py ? s3.y : s2.y;
    const uint32_t nz = pz ? s2.z : s3.z, fz = pz ? s3.z : s2.z;
    uint32_t k0 = wide_child_key<0>(nx, ny, nz, fx, fy, fz, ax, ay, az, bx, by, bz, tmin, tmax);
    uint32_t k1 = wide_child_key<1>(nx, ny, nz, fx, fy, fz, ax, ay, az, bx, by, bz, tmin, tmax);
    uint32_t k2 = wide_child_key<2>(nx, ny, nz, fx, fy, fz, ax, ay, az, bx, by, bz, tmin, tmax);
    uint32_t k3 = wide_child_key<3>(nx, ny, nz, fx, fy, fz, ax, ay, az, bx, by, bz, tmin, tmax);
    // leaves that were hit hand out their primitives
    uint32_t l0, l1, l2, l3;
    const uint32_t m0 = wide_leaf_bits<0>(leaves, l0), m1 = wide_leaf_bits<1>(leaves, l1), m2 = wide_leaf_bits<2>(leaves, l2), m3 = wide_leaf_bits<3>(leaves, l3);
    tbase = s1.w;
    tmask = (k0 != 0xFFFFFFFFu ? m0 : 0u) | (k1 != 0xFFFFFFFFu ? m1 : 0u) | (k2 != 0xFFFFFFFFu ? m2 : 0u) | (k3 != 0xFFFFFFFFu ? m3 : 0u);
    n_children = s3.w >> 8;
    // child nodes that were hit, nearest first: a leaf's key becomes all ones like a miss
    k0 |= l0;
    k1 |= l1;
    k2 |= l2;
    k3 |= l3;
    const uint32_t a = min(k0, k1), b = max(k0, k1), c = min(k2, k3), d = max(k2, k3);
    k0 = min(a, c);
    const uint32_t m1s = max(a, c), m2s = min(b, d);
    k3 = max(b, d);
    k1 = min(m1s, m2s);
    k2 = max(m1s, m2s);
    next = k0 != 0xFFFFFFFFu ? child_base + (k0 & 3u) : WIDE_NONE;
    // (a key of all ones leaves 7 = "no further child" in its entry)
    group = (k1 & 4u) == 0u ? (child_base << 9) | (k1 & 7u) | ((k2 & 7u) << 3) | ((k3 & 7u) << 6) : 0u;
}

// Takes the next child node out of a pending group: returns its node index and leaves the group with one child fewer
// (0 when it was the last).  The group must not be 0.
__device__ __forceinline__ uint32_t wide_group_pop(uint32_t& group) {
    const uint32_t node = (group >> 9) + (group & 3u);
    const uint32_t rest = ((group >> 3) & 0x3Fu) | 0x1C0u;  // the entries move down, "no further child" moves in
    group = (rest & 4u) == 0u ? ((group & 0xFFFFFE00u) | rest) : 0u;
    return node;
}

// ---------------------------------------------------------------------------------------------- brackets
// Translate / RotateY / ConstantMedium (mesh.rs:324-432, :627-696) appear in the threaded array as bracket entries
// around their child's entries.  Extra per-ray walk state they need:
struct WalkX {
    uint32_t cur_xf;   // innermost transform-enter entry the walk is inside, XF_NONE at top level
    uint32_t best_xf;  // cur_xf at the moment the current best hit was accepted
    float tmin;        // current interval start (differs from the caller's inside a medium's boundary queries)
    float sv_t;        // while inside a medium: the caller's best so far ...
    uint32_t sv_ent, sv_xf;
    float med_t1;      // NaN during the first boundary query, hit1.distance during the second
};
// A ConstantMedium's boundary may hold further media (mesh.rs:627-631 takes any `B: Mesh`; :635-638 then calls their
// hit() inside the boundary queries).  WalkX holds the boundary-query state of the innermost medium a walk is inside; the
// states of the media around it wait in DevScene::med_stack, one five-word record per nesting level and lane (the begin /
// end entries carry their level), word j of level l of lane g at [(5 * (l - 1) + j) * lanes + g].  Memory, not
// registers: it is touched only when a medium begins or ends inside another one.
constexpr int MED_MAX_DEPTH = 4;  // the host rejects deeper nesting (rtx_accel::MAX_MED_DEPTH)
struct MedStack {
    uint32_t* base;  // DevScene::med_stack (null when the scene nests no media: no entry then has a level above 0)
    uint32_t lanes;  // DevScene::med_lanes
    uint32_t lane;   // this lane's index among the lanes of the launch (< lanes: checked by the host at launch)
    __device__ __forceinline__ uint32_t* rec(uint32_t level) const {  // level of the begin / end entry, 1 <= level < MED_MAX_DEPTH
        return base + (size_t)(5u * (min(level, (uint32_t)MED_MAX_DEPTH - 1u) - 1u)) * lanes + lane;
    }
};
__device__ __forceinline__ void walkx_init(WalkX& x, float tmin) {
    x.cur_xf = XF_NONE;
    x.best_xf = XF_NONE;
    x.tmin = tmin;
    x.sv_t = 0.0f;
    x.sv_ent = NO_HIT;
    x.sv_xf = XF_NONE;
    x.med_t1 = __builtin_nanf("");
}

struct EntReader {  // reads entries of an array in global memory by byte offset (the bracket paths, hit records)
    const uint4* base;
    __device__ __forceinline__ void load(uint32_t off, uint4& a, uint4& b) const {
        const char* p = reinterpret_cast<const char*>(base) + off;
        a = *reinterpret_cast<const uint4*>(p);
        b = *reinterpret_cast<const uint4*>(p + 16);
    }
};
struct EntReaderLds {  // the same for a copy of the array that starts at LDS address 0: a byte offset is the LDS address
    __device__ __forceinline__ void load(uint32_t off, uint4& a, uint4& b) const {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const u32x4 lds_u4;
        lds_u4* p = (lds_u4*)(uintptr_t)off;
        const u32x4 a_ = p[0], b_ = p[1];
        a = make_uint4(a_.x, a_.y, a_.z, a_.w);
        b = make_uint4(b_.x, b_.y, b_.z, b_.w);
    }
};

// Translate::hit (mesh.rs:332) / RotateY::hit (mesh.rs:365-372): the ray handed to the wrapped object
__device__ __forceinline__ void xf_apply(const uint4& ea, Ray& r) {
    if (ea.x == 1u) {
        r.ox = r.ox - as_f(ea.y);
        r.oy = r.oy - as_f(ea.z);
        r.oz = r.oz - as_f(ea.w);
    } else {
        const float s = as_f(ea.y), c = as_f(ea.z);
        const float ox = c * r.ox - s * r.oz, oz = s * r.ox + c * r.oz;
        const float dx = c * r.dx - s * r.dz, dz = s * r.dx + c * r.dz;
        r.ox = ox;
        r.oz = oz;
        r.dx = dx;
        r.dz = dz;
    }
}
// The ray as seen inside transform-enter entry `xf` (and all transforms enclosing it), derived from the world ray: the
// enclosing transforms are applied outermost first, as the nested hit() calls of the reference do (mesh.rs:332, :365-372).
// An entry only links to its parent, so the chain is walked once to count it and then once per level from the top;
// nesting is shallow (two levels in the reference's scenes), and any depth the host accepts works.
constexpr int XF_MAX_DEPTH = 64;  // the host rejects deeper nesting (rtx_accel::MAX_XF_DEPTH)
template <class READER>
__device__ inline Ray xf_chain_ray(const READER& ents, uint32_t xf, const Ray& w) {
    Ray r = w;
    if (xf == XF_NONE) return r;
    uint4 a, b;
    ents.load(xf, a, b);
    if (b.x == XF_NONE) {  // one level: the common case
        xf_apply(a, r);
        return r;
    }
    int depth = 1;
    for (uint32_t x = b.x; x != XF_NONE && depth < XF_MAX_DEPTH; depth++) {
        uint4 pa, pb;
        ents.load(x, pa, pb);
        x = pb.x;
    }
    for (int level = depth - 1; level >= 0; level--) {  // level 0 is xf itself, depth - 1 the outermost
        uint32_t x = xf;
        uint4 ta, tb;
        ents.load(x, ta, tb);
        for (int k = 0; k < level; k++) {
            x = tb.x;
            ents.load(x, ta, tb);
        }
        xf_apply(ta, r);
    }
    return r;
}

// An ENT_INST entry: moves the ray from the frame the walk is in (the caller's: such entries exist only outside transform
// brackets) into the primitive's own frame through the entry's chain record -- Translate::hit (mesh.rs:332), RotateY::hit
// (mesh.rs:365-372), outermost first, the very operations the bracket entries of the instance would have applied -- and
// turns the entry into the plain one of its primitive.  a = direction.length_squared() in that frame (mesh.rs:116).
template <class READER>
__device__ __forceinline__ void inst_enter(const READER& ents, uint4& ea, uint4& eb, Ray& r, float& a) {
    const bool sph = (eb.w & 3u) == ENT_SPHERE;
    uint4 c0, c1;
    ents.load(sph ? ea.w : ea.y, c0, c1);
    xf_apply(c0, r);
    if (c1.x != 0u) xf_apply(c1, r);
    a = dot3(r.dx, r.dy, r.dz, r.dx, r.dy, r.dz);
    if (sph) ea.w = 0u;  // (the word held the chain; a sphere placed this way is at rest)
    eb.w &= ~ENT_INST;
}

// One step of the walk for the entry (ea, eb) at byte offset off; returns the next offset.
// QUICK: the primitive tests as the render kernels' fast walk evaluates them (hardware square root / reciprocal).
template <bool STRICT, int SLAB, bool QUICK = false, class READER>
__device__ __forceinline__ uint32_t walk_step(const DevScene& S, const READER& ents, uint4 ea, uint4 eb, uint32_t off, const Ray& r, const RayConst& rc,
                                              float tmin, Best& best, const uint32_t* __restrict__ rank) {
    if (eb.w == ENT_BOX) return slab_hit<SLAB>(ea, eb, r, rc, tmin, best.t) ? off + ENT_BYTES : eb.z;
    Ray tr = r;
    RayConst trc = rc;
    if (!STRICT && (eb.w & ENT_INST) != 0u) inst_enter(ents, ea, eb, tr, trc.a);
    const uint32_t kind = eb.w & 15u;
    if (kind == ENT_SPHERE) {
        sphere_test<STRICT, 5, QUICK>(ea, eb, tr, trc, tmin, best, off, rank);
    } else {
        const uint32_t pi = eb.w >> 4;
        const bool tri = kind == ENT_TRI;
        planar_test<STRICT, QUICK>(S.planar, tri ? pi + S.n_quads : pi, tri, tr, tmin, best, off, rank);
    }
    return off + ENT_BYTES;
}

// The part of ConstantMedium::hit behind its two boundary queries (mesh.rs:640-665): the boundary hits t1, t2 clipped to
// the interval [t_start, t_end) the medium was called with, the free-flight draw, the hit distance.  false: no hit.
__device__ __forceinline__ bool medium_flight(float t1, float t2, float t_start, float t_end, float medium_density, const Ray& ray, Rng& rng, float& distance) {
    if (t1 < t_start) t1 = t_start;
    if (t2 > t_end) t2 = t_end;
    if (t1 >= t2) return false;
    if (t1 < 0.0f) t1 = 0.0f;
    const float ray_length = __builtin_sqrtf(dot3(ray.dx, ray.dy, ray.dz, ray.dx, ray.dy, ray.dz));
    const float inside = (t2 - t1) * ray_length;
    const float density = -1.0f / medium_density;
    const float hit_distance = density * logf(rng_f32(rng));
    if (hit_distance > inside) return false;
    distance = t1 + hit_distance / ray_length;
    return true;
}

// One walk step at a bracket entry (ea, eb) at byte offset off.  `ray` is the ray in the current frame, `wray` the
// caller's ray; `main_tmin` the start of the caller's interval; the caller's interval end is the running best.
// FRAMES = false: the array holds no transform-enter / leave entries (media only); wray is not read.
// NEST: the array holds media inside other media's boundaries, or media whose boundary is one sphere (both get code
// that scenes without them should not carry: the parking code costs the frame-less render variants ten spilled registers,
// the one-step sphere boundary three); med(): the MedStack of this lane, asked for only at a medium inside a medium.
// QUICK: how the walk this step belongs to evaluates its sphere tests (the one-sphere boundary below uses the same).
template <bool STRICT, int SLAB, bool FRAMES = true, bool NEST = false, bool QUICK = false, class READER, class MED>
__device__ inline void bracket_step(const READER& ents, const uint4& ea, const uint4& eb, uint32_t& off, Ray& ray, RayConst& rc, const Ray& wray,
                                    Best& best, WalkX& x, const MED& med, float main_tmin, Rng& rng, const uint32_t* __restrict__ rank) {
    const uint32_t kind = eb.w & 15u;
    if (FRAMES && kind == ENT_XF_PUSH) {
        x.cur_xf = off;
        xf_apply(ea, ray);
        rc = ray_const<SLAB>(ray);
        off += ENT_BYTES;
    } else if (FRAMES && kind == ENT_XF_POP) {
        x.cur_xf = eb.x;
        ray = xf_chain_ray(ents, x.cur_xf, wray);
        rc = ray_const<SLAB>(ray);
        off += ENT_BYTES;
    } else if (!STRICT && NEST && kind == ENT_MED_BEGIN && eb.x != 0u) {
        // A medium whose boundary is one sphere (the entry behind this one; the fast builder says so in word 4): both
        // boundary queries (mesh.rs:635-638: the whole line, then from just behind the first hit) are that sphere's test
        // with those two intervals, made here in one step instead of five trips through the entries.  The interval the
        // medium is called with is the walk's current one: x.tmin (the caller's start, or inside another medium the
        // start of the query that came by) to the running best.
        uint4 sa, sb;
        ents.load(off + ENT_BYTES, sa, sb);
        const uint32_t end_off = eb.z;
        Best q1;
        q1.t = __builtin_inff();
        q1.ent = NO_HIT;
        sphere_test<false, 5, QUICK>(sa, sb, ray, rc, -__builtin_inff(), q1, off + ENT_BYTES, rank);
        if (q1.ent != NO_HIT) {
            Best q2;
            q2.t = __builtin_inff();
            q2.ent = NO_HIT;
            sphere_test<false, 5, QUICK>(sa, sb, ray, rc, q1.t + 0.0001f, q2, off + ENT_BYTES, rank);
            float distance = 0.0f;
            if (q2.ent != NO_HIT && medium_flight(q1.t, q2.t, x.tmin, best.t, as_f(ea.x), ray, rng, distance) &&
                in_range<false>(distance, x.tmin, best, end_off, rank)) {
                best.t = distance;
                best.ent = end_off;
                x.best_xf = x.cur_xf;
            }
        }
        off = end_off + ENT_BYTES;
    } else if (kind == ENT_MED_BEGIN) {
        // ConstantMedium::hit (mesh.rs:634-638): first boundary query over the whole line
        if (NEST && ea.w != 0u) {  // inside another medium's boundary query: that query's state waits at this entry's level
            const MedStack m = med();
            uint32_t* s = m.rec(ea.w);
            s[0] = __float_as_uint(x.sv_t);
            s[m.lanes] = x.sv_ent;
            s[2u * m.lanes] = x.sv_xf;
            s[3u * m.lanes] = __float_as_uint(x.med_t1);
            s[4u * m.lanes] = __float_as_uint(x.tmin);
        }
        x.sv_t = best.t;
        x.sv_ent = best.ent;
        x.sv_xf = x.best_xf;
        best.t = __builtin_inff();
        best.ent = NO_HIT;
        x.tmin = -__builtin_inff();
        x.med_t1 = __builtin_nanf("");
        off += ENT_BYTES;
    } else {  // ENT_MED_END
        if (x.med_t1 != x.med_t1 && best.ent != NO_HIT) {
            // second boundary query, from just behind the first hit (mesh.rs:636-638)
            x.med_t1 = best.t;
            x.tmin = best.t + 0.0001f;
            best.t = __builtin_inff();
            best.ent = NO_HIT;
            off = eb.z + ENT_BYTES;
        } else {
            bool some = false;
            float distance = 0.0f;
            // the interval this medium's hit() was called with starts where its caller's does: the walk's own interval, or,
            // inside another medium, the start of the boundary query that came by (mesh.rs:635 -inf, :638 hit1 + 0.0001)
            const uint32_t level = NEST ? ea.x : 0u;
            if (NEST && level != 0u) {
                const MedStack m = med();
                main_tmin = __uint_as_float(m.rec(level)[4u * m.lanes]);
            }
            if (x.med_t1 == x.med_t1 && best.ent != NO_HIT) {  // both boundary hits exist (mesh.rs:640-665)
                uint4 ba, bb;
                ents.load(eb.z, ba, bb);
                some = medium_flight(x.med_t1, best.t, main_tmin, x.sv_t, as_f(ba.x), ray, rng, distance);
            }
            best.t = x.sv_t;
            best.ent = x.sv_ent;
            x.best_xf = x.sv_xf;
            x.tmin = main_tmin;
            x.med_t1 = __builtin_nanf("");
            if (NEST && level != 0u) {  // back in the enclosing medium's boundary query
                const MedStack m = med();
                const uint32_t* outer = m.rec(level);
                x.sv_t = __uint_as_float(outer[0]);
                x.sv_ent = outer[m.lanes];
                x.sv_xf = outer[2u * m.lanes];
                x.med_t1 = __uint_as_float(outer[3u * m.lanes]);
            }
            // strict: the caller takes whatever the medium returns (mesh.rs:76-78, :611-616); fast: closest hit, reference tie order
            if (some && (STRICT || in_range<false>(distance, main_tmin, best, off, rank))) {
                best.t = distance;
                best.ent = off;
                x.best_xf = x.cur_xf;
            }
            off += ENT_BYTES;
        }
    }
}

// Hit record of the accepted entry at best.ent when the scene has brackets: evaluate it with the ray of its own
// frame, then carry point and normal out through the enclosing transforms as the reference does -- Translate adds its
// offset to the point (mesh.rs:334); RotateY re-applies its FORWARD rotation and reads the already overwritten x
// (mesh.rs:378-382).
template <bool WANT_UV, bool EXACT, bool INST = true, class READER>
__device__ inline HitRec make_hit_x(const DevScene& S, const READER& ents, const Ray& wray, const Best& best, uint32_t best_xf) {
    uint4 ha, hb;
    ents.load(best.ent, ha, hb);
    auto carry_out = [](const uint4& a, HitRec& h) {  // one level, as its hit() returns (mesh.rs:334, :378-382)
        if (a.x == 1u) {
            h.px += as_f(a.y);
            h.py += as_f(a.z);
            h.pz += as_f(a.w);
        } else {
            const float s = as_f(a.y), c = as_f(a.z);
            h.px = c * h.px - s * h.pz;
            h.pz = s * h.px + c * h.pz;
            h.nx = c * h.nx - s * h.nz;
            h.nz = s * h.nx + c * h.nz;
        }
    };
    if (INST && (hb.w & 15u) >= ENT_INST) {  // primitive of an instance placed in the caller's frame: its chain record says how it got there
        const uint32_t chain = (hb.w & 3u) == ENT_SPHERE ? ha.w : ha.y;
        Ray lr = wray;
        float a_unused;
        inst_enter(ents, ha, hb, lr, a_unused);
        HitRec h = make_hit<WANT_UV, EXACT>(S, ha, hb, lr, best.t);
        uint4 c0, c1;
        ents.load(chain, c0, c1);
        if (c1.x != 0u) carry_out(c1, h);
        carry_out(c0, h);
        return h;
    }
    const Ray lr = xf_chain_ray(ents, best_xf, wray);
    HitRec h;
    if ((hb.w & 15u) == ENT_MED_END) {  // mesh.rs:667-674
        uint4 ba, bb;
        ents.load(hb.z, ba, bb);
        h.px = lr.ox + lr.dx * best.t;
        h.py = lr.oy + lr.dy * best.t;
        h.pz = lr.oz + lr.dz * best.t;
        h.nx = 1.0f;
        h.ny = 0.0f;
        h.nz = 0.0f;
        h.u = 0.0f;
        h.v = 0.0f;
        h.mat = ba.y;
        h.prim_id = ba.z;
    } else {
        h = make_hit<WANT_UV, EXACT>(S, ha, hb, lr, best.t);
    }
    uint32_t xf = best_xf;
    for (int level = 0; level < XF_MAX_DEPTH && xf != XF_NONE; level++) {  // innermost first, as the calls return
        uint4 a, b;
        ents.load(xf, a, b);
        carry_out(a, h);
        xf = b.x;
    }
    return h;
}

// ---------------------------------------------------------------------------------------------- textures
// Rust `as i32` saturates and maps NaN to 0 -- exactly what v_cvt_i32_f32 does in hardware.
__device__ __forceinline__ int32_t as_i32_sat(float v) {
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t as_u32_sat(float v) {
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
__device__ __forceinline__ float clamp01(float v) {  // Interval(0..1).clamp (lib.rs:97-99); NaN stays NaN
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return v;
}

// Perlin::noise / perlin_interp / turb (texture.rs:187-243)
// TAB is a pointer to one table set (768 floats, then 3 * 256 permutation words) in device memory or in LDS.
template <class TAB>
__device__ inline float perlin_noise(TAB tab, float px, float py, float pz) {
    float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    float u = px - fx, v = py - fy, w = pz - fz;
    int32_t i = as_i32_sat(fx), j = as_i32_sat(fy), k = as_i32_sat(fz);
    float uu = u * u * (3.0f - 2.0f * u);
    float vv = v * v * (3.0f - 2.0f * v);
    float ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
    // the two x-planes of the lattice cell are taken one after the other: fully unrolled, the 32 table loads of a cell
    // are all hoisted to the front and hold ~30 more registers across the shading code (3 instead of 4 waves per SIMD)
#pragma unroll 1
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                uint32_t g = __float_as_uint(tab[768 + ((uint32_t)(i + di) & 255u)]) ^ __float_as_uint(tab[1024 + ((uint32_t)(j + dj) & 255u)]) ^
                             __float_as_uint(tab[1280 + ((uint32_t)(k + dk) & 255u)]);
                float cx = tab[3 * g], cy = tab[3 * g + 1], cz = tab[3 * g + 2];
                float fi = (float)di, fj = (float)dj, fk = (float)dk;
                accum += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) * (fk * ww + (1.0f - fk) * (1.0f - ww)) *
                         dot3(cx, cy, cz, u - fi, v - fj, w - fk);
            }
    return accum;
}
template <class TAB>
__device__ inline float perlin_turb(TAB tab, float px, float py, float pz, int depth) {
    float accum = 0.0f, weight = 1.0f;
    for (int n = 0; n < depth; n++) {
        accum += weight * perlin_noise(tab, px, py, pz);
        px = px * 2.0f;
        py = py * 2.0f;
        pz = pz * 2.0f;
        weight *= 0.5f;
    }
    return fabsf(accum);
}

// Texture::value (texture.rs:34-37, :63-74, :100-114, :132-137)
// lds_perlin: LDS byte address of a copy of Perlin table set 0, or 0xFFFFFFFF when the block keeps none.
template <bool FULL>
__device__ inline void tex_value(const DevScene& S, uint32_t tex, float u, float v, float px, float py, float pz, float& r, float& g, float& b,
                                 uint32_t lds_perlin = 0xFFFFFFFFu) {
    uint4 t0 = S.texs[2 * tex];
    for (int guard = 0; guard < 16 && t0.x == RTX_TEX_CHECKER; guard++) {
        float scale = as_f(t0.y);
        int32_t x = as_i32_sat(floorf(px * scale));
        int32_t y = as_i32_sat(floorf(py * scale));
        int32_t z = as_i32_sat(floorf(pz * scale));
        int32_t sum = (int32_t)((uint32_t)x + (uint32_t)y + (uint32_t)z);
        tex = (sum % 2 == 0) ? t0.z : t0.w;
        t0 = S.texs[2 * tex];
    }
    if (t0.x == RTX_TEX_SOLID) {
        uint4 t1 = S.texs[2 * tex + 1];
        r = as_f(t1.x);
        g = as_f(t1.y);
        b = as_f(t1.z);
    } else if (FULL && t0.x == RTX_TEX_IMAGE) {
        DevImage im = S.images[t0.z];
        float uc = clamp01(u);
        float vc = 1.0f - clamp01(v);
        uint32_t i = as_u32_sat(uc * ((float)im.width - 1.0f));
        uint32_t j = as_u32_sat(vc * ((float)im.height - 1.0f));
        const uint8_t* px8 = im.rgb + ((size_t)j * im.width + i) * 3;
        const float color_scale = 1.0f / 255.0f;
        r = (float)px8[0] * color_scale;
        g = (float)px8[1] * color_scale;
        b = (float)px8[2] * color_scale;
    } else if (FULL && t0.x == RTX_TEX_NOISE) {
        float turb;
        if (t0.z == 0u && lds_perlin != 0xFFFFFFFFu) {  // 336 table reads per value: from LDS when the table is there
            turb = perlin_turb((__attribute__((address_space(3))) const float*)(uintptr_t)lds_perlin, px, py, pz, 7);
        } else {
            turb = perlin_turb(S.perlin + (size_t)t0.z * 1536, px, py, pz, 7);
        }
        float s = 1.0f + sinf(as_f(t0.y) * pz + 10.0f * turb);
        r = g = b = (1.0f * 0.5f) * s;
    } else {
        r = g = b = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------- materials
// material.rs:13-17; powi(5) as llvm.powi expands it: x * ((x*x)*(x*x))
__device__ __forceinline__ float reflectance(float cosine, float ri) {
    float r0 = (1.0f - ri) * __builtin_amdgcn_rcpf(1.0f + ri);
    r0 *= r0;
    float b = 1.0f - cosine;
    float b2 = b * b;
    float b4 = b2 * b2;
    return r0 + (1.0f - r0) * (b * b4);
}

struct ScatterOut {
    bool scattered;
    float ar, ag, ab;  // attenuation
    float er, eg, eb;  // emitted
    Ray ray;
};

// Material::emitted + Material::scatter for the hit (camera.rs:128-130; material.rs:70-96, :104-119, :149-173,
// :198-206, :231-240), arranged for a wave whose lanes hold different materials: the two random draws, the unit
// vector and the texture lookup that several materials need are evaluated once, ahead of the per-material part.
// Every scatter consumes exactly two draws of the path's stream (the dielectric uses the first as its xi).
// Shading arithmetic only has to agree with the reference within tolerance (the streams differ anyway), so this
// function may fuse multiply-adds and uses the hardware reciprocal / reciprocal square root.
// `m` = (kind, texture, param bits, flags) as in DevScene::mats.  With `solid` the texture colour is already known
// (rgb): the common case of a solid colour needs no dependent table reads at all.
template <bool FULL>
__device__ inline ScatterOut shade(const DevScene& S, const Ray& rin, const HitRec& h, Rng& rng, const uint4 m, const bool solid, const float sr,
                                   const float sg, const float sb, const uint32_t lds_perlin = 0xFFFFFFFFu) {
#pragma clang fp contract(fast)
    ScatterOut o;
    o.scattered = false;
    o.ar = o.ag = o.ab = 0.0f;
    o.er = o.eg = o.eb = 0.0f;
    o.ray.ox = h.px;
    o.ray.oy = h.py;
    o.ray.oz = h.pz;
    o.ray.time = rin.time;
    // shared by all materials: two draws, a uniform unit vector (utils.rs:37-39), the texture colour
    const float u1 = rng_f32(rng);
    const float u2 = rng_f32(rng);
    const float rz = 1.0f - 2.0f * u1;
    const float rr = __builtin_amdgcn_sqrtf(fmaxf(0.0f, 1.0f - rz * rz));
    const float rx = rr * __builtin_amdgcn_cosf(u2);
    const float ry = rr * __builtin_amdgcn_sinf(u2);
    float tr = sr, tg = sg, tb = sb;
    if (!solid) tex_value<FULL>(S, m.y, h.u, h.v, h.px, h.py, h.pz, tr, tg, tb, lds_perlin);
    float dx = 0.0f, dy = 0.0f, dz = 0.0f;
    if (m.x == RTX_MAT_LAMBERTIAN) {
        dx = h.nx + rx;
        dy = h.ny + ry;
        dz = h.nz + rz;
        const float s = 1e-8f;  // near_zero (utils.rs:41-44)
        if (fabsf(dx) < s && fabsf(dy) < s && fabsf(dz) < s) {
            dx = h.nx;
            dy = h.ny;
            dz = h.nz;
        }
        o.scattered = true;
    } else if (m.x == RTX_MAT_METAL) {
        // reflect(ray.direction, normal) on the UN-normalised direction, plus roughness * unit vector (material.rs:107-108)
        const float k = 2.0f * dot3(rin.dx, rin.dy, rin.dz, h.nx, h.ny, h.nz);
        const float rough = as_f(m.z);
        dx = (rin.dx - k * h.nx) + (rough * rx);
        dy = (rin.dy - k * h.ny) + (rough * ry);
        dz = (rin.dz - k * h.nz) + (rough * rz);
        o.scattered = dot3(dx, dy, dz, h.nx, h.ny, h.nz) > 0.0f;
    } else if (m.x == RTX_MAT_DIELECTRIC) {
        // Hit::front_face is always true (mesh.rs:47) so ri = 1 / refraction_index on entry AND exit (material.rs:151-155)
        const float ri = __builtin_amdgcn_rcpf(as_f(m.z));
        const float inv_len = __builtin_amdgcn_rsqf(dot3(rin.dx, rin.dy, rin.dz, rin.dx, rin.dy, rin.dz));
        const float ux = rin.dx * inv_len, uy = rin.dy * inv_len, uz = rin.dz * inv_len;
        const float udn = dot3(ux, uy, uz, h.nx, h.ny, h.nz);
        const float cos_theta = -fminf(udn, 1.0f);
        const float sin_theta = __builtin_amdgcn_sqrtf(1.0f - cos_theta * cos_theta);
        const bool cannot_refract = ri * sin_theta > 1.0f;
        if (cannot_refract || reflectance(cos_theta, ri) > u1) {
            const float k = 2.0f * udn;
            dx = ux - k * h.nx;
            dy = uy - k * h.ny;
            dz = uz - k * h.nz;
        } else {  // refract (material.rs:23-28)
            const float ppx = ri * (ux + cos_theta * h.nx), ppy = ri * (uy + cos_theta * h.ny), ppz = ri * (uz + cos_theta * h.nz);
            const float par = -__builtin_amdgcn_sqrtf(fabsf(1.0f - dot3(ppx, ppy, ppz, ppx, ppy, ppz)));
            dx = ppx + par * h.nx;
            dy = ppy + par * h.ny;
            dz = ppz + par * h.nz;
        }
        o.scattered = true;
    } else if (m.x == RTX_MAT_DIFFUSE_LIGHT) {
        o.er = tr;
        o.eg = tg;
        o.eb = tb;
    } else if (m.x == RTX_MAT_ISOTROPIC) {
        dx = rx;
        dy = ry;
        dz = rz;
        o.scattered = true;
    }
    o.ar = tr;
    o.ag = tg;
    o.ab = tb;
    o.ray.dx = dx;
    o.ray.dy = dy;
    o.ray.dz = dz;
    return o;
}

// bevy_color Srgba::gamma_function_inverse + to_u8_array_no_alpha (camera.rs:101)
__device__ __forceinline__ uint8_t srgb_u8(float value) {
    float s;
    if (value <= 0.0f)
        s = value;
    else if (value <= 0.0031308f)
        s = value * 12.92f;
    else
        s = (1.055f * powf(value, 1.0f / 2.4f)) - 0.055f;
    if (s < 0.0f) s = 0.0f;
    if (s > 1.0f) s = 1.0f;
    float q = roundf(s * 255.0f);
    if (!(q == q)) return 0;
    return (uint8_t)q;
}

}  // namespace rtxd
