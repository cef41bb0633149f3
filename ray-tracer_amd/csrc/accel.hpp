// Host-side construction of the traversal data the kernels walk.
//
// Both traversal modes use the same device structure: a "threaded" array of 32-byte entries in depth-first
// order.  An entry is a box (six bounds + where to jump when the ray misses it; a hit falls through to the next
// entry), a primitive (spheres inline, planar primitives by index) or a bracket (transform enter / leave, medium
// begin / end).  Walking the array front to back therefore IS a left-first depth-first traversal without a stack.
//
//  strict: the array is the reference's own tree (Bvh::from, mesh.rs:560-605, handed over in bvh_nodes) with the
//          reference's padded, origin-including boxes, so Bvh::hit (mesh.rs:608-620) is reproduced test for test,
//          cull for cull; World / Cube lists are scanned linearly as World::hit does (mesh.rs:66-82).
//  fast  : binned-SAH trees built here over tight (slightly padded) boxes -- one over the top level, one inside every
//          Translate / RotateY / ConstantMedium -- whose closest hit is order independent; ties are resolved with
//          the rank a primitive has in the strict order.
//
// Word 7 of an entry doubles as the walk state of a lane standing at it:
//   0                      box         words 0-5 bounds, word 6 = entry to jump to on a miss
//   (index << 4) | kind    kind 1 sphere (words 0-6 = c0, dc, r), 2 quad, 3 triangle (planar record index)
//                          kind 4 transform enter: word 0 = 1 translate (words 1-3 offset) | 2 rotate-y (word 1 sin,
//                                 word 2 cos); word 4 = enclosing transform-enter entry or NONE
//                          kind 5 transform leave: word 4 = enclosing transform-enter entry or NONE
//                          kind 6 medium begin:    word 0 density, word 1 material, word 2 prim id, word 3 nesting level, word 4 = 1 when
//                                                  its boundary is the one sphere entry behind it
//                                                  (number of media whose boundary it lies in), word 6 = its end
//                          kind 7 medium end:      word 0 nesting level, word 6 = its begin
//   ENT_END                terminator appended after the last entry: the walk is complete
// Links are entry indices while building and byte offsets after finalize().
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "rtx.h"

namespace rtx_accel {

enum : uint32_t { ENT_BOX = 0, ENT_SPHERE = 1, ENT_QUAD = 2, ENT_TRI = 3, ENT_XF_PUSH = 4, ENT_XF_POP = 5, ENT_MED_BEGIN = 6, ENT_MED_END = 7 };
// Fast arrays only: ENT_INST | ENT_SPHERE / ENT_QUAD / ENT_TRI is a primitive of a Translate / RotateY instance placed in
// the tree of the enclosing frame under a box of its own (in that frame's coordinates) instead of behind the
// instance's brackets.  The entry is the primitive's own (object space) with the index of its chain record in a spare
// word -- w[3] of a sphere (only spheres that do not move are placed this way), w[1] of a quad / triangle; the test moves
// the ray into object space through the chain, operation for operation what the bracket entries would have done, so
// hits are the same bit for bit.  A chain record is one entry behind the terminator: two levels of (kind 1 = translate
// | 2 = rotate-y | 0 = none, three parameters), outermost first.
constexpr uint32_t ENT_INST = 8;
constexpr int MAX_INST_LEVELS = 2;
constexpr uint32_t ENT_END = 0xFFFFFFFEu;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int MAX_MED_DEPTH = 4;  // ConstantMedium boundaries nested inside one another (a walk saves one state record per enclosing medium)
constexpr int MAX_XF_DEPTH = 64;  // nested Translate / RotateY levels accepted (the kernels walk the chain of enclosing transforms)

struct Ent {
    uint32_t w[8];
};
static_assert(sizeof(Ent) == 32, "entry must be 32 bytes");

struct Threaded {
    std::vector<Ent> ents;
    std::vector<uint32_t> rank;  // per entry: tie-break rank of a primitive / medium entry (first position in the strict order)
    std::vector<uint32_t> mat;   // per entry: material of a primitive / medium-end entry (what a hit there is shaded with), else NONE
    uint32_t n_boxes = 0, n_prims = 0;
    bool has_brackets = false;   // contains transform or medium entries (or ENT_INST primitives)
    bool has_bare = false;       // some subtree stands without a box of its own (see Builder::never_culls): not a strict binary tree of boxes
    bool sphere_media = false;   // some medium's boundary is a single sphere (its begin entry says so in word 4)
    int med_levels = 0;          // deepest nesting of media: 1 = media side by side only, 2 = one inside another's boundary, ...
    bool has_frames = false;     // contains transform-enter / leave entries: a walk changes frames
    std::vector<uint32_t> inner; // fast only: boxes whose two children are boxes that may be visited in either order
    std::vector<Ent> chains;     // fast only: chain records of the ENT_INST entries (appended behind the terminator by finalize)
};

static inline uint32_t fbits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static inline Ent box_entry(const float b[6], uint32_t skip) {
    Ent e;
    for (int i = 0; i < 6; i++) e.w[i] = fbits(b[i]);
    e.w[6] = skip;
    e.w[7] = ENT_BOX;
    return e;
}

static inline Ent prim_entry(const rtx_scene_desc_t& d, uint32_t kind, uint32_t index) {
    Ent e;
    std::memset(&e, 0, sizeof(e));
    if (kind == RTX_REF_SPHERE) {
        e.w[0] = fbits(d.sph_cx[index]); e.w[1] = fbits(d.sph_cy[index]); e.w[2] = fbits(d.sph_cz[index]);
        e.w[3] = fbits(d.sph_dx[index]); e.w[4] = fbits(d.sph_dy[index]); e.w[5] = fbits(d.sph_dz[index]);
        e.w[6] = fbits(d.sph_radius[index]);
        e.w[7] = ENT_SPHERE | (index << 4);
    } else if (kind == RTX_REF_QUAD) {
        e.w[0] = index;
        e.w[7] = ENT_QUAD | (index << 4);
    } else {
        e.w[0] = index;
        e.w[7] = ENT_TRI | (index << 4);
    }
    return e;
}

struct Box3 {
    float lo[3], hi[3];
    void reset() {
        const float INF = std::numeric_limits<float>::infinity();
        for (int a = 0; a < 3; a++) { lo[a] = INF; hi[a] = -INF; }
    }
    void grow(const Box3& o) {
        for (int a = 0; a < 3; a++) { lo[a] = std::fmin(lo[a], o.lo[a]); hi[a] = std::fmax(hi[a], o.hi[a]); }
    }
    bool valid() const {
        for (int a = 0; a < 3; a++)
            if (lo[a] != lo[a] || hi[a] != hi[a]) return false;
        return true;
    }
    float max_abs() const {
        float m = 0;
        for (int a = 0; a < 3; a++) m = std::fmax(m, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
        return m;
    }
};

// Tight bounds of a primitive (spheres over their whole motion).
static inline Box3 prim_box(const rtx_scene_desc_t& d, uint32_t kind, uint32_t i) {
    Box3 b;
    if (kind == RTX_REF_SPHERE) {
        const float c0[3] = {d.sph_cx[i], d.sph_cy[i], d.sph_cz[i]};
        const float dc[3] = {d.sph_dx[i], d.sph_dy[i], d.sph_dz[i]};
        float r = std::fabs(d.sph_radius[i]);
        for (int a = 0; a < 3; a++) {
            float c1 = c0[a] + dc[a];  // centre over time in [0,1) sweeps c0 .. c0+dc
            b.lo[a] = std::fmin(c0[a], c1) - r;
            b.hi[a] = std::fmax(c0[a], c1) + r;
        }
    } else if (kind == RTX_REF_QUAD) {
        for (int a = 0; a < 3; a++) {
            float q = d.quad_f[a][i], u = d.quad_f[3 + a][i], v = d.quad_f[6 + a][i];
            float p[4] = {q, q + u, q + v, q + u + v};
            b.lo[a] = std::fmin(std::fmin(p[0], p[1]), std::fmin(p[2], p[3]));
            b.hi[a] = std::fmax(std::fmax(p[0], p[1]), std::fmax(p[2], p[3]));
        }
    } else {
        for (int a = 0; a < 3; a++) {
            float p0 = d.tri_f[a][i], p1 = d.tri_f[3 + a][i], p2 = d.tri_f[6 + a][i];
            b.lo[a] = std::fmin(p0, std::fmin(p1, p2));
            b.hi[a] = std::fmax(p0, std::fmax(p1, p2));
        }
    }
    return b;
}

// Outward padding of a fast-mode box.  The slab test and the primitive tests round differently, so a box that is
// exactly tight could cull a ray the primitive test would still accept: 1e-3 of the box diagonal, a few ulps of the
// coordinates, and the absolute error of the fused box tests (two rounding steps of ~1.2e-7 * (|origin| + |plane|) each in
// the centre / half-extent form: <= 4.8e-7 * range).
static inline void pad_box(Box3& b, float coord_range) {
    float diag2 = 0;
    for (int a = 0; a < 3; a++) diag2 += (b.hi[a] - b.lo[a]) * (b.hi[a] - b.lo[a]);
    float pad = std::isfinite(diag2) ? 1e-3f * std::sqrt(diag2) : 0.0f;
    for (int a = 0; a < 3; a++) {
        float m = std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
        float p = pad + 6e-7f * m + 5e-7f * coord_range + 1e-30f;
        b.lo[a] -= p;
        b.hi[a] += p;
    }
}

struct Unit {  // something a fast tree places under a leaf box: a primitive or a whole bracketed object
    Box3 box;
    rtx_ref_t ref;
    uint32_t rank;
    uint32_t chain = 0xFFFFFFFFu;  // primitive of an instance placed in the enclosing frame: its chain record
};

class Builder {
  public:
    const rtx_scene_desc_t& d;
    std::string err;
    bool unsupported = false;
    std::unordered_map<uint32_t, uint32_t> first_rank;  // prim / medium ref -> first position among the strict order's tests
    uint32_t MAX_LEAF = 4;  // most primitives under one leaf box
    float C_TRAV = 1.0f;    // cost of a box test in units of a sphere test
    float PLANAR_COST = 1.0f;  // cost of a quad / triangle test in the same units: the caller says (RTX_SAH_PLANAR_COST overrides)
    float coord_range = 0;

    explicit Builder(const rtx_scene_desc_t& d_) : d(d_) {}

    bool fail(const char* m, bool unsup = false) {
        if (err.empty()) err = m;
        unsupported = unsupported || unsup;
        return false;
    }

    // ------------------------------------------------------------------------------------------ shared pieces
    bool check_ref(rtx_ref_t r) {
        uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        switch (k) {
            case RTX_REF_NODE: return i < d.n_bvh_nodes || fail("bvh node reference out of range");
            case RTX_REF_SPHERE: return i < d.n_spheres || fail("sphere reference out of range");
            case RTX_REF_QUAD: return i < d.n_quads || fail("quad reference out of range");
            case RTX_REF_TRI: return i < d.n_tris || fail("triangle reference out of range");
            case RTX_REF_OBJECT: return i < d.n_objects || fail("object reference out of range");
        }
        return fail("unknown reference kind");
    }
    static bool is_prim(rtx_ref_t r) {
        uint32_t k = RTX_REF_KIND(r);
        return k == RTX_REF_SPHERE || k == RTX_REF_QUAD || k == RTX_REF_TRI;
    }

    void push_prim(Threaded& out, rtx_ref_t r, uint32_t rank) {
        out.ents.push_back(prim_entry(d, RTX_REF_KIND(r), RTX_REF_INDEX(r)));
        out.rank.push_back(rank);
        const uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        out.mat.push_back(k == RTX_REF_SPHERE ? d.sph_material[i] : k == RTX_REF_QUAD ? d.quad_material[i] : d.tri_material[i]);
        out.n_prims++;
    }

    // Emit a Translate / RotateY / ConstantMedium object: opening bracket, its child through `child_fn`, closing bracket.
    template <class F>
    bool emit_bracketed(Threaded& out, const rtx_object_t& o, uint32_t obj_index, uint32_t rank, uint32_t enclosing_xf, int xf_depth,
                        int med_depth, F child_fn) {
        out.has_brackets = true;
        Ent a, b;
        std::memset(&a, 0, sizeof(a));
        std::memset(&b, 0, sizeof(b));
        uint32_t at = (uint32_t)out.ents.size();
        if (o.kind == RTX_OBJ_MEDIUM) {
            if (med_depth >= MAX_MED_DEPTH) return fail("ConstantMedium boundaries nested deeper than 4 levels", true);
            if (o.material >= d.n_materials) return fail("medium material out of range");
            if (!(o.f[0] == o.f[0])) return fail("NaN medium density");
            a.w[0] = fbits(o.f[0]);
            a.w[1] = o.material;
            a.w[2] = o.prim_id;
            out.med_levels = std::max(out.med_levels, med_depth + 1);
            a.w[3] = (uint32_t)med_depth;  // media whose boundary this one lies in: the level its saved walk state goes to
            a.w[7] = ENT_MED_BEGIN | (obj_index << 4);
            out.ents.push_back(a);
            out.rank.push_back(rank);
            out.mat.push_back(NONE);
            if (!child_fn(o.child, enclosing_xf, xf_depth, med_depth + 1)) return false;
            b.w[0] = (uint32_t)med_depth;
            // (the boundary is one sphere, nothing else: the fast walk then makes both boundary queries in one step)
            if (out.ents.size() == (size_t)at + 2 && out.ents[at + 1].w[7] != ENT_BOX && (out.ents[at + 1].w[7] & 15u) == ENT_SPHERE) {
                out.ents[at].w[4] = 1u;
                out.sphere_media = true;
            }
            b.w[6] = at;
            b.w[7] = ENT_MED_END | (obj_index << 4);
            out.ents[at].w[6] = (uint32_t)out.ents.size();
            out.ents.push_back(b);
            out.rank.push_back(rank);
            out.mat.push_back(o.material);
            out.n_prims++;
        } else {
            if (xf_depth >= MAX_XF_DEPTH) return fail("Translate / RotateY nested deeper than 64 levels", true);
            out.has_frames = true;
            a.w[0] = o.kind == RTX_OBJ_TRANSLATE ? 1u : 2u;
            a.w[1] = fbits(o.f[0]);
            a.w[2] = fbits(o.f[1]);
            a.w[3] = fbits(o.f[2]);
            a.w[4] = enclosing_xf;
            a.w[7] = ENT_XF_PUSH | (obj_index << 4);
            out.ents.push_back(a);
            out.rank.push_back(0);
            out.mat.push_back(NONE);
            if (!child_fn(o.child, at, xf_depth + 1, med_depth)) return false;
            b.w[4] = enclosing_xf;
            b.w[7] = ENT_XF_POP | (obj_index << 4);
            out.ents[at].w[6] = (uint32_t)out.ents.size();
            out.ents.push_back(b);
            out.rank.push_back(0);
            out.mat.push_back(NONE);
        }
        return true;
    }

    // ------------------------------------------------------------------------------------------ strict
    uint32_t next_rank = 0;
    int depth = 0;
    bool emit_strict(Threaded& out, rtx_ref_t r, uint32_t enclosing_xf, int xf_depth, int med_depth) {
        if (!check_ref(r)) return false;
        if (++depth > 4096) return fail("scene graph too deep (cycle?)");
        uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        bool ok = true;
        if (k == RTX_REF_NODE) {
            const rtx_bvh_node_t& n = d.bvh_nodes[i];
            for (int c = 0; c < 6; c++)
                if (n.bounds[c] != n.bounds[c]) ok = fail("NaN in bvh node bounds");
            if (ok) {
                uint32_t at = (uint32_t)out.ents.size();
                out.ents.push_back(box_entry(n.bounds, 0));
                out.rank.push_back(0);
                out.mat.push_back(NONE);
                out.n_boxes++;
                // a one-element Bvh has left == right (mesh.rs:572-577): the reference tests it twice, so do we
                ok = emit_strict(out, n.left, enclosing_xf, xf_depth, med_depth) && emit_strict(out, n.right, enclosing_xf, xf_depth, med_depth);
                out.ents[at].w[6] = (uint32_t)out.ents.size();
            }
        } else if (is_prim(r)) {
            auto it = first_rank.find(r);
            uint32_t rk = it == first_rank.end() ? (first_rank[r] = next_rank) : it->second;
            next_rank++;
            push_prim(out, r, rk);
        } else {
            const rtx_object_t& o = d.objects[i];
            if (o.kind == RTX_OBJ_LIST) {
                // World::hit (mesh.rs:66-82) is a linear scan with a shrinking interval and no box test
                if ((uint64_t)o.first + o.count > d.n_refs) ok = fail("list object exceeds refs");
                for (uint32_t c = 0; c < o.count && ok; c++) ok = emit_strict(out, d.refs[o.first + c], enclosing_xf, xf_depth, med_depth);
            } else if (o.kind == RTX_OBJ_TRANSLATE || o.kind == RTX_OBJ_ROTATE_Y || o.kind == RTX_OBJ_MEDIUM) {
                uint32_t rk = 0;
                if (o.kind == RTX_OBJ_MEDIUM) {
                    auto it = first_rank.find(r);
                    rk = it == first_rank.end() ? (first_rank[r] = next_rank) : it->second;
                    next_rank++;
                }
                ok = emit_bracketed(out, o, i, rk, enclosing_xf, xf_depth, med_depth,
                                    [&](rtx_ref_t c, uint32_t xf, int xd, int med) { return emit_strict(out, c, xf, xd, med); });
            } else {
                ok = fail("unknown object kind");
            }
        }
        depth--;
        return ok;
    }

    // ------------------------------------------------------------------------------------------ fast
    // Bounds, in the enclosing frame, of a box given in the frame of a Translate / RotateY object.
    static Box3 xf_box(const rtx_object_t& o, const Box3& c) {
        Box3 out;
        out.reset();
        if (o.kind == RTX_OBJ_TRANSLATE) {
            for (int a = 0; a < 3; a++) { out.lo[a] = c.lo[a] + o.f[a]; out.hi[a] = c.hi[a] + o.f[a]; }
        } else {
            // object -> world rotation of the eight corners (mesh.rs:405-423)
            const float s = o.f[0], co = o.f[1];
            for (int ci = 0; ci < 8; ci++) {
                float x = (ci & 1) ? c.hi[0] : c.lo[0], y = (ci & 2) ? c.hi[1] : c.lo[1], z = (ci & 4) ? c.hi[2] : c.lo[2];
                float p[3] = {co * x + s * z, y, -s * x + co * z};
                for (int a = 0; a < 3; a++) { out.lo[a] = std::fmin(out.lo[a], p[a]); out.hi[a] = std::fmax(out.hi[a], p[a]); }
            }
            float m = out.max_abs();
            for (int a = 0; a < 3; a++) { out.lo[a] -= 1e-6f * m; out.hi[a] += 1e-6f * m; }
        }
        return out;
    }

    // Instances whose primitives are placed in the enclosing frame's tree (ENT_INST): a Translate / RotateY, at most
    // MAX_INST_LEVELS of them nested, over nothing but primitives (spheres at rest), lists and Bvh nodes.
    bool place_instances = true;
    std::vector<std::vector<uint32_t>> inst_chains;  // per chain record: its objects, outermost first
    bool flattenable(rtx_ref_t r, int levels, int guard = 0) {
        if (guard > 4096 || !check_ref(r)) return false;
        const uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        if (k == RTX_REF_NODE) return flattenable(d.bvh_nodes[i].left, levels, guard + 1) && flattenable(d.bvh_nodes[i].right, levels, guard + 1);
        if (k == RTX_REF_SPHERE) return d.sph_dx[i] == 0.0f && d.sph_dy[i] == 0.0f && d.sph_dz[i] == 0.0f;
        if (is_prim(r)) return true;
        const rtx_object_t& o = d.objects[i];
        if (o.kind == RTX_OBJ_LIST) {
            if ((uint64_t)o.first + o.count > d.n_refs) return false;
            for (uint32_t c = 0; c < o.count; c++)
                if (!flattenable(d.refs[o.first + c], levels, guard + 1)) return false;
            return true;
        }
        if (o.kind == RTX_OBJ_TRANSLATE || o.kind == RTX_OBJ_ROTATE_Y) return levels < MAX_INST_LEVELS && flattenable(o.child, levels + 1, guard + 1);
        return false;
    }
    uint32_t chain_id(const std::vector<uint32_t>& objs) {
        for (uint32_t c = 0; c < inst_chains.size(); c++)
            if (inst_chains[c] == objs) return c;
        inst_chains.push_back(objs);
        return (uint32_t)inst_chains.size() - 1u;
    }

    // Local-space bounds of everything under a reference (tight; padding is applied per unit).
    bool bounds_of(rtx_ref_t r, Box3& out_box, int guard = 0) {
        if (!check_ref(r)) return false;
        if (guard > 4096) return fail("scene graph too deep (cycle?)");
        uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        out_box.reset();
        if (k == RTX_REF_NODE) {
            Box3 a, b;
            if (!bounds_of(d.bvh_nodes[i].left, a, guard + 1) || !bounds_of(d.bvh_nodes[i].right, b, guard + 1)) return false;
            out_box.grow(a);
            out_box.grow(b);
        } else if (is_prim(r)) {
            out_box = prim_box(d, k, i);
            if (!out_box.valid()) return fail("NaN in primitive bounds");
        } else {
            const rtx_object_t& o = d.objects[i];
            if (o.kind == RTX_OBJ_LIST) {
                if ((uint64_t)o.first + o.count > d.n_refs) return fail("list object exceeds refs");
                for (uint32_t c = 0; c < o.count; c++) {
                    Box3 b;
                    if (!bounds_of(d.refs[o.first + c], b, guard + 1)) return false;
                    out_box.grow(b);
                }
            } else {
                Box3 c;
                if (!bounds_of(o.child, c, guard + 1)) return false;
                coord_range = std::fmax(coord_range, 2.0f * c.max_abs());
                if (o.kind == RTX_OBJ_TRANSLATE || o.kind == RTX_OBJ_ROTATE_Y) {
                    out_box = xf_box(o, c);
                } else if (o.kind == RTX_OBJ_MEDIUM) {
                    out_box = c;
                } else {
                    return fail("unknown object kind");
                }
                if (!out_box.valid()) return fail("NaN in object bounds");
            }
        }
        return true;
    }

    // Units under a reference: BVH nodes and lists are dissolved, primitives and bracketed objects are kept whole.
    // may_place: instances met here may have their primitives placed in this frame's tree (top-level frame only: the
    // chain then starts from the caller's ray); chain: the instances already entered on the way, outermost first.
    bool collect_units(rtx_ref_t r, std::vector<Unit>& units, bool may_place, int guard = 0, std::vector<uint32_t>* chain = nullptr) {
        if (!check_ref(r)) return false;
        if (guard > 4096) return fail("scene graph too deep (cycle?)");
        uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        if (k == RTX_REF_NODE) {
            const rtx_bvh_node_t& n = d.bvh_nodes[i];
            if (!collect_units(n.left, units, may_place, guard + 1, chain)) return false;
            return n.right == n.left || collect_units(n.right, units, may_place, guard + 1, chain);
        }
        if (k == RTX_REF_OBJECT && d.objects[i].kind == RTX_OBJ_LIST) {
            const rtx_object_t& o = d.objects[i];
            if ((uint64_t)o.first + o.count > d.n_refs) return fail("list object exceeds refs");
            for (uint32_t c = 0; c < o.count; c++)
                if (!collect_units(d.refs[o.first + c], units, may_place, guard + 1, chain)) return false;
            return true;
        }
        if (k == RTX_REF_OBJECT && (d.objects[i].kind == RTX_OBJ_TRANSLATE || d.objects[i].kind == RTX_OBJ_ROTATE_Y) && may_place && place_instances &&
            flattenable(r, chain ? (int)chain->size() : 0)) {
            std::vector<uint32_t> inner = chain ? *chain : std::vector<uint32_t>();
            inner.push_back(i);
            Box3 c;
            if (!bounds_of(r, c)) return false;  // (validates the subtree and feeds coord_range as for a bracketed instance)
            return collect_units(d.objects[i].child, units, may_place, guard + 1, &inner);
        }
        Unit u;
        u.ref = r;
        auto it = first_rank.find(r);
        u.rank = it == first_rank.end() ? 0u : it->second;
        if (!bounds_of(r, u.box)) return false;
        if (chain != nullptr && !chain->empty()) {
            for (size_t l = chain->size(); l-- > 0;) u.box = xf_box(d.objects[(*chain)[l]], u.box);  // innermost level first
            if (!u.box.valid()) return fail("NaN in object bounds");
            u.chain = chain_id(*chain);
        }
        units.push_back(u);
        return true;
    }

    static float half_area(const Box3& b) {
        float e[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }

    bool emit_fast_subtree(Threaded& out, rtx_ref_t r, uint32_t enclosing_xf, int xf_depth, int med_depth) {
        std::vector<Unit> units;
        if (!collect_units(r, units, enclosing_xf == NONE)) return false;
        if (units.empty()) return fail("empty subtree");
        // drop duplicates (a one-element Bvh lists its child twice, shared objects may repeat): keep the first by rank
        std::sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) {
            return a.ref != b.ref ? a.ref < b.ref : a.chain != b.chain ? a.chain < b.chain : a.rank < b.rank;
        });
        units.erase(std::unique(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.ref == b.ref && a.chain == b.chain; }), units.end());
        std::sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.rank < b.rank; });
        for (Unit& u : units) {
            coord_range_seen = std::fmax(coord_range_seen, u.box.max_abs());
            pad_box(u.box, coord_range);
        }
        // The whole boundary of a medium is one primitive: the box it would get is the box the medium itself stands under
        // (bounds_of: a medium's box is its boundary's), which a walk has passed when it arrives here.
        if (units.size() == 1 && is_prim(units[0].ref) && units[0].chain == NONE && med_depth > 0 && !out.ents.empty() && out.ents.back().w[7] != ENT_BOX &&
            (out.ents.back().w[7] & 15u) == ENT_MED_BEGIN)
            return emit_unit_fast(out, units[0], false, enclosing_xf, xf_depth, med_depth);
        // ... and for a boundary of several units the same holds for the box around all of them: its two halves stand
        // between the medium's begin and end without it
        skip_box_once = units.size() > 1 && med_depth > 0 && !out.ents.empty() && out.ents.back().w[7] != ENT_BOX && (out.ents.back().w[7] & 15u) == ENT_MED_BEGIN;
        // Units whose box no ray of the scene can miss (never_culls: a ground sphere whose padded box reaches over
        // everything, a medium that fills the scene) are kept out of the tree: inside it they would stretch every box on
        // the way down to them to their own size, and none of those boxes could cull either.  They stand right behind the
        // root, without boxes of their own, and the tree of the others follows as the root's last child.
        const uint32_t n_units = (uint32_t)units.size();
        if (drop_unculling && enclosing_xf == NONE && med_depth == 0 && out.ents.empty() && n_units > 1) {
            const auto rest = std::stable_partition(units.begin(), units.end(), [&](const Unit& u) { return u.chain == NONE && never_culls(u.box); });
            const uint32_t n_fill = (uint32_t)(rest - units.begin());
            if (n_fill > 0 && n_fill < n_units) {
                Box3 bb;
                bb.reset();
                for (const Unit& u : units) bb.grow(u.box);
                const float b6[6] = {bb.lo[0], bb.hi[0], bb.lo[1], bb.hi[1], bb.lo[2], bb.hi[2]};
                out.ents.push_back(box_entry(b6, 0));
                out.rank.push_back(0);
                out.mat.push_back(NONE);
                out.n_boxes++;
                out.has_bare = true;
                bool ok = true;
                for (uint32_t i = 0; i < n_fill && ok; i++) ok = emit_unit_fast(out, units[i], false, enclosing_xf, xf_depth, med_depth);
                ok = ok && sah(out, units, n_fill, n_units, enclosing_xf, xf_depth, med_depth);
                out.ents[0].w[6] = (uint32_t)out.ents.size();
                return ok;
            }
            std::sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.rank < b.rank; });
        }
        return sah(out, units, 0, n_units, enclosing_xf, xf_depth, med_depth);
    }
    float coord_range_seen = 0;

    bool emit_unit_fast(Threaded& out, const Unit& u, bool guard_box, uint32_t enclosing_xf, int xf_depth, int med_depth) {
        if (is_prim(u.ref)) {
            push_prim(out, u.ref, u.rank);
            if (u.chain != NONE) {
                Ent& e = out.ents.back();
                e.w[7] |= ENT_INST;
                e.w[RTX_REF_KIND(u.ref) == RTX_REF_SPHERE ? 3 : 1] = u.chain;
                out.has_brackets = true;  // rendered by the variants that know transforms
            }
            return true;
        }
        const rtx_object_t& o = d.objects[RTX_REF_INDEX(u.ref)];
        uint32_t at = NONE;
        // a bracketed object is a whole subtree: never enter it unless its own box is hit -- unless no ray can miss that
        // box (never_culls: a medium that fills the scene)
        if (guard_box && !(drop_unculling && enclosing_xf == NONE && never_culls(u.box))) {
            at = (uint32_t)out.ents.size();
            const float b6[6] = {u.box.lo[0], u.box.hi[0], u.box.lo[1], u.box.hi[1], u.box.lo[2], u.box.hi[2]};
            out.ents.push_back(box_entry(b6, 0));
            out.rank.push_back(0);
            out.mat.push_back(NONE);
            out.n_boxes++;
        }
        bool ok = emit_bracketed(out, o, RTX_REF_INDEX(u.ref), u.rank, enclosing_xf, xf_depth, med_depth,
                                 [&](rtx_ref_t c, uint32_t xf, int xd, int med) { return emit_fast_subtree(out, c, xf, xd, med); });
        if (at != NONE) out.ents[at].w[6] = (uint32_t)out.ents.size();
        return ok;
    }

    static constexpr int NBINS = 16;
    bool bare_leaves = false;  // experiment (RTX_SAH_BARE_LEAF): a leaf of one primitive gets no box of its own
    // A box that contains every surface of the scene contains the origin of every bounce ray, and such a ray cannot miss
    // it: its test is a step that never culls.  (The ground sphere of radius 1000 under a scene two units high: its padded
    // box reaches above everything that stands on it.)  A single primitive under such a box stands in its parent's
    // sequence without it (also the boundary of a medium that fills the scene: its two queries start inside it like any
    // other ray).  Top-level frame only, never the root (whose test the render skips by itself when the camera
    // is inside).  The wide form takes the primitives that stand right behind the root as a list of its own (Wide::n_first).
    bool drop_unculling = false;
    bool skip_box_once = false;  // the next call of sah() emits its children without the box around them (see emit_fast_subtree)
    Box3 scene_tight;          // tight bounds of everything (world frame)
    bool never_culls(const Box3& b) const {
        for (int a = 0; a < 3; a++)
            if (!(b.lo[a] < scene_tight.lo[a] && b.hi[a] > scene_tight.hi[a])) return false;
        return true;
    }
    bool sah(Threaded& out, std::vector<Unit>& units, uint32_t begin, uint32_t end, uint32_t enclosing_xf, int xf_depth, int med_depth) {
        const float INF = std::numeric_limits<float>::infinity();
        if (bare_leaves && end - begin == 1 && is_prim(units[begin].ref) && !out.ents.empty())
            return emit_unit_fast(out, units[begin], false, enclosing_xf, xf_depth, med_depth);
        Box3 bb, cb;
        bb.reset();
        cb.reset();
        for (uint32_t i = begin; i < end; i++) {
            bb.grow(units[i].box);
            for (int a = 0; a < 3; a++) {
                float c = 0.5f * (units[i].box.lo[a] + units[i].box.hi[a]);
                cb.lo[a] = std::fmin(cb.lo[a], c);
                cb.hi[a] = std::fmax(cb.hi[a], c);
            }
        }
        const uint32_t n = end - begin;
        if (drop_unculling && n == 1 && is_prim(units[begin].ref) && units[begin].chain == NONE && enclosing_xf == NONE && !out.ents.empty() && never_culls(bb)) {
            out.has_bare = true;
            return emit_unit_fast(out, units[begin], false, enclosing_xf, xf_depth, med_depth);
        }
        const bool no_box = skip_box_once && n > 1;
        skip_box_once = false;
        const uint32_t at = (uint32_t)out.ents.size();
        if (!no_box) {
            const float b6[6] = {bb.lo[0], bb.hi[0], bb.lo[1], bb.hi[1], bb.lo[2], bb.hi[2]};
            out.ents.push_back(box_entry(b6, 0));
            out.rank.push_back(0);
            out.mat.push_back(NONE);
            out.n_boxes++;
        }

        int best_axis = -1, best_bin = -1;
        float best_cost = INF;
        const float parent_area = half_area(bb);
        if (n > 1) {
            for (int a = 0; a < 3; a++) {
                float ext = cb.hi[a] - cb.lo[a];
                if (!(ext > 0) || !std::isfinite(ext)) continue;
                uint32_t cnt[NBINS] = {0};
                Box3 bins[NBINS];
                for (int b = 0; b < NBINS; b++) bins[b].reset();
                float scale = (float)NBINS / ext;
                for (uint32_t i = begin; i < end; i++) {
                    float c = 0.5f * (units[i].box.lo[a] + units[i].box.hi[a]);
                    int b = std::min(NBINS - 1, std::max(0, (int)((c - cb.lo[a]) * scale)));
                    cnt[b]++;
                    bins[b].grow(units[i].box);
                }
                float rarea[NBINS];
                uint32_t rcnt[NBINS];
                Box3 acc;
                acc.reset();
                uint32_t rc = 0;
                for (int b = NBINS - 1; b > 0; b--) {
                    if (cnt[b]) acc.grow(bins[b]);
                    rc += cnt[b];
                    rarea[b] = rc ? half_area(acc) : 0;
                    rcnt[b] = rc;
                }
                acc.reset();
                uint32_t lc = 0;
                for (int b = 0; b < NBINS - 1; b++) {
                    if (cnt[b]) acc.grow(bins[b]);
                    lc += cnt[b];
                    if (lc == 0 || rcnt[b + 1] == 0) continue;
                    float cost = half_area(acc) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        // SAH: compare (C_TRAV + split) against the cost of testing everything here.  A sphere test costs about what a box
        // test costs (weight 1); a quad or triangle test costs about five box steps (PLANAR_COST), so leaves of several of
        // them rarely pay: cornell-box spent 5.6 primitive tests per ray on leaves of four quads, 1.8 with leaves of one (+29 %).
        float weight = 0.0f;
        for (uint32_t i = begin; i < end; i++) weight += is_prim(units[i].ref) && RTX_REF_KIND(units[i].ref) != RTX_REF_SPHERE ? PLANAR_COST : 1.0f;
        bool make_leaf = n <= 1 || best_axis < 0 ||
                         (n <= MAX_LEAF && !(parent_area > 0 && C_TRAV + (best_cost / parent_area) * (weight / (float)n) < weight));
        bool ok = true;
        if (make_leaf && n > MAX_LEAF) {
            uint32_t mid = begin + n / 2;  // all centroids coincide: halve the range to bound the leaf size
            ok = sah(out, units, begin, mid, enclosing_xf, xf_depth, med_depth) && sah(out, units, mid, end, enclosing_xf, xf_depth, med_depth);
        } else if (make_leaf) {
            std::sort(units.begin() + begin, units.begin() + end, [](const Unit& x, const Unit& y) { return x.rank < y.rank; });
            for (uint32_t i = begin; i < end && ok; i++) ok = emit_unit_fast(out, units[i], n > 1, enclosing_xf, xf_depth, med_depth);
        } else {
            const int a = best_axis;
            const float base = cb.lo[a], scale = (float)NBINS / (cb.hi[a] - cb.lo[a]);
            auto midit = std::partition(units.begin() + begin, units.begin() + end, [&](const Unit& p) {
                float c = 0.5f * (p.box.lo[a] + p.box.hi[a]);
                int b = std::min(NBINS - 1, std::max(0, (int)((c - base) * scale)));
                return b <= best_bin;
            });
            uint32_t mid = (uint32_t)(midit - units.begin());
            if (mid == begin || mid == end) mid = begin + n / 2;
            ok = sah(out, units, begin, mid, enclosing_xf, xf_depth, med_depth);
            const uint32_t right_at = (uint32_t)out.ents.size();
            ok = ok && sah(out, units, mid, end, enclosing_xf, xf_depth, med_depth);
            // (children that may be visited in either order: both must be boxes, whose links say where they end)
            if (!no_box && ok && out.ents[at + 1].w[7] == ENT_BOX && out.ents[at + 1].w[6] == right_at && right_at < out.ents.size() && out.ents[right_at].w[7] == ENT_BOX)
                out.inner.push_back(at);
        }
        if (!no_box) out.ents[at].w[6] = (uint32_t)out.ents.size();
        return ok;
    }
};

// A copy of a fast array (entry indices, before finalize) in which the two children of every inner box are visited
// nearer-to-`view` first.  Rays that start at the viewpoint then meet their closest hit early and cull the far child
// by distance; the closest hit itself does not depend on the order.  Inside a Translate / RotateY the viewpoint is
// moved into the object's space.
class Reorder {
  public:
    Reorder(const Threaded& src_, Threaded& dst_) : src(src_), dst(dst_) {}

    // grow: every box is enlarged by this much on all sides (a viewpoint outside the range the boxes were padded for)
    void run(const float view[3], float grow = 0.0f, bool reorder = true) {
        grow_ = grow;
        reorder_ = reorder;  // false: keep the build order, only enlarge the boxes
        const uint32_t n = (uint32_t)src.ents.size();
        is_inner.assign(n, 0);
        for (uint32_t i : src.inner) is_inner[i] = 1;
        new_index.assign(n, NONE);
        old_index.clear();
        old_index.reserve(n);
        dst.ents.clear();
        dst.rank.clear();
        dst.mat.clear();
        dst.ents.reserve(n + 1);
        dst.rank.reserve(n + 1);
        dst.mat.reserve(n + 1);
        dst.n_boxes = src.n_boxes;
        dst.n_prims = src.n_prims;
        dst.has_brackets = src.has_brackets;
        dst.has_frames = src.has_frames;
        dst.has_bare = src.has_bare;
        dst.sphere_media = src.sphere_media;
        dst.chains = src.chains;
        dst.inner.clear();
        for (uint32_t j = 0; j < n;) j = place(j, view);
        // links that name an entry (not the end of the own subtree) go through the index map
        for (uint32_t p = 0; p < n; p++) {
            Ent& e = dst.ents[p];
            if (e.w[7] == ENT_BOX) continue;
            const uint32_t kind = e.w[7] & 15u;
            if (kind == ENT_XF_PUSH || kind == ENT_MED_BEGIN || kind == ENT_MED_END) e.w[6] = new_index[e.w[6]];
            if ((kind == ENT_XF_PUSH || kind == ENT_XF_POP) && e.w[4] != NONE) e.w[4] = new_index[e.w[4]];
        }
    }

  private:
    const Threaded& src;
    Threaded& dst;
    std::vector<uint8_t> is_inner;
    std::vector<uint32_t> new_index, old_index;
    float grow_ = 0.0f;
    bool reorder_ = true;

    static float bits_f(uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    float dist2(uint32_t box, const float view[3]) const {  // squared distance from the viewpoint to a box entry (0 inside)
        const Ent& e = src.ents[box];
        float s = 0;
        for (int a = 0; a < 3; a++) {
            float g = std::fmax(std::fmax(bits_f(e.w[2 * a]) - view[a], view[a] - bits_f(e.w[2 * a + 1])), 0.0f);
            s += g * g;
        }
        return s;
    }
    uint32_t copy(uint32_t i) {
        const uint32_t p = (uint32_t)dst.ents.size();
        new_index[i] = p;
        dst.ents.push_back(src.ents[i]);
        dst.rank.push_back(src.rank[i]);
        dst.mat.push_back(src.mat[i]);
        if (grow_ > 0.0f && src.ents[i].w[7] == ENT_BOX) {
            Ent& e = dst.ents.back();
            for (int a = 0; a < 3; a++) {
                e.w[2 * a] = fbits(bits_f(e.w[2 * a]) - grow_);
                e.w[2 * a + 1] = fbits(bits_f(e.w[2 * a + 1]) + grow_);
            }
        }
        return p;
    }
    // Places entry i with everything under it; returns the old index of the entry that followed it.
    uint32_t place(uint32_t i, const float view[3]) {
        const Ent& e = src.ents[i];
        const uint32_t p = copy(i);
        if (e.w[7] == ENT_BOX) {
            const uint32_t end = e.w[6];
            if (is_inner[i]) {
                const uint32_t l = i + 1, r = src.ents[l].w[6];
                dst.inner.push_back(p);
                if (reorder_ && dist2(r, view) < dist2(l, view)) {
                    place(r, view);
                    place(l, view);
                } else {
                    place(l, view);
                    place(r, view);
                }
            } else {
                for (uint32_t j = i + 1; j < end;) j = place(j, view);
            }
            dst.ents[p].w[6] = (uint32_t)dst.ents.size();
            return end;
        }
        const uint32_t kind = e.w[7] & 15u;
        if (kind == ENT_XF_PUSH || kind == ENT_MED_BEGIN) {
            const uint32_t close = e.w[6];
            float local[3] = {view[0], view[1], view[2]};
            if (kind == ENT_XF_PUSH) {
                if (e.w[0] == 1u) {
                    for (int a = 0; a < 3; a++) local[a] = view[a] - bits_f(e.w[1 + a]);
                } else {
                    const float sn = bits_f(e.w[1]), cs = bits_f(e.w[2]);
                    local[0] = cs * view[0] - sn * view[2];
                    local[2] = sn * view[0] + cs * view[2];
                }
            }
            for (uint32_t j = i + 1; j < close;) j = place(j, local);
            copy(close);
            return close + 1;
        }
        return i + 1;
    }
};

// ---------------------------------------------------------------------------------------------- packed fast array
// The form of the fast array the render kernels walk when the scene has no transform / medium brackets: 16-byte slots,
//   box        half_boxes: 1 slot   6 halfs (centre x y z, half extent x y z), word 3 = first slot of the entry to jump
//                                   to on a miss
//              else:       2 slots  (centre.xyz as floats, miss link as a BYTE offset) (half extent.xyz as floats, 0)
//   sphere     2 slots  (c0.xyz, tag) (dc.xyz, r)
//   quad/tri   5 slots  (n.xyz, tag) (Q.xyz, d) (u.xyz, material) (v.xyz, prim id) (w.xyz, 0)
//   end        1 slot   word 3 = ENT_END
// tag = kind << 28 | index (kind >= 1), so a word 3 below PK_PRIM says "box" and is its miss link at the same time.
// Centre / half-extent form needs no min / max per axis: the near and far plane distances are
// centre_distance -+ half_extent * |1 / d|, three fused multiply-adds, which gfx950 issues at nearly twice the rate of
// min / max / compare.  Arrays small enough for LDS keep float boxes (the walk is bound by vector issue there, and a
// multiply-add that reads a half in place issues at the slow rate); larger arrays are walked from HBM / L2, where bytes
// are what counts, and get half-precision boxes: one 16-byte read per box.  Rounding only loosens a box: the centre is
// rounded to nearest and the half extent is rounded UP from what is needed to cover the given box around the rounded
// centre; where a value does not fit a half (beyond 65504) the axis gets an infinite extent around 0 and never culls.
constexpr uint32_t PK_PRIM = 0x10000000u;
constexpr uint32_t PK_PAD_FLAG = 0x08000000u;   // in a sphere's tag: a pad slot follows the entry
constexpr uint32_t PK_INDEX_MASK = 0x07FFFFFFu;  // the index part of a tag
constexpr uint32_t PK_SLOT_BYTES = 16u;

static inline uint16_t f16_nearest(float f) {  // IEEE binary16, round to nearest even; overflow -> inf, NaN -> NaN
    uint32_t x = fbits(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to or beyond 65520 -> inf
    if (x < 0x33000001u) return sign;                          // <= 2^-25 -> 0 (2^-25 itself ties to even = 0)
    if (x < 0x38800000u) {                                     // result is a half denormal
        const int shift = 126 - (int)(x >> 23);                // 14 .. 24
        const uint32_t mant = (x & 0x7FFFFFu) | 0x800000u;
        const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t rem = x & 0x1FFFu;
    uint32_t h = ((x - 0x38000000u) >> 13);
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
static inline float f16_value(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            float v = (float)m * (1.0f / 16777216.0f);  // m * 2^-24, exact
            uint32_t b;
            std::memcpy(&b, &v, 4);
            x = sign | b;
        }
    } else if (e == 31) {
        x = sign | 0x7F800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}
static inline uint16_t f16_at_least(double v) {  // smallest non-negative half >= v (v >= 0)
    uint16_t h = f16_nearest((float)v);
    if (h >= 0x7C00u) return 0x7C00u;
    if ((double)f16_value(h) < v) h++;  // next half up; 0x7BFF + 1 is +inf
    return h;
}

struct Packed {
    std::vector<uint32_t> words;  // 4 per slot
    std::vector<uint32_t> rank;   // per slot: tie-break rank of the primitive whose entry starts there
    std::vector<uint32_t> mat;    // per slot: material of that primitive, NONE elsewhere
    size_t n_slots() const { return words.size() / 4; }
};

// planar: 5 x 4 floats per quad / triangle as the device keeps them ((Q,d) (u,mat) (v,0) (n,0) (w,0)), quads first, and
// planar_id: their prim ids in the same order.  `t` is a fast array in entry-index form (before finalize).
// Returns false when it holds brackets (or is too large for 28-bit slot links).
// Arrays with float boxes (the ones walked from LDS): every entry is 32 bytes or 80, so without further ado the first
// slot of every entry stands at an even multiple of 16 bytes and the second at an odd one -- and a 16-byte LDS read of one
// wave instruction, which goes for the same slot of every lane's entry, uses only every other group of four banks.  A
// one-slot *pad* behind every pk_pad_every()-th sphere shifts what follows by 16 bytes, so that entries of both
// parities exist and a read spreads over all banks.  Nobody reads a pad: links lead to the entry behind it, and the
// sphere in front of it says in its tag (PK_PAD_FLAG) that it is three slots long.  RTX_PK_PAD=0: no pads.
static inline uint32_t pk_pad_every() {
    static const uint32_t v = [] {
        const char* e = std::getenv("RTX_PK_PAD");  // (measured on the benchmark scene: 2 to 32 all gain, flat between 8 and 16)
        return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 8u;
    }();
    return v;
}
static inline size_t packed_slots(const Threaded& t, bool half_boxes) {  // terminator included
    size_t slots = 1, spheres = 0;
    const uint32_t pad = half_boxes ? 0u : pk_pad_every();
    for (const Ent& e : t.ents) {
        const uint32_t kind = e.w[7] == ENT_BOX ? ENT_BOX : (e.w[7] & 15u);
        slots += kind == ENT_BOX ? (half_boxes ? 1 : 2) : kind == ENT_SPHERE ? 2 : 5;
        if (kind == ENT_SPHERE && pad != 0u && ++spheres % pad == 0) slots++;
    }
    return slots;
}

static inline bool pack(const Threaded& t, const float* planar, const uint32_t* planar_id, uint32_t n_quads, bool half_boxes, Packed& out) {
    if (t.has_brackets) return false;
    const size_t n = t.ents.size();
    std::vector<uint32_t> first(n + 1);
    std::vector<uint32_t> pads;  // slot indices of the pads (see pk_pad_every)
    const uint32_t pad_every = half_boxes ? 0u : pk_pad_every();
    size_t slots = 0, spheres = 0;
    for (size_t i = 0; i < n; i++) {
        first[i] = (uint32_t)slots;
        const uint32_t w7 = t.ents[i].w[7];
        const uint32_t kind = w7 == ENT_BOX ? ENT_BOX : (w7 & 15u);
        slots += kind == ENT_BOX ? (half_boxes ? 1 : 2) : kind == ENT_SPHERE ? 2 : 5;
        if (kind == ENT_SPHERE && pad_every != 0u && ++spheres % pad_every == 0) pads.push_back((uint32_t)slots++);
    }
    first[n] = (uint32_t)slots;
    slots += 1;  // terminator
    if (slots >= PK_PRIM) return false;
    out.words.assign(4 * slots, 0u);
    out.rank.assign(slots, 0u);
    out.mat.assign(slots, NONE);
    auto fval = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    for (size_t i = 0; i < n; i++) {
        const Ent& e = t.ents[i];
        uint32_t* w = &out.words[4 * (size_t)first[i]];
        const uint32_t w7 = e.w[7];
        if (w7 == ENT_BOX && !half_boxes) {
            for (int a = 0; a < 3; a++) {
                const float lo = fval(e.w[2 * a]), hi = fval(e.w[2 * a + 1]);
                float c = 0.5f * lo + 0.5f * hi;
                if (!std::isfinite(c)) c = 0.0f;
                const double need = std::fmax((double)hi - (double)c, (double)c - (double)lo);
                float h = (float)need;
                if ((double)h < need) h = std::nextafter(h, std::numeric_limits<float>::infinity());
                if (!(need == need)) h = std::numeric_limits<float>::infinity();  // NaN bounds: never cull
                w[a] = fbits(c);
                w[4 + a] = fbits(h);
            }
            w[3] = first[e.w[6]] * PK_SLOT_BYTES;  // these arrays live in LDS: the link is the LDS address to continue at
            w[7] = 0;
            continue;
        }
        if (w7 == ENT_BOX) {
            uint16_t c[3], h[3];
            for (int a = 0; a < 3; a++) {
                const float lo = fval(e.w[2 * a]), hi = fval(e.w[2 * a + 1]);
                c[a] = f16_nearest(0.5f * lo + 0.5f * hi);
                const double cv = (double)f16_value(c[a]);
                const double need = std::fmax((double)hi - cv, cv - (double)lo);
                h[a] = f16_at_least(std::fmax(need, 0.0));
                if ((c[a] & 0x7FFFu) >= 0x7C00u || !(need == need)) {  // centre does not fit a half (or NaN bounds): never cull on this axis
                    c[a] = 0;
                    h[a] = 0x7C00u;
                }
            }
            w[0] = (uint32_t)c[0] | ((uint32_t)c[1] << 16);
            w[1] = (uint32_t)c[2] | ((uint32_t)h[0] << 16);
            w[2] = (uint32_t)h[1] | ((uint32_t)h[2] << 16);
            w[3] = first[e.w[6]];  // e.w[6] <= n: the entry after the subtree, or the terminator
            continue;
        }
        const uint32_t kind = w7 & 15u, index = w7 >> 4;
        if (index > PK_INDEX_MASK) return false;
        out.rank[first[i]] = t.rank[i];
        out.mat[first[i]] = t.mat[i];
        if (kind == ENT_SPHERE) {
            w[0] = e.w[0]; w[1] = e.w[1]; w[2] = e.w[2]; w[3] = (ENT_SPHERE << 28) | index;
            w[4] = e.w[3]; w[5] = e.w[4]; w[6] = e.w[5]; w[7] = e.w[6];
        } else {
            const size_t p = kind == ENT_TRI ? (size_t)index + n_quads : index;
            const float* r = planar + 20 * p;  // (Q,d) (u,mat) (v,0) (n,0) (w,0)
            auto put3 = [&](uint32_t* dst, const float* src) { std::memcpy(dst, src, 12); };
            put3(w + 0, r + 12);  w[3] = (kind << 28) | index;                     // n, tag
            put3(w + 4, r + 0);   std::memcpy(w + 7, r + 3, 4);                    // Q, d
            put3(w + 8, r + 4);   std::memcpy(w + 11, r + 7, 4);                   // u, material
            put3(w + 12, r + 8);  w[15] = planar_id[p];             // v, prim id
            put3(w + 16, r + 16); w[19] = 0;                                       // w
        }
    }
    out.words[4 * (size_t)first[n] + 3] = ENT_END;
    for (uint32_t p : pads) out.words[4 * (size_t)(p - 2u) + 3] |= PK_PAD_FLAG;  // (the sphere in front of the pad)
    return true;
}

// ---------------------------------------------------------------------------------------------- wide fast array
// What the render kernels walk when a bracket-free scene is too large for LDS: the binary fast tree collapsed into nodes
// of up to four children, each node 64 bytes (half a cache line), and the primitives in a separate array of five-slot
// records.  Such a walk is bound by the bytes its lanes pull through the vector memory path, one dependent fetch after
// the other: a node brings in two binary levels with one 64-byte fetch (the binary form: 16 bytes per box, but one fetch
// per box), and the primitives of every leaf a node hands out are known at once.  Walking needs a per-lane stack (kept
// in LDS); its depth is bounded by `levels`.
//   node, 4 slots of 16 bytes:
//     (origin.xyz, child_base)   (scale.xyz, prim_base)   (lo_x, lo_y, lo_z, leaves)   (hi_x, hi_y, hi_z, n_internal | n_children << 8)
//   origin = the low corner of the box around all children, scale = its extent / 255 (rounded up).  lo_a / hi_a hold one
//   byte per child (child j in bits 8j .. 8j+7): the child's box on axis a is [origin_a + scale_a * lo, origin_a +
//   scale_a * hi], rounded outward from the fast tree's (padded) box, so a quantised box always contains the box it
//   stands for.  An unused child has lo = 255, hi = 0 on every axis (never hit).
//   Children that are nodes come first: child j < n_internal is node child_base + j (children of one node are
//   contiguous).  A child that is a leaf owns some of the node's primitives, which follow each other from slot prim_base
//   in child order: byte j of `leaves` = count << 5 | index of its first primitive record among the node's (0 for a child
//   node and for an unused child).
//   primitive, 5 slots: a sphere (c0.xyz, tag) (dc.xyz, r) + 3 unused slots; a quad / triangle as in the packed array.
constexpr uint32_t WIDE_NODE_WORDS = 16;
constexpr uint32_t WIDE_PRIM_SLOTS = 5;
constexpr uint32_t WIDE_MAX_LEVELS = 32;  // deeper trees are not collapsed (the scene then keeps the packed array)
constexpr uint32_t WIDE_MAX_LEAF = 7;     // primitives per leaf: three bits count them, five bits say where they start

struct Wide {
    std::vector<uint32_t> nodes;  // WIDE_NODE_WORDS per node, root first
    std::vector<uint32_t> prims;  // 4 words per slot, WIDE_PRIM_SLOTS slots per primitive
    std::vector<uint32_t> rank;   // per slot: tie-break rank of the primitive whose record starts there
    std::vector<uint32_t> mat;    // per slot: its material, NONE elsewhere
    uint32_t levels = 0;          // nodes on the longest path from the root
    uint32_t n_first = 0;         // primitives every walk tests before it visits the root node: the first n_first records of
                                  // `prims` (units no ray can miss, which the builder keeps out of the tree)
    size_t n_nodes() const { return nodes.size() / WIDE_NODE_WORDS; }
    size_t n_slots() const { return prims.size() / 4; }
};

class WideBuilder {
  public:
    WideBuilder(const Threaded& t_, const float* planar_, const uint32_t* planar_id_, uint32_t n_quads_, float grow_, Wide& out_)
        : t(t_), planar(planar_), planar_id(planar_id_), n_quads(n_quads_), grow(grow_), out(out_) {}

    bool run() {
        if (t.has_brackets || t.ents.empty() || t.ents[0].w[7] != ENT_BOX) return false;
        out = Wide();
        out.nodes.assign(WIDE_NODE_WORDS, 0u);
        // Primitives that stand right behind the root without boxes (Builder: units no ray can miss) are not part of the
        // tree: every walk tests them first, and the root node is the root of what follows them.
        uint32_t top = 0;
        const uint32_t n = (uint32_t)t.ents.size();
        if (n > 2 && t.ents[1].w[7] != ENT_BOX) {
            uint32_t p = 1;
            while (p < n && t.ents[p].w[7] != ENT_BOX && (t.ents[p].w[7] & 15u) <= ENT_TRI) p++;
            if (p - 1 > 16u || p >= n || t.ents[p].w[7] != ENT_BOX || t.ents[p].w[6] != t.ents[0].w[6]) return false;
            for (uint32_t q = 1; q < p; q++) put_prim(q);
            out.n_first = p - 1;
            top = p;
        }
        if (!node(0, top, 1)) return false;
        return out.n_nodes() < (1u << 23) && out.n_slots() < (1u << 28);
    }

  private:
    const Threaded& t;
    const float* planar;
    const uint32_t* planar_id;
    uint32_t n_quads;
    float grow;
    Wide& out;

    static float fval(uint32_t u) {
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    bool is_box(uint32_t e) const { return e < t.ents.size() && t.ents[e].w[7] == ENT_BOX; }
    bool internal(uint32_t e) const { return is_box(e) && is_box(e + 1); }  // a box whose first child is a box has two box children
    float area(uint32_t e) const {
        const Ent& b = t.ents[e];
        const float x = fval(b.w[1]) - fval(b.w[0]), y = fval(b.w[3]) - fval(b.w[2]), z = fval(b.w[5]) - fval(b.w[4]);
        const float a = x * y + y * z + z * x;
        return a == a ? a : std::numeric_limits<float>::infinity();
    }
    bool box_of(uint32_t e, double lo[3], double hi[3]) const {  // box entry e, enlarged by `grow`; false when it is not finite
        for (int a = 0; a < 3; a++) {
            lo[a] = (double)(fval(t.ents[e].w[2 * a]) - grow);
            hi[a] = (double)(fval(t.ents[e].w[2 * a + 1]) + grow);
            if (!std::isfinite(lo[a]) || !std::isfinite(hi[a]) || std::fabs(lo[a]) > 1e30 || std::fabs(hi[a]) > 1e30) return false;
        }
        return true;
    }
    float extent(uint32_t e) const {  // largest side of box entry e
        const Ent& b = t.ents[e];
        const float m = std::fmax(fval(b.w[1]) - fval(b.w[0]), std::fmax(fval(b.w[3]) - fval(b.w[2]), fval(b.w[5]) - fval(b.w[4])));
        return m == m ? m : std::numeric_limits<float>::infinity();
    }
    void put_prim(uint32_t p) {  // appends the five-slot record of primitive entry p
        const Ent& e = t.ents[p];
        const uint32_t kind = e.w[7] & 15u, index = e.w[7] >> 4;
        const size_t at = out.prims.size();
        out.prims.resize(at + 4 * WIDE_PRIM_SLOTS, 0u);
        out.rank.resize(out.rank.size() + WIDE_PRIM_SLOTS, 0u);
        out.mat.resize(out.mat.size() + WIDE_PRIM_SLOTS, NONE);
        uint32_t* w = &out.prims[at];
        out.rank[at / 4] = t.rank[p];
        out.mat[at / 4] = t.mat[p];
        if (kind == ENT_SPHERE) {
            w[0] = e.w[0]; w[1] = e.w[1]; w[2] = e.w[2]; w[3] = (ENT_SPHERE << 28) | index;
            w[4] = e.w[3]; w[5] = e.w[4]; w[6] = e.w[5]; w[7] = e.w[6];
        } else {
            const size_t q = kind == ENT_TRI ? (size_t)index + n_quads : index;
            const float* r = planar + 20 * q;  // (Q,d) (u,mat) (v,0) (n,0) (w,0)
            std::memcpy(w + 0, r + 12, 12);  w[3] = (kind << 28) | index;   // n, tag
            std::memcpy(w + 4, r + 0, 16);                                  // Q, d
            std::memcpy(w + 8, r + 4, 16);                                  // u, material
            std::memcpy(w + 12, r + 8, 12);  w[15] = planar_id[q];          // v, prim id
            std::memcpy(w + 16, r + 16, 12); w[19] = 0;                     // w
        }
    }

    // Fills node `ni` from the binary subtree under box entry `e`.
    bool node(uint32_t ni, uint32_t e, uint32_t level) {
        if (level > WIDE_MAX_LEVELS) return false;
        out.levels = std::max(out.levels, level);
        uint32_t kids[4];
        uint32_t n = 0;
        if (internal(e)) {
            kids[n++] = e + 1;
            kids[n++] = t.ents[e + 1].w[6];
        } else {
            kids[n++] = e;  // a tree that is one leaf
        }
        // Open the largest child that has children of its own -- unless it is small against the node: its children's boxes
        // go onto the node's 256-step grid, where anything much smaller than a grid step is as good as lost (a mesh of
        // size 1 beside a ground sphere of radius 1000: the mesh stays one child here and gets a grid of its own below).
        const float open_min = extent(e) * (1.0f / 16.0f);
        while (n < 4) {
            int pick = -1;
            float best = -1.0f;
            for (uint32_t k = 0; k < n; k++)
                if (internal(kids[k]) && extent(kids[k]) >= open_min && area(kids[k]) > best) {
                    best = area(kids[k]);
                    pick = (int)k;
                }
            if (pick < 0) break;
            const uint32_t l = kids[pick] + 1, r = t.ents[l].w[6];
            kids[pick] = l;
            kids[n++] = r;
        }
        std::stable_partition(kids, kids + n, [&](uint32_t k) { return internal(k); });
        uint32_t n_int = 0;
        while (n_int < n && internal(kids[n_int])) n_int++;
        const uint32_t child_base = (uint32_t)out.n_nodes();
        out.nodes.resize(out.nodes.size() + (size_t)n_int * WIDE_NODE_WORDS, 0u);
        const uint32_t prim_base = (uint32_t)out.n_slots();
        uint32_t leaves = 0u, n_prims = 0;
        for (uint32_t k = n_int; k < n; k++) {
            const uint32_t first = kids[k] + 1, end = t.ents[kids[k]].w[6], count = end - first;
            if (count == 0 || count > WIDE_MAX_LEAF) return false;
            for (uint32_t p = first; p < end; p++) {
                if (t.ents[p].w[7] == ENT_BOX || (t.ents[p].w[7] & 15u) > ENT_TRI) return false;
                put_prim(p);
            }
            leaves |= ((count << 5) | n_prims) << (8 * k);
            n_prims += count;
        }
        // the box around all children and the children's boxes on its 256-step grid, rounded outward
        double lo[4][3], hi[4][3], nlo[3], nhi[3];
        for (uint32_t k = 0; k < n; k++)
            if (!box_of(kids[k], lo[k], hi[k])) return false;
        uint32_t* w = &out.nodes[(size_t)ni * WIDE_NODE_WORDS];
        uint32_t qlo[3] = {0u, 0u, 0u}, qhi[3] = {0u, 0u, 0u};
        for (int a = 0; a < 3; a++) {
            nlo[a] = lo[0][a];
            nhi[a] = hi[0][a];
            for (uint32_t k = 1; k < n; k++) {
                nlo[a] = std::fmin(nlo[a], lo[k][a]);
                nhi[a] = std::fmax(nhi[a], hi[k][a]);
            }
            float origin = (float)nlo[a];
            if ((double)origin > nlo[a]) origin = std::nextafter(origin, -std::numeric_limits<float>::infinity());
            float scale = (float)((nhi[a] - (double)origin) / 255.0);
            scale = std::nextafter(std::fmax(scale, 1e-30f), std::numeric_limits<float>::infinity());
            while ((double)origin + 255.0 * (double)scale < nhi[a]) scale = std::nextafter(scale, std::numeric_limits<float>::infinity());
            w[a] = fbits(origin);
            w[4 + a] = fbits(scale);
            for (uint32_t k = 0; k < 4; k++) {
                uint32_t l = 255u, h = 0u;  // an unused child is never hit
                if (k < n) {
                    double fl = std::floor((lo[k][a] - (double)origin) / (double)scale), fh = std::ceil((hi[k][a] - (double)origin) / (double)scale);
                    fl = std::fmin(std::fmax(fl, 0.0), 255.0);
                    fh = std::fmin(std::fmax(fh, 0.0), 255.0);
                    while (fl > 0.0 && (double)origin + fl * (double)scale > lo[k][a]) fl -= 1.0;
                    while (fh < 255.0 && (double)origin + fh * (double)scale < hi[k][a]) fh += 1.0;
                    if ((double)origin + fl * (double)scale > lo[k][a] || (double)origin + fh * (double)scale < hi[k][a]) return false;
                    l = (uint32_t)fl;
                    h = (uint32_t)fh;
                }
                qlo[a] |= l << (8 * k);
                qhi[a] |= h << (8 * k);
            }
            w[8 + a] = qlo[a];
            w[12 + a] = qhi[a];
        }
        w[3] = child_base;
        w[7] = prim_base;
        w[11] = leaves;
        w[15] = n_int | (n << 8);
        for (uint32_t k = 0; k < n_int; k++)
            if (!node(child_base + k, kids[k], level + 1)) return false;
        return true;
    }
};

// Turn entry indices into byte offsets and append the terminator.
static inline void finalize(Threaded& t) {
    const uint32_t S = (uint32_t)sizeof(Ent);
    for (Ent& e : t.ents) {
        uint32_t kind = e.w[7] == ENT_BOX ? ENT_BOX : (e.w[7] & 15u);
        if (kind == ENT_BOX || kind == ENT_MED_BEGIN || kind == ENT_MED_END || kind == ENT_XF_PUSH) e.w[6] *= S;
        if ((kind == ENT_XF_PUSH || kind == ENT_XF_POP) && e.w[4] != NONE) e.w[4] *= S;
    }
    Ent end;
    std::memset(&end, 0, sizeof(end));
    end.w[7] = ENT_END;
    t.ents.push_back(end);
    t.rank.push_back(0);
    t.mat.push_back(NONE);
    // chain records of the ENT_INST entries go behind the terminator, where no walk comes by; the entries name them by byte offset
    const uint32_t chains_at = (uint32_t)t.ents.size();
    for (uint32_t i = 0; i + 1 < chains_at; i++) {
        Ent& e = t.ents[i];
        if (e.w[7] == ENT_BOX || (e.w[7] & ENT_INST) == 0u) continue;
        uint32_t& c = e.w[(e.w[7] & 3u) == ENT_SPHERE ? 3 : 1];
        c = (chains_at + c) * S;
    }
    for (const Ent& c : t.chains) {
        t.ents.push_back(c);
        t.rank.push_back(0);
        t.mat.push_back(NONE);
    }
}

// Build both arrays.  Returns RTX_OK or an error status with `err` set.
// drop_unculling: see Builder::never_culls (the caller asks for it when the fast array will be walked in its threaded form).
// max_leaf: most primitives under one leaf box of the fast tree (RTX_SAH_MAXLEAF overrides it); planar_cost: what a quad or
// triangle test costs in the builder's leaf decision, in sphere tests.
static inline int build(const rtx_scene_desc_t& d, Threaded& strict, Threaded& fast, Threaded& fast_indexed, float& coord_range, std::string& err,
                        bool drop_unculling = false, uint32_t max_leaf = 4, float planar_cost = 1.0f) {
    Builder b(d);
    b.MAX_LEAF = std::max(1u, max_leaf);
    b.PLANAR_COST = planar_cost;
    if (const char* e = std::getenv("RTX_SAH_MAXLEAF")) b.MAX_LEAF = (uint32_t)std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RTX_SAH_CTRAV")) b.C_TRAV = (float)std::atof(e);
    if (const char* e = std::getenv("RTX_SAH_BARE_LEAF")) b.bare_leaves = std::atoi(e) != 0;
    if (const char* e = std::getenv("RTX_SAH_PLANAR_COST")) b.PLANAR_COST = (float)std::atof(e);
    if (!b.emit_strict(strict, d.root, NONE, 0, 0)) {
        err = b.err;
        return b.unsupported ? RTX_ERR_UNSUPPORTED : RTX_ERR_INVALID;
    }
    if (strict.n_prims == 0) { err = "empty scene: no primitive is reachable from the root"; return RTX_ERR_INVALID; }
    // coordinate range of the scene: every ray the render kernels trace starts at the camera or on a primitive, in
    // world space or in the local space of a transform (bounds_of doubles the local ranges it meets)
    Box3 world;
    b.coord_range = 0.0f;
    if (!b.bounds_of(d.root, world)) {
        err = b.err;
        return b.unsupported ? RTX_ERR_UNSUPPORTED : RTX_ERR_INVALID;
    }
    b.coord_range = std::fmax(b.coord_range, world.max_abs());
    if (!std::isfinite(b.coord_range)) { err = "non-finite scene bounds"; return RTX_ERR_INVALID; }
    b.scene_tight = world;
    b.drop_unculling = drop_unculling;
    if (const char* e = std::getenv("RTX_PLACE_INSTANCES")) b.place_instances = std::atoi(e) != 0;
    if (!b.emit_fast_subtree(fast, d.root, NONE, 0, 0)) {
        err = b.err;
        return b.unsupported ? RTX_ERR_UNSUPPORTED : RTX_ERR_INVALID;
    }
    // Instances are placed all or none: an array that still needs transform brackets for some instance (nested deeper
    // than the chain record holds, over a moving sphere or a medium) is walked by the variants that change frames, and
    // those are kept free of the placed-primitive path.
    if (b.place_instances && fast.has_frames && !b.inst_chains.empty()) {
        b.place_instances = false;
        b.inst_chains.clear();
        fast = Threaded();
        if (!b.emit_fast_subtree(fast, d.root, NONE, 0, 0)) {
            err = b.err;
            return b.unsupported ? RTX_ERR_UNSUPPORTED : RTX_ERR_INVALID;
        }
    }
    for (const std::vector<uint32_t>& objs : b.inst_chains) {
        Ent c;
        std::memset(&c, 0, sizeof(c));
        for (size_t l = 0; l < objs.size(); l++) {
            const rtx_object_t& o = d.objects[objs[l]];
            c.w[4 * l] = o.kind == RTX_OBJ_TRANSLATE ? 1u : 2u;
            for (int a = 0; a < 3; a++) c.w[4 * l + 1 + a] = fbits(o.f[a]);
        }
        fast.chains.push_back(c);
    }
    coord_range = b.coord_range;
    if (strict.ents.size() >= (1u << 26) || fast.ents.size() >= (1u << 26)) { err = "scene too large for 32-bit entry offsets"; return RTX_ERR_INVALID; }
    fast_indexed = fast;  // kept by the scene: re-ordered for a viewpoint when a render asks for it
    finalize(strict);
    finalize(fast);
    return RTX_OK;
}

}  // namespace rtx_accel
