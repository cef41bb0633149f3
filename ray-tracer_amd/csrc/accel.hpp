// Host-side construction of the traversal data the kernels walk.
//
// Both traversal modes use the same device structure: a "threaded" array of 32-byte entries in depth-first
// order.  An entry is either a box (six bounds + the index to jump to when the ray misses it) or a primitive
// (spheres are stored inline, planar primitives by index); a hit box falls through to the next entry.  Walking
// the array front to back therefore IS a left-first depth-first traversal without a stack.
//
//  strict: the array is the reference's own tree (Bvh::from, mesh.rs:560-605, handed over in bvh_nodes) with
//          the reference's padded, origin-including boxes, so Bvh::hit (mesh.rs:608-620) is reproduced test
//          for test, cull for cull.
//  fast  : a binned-SAH tree built here over tight (slightly padded) primitive boxes; the closest hit is
//          order independent, and ties are resolved with the rank a primitive has in the strict order.
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

enum : uint32_t { ENT_BOX = 0, ENT_SPHERE = 1, ENT_QUAD = 2, ENT_TRI = 3 };
// word 7 of an entry doubles as the walk state of a lane standing at it:
//   0                      a box            (words 0-5 bounds, word 6 = byte offset of the entry to jump to on a miss)
//   (index << 4) | kind    a primitive      (kind 1..3)
//   ENT_END                the terminator appended after the last entry: the walk is complete
constexpr uint32_t ENT_END = 0xFFFFFFFEu;

struct Ent {
    uint32_t w[8];
};
static_assert(sizeof(Ent) == 32, "entry must be 32 bytes");

struct Threaded {
    std::vector<Ent> ents;
    std::vector<uint32_t> rank;  // per entry: tie-break rank of a primitive entry (its first position in the strict order)
    uint32_t n_boxes = 0, n_prims = 0;
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

// -------------------------------------------------------------------------------------------------- strict
struct StrictBuilder {
    const rtx_scene_desc_t& d;
    Threaded& out;
    std::unordered_map<uint32_t, uint32_t> first_rank;  // prim ref -> first position among primitive tests
    uint32_t next_rank = 0;
    std::string err;
    int depth = 0;

    StrictBuilder(const rtx_scene_desc_t& d_, Threaded& o) : d(d_), out(o) {}

    bool emit(rtx_ref_t r) {
        if (++depth > 4096) { err = "scene graph too deep (cycle?)"; return false; }
        uint32_t k = RTX_REF_KIND(r), i = RTX_REF_INDEX(r);
        bool ok = true;
        switch (k) {
            case RTX_REF_NODE: {
                if (i >= d.n_bvh_nodes) { err = "bvh node reference out of range"; ok = false; break; }
                const rtx_bvh_node_t& n = d.bvh_nodes[i];
                for (int c = 0; c < 6; c++)
                    if (n.bounds[c] != n.bounds[c]) { err = "NaN in bvh node bounds"; ok = false; }
                if (!ok) break;
                uint32_t at = (uint32_t)out.ents.size();
                out.ents.push_back(box_entry(n.bounds, 0));
                out.rank.push_back(0);
                out.n_boxes++;
                // a one-element Bvh has left == right (mesh.rs:572-577): the reference tests it twice, so do we
                ok = emit(n.left) && emit(n.right);
                out.ents[at].w[6] = (uint32_t)out.ents.size();
                break;
            }
            case RTX_REF_SPHERE:
            case RTX_REF_QUAD:
            case RTX_REF_TRI: {
                uint32_t cnt = k == RTX_REF_SPHERE ? d.n_spheres : (k == RTX_REF_QUAD ? d.n_quads : d.n_tris);
                if (i >= cnt) { err = "primitive reference out of range"; ok = false; break; }
                auto it = first_rank.find(r);
                uint32_t rk = it == first_rank.end() ? (first_rank[r] = next_rank) : it->second;
                next_rank++;
                out.ents.push_back(prim_entry(d, k, i));
                out.rank.push_back(rk);
                out.n_prims++;
                break;
            }
            case RTX_REF_OBJECT: {
                if (i >= d.n_objects) { err = "object reference out of range"; ok = false; break; }
                const rtx_object_t& o = d.objects[i];
                if (o.kind == RTX_OBJ_LIST) {
                    // World::hit (mesh.rs:66-82) is a linear scan with a shrinking interval and no box test
                    if ((uint64_t)o.first + o.count > d.n_refs) { err = "list object exceeds refs"; ok = false; break; }
                    for (uint32_t c = 0; c < o.count && ok; c++) ok = emit(d.refs[o.first + c]);
                } else {
                    err = "Translate / RotateY / ConstantMedium are not implemented by this build";
                    ok = false;
                    unsupported = true;
                }
                break;
            }
            default: err = "unknown reference kind"; ok = false;
        }
        depth--;
        return ok;
    }
    bool unsupported = false;
};

// -------------------------------------------------------------------------------------------------- fast (binned SAH)
struct PrimBox {
    float lo[3], hi[3];
    uint32_t ref;   // RTX_REF(kind, index)
    uint32_t rank;  // tie-break rank
};

static inline void prim_bounds(const rtx_scene_desc_t& d, uint32_t kind, uint32_t i, float lo[3], float hi[3], float coord_range) {
    if (kind == RTX_REF_SPHERE) {
        const float c0[3] = {d.sph_cx[i], d.sph_cy[i], d.sph_cz[i]};
        const float dc[3] = {d.sph_dx[i], d.sph_dy[i], d.sph_dz[i]};
        float r = std::fabs(d.sph_radius[i]);
        for (int a = 0; a < 3; a++) {
            float c1 = c0[a] + dc[a];  // centre over time in [0,1) sweeps c0 .. c0+dc
            lo[a] = std::fmin(c0[a], c1) - r;
            hi[a] = std::fmax(c0[a], c1) + r;
        }
    } else if (kind == RTX_REF_QUAD) {
        for (int a = 0; a < 3; a++) {
            float q = d.quad_f[a][i], u = d.quad_f[3 + a][i], v = d.quad_f[6 + a][i];
            float p[4] = {q, q + u, q + v, q + u + v};
            lo[a] = std::fmin(std::fmin(p[0], p[1]), std::fmin(p[2], p[3]));
            hi[a] = std::fmax(std::fmax(p[0], p[1]), std::fmax(p[2], p[3]));
        }
    } else {
        for (int a = 0; a < 3; a++) {
            float p0 = d.tri_f[a][i], p1 = d.tri_f[3 + a][i], p2 = d.tri_f[6 + a][i];
            lo[a] = std::fmin(p0, std::fmin(p1, p2));
            hi[a] = std::fmax(p0, std::fmax(p1, p2));
        }
    }
    // Outward padding: the slab test and the primitive tests round differently, so a box that is exactly tight could
    // cull a ray the primitive test would still accept.  1e-3 of the box diagonal plus a few ulps of the coordinates.
    float diag2 = 0;
    for (int a = 0; a < 3; a++) diag2 += (hi[a] - lo[a]) * (hi[a] - lo[a]);
    float pad = 1e-3f * std::sqrt(diag2);
    for (int a = 0; a < 3; a++) {
        float m = std::fmax(std::fabs(lo[a]), std::fabs(hi[a]));
        // + the absolute error of the fused box test: ~1.2e-7 * (|origin| + |hit point|) <= 2.4e-7 * coord_range
        float p = pad + 4e-7f * m + 3e-7f * coord_range + 1e-30f;
        lo[a] -= p;
        hi[a] += p;
    }
}

struct FastBuilder {
    std::vector<PrimBox>& prims;
    const rtx_scene_desc_t& d;
    Threaded& out;
    static constexpr int NBINS = 16;
    uint32_t MAX_LEAF = 4;   // most primitives under one leaf box
    float C_TRAV = 1.0f;     // cost of a box test in units of a primitive test

    FastBuilder(std::vector<PrimBox>& p, const rtx_scene_desc_t& d_, Threaded& o) : prims(p), d(d_), out(o) {}

    static float half_area(const float lo[3], const float hi[3]) {
        float e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }

    void build(uint32_t begin, uint32_t end) {
        const float INF = std::numeric_limits<float>::infinity();
        float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF}, clo[3] = {INF, INF, INF}, chi[3] = {-INF, -INF, -INF};
        for (uint32_t i = begin; i < end; i++)
            for (int a = 0; a < 3; a++) {
                lo[a] = std::fmin(lo[a], prims[i].lo[a]);
                hi[a] = std::fmax(hi[a], prims[i].hi[a]);
                float c = 0.5f * (prims[i].lo[a] + prims[i].hi[a]);
                clo[a] = std::fmin(clo[a], c);
                chi[a] = std::fmax(chi[a], c);
            }
        uint32_t n = end - begin;
        uint32_t at = (uint32_t)out.ents.size();
        const float b6[6] = {lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]};
        out.ents.push_back(box_entry(b6, 0));
        out.rank.push_back(0);
        out.n_boxes++;

        // choose a split: binned SAH over the centroid bounds
        int best_axis = -1;
        int best_bin = -1;
        float best_cost = INF;
        const float parent_area = half_area(lo, hi);
        if (n > 1) {
            for (int a = 0; a < 3; a++) {
                float ext = chi[a] - clo[a];
                if (!(ext > 0)) continue;
                uint32_t cnt[NBINS] = {0};
                float blo[NBINS][3], bhi[NBINS][3];
                for (int b = 0; b < NBINS; b++)
                    for (int k = 0; k < 3; k++) { blo[b][k] = INF; bhi[b][k] = -INF; }
                float scale = (float)NBINS / ext;
                for (uint32_t i = begin; i < end; i++) {
                    float c = 0.5f * (prims[i].lo[a] + prims[i].hi[a]);
                    int b = std::min(NBINS - 1, std::max(0, (int)((c - clo[a]) * scale)));
                    cnt[b]++;
                    for (int k = 0; k < 3; k++) {
                        blo[b][k] = std::fmin(blo[b][k], prims[i].lo[k]);
                        bhi[b][k] = std::fmax(bhi[b][k], prims[i].hi[k]);
                    }
                }
                float rarea[NBINS];
                uint32_t rcnt[NBINS];
                float rl[3] = {INF, INF, INF}, rh[3] = {-INF, -INF, -INF};
                uint32_t rc = 0;
                for (int b = NBINS - 1; b > 0; b--) {
                    for (int k = 0; k < 3; k++) { rl[k] = std::fmin(rl[k], blo[b][k]); rh[k] = std::fmax(rh[k], bhi[b][k]); }
                    rc += cnt[b];
                    rarea[b] = rc ? half_area(rl, rh) : 0;
                    rcnt[b] = rc;
                }
                float ll[3] = {INF, INF, INF}, lh[3] = {-INF, -INF, -INF};
                uint32_t lc = 0;
                for (int b = 0; b < NBINS - 1; b++) {
                    for (int k = 0; k < 3; k++) { ll[k] = std::fmin(ll[k], blo[b][k]); lh[k] = std::fmax(lh[k], bhi[b][k]); }
                    lc += cnt[b];
                    if (lc == 0 || rcnt[b + 1] == 0) continue;
                    float cost = half_area(ll, lh) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        // SAH: one box test costs about as much as a primitive test here, so compare (1 + split) against n
        bool make_leaf = n <= 1 || best_axis < 0 ||
                         (n <= MAX_LEAF && !(parent_area > 0 && C_TRAV + best_cost / parent_area < (float)n));
        if (make_leaf && n > MAX_LEAF) {
            // all centroids coincide: split the range in half to bound the leaf size
            uint32_t mid = begin + n / 2;
            build(begin, mid);
            build(mid, end);
        } else if (make_leaf) {
            std::sort(prims.begin() + begin, prims.begin() + end, [](const PrimBox& x, const PrimBox& y) { return x.rank < y.rank; });
            for (uint32_t i = begin; i < end; i++) {
                out.ents.push_back(prim_entry(d, RTX_REF_KIND(prims[i].ref), RTX_REF_INDEX(prims[i].ref)));
                out.rank.push_back(prims[i].rank);
                out.n_prims++;
            }
        } else {
            float ext = chi[best_axis] - clo[best_axis];
            float scale = (float)NBINS / ext;
            int a = best_axis;
            float base = clo[a];
            auto midit = std::partition(prims.begin() + begin, prims.begin() + end, [&](const PrimBox& p) {
                float c = 0.5f * (p.lo[a] + p.hi[a]);
                int b = std::min(NBINS - 1, std::max(0, (int)((c - base) * scale)));
                return b <= best_bin;
            });
            uint32_t mid = (uint32_t)(midit - prims.begin());
            if (mid == begin || mid == end) mid = begin + n / 2;
            build(begin, mid);
            build(mid, end);
        }
        out.ents[at].w[6] = (uint32_t)out.ents.size();
    }
};

// Turn entry indices into byte offsets and append the terminator.
static inline void finalize(Threaded& t) {
    for (Ent& e : t.ents)
        if (e.w[7] == ENT_BOX) e.w[6] *= (uint32_t)sizeof(Ent);
    Ent end;
    std::memset(&end, 0, sizeof(end));
    end.w[7] = ENT_END;
    t.ents.push_back(end);
    t.rank.push_back(0);
}

// Build both arrays.  Returns RTX_OK or an error status with `err` set.
static inline int build(const rtx_scene_desc_t& d, Threaded& strict, Threaded& fast, float& coord_range, std::string& err) {
    StrictBuilder sb(d, strict);
    if (!sb.emit(d.root)) {
        err = sb.err;
        return sb.unsupported ? RTX_ERR_UNSUPPORTED : RTX_ERR_INVALID;
    }
    if (strict.n_prims == 0) { err = "empty scene: no primitive is reachable from the root"; return RTX_ERR_INVALID; }
    std::vector<PrimBox> prims;
    prims.reserve(sb.first_rank.size());
    // coordinate range of the scene: every ray the render kernels trace starts at the camera or on a primitive
    coord_range = 0.0f;
    for (auto& kv : sb.first_rank) {
        float lo[3], hi[3];
        prim_bounds(d, RTX_REF_KIND(kv.first), RTX_REF_INDEX(kv.first), lo, hi, 0.0f);
        for (int a = 0; a < 3; a++) {
            if (lo[a] != lo[a] || hi[a] != hi[a]) { err = "NaN in primitive bounds"; return RTX_ERR_INVALID; }
            coord_range = std::fmax(coord_range, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
        }
    }
    for (auto& kv : sb.first_rank) {
        PrimBox p;
        p.ref = kv.first;
        p.rank = kv.second;
        prim_bounds(d, RTX_REF_KIND(kv.first), RTX_REF_INDEX(kv.first), p.lo, p.hi, coord_range);
        prims.push_back(p);
    }
    std::sort(prims.begin(), prims.end(), [](const PrimBox& x, const PrimBox& y) { return x.rank < y.rank; });
    FastBuilder fb(prims, d, fast);
    if (const char* e = std::getenv("RTX_SAH_MAXLEAF")) fb.MAX_LEAF = (uint32_t)std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RTX_SAH_CTRAV")) fb.C_TRAV = (float)std::atof(e);
    fb.build(0, (uint32_t)prims.size());
    if (strict.ents.size() >= (1u << 26) || fast.ents.size() >= (1u << 26)) { err = "scene too large for 32-bit entry offsets"; return RTX_ERR_INVALID; }
    finalize(strict);
    finalize(fast);
    return RTX_OK;
}

}  // namespace rtx_accel
