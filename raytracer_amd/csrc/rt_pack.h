// rt_pack.h -- host-side scene preparation: the reference BVH builder, tree flattening, the
// packer that turns (triangles, flat nodes) into the kernel layout of rt_layout.h, and the
// per-frame constants.  Pure host C++ (no HIP calls), shared by the product library and by the
// host build of the pixel pipeline that the CPU tests use.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <future>
#include <map>
#include <system_error>
#include <vector>

#include "../../include/rt_mi355.h"
#include "rt_exact.h"
#include "rt_layout.h"

namespace rt {

static_assert(sizeof(rt_triangle) == 96, "reference Triangle is 96 bytes");
static_assert(sizeof(rt_flat_node) == 40, "flat node is 40 bytes");
static_assert(sizeof(rt_transform) == 24 && sizeof(rt_camera) == 40 && sizeof(rt_light) == 24, "reference struct sizes");

inline F3 f3(const rt_vec3& v) { return f3(v.x, v.y, v.z); }
inline float bits_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// MeshArgs::wide.  The triangle test's distance t = (e2 . (s x e1)) / a (ray.c:47-63) is off by about 8 u |e1| |e2| |s| / |a|
// (u = 2^-24), and the only rays it turns away are those with |a| = |e1 x e2| |cos| < 1e-7 (ray.c:50): the larger the triangle,
// the closer to its plane a ray may run and still be tested, and the farther off t -- and with it u, v, so that rays which
// miss the triangle are accepted as well.  For the reference scene's drone (rotor discs of 2.8 x 2.8) such hits came out 8 %
// in front of the plane; the distance cull's margin is 0.2 %.  Three rays in 3.6e10 of a self-checking run were affected, all
// on those triangles.  Triangles below the threshold can misbehave in the same way only within a far narrower band of
// directions and over an area hundreds of times smaller.
constexpr float kWideTriangle = 0.25f;
inline bool wide_triangle(const rt_triangle& t) {
    const float a[3] = {t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z}, b[3] = {t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z};
    const double k2 = ((double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2]) * ((double)b[0] * b[0] + (double)b[1] * b[1] + (double)b[2] * b[2]);
    return !(k2 < (double)kWideTriangle * (double)kWideTriangle);      // a NaN counts as wide
}
inline bool any_wide_triangle(const rt_triangle* tris, int32_t n) {
    for (int32_t i = 0; i < n; i++) if (wide_triangle(tris[i])) return true;
    return false;
}

// ------------------------------------------------------------------------------------------
// Reference builder (accel/bvh.c:4-59).  The tree is produced in creation order
// (node, left subtree, right subtree) and the triangle array is permuted in place.
// ------------------------------------------------------------------------------------------
class BvhBuilder {
public:
    BvhBuilder(rt_triangle* tris, rt_flat_node* out, int32_t cap) : tris_(tris), out_(out), cap_(cap) {}
    // Returns node index or -1 when out of space.
    int32_t build(int32_t start, int32_t count) {
        if (used_ >= cap_) return -1;
        const int32_t me = used_++;
        rt_flat_node& n = out_[me];
        n.start_idx = start; n.triangle_count = count; n.left = n.right = -1;
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};       // aabb.c:3-8
        for (int32_t i = 0; i < count; i++) {
            const rt_vec3* corner = &tris_[start + i].v0;                             // v0, v1, v2 are contiguous
            for (int c = 0; c < 3; c++) {                                             // aabb.c:10-23
                lo[0] = min_later(lo[0], corner[c].x); lo[1] = min_later(lo[1], corner[c].y); lo[2] = min_later(lo[2], corner[c].z);
                hi[0] = max_later(hi[0], corner[c].x); hi[1] = max_later(hi[1], corner[c].y); hi[2] = max_later(hi[2], corner[c].z);
            }
        }
        n.bounds_min = {lo[0], lo[1], lo[2]};
        n.bounds_max = {hi[0], hi[1], hi[2]};
        if (count <= 4) return me;                                                    // bvh.c:19
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        int axis = 0;
        if (ey > ex) axis = 1;                                                        // bvh.c:23
        if (ez > ex && ez > ey) axis = 2;                                             // bvh.c:24 (compares with ex, not the running max)
        float split = 0.0f;
        for (int32_t i = 0; i < count; i++) split += key(tris_[start + i], axis);     // bvh.c:28-32
        split /= (float)count;                                                        // bvh.c:33
        int32_t mid = start;
        for (int32_t i = 0; i < count; i++) {                                         // bvh.c:37-48
            if (key(tris_[start + i], axis) < split) {
                rt_triangle tmp = tris_[start + i];
                tris_[start + i] = tris_[mid];
                tris_[mid] = tmp;
                mid++;
            }
        }
        const int32_t nl = mid - start;
        if (nl > 0 && nl < count) {                                                   // bvh.c:52-55
            const int32_t l = build(start, nl);
            if (l < 0) return -1;
            const int32_t r = build(mid, count - nl);
            if (r < 0) return -1;
            out_[me].left = l; out_[me].right = r;
        }
        return me;
    }
    int32_t used() const { return used_; }

private:
    // bvh.c:30-31: one coordinate of ((v0+v1)+v2) * (1.0f/3.0f)
    static float key(const rt_triangle& t, int axis) {
        const float third = 1.0f / 3.0f;
        const float a = axis == 0 ? t.v0.x : (axis == 1 ? t.v0.y : t.v0.z);
        const float b = axis == 0 ? t.v1.x : (axis == 1 ? t.v1.y : t.v1.z);
        const float c = axis == 0 ? t.v2.x : (axis == 1 ? t.v2.y : t.v2.z);
        return ((a + b) + c) * third;
    }
    rt_triangle* tris_;
    rt_flat_node* out_;
    int32_t cap_;
    int32_t used_ = 0;
};

inline int32_t flatten_tree(const rt_tree_node* n, rt_flat_node* out, int32_t cap, int32_t& used, int depth) {
    if (used >= cap || depth > 4096) return -1;
    const int32_t me = used++;
    out[me].bounds_min = n->bounds_min; out[me].bounds_max = n->bounds_max;
    out[me].start_idx = n->start_idx; out[me].triangle_count = n->triangle_count;
    out[me].left = out[me].right = -1;
    if (n->left) { const int32_t l = flatten_tree(n->left, out, cap, used, depth + 1); if (l < 0) return -1; out[me].left = l; }
    if (n->right) { const int32_t r = flatten_tree(n->right, out, cap, used, depth + 1); if (r < 0) return -1; out[me].right = r; }
    return me;
}

// ------------------------------------------------------------------------------------------
// Packer.
// ------------------------------------------------------------------------------------------
struct PackedMesh {
    std::vector<PairNode> pairs;
    std::vector<TriRec> tris;
    std::vector<AttrRec> attrs;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint32_t root_ref = kLeafBit;     // empty leaf
    bool empty = true;
    int depth = 0;                    // deepest chain of PairNodes from the root
    // With a regrouped tree (see Packer::regroup) pairs = [regrouped tree][the reference's tree]; root_ref
    // enters the first, ref_root the second (the visiting order and the work counts of the reference).
    // Without one both are the same.  tree_pairs = nodes of the tree root_ref enters.
    uint32_t ref_root = kLeafBit;
    uint32_t tree_pairs = 0;
    bool wide = false;                // a triangle with |e1| |e2| >= kWideTriangle (MeshArgs::wide)
    // FlatGrid (see Packer::build_grid): entry references per cell, row-major [second in-plane axis][first]; axis 3 = none
    std::vector<uint32_t> grid;
    uint32_t grid_axis = 3;
    int32_t grid_n = 0;
    float grid_lo[2] = {0, 0}, grid_inv[2] = {0, 0}, grid_slack[2] = {0, 0}, grid_bmax[2] = {0, 0};
    // A third tree, behind the other two in `pairs`, for renderings that need not be bit-identical (see Packer::build_loose):
    // its leaves index loose_tris, the triangle records in that tree's own order.  loose_root = root_ref (and loose_tris
    // empty) where there is none.
    uint32_t loose_root = kLeafBit;
    uint32_t loose_pairs = 0;
    int loose_depth = 0;
    std::vector<TriRec> loose_tris;
    // A tree over the reference's leaves in a sheared frame (Packer::build_shear): grid_axis = 4 + kept axis, `grid` holds its
    // header, directions (grid_n of them) and leaf records; shear_root enters it.
    uint32_t shear_root = kLeafBit;
    uint32_t shear_pairs = 0;
};

// Per-frame part of a sheared tree's gate (rt_pixel.h: shear_gate), for a MeshArgs whose placement constants are filled in;
// `blob` is the host copy of what MeshArgs::grid points to.  Which families of the mesh's triangles a ray runs too close to is
// a question about its direction alone.  Every shadow ray of the frame has the direction lit_d: the answer is given here, once
// (grid_inv: the mask of close families).  The camera rays that reach the gate pass the mesh's root box, so their directions
// lie in the cone from cam_o around that box: families whose plane stays clear of the whole cone need no test per ray (grid_lo:
// the mask of families still to test).  grid_slack[0] says which of the two is valid (bit 0 camera, bit 1 shadow).  Both err
// on the side of testing: a family is left out only where double arithmetic clears it with room to spare.
inline void shear_frame_masks(const uint32_t* blob, MeshArgs& M) {
    if (M.grid_axis < 4u || !blob) return;
    const int n = M.grid_n;
    uint64_t shadow_close = 0, camera_check = 0;
    bool shadow_ok = true, camera_ok = true;
    double d[3] = {M.lit_d[0], M.lit_d[1], M.lit_d[2]};
    // the cone of the camera rays: from cam_o around the sphere that holds the root box (a tenth larger: a ray that passes the
    // float slab test passes the box itself to within rounding)
    double centre[3], half = 0.0, to[3], dist = 0.0;
    for (int a = 0; a < 3; a++) {
        centre[a] = 0.5 * ((double)M.root_min[a] + (double)M.root_max[a]);
        const double h = 0.5 * ((double)M.root_max[a] - (double)M.root_min[a]);
        half += h * h;
        to[a] = centre[a] - (double)M.cam_o[a];
        dist += to[a] * to[a];
    }
    half = std::sqrt(half) * 1.1 + 1e-3;
    dist = std::sqrt(dist);
    if (!(dist > 1.5 * half) || !(dist < 1e15)) camera_ok = false;
    const double sin_cone = camera_ok ? half / dist : 1.0, cos_cone = std::sqrt(1.0 - sin_cone * sin_cone);
    for (int k = 0; k < n && k < 64; k++) {
        float rec[4];
        memcpy(rec, blob + 16 + 8 * (size_t)k, 16);
        const double c = rec[3];
        const double along = std::fabs(d[0] * rec[0] + d[1] * rec[1] + d[2] * rec[2]);
        if (!(along >= c + 4e-6)) shadow_close |= (uint64_t)1 << k;          // also when a NaN is about
        if (camera_ok) {
            const double axis = std::fabs((to[0] * rec[0] + to[1] * rec[1] + to[2] * rec[2]) / dist);
            const double across = std::sqrt(std::max(0.0, 1.0 - axis * axis));
            const double least = axis * cos_cone - across * sin_cone;                    // smallest |d . n| over unit directions in the cone
            if (!(least * 0.998 >= c + 1e-5)) camera_check |= (uint64_t)1 << k;           // 0.998: the gate admits |d| within a thousandth of 1
        }
    }
    if (n > 64) shadow_ok = camera_ok = false;
    uint32_t w[4] = {(uint32_t)camera_check, (uint32_t)(camera_check >> 32), (uint32_t)shadow_close, (uint32_t)(shadow_close >> 32)};
    memcpy(&M.grid_lo[0], &w[0], 4); memcpy(&M.grid_lo[1], &w[1], 4);
    memcpy(&M.grid_inv[0], &w[2], 4); memcpy(&M.grid_inv[1], &w[3], 4);
    const uint32_t valid = (camera_ok ? 1u : 0u) | (shadow_ok ? 2u : 0u);
    memcpy(&M.grid_slack[0], &valid, 4);
}

// The grid fields of a MeshArgs from a packed mesh; d_grid = where pk.grid lives for the code that will traverse.
inline void set_grid_args(const PackedMesh& pk, const uint32_t* d_grid, MeshArgs& M) {
    M.grid = pk.grid_axis != 3 ? d_grid : nullptr;
    M.grid_axis = M.grid ? pk.grid_axis : 3u;
    M.grid_n = pk.grid_n;
    M.shear_root = pk.shear_root;
    for (int j = 0; j < 2; j++) { M.grid_lo[j] = pk.grid_lo[j]; M.grid_inv[j] = pk.grid_inv[j]; M.grid_slack[j] = pk.grid_slack[j]; M.grid_bmax[j] = pk.grid_bmax[j]; }
}

class Packer {
public:
    Packer(const rt_triangle* tris, int32_t n_tris, const rt_flat_node* nodes, int32_t n_nodes, PackedMesh& out, bool regroup = false)
        : tris_(tris), n_tris_(n_tris), nodes_(nodes), n_nodes_(n_nodes), out_(out), regroup_(regroup) {}

    // Returns RT_OK or an rt_status; err gets the reason.
    int run(std::string& err) {
        out_ = PackedMesh();
        if (n_tris_ < 0 || n_nodes_ < 0 || (n_tris_ > 0 && !tris_) || (n_nodes_ > 0 && !nodes_)) { err = "null or negative mesh arrays"; return RT_ERR_INVALID; }
        if (n_tris_ > RT_MAX_MESH_TRIS) { err = "too many triangles in one mesh"; return RT_ERR_UNSUPPORTED; }
        out_.tris.resize((size_t)n_tris_);
        out_.attrs.resize((size_t)n_tris_);
        for (int32_t i = 0; i < n_tris_; i++) {
            const rt_triangle& s = tris_[i];
            TriRec& t = out_.tris[(size_t)i];
            const F3 v0 = f3(s.v0), e1 = sub(f3(s.v1), v0), e2 = sub(f3(s.v2), v0);   // ray.c:45-46
            t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z; t.prio = 0;
            t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z; t.pad1 = bits_as_float((uint32_t)i);   // the triangle's id (see TriRec)
            t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z; t.pad2 = 0;
            AttrRec& a = out_.attrs[(size_t)i];
            a.t0[0] = s.t0.u; a.t0[1] = s.t0.v; a.t1[0] = s.t1.u; a.t1[1] = s.t1.v; a.t2[0] = s.t2.u; a.t2[1] = s.t2.v;
            a.n0[0] = s.n0.x; a.n0[1] = s.n0.y; a.n0[2] = s.n0.z;
            a.n1[0] = s.n1.x; a.n1[1] = s.n1.y; a.n1[2] = s.n1.z;
            a.n2[0] = s.n2.x; a.n2[1] = s.n2.y; a.n2[2] = s.n2.z;
            a.pad = 0;
        }
        out_.wide = any_wide_triangle(tris_, n_tris_);
        if (n_nodes_ == 0) return RT_OK;                         // nothing to traverse
        covered_.assign((size_t)n_tris_, 0);
        visited_.assign((size_t)n_nodes_, 0);
        const rt_flat_node& root = nodes_[0];
        memcpy(out_.root_min, &root.bounds_min, 12);
        memcpy(out_.root_max, &root.bounds_max, 12);
        uint32_t ref = 0;
        const int st = emit(0, 0, ref, err);
        if (st != RT_OK) return st;
        out_.root_ref = ref;
        out_.ref_root = ref;
        out_.tree_pairs = (uint32_t)out_.pairs.size();
        out_.empty = n_tris_ == 0;                               // a tree over no triangles (one node, bvh.c:8-18) never yields a hit either
        if (out_.depth + 2 > kStackDepth || out_.depth > RT_MAX_TREE_DEPTH) { err = "BVH deeper than the traversal stack"; return RT_ERR_UNSUPPORTED; }
        if (regroup_ && leaves_ok_ && leaves_.size() >= 4 && leaves_.size() <= ((size_t)1 << 23)) regroup();
        if (regroup_) build_grid();
        out_.loose_root = out_.root_ref; out_.loose_pairs = out_.tree_pairs;
        if (regroup_ && out_.grid_axis == 3) build_loose();
        if (regroup_ && out_.grid_axis == 3) build_shear();
        return RT_OK;
    }

    // The second tree for a mesh whose reference tree was packed elsewhere (on the device): pk.pairs / pk.root_ref hold that
    // tree, every leaf one reference.  Its leaves are collected in the reference's visiting order (everything below the left
    // child, then everything below the right one) with the boxes their parents hold for them, and pk gets what run() would
    // have given it.  Returns false when nothing was done (a single leaf, fewer than four leaves, more than the limit, a
    // malformed or too deep tree).
    static bool regroup_packed(PackedMesh& pk) {
        if ((pk.root_ref & kLeafBit) || pk.pairs.empty()) return false;
        Packer P(nullptr, 0, nullptr, 0, pk, true);
        struct Visit { uint32_t node; int stage; };
        std::vector<Visit> path(1, Visit{pk.root_ref, 0});
        size_t visited = 0;
        while (!path.empty()) {
            const size_t top = path.size() - 1;
            const uint32_t node = path[top].node;
            const int stage = path[top].stage++;
            if (node >= pk.pairs.size() || path.size() > (size_t)RT_MAX_TREE_DEPTH + 8) return false;
            const PairNode& n = pk.pairs[node];
            if (stage == 0) {
                if (++visited > pk.pairs.size()) return false;                 // not a tree
                if (n.lref & kLeafBit) P.note_leaf(n.lmin, n.lmax, n.lref); else path.push_back(Visit{n.lref, 0});
            } else if (stage == 1) {
                if (n.rref & kLeafBit) P.note_leaf(n.rmin, n.rmax, n.rref); else path.push_back(Visit{n.rref, 0});
            } else {
                path.pop_back();
            }
        }
        if (P.leaves_.size() < 4 || P.leaves_.size() > ((size_t)1 << 23)) return false;
        const uint32_t before = pk.root_ref;
        P.regroup();
        return pk.root_ref != before || pk.ref_root != before;
    }

private:
    void note_leaf(const float* lo, const float* hi, uint32_t ref) {
        LeafItem it;
        memcpy(it.lo, lo, 12); memcpy(it.hi, hi, 12);
        it.ref = ref;
        leaves_.push_back(it);
    }
    int emit(int32_t ni, int depth, uint32_t& ref, std::string& err) {
        if (ni < 0 || ni >= n_nodes_) { err = "node index out of range"; return RT_ERR_INVALID; }
        if (visited_[(size_t)ni]) { err = "node reachable twice (not a tree)"; return RT_ERR_INVALID; }
        visited_[(size_t)ni] = 1;
        if (depth > RT_MAX_TREE_DEPTH + 8) { err = "BVH deeper than the traversal stack"; return RT_ERR_UNSUPPORTED; }
        const rt_flat_node& n = nodes_[ni];
        if (n.left < 0 && n.right < 0) {
            const int64_t s = n.start_idx, c = n.triangle_count;
            if (s < 0 || c < 0 || s + c > n_tris_) { err = "leaf triangle range out of bounds"; return RT_ERR_INVALID; }
            for (int64_t i = c - 1; i >= 0; i--) {                // later leaf > earlier leaf; inside a leaf lower index > higher
                if (covered_[(size_t)(s + i)]) { err = "triangle referenced by two leaves"; return RT_ERR_INVALID; }
                covered_[(size_t)(s + i)] = 1;
                out_.tris[(size_t)(s + i)].prio = ++next_prio_;
            }
            const int st = emit_leaf((uint32_t)s, (uint32_t)c, n, depth, ref, err);
            if (c <= kMaxLeafTris) {
                LeafItem it;
                memcpy(it.lo, &n.bounds_min, 12); memcpy(it.hi, &n.bounds_max, 12);
                it.ref = ref;
                leaves_.push_back(it);
            } else {
                leaves_ok_ = false;                               // a chain of PairNodes stands for this leaf
            }
            return st;
        }
        if (n.left < 0 || n.right < 0) { err = "internal node with a single child"; return RT_ERR_UNSUPPORTED; }
        if (out_.pairs.size() >= (size_t)kLeafBit) { err = "too many internal nodes"; return RT_ERR_UNSUPPORTED; }
        const uint32_t me = (uint32_t)out_.pairs.size();
        out_.pairs.emplace_back();
        if (depth + 1 > out_.depth) out_.depth = depth + 1;
        uint32_t l = 0, r = 0;
        int st = emit(n.left, depth + 1, l, err);
        if (st != RT_OK) return st;
        st = emit(n.right, depth + 1, r, err);
        if (st != RT_OK) return st;
        PairNode& p = out_.pairs[me];
        memcpy(p.lmin, &nodes_[n.left].bounds_min, 12);  memcpy(p.lmax, &nodes_[n.left].bounds_max, 12);
        memcpy(p.rmin, &nodes_[n.right].bounds_min, 12); memcpy(p.rmax, &nodes_[n.right].bounds_max, 12);
        p.lref = l; p.rref = r; p.pad[0] = p.pad[1] = 0;
        ref = me;
        return RT_OK;
    }
    // A leaf of up to kMaxLeafTris triangles is one reference.  A larger one (the reference makes
    // them when a split leaves one side empty) becomes a chain of PairNodes whose two boxes both
    // repeat the leaf's own box: the slab test on them repeats the test that already passed, so
    // the set of triangles tested, and the result, are unchanged.
    int emit_leaf(uint32_t start, uint32_t count, const rt_flat_node& box, int depth, uint32_t& ref, std::string& err) {
        if (count <= (uint32_t)kMaxLeafTris) { ref = leaf_ref(start, count); return RT_OK; }
        if (depth > RT_MAX_TREE_DEPTH + 8) { err = "oversize leaf too deep"; return RT_ERR_UNSUPPORTED; }
        const uint32_t me = (uint32_t)out_.pairs.size();
        out_.pairs.emplace_back();
        if (depth + 1 > out_.depth) out_.depth = depth + 1;
        const uint32_t half = count / 2;
        uint32_t l = 0, r = 0;
        int st = emit_leaf(start, half, box, depth + 1, l, err);
        if (st != RT_OK) return st;
        st = emit_leaf(start + half, count - half, box, depth + 1, r, err);
        if (st != RT_OK) return st;
        PairNode& p = out_.pairs[me];
        memcpy(p.lmin, &box.bounds_min, 12); memcpy(p.lmax, &box.bounds_max, 12);
        memcpy(p.rmin, &box.bounds_min, 12); memcpy(p.rmax, &box.bounds_max, 12);
        p.lref = l; p.rref = r; p.pad[0] = p.pad[1] = 0;
        ref = me;
        return RT_OK;
    }

    // ---- a second tree over the same leaves --------------------------------------------------------
    // The reference reaches a leaf iff the slab test passes on the leaf's box and on the box of every
    // ancestor (bvh.c:76-78, 113-124).  Every ancestor's box contains the leaf's, and the slab arithmetic
    // (aabb.c:39-55: subtract the origin, multiply by the reciprocal direction, min/max) is monotone in the
    // box's bounds -- a float subtraction and a float multiplication round monotonically, infinities
    // included -- so an enclosing box's interval contains the leaf's on every axis and "tmax >= tmin &&
    // tmax > 0" holds for it whenever it holds for the leaf.  Hence the reference tests exactly the leaves
    // whose own box the ray passes, and ANY tree over these leaves whose inner boxes are unions of its
    // leaves' boxes leads to the same leaves.  (Only a product 0 * inf = NaN, which fminf/fmaxf drop, can
    // break the chain: a direction component that is exactly zero.  Rays with one take the reference's
    // tree, see trace_mesh.)  Which leaf is visited first does not matter either: of equal t the triangle
    // with the higher priority wins, and priorities were given in the reference's order above.
    // The tree built here splits by a binned surface-area heuristic over the leaves' boxes; the
    // reference's mean split costs about a sixth more box tests per ray on its own scene.
    struct LeafItem { float lo[3], hi[3]; uint32_t ref; };

    static float half_area(const float* lo, const float* hi) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }

    // Builds the subtree over items [first, first + count) into `fast`; returns its reference and box.
    // Node numbering is pre-order (a node, its left subtree, its right subtree): the subtree over `count` leaves owns the
    // count - 1 records from `base` on, so the two halves of a large range can be built by two threads at once.
    uint32_t regroup_range(std::vector<PairNode>& fast, uint32_t base, size_t first, size_t count, int depth, int& deepest, float* lo, float* hi) {
        LeafItem* it = leaves_.data() + first;
        for (int a = 0; a < 3; a++) { lo[a] = it[0].lo[a]; hi[a] = it[0].hi[a]; }
        if (count == 1 && items_per_leaf_ == 0) return it[0].ref;
        if (items_per_leaf_ > 0 && count <= items_per_leaf_) {          // build_loose: the items themselves, in their final order, are the leaf
            for (size_t i = 1; i < count; i++)
                for (int a = 0; a < 3; a++) { lo[a] = it[i].lo[a] < lo[a] ? it[i].lo[a] : lo[a]; hi[a] = it[i].hi[a] > hi[a] ? it[i].hi[a] : hi[a]; }
            return lists_ ? (kGridListBit | ((uint32_t)count << 24) | (lists_first_ + (uint32_t)first)) : leaf_ref((uint32_t)first, (uint32_t)count);
        }
        float clo[3], chi[3];
        for (int a = 0; a < 3; a++) { clo[a] = chi[a] = it[0].lo[a] + it[0].hi[a]; }
        for (size_t i = 1; i < count; i++)
            for (int a = 0; a < 3; a++) {
                lo[a] = it[i].lo[a] < lo[a] ? it[i].lo[a] : lo[a];
                hi[a] = it[i].hi[a] > hi[a] ? it[i].hi[a] : hi[a];
                const float c = it[i].lo[a] + it[i].hi[a];
                clo[a] = c < clo[a] ? c : clo[a]; chi[a] = c > chi[a] ? c : chi[a];
            }
        constexpr int kBins = 16;
        float best = 3.0e38f;
        int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const float extent = chi[a] - clo[a];
            if (!(extent > 0.0f)) continue;
            float blo[kBins][3], bhi[kBins][3];
            size_t bn[kBins] = {};
            for (int b = 0; b < kBins; b++) for (int k = 0; k < 3; k++) { blo[b][k] = 3.0e38f; bhi[b][k] = -3.0e38f; }
            for (size_t i = 0; i < count; i++) {
                int b = (int)(((it[i].lo[a] + it[i].hi[a]) - clo[a]) / extent * (float)kBins);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bn[b]++;
                for (int k = 0; k < 3; k++) { blo[b][k] = it[i].lo[k] < blo[b][k] ? it[i].lo[k] : blo[b][k]; bhi[b][k] = it[i].hi[k] > bhi[b][k] ? it[i].hi[k] : bhi[b][k]; }
            }
            float rlo[kBins][3], rhi[kBins][3];
            size_t rn[kBins];
            float alo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, ahi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
            size_t an = 0;
            for (int b = kBins - 1; b >= 0; b--) {                 // suffix boxes
                for (int k = 0; k < 3; k++) { alo[k] = blo[b][k] < alo[k] ? blo[b][k] : alo[k]; ahi[k] = bhi[b][k] > ahi[k] ? bhi[b][k] : ahi[k]; }
                an += bn[b];
                for (int k = 0; k < 3; k++) { rlo[b][k] = alo[k]; rhi[b][k] = ahi[k]; }
                rn[b] = an;
            }
            for (int k = 0; k < 3; k++) { alo[k] = 3.0e38f; ahi[k] = -3.0e38f; }
            an = 0;
            for (int split = 1; split < kBins; split++) {          // left = bins [0, split)
                const int b = split - 1;
                for (int k = 0; k < 3; k++) { alo[k] = blo[b][k] < alo[k] ? blo[b][k] : alo[k]; ahi[k] = bhi[b][k] > ahi[k] ? bhi[b][k] : ahi[k]; }
                an += bn[b];
                if (an == 0 || rn[split] == 0) continue;
                const float cost = half_area(alo, ahi) * (float)an + half_area(rlo[split], rhi[split]) * (float)rn[split];
                if (cost < best) { best = cost; best_axis = a; best_split = split; }
            }
        }
        size_t n_left = count / 2;
        // A run of lopsided splits could take the tree past the depth the traversal stack covers (and this recursion past its
        // own stack): from the point where only halving still fits below the limit, halve.
        int halvings = 0;
        const size_t unit = items_per_leaf_ ? items_per_leaf_ : 1;
        while ((unit << halvings) < count) halvings++;
        if (depth + halvings >= depth_cap_) best_axis = -1;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float extent = chi[a] - clo[a];
            size_t l = 0;
            for (size_t i = 0; i < count; i++) {
                int b = (int)(((it[i].lo[a] + it[i].hi[a]) - clo[a]) / extent * (float)kBins);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                if (b < best_split) { std::swap(it[i], it[l]); l++; }
            }
            n_left = l;
        }
        const uint32_t me = base;
        if (depth + 1 > deepest) deepest = depth + 1;
        PairNode p;
        uint32_t l, r;
        const uint32_t rbase = base + (uint32_t)n_left;                  // the left subtree takes base + 1 .. base + n_left - 1
        if (count >= kRegroupParallelMin && depth < 5) {
            int deep_left = 0;
            std::future<uint32_t> left;
            bool threaded = true;
            try {
                left = std::async(std::launch::async, [&]() { return regroup_range(fast, base + 1, first, n_left, depth + 1, deep_left, p.lmin, p.lmax); });
            } catch (const std::system_error&) {                         // no thread to be had: this one does both halves
                threaded = false;
            }
            r = regroup_range(fast, rbase, first + n_left, count - n_left, depth + 1, deepest, p.rmin, p.rmax);
            l = threaded ? left.get() : regroup_range(fast, base + 1, first, n_left, depth + 1, deep_left, p.lmin, p.lmax);
            if (deep_left > deepest) deepest = deep_left;
        } else {
            l = regroup_range(fast, base + 1, first, n_left, depth + 1, deepest, p.lmin, p.lmax);
            r = regroup_range(fast, rbase, first + n_left, count - n_left, depth + 1, deepest, p.rmin, p.rmax);
        }
        p.lref = l; p.rref = r; p.pad[0] = p.pad[1] = 0;
        fast[me] = p;
        return me;
    }
    static constexpr size_t kRegroupParallelMin = (size_t)1 << 16;

    void regroup() {
        std::vector<PairNode> fast(leaves_.size() - 1);
        int deepest = 0;
        float lo[3], hi[3];
        const uint32_t root = regroup_range(fast, 0, 0, leaves_.size(), 0, deepest, lo, hi);
        if (deepest + 2 > kStackDepth || deepest > RT_MAX_TREE_DEPTH) return;        // keep the reference's tree only
        const uint32_t shift = (uint32_t)fast.size();
        for (PairNode p : out_.pairs) {                           // the reference's tree moves behind the new one
            if (!(p.lref & kLeafBit)) p.lref += shift;
            if (!(p.rref & kLeafBit)) p.rref += shift;
            fast.push_back(p);
        }
        // leaf references of the new tree that point at the reference's chain nodes do not exist (leaves_ok_)
        out_.ref_root = (out_.root_ref & kLeafBit) ? out_.root_ref : out_.root_ref + shift;
        out_.root_ref = root;
        out_.tree_pairs = shift;
        out_.pairs.swap(fast);
        if (deepest > out_.depth) out_.depth = deepest;
    }

    // ---- a tree of the triangles themselves -----------------------------------------------------------
    // The reference tests a triangle when the ray passes the box of the triangle's LEAF, and its leaves are what a split at
    // the mean centroid leaves over (bvh.c:19-55): up to four triangles that need not be neighbours, in a box that overlaps
    // its neighbours'.  A tree built over the triangles themselves by the surface-area split of regroup_range, with leaves of
    // up to kLooseLeafTris triangles that ARE neighbours, tests fewer boxes and fewer triangles -- but not the same ones: a ray
    // that grazes a triangle's edge can be accepted by the triangle test (ray.c:42-67) and yet fail the slab test of a tight
    // box around it, or the other way round, so single pixels along silhouettes may differ from the reference's.  Only the
    // fast build of the kernels, whose arithmetic is not the reference's either, walks this tree.  Its leaves index
    // loose_tris, a copy of the triangle records in the tree's order; each record carries the triangle's id (TriRec::pad1),
    // which is what a hit reports, so attributes and hit ids are the reference's.  Triangles that no leaf of the reference's
    // tree holds are left out.  Appended to `pairs` behind the other trees, which keep their places.  Not built for flat
    // meshes (their grid already leads to single leaves).
    static constexpr size_t kLooseLeafTris = 8;      // measured: 4: 0.736 ms, 6: 0.732, 8: 0.662, 12: 0.663, 16: 0.676 per 8-frame launch of the animation
    void build_loose() {
        if (out_.empty || !tris_ || n_tris_ < 8 || (out_.root_ref & kLeafBit)) return;
        std::vector<LeafItem> items;
        items.reserve((size_t)n_tris_);
        for (int32_t i = 0; i < n_tris_; i++) {
            if (!covered_[(size_t)i]) continue;
            const rt_vec3* corner = &tris_[i].v0;
            LeafItem it;
            it.lo[0] = it.hi[0] = corner[0].x; it.lo[1] = it.hi[1] = corner[0].y; it.lo[2] = it.hi[2] = corner[0].z;
            for (int c = 1; c < 3; c++) {
                const float v[3] = {corner[c].x, corner[c].y, corner[c].z};
                for (int a = 0; a < 3; a++) { it.lo[a] = v[a] < it.lo[a] ? v[a] : it.lo[a]; it.hi[a] = v[a] > it.hi[a] ? v[a] : it.hi[a]; }
            }
            it.ref = (uint32_t)i;
            items.push_back(it);
        }
        if (items.size() < 8 || out_.pairs.size() + items.size() >= (size_t)kLeafBit) return;
        leaves_.swap(items);
        items_per_leaf_ = kLooseLeafTris;
        // no deeper than the mesh's other trees, if a balanced tree fits below that: the traversal stack of every kernel,
        // the exact ones included, is sized by the deepest tree of the scene
        int balanced = 1;
        while ((items_per_leaf_ << balanced) < leaves_.size()) balanced++;
        depth_cap_ = out_.depth > balanced ? out_.depth : balanced + 1;
        if (depth_cap_ > RT_MAX_TREE_DEPTH) depth_cap_ = RT_MAX_TREE_DEPTH;
        std::vector<PairNode> loose(leaves_.size() - 1);
        memset(loose.data(), 0, loose.size() * sizeof(PairNode));          // records that multi-triangle leaves leave unused
        int deepest = 0;
        float lo[3], hi[3];
        const uint32_t root = regroup_range(loose, 0, 0, leaves_.size(), 0, deepest, lo, hi);
        items_per_leaf_ = 0;
        depth_cap_ = RT_MAX_TREE_DEPTH;
        leaves_.swap(items);                                               // items: the triangles in the tree's order
        if ((root & kLeafBit) || deepest + 2 > kStackDepth || deepest > RT_MAX_TREE_DEPTH) return;
        const uint32_t shift = (uint32_t)out_.pairs.size();
        for (PairNode p : loose) {
            if (!(p.lref & kLeafBit)) p.lref += shift;
            if (!(p.rref & kLeafBit)) p.rref += shift;
            out_.pairs.push_back(p);
        }
        out_.loose_tris.resize(items.size());
        for (size_t i = 0; i < items.size(); i++) out_.loose_tris[i] = out_.tris[items[i].ref];
        out_.loose_root = root + shift;
        out_.loose_pairs = (uint32_t)loose.size();
        out_.loose_depth = deepest;
        if (deepest > out_.depth) out_.depth = deepest;
    }

    // ---- a tree in a sheared frame ---------------------------------------------------------------------
    // Axis-aligned boxes are the worst fit for long thin triangles that run along a diagonal: the box of a beam from
    // (-1, -1) to (1, 1) is the whole square, and a ray anywhere through the square tests every triangle of every such beam
    // (the reference scene's drone: 109 triangle tests per ray through its root box, nine in ten of all triangle tests of
    // a frame).  In the frame (p_a + p_b, p_a - p_b, p_k) those beams are axis-aligned.  This builds a further tree over the
    // reference's leaves whose boxes live in that frame and are walked with the sheared ray (rt_pixel.h: shear, trace_mesh);
    // a leaf of it is a record that holds the reference's own box of that leaf, which is tested the reference's way before
    // the leaf's triangles are.
    //
    // What the tree must promise: a ray that the reference's triangle test (ray.c:42-67, float) accepts on triangle i, after
    // passing the reference's box of i's leaf, reaches that leaf's record.  The boxes above a record contain its box and the
    // slab arithmetic is monotone (see regroup), so it is enough that the ray passes the record's own box, and that is
    // arranged as follows.  u = 2^-24, the ray o + t d with |d| = 1, s = o - v0, S >= |s|, a = e1 . (d x e2) = -d . N with
    // N = e1 x e2, |a| = |N| |cos| where cos = d . N / |N|.  In floats every one of a, U = s . (d x e2), V = d . (s x e1),
    // T = e2 . (s x e1) is a triple product whose computed value is off by at most 11 u times the product of the three
    // lengths (three roundings per cross component, three per dot, one for s); the code below takes 12 u and doubles every
    // bound once more.  With |cos| >= c:
    //   * the computed a has the sign of the true one and at least three quarters of its size as soon as
    //     24 u |e1| |e2| <= |N| c / 4;
    //   * the ray meets the triangle's plane in X = v0 + u* e1 + v* e2; u* < -eps_u with eps_u = 24 u S |e2| / (|N| c) makes
    //     the computed U carry the sign that gives u < 0 (ray.c:56), likewise v* < -eps_v with |e1| (ray.c:61), and u* + v* >
    //     1 + eps_3, eps_3 = 24 u (|e1| |e2| + S (|e1| + |e2|)) / (|N| c) + 8 u, makes the computed u + v exceed 1 (ray.c:61):
    //     the triangle test accepts only if X lies in the triangle widened by those three amounts;
    //   * the computed t differs from X's parameter t* by at most 24 u S |e1| |e2| / (|N| c); an accepted t is positive
    //     (ray.c:66), so either t* > 0 or the origin lies within that distance of X.
    // The box of a record is the box, in the sheared frame, of its triangles widened like that (with S = the largest |o| +
    // |vertex| a ray may have to enter the tree, c = the triangle's own bound), plus room for the rounding of the sheared ray
    // and of the slab test; a ray enters the tree only if it keeps |cos| >= c for every triangle of the mesh and starts
    // outside the hull of all widened triangles by more than the last distance (shear_gate).  For such a ray t* > 0, X lies on
    // the ray inside the record's box with that room to spare, and the slab test passes.  c is the smallest bound that keeps
    // the widening small as a distance: a hundred-and-twenty-eighth of the mesh's radius across the two edges at v0, one radius along
    // the triangle's own edges (where a needle's widening goes).  Triangles whose bound c would have
    // to exceed 1/16 (needle-thin beyond what the drone has, or of no area) make the mesh keep its other trees only; so does a
    // mesh whose triangles face too many different ways to list, or whose boxes the frame does not shrink by half.
    static constexpr double kShearGrowth = 1.0;        // how far along its own edges a widened triangle may grow, in radii of the mesh
    static constexpr double kShearCosMax = 1.0 / 16.0, kShearCosMin = 1.0 / 1024.0;
    static constexpr double kShearBeyond = 8.0;        // rays may start this far outside the mesh's radius
    static constexpr size_t kShearMaxDirections = 64;
    // Leaf records per leaf of the sheared trees (a list, at most 15): a list costs one box test per record, a level of the tree
    // two per visit and a dependent fetch.  Measured per 8-frame launch of the animation: 1: 0.565 ms, 2: 0.546, 3: 0.537,
    // 4: 0.534, 6: 0.532, 8: 0.546, 12: 0.613, 15: 0.699.
    static constexpr size_t kShearListRecords = 6;
    static constexpr double kShearAngle = 1.0 / 512.0;      // triangles whose normals differ by less share a direction

    static void shear_point(const double* p, int k, double* out) {
        if (k == 0) { out[0] = p[0]; out[1] = p[1] + p[2]; out[2] = p[1] - p[2]; }
        else if (k == 1) { out[0] = p[2] - p[0]; out[1] = p[1]; out[2] = p[2] + p[0]; }
        else { out[0] = p[0] + p[1]; out[1] = p[0] - p[1]; out[2] = p[2]; }
    }
    static float round_down(double v) { float f = (float)v; return (double)f > v ? std::nextafterf(f, -INFINITY) : f; }
    static float round_up(double v) { float f = (float)v; return (double)f < v ? std::nextafterf(f, INFINITY) : f; }

    void build_shear() {
        if (out_.empty || out_.tris.empty() || !leaves_ok_ || leaves_.size() < 8 || leaves_.size() >= ((size_t)1 << 23) || (out_.root_ref & kLeafBit)) return;
        const double U = std::ldexp(1.0, -24);
        double radius = 0.0;
        for (const TriRec& t : out_.tris)
            for (int c = 0; c < 3; c++) {
                double v[3];
                for (int a = 0; a < 3; a++) v[a] = (double)t.v0[a] + (c == 1 ? (double)t.e1[a] : (c == 2 ? (double)t.e2[a] : 0.0));
                const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                if (!(len <= 1e15)) return;                                  // NaN, infinite or huge: the other trees only
                radius = len > radius ? len : radius;
            }
        if (!(radius >= 1e-6)) return;
        const double S = radius + kShearBeyond;
        const double pad_uv = radius / 128.0;              // how far outside the two edges at v0 a widened triangle may reach at most
        struct Widened { double corner[3][3]; double behind; };    // behind: how far behind the origin, per unit of S, the plane may lie
        std::vector<Widened> widened(out_.tris.size());
        struct Direction { double n[3]; double c; std::map<uint32_t, uint32_t> members; };   // leaf (its place in leaves_) -> mask of its triangles
        std::vector<Direction> directions;
        double origin_factor = 0.0;
        double hull_lo[3] = {1e300, 1e300, 1e300}, hull_hi[3] = {-1e300, -1e300, -1e300};
        for (size_t at = 0; at < leaves_.size(); at++) {
            const LeafItem& leaf = leaves_[at];
            const uint32_t first = leaf.ref & 0x00FFFFFFu, count = (leaf.ref >> 24) & 0x7Fu;
            for (uint32_t i = first; i < first + count; i++) {
                if (i >= out_.tris.size()) return;
                const uint32_t bit = count <= 32 ? (1u << (i - first)) : 0xFFFFFFFFu;     // test_leaf's masks cover leaves of up to 32
                const TriRec& t = out_.tris[i];
                const double v0[3] = {t.v0[0], t.v0[1], t.v0[2]}, e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
                const double N[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double n2 = std::sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
                const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                if (!(n2 > 0.0) || !(l1 > 1e-9) || !(l2 > 1e-9)) return;
                const double q = l1 * l2 / n2, p = (l1 + l2) / n2;
                double c = kShearCosMin;
                c = std::max(c, 96.0 * U * q);                                                   // the sign and size of a
                c = std::max(c, 24.0 * U * S / pad_uv);                                          // eps_u, eps_v as distances
                const double longest = l1 > l2 ? l1 : l2;
                c = std::max(c, 24.0 * U * (q + S * p) * longest / (kShearGrowth * radius - 8.0 * U * longest));   // eps_3 as a distance
                c = std::max(c, 24.0 * U * S * q / (kShearGrowth * radius));                     // eps_u, eps_v along the edges
                if (!(c <= kShearCosMax)) return;
                const double eps_u = 24.0 * U * S * l2 / (n2 * c), eps_v = 24.0 * U * S * l1 / (n2 * c);
                const double eps_3 = 24.0 * U * (q + S * p) / c + 8.0 * U;
                origin_factor = std::max(origin_factor, 24.0 * U * q / c);
                widened[i].behind = 24.0 * U * q / c;
                const double bary[3][2] = {{-eps_u, -eps_v}, {1.0 + eps_3 + eps_v, -eps_v}, {-eps_u, 1.0 + eps_3 + eps_u}};
                for (int k = 0; k < 3; k++)
                    for (int a = 0; a < 3; a++) {
                        widened[i].corner[k][a] = v0[a] + bary[k][0] * e1[a] + bary[k][1] * e2[a];
                        hull_lo[a] = std::min(hull_lo[a], widened[i].corner[k][a]); hull_hi[a] = std::max(hull_hi[a], widened[i].corner[k][a]);
                    }
                // the direction family of this triangle: an existing one within kShearAngle, or a new one
                const double unit_n[3] = {N[0] / n2, N[1] / n2, N[2] / n2};
                bool found = false;
                for (Direction& dir : directions) {
                    const double along = unit_n[0] * dir.n[0] + unit_n[1] * dir.n[1] + unit_n[2] * dir.n[2];
                    const double sgn = along < 0.0 ? -1.0 : 1.0;
                    const double off[3] = {unit_n[0] - sgn * dir.n[0], unit_n[1] - sgn * dir.n[1], unit_n[2] - sgn * dir.n[2]};
                    const double away = std::sqrt(off[0] * off[0] + off[1] * off[1] + off[2] * off[2]);
                    if (away <= kShearAngle) { dir.c = std::max(dir.c, c + away); dir.members[(uint32_t)at] |= bit; found = true; break; }   // |d . n_i| >= |d . n| - |n_i - n|
                }
                if (!found) {
                    if (directions.size() >= kShearMaxDirections) return;
                    Direction dir; dir.n[0] = unit_n[0]; dir.n[1] = unit_n[1]; dir.n[2] = unit_n[2]; dir.c = c;
                    dir.members[(uint32_t)at] = bit;
                    directions.push_back(dir);
                }
            }
        }
        // room for the rounding of the sheared origin and direction and of the slab arithmetic on coordinates of up to 2 S
        const double room = S / 8192.0;
        // the frame that shrinks the leaves' boxes most, measured by the surface of the boxes as they lie in the mesh's own space
        int best_k = -1;
        double best_area = 0.0, plain_area = 0.0;
        for (const LeafItem& leaf : leaves_) plain_area += (double)half_area(leaf.lo, leaf.hi);
        std::vector<LeafItem> items(leaves_.size()), best_items;
        for (int k = 0; k < 3; k++) {
            double area = 0.0;
            for (size_t j = 0; j < leaves_.size(); j++) {
                const uint32_t first = leaves_[j].ref & 0x00FFFFFFu, count = (leaves_[j].ref >> 24) & 0x7Fu;
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for (uint32_t i = first; i < first + count; i++)
                    for (int c = 0; c < 3; c++) {
                        double g[3];
                        shear_point(widened[i].corner[c], k, g);
                        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], g[a]); hi[a] = std::max(hi[a], g[a]); }
                    }
                for (int a = 0; a < 3; a++) { items[j].lo[a] = round_down(lo[a] - room); items[j].hi[a] = round_up(hi[a] + room); }
                items[j].ref = kGridListBit | (1u << 24) | (uint32_t)j;
                // sides of the box in the mesh's space: the two sheared coordinates stretch lengths by sqrt(2)
                double side[3];
                for (int a = 0; a < 3; a++) side[a] = (double)items[j].hi[a] - (double)items[j].lo[a];
                const int kept = k == 0 ? 0 : (k == 1 ? 1 : 2);
                for (int a = 0; a < 3; a++) if (a != kept) side[a] /= std::sqrt(2.0);
                area += side[0] * side[1] + side[1] * side[2] + side[2] * side[0];
            }
            if (best_k < 0 || area < best_area) { best_k = k; best_area = area; best_items = items; }
        }
        if (!(best_area <= 0.5 * plain_area)) return;
        if (out_.pairs.size() + leaves_.size() >= (size_t)kGridListBit) return;
        // records, in the order of leaves_ as it stands now; the tree is built over copies that regroup_range may reorder
        std::vector<LeafItem> records = leaves_;
        const std::vector<LeafItem> best_items_copy = best_items;            // item j belongs to record j; regroup_range reorders its input
        leaves_.swap(best_items);
        int balanced = 1;
        while (((size_t)1 << balanced) < leaves_.size()) balanced++;
        depth_cap_ = out_.depth > balanced ? out_.depth : balanced + 1;
        if (depth_cap_ > RT_MAX_TREE_DEPTH) depth_cap_ = RT_MAX_TREE_DEPTH;
        std::vector<PairNode> tree(leaves_.size() - 1);
        memset(tree.data(), 0, tree.size() * sizeof(PairNode));
        int deepest = 0;
        float lo[3], hi[3];
        items_per_leaf_ = kShearListRecords; lists_ = true; lists_first_ = 0;
        const uint32_t root = regroup_range(tree, 0, 0, leaves_.size(), 0, deepest, lo, hi);
        std::vector<uint32_t> order_main(leaves_.size());                 // record position in the main tree's order -> leaf (place in `records`)
        for (size_t j = 0; j < leaves_.size(); j++) order_main[j] = leaves_[j].ref & 0x00FFFFFFu;
        // The same once more for rays that start inside the hull (a shadow ray leaving the mesh's own surface): there the plane
        // of a triangle may lie behind the origin by up to `behind` * S and the triangle still be accepted, so every box gets
        // that much room on all sides (sqrt(2) of it along the sheared axes) -- then it holds the origin whenever that happens,
        // and a box that holds the origin passes the slab test.  S is bounded by letting in origins with |o| <= near_reach only.
        const double near_reach = 1.5 * radius;
        std::vector<LeafItem> near_items(records.size());
        for (size_t j = 0; j < records.size(); j++) {
            const uint32_t first = records[j].ref & 0x00FFFFFFu, count = (records[j].ref >> 24) & 0x7Fu;
            double behind = 0.0;
            for (uint32_t i = first; i < first + count; i++) behind = std::max(behind, widened[i].behind);
            const double more = std::sqrt(2.0) * behind * (near_reach + radius) * 1.001;       // 1.001: |d| may differ from 1 by that much
            near_items[j] = best_items_copy[j];
            for (int a = 0; a < 3; a++) { near_items[j].lo[a] = round_down((double)near_items[j].lo[a] - more); near_items[j].hi[a] = round_up((double)near_items[j].hi[a] + more); }
        }
        leaves_.swap(near_items);
        std::vector<PairNode> near_tree(leaves_.size() - 1);
        memset(near_tree.data(), 0, near_tree.size() * sizeof(PairNode));
        int near_deepest = 0;
        lists_first_ = (uint32_t)leaves_.size();                           // its records follow the main tree's
        const uint32_t near_root = regroup_range(near_tree, 0, 0, leaves_.size(), 0, near_deepest, lo, hi);
        std::vector<uint32_t> order_near(leaves_.size());
        for (size_t j = 0; j < leaves_.size(); j++) order_near[j] = leaves_[j].ref & 0x00FFFFFFu;
        items_per_leaf_ = 0; lists_ = false; lists_first_ = 0;
        leaves_.swap(near_items);
        const bool near_ok = !(near_root & (kLeafBit | kGridListBit)) && near_deepest + 2 <= kStackDepth && near_deepest <= RT_MAX_TREE_DEPTH;
        depth_cap_ = RT_MAX_TREE_DEPTH;
        leaves_.swap(records);                                              // leaves_ as before; `records` is scratch from here on
        if ((root & (kLeafBit | kGridListBit)) || deepest + 2 > kStackDepth || deepest > RT_MAX_TREE_DEPTH) return;
        const uint32_t shift = (uint32_t)out_.pairs.size();
        for (PairNode p : tree) {
            if (!(p.lref & (kLeafBit | kGridListBit))) p.lref += shift;
            if (!(p.rref & (kLeafBit | kGridListBit))) p.rref += shift;
            out_.pairs.push_back(p);
        }
        const uint32_t near_shift = (uint32_t)out_.pairs.size();
        if (near_ok) {
            for (PairNode p : near_tree) {
                if (!(p.lref & (kLeafBit | kGridListBit))) p.lref += near_shift;
                if (!(p.rref & (kLeafBit | kGridListBit))) p.rref += near_shift;
                out_.pairs.push_back(p);
            }
            if (near_deepest > deepest) deepest = near_deepest;
        }
        // header, directions, records
        const size_t n_dir = directions.size();
        size_t n_members = 0;
        for (const Direction& dir : directions) n_members += dir.members.size();
        const size_t members_at = (size_t)16 + 8 * n_dir;
        const size_t rec_base = (members_at + 2 * n_members + 7) & ~(size_t)7;
        out_.grid.assign(rec_base + 8 * 2 * leaves_.size(), 0u);
        std::vector<uint32_t> place_main(leaves_.size());                  // leaf -> its record's position in the main tree's order
        for (size_t j = 0; j < order_main.size(); j++) place_main[order_main[j]] = (uint32_t)j;
        ShearHeader header;
        memset(&header, 0, sizeof header);
        header.reach = round_down(kShearBeyond);
        header.origin_factor = round_up(origin_factor * 1.0001);
        header.radius = round_up(radius);
        for (int a = 0; a < 3; a++) { header.hull_min[a] = round_down(hull_lo[a]); header.hull_max[a] = round_up(hull_hi[a]); }
        header.near_root = near_ok ? near_root + near_shift : 0u;
        header.near_reach = near_ok ? round_down(near_reach) : 0.0f;          // 0: there is no such tree
        header.lists_at = (uint32_t)rec_base;
        header.members_at = (uint32_t)members_at;
        memcpy(out_.grid.data(), &header, sizeof header);
        size_t next_member = 0;
        for (size_t k = 0; k < n_dir; k++) {
            const float rec[4] = {(float)directions[k].n[0], (float)directions[k].n[1], (float)directions[k].n[2],
                                  round_up(directions[k].c + 1.0 / 65536.0)};          // the rounding of n and of the dot product
            uint32_t* at = out_.grid.data() + 16 + 8 * k;
            memcpy(at, rec, 16);
            at[4] = (uint32_t)next_member; at[5] = (uint32_t)directions[k].members.size();
            for (const auto& member : directions[k].members) {
                out_.grid[members_at + 2 * next_member] = place_main[member.first];  // the leaf's record among the main tree's
                out_.grid[members_at + 2 * next_member + 1] = member.second;
                next_member++;
            }
        }
        for (size_t j = 0; j < 2 * leaves_.size(); j++) {                  // the main tree's records in its order, then the near tree's in its
            const LeafItem& leaf = leaves_[j < leaves_.size() ? order_main[j] : order_near[j - leaves_.size()]];
            uint32_t* rec = out_.grid.data() + rec_base + 8 * j;
            memcpy(rec, leaf.lo, 12); memcpy(rec + 3, leaf.hi, 12);
            rec[6] = leaf.ref; rec[7] = 0xFFFFFFFFu;
        }
        out_.grid_axis = 4u + (uint32_t)best_k;
        out_.grid_n = (int32_t)n_dir;
        out_.shear_root = root + shift;
        out_.shear_pairs = (uint32_t)(tree.size() + (near_ok ? near_tree.size() : 0));
        // A mesh that gains from this tree is made of needles, and a needle's computed distance is the least certain thing
        // about it (the last bound above): nothing of such a mesh is skipped for being farther away than the closest hit.
        out_.wide = true;
        if (deepest > out_.depth) out_.depth = deepest;
    }

    // ---- flat meshes: where a ray can start its walk ------------------------------------------------
    // Cells are padded by this share of their size on every side; half of it is what a ray's own uncertainty may use up.
    static constexpr double kGridSlackShare = 64.0;

    // Which triangles of a leaf a ray through a cell can be accepted by (ray.c:42-67) -- bit i for the leaf's i-th triangle.
    // The ray meets the mesh's plane in X, inside the padded cell [rlo, rhi]; its uncertainty about X is at most `gate` per
    // in-plane axis (16 u (|o| + |t d| + max |coordinate|), what grid_entry admits).  With the triangle's edges e1, e2 in the
    // plane (component k exactly zero), the reference's u is [s, d, e2] / [e1, d, e2] with s = o - v0: the products that
    // involve the ray's path to the plane cancel exactly in real arithmetic and to within 15 u (|o| + 3 |t d| + max
    // |coordinate|, summed over both axes) times |d_k| |e2| in floats; the denominator is d_k D, D = e1 x e2, off by at most
    // 3 u kappa of itself (kappa = (|e1_a e2_b| + |e1_b e2_a|) / |D|).  So a ray whose X lies farther than about (gate_a +
    // gate_b) outside the line through v0 along e2 gets u < 0 for certain, likewise v < 0 beyond the line along e1, and
    // u + v > 1 beyond the line v1 v2 if the distance also covers the factor (|e1| + |e2|) / |v2 - v1| by which the two
    // errors add up there and kappa's share of the triangle's height.  The three lines are moved outward by
    //     mu = 8 (gate_a + gate_b) max(1, (|e1| + |e2|) / |v2 - v1|) + 2^-18 (kappa + 1) (|e1| + |e2|)
    // -- eight times the bound -- and the triangle is left out of the cell only if the padded cell lies entirely beyond one
    // of the moved lines.  Triangles for which the argument does not hold (kappa or the factor above 4096: needles, edges of
    // length zero, non-finite vertices) are always kept, as are all triangles of a leaf with more than 32.
    uint32_t tri_mask(uint32_t leaf, int k, const int* ax, const double* rlo, const double* rhi, const double* gate) const {
        const uint32_t first = leaf & 0x00FFFFFFu, count = (leaf >> 24) & 0x7Fu;
        if (count == 0) return 0u;
        if (count > 32 || out_.tris.size() < (size_t)first + count) return 0xFFFFFFFFu;
        uint32_t mask = 0;
        for (uint32_t i = 0; i < count; i++) {
            const TriRec& t = out_.tris[(size_t)first + i];
            bool keep = true;
            const double p0[2] = {t.v0[ax[0]], t.v0[ax[1]]};
            const double e1[2] = {t.e1[ax[0]], t.e1[ax[1]]}, e2[2] = {t.e2[ax[0]], t.e2[ax[1]]};
            const double l1 = sqrt(e1[0] * e1[0] + e1[1] * e1[1]), l2 = sqrt(e2[0] * e2[0] + e2[1] * e2[1]);
            const double h[2] = {e2[0] - e1[0], e2[1] - e1[1]};
            const double l3 = sqrt(h[0] * h[0] + h[1] * h[1]);
            const double D = e1[1] * e2[0] - e1[0] * e2[1];
            const double spread = fabs(e1[0] * e2[1]) + fabs(e1[1] * e2[0]);
            const bool sound = t.e1[k] == 0.0f && t.e2[k] == 0.0f && std::isfinite(p0[0]) && std::isfinite(p0[1]) && std::isfinite(l1) &&
                               std::isfinite(l2) && std::isfinite(l3) && l1 > 0 && l2 > 0 && l3 > 0 && fabs(D) > 0 && spread <= 4096.0 * fabs(D) &&
                               l1 + l2 <= 4096.0 * l3;
            if (sound) {
                const double kappa = spread / fabs(D), worst = (l1 + l2) / l3;
                const double mu = 8.0 * (gate[0] + gate[1]) * (worst > 1.0 ? worst : 1.0) + ldexp(1.0, -18) * (kappa + 1.0) * (l1 + l2);
                // edges: (p, direction, a vertex on the inner side)
                const double v1[2] = {p0[0] + e1[0], p0[1] + e1[1]}, v2[2] = {p0[0] + e2[0], p0[1] + e2[1]};
                const double* from[3] = {p0, p0, v1};
                const double* along[3] = {e2, e1, h};
                const double len[3] = {l2, l1, l3};
                const double* inner[3] = {v1, v2, p0};
                for (int e = 0; e < 3 && keep; e++) {
                    double nx = along[e][1] / len[e], ny = -along[e][0] / len[e];           // unit normal of the edge's line
                    if (nx * (inner[e][0] - from[e][0]) + ny * (inner[e][1] - from[e][1]) > 0) { nx = -nx; ny = -ny; }   // pointing outward
                    // the corner of the padded cell that lies least far out
                    const double cx = nx > 0 ? rlo[0] : rhi[0], cy = ny > 0 ? rlo[1] : rhi[1];
                    const double out = nx * (cx - from[e][0]) + ny * (cy - from[e][1]);
                    if (out > mu) keep = false;
                }
            }
            if (keep) mask |= 1u << i;
        }
        return mask;
    }

    // Whether every ray that grid_entry sends to a cell passes the slab test of a flat leaf box (record: min, max as float
    // bits) whenever the plane's parameter ty is positive.  On the flat axis both slab parameters ARE ty (the same float
    // expression as in grid_entry).  On an in-plane axis a the test needs ty between t1 = fl(fl(min_a - o_a) inv_a) and t2
    // (max_a likewise); their real values differ from ty's by (X_a - min_a) / d_a and (max_a - X_a) / d_a, and the three
    // roundings involved move them by at most 3 u (|o_a| + |ty d_a| + max |coordinate|) / |d_a| -- less than a fifth of
    // what the ray's admitted uncertainty (gate_a = 16 u of the same sum) allows.  So if the padded cell keeps a distance
    // of gate_a from the rectangle's sides, t1 <= ty <= t2 holds in floats, tmin = tmax = ty, and "tmax >= tmin && tmax > 0"
    // (aabb.c:55) is "ty > 0".
    static bool inside_for_certain(const uint32_t* rec, const int* ax, const double* rlo, const double* rhi, const double* gate) {
        for (int j = 0; j < 2; j++) {
            const double bmin = bits_as_float(rec[ax[j]]), bmax = bits_as_float(rec[3 + ax[j]]);
            if (!(rlo[j] >= bmin + gate[j] && rhi[j] <= bmax - gate[j])) return false;
        }
        return true;
    }

    // A mesh whose root box has zero thickness along one axis k (a ground plane) is met by a ray in ONE point X of its plane,
    // and a leaf's slab test can only pass if X lies in the leaf's rectangle, up to rounding: with all direction
    // components finite and non-zero, "tmax >= tmin" needs the plane's parameter ty to lie between the two parameters of
    // each in-plane axis a, t1 = fl(fl(min_a - o_a) inv_a) and t2 likewise, and then X_a = fl(o_a + fl(ty d_a)) is within
    // 3.2 u (|o_a| + |ty d_a| + max|bound_a|) of [min_a, max_a] (u = 2^-24: two roundings in t1/t2, one in inv_a = fl(1/d_a),
    // two in X_a).  The traversal takes 16 u of that sum as the ray's uncertainty E and uses the grid only when E is at most
    // half the slack the cells were built with, the other half covering the rounding of the cell index.
    // Per cell, padded by the slack on every side: walk down from the root while only ONE child's box overlaps the padded
    // cell -- no leaf below the other child can pass for a ray whose X is in the cell -- and store where that stops (a
    // node with two overlapping children, a leaf, or nothing at all: then no leaf can pass and the walk is empty).
    // Starting there skips the top of the tree without changing which leaves are reached.
    void build_grid() {
        if (out_.empty || (out_.root_ref & kLeafBit) || out_.tree_pairs < 8) return;
        int k = -1;
        for (int a = 0; a < 3; a++)
            if (out_.root_min[a] == out_.root_max[a]) { if (k >= 0) return; k = a; }        // flat along exactly one axis
        if (k < 0) return;
        const int ax[2] = {(k + 1) % 3, (k + 2) % 3};
        double lo[2], ext[2];
        for (int j = 0; j < 2; j++) {
            lo[j] = out_.root_min[ax[j]]; ext[j] = (double)out_.root_max[ax[j]] - lo[j];
            if (!(ext[j] > 0) || !(ext[j] < 1e30)) return;
        }
        // cells well below the size of a triangle, so that most cells lie inside one: eight per side and triangle on average
        int n = (int)ceil(sqrt(64.0 * (double)(out_.tris.empty() ? (size_t)out_.tree_pairs : out_.tris.size())));
        n = n < 8 ? 8 : (n > 256 ? 256 : n);
        double cs[2], slack[2];
        for (int j = 0; j < 2; j++) { cs[j] = ext[j] / n; slack[j] = cs[j] / kGridSlackShare; }
        out_.grid.assign((size_t)n * (size_t)n, leaf_ref(0, 0));
        auto overlaps = [&](const float* bmin, const float* bmax, const double* rlo, const double* rhi) {
            for (int j = 0; j < 2; j++)
                if ((double)bmax[ax[j]] < rlo[j] || (double)bmin[ax[j]] > rhi[j]) return false;
            return true;
        };
        for (int iy = 0; iy < n; iy++)
            for (int ix = 0; ix < n; ix++) {
                const double rlo[2] = {lo[0] + ix * cs[0] - slack[0], lo[1] + iy * cs[1] - slack[1]};
                const double rhi[2] = {lo[0] + (ix + 1) * cs[0] + slack[0], lo[1] + (iy + 1) * cs[1] + slack[1]};
                uint32_t ref = out_.root_ref;
                while (!(ref & kLeafBit)) {
                    const PairNode& N = out_.pairs[ref];
                    const bool l = overlaps(N.lmin, N.lmax, rlo, rhi), r = overlaps(N.rmin, N.rmax, rlo, rhi);
                    if (l && r) break;
                    if (!l && !r) { ref = leaf_ref(0, 0); break; }
                    ref = l ? N.lref : N.rref;
                }
                out_.grid[(size_t)iy * (size_t)n + (size_t)ix] = ref;
            }
        // What a cell finally names is a list of up to kGridListMax records {leaf box, leaf reference, triangle mask}: the
        // leaves whose boxes overlap the padded cell -- a ray whose X lies in the cell can pass no other leaf's box, and a leaf
        // is reached exactly when its own box is passed (ancestors' boxes contain it), so testing the listed boxes one by one
        // reaches the leaves the walk would reach, without the boxes above them -- and, per leaf, the triangles that such a
        // ray can possibly be accepted by (tri_mask below); a leaf none of whose triangles qualifies is left out.  Only where
        // more leaves than fit overlap the cell does the walk start at the node found above.  Equal lists are stored once.
        const size_t cells = (size_t)n * (size_t)n, list_base = (cells + 7) & ~(size_t)7;
        out_.grid.resize(list_base, leaf_ref(0, 0));
        const double gate[2] = {slack[0] * 0.5, slack[1] * 0.5};
        std::map<std::vector<uint32_t>, uint32_t> known;
        std::vector<uint32_t> stack, list;
        for (int iy = 0; iy < n; iy++)
            for (int ix = 0; ix < n; ix++) {
                const size_t ci = (size_t)iy * (size_t)n + (size_t)ix;          // (the vector grows below: no reference into it)
                if (out_.grid[ci] == leaf_ref(0, 0)) continue;                              // no leaf's box overlaps the padded cell
                const double rlo[2] = {lo[0] + ix * cs[0] - slack[0], lo[1] + iy * cs[1] - slack[1]};
                const double rhi[2] = {lo[0] + (ix + 1) * cs[0] + slack[0], lo[1] + (iy + 1) * cs[1] + slack[1]};
                list.clear();
                uint32_t count = 0;
                bool fits = true;
                stack.assign(1, out_.root_ref);
                while (!stack.empty() && fits) {
                    const PairNode& N = out_.pairs[stack.back()];
                    stack.pop_back();
                    for (int side = 0; side < 2 && fits; side++) {
                        const float* bmin = side ? N.rmin : N.lmin;
                        const float* bmax = side ? N.rmax : N.lmax;
                        const uint32_t child = side ? N.rref : N.lref;
                        if (!overlaps(bmin, bmax, rlo, rhi)) continue;
                        if (!(child & kLeafBit)) { stack.push_back(child); continue; }
                        const uint32_t mask = tri_mask(child, k, ax, rlo, rhi, gate);
                        if (mask == 0) continue;
                        if (count == kGridListMax) { fits = false; break; }
                        uint32_t rec[8];
                        memcpy(&rec[0], bmin, 12); memcpy(&rec[3], bmax, 12);
                        rec[6] = child; rec[7] = mask;
                        list.insert(list.end(), rec, rec + 8);
                        count++;
                    }
                }
                if (!fits) {                                                       // too many leaves: the walk starts at a node, as found above
                    if (out_.grid[ci] & kLeafBit) out_.grid[ci] = out_.root_ref;                     // (never a leaf by itself: a leaf's own box must be tested)
                    continue;
                }
                if (count == 0) { out_.grid[ci] = leaf_ref(0, 0); continue; }
                // One leaf, one triangle, and the padded cell well inside the leaf's rectangle: a ray through the cell passes the
                // leaf's box exactly when it meets the plane in front of its origin (inside_for_certain), so the cell names the
                // triangle itself and the walk tests that distance instead of the box.
                if (count == 1 && (list[7] & (list[7] - 1u)) == 0u && ((list[6] >> 24) & 0x7Fu) <= 32u && inside_for_certain(&list[0], ax, rlo, rhi, gate)) {
                    uint32_t bit = 0;
                    while (!((list[7] >> bit) & 1u)) bit++;
                    out_.grid[ci] = leaf_ref((list[6] & 0x00FFFFFFu) + bit, 1u);
                    continue;
                }
                auto seen = known.find(list);
                uint32_t at;
                if (seen != known.end()) {
                    at = seen->second;
                } else {
                    at = (uint32_t)((out_.grid.size() - list_base) / 8);
                    if ((size_t)at + count > 0x00FFFFFFu) { if (out_.grid[ci] & kLeafBit) out_.grid[ci] = out_.root_ref; continue; }
                    out_.grid.insert(out_.grid.end(), list.begin(), list.end());
                    known.emplace(list, at);
                }
                out_.grid[ci] = kGridListBit | (count << 24) | at;
            }
        out_.grid_axis = (uint32_t)k; out_.grid_n = n;
        for (int j = 0; j < 2; j++) {
            out_.grid_lo[j] = (float)lo[j];
            out_.grid_inv[j] = (float)(1.0 / cs[j]);
            out_.grid_slack[j] = (float)(slack[j] * 0.5);                 // what the ray's own uncertainty may use up
            const float a = fabsf(out_.root_min[ax[j]]), b = fabsf(out_.root_max[ax[j]]);
            out_.grid_bmax[j] = a > b ? a : b;
        }
    }

    const rt_triangle* tris_;
    int32_t n_tris_;
    const rt_flat_node* nodes_;
    int32_t n_nodes_;
    PackedMesh& out_;
    std::vector<uint8_t> covered_, visited_;
    uint32_t next_prio_ = 0;
    bool regroup_ = false, leaves_ok_ = true;
    int depth_cap_ = RT_MAX_TREE_DEPTH;  // regroup_range: no chain of nodes longer than this
    size_t items_per_leaf_ = 0;          // regroup_range: 0 = every item is a leaf reference of its own; n = up to n items form a leaf
    bool lists_ = false;                 // regroup_range with items_per_leaf_: such a leaf is a list of records (kGridListBit), not of triangles
    uint32_t lists_first_ = 0;           // ... whose first item is this record
    std::vector<LeafItem> leaves_;
};

// ------------------------------------------------------------------------------------------
// Per-frame constants.
// ------------------------------------------------------------------------------------------
// Camera part of FrameArgs (camera.c:13-17 runs these per ray; they depend on the camera only).
inline void fill_camera(const rt_camera& cam, const rt_light& light, int32_t width, int32_t height, FrameArgs& A) {
    const F3 fwd = unit(sub(f3(cam.look_at), f3(cam.position)));
    const F3 right = unit(cross(fwd, f3(cam.up)));
    const F3 up = cross(right, fwd);
    A.width = width; A.height = height;
    A.full_height = height; A.strip_stride = 1; A.strip_offset = 0; A.first_tile_row = 0;
    A.cam_pos[0] = cam.position.x; A.cam_pos[1] = cam.position.y; A.cam_pos[2] = cam.position.z;
    A.cam_right[0] = right.x; A.cam_right[1] = right.y; A.cam_right[2] = right.z;
    A.cam_up[0] = up.x; A.cam_up[1] = up.y; A.cam_up[2] = up.z;
    A.cam_fwd[0] = fwd.x; A.cam_fwd[1] = fwd.y; A.cam_fwd[2] = fwd.z;
    A.aspect = (float)width / (float)height;                                        // scene.c:53
    // camera.c:17: (fov*0.5f) in float, then * M_PI / 180.0f in double, narrowed for tanf.
    A.scale = tanf((float)((double)(cam.fov * 0.5f) * 3.14159265358979323846 / (double)180.0f));
    A.light_dir[0] = light.direction.x; A.light_dir[1] = light.direction.y; A.light_dir[2] = light.direction.z;
    A.light_color[0] = light.color.x; A.light_color[1] = light.color.y; A.light_color[2] = light.color.z;
}

// Placement part of one MeshArgs (ray.c:5-16 and ray.c:27-37 run these per ray).
inline void fill_placement(const rt_transform& tf, const FrameArgs& A, MeshArgs& M) {
    const M4 rot = m4_mul(m4_rotation(2, tf.rotation.z), m4_mul(m4_rotation(1, tf.rotation.y), m4_rotation(0, tf.rotation.x)));
    const M4 inv = m4_inverse(m4_mul(m4_translation(f3(tf.position)), rot));
    const M4 nrm = m4_transpose(m4_inverse(rot));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) M.inv[i][j] = inv.m[i][j];
    for (int j = 0; j < 4; j++) M.wrow[j] = inv.m[3][j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M.nrm[i][j] = nrm.m[i][j];
    const F3 co = xform_point(M.inv, M.wrow, f3(A.cam_pos));
    M.cam_o[0] = co.x; M.cam_o[1] = co.y; M.cam_o[2] = co.z;
    const F3 ld = unit(xform_vector<4>(M.inv, f3(A.light_dir)));
    const F3 li = recip(ld);
    M.lit_d[0] = ld.x; M.lit_d[1] = ld.y; M.lit_d[2] = ld.z;
    M.lit_inv[0] = li.x; M.lit_inv[1] = li.y; M.lit_inv[2] = li.z;
    // With a bottom row of (+-0, +-0, +-0, w33), w33 != 0, the w of mat4_transform_point equals
    // w33 exactly for every finite point.
    M.w_mode = 0;
    if (M.wrow[0] == 0.0f && M.wrow[1] == 0.0f && M.wrow[2] == 0.0f && M.wrow[3] != 0.0f) M.w_mode = (M.wrow[3] == 1.0f) ? 2u : 1u;
}

// Sampling part of a frame: the lights and, per mesh, their directions in mesh space (ray.c:20,22
// applied to the shadow ray's direction, which is the light's for every pixel).  Call after every
// MeshArgs of A has its placement.
inline void fill_samples(const rt_light* lights, int32_t n_lights, int32_t spp, uint32_t seed, const FrameArgs& A, SampleArgs& X) {
    memset(&X, 0, sizeof(X));
    X.spp = spp; X.seed = seed; X.n_lights = n_lights;
    for (int32_t l = 0; l < n_lights; l++) {
        const F3 d = f3(lights[l].direction.x, lights[l].direction.y, lights[l].direction.z);
        X.light_dir[l][0] = d.x; X.light_dir[l][1] = d.y; X.light_dir[l][2] = d.z;
        X.light_color[l][0] = lights[l].color.x; X.light_color[l][1] = lights[l].color.y; X.light_color[l][2] = lights[l].color.z;
        for (int32_t m = 0; m < A.n_mesh; m++) {
            const F3 ld = unit(xform_vector<4>(A.mesh[m].inv, d));
            const F3 li = recip(ld);
            X.lit_d[m][l][0] = ld.x; X.lit_d[m][l][1] = ld.y; X.lit_d[m][l][2] = ld.z;
            X.lit_inv[m][l][0] = li.x; X.lit_inv[m][l][1] = li.y; X.lit_inv[m][l][2] = li.z;
        }
    }
}

// ---- work layout helpers (never used for results) -------------------------------------------
struct WorldBox { double lo[3], hi[3]; double corner[8][3]; };

// World-space corners and bounds of a mesh-space box placed by tf.
inline WorldBox world_box(const rt_transform& tf, const float* bmin, const float* bmax) {
    const M4 rot = m4_mul(m4_rotation(2, tf.rotation.z), m4_mul(m4_rotation(1, tf.rotation.y), m4_rotation(0, tf.rotation.x)));
    const M4 fwd = m4_mul(m4_translation(f3(tf.position)), rot);
    WorldBox w;
    for (int i = 0; i < 3; i++) { w.lo[i] = 1e300; w.hi[i] = -1e300; }
    for (int c = 0; c < 8; c++) {
        const double p[3] = {(c & 1) ? bmax[0] : bmin[0], (c & 2) ? bmax[1] : bmin[1], (c & 4) ? bmax[2] : bmin[2]};
        for (int i = 0; i < 3; i++) {
            const double v = fwd.m[i][0] * p[0] + fwd.m[i][1] * p[1] + fwd.m[i][2] * p[2] + fwd.m[i][3];
            w.corner[c][i] = v;
            w.lo[i] = v < w.lo[i] ? v : w.lo[i];
            w.hi[i] = v > w.hi[i] ? v : w.hi[i];
        }
    }
    return w;
}

// Screen rectangle, in 8x8-pixel tiles {x0, y0, x1, y1} (half open), of every point from which a ray
// can meet the box `b`: the box itself (camera rays) and the box swept against the light direction
// as far as the scene's bounds reach (points whose shadow ray can enter it).  Two pixels of margin.
// Returns false if the region cannot be bounded on screen (a point at or behind the camera plane).
inline bool coop_rect(const WorldBox& b, const double scene_lo[3], const double scene_hi[3], const FrameArgs& A, bool with_shadow, int rect[4]) {
    double s_max = 1e300;
    for (int i = 0; i < 3; i++) {
        const double L = A.light_dir[i];
        if (L > 0) { const double s = (b.hi[i] - scene_lo[i]) / L; s_max = s < s_max ? s : s_max; }
        else if (L < 0) { const double s = (scene_hi[i] - b.lo[i]) / -L; s_max = s < s_max ? s : s_max; }
    }
    if (!(s_max > 0)) s_max = 0;
    if (!(s_max < 1e6)) s_max = 1e6;
    if (!with_shadow) s_max = 0;
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int c = 0; c < 16; c++) {
        const double s = c < 8 ? 0.0 : s_max;
        double w[3];
        for (int i = 0; i < 3; i++) w[i] = b.corner[c & 7][i] - s * A.light_dir[i] - A.cam_pos[i];
        const double z = w[0] * A.cam_fwd[0] + w[1] * A.cam_fwd[1] + w[2] * A.cam_fwd[2];
        if (!(z > 1e-4)) return false;
        const double sx = (w[0] * A.cam_right[0] + w[1] * A.cam_right[1] + w[2] * A.cam_right[2]) / z;
        const double sy = (w[0] * A.cam_up[0] + w[1] * A.cam_up[1] + w[2] * A.cam_up[2]) / z;
        const double px = (sx / ((double)A.aspect * A.scale) + 1.0) * 0.5 * A.width - 0.5;     // inverse of camera.c:18
        const double py = (1.0 - sy / (double)A.scale) * 0.5 * A.height - 0.5;                  // inverse of camera.c:19
        if (!(px == px) || !(py == py)) return false;
        lo[0] = px < lo[0] ? px : lo[0]; hi[0] = px > hi[0] ? px : hi[0];
        lo[1] = py < lo[1] ? py : lo[1]; hi[1] = py > hi[1] ? py : hi[1];
    }
    const int tiles_x = (A.width + 7) / 8, tiles_y = (A.height + 7) / 8;
    auto tile_lo = [](double v, int n) { const double t = floor((v - 2.0) / 8.0); return t < 0 ? 0 : (t > n ? n : (int)t); };
    auto tile_hi = [](double v, int n) { const double t = floor((v + 2.0) / 8.0) + 1.0; return t < 0 ? 0 : (t > n ? n : (int)t); };
    rect[0] = tile_lo(lo[0], tiles_x); rect[1] = tile_lo(lo[1], tiles_y);
    rect[2] = tile_hi(hi[0], tiles_x); rect[3] = tile_hi(hi[1], tiles_y);
    return true;
}


// Rectangle of tiles around the projections of `n` world points, each allowed to be off by `slack` in space; false (rect =
// whole frame) when a point is at or behind the camera plane or the allowance exceeds what the projection can tell.
// The camera basis as camera.c:13-15 leaves it is orthonormal up to rounding -- unless its inputs are degenerate: an up vector
// of 1e30 overflows the normalisation and leaves right = up = 0 (every pixel then has the same ray), one of 1e-20 takes it
// into denormals and leaves vectors that are not of unit length; nothing can be told from a projection then.  True when pixels and directions correspond as the projections below assume.
inline bool camera_basis_is_orthonormal(const FrameArgs& A) {
    const float* v[3] = {A.cam_right, A.cam_up, A.cam_fwd};
    for (int i = 0; i < 3; i++)
        for (int j = i; j < 3; j++) {
            const double d = (double)v[i][0] * v[j][0] + (double)v[i][1] * v[j][1] + (double)v[i][2] * v[j][2];
            if (!(fabs(d - (i == j ? 1.0 : 0.0)) < 4e-6)) return false;           // also false for NaN; float normalisation leaves 2e-7
        }
    return true;
}

inline bool tiles_around(const double (*pts)[3], int n, double slack, const FrameArgs& A, int32_t rect[4]) {
    const int tiles_x = (A.width + 7) / 8, tiles_y = (A.full_height + 7) / 8;
    rect[0] = 0; rect[1] = 0; rect[2] = tiles_x; rect[3] = tiles_y;
    if (!camera_basis_is_orthonormal(A)) return false;
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300}, z_min = 1e300, s_max = 0.0;
    for (int c = 0; c < n; c++) {
        double w[3];
        for (int i = 0; i < 3; i++) w[i] = pts[c][i] - (double)A.cam_pos[i];
        const double z = w[0] * A.cam_fwd[0] + w[1] * A.cam_fwd[1] + w[2] * A.cam_fwd[2];
        if (!(z > 1e-4) || !(z - 2.0 * slack > 0.25 * z)) return false;      // behind the camera plane, or closer than the slack can tell
        const double sx = (w[0] * A.cam_right[0] + w[1] * A.cam_right[1] + w[2] * A.cam_right[2]) / z;
        const double sy = (w[0] * A.cam_up[0] + w[1] * A.cam_up[1] + w[2] * A.cam_up[2]) / z;
        if (!(sx == sx) || !(sy == sy)) return false;
        lo[0] = sx < lo[0] ? sx : lo[0]; hi[0] = sx > hi[0] ? sx : hi[0];
        lo[1] = sy < lo[1] ? sy : lo[1]; hi[1] = sy > hi[1] ? sy : hi[1];
        z_min = z < z_min ? z : z_min;
        const double a = fabs(sx) > fabs(sy) ? fabs(sx) : fabs(sy);
        s_max = a > s_max ? a : s_max;
    }
    if (n <= 0) return false;
    // moving a point by `slack` moves its projection by at most slack * (1 + |s|) / (z - slack); z - slack >= z / 2 here
    const double allow = 2.0 * slack * (1.0 + s_max) / z_min + (1.0 + s_max * s_max) / 16384.0;
    const double kx = 0.5 * A.width / ((double)A.aspect * A.scale), ky = 0.5 * A.full_height / (double)A.scale;
    if (!(kx > 0) || !(ky > 0) || !(kx < 1e12) || !(ky < 1e12)) return false;
    // pixel coordinate of a projection (inverse of camera.c:18-19 and scene.c:61-62): px = (sx / (aspect*scale) + 1) / 2 * width - 1/2
    const double px_lo = (lo[0] - allow) * kx + 0.5 * A.width - 0.5 - 2.0, px_hi = (hi[0] + allow) * kx + 0.5 * A.width - 0.5 + 2.0;
    const double py_lo = 0.5 * A.full_height - (hi[1] + allow) * ky - 0.5 - 2.0, py_hi = 0.5 * A.full_height - (lo[1] - allow) * ky - 0.5 + 2.0;
    auto tile_lo = [](double v, int m) { const double t = floor(v / 8.0); return t < 0 ? 0 : (t > m ? m : (int)t); };
    auto tile_hi = [](double v, int m) { const double t = floor(v / 8.0) + 1.0; return t < 0 ? 0 : (t > m ? m : (int)t); };
    rect[0] = tile_lo(px_lo, tiles_x); rect[1] = tile_lo(py_lo, tiles_y);
    rect[2] = tile_hi(px_hi, tiles_x); rect[3] = tile_hi(py_hi, tiles_y);
    return true;
}

// Sum of the magnitudes a placed box's root test works with (camera position, placement, box coordinates).
inline double placement_magnitude(const rt_transform& tf, const float* bmin, const float* bmax, const FrameArgs& A) {
    double mag = fabs(tf.position.x) + fabs(tf.position.y) + fabs(tf.position.z);
    for (int i = 0; i < 3; i++) {
        mag += fabs((double)A.cam_pos[i]);
        const double m = fabs((double)bmin[i]) > fabs((double)bmax[i]) ? fabs((double)bmin[i]) : fabs((double)bmax[i]);
        mag += m;
    }
    return mag;
}

// Tiles (8x8 pixels, {x0, y0, x1, y1} half open) outside of which a camera ray of the frame cannot pass the root box of a
// mesh placed by tf -- with certainty, i.e. including what float rounding can do to the reference's test in mesh space
// (ray.c:5-22 + aabb.c:33-56).  That test works on values whose absolute errors are a few units in the last place of
// the magnitudes involved (camera position, placement, box coordinates; the translation and the camera position cancel in
// mesh space), so a ray it accepts passes, in exact arithmetic, within `slack` of the box, where slack is taken as 2^-18 of
// the sum of those magnitudes (64 units in the last place).  The box corners are projected in double precision; the slack,
// seen from the nearest corner, an angular allowance of 2^-14 for the rounding of the ray direction itself, and two
// pixels (sample jitter, tile rounding) are added on every side.  A corner at or behind the camera plane, or a frame so
// narrow that the allowance exceeds it, gives the whole frame.
inline void mesh_screen_rect(const rt_transform& tf, const float* bmin, const float* bmax, const FrameArgs& A, int32_t rect[4]) {
    const WorldBox b = world_box(tf, bmin, bmax);
    const double slack = placement_magnitude(tf, bmin, bmax, A) * (1.0 / 262144.0);
    (void)tiles_around(b.corner, 8, slack, A, rect);
}

// The same for shadow rays (scene.c:117-139).  A shadow ray starts at a surface point P of the scene (moved 0.001 along
// the normal) and runs along the light direction L; it can pass the box only if P lies in the box swept against L, and P
// lies inside the scene's bounds, which limits the sweep to s_max.  The swept box is the convex hull of the box's corners
// and the same corners moved by -s_max L, so its projection lies inside the rectangle around those sixteen points.  P is
// seen in the pixel whose ray found it (it is computed as origin + t * direction, scene.c:118), so pixels outside the
// rectangle have no such P.  Slack as for camera rays, with the scene's extent (P's magnitude) and the 0.001 added.
inline void mesh_shadow_rect(const rt_transform& tf, const float* bmin, const float* bmax, const double scene_lo[3], const double scene_hi[3],
                             const FrameArgs& A, int32_t rect[4]) {
    const int tiles_x = (A.width + 7) / 8, tiles_y = (A.full_height + 7) / 8;
    rect[0] = 0; rect[1] = 0; rect[2] = tiles_x; rect[3] = tiles_y;
    const WorldBox b = world_box(tf, bmin, bmax);
    double mag = placement_magnitude(tf, bmin, bmax, A);
    for (int i = 0; i < 3; i++) mag += (fabs(scene_lo[i]) > fabs(scene_hi[i]) ? fabs(scene_lo[i]) : fabs(scene_hi[i]));
    const double slack = mag * (1.0 / 262144.0) + 0.002;
    // The kernels normalise the light direction in mesh space in float (ray.c:22).  A caller's own, unnormalised light whose
    // length squared leaves the float range comes out of that as zeros or NaNs, and the reference's root test then accepts by
    // the origin alone (0 * inf): no direction to sweep along.  Such lights get the whole frame.
    const double len2 = (double)A.light_dir[0] * A.light_dir[0] + (double)A.light_dir[1] * A.light_dir[1] + (double)A.light_dir[2] * A.light_dir[2];
    if (!(len2 > 1e-30 && len2 < 1e30)) return;
    double s_max = 1e300;
    for (int i = 0; i < 3; i++) {
        const double L = A.light_dir[i];
        if (L > 0) { const double s = (b.hi[i] + slack - (scene_lo[i] - slack)) / L; s_max = s < s_max ? s : s_max; }
        else if (L < 0) { const double s = (scene_hi[i] + slack - (b.lo[i] - slack)) / -L; s_max = s < s_max ? s : s_max; }
    }
    if (!(s_max < 1e9)) return;                     // the light direction has no component the scene's bounds could stop
    if (!(s_max > 0)) s_max = 0;
    double pts[16][3];
    for (int c = 0; c < 16; c++)
        for (int i = 0; i < 3; i++) pts[c][i] = b.corner[c & 7][i] - (c < 8 ? 0.0 : s_max) * (double)A.light_dir[i];
    (void)tiles_around(pts, 16, slack, A, rect);
}

// ---- hierarchy over placed instances (never used for results, only to propose instances) ----------
// World-space box of an instance, padded so that float rounding in the world-space slab test can never
// reject a ray that the exact test in the instance's own space would accept: 1e-4 of the box's extent
// plus 1e-4 absolute on every side, then rounded outwards to float.
struct InstanceBox { float lo[3], hi[3]; };

inline InstanceBox instance_box(const rt_transform& tf, const float* bmin, const float* bmax) {
    const WorldBox w = world_box(tf, bmin, bmax);
    InstanceBox b;
    for (int i = 0; i < 3; i++) {
        const double pad = 1e-4 * (w.hi[i] - w.lo[i]) + 1e-4 + 1e-6 * (fabs(w.lo[i]) + fabs(w.hi[i]));
        b.lo[i] = nextafterf((float)(w.lo[i] - pad), -INFINITY);
        b.hi[i] = nextafterf((float)(w.hi[i] + pad), INFINITY);
    }
    return b;
}

// The box of an instance of a mesh with MeshArgs::wide reaches as far as the camera: every camera ray then starts inside it
// (and inside every box of the hierarchy above it), enters it at a distance of zero or less, and no walk over the hierarchy
// drops it for lying behind the closest hit so far -- the instance is proposed to every camera ray and gets the reference's
// own root test.  (Shadow rays are not culled by distance in the first place.)
inline void reach_camera(InstanceBox& b, const float* cam) {
    for (int i = 0; i < 3; i++) {
        if (!std::isfinite(cam[i])) continue;
        const float pad = 1e-3f * (1.0f + fabsf(cam[i]));
        if (cam[i] - pad < b.lo[i]) b.lo[i] = cam[i] - pad;
        if (cam[i] + pad > b.hi[i]) b.hi[i] = cam[i] + pad;
    }
}

// What the padding above leaves out: the exact test in an instance's own space works on the ray's origin moved into that
// space, whose rounding error goes with the magnitudes involved -- the instance's position, but also the origin itself: the
// camera, or, for a shadow ray, a surface point anywhere in the scene.  With an entry placed at 1e30 (the reference accepts any
// placement) those errors are of the order of 1e23, the exact test then accepts or rejects by rounding alone, and rays that start
// on that entry do the same against every other one.  So every box is widened once more, on all sides, by 2^-18 of the
// largest finite coordinate magnitude of the scene and the camera (64 units in the last place, the allowance of
// mesh_screen_rect), rounded up to a power of two so that it changes rarely and a kept hierarchy stays valid from frame to
// frame.  An ordinary scene (coordinates of tens) gets 5e-4.
inline float scene_box_allowance(const std::vector<InstanceBox>& boxes, const float* cam_pos) {
    double mag = 0.0;
    auto take = [&](float v) { const double a = fabs((double)v); if (a <= 3.0e38 && a > mag) mag = a; };       // false for NaN and infinity
    for (int i = 0; i < 3; i++) take(cam_pos[i]);
    for (const InstanceBox& b : boxes)
        for (int i = 0; i < 3; i++) { take(b.lo[i]); take(b.hi[i]); }
    if (!(mag > 0.0)) return 0.0f;
    int e = 0;
    (void)frexp(mag, &e);                                      // mag < 2^e
    return (float)ldexp(1.0, e - 18);                          // overflows to infinity only beyond 2^127 * 2^18: never
}
// The allowance to use given the one used last time (0 at first): kept while it is at least what the scene needs now and at
// most four times that, so that entries moving about a scene do not change every box each time the largest coordinate
// crosses a power of two.
inline float settled_allowance(float kept, float need) { return (kept >= need && kept <= 4.0f * need) ? kept : need; }
inline void widen_boxes(std::vector<InstanceBox>& boxes, float by) {
    if (!(by > 0.0f)) return;
    for (InstanceBox& b : boxes)
        for (int i = 0; i < 3; i++) {
            b.lo[i] = nextafterf(b.lo[i] - by, -INFINITY);
            b.hi[i] = nextafterf(b.hi[i] + by, INFINITY);
        }
}

struct Tlas {
    std::vector<PairNode> pairs;
    std::vector<uint32_t> order;      // instance indices, leaf after leaf
    uint32_t root_ref = kLeafBit;     // empty leaf
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    int depth = 0;
};

constexpr size_t kTlasLeaf = 1;      // instances per leaf of the hierarchy

// The hierarchy as the traversal reads it: every leaf names its instance itself, leaf_ref(instance index, 1), instead of a
// position in `order` -- one dependent memory read less per proposed instance and ray.  (One instance per leaf; an emptied
// leaf stays leaf_ref(0, 0).)
inline uint32_t direct_ref(const Tlas& T, uint32_t ref) {
    static_assert(kTlasLeaf == 1, "a leaf reference holds one instance index");
    if (!(ref & kLeafBit)) return ref;
    const uint32_t first = ref & 0x00FFFFFFu, count = (ref >> 24) & 0x7Fu;
    return count ? leaf_ref(T.order[first], 1) : leaf_ref(0, 0);
}
inline void write_direct(const Tlas& T, PairNode* out) {
    for (size_t i = 0; i < T.pairs.size(); i++) {
        out[i] = T.pairs[i];
        out[i].lref = direct_ref(T, T.pairs[i].lref);
        out[i].rref = direct_ref(T, T.pairs[i].rref);
    }
}

class TlasBuilder {
public:
    // boxes[k] belongs to instance ids[k]; binned surface-area split of the centroids (median split along the
    // longest axis where that finds nothing, or for the whole tree if it comes out too deep), leaves of at most
    // two instances.
    TlasBuilder(const std::vector<InstanceBox>& boxes, const std::vector<uint32_t>& ids, Tlas& out) : boxes_(boxes), ids_(ids), out_(out) {}

    void run() {
        build(true);
        if (out_.depth + 2 > kTlasStackDepth) build(false);      // a lopsided surface-area tree: the median split is balanced
    }

private:
    void build(bool by_area) {
        by_area_ = by_area;
        out_ = Tlas();
        std::vector<uint32_t> items(boxes_.size());
        for (size_t i = 0; i < items.size(); i++) items[i] = (uint32_t)i;
        if (items.empty()) return;
        out_.order.reserve(items.size());
        InstanceBox root;
        out_.root_ref = emit(items.data(), items.size(), 0, root);
        memcpy(out_.root_min, root.lo, 12); memcpy(out_.root_max, root.hi, 12);
    }

    uint32_t emit(uint32_t* items, size_t n, int depth, InstanceBox& box) {
        for (int i = 0; i < 3; i++) { box.lo[i] = INFINITY; box.hi[i] = -INFINITY; }
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = 0; k < n; k++) {
            const InstanceBox& b = boxes_[items[k]];
            for (int i = 0; i < 3; i++) {
                box.lo[i] = b.lo[i] < box.lo[i] ? b.lo[i] : box.lo[i];
                box.hi[i] = b.hi[i] > box.hi[i] ? b.hi[i] : box.hi[i];
                const float c = 0.5f * b.lo[i] + 0.5f * b.hi[i];
                clo[i] = c < clo[i] ? c : clo[i]; chi[i] = c > chi[i] ? c : chi[i];
            }
        }
        if (n <= kTlasLeaf) {
            const uint32_t first = (uint32_t)out_.order.size();
            for (size_t k = 0; k < n; k++) out_.order.push_back(ids_[items[k]]);
            return leaf_ref(first, (uint32_t)n);
        }
        // Binned surface-area split over the centroids (a huge instance, e.g. a ground plane, ends up alone near
        // the root instead of inflating every box above it); median split along the longest axis when no bin
        // boundary separates the items.
        size_t mid = n / 2;
        {
            constexpr int kBins = 16;
            float best = INFINITY;
            int best_axis = -1, best_split = 0;
            for (int a = 0; a < 3 && by_area_; a++) {
                const float extent = chi[a] - clo[a];
                if (!(extent > 0.0f)) continue;
                InstanceBox bb[kBins];
                size_t bn[kBins] = {};
                for (int b = 0; b < kBins; b++) for (int i = 0; i < 3; i++) { bb[b].lo[i] = INFINITY; bb[b].hi[i] = -INFINITY; }
                auto bin_of = [&](uint32_t it) {
                    const float c = 0.5f * boxes_[it].lo[a] + 0.5f * boxes_[it].hi[a];
                    int b = (int)((c - clo[a]) / extent * (float)kBins);
                    return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                };
                for (size_t k = 0; k < n; k++) {
                    const int b = bin_of(items[k]);
                    bn[b]++;
                    for (int i = 0; i < 3; i++) {
                        bb[b].lo[i] = boxes_[items[k]].lo[i] < bb[b].lo[i] ? boxes_[items[k]].lo[i] : bb[b].lo[i];
                        bb[b].hi[i] = boxes_[items[k]].hi[i] > bb[b].hi[i] ? boxes_[items[k]].hi[i] : bb[b].hi[i];
                    }
                }
                auto area = [](const InstanceBox& x) { const float dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2]; return dx * dy + dy * dz + dz * dx; };
                InstanceBox suffix[kBins];
                size_t sn[kBins];
                InstanceBox acc; size_t an = 0;
                for (int i = 0; i < 3; i++) { acc.lo[i] = INFINITY; acc.hi[i] = -INFINITY; }
                for (int b = kBins - 1; b >= 0; b--) {
                    for (int i = 0; i < 3; i++) { acc.lo[i] = bb[b].lo[i] < acc.lo[i] ? bb[b].lo[i] : acc.lo[i]; acc.hi[i] = bb[b].hi[i] > acc.hi[i] ? bb[b].hi[i] : acc.hi[i]; }
                    an += bn[b]; suffix[b] = acc; sn[b] = an;
                }
                for (int i = 0; i < 3; i++) { acc.lo[i] = INFINITY; acc.hi[i] = -INFINITY; }
                an = 0;
                for (int split = 1; split < kBins; split++) {
                    const int b = split - 1;
                    for (int i = 0; i < 3; i++) { acc.lo[i] = bb[b].lo[i] < acc.lo[i] ? bb[b].lo[i] : acc.lo[i]; acc.hi[i] = bb[b].hi[i] > acc.hi[i] ? bb[b].hi[i] : acc.hi[i]; }
                    an += bn[b];
                    if (an == 0 || sn[split] == 0) continue;
                    const float cost = area(acc) * (float)an + area(suffix[split]) * (float)sn[split];
                    if (cost < best) { best = cost; best_axis = a; best_split = split; }
                }
            }
            if (best_axis >= 0) {
                const int a = best_axis;
                const float extent = chi[a] - clo[a];
                uint32_t* m = std::partition(items, items + n, [&](uint32_t it) {
                    const float c = 0.5f * boxes_[it].lo[a] + 0.5f * boxes_[it].hi[a];
                    int b = (int)((c - clo[a]) / extent * (float)kBins);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    return b < best_split;
                });
                mid = (size_t)(m - items);
            } else {
                int axis = 0;
                if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
                if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
                std::nth_element(items, items + mid, items + n, [&](uint32_t a, uint32_t b) {
                    const float ca = boxes_[a].lo[axis] + boxes_[a].hi[axis], cb = boxes_[b].lo[axis] + boxes_[b].hi[axis];
                    return ca < cb || (ca == cb && a < b);
                });
            }
        }
        const uint32_t me = (uint32_t)out_.pairs.size();
        out_.pairs.emplace_back();
        if (depth + 1 > out_.depth) out_.depth = depth + 1;
        InstanceBox lb, rb;
        const uint32_t l = emit(items, mid, depth + 1, lb);
        const uint32_t r = emit(items + mid, n - mid, depth + 1, rb);
        PairNode& p = out_.pairs[me];
        memcpy(p.lmin, lb.lo, 12); memcpy(p.lmax, lb.hi, 12);
        memcpy(p.rmin, rb.lo, 12); memcpy(p.rmax, rb.hi, 12);
        p.lref = l; p.rref = r; p.pad[0] = p.pad[1] = 0;
        return me;
    }

    const std::vector<InstanceBox>& boxes_;
    const std::vector<uint32_t>& ids_;
    Tlas& out_;
    bool by_area_ = true;
};

// The hierarchy of a scene that is rendered frame after frame with most instances standing still.
//
// A hierarchy built once is kept.  For the next frame the instances whose box changed (moved, swapped
// for another mesh, emptied or filled) are counted; if they are few, the kept tree is copied, their
// leaves in it are emptied (the boxes above stay as they are: larger than needed, never too small),
// a small tree is built over them where they are now, and a new root joins the two.  Any hierarchy
// whose boxes contain the instances below them proposes every instance a ray can reach, so the
// rendered frame is the same; only the walk's cost depends on the tree.  The kept tree is the one
// from the last full build, so patches do not pile up; when many instances changed, or the list's
// length did, the tree is built anew and kept.
class StandingTlas {
public:
    // boxes[k] belongs to instance ids[k] (the non-empty instances, ids ascending) of a list of n_instances.
    // Fills `out` with this frame's hierarchy.  Returns true when the kept tree was reused.
    bool update(const std::vector<InstanceBox>& boxes, const std::vector<uint32_t>& ids, size_t n_instances, Tlas& out) {
        static_assert(kTlasLeaf == 1, "emptying a leaf removes exactly one instance");
        has_now_.assign(n_instances, 0);
        where_now_.assign(n_instances, 0);
        for (size_t k = 0; k < ids.size(); k++) { has_now_[ids[k]] = 1; where_now_[ids[k]] = (uint32_t)k; }
        if (valid_ && kept_has_.size() == n_instances && !kept_.pairs.empty()) {
            const size_t limit = n_instances / 16 > 8 ? n_instances / 16 : 8;
            std::vector<uint32_t> changed;
            for (size_t i = 0; i < n_instances && changed.size() <= limit; i++) {
                const bool same = has_now_[i] == kept_has_[i] && (!has_now_[i] || memcmp(&boxes[where_now_[i]], &kept_box_[i], sizeof(InstanceBox)) == 0);
                if (!same) changed.push_back((uint32_t)i);
            }
            if (changed.size() <= limit && patch(boxes, ids, changed, out)) return true;
        }
        TlasBuilder(boxes, ids, kept_).run();
        kept_has_ = has_now_;
        kept_box_.assign(n_instances, InstanceBox());
        for (size_t k = 0; k < ids.size(); k++) kept_box_[ids[k]] = boxes[k];
        slot_.assign(n_instances, kNoSlot);
        for (size_t p = 0; p < kept_.pairs.size(); p++)
            for (uint32_t side = 0; side < 2; side++) {
                const uint32_t ref = side ? kept_.pairs[p].rref : kept_.pairs[p].lref;
                if (!(ref & kLeafBit)) continue;
                const uint32_t first = ref & 0x00FFFFFFu, count = (ref >> 24) & 0x7Fu;
                for (uint32_t k = 0; k < count; k++) slot_[kept_.order[first + k]] = (uint32_t)p * 2u + side;
            }
        valid_ = true;
        out = kept_;
        return false;
    }

    void clear() { valid_ = false; }

private:
    static constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

    bool patch(const std::vector<InstanceBox>& boxes, const std::vector<uint32_t>& ids, const std::vector<uint32_t>& changed, Tlas& out) {
        out = kept_;
        if (changed.empty()) return true;
        std::vector<InstanceBox> moved_boxes;
        std::vector<uint32_t> moved_ids;
        for (uint32_t i : changed) {
            if (kept_has_[i]) {
                if (slot_[i] == kNoSlot) return false;
                PairNode& p = out.pairs[slot_[i] >> 1];
                ((slot_[i] & 1u) ? p.rref : p.lref) = leaf_ref(0, 0);          // a leaf without instances
            }
            if (has_now_[i]) { moved_boxes.push_back(boxes[where_now_[i]]); moved_ids.push_back(ids[where_now_[i]]); }
        }
        if (moved_ids.empty()) return true;
        Tlas small;
        TlasBuilder(moved_boxes, moved_ids, small).run();
        const uint32_t pair_shift = (uint32_t)out.pairs.size(), order_shift = (uint32_t)out.order.size();
        if ((size_t)order_shift + small.order.size() > 0x00FFFFFFu) return false;
        auto shifted = [&](uint32_t ref) { return (ref & kLeafBit) ? ref + order_shift : ref + pair_shift; };
        for (PairNode p : small.pairs) { p.lref = shifted(p.lref); p.rref = shifted(p.rref); out.pairs.push_back(p); }
        out.order.insert(out.order.end(), small.order.begin(), small.order.end());
        PairNode join;
        memcpy(join.lmin, kept_.root_min, 12); memcpy(join.lmax, kept_.root_max, 12);
        memcpy(join.rmin, small.root_min, 12); memcpy(join.rmax, small.root_max, 12);
        join.lref = kept_.root_ref; join.rref = shifted(small.root_ref); join.pad[0] = join.pad[1] = 0;
        out.root_ref = (uint32_t)out.pairs.size();
        out.pairs.push_back(join);
        for (int i = 0; i < 3; i++) {
            out.root_min[i] = small.root_min[i] < kept_.root_min[i] ? small.root_min[i] : kept_.root_min[i];
            out.root_max[i] = small.root_max[i] > kept_.root_max[i] ? small.root_max[i] : kept_.root_max[i];
        }
        out.depth = (small.depth > kept_.depth ? small.depth : kept_.depth) + 1;
        return out.depth + 2 <= kTlasStackDepth;
    }

    Tlas kept_;
    std::vector<InstanceBox> kept_box_;     // per instance of the list the kept tree was built for
    std::vector<uint8_t> kept_has_;         // ... whether it is in the tree (not an empty mesh)
    std::vector<uint32_t> slot_;            // ... and where its leaf hangs: PairNode index * 2 + side
    std::vector<uint8_t> has_now_;
    std::vector<uint32_t> where_now_;
    bool valid_ = false;
};

}  // namespace rt
