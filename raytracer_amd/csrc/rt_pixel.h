// rt_pixel.h -- one pixel of the frame loop (reference scene.c:58-158), written once for the
// device kernels and for a host build that the CPU tests use to check the packed scene and the
// traversal logic without a GPU.
//
// Equivalence with the reference's recursive, un-culled traversal (accel/bvh.c:80-129):
//  * The reference visits exactly the nodes whose own box and all ancestors' boxes pass the slab
//    test, in any order the result is the minimum-t triangle among the visited leaves.
//  * Equal t: inside a leaf the lowest index wins (strict '<', bvh.c:95); between siblings the
//    right child wins (bvh.c:113-124), i.e. the later leaf in index order.  The packer gives every
//    triangle a priority that encodes exactly that order, so "t smaller, or t equal and priority
//    higher" picks the reference's winner regardless of visiting order -- and regardless of how
//    the visited set is divided among cooperating lanes.
//  * Across meshes an earlier mesh wins ties (strict '<', scene.c:82).
//  * The shadow pass only uses "any triangle accepted" (scene.c:133-138), so it may stop at the
//    first accepted triangle.
#pragma once
#include "rt_exact.h"
#include "rt_layout.h"

#ifndef RT_UNROLL
#if defined(__HIPCC__) || defined(__clang__)
#define RT_UNROLL _Pragma("unroll")
#else
#define RT_UNROLL _Pragma("GCC unroll 8")
#endif
#endif

namespace rt {

struct Hit {
    float t, u, v;
    uint32_t tri;
    uint32_t prio;
};
RT_HD Hit no_hit() { Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0; h.prio = 0; return h; }

struct Work {
    uint32_t slabs = 0, tris = 0;
    uint32_t idle = 0;           // counting runs: shadow rays sent although their result cannot change the colour (max(0.2, n.L) = 0.2)
    // Filled by the self-checking party only (SoloVerify): shortcut decisions that were examined again the reference's way,
    // and how many of them the reference's way contradicts, by kind.
    uint32_t checks = 0;
    uint32_t bad_vis = 0;        // a camera ray passes the root box of a mesh that its tile's screen rectangle leaves out
    uint32_t bad_shadow_vis = 0; // the same for a shadow ray and the shadow rectangle
    uint32_t bad_entry = 0;      // a walk entered through a flat mesh's grid ends differently from the walk from the tree's root
    uint32_t bad_cull = 0;       // the ordered, distance-culled walk (second tree included) ends differently from the reference's own
    uint32_t bad_inst = 0;       // the walk over the instance hierarchy ends differently from the reference's loop over all instances
};

RT_HD float uint_as_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
RT_HD uint32_t f32_as_uint(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
// Reads through the pointers a MeshArgs carries (triangles, attributes, texels, grid cells).  Device code reads such a
// pointer from memory first, so the compiler cannot know which memory it points to and would use FLAT accesses; those
// count against both of a wave's memory counters, and every wait for an LDS read (the traversal stack) would then wait
// for them as well.  The arrays live in global memory: the device build says so.  Host build: plain reads.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_GLOBAL_AS __attribute__((address_space(1)))
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class V>
__device__ __forceinline__ V global_read(const void* base, size_t byte_offset) {
    return *(const RT_GLOBAL_AS V*)((const RT_GLOBAL_AS char*)base + byte_offset);
}
__device__ __forceinline__ TriRec read_tri(const TriRec* tris, uint32_t index) {
    const size_t at = (size_t)index * sizeof(TriRec);
    const f32x4 a = global_read<f32x4>(tris, at);
#if defined(RT_FAST)
    const f32x4 b = global_read<f32x4>(tris, at + 16);                     // with the triangle's id
#else
    const f32x3 b = global_read<f32x3>(tris, at + 16);
#endif
    const f32x3 c = global_read<f32x3>(tris, at + 32);
    TriRec t;
    t.v0[0] = a.x; t.v0[1] = a.y; t.v0[2] = a.z; t.prio = __float_as_uint(a.w);
#if defined(RT_FAST)
    t.e1[0] = b.x; t.e1[1] = b.y; t.e1[2] = b.z; t.pad1 = b.w;
#else
    t.e1[0] = b.x; t.e1[1] = b.y; t.e1[2] = b.z; t.pad1 = 0.0f;
#endif
    t.e2[0] = c.x; t.e2[1] = c.y; t.e2[2] = c.z; t.pad2 = 0.0f;
    return t;
}
__device__ __forceinline__ AttrRec read_attrs(const AttrRec* attrs, uint32_t index) {
    const size_t at = (size_t)index * sizeof(AttrRec);
    const f32x4 a = global_read<f32x4>(attrs, at), b = global_read<f32x4>(attrs, at + 16), c = global_read<f32x4>(attrs, at + 32);
    const f32x3 d = global_read<f32x3>(attrs, at + 48);
    AttrRec r;
    r.t0[0] = a.x; r.t0[1] = a.y; r.t1[0] = a.z; r.t1[1] = a.w; r.t2[0] = b.x; r.t2[1] = b.y;
    r.n0[0] = b.z; r.n0[1] = b.w; r.n0[2] = c.x; r.n1[0] = c.y; r.n1[1] = c.z; r.n1[2] = c.w;
    r.n2[0] = d.x; r.n2[1] = d.y; r.n2[2] = d.z; r.pad = 0.0f;
    return r;
}
__device__ __forceinline__ uint32_t read_word(const uint32_t* words, size_t index) { return global_read<uint32_t>(words, index * 4); }
#else
RT_HD TriRec read_tri(const TriRec* tris, uint32_t index) { return tris[index]; }
RT_HD AttrRec read_attrs(const AttrRec* attrs, uint32_t index) { return attrs[index]; }
RT_HD uint32_t read_word(const uint32_t* words, size_t index) { return words[index]; }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// A node that every active lane of the wave is at: one read through the scalar cache into scalar registers instead of
// sixty-four through the vector memory path.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ PairNode read_pair_uniform(const PairNode* pairs, uint32_t index) {
    const u32x16 w = *(const __attribute__((address_space(4))) u32x16*)((const __attribute__((address_space(4))) char*)pairs + (size_t)index * sizeof(PairNode));
    PairNode n;
    n.lmin[0] = __uint_as_float(w[0]); n.lmin[1] = __uint_as_float(w[1]); n.lmin[2] = __uint_as_float(w[2]);
    n.lmax[0] = __uint_as_float(w[3]); n.lmax[1] = __uint_as_float(w[4]); n.lmax[2] = __uint_as_float(w[5]);
    n.rmin[0] = __uint_as_float(w[6]); n.rmin[1] = __uint_as_float(w[7]); n.rmin[2] = __uint_as_float(w[8]);
    n.rmax[0] = __uint_as_float(w[9]); n.rmax[1] = __uint_as_float(w[10]); n.rmax[2] = __uint_as_float(w[11]);
    n.lref = w[12]; n.rref = w[13]; n.pad[0] = n.pad[1] = 0;
    return n;
}
template <class Nodes> struct ScalarNodes { static constexpr bool on = false; __device__ static const PairNode* base(const Nodes&) { return nullptr; } };
#endif

// Node source: where PairNodes are read from.  GlobalNodes reads the scene's combined node array
// through the vector cache; the kernels also have a source that reads a copy staged in LDS.
struct GlobalNodes {
    const PairNode* all;
    RT_HD const PairNode& operator()(const MeshArgs& M, uint32_t ref) const { return all[M.pair_base + ref]; }
    RT_HD TriRec tri(const MeshArgs& M, uint32_t index) const { return read_tri(M.tris, index); }
};

// The same source for a kernel that reads its frame constants from memory (rt_render_frames): its walks may use the per-frame
// part of a sheared tree's gate (shear_gate).
struct BatchNodes : GlobalNodes {};
template <class Nodes> struct FrameMasks { static constexpr bool on = false; };
template <> struct FrameMasks<BatchNodes> { static constexpr bool on = true; };

#if defined(__HIP_DEVICE_COMPILE__)
template <> struct ScalarNodes<GlobalNodes> { static constexpr bool on = true; __device__ static const PairNode* base(const GlobalNodes& n) { return n.all; } };
template <> struct ScalarNodes<BatchNodes> { static constexpr bool on = true; __device__ static const PairNode* base(const BatchNodes& n) { return n.all; } };
#endif

// Distance culling (used unless the caller asks for the reference's visiting order).
// A subtree is skipped when its box is entered farther away than the closest accepted hit.  Both
// distances carry rounding error (the slab entry is within a few ulp of the true one; Moeller-
// Trumbore's t is not as tight on slivers), so a box is only skipped when it starts beyond
// best*(1 + 2^-9) + 2^-13: a triangle inside it would need a t that is wrong by 0.2 % to have won.
// Ties (equal t) lie well inside the margin and are still visited, so the tie rule is unaffected.
RT_HD float cull_limit(float best) { return best * 1.001953125f + 0.0001220703125f; }
// The limit of a query that skips nothing by distance (MeshArgs::wide): the margin above covers a triangle test whose distance
// is off by up to 0.2 %.  That holds for rays that meet a triangle's plane at more than about 2e-4 rad; ray.c:50 only turns away
// rays with |a| < 1e-7, and for a triangle with edges of 0.5 or more |a| = 2 area |cos| stays above that down to 1e-6 rad and
// less, where the computed distance (and u, v) are off by percents.  The reference accepts such a hit wherever it falls.
constexpr float kNoCull = 3.0e38f;

// True when hit a (if found) is the one the reference would keep over hit b.
RT_HD bool beats(bool a_found, const Hit& a, bool b_found, const Hit& b) {
    return a_found && (!b_found || a.t < b.t || (a.t == b.t && a.prio > b.prio));
}

// Triangles fetched together in a leaf.  Two at a time halve the memory round trips of a leaf, but hold ten more registers
// and cost a wave of occupancy; with the kernels at 59-90 registers one at a time is faster (render_batch: 0.979 -> 0.935 ms).
constexpr uint32_t kLeafBatch = 1;

// Slab tests of both children of a PairNode (aabb.c:33-56 twice), entry distances reported.
#if defined(__HIP_DEVICE_COMPILE__)
// Device form: the 24 subtractions and multiplications run as 12 two-wide packed operations
// (v_pk_add_f32 / v_pk_mul_f32 round each half exactly like the scalar instructions).  The twelve
// bounds are taken as six register pairs in memory order -- (lmin.x,lmin.y) (lmin.z,lmax.x)
// (lmax.y,lmax.z) and the same for the right box -- against matching pairs of the ray's origin and
// reciprocal direction.
typedef float v2f __attribute__((ext_vector_type(2)));
#if defined(RT_FAST)
// Fast build (not bit-exact): (bound - o) * inv becomes bound * inv - o * inv with o * inv computed once per ray and
// mesh, one packed fused multiply-add per pair of bounds.  Where a direction component is zero (inv = inf) the
// products turn into NaNs that fminf/fmaxf drop, which only ever lets MORE boxes through than the exact form.
struct SlabRay {
    v2f ixy, izx, iyz, pxy, pzx, pyz;            // reciprocal direction pairs, origin * reciprocal direction pairs
};
RT_HD SlabRay slab_ray(F3 o, F3 inv) {
    SlabRay r;
    r.ixy = v2f{inv.x, inv.y}; r.izx = v2f{inv.z, inv.x}; r.iyz = v2f{inv.y, inv.z};
    r.pxy = v2f{o.x, o.y} * r.ixy; r.pzx = v2f{o.z, o.x} * r.izx; r.pyz = v2f{o.y, o.z} * r.iyz;
    return r;
}
RT_HD bool slab_half(v2f p0, v2f p1, v2f p2, const SlabRay& r, float& entry) {
    const v2f a0 = p0 * r.ixy - r.pxy;          // min.x, min.y
    const v2f a1 = p1 * r.izx - r.pzx;          // min.z, max.x
    const v2f a2 = p2 * r.iyz - r.pyz;          // max.y, max.z
#else
struct SlabRay {
    v2f oxy, ozx, oyz, ixy, izx, iyz;
};
RT_HD SlabRay slab_ray(F3 o, F3 inv) {
    SlabRay r;
    r.oxy = v2f{o.x, o.y}; r.ozx = v2f{o.z, o.x}; r.oyz = v2f{o.y, o.z};
    r.ixy = v2f{inv.x, inv.y}; r.izx = v2f{inv.z, inv.x}; r.iyz = v2f{inv.y, inv.z};
    return r;
}
RT_HD bool slab_half(v2f p0, v2f p1, v2f p2, const SlabRay& r, float& entry) {
    const v2f a0 = (p0 - r.oxy) * r.ixy;        // min.x, min.y
    const v2f a1 = (p1 - r.ozx) * r.izx;        // min.z, max.x
    const v2f a2 = (p2 - r.oyz) * r.iyz;        // max.y, max.z
#endif
    float tmin = fminf(a0.x, a1.y), tmax = fmaxf(a0.x, a1.y);
    tmin = fmaxf(tmin, fminf(a0.y, a2.x)); tmax = fminf(tmax, fmaxf(a0.y, a2.x));
    tmin = fmaxf(tmin, fminf(a1.x, a2.y)); tmax = fminf(tmax, fmaxf(a1.x, a2.y));
    entry = tmin;
    return tmax >= tmin && tmax > 0;
}
RT_HD void slab_pair(const SlabRay& r, const PairNode& N, float& tl, float& tr, bool& hl, bool& hr) {
    hl = slab_half(v2f{N.lmin[0], N.lmin[1]}, v2f{N.lmin[2], N.lmax[0]}, v2f{N.lmax[1], N.lmax[2]}, r, tl);
    hr = slab_half(v2f{N.rmin[0], N.rmin[1]}, v2f{N.rmin[2], N.rmax[0]}, v2f{N.rmax[1], N.rmax[2]}, r, tr);
}
#else
struct SlabRay {
    F3 o, inv;
};
RT_HD SlabRay slab_ray(F3 o, F3 inv) { SlabRay r; r.o = o; r.inv = inv; return r; }
RT_HD void slab_pair(const SlabRay& r, const PairNode& N, float& tl, float& tr, bool& hl, bool& hr) {
    hl = slab_t(r.o, r.inv, N.lmin, N.lmax, tl);
    hr = slab_t(r.o, r.inv, N.rmin, N.rmax, tr);
}
#endif

// State of one closest-hit / any-hit query while it runs.
struct Query {
    float best_t;        // bvh.c:85 / bvh.c:106: bound of the leaf tests
    float bound;         // the caller's closest_t (scene.c:74)
    float limit;         // culling distance (only used when culling)
    bool found;
    Hit hit;
};

// Tests the triangles of one leaf (bvh.c:89-103).  Returns true when an any-hit query may stop.
// Triangles are fetched LB at a time before any of them is tested (one memory round trip per batch, at ten registers per
// triangle held).
// `mask` (leaves of at most 32 triangles): only the triangles whose bit is set -- the others are known to reject this ray
// (the lists of a flat mesh's grid, Packer::tri_mask).
template <bool ANY, bool COUNT, bool FULL, uint32_t LB, class Nodes>
RT_HD bool test_leaf(const MeshArgs& M, const Nodes& nodes, uint32_t ref, F3 o, F3 d, Query& q, Work& work, const uint32_t mask = 0xFFFFFFFFu) {
    constexpr bool CULL = !FULL && !ANY;
    const uint32_t first = ref & 0x00FFFFFFu, count = (ref >> 24) & 0x7Fu;
    for (uint32_t i0 = 0; i0 < count; i0 += LB) {
        if (LB == 1 && mask != 0xFFFFFFFFu && !((mask >> (i0 & 31u)) & 1u)) continue;
        TriRec T[LB];                                          // registers: every index below is a constant
        const uint32_t n = count - i0 < LB ? count - i0 : LB;
        RT_UNROLL
        for (uint32_t k = 0; k < LB; k++) T[k] = nodes.tri(M, first + i0 + (k < n ? k : 0));
        RT_UNROLL
        for (uint32_t k = 0; k < LB; k++) {
            if (k >= n) continue;
            if (LB > 1 && mask != 0xFFFFFFFFu && !((mask >> ((i0 + k) & 31u)) & 1u)) continue;
            float t, u, v;
            if (COUNT) work.tris++;
            if (!moller_trumbore(o, d, f3(T[k].v0), f3(T[k].e1), f3(T[k].e2), t, u, v)) continue;
            if (ANY) {
                if (t < q.best_t) {
                    q.found = true;
                    if (!FULL) return true;
                }
            } else if (t < q.best_t || (q.found && t == q.best_t && T[k].prio > q.hit.prio)) {
                q.best_t = t; q.found = true;
                q.hit.t = t; q.hit.u = u; q.hit.v = v; q.hit.prio = T[k].prio;
#if defined(RT_FAST) || !defined(__HIP_DEVICE_COMPILE__)
                q.hit.tri = f32_as_uint(T[k].pad1);          // the record names its triangle: the tree walked may keep its records in its own order
#else
                q.hit.tri = first + i0 + k;
#endif
                if (CULL && t < q.bound && q.limit < kNoCull) q.limit = cull_limit(t);
            }
        }
    }
    return false;
}

// Start of a query against one mesh: the root box test (bvh.c:82 at the root) and the bounds.
// Returns false when the traversal has nothing to visit.
template <bool ANY, bool COUNT, bool FULL>
RT_HD bool begin_query(const MeshArgs& M, F3 o, F3 inv, float bound, Query& q, Work& work) {
    constexpr bool CULL = !FULL && !ANY;
    if (COUNT) work.slabs++;
    float entry;
    if (!slab_t(o, inv, M.root_min, M.root_max, entry)) return false;
    if (M.empty) return false;
    // `bound` is the t the caller passes in: the reference only honours it inside a root that is
    // a leaf (bvh.c:85); otherwise leaves start from 1e30f (bvh.c:106).  The caller accepts the
    // result only if it is below `bound` (scene.c:82), so hits at or beyond it need not be found.
    q.best_t = (M.root_ref & kLeafBit) ? bound : 1e30f;
    q.bound = bound;
    q.limit = M.wide ? kNoCull : cull_limit(bound);
    q.found = false;
    q.hit = no_hit();
    return !(CULL && entry > q.limit);
}

// Where a ray enters a mesh's nodes.  The regrouped tree leads to the leaves the reference reaches as long as
// no slab product is 0 * inf (see Packer::regroup): a ray with a direction component of exactly zero, and the
// reference-order traversal, start in the reference's own tree.
template <bool FULL>
RT_HD uint32_t entry_ref(const MeshArgs& M, F3 d) {
    if (FULL) return M.ref_root;
    return (d.x != 0.0f && d.y != 0.0f && d.z != 0.0f) ? M.root_ref : M.ref_root;
}

RT_HD float comp(F3 v, uint32_t axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// The sheared frame of MeshArgs::grid_axis = 4 + k (axis k kept; a = (k + 1) % 3, b = (k + 2) % 3 become p_a + p_b and p_a - p_b).
// Linear, so a ray o + t d maps to shear(o) + t shear(d) with the same t.
RT_HD F3 shear(F3 p, uint32_t k) {
    if (k == 0) return f3(p.x, p.y + p.z, p.y - p.z);
    if (k == 1) return f3(p.z - p.x, p.y, p.z + p.x);
    return f3(p.x + p.y, p.x - p.y, p.z);
}

// Four words of a MeshArgs::grid that every lane reads at the same place (header and directions of a sheared tree).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ f32x4 read_uniform4(const uint32_t* base, uint32_t word) {
    return *(const __attribute__((address_space(4))) f32x4*)((const __attribute__((address_space(4))) char*)base + (size_t)word * 4);
}
#else
struct f32x4 { float x, y, z, w; };
RT_HD f32x4 read_uniform4(const uint32_t* base, uint32_t word) { f32x4 r; memcpy(&r, base + word, 16); return r; }
#endif

// First record of the leaf lists in MeshArgs::grid (in words): behind the cells of a flat mesh's grid, or where the header of a
// sheared tree says.
RT_HD size_t list_base(const MeshArgs& M) {
    if (M.grid_axis < 3u) return (((size_t)M.grid_n * (size_t)M.grid_n) + 7) & ~(size_t)7;
    return (size_t)f32_as_uint(read_uniform4(M.grid, 8).y);
}

RT_HD int bits_set(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(v);
#else
    return __builtin_popcountll(v);
#endif
}
RT_HD int lowest_bit(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)v) - 1;
#else
    return __builtin_ctzll(v);
#endif
}

// Whether a ray may walk a mesh's sheared tree (Packer::build_shear states what the tree promises and why these conditions
// make it hold): origin within reach and outside the widened hull, direction of unit length.  `close` gets a bit for every
// family of the mesh's triangles whose plane the direction comes closer to than the family allows: the tree's boxes say
// nothing about those triangles for this ray, and the caller tests them apart.  Every comparison is false for a NaN.
// Per frame the host has answered part of the question (shear_frame_masks in rt_pack.h): for the shadow rays' common direction
// MeshArgs::lit_d it names the close families outright, and for rays from MeshArgs::cam_o it names the families that any ray
// through the root box can come close to at all -- the others are not looked at.
// MASKS = false leaves the host's answers alone and looks at every family (kernels that get their frame constants by value
// would copy all of them to memory for the sake of these few words).
// `root` gets the tree to walk: MeshArgs::shear_root, or, for an origin inside the widened hull but no farther out than
// ShearHeader::near_reach, the tree whose boxes have room for such origins.
template <bool MASKS>
RT_HD bool shear_gate(const MeshArgs& M, F3 o, F3 d, uint64_t& close, uint32_t& root) {
    const f32x4 h0 = read_uniform4(M.grid, 0), h1 = read_uniform4(M.grid, 4), h2 = read_uniform4(M.grid, 8);
    const float reach = h0.x, origin_factor = h0.y, radius = h0.z;
    const float oo = dot(o, o), dd = dot(d, d);
    close = 0;
    root = M.shear_root;
    if (!(oo <= reach * reach) || !(fabsf(dd - 1.0f) <= 0.0009765625f)) return false;
    const float wide = origin_factor * (sqrtf(oo) + radius);
    const bool outside = o.x < h0.w - wide || o.y < h1.x - wide || o.z < h1.y - wide ||
                         o.x > h1.z + wide || o.y > h1.w + wide || o.z > h2.x + wide;
    if (!outside) {
        const f32x4 h3 = read_uniform4(M.grid, 12);
        if (!(h3.x > 0.0f && oo <= h3.x * h3.x)) return false;
        root = f32_as_uint(h2.w);
    }
    const uint32_t valid = MASKS ? f32_as_uint(M.grid_slack[0]) : 0u;
    if ((valid & 2u) && f32_as_uint(d.x) == f32_as_uint(M.lit_d[0]) && f32_as_uint(d.y) == f32_as_uint(M.lit_d[1]) &&
        f32_as_uint(d.z) == f32_as_uint(M.lit_d[2])) {
        close = (uint64_t)f32_as_uint(M.grid_inv[0]) | ((uint64_t)f32_as_uint(M.grid_inv[1]) << 32);
        return true;
    }
    uint64_t check = M.grid_n >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << M.grid_n) - 1);
    if ((valid & 1u) && f32_as_uint(o.x) == f32_as_uint(M.cam_o[0]) && f32_as_uint(o.y) == f32_as_uint(M.cam_o[1]) &&
        f32_as_uint(o.z) == f32_as_uint(M.cam_o[2]))
        check = (uint64_t)f32_as_uint(M.grid_lo[0]) | ((uint64_t)f32_as_uint(M.grid_lo[1]) << 32);
    while (check) {
        const int k = lowest_bit(check);
        check &= check - 1;
        const f32x4 n = read_uniform4(M.grid, 16u + 8u * (uint32_t)k);
        if (!(fabsf(d.x * n.x + d.y * n.y + d.z * n.z) >= n.w)) close |= (uint64_t)1 << k;
    }
    return true;
}

// Where a ray starts its walk in a flat mesh (MeshArgs::grid, built by Packer::build_grid): the node stored for the cell
// that holds the point X in which the ray meets the mesh's plane -- if the ray qualifies: all direction components finite
// and non-zero (the caller checks the zeros), X inside the grid, and the ray's own uncertainty about X (16 u of the
// magnitudes that enter it) no more than the slack the cells were built with.  Otherwise the tree's root.
RT_HD uint32_t grid_entry(const MeshArgs& M, F3 o, F3 d, F3 inv, uint32_t root, float& plane_t) {
    const uint32_t k = M.grid_axis, a = k == 2 ? 0u : k + 1u, b = k == 0 ? 2u : (k == 1 ? 0u : 1u);
    const float ty = (M.root_min[k] - comp(o, k)) * comp(inv, k);              // aabb.c:39-53 for the flat axis
    plane_t = ty;
    const float sa = ty * comp(d, a), sb = ty * comp(d, b);
    const float xa = comp(o, a) + sa, xb = comp(o, b) + sb;
    const float ea = (fabsf(comp(o, a)) + fabsf(sa) + M.grid_bmax[0]) * 9.5367431640625e-7f;      // 2^-20 = 16 u
    const float eb = (fabsf(comp(o, b)) + fabsf(sb) + M.grid_bmax[1]) * 9.5367431640625e-7f;
    const float fa = (xa - M.grid_lo[0]) * M.grid_inv[0], fb = (xb - M.grid_lo[1]) * M.grid_inv[1];
    const float n = (float)M.grid_n;
    // every comparison is false for a NaN, so a ray whose numbers overflow keeps the root
    if (!(ea <= M.grid_slack[0] && eb <= M.grid_slack[1] && fa >= 0.0f && fa < n && fb >= 0.0f && fb < n)) return root;
    return read_word(M.grid, (size_t)((uint32_t)(int)fb * (uint32_t)M.grid_n + (uint32_t)(int)fa));
}

// Closest hit (ANY=false) or any hit (ANY=true) of one mesh by ONE lane, ray already in mesh space.
// FULL=true reproduces the reference's work exactly: every box whose ancestors' slabs all pass is
// visited, and an any-hit query keeps going after its first accepted triangle.
// FULL=false visits the nearer child first, skips boxes beyond the closest hit so far, and ends an
// any-hit query at the first accepted triangle; the result is the same.
// BUDGET=true counts the nodes taken off the stack against `steps`; when it runs out the query is
// abandoned (returns false with steps < 0) and the caller hands the pixel to the cooperative pass.
// use_grid=false starts a flat mesh's walk at the tree's root as well (the self-checking party compares the two).
template <bool ANY, bool COUNT, bool FULL, bool BUDGET, uint32_t LB, class Stack, class Nodes>
RT_HD bool trace_mesh(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, float bound, Hit& hit, Stack& stack, Work& work,
                      int& steps, const bool use_grid = true) {
    constexpr bool CULL = !FULL && !ANY;
    Query q;
    if (!begin_query<ANY, COUNT, FULL>(M, o, inv, bound, q, work)) return false;
    stack.clear();
    SlabRay sr = slab_ray(o, inv);
    if (!FULL) {
        // Descend through PairNodes without touching the stack for the nearer child; only the
        // farther child is parked.  The inner loop holds nothing but box tests, the outer step
        // nothing but triangle tests, so the lanes of a wave do not drag each other through both
        // kinds of work on every step.
        uint32_t ref = entry_ref<false>(M, d);
        if (use_grid && M.grid_axis < 3u && ref == M.root_ref && d.x != 0.0f && d.y != 0.0f && d.z != 0.0f) {
            float plane_t = 0.0f;
            ref = grid_entry(M, o, d, inv, ref, plane_t);
            if (ref & kLeafBit) {
                // the cell names one triangle (or nothing): the ray passes its leaf's box exactly when it meets the mesh's plane in
                // front of its origin (Packer::inside_for_certain), so that is tested in the box test's place
                if (COUNT) work.slabs++;
                if (!(plane_t > 0.0f) || (CULL && plane_t > q.limit)) return false;
            }
        }
        uint64_t apart = 0;                  // families of triangles that a walk through the sheared tree has to test apart
        if (use_grid && M.grid_axis >= 4u && ref == M.root_ref) {
            // the tree in the sheared frame: same records, same slab arithmetic, on the sheared ray (same t); its leaves are
            // lists of one record, whose own box test is the reference's for that leaf
            uint64_t close;
            uint32_t root;
            const F3 sd = shear(d, M.grid_axis - 4u);
            if (shear_gate<FrameMasks<Nodes>::on>(M, o, d, close, root) && bits_set(close) <= 3 && sd.x != 0.0f && sd.y != 0.0f && sd.z != 0.0f) {
                sr = slab_ray(shear(o, M.grid_axis - 4u), recip(sd));
                ref = root;
                apart = close;
            }
        }
        const size_t lists_at = M.grid_axis != 3u ? list_base(M) : 0;
        // A list of leaves (a flat mesh's cell, a leaf of a sheared tree): each one's own box test (the reference reaches a leaf
        // exactly when that passes, the boxes above contain it), then its triangles.  True when an any-hit query may stop.
        // `only`: the triangles of the leaf to look at (a mask like the record's own)
        // (always inlined: a call would take the address of everything captured, the mesh's constants among them, and a kernel
        // that gets its frame constants by value would have to copy them to memory first)
        auto run_record = [&](const size_t at, const uint32_t only) __attribute__((always_inline)) -> bool {
            {
                float bmin[3], bmax[3];
#if defined(__HIP_DEVICE_COMPILE__)
                const f32x4 r0 = global_read<f32x4>(M.grid, at * 4), r1 = global_read<f32x4>(M.grid, at * 4 + 16);     // records are 32-byte aligned
                bmin[0] = r0.x; bmin[1] = r0.y; bmin[2] = r0.z; bmax[0] = r0.w; bmax[1] = r1.x; bmax[2] = r1.y;
                const uint32_t leaf = f32_as_uint(r1.z), mask = f32_as_uint(r1.w) & only;
#else
                RT_UNROLL
                for (int c = 0; c < 3; c++) { bmin[c] = uint_as_f32(read_word(M.grid, at + c)); bmax[c] = uint_as_f32(read_word(M.grid, at + 3 + c)); }
                const uint32_t leaf = read_word(M.grid, at + 6), mask = read_word(M.grid, at + 7) & only;
#endif
                if (COUNT) work.slabs++;
                float entry;
                if (!slab_t(o, inv, bmin, bmax, entry)) return false;
                if (CULL && entry > q.limit) return false;
                return test_leaf<ANY, COUNT, FULL, LB>(M, nodes, leaf, o, d, q, work, mask);
            }
        };
        auto run_list = [&](const uint32_t list) __attribute__((always_inline)) -> bool {
            const uint32_t count = (list >> 24) & 0xFu;
            const size_t first = lists_at + (size_t)(list & 0x00FFFFFFu) * 8;
            for (uint32_t i = 0; i < count; i++)
                if (run_record(first + (size_t)i * 8, 0xFFFFFFFFu)) return true;
            return false;
        };
        // the families the sheared tree cannot speak for: every leaf that holds one of their triangles, those triangles only
        while (apart) {
            const int k = lowest_bit(apart);
            apart &= apart - 1;
            const uint32_t first = read_word(M.grid, 16 + 8 * (size_t)k + 4), count = read_word(M.grid, 16 + 8 * (size_t)k + 5);
            for (uint32_t i = 0; i < count; i++) {
                const size_t member = (size_t)f32_as_uint(read_uniform4(M.grid, 8).z) + 2 * (size_t)(first + i);
                if (run_record(lists_at + (size_t)read_word(M.grid, member) * 8, read_word(M.grid, member + 1))) return true;
            }
        }
        if (!(ref & kLeafBit) && (ref & kGridListBit)) {             // the cell of a flat mesh's grid names its leaves: nothing else to do
            if (run_list(ref)) return true;
            hit = q.hit;
            return q.found;
        }
        bool live = true;
        while (live) {
            auto step = [&](const PairNode& N) __attribute__((always_inline)) {
                if (COUNT) work.slabs += 2;
                float tl, tr;
                bool hl, hr;
                slab_pair(sr, N, tl, tr, hl, hr);
                if (CULL) { hl = hl && !(tl > q.limit); hr = hr && !(tr > q.limit); }
                const bool left_first = !(tr < tl);
                const uint32_t nearer = left_first ? N.lref : N.rref, farther = left_first ? N.rref : N.lref;
                const bool both = hl && hr;
                stack.push_if(both, farther);
                if (hl || hr) {
                    ref = both ? nearer : (hl ? N.lref : N.rref);
                } else if (stack.empty()) {
                    live = false;
                } else {
                    ref = stack.pop();
                }
            };
            while (live && !(ref & (kLeafBit | kGridListBit))) {
                if (BUDGET && --steps < 0) return false;
#if defined(__HIP_DEVICE_COMPILE__)
                // Neighbouring rays mostly walk the same nodes at the same time (two thirds of all node visits happen with every
                // active lane of the wave at one node): such a node is read once through the scalar cache, and the vector memory
                // path -- whose address unit is busy 16 cycles per 16-byte-per-lane load and was this loop's limit -- is spared
                // four loads.  Same arithmetic on the same bits either way.
                if (ScalarNodes<Nodes>::on) {
                    const uint32_t idx = M.pair_base + ref;
                    const uint32_t uidx = __builtin_amdgcn_readfirstlane(idx);
                    if (__builtin_amdgcn_ballot_w64(idx != uidx) == 0) {
                        step(read_pair_uniform(ScalarNodes<Nodes>::base(nodes), uidx));
                        continue;
                    }
                }
#endif
                step(nodes(M, ref));
            }
            if (live) {
                if (BUDGET && --steps < 0) return false;
                if ((ref & kLeafBit) ? test_leaf<ANY, COUNT, FULL, LB>(M, nodes, ref, o, d, q, work) : run_list(ref)) return true;
                if (stack.empty()) live = false;
                else ref = stack.pop();
            }
        }
        hit = q.hit;
        return q.found;
    }
    // Reference order: every box whose ancestors' boxes pass, no culling.
    stack.push(entry_ref<true>(M, d));
    while (!stack.empty()) {
        if (BUDGET && --steps < 0) return false;
        const uint32_t ref = stack.pop();
        if (ref & kLeafBit) {
            if (test_leaf<ANY, COUNT, FULL, LB>(M, nodes, ref, o, d, q, work)) return true;
        } else {
            const PairNode N = nodes(M, ref);
            if (COUNT) work.slabs += 2;
            float tl, tr;
            bool hl, hr;
            slab_pair(sr, N, tl, tr, hl, hr);
            // the reference goes left first (bvh.c:113-124): left child on top
            if (hr) stack.push(N.rref);
            if (hl) stack.push(N.lref);
        }
    }
    hit = q.hit;
    return q.found;
}

// ---- who traverses ------------------------------------------------------------------------
// A "party" runs the traversal of one ray against one mesh.  Solo: the calling lane alone.  The
// kernels add a party in which kGroupLanes adjacent lanes hold the same ray and share its stack.
template <uint32_t LB>
struct SoloT {
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    RT_HD bool closest(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, float bound, Hit& h, Stack& st, Work& w) const {
        int unused = 0;
        return trace_mesh<false, COUNT, FULL, false, LB>(M, nodes, o, d, inv, bound, h, st, w, unused);
    }
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    RT_HD bool any(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, Stack& st, Work& w) const {
        Hit h = no_hit();
        int unused = 0;
        return trace_mesh<true, COUNT, FULL, false, LB>(M, nodes, o, d, inv, 1e30f, h, st, w, unused);
    }
    RT_HD bool exhausted() const { return false; }
    static constexpr bool kVerify = false;
};
using Solo = SoloT<kLeafBatch>;

// What a query's caller can see of its result: whether it found a hit below `bound` (scene.c:82 discards the others), and
// which.  Two walks agree when this is the same.
RT_HD bool same_outcome(bool a_found, const Hit& a, bool b_found, const Hit& b, float bound) {
    const bool ea = a_found && a.t < bound, eb = b_found && b.t < bound;
    if (ea != eb) return false;
    if (!ea) return true;
    return f32_as_uint(a.t) == f32_as_uint(b.t) && a.tri == b.tri;
}

// The self-checking party (RT_FLAG_VERIFY_SKIPS): one lane per ray like Solo, but every query is walked three ways -- as the
// kernels walk it (second tree, grid entry of flat meshes, nearer child first, distance culling, any-hit stops at the first
// blocker), the same from the tree's root without the grid, and the reference's own way (its tree, every box whose
// ancestors' boxes pass, no culling: bvh.c:80-129) -- and the outcomes are compared.  The result handed back is the first
// one, so a frame rendered with this party is the frame the kernels render; what the reference's way would have changed is
// counted in Work.
struct SoloVerify {
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    RT_HD bool closest(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, float bound, Hit& h, Stack& st, Work& w) const {
        int unused = 0;
        Work scratch;
        Hit h2 = no_hit(), h3 = no_hit();
        const bool f1 = trace_mesh<false, COUNT, false, false, kLeafBatch>(M, nodes, o, d, inv, bound, h, st, w, unused);
        const bool f2 = trace_mesh<false, false, false, false, kLeafBatch>(M, nodes, o, d, inv, bound, h2, st, scratch, unused, false);
        const bool f3 = trace_mesh<false, false, true, false, kLeafBatch>(M, nodes, o, d, inv, bound, h3, st, scratch, unused);
        w.checks += 2;
        if (!same_outcome(f1, h, f2, h2, bound)) w.bad_entry++;
        if (!same_outcome(f2, h2, f3, h3, bound)) w.bad_cull++;
        return f1;
    }
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    RT_HD bool any(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, Stack& st, Work& w) const {
        int unused = 0;
        Work scratch;
        Hit h = no_hit();
        const bool f1 = trace_mesh<true, COUNT, false, false, kLeafBatch>(M, nodes, o, d, inv, 1e30f, h, st, w, unused);
        const bool f2 = trace_mesh<true, false, false, false, kLeafBatch>(M, nodes, o, d, inv, 1e30f, h, st, scratch, unused, false);
        // the reference's shadow pass is its closest-hit walk with only the flag used (scene.c:133-138)
        const bool f3 = trace_mesh<false, false, true, false, kLeafBatch>(M, nodes, o, d, inv, 1e30f, h, st, scratch, unused);
        w.checks += 2;
        if (f1 != f2) w.bad_entry++;
        if (f2 != f3) w.bad_cull++;
        return f1;
    }
    RT_HD bool exhausted() const { return false; }
    static constexpr bool kVerify = true;
};

// One lane alone, but only for as long as the pixel stays cheap: all queries of the pixel draw on
// one budget of node visits.
struct SoloBudget {
    mutable int steps;
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    RT_HD bool closest(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, float bound, Hit& h, Stack& st, Work& w) const {
        return trace_mesh<false, COUNT, FULL, true, kLeafBatch>(M, nodes, o, d, inv, bound, h, st, w, steps);
    }
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    RT_HD bool any(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, Stack& st, Work& w) const {
        Hit h = no_hit();
        return trace_mesh<true, COUNT, FULL, true, kLeafBatch>(M, nodes, o, d, inv, 1e30f, h, st, w, steps);
    }
    RT_HD bool exhausted() const { return steps < 0; }
    static constexpr bool kVerify = false;
};

struct PixelOut {
    uint8_t r, g, b;
    uint8_t shadow;      // 1 blocked, 0 lit, 255 no primary hit (2: no shadow ray sent, colour_only)
    int32_t hit_id;      // (mesh<<24)|tri or -1
    float hit_t;
    uint32_t hit;        // 0/1
    uint32_t dark;       // 0/1
    uint32_t deferred;   // 1: not rendered, the party ran out of budget (DEFER only)
};

// Row of the whole frame that row py of this launch is (see FrameArgs).
RT_HD int frame_row(const FrameArgs& A, int py) { return ((py >> 3) * A.strip_stride + A.strip_offset) * 8 + (py & 7); }

// DEFER=true gives up as soon as the party reports that its budget is exhausted and returns with
// `deferred` set; nothing of such a pixel is stored.
// colour_only is for callers that store nothing but the colour (no shadow plane, no counters): where the surface's own term
// max(0.2, n.L) is 0.2 already (it faces away from the light, or nearly), scene.c:143-144 gives diffuse = 0.2 blocked or not,
// and the shadow ray is not sent (`shadow` reads 2, `dark` 0).  Counting, reference-order and self-checking runs send every ray.
template <bool COUNT, bool FULL, bool DEFER, class Stack, class Nodes, class Party>
RT_HD PixelOut shade_pixel(const FrameArgs& A, const Nodes& nodes, const Party& party, int px, int py, Stack& stack, Work& work,
                           const bool colour_only = false) {
    PixelOut out;
    out.deferred = 0;
    // ---- camera ray (camera.c:12-30; basis and tan() hoisted to the host) ----
    const float fx = ((float)px + 0.5f) / (float)A.width;             // scene.c:61
    const float fy = ((float)frame_row(A, py) + 0.5f) / (float)A.full_height;   // scene.c:62
    const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale;         // camera.c:18
    const float ry = (1.0f - 2.0f * fy) * A.scale;                    // camera.c:19
    const F3 wo = f3(A.cam_pos);
    const F3 wd = unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));

    // ---- closest hit over the meshes (scene.c:72-108) ----
    float closest = 1e30f;
    int who = -1;
    Hit best = no_hit();
    const int tile_x = px >> 3, tile_y = frame_row(A, py) >> 3;
    for (int m = 0; m < A.n_mesh; m++) {
        const MeshArgs& M = A.mesh[m];
        // outside A.vis[m] the root test below fails for certain; counting and reference-order runs make it anyway
        if (!COUNT && !FULL && (tile_x < A.vis[m][0] || tile_x >= A.vis[m][2] || tile_y < A.vis[m][1] || tile_y >= A.vis[m][3])) {
            if (Party::kVerify) {                                      // the reference's root test on this ray (scene.c:79, bvh.c:82)
                const F3 vd = unit(xform_vector<4>(M.inv, wd));
                float entry;
                work.checks++;
                if (slab_t(f3(M.cam_o), recip(vd), M.root_min, M.root_max, entry)) work.bad_vis++;
            }
            continue;
        }
        const F3 lo = f3(M.cam_o);                                     // ray.c:19 (same for every pixel)
        const F3 ld = unit(xform_vector<4>(M.inv, wd));                // ray.c:20,22
        Hit h = no_hit();
        if (party.template closest<COUNT, FULL>(M, nodes, lo, ld, recip(ld), closest, h, stack, work) && h.t < closest) {
            closest = h.t; who = m; best = h;
        }
        if (DEFER && party.exhausted()) { out.deferred = 1; return out; }
    }
    out.hit_t = closest;
    if (who < 0) {                                                     // scene.c:156
        out.r = out.g = out.b = 50; out.shadow = 255; out.hit_id = -1; out.hit = 0; out.dark = 0;
        return out;
    }
    const MeshArgs& M = A.mesh[who];
    out.hit = 1;
    out.hit_id = (int32_t)(((uint32_t)who << 24) | best.tri);

    // ---- attributes of the winning hit (scene.c:86-106) ----
    const AttrRec R = read_attrs(M.attrs, best.tri);
    const float u = best.u, v = best.v;
    const float w = 1.0f - u - v;
    const float tu = w * R.t0[0] + u * R.t1[0] + v * R.t2[0];
    const float tv = w * R.t0[1] + u * R.t1[1] + v * R.t2[1];
    F3 n = unit(add(add(scale(f3(R.n0), w), scale(f3(R.n1), u)), scale(f3(R.n2), v)));
    n = unit(xform_vector<3>(M.nrm, n));                               // ray.c:39

    // ---- texel (mesh.c:138-149) ----
    const float su = tu - floorf(tu), sv = tv - floorf(tv);
    const int tx = (int)(su * (float)(M.tex_w - 1));
    const int ty = (int)(sv * (float)(M.tex_h - 1));
    const uint32_t texel = read_word(M.texels, (size_t)(ty * M.tex_w + tx));
    F3 color = f3((float)(texel & 0xFFu) / 255.0f, (float)((texel >> 8) & 0xFFu) / 255.0f,
                  (float)((texel >> 16) & 0xFFu) / 255.0f);

    // ---- shadow ray (scene.c:117-139) ----
    const F3 p = add(wo, scale(wd, closest));
    const F3 so = add(p, scale(n, 0.001f));
    bool dark = false;
    const F3 L = f3(A.light_dir);
    const float facing = dot(n, L);
    const bool unsent = colour_only && !COUNT && !FULL && !Party::kVerify && !(facing > 0.2f);   // a NaN compares false: fmaxf(0.2f, NaN) is 0.2f as well
    const int n_shadow = unsent ? 0 : A.n_mesh;
    if (COUNT && !(facing > 0.2f)) work.idle++;
    for (int m = 0; m < n_shadow && !dark; m++) {
        const MeshArgs& S = A.mesh[m];
        // outside A.vis_shadow[m] no surface point seen in this tile can send its shadow ray through the mesh's root box
        if (!COUNT && !FULL && (tile_x < A.vis_shadow[m][0] || tile_x >= A.vis_shadow[m][2] || tile_y < A.vis_shadow[m][1] || tile_y >= A.vis_shadow[m][3])) {
            if (Party::kVerify) {                                      // the reference's root test on this shadow ray (scene.c:131, bvh.c:82)
                const F3 vo = S.w_mode == 2 ? xform_point_w1(S.inv, so) : xform_point(S.inv, S.wrow, so);
                float entry;
                work.checks++;
                if (slab_t(vo, f3(S.lit_inv), S.root_min, S.root_max, entry)) work.bad_shadow_vis++;
            }
            continue;
        }
        const F3 lo = S.w_mode == 2 ? xform_point_w1(S.inv, so) : xform_point(S.inv, S.wrow, so);   // ray.c:19
        dark = party.template any<COUNT, FULL>(S, nodes, lo, f3(S.lit_d), f3(S.lit_inv), stack, work);
        if (DEFER && party.exhausted()) { out.deferred = 1; return out; }
    }
    float diffuse = 0.2f;                                              // scene.c:115
    if (!dark) diffuse = fmaxf(diffuse, facing);                       // scene.c:143-144
    color = scale(mul(color, f3(A.light_color)), diffuse);             // scene.c:148-149
    out.r = colour_byte(color.x);                // scene.c:152-154 (truncation)
    out.g = colour_byte(color.y);
    out.b = colour_byte(color.z);
    out.shadow = dark ? 1 : (unsent ? 2 : 0);
    out.dark = dark ? 1 : 0;
    return out;
}

}  // namespace rt
