// rt_instanced_wave.h -- the instanced frame (rt_instanced.h) scheduled per wavefront instead of per lane.
//
// One lane per pixel walking "hierarchy over the instances, then the instance's own BVH, then its triangles"
// as nested loops leaves most lanes of a wave waiting: neighbouring rays enter different instances at different
// times, many proposed instances fail the exact root-box test at once, and every inner loop lasts as long as its
// slowest lane.  Here every lane is a small state machine -- its ray is either at a box pair (of the hierarchy or of
// an instance's BVH: the same 64-byte record and the same slab arithmetic, only the ray's space and the node array
// differ), at a leaf of triangles, at a change of space (entering an instance, leaving it), at the end of a ray
// (shading, then the shadow ray through the same machine; storing the pixel) -- and the WAVE decides what to execute
// next: it counts the lanes in each state with ballots and runs the step most lanes wait for.  Lanes that finish
// their pixel take the next pixel of the wave's strip of tiles, so a wave stays full until the strip runs out.
//
// Result: the reference's (scene.c:72-139 over the mesh list): closest hit = minimum of (t, list index) over the
// instances' own closest hits, ties inside an instance by the triangles' priorities (rt_pixel.h); shadow = any
// instance blocks.  One running best per ray replaces the per-instance query: a triangle replaces it iff
// (t, instance index, -priority) is lexicographically smaller, which selects the same winner.
// Device code only (wave64 ballots).
#pragma once
#include "rt_instanced.h"
#include "rt_lds_stack.h"

namespace rt {

constexpr uint32_t kRefSentinel = 0xFFFFFFFFu;   // on the stack below an instance's nodes: back to world space
constexpr uint32_t kRefEnd = 0xFFFFFFFEu;        // bottom of the stack: the ray is through
// Neither can be a real leaf reference: that would need a leaf of 127 triangles starting at triangle 2^24 - 1 or - 2.

enum : int { W_BOX = 0, W_LEAF = 1, W_SWITCH = 2, W_SHADE = 3, W_STORE = 4, W_DONE = 5 };

template <int BLOCK>
struct WaveLane {
    // the ray in world space and in the space it is traversed in
    F3 wo, wd, winv;
    F3 lo, ld;                   // current space: origin, direction (Moeller-Trumbore needs them)
    SlabRay sr;                  // current space, packed for the box tests
    const PairNode* base;        // node array of the current space
    const TriRec* tris;          // triangles of the current instance
    uint32_t ref;
    LdsStack<BLOCK> stack;
    bool in_inst, shadow_mode;
    int inst;                    // instance being traversed
    // running best of the camera ray / result of the shadow ray
    float closest, limit;
    int who;
    float bu, bv;
    uint32_t btri, bprio;
    bool dark;
    // pixel
    int pixel;                   // index inside the wave's strip, -1 = none
    int px, py;
    F3 color;                    // texel * light colour, waiting for the shadow ray
    float lit;                   // max(0.2, n.L)
};

template <int BLOCK>
__device__ __forceinline__ int lane_state(const WaveLane<BLOCK>& L) {
    if (L.pixel < 0) return W_DONE;
    if (!(L.ref & kLeafBit)) return W_BOX;
    if (L.ref == kRefEnd) return L.shadow_mode ? W_STORE : W_SHADE;
    if (L.in_inst && L.ref != kRefSentinel) return W_LEAF;
    return W_SWITCH;
}

// Starts a ray (camera or shadow) at the root of the instance hierarchy.  Sets ref = kRefEnd when it cannot hit anything.
template <int BLOCK>
__device__ __forceinline__ void start_ray(WaveLane<BLOCK>& L, const InstancedArgs& I) {
    L.stack.clear();
    L.stack.push(kRefEnd);
    L.in_inst = false;
    L.base = I.tlas;
    L.lo = L.wo; L.ld = L.wd;
    L.sr = slab_ray(L.wo, L.winv);
    float entry;
    if (I.n_inst <= 0 || !slab_t(L.wo, L.winv, I.root_min, I.root_max, entry)) L.ref = L.stack.pop();
    else L.ref = I.tlas_root;
}

template <int BLOCK>
__device__ __forceinline__ void start_pixel(WaveLane<BLOCK>& L, const FrameArgs& A, const InstancedArgs& I) {
    const float fx = ((float)L.px + 0.5f) / (float)A.width;                     // scene.c:61
    const float fy = ((float)frame_row(A, L.py) + 0.5f) / (float)A.full_height; // scene.c:62
    const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale;                   // camera.c:18
    const float ry = (1.0f - 2.0f * fy) * A.scale;                              // camera.c:19
    L.wo = f3(A.cam_pos);
    L.wd = unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));
    L.winv = recip(L.wd);
    L.shadow_mode = false; L.dark = false;
    L.closest = 1e30f; L.limit = cull_limit(1e30f); L.who = -1;
    L.bu = L.bv = 0.0f; L.btri = 0; L.bprio = 0;
    start_ray(L, I);
}

// One box pair: both children's slab tests, nearer child first, the farther one parked (trace_mesh's inner step).
// (This form keeps the distance cull inside every instance; the host does not launch it for a list that places a MeshArgs::wide mesh.)
template <int BLOCK>
__device__ __forceinline__ void step_box(WaveLane<BLOCK>& L) {
    const PairNode N = L.base[L.ref];
    float tl, tr;
    bool hl, hr;
    slab_pair(L.sr, N, tl, tr, hl, hr);
    if (!L.shadow_mode) { hl = hl && !(tl > L.limit); hr = hr && !(tr > L.limit); }
    const bool left_first = !(tr < tl);
    const uint32_t nearer = left_first ? N.lref : N.rref, farther = left_first ? N.rref : N.lref;
    const bool both = hl && hr;
    L.stack.push_if(both, farther);
    if (hl || hr) L.ref = both ? nearer : (hl ? N.lref : N.rref);
    else L.ref = L.stack.pop();
}

// The triangles of one leaf of the current instance (bvh.c:89-103), two fetched at a time.
template <int BLOCK>
__device__ __forceinline__ void step_leaf(WaveLane<BLOCK>& L) {
    const uint32_t first = L.ref & 0x00FFFFFFu, count = (L.ref >> 24) & 0x7Fu;
    bool stop = false;
    for (uint32_t i0 = 0; i0 < count && !stop; i0 += kLeafBatch) {
        TriRec T[kLeafBatch];
        const uint32_t n = count - i0 < kLeafBatch ? count - i0 : kLeafBatch;
        RT_UNROLL
        for (uint32_t k = 0; k < kLeafBatch; k++) T[k] = read_tri(L.tris, first + i0 + (k < n ? k : 0));
        RT_UNROLL
        for (uint32_t k = 0; k < kLeafBatch; k++) {
            if (k >= n || stop) continue;
            float t, u, v;
            if (!moller_trumbore(L.lo, L.ld, f3(T[k].v0), f3(T[k].e1), f3(T[k].e2), t, u, v)) continue;
            if (L.shadow_mode) {
                if (t < 1e30f) { L.dark = true; stop = true; }         // scene.c:133-138: any accepted triangle blocks
            } else if (t < L.closest ||
                       (t == L.closest && (L.inst < L.who || (L.inst == L.who && T[k].prio > L.bprio)))) {
                L.closest = t; L.who = L.inst; L.bu = u; L.bv = v; L.btri = first + i0 + k; L.bprio = T[k].prio;
                L.limit = cull_limit(t);
            }
        }
    }
    L.ref = L.stack.pop();
}

// Entering the instance a hierarchy leaf names, or leaving the one just finished.
template <int BLOCK>
__device__ __forceinline__ void step_switch(WaveLane<BLOCK>& L, const InstancedArgs& I, const SceneArgs& S) {
    if (L.in_inst) {                                  // the sentinel: the instance's nodes are used up
        L.in_inst = false;
        L.base = I.tlas;
        L.lo = L.wo; L.ld = L.wd;
        L.sr = slab_ray(L.wo, L.winv);
        L.ref = L.stack.pop();
        return;
    }
    const int i = (int)(L.ref & 0x00FFFFFFu);                                // the leaf names its instance
    if (((L.ref >> 24) & 0x7Fu) == 0) { L.ref = L.stack.pop(); return; }
    const MeshArgs& M = I.inst[i];
    F3 o, d, inv;
    if (L.shadow_mode) {
        o = M.w_mode == 2 ? xform_point_w1(M.inv, L.wo) : xform_point(M.inv, M.wrow, L.wo);      // ray.c:19
        d = f3(M.lit_d); inv = f3(M.lit_inv);
    } else {
        o = f3(M.cam_o);                                                                          // ray.c:19, same for every pixel
        d = unit(xform_vector<4>(M.inv, L.wd));                                                   // ray.c:20,22
        inv = recip(d);
    }
    float entry;
    bool ok = slab_t(o, inv, M.root_min, M.root_max, entry) && !M.empty;                          // bvh.c:82 at the root
    if (!L.shadow_mode && entry > L.limit) ok = false;
    if (!ok) { L.ref = L.stack.pop(); return; }
    L.stack.push(kRefSentinel);
    L.in_inst = true; L.inst = i;
    L.lo = o; L.ld = d;
    L.sr = slab_ray(o, inv);
    L.base = S.pairs + M.pair_base;
    L.tris = M.tris;
    L.ref = entry_ref<false>(M, d);
}

// End of the camera ray: background (the pixel goes straight to its store), or attributes + texel of the winner and the
// shadow ray's start (scene.c:86-139).
template <int BLOCK>
__device__ __forceinline__ void step_shade(WaveLane<BLOCK>& L, const FrameArgs& A, const InstancedArgs& I) {
    L.shadow_mode = true;
    if (L.who < 0) { L.ref = kRefEnd; return; }
    const MeshArgs& M = I.inst[L.who];
    const AttrRec R = read_attrs(M.attrs, L.btri);
    const float u = L.bu, v = L.bv;
    const float w = 1.0f - u - v;
    const float tu = w * R.t0[0] + u * R.t1[0] + v * R.t2[0];
    const float tv = w * R.t0[1] + u * R.t1[1] + v * R.t2[1];
    F3 n = unit(add(add(scale(f3(R.n0), w), scale(f3(R.n1), u)), scale(f3(R.n2), v)));
    n = unit(xform_vector<3>(M.nrm, n));                                                          // ray.c:39
    const float su = tu - floorf(tu), sv = tv - floorf(tv);                                       // mesh.c:138-149
    const int tx = (int)(su * (float)(M.tex_w - 1));
    const int ty = (int)(sv * (float)(M.tex_h - 1));
    const uint32_t texel = read_word(M.texels, (size_t)(ty * M.tex_w + tx));
    const F3 tex = f3((float)(texel & 0xFFu) / 255.0f, (float)((texel >> 8) & 0xFFu) / 255.0f, (float)((texel >> 16) & 0xFFu) / 255.0f);
    L.color = mul(tex, f3(A.light_color));                                                        // scene.c:148
    const F3 lightdir = f3(A.light_dir);
    L.lit = fmaxf(0.2f, dot(n, lightdir));                                                        // scene.c:143-144
    L.wo = add(add(L.wo, scale(L.wd, L.closest)), scale(n, 0.001f));                              // scene.c:118-120
    L.wd = lightdir;
    L.winv = recip(lightdir);
    L.limit = 3.0e38f;
    start_ray(L, I);
}

}  // namespace rt
