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
//    higher" picks the reference's winner regardless of visiting order.
//  * Across meshes an earlier mesh wins ties (strict '<', scene.c:82).
//  * The shadow pass only uses "any triangle accepted" (scene.c:133-138), so it may stop at the
//    first accepted triangle.
#pragma once
#include "rt_exact.h"
#include "rt_layout.h"

namespace rt {

struct Hit {
    float t, u, v;
    uint32_t tri;
    uint32_t prio;
};

struct Work {
    uint32_t slabs, tris;
};

// Closest hit (ANY=false) or any hit (ANY=true) of one mesh, ray already in mesh space.
// `bound` is the t the caller passes in (scene.c:74): the reference only honours it when the
// root is a leaf (bvh.c:85); otherwise leaves start from 1e30f (bvh.c:106).
// FULL=true keeps going after the first accepted triangle of an any-hit query, so that the
// work counters equal the reference's (which always finishes the traversal).
template <bool ANY, bool COUNT, bool FULL, class Stack>
RT_HD bool trace_mesh(const MeshArgs& M, F3 o, F3 d, F3 inv, float bound, Hit& hit, Stack& stack, Work& work) {
    if (COUNT) work.slabs++;
    if (!slab(o, inv, M.root_min, M.root_max)) return false;           // bvh.c:82 at the root
    if (M.empty) return false;
    const bool root_leaf = (M.root_ref & kLeafBit) != 0;
    float best_t = root_leaf ? bound : 1e30f;
    bool found = false;
    stack.clear();
    stack.push(M.root_ref);
    while (!stack.empty()) {
        const uint32_t ref = stack.pop();
        if (ref & kLeafBit) {
            const uint32_t first = ref & 0x00FFFFFFu, count = (ref >> 24) & 0x7Fu;
            for (uint32_t i = 0; i < count; i++) {
                const TriRec& T = M.tris[first + i];
                float t, u, v;
                if (COUNT) work.tris++;
                if (!moller_trumbore(o, d, f3(T.v0), f3(T.e1), f3(T.e2), t, u, v)) continue;
                if (ANY) {
                    if (t < best_t) {
                        if (!FULL) return true;
                        found = true;
                    }
                } else if (t < best_t || (found && t == best_t && T.prio > hit.prio)) {
                    best_t = t; found = true;
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = first + i; hit.prio = T.prio;
                }
            }
        } else {
            const PairNode& N = M.pairs[ref];
            if (COUNT) work.slabs += 2;
            const bool hl = slab(o, inv, N.lmin, N.lmax);
            const bool hr = slab(o, inv, N.rmin, N.rmax);
            if (hr) stack.push(N.rref);
            if (hl) stack.push(N.lref);
        }
    }
    return found;
}

struct PixelOut {
    uint8_t r, g, b;
    uint8_t shadow;      // 1 blocked, 0 lit, 255 no primary hit
    int32_t hit_id;      // (mesh<<24)|tri or -1
    float hit_t;
    uint32_t hit;        // 0/1
    uint32_t dark;       // 0/1
};

template <bool COUNT, bool FULL, class Stack>
RT_HD PixelOut shade_pixel(const FrameArgs& A, int px, int py, Stack& stack, Work& work) {
    PixelOut out;
    // ---- camera ray (camera.c:12-30; basis and tan() hoisted to the host) ----
    const float fx = ((float)px + 0.5f) / (float)A.width;             // scene.c:61
    const float fy = ((float)py + 0.5f) / (float)A.height;            // scene.c:62
    const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale;         // camera.c:18
    const float ry = (1.0f - 2.0f * fy) * A.scale;                    // camera.c:19
    const F3 wo = f3(A.cam_pos);
    const F3 wd = unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));

    // ---- closest hit over the meshes (scene.c:72-108) ----
    float closest = 1e30f;
    int who = -1;
    Hit best; best.t = 0; best.u = 0; best.v = 0; best.tri = 0; best.prio = 0;
    for (int m = 0; m < A.n_mesh; m++) {
        const MeshArgs& M = A.mesh[m];
        const F3 lo = f3(M.cam_o);                                     // ray.c:19 (same for every pixel)
        const F3 ld = unit(xform_vector<4>(M.inv, wd));                // ray.c:20,22
        Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0; h.prio = 0;
        if (trace_mesh<false, COUNT, FULL>(M, lo, ld, recip(ld), closest, h, stack, work) && h.t < closest) {
            closest = h.t; who = m; best = h;
        }
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
    const AttrRec& R = M.attrs[best.tri];
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
    const uint32_t texel = M.texels[ty * M.tex_w + tx];
    F3 color = f3((float)(texel & 0xFFu) / 255.0f, (float)((texel >> 8) & 0xFFu) / 255.0f,
                  (float)((texel >> 16) & 0xFFu) / 255.0f);

    // ---- shadow ray (scene.c:117-139) ----
    const F3 p = add(wo, scale(wd, closest));
    const F3 so = add(p, scale(n, 0.001f));
    bool dark = false;
    for (int m = 0; m < A.n_mesh && !dark; m++) {
        const MeshArgs& S = A.mesh[m];
        const F3 lo = xform_point(S.inv, S.wrow, so);                  // ray.c:19
        Hit h; h.t = 0; h.u = 0; h.v = 0; h.tri = 0; h.prio = 0;
        dark = trace_mesh<true, COUNT, FULL>(S, lo, f3(S.lit_d), f3(S.lit_inv), 1e30f, h, stack, work);
    }
    float diffuse = 0.2f;                                              // scene.c:115
    const F3 L = f3(A.light_dir);
    if (!dark) diffuse = fmaxf(diffuse, dot(n, L));                    // scene.c:143-144
    color = scale(mul(color, f3(A.light_color)), diffuse);             // scene.c:148-149
    out.r = (uint8_t)(fminf(color.x * 255.0f, 255.0f));                // scene.c:152-154 (truncation)
    out.g = (uint8_t)(fminf(color.y * 255.0f, 255.0f));
    out.b = (uint8_t)(fminf(color.z * 255.0f, 255.0f));
    out.shadow = dark ? 1 : 0;
    out.dark = dark ? 1 : 0;
    return out;
}

}  // namespace rt
