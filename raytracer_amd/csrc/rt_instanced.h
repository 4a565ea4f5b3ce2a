// rt_instanced.h -- one pixel of the frame loop for scenes with many placed meshes.
//
// The reference tests a ray against every mesh of the scene in order (scene.c:72-108) and keeps a hit
// only if it is strictly closer than the closest so far, so its result is the minimum of
// (t, mesh index) over the meshes' own closest hits.  Written like that it does not depend on the
// order in which meshes are tried, and a mesh whose box the ray cannot reach need not be tried at
// all.  Here a hierarchy over the instances' world-space boxes (padded, see the host builder)
// proposes the instances; each proposed instance gets the reference's own test in its own space
// (ray.c:5-22 transform, root box, BVH), and the winner is chosen by (t, instance index).
// The shadow pass (scene.c:124-139) is "any instance blocks".
#pragma once
#include "rt_pixel.h"

namespace rt {

struct InstancedOut {
    uint8_t r, g, b;
    uint8_t shadow;      // 1 blocked, 0 lit, 255 no primary hit (2: no shadow ray sent, colour_only)
    int32_t inst;        // winning instance or -1
    int32_t tri;         // its triangle or -1
    float hit_t;
    uint32_t hit, dark;
};

// World-space walk over the instance hierarchy, one instance at a time.  next() goes on until it reaches an instance whose
// padded box the ray enters no farther than limit() AND that accept(instance index) takes (the caller's exact test in the
// instance's own space); it returns that index, or -1 when the hierarchy is used up.  The caller then walks the instance.
// Written like this the lanes of a wave look for their next instance together and walk together: a lane whose proposal
// fails the exact test goes on searching at once instead of sitting out the others' walk.
template <class TStack>
struct InstanceWalk {
    SlabRay sr;
    uint32_t ref;
    bool live;
    RT_HD void begin(const InstancedArgs& I, F3 o, F3 inv, TStack& ts) {
        float entry;
        live = I.n_inst > 0 && slab_t(o, inv, I.root_min, I.root_max, entry);
        sr = slab_ray(o, inv);
        ts.clear();
        ref = I.tlas_root;
    }
    template <class Limit, class Accept>
    RT_HD int next(const InstancedArgs& I, TStack& ts, Limit limit, Accept accept) {
        while (live) {
            while (live && !(ref & kLeafBit)) {
                const PairNode N = I.tlas[ref];
                float tl, tr;
                bool hl, hr;
                slab_pair(sr, N, tl, tr, hl, hr);
                const float lim = limit();
                hl = hl && !(tl > lim); hr = hr && !(tr > lim);
                const bool left_first = !(tr < tl);
                if (hl && hr) { ts.push(left_first ? N.rref : N.lref); ref = left_first ? N.lref : N.rref; }
                else if (hl || hr) ref = hl ? N.lref : N.rref;
                else if (ts.empty()) live = false;
                else ref = ts.pop();
            }
            if (live) {
                const uint32_t i = ref & 0x00FFFFFFu, count = (ref >> 24) & 0x7Fu;      // the leaf names its instance
                if (ts.empty()) live = false;
                else ref = ts.pop();
                if (count > 0 && accept(i)) return (int)i;
            }
        }
        return -1;
    }
};

// The fields of an instance's MeshArgs that a walk through its tree reads (begin_query, entry_ref, the node and triangle
// sources), copied to a record the caller keeps in registers.  The flat-mesh grid is left out (grid_axis = 3: none): its
// nine fields would have to ride along for every instance, and the instances that are not flat pay for them.
RT_HD void take_walk_fields(const MeshArgs& M, MeshArgs& W) {
    W.pair_base = M.pair_base; W.tris = M.tris;
    W.root_min[0] = M.root_min[0]; W.root_min[1] = M.root_min[1]; W.root_min[2] = M.root_min[2];
    W.root_max[0] = M.root_max[0]; W.root_max[1] = M.root_max[1]; W.root_max[2] = M.root_max[2];
    W.root_ref = M.root_ref; W.ref_root = M.ref_root; W.empty = M.empty; W.wide = M.wide;
    W.res_pairs = kNotResident; W.res_tris = kNotResident;
    W.grid = nullptr; W.grid_axis = 3u;
}

template <bool COUNT, class Stack, class TStack, class Nodes, class Party>
RT_HD InstancedOut shade_instanced_pixel(const FrameArgs& A, const InstancedArgs& I, const Nodes& nodes, const Party& party, int px, int py,
                                         Stack& stack, TStack& tstack, Work& work, const bool colour_only = false) {
    InstancedOut out;
    // ---- camera ray (camera.c:12-30) ----
    const float fx = ((float)px + 0.5f) / (float)A.width;
    const float fy = ((float)frame_row(A, py) + 0.5f) / (float)A.full_height;
    const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale;
    const float ry = (1.0f - 2.0f * fy) * A.scale;
    const F3 wo = f3(A.cam_pos);
    const F3 wd = unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));

    // ---- closest hit over the instances ----
    float closest = 1e30f;
    int who = -1;
    Hit best = no_hit();
    InstanceWalk<TStack> walk;
    walk.begin(I, wo, recip(wd), tstack);
    for (;;) {
        // What the walk inside an instance reads of its MeshArgs is taken over into registers in one go when the instance is
        // proposed: read field by field where each is first needed, every one costs its own trip to memory, one after the
        // other (instances differ from lane to lane, so these are vector loads of a record that is rarely in the L1).
        MeshArgs W;
        F3 ld = f3(0.0f, 0.0f, 0.0f), linv = ld, lo = ld;
        const int i = walk.next(I, tstack, [&]() { return cull_limit(closest); }, [&](uint32_t k) {
            const MeshArgs& M = I.inst[k];
            take_walk_fields(M, W);
            lo = f3(M.cam_o);
            ld = unit(xform_vector<4>(M.inv, wd));                                               // ray.c:20,22
            linv = recip(ld);
            // what begin_query decides (bvh.c:82 at the root, and the distance the closest hit so far allows); a mesh that is
            // a single leaf honours the bound inside the leaf (bvh.c:85): it gets none, the comparison below decides
            float entry;
            if (!slab_t(lo, linv, W.root_min, W.root_max, entry) || W.empty) return false;
            return W.wide || !(entry > cull_limit((W.root_ref & kLeafBit) ? 1e30f : closest));
        });
        if (i < 0) break;
        Hit h = no_hit();
        const float bound = (W.root_ref & kLeafBit) ? 1e30f : closest;
        if (party.template closest<COUNT, false>(W, nodes, lo, ld, linv, bound, h, stack, work) &&
            (h.t < closest || (h.t == closest && i < who))) {
            closest = h.t; who = i; best = h;
        }
    }
    if (Party::kVerify) {
        // the reference's loop over the whole list (scene.c:72-108): every entry's own walk, strictly closer wins
        float ref_closest = 1e30f;
        int ref_who = -1;
        uint32_t ref_tri = 0;
        Work scratch;
        int unused = 0;
        for (int k = 0; k < I.n_total; k++) {
            const MeshArgs& M = I.inst[k];
            if (M.empty) continue;                    // no triangles: nothing the reference could accept
            const F3 ld = unit(xform_vector<4>(M.inv, wd));
            Hit h = no_hit();
            if (trace_mesh<false, false, true, false, kLeafBatch>(M, nodes, f3(M.cam_o), ld, recip(ld), ref_closest, h, stack, scratch, unused) &&
                h.t < ref_closest) {
                ref_closest = h.t; ref_who = k; ref_tri = h.tri;
            }
        }
        work.checks++;
        if (ref_who != who || (who >= 0 && (f32_as_uint(ref_closest) != f32_as_uint(closest) || ref_tri != best.tri))) work.bad_inst++;
    }
    out.hit_t = closest;
    if (who < 0) {
        out.r = out.g = out.b = 50; out.shadow = 255; out.inst = -1; out.tri = -1; out.hit = 0; out.dark = 0;
        return out;
    }
    const MeshArgs& M = I.inst[who];
    out.hit = 1; out.inst = who; out.tri = (int32_t)best.tri;

    // ---- attributes, texel (scene.c:86-106, mesh.c:138-149) ----
    const AttrRec R = read_attrs(M.attrs, best.tri);
    const float u = best.u, v = best.v;
    const float w = 1.0f - u - v;
    const float tu = w * R.t0[0] + u * R.t1[0] + v * R.t2[0];
    const float tv = w * R.t0[1] + u * R.t1[1] + v * R.t2[1];
    F3 n = unit(add(add(scale(f3(R.n0), w), scale(f3(R.n1), u)), scale(f3(R.n2), v)));
    n = unit(xform_vector<3>(M.nrm, n));
    const float su = tu - floorf(tu), sv = tv - floorf(tv);
    const int tx = (int)(su * (float)(M.tex_w - 1));
    const int ty = (int)(sv * (float)(M.tex_h - 1));
    const uint32_t texel = read_word(M.texels, (size_t)(ty * M.tex_w + tx));
    F3 color = f3((float)(texel & 0xFFu) / 255.0f, (float)((texel >> 8) & 0xFFu) / 255.0f, (float)((texel >> 16) & 0xFFu) / 255.0f);

    // ---- shadow ray (scene.c:117-139): any instance ----
    const F3 so = add(add(wo, scale(wd, closest)), scale(n, 0.001f));
    const F3 L = f3(A.light_dir);
    bool dark = false;
    // a caller that stores nothing but the colour: where max(0.2, n.L) is 0.2 already, scene.c:143-144 gives 0.2 blocked or not,
    // and the shadow ray is not sent (shade_pixel does the same)
    const float facing = dot(n, L);
    const bool unsent = colour_only && !COUNT && !Party::kVerify && !(facing > 0.2f);
    if (COUNT && !(facing > 0.2f)) work.idle++;
    if (!unsent) walk.begin(I, so, recip(L), tstack);
    while (!dark && !unsent) {
        MeshArgs W;
        F3 lo = so, sd = L, sinv = L;
        const int i = walk.next(I, tstack, []() { return 3.0e38f; }, [&](uint32_t k) {
            const MeshArgs& S = I.inst[k];
            take_walk_fields(S, W);
            sd = f3(S.lit_d); sinv = f3(S.lit_inv);
            lo = S.w_mode == 2 ? xform_point_w1(S.inv, so) : xform_point(S.inv, S.wrow, so);     // ray.c:19
            float entry;
            return slab_t(lo, sinv, W.root_min, W.root_max, entry) && !W.empty;                  // bvh.c:82 at the root
        });
        if (i < 0) break;
        dark = party.template any<COUNT, false>(W, nodes, lo, sd, sinv, stack, work);
    }
    if (Party::kVerify) {
        // the reference's shadow loop (scene.c:124-139): every entry until one blocks
        bool ref_dark = false;
        Work scratch;
        int unused = 0;
        for (int k = 0; k < I.n_total && !ref_dark; k++) {
            const MeshArgs& S = I.inst[k];
            if (S.empty) continue;
            const F3 lo = S.w_mode == 2 ? xform_point_w1(S.inv, so) : xform_point(S.inv, S.wrow, so);
            Hit h = no_hit();
            ref_dark = trace_mesh<false, false, true, false, kLeafBatch>(S, nodes, lo, f3(S.lit_d), f3(S.lit_inv), 1e30f, h, stack, scratch, unused);
        }
        work.checks++;
        if (ref_dark != dark) work.bad_inst++;
    }
    float diffuse = 0.2f;
    if (!dark) diffuse = fmaxf(diffuse, facing);
    color = scale(mul(color, f3(A.light_color)), diffuse);
    out.r = colour_byte(color.x);
    out.g = colour_byte(color.y);
    out.b = colour_byte(color.z);
    out.shadow = dark ? 1 : (unsent ? 2 : 0);
    out.dark = dark ? 1 : 0;
    return out;
}

}  // namespace rt
