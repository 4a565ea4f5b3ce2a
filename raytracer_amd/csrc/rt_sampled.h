// rt_sampled.h -- the frame loop with several camera rays per pixel and several directional lights.
//
// The reference traces one ray through the pixel centre and has one light (scene.c:61-62, scene.h:17).
// This extension keeps every per-ray operation of rt_pixel.h and defines what surrounds it:
//
//   sample s of pixel (x, y):  spp == 1: offset (0.5, 0.5), the reference's ray.
//                              spp  > 1: offset (jx, jy) in [0,1)^2 from a counter-based hash of
//                              (x, y, s, seed), see sample_offset().
//   lights:   for every light l, in order: one shadow ray from the same origin (scene.c:118-119),
//             d_l = 0.2, or max(0.2, n.L_l) when nothing blocks it (scene.c:115,141-146);
//             colour = sum over l of (texel * color_l) * d_l, the first term as it stands, the others
//             added in order.
//   sample value:  min(colour * 255, 255) per channel (scene.c:152-154 before the cast); 50 for a miss
//                  (scene.c:156).
//   pixel:    the sample values summed in order, divided by (float)spp when spp > 1, then cast to
//             unsigned char.
// With spp == 1 and one light this is the reference's pixel, bit for bit.
#pragma once
#include "rt_pixel.h"

namespace rt {

// PCG-style 32-bit mix (Jarzynski & Olano, "Hash Functions for GPU Rendering").
RT_HD uint32_t mix32(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    const uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}

// Offset of sample s inside pixel (x, y): 24-bit fractions, exact in float.
RT_HD void sample_offset(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float& jx, float& jy) {
    const uint32_t h0 = mix32(x ^ mix32(y ^ mix32(s ^ mix32(seed))));
    const uint32_t h1 = mix32(h0);
    jx = (float)(h0 >> 8) * (1.0f / 16777216.0f);
    jy = (float)(h1 >> 8) * (1.0f / 16777216.0f);
}

struct SampledOut {
    uint8_t r, g, b;
    uint32_t hits;       // samples that hit something
    uint32_t darks;      // (sample, light) pairs whose shadow ray was blocked
    int32_t hit_id;      // of sample 0: (mesh<<24)|tri or -1
};

// Value of sample s of pixel (px, row of the whole frame): three floats in [0, 255].  Adds to hits / darks;
// sample 0 sets hit_id.
template <bool COUNT, class Stack, class Nodes, class Party>
RT_HD F3 sample_value(const FrameArgs& A, const SampleArgs& X, const Nodes& nodes, const Party& party, int px, int row, int s,
                      Stack& stack, Work& work, SampledOut& out) {
    const F3 wo = f3(A.cam_pos);
    float jx = 0.5f, jy = 0.5f;
    if (X.spp > 1) sample_offset((uint32_t)px, (uint32_t)row, (uint32_t)s, X.seed, jx, jy);
    // ---- camera ray (scene.c:61-62 with the offset in place of 0.5f; camera.c:12-30) ----
    const float fx = ((float)px + jx) / (float)A.width;
    const float fy = ((float)row + jy) / (float)A.full_height;
    const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale;
    const float ry = (1.0f - 2.0f * fy) * A.scale;
    const F3 wd = unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));

    // ---- closest hit over the meshes (scene.c:72-108) ----
    float closest = 1e30f;
    int who = -1;
    Hit best = no_hit();
    const int tile_x = px >> 3, tile_y = row >> 3;
    for (int m = 0; m < A.n_mesh; m++) {
        const MeshArgs& M = A.mesh[m];
        // outside A.vis[m] (two pixels wider than the mesh's root box can reach: a sample stays inside its pixel) the root
        // test fails for certain
        if (!COUNT && (tile_x < A.vis[m][0] || tile_x >= A.vis[m][2] || tile_y < A.vis[m][1] || tile_y >= A.vis[m][3])) {
            if (Party::kVerify) {                                      // the reference's root test on this ray (scene.c:79, bvh.c:82)
                const F3 vd = unit(xform_vector<4>(M.inv, wd));
                float entry;
                work.checks++;
                if (slab_t(f3(M.cam_o), recip(vd), M.root_min, M.root_max, entry)) work.bad_vis++;
            }
            continue;
        }
        const F3 ld = unit(xform_vector<4>(M.inv, wd));
        Hit h = no_hit();
        if (party.template closest<COUNT, false>(M, nodes, f3(M.cam_o), ld, recip(ld), closest, h, stack, work) && h.t < closest) {
            closest = h.t; who = m; best = h;
        }
    }
    F3 value = f3(50.0f, 50.0f, 50.0f);                                   // scene.c:156
    if (who >= 0) {
        const MeshArgs& M = A.mesh[who];
        out.hits++;
        if (s == 0) out.hit_id = (int32_t)(((uint32_t)who << 24) | best.tri);
        // ---- attributes and texel of the hit (scene.c:86-106, mesh.c:138-149) ----
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
        const F3 tex = f3((float)(texel & 0xFFu) / 255.0f, (float)((texel >> 8) & 0xFFu) / 255.0f,
                          (float)((texel >> 16) & 0xFFu) / 255.0f);
        // ---- one shadow ray per light (scene.c:117-149) ----
        const F3 so = add(add(wo, scale(wd, closest)), scale(n, 0.001f));
        F3 color = f3(0.0f, 0.0f, 0.0f);
        for (int l = 0; l < X.n_lights; l++) {
            bool dark = false;
            // Only the colour leaves a sampled frame (and, when counting, the number of blocked rays): where max(0.2, n.L) is 0.2
            // already, scene.c:143-144 gives 0.2 blocked or not, and the shadow ray towards this light is not sent.
            const float facing = dot(n, f3(X.light_dir[l]));
            const int n_shadow = (!COUNT && !Party::kVerify && !(facing > 0.2f)) ? 0 : A.n_mesh;
            if (COUNT && !(facing > 0.2f)) work.idle++;
            for (int m = 0; m < n_shadow && !dark; m++) {
                const MeshArgs& S = A.mesh[m];
                const F3 lo = S.w_mode == 2 ? xform_point_w1(S.inv, so) : xform_point(S.inv, S.wrow, so);
                dark = party.template any<COUNT, false>(S, nodes, lo, f3(X.lit_d[m][l]), f3(X.lit_inv[m][l]), stack, work);
            }
            float diffuse = 0.2f;
            if (!dark) diffuse = fmaxf(diffuse, facing);
            else out.darks++;
            const F3 term = scale(mul(tex, f3(X.light_color[l])), diffuse);
            color = l == 0 ? term : add(color, term);
        }
        value = f3(fminf(color.x * 255.0f, 255.0f), fminf(color.y * 255.0f, 255.0f), fminf(color.z * 255.0f, 255.0f));
    }
    return value;
}

// Sum of the sample values in order -> bytes.
RT_HD void finish_sampled(const SampleArgs& X, F3 acc, SampledOut& out) {
    if (X.spp > 1) { const float n = (float)X.spp; acc = f3(acc.x / n, acc.y / n, acc.z / n); }
    out.r = byte_of(acc.x); out.g = byte_of(acc.y); out.b = byte_of(acc.z);
}

template <bool COUNT, class Stack, class Nodes, class Party>
RT_HD SampledOut shade_sampled_pixel(const FrameArgs& A, const SampleArgs& X, const Nodes& nodes, const Party& party, int px, int py,
                                     Stack& stack, Work& work) {
    SampledOut out;
    out.hits = 0; out.darks = 0; out.hit_id = -1;
    const int row = frame_row(A, py);          // the pixel's row in the whole frame
    F3 acc = f3(0.0f, 0.0f, 0.0f);
    for (int s = 0; s < X.spp; s++) {
        const F3 value = sample_value<COUNT>(A, X, nodes, party, px, row, s, stack, work, out);
        acc = s == 0 ? value : add(acc, value);
    }
    finish_sampled(X, acc, out);
    return out;
}

}  // namespace rt
