// rt_binned.hip -- instanced scenes rendered instance by instance (gfx950).
//
// The nested-loop kernel (render_instanced, rt_kernels.hip) gives every lane its own walk through the hierarchy over the
// instances and its own walks inside them: the lanes of a wave sit in different instances at different depths and fewer
// than half of them are active per instruction.  Here the work is turned round and split into passes:
//   * root-test passes: a wave takes ONE instance and some 8x8 tiles of that instance's screen rectangle (outside of
//     which its root test fails for certain: instance_screen_rect).  The instance's constants -- inverse matrix, root box --
//     are wave-uniform and live in scalar registers.  Every pixel gets the reference's exact root test in the instance's
//     space (ray.c:5-22, bvh.c:82); the pixels that pass are appended to the instance's bucket in HBM (one atomic add
//     per wave reserves the places, ballot + popcount give each lane its own);
//   * walk passes: a wave takes 64 consecutive entries of ONE instance's bucket -- a full wave of rays that are
//     neighbours in the instance's own space -- and walks the instance's tree with them.  A hit is merged into the pixel's
//     64-bit key (float bits of t << 32 | list index) with an atomic minimum, which is the reference's rule -- strictly
//     closer wins, so of equal distances the earlier list entry stays (scene.c:82) -- in any order of arrival; triangle
//     and barycentrics go to a record of their own per (instance, pixel).
// Camera rays: `near` finds, per pixel, the instance whose root box the ray enters first; `sort` puts every passing
// (instance, pixel) pair into the instance's bucket, the pixel's nearest-entry instance from the front, the others from the
// back; `first` walks the front parts (one walk per pixel, all of them useful); `rest` walks the back parts against the
// distances now known, so that boxes behind a hit are skipped as a front-to-back walk would skip them.  Then one pass
// per pixel computes normal, texel and shadow ray origin of the winner, the shadow rays get a root-test pass and an
// any-hit walk pass of the same kind, and a last pass writes the colours.
// Results do not depend on any of this: every (instance, pixel) pair whose exact root test passes is either walked or
// skipped because its box starts beyond cull_limit(closest hit), the nested kernel's own condition.
#include <hip/hip_runtime.h>

#include "rt_kernels.h"
#include "rt_pixel.h"
#include "rt_lds_stack.h"

namespace rt {

namespace {

#define RT_CONST_AS __attribute__((address_space(4)))

// Wave-uniform reads through the scalar cache: the record arrives in scalar registers, one read for the whole wave.
template <class T>
__device__ __forceinline__ T read_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "whole words");
    union { T t; uint32_t w[sizeof(T) / 4]; } u;
    const RT_CONST_AS uint32_t* s = (const RT_CONST_AS uint32_t*)(const RT_CONST_AS char*)p;
    RT_UNROLL
    for (size_t i = 0; i < sizeof(T) / 4; i++) u.w[i] = s[i];
    return u.t;
}

// An instance's constants in scalar registers.  The flat-mesh grid is switched off: it indexes the root box by an axis
// known only at run time, which would put the whole record into per-lane memory.
__device__ __forceinline__ MeshArgs instance_constants(const MeshArgs* p) {
    MeshArgs M = read_uniform(p);
    M.grid = nullptr; M.grid_axis = 3u;
    M.res_pairs = kNotResident; M.res_tris = kNotResident;
    return M;
}

__device__ __forceinline__ F3 camera_dir(const FrameArgs& A, int px, int py) {
    const float fx = ((float)px + 0.5f) / (float)A.width;                        // scene.c:61
    const float fy = ((float)py + 0.5f) / (float)A.full_height;                  // scene.c:62
    const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale;                    // camera.c:18
    const float ry = (1.0f - 2.0f * fy) * A.scale;                               // camera.c:19
    return unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));
}

__device__ __forceinline__ float key_t(unsigned long long key) { return __uint_as_float((uint32_t)(key >> 32)); }

// Entry distance as an unsigned number that orders like the float (negative entries -- the camera inside the box -- count as 0).
__device__ __forceinline__ uint32_t entry_bits(float entry) { return entry > 0.0f ? __float_as_uint(entry) : 0u; }

__device__ __forceinline__ void tile_of(const BinRect& R, bool shadow, uint32_t k, int& tx, int& ty) {
    const int x0 = shadow ? R.sx0 : R.x0, y0 = shadow ? R.sy0 : R.y0, w = (shadow ? R.sx1 : R.x1) - x0;
    const int row = (int)k / w;
    tx = x0 + ((int)k - row * w); ty = y0 + row;
}

// Places for the lanes with `want` in a bucket whose fill count is *counter: one atomic add per wave; returns the lane's place.
__device__ __forceinline__ uint32_t reserve(uint32_t* counter, bool want) {
    const unsigned long long votes = __ballot(want);
    if (votes == 0) return 0;
    const uint32_t lane = threadIdx.x & 63;
    const int leader = __builtin_ctzll(votes);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(votes));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
}

}  // namespace

// ---- keys to "no hit", entries to "none", buckets empty ------------------------------------------------------------------
__global__ void __launch_bounds__(256) binned_clear(unsigned long long* __restrict__ keys, unsigned long long* __restrict__ nearest, size_t n,
                                                  uint32_t* __restrict__ counts, uint32_t n_counts) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { keys[i] = kBinNoHit; nearest[i] = ~0ull; }
    if (i < n_counts) counts[i] = 0;
}

// ---- camera rays, root tests ---------------------------------------------------------------------------------------------
// SORT = false (near): per pixel the minimum of (entry distance of the instance's root box, list index) over the instances
//                      whose exact root test the ray passes.
// SORT = true: every passing pair into its instance's bucket: the pixel's nearest-entry instance from the front, the others
//              from the back.
template <bool SORT>
__global__ void __launch_bounds__(64) binned_camera_roots(const FrameArgs A, const InstancedArgs I, const BinArgs B) {
    const uint32_t lane = threadIdx.x;
    const BinItem item = read_uniform(&B.items[blockIdx.x]);
    const uint32_t inst = item.inst;
    const MeshArgs M = instance_constants(&I.inst[inst]);
    const BinRect R = read_uniform(&B.rects[inst]);
    const F3 lo = f3(M.cam_o);                              // ray.c:19: the same for every pixel
    const uint32_t cap = (uint32_t)((R.x1 - R.x0) * (R.y1 - R.y0)) * 64u;
    for (uint32_t k = item.first; k < item.first + item.count; k++) {
        int tx, ty;
        tile_of(R, false, k, tx, ty);
        const int px = tx * 8 + (int)(lane & 7), py = ty * 8 + (int)(lane >> 3);
        bool first = false, rest = false;
        if (px < A.width && py < A.height) {
            const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
            const F3 ld = unit(xform_vector<4>(M.inv, camera_dir(A, px, py)));   // ray.c:20,22
            float entry;
            if (slab_t(lo, recip(ld), M.root_min, M.root_max, entry)) {           // bvh.c:82 at the root
                const unsigned long long mine = ((unsigned long long)entry_bits(entry) << 32) | inst;
                if (!SORT) atomicMin(&B.nearest[p], mine);
                else { first = B.nearest[p] == mine; rest = !first; }
            }
        }
        if (SORT) {
            const uint32_t pixel = (uint32_t)px | ((uint32_t)py << 16);
            const uint32_t at_f = reserve(&B.counts[inst], first);
            if (first) B.bucket[(size_t)R.slot0 + at_f] = pixel;
            const uint32_t at_r = reserve(&B.counts[B.n_inst + inst], rest);
            if (rest) B.bucket[(size_t)R.slot0 + (cap - 1u - at_r)] = pixel;
        }
    }
}

// ---- camera rays, walks --------------------------------------------------------------------------------------------------
// REST = false: the front part of the instance's bucket (pixels whose ray enters this instance's box first).
// REST = true: the back part, each pair culled against the closest hit its pixel has by now.
template <bool REST>
__global__ void __launch_bounds__(64, 7) binned_camera_walk(const FrameArgs A, const InstancedArgs I, const SceneArgs S, const BinArgs B) {
    extern __shared__ uint32_t lds[];                       // stack: S.stack_entries x 64
    const uint32_t lane = threadIdx.x;
    const BinItem item = read_uniform(&B.items[blockIdx.x]);
    const uint32_t inst = item.inst;
    const uint32_t n = read_uniform(&B.counts[(REST ? B.n_inst : 0u) + inst]);
    if (item.first * 64u >= n) return;                      // this instance's bucket ends before this wave's part
    LdsStack<64> stack; stack.init(&lds[lane]);
    GlobalNodes nodes; nodes.all = S.pairs;
    const MeshArgs M = instance_constants(&I.inst[inst]);
    const BinRect R = read_uniform(&B.rects[inst]);
    const F3 lo = f3(M.cam_o);
    const bool leaf_root = (M.root_ref & kLeafBit) != 0;    // such a mesh honours the caller's bound inside the leaf (bvh.c:85): it gets none
    const int rect_w = R.x1 - R.x0;
    const uint32_t cap = (uint32_t)(rect_w * (R.y1 - R.y0)) * 64u;
    Work work; work.slabs = 0; work.tris = 0;
    for (uint32_t b = item.first; b < item.first + item.count && b * 64u < n; b++) {
        const uint32_t e = b * 64u + lane;
        if (e >= n) continue;
        const uint32_t pixel = B.bucket[(size_t)R.slot0 + (REST ? cap - 1u - e : e)];
        const int px = (int)(pixel & 0xFFFFu), py = (int)(pixel >> 16);
        const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
        const F3 ld = unit(xform_vector<4>(M.inv, camera_dir(A, px, py)));
        const unsigned long long seen = B.keys[p];
        const float bound = leaf_root ? 1e30f : key_t(seen);
        Hit h = no_hit();
        // the walk repeats the root test and, with a bound, skips the instance if its box starts beyond cull_limit(bound)
        if (Solo().template closest<false, false>(M, nodes, lo, ld, recip(ld), bound, h, stack, work) && h.t < 1e30f) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(h.t) << 32) | inst;
            if (key < seen) {                                // `seen` only ever goes down: what fails here fails later too
                const int tx = px >> 3, ty = py >> 3;
                BinHit rec; rec.tri = h.tri; rec.u = h.u; rec.v = h.v;
                B.hits[(size_t)R.slot0 + (size_t)((ty - R.y0) * rect_w + (tx - R.x0)) * 64 + (size_t)((py & 7) * 8 + (px & 7))] = rec;
                atomicMin(&B.keys[p], key);
            }
        }
    }
}

// ---- the winner's surface point: attributes, texel, shadow ray origin (scene.c:86-122) ----------------------------------
__global__ void __launch_bounds__(256) binned_shade(const FrameArgs A, const InstancedArgs I, const BinArgs B, const OutputPlanes O,
                                                  int32_t* __restrict__ hit_tri) {
    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    if (px >= A.width || py >= A.height) return;
    const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
    const unsigned long long key = B.keys[p];
    const uint32_t who = (uint32_t)key;
    const float closest = key_t(key);
    if (O.hit_t) O.hit_t[p] = closest;
    if (who == 0xFFFFFFFFu) {                                                    // scene.c:156
        uint8_t* rgb = O.rgb + p * 3;
        rgb[0] = rgb[1] = rgb[2] = 50;
        if (O.hit_id) O.hit_id[p] = -1;
        if (hit_tri) hit_tri[p] = -1;
        if (O.shadow) O.shadow[p] = 255;
        B.dark[p] = 2;
        return;
    }
    const MeshArgs& M = I.inst[who];
    const BinRect& R = B.rects[who];
    const BinHit h = B.hits[(size_t)R.slot0 + (size_t)((ty - R.y0) * (R.x1 - R.x0) + (tx - R.x0)) * 64 + (size_t)lane];
    if (O.hit_id) O.hit_id[p] = (int32_t)who;
    if (hit_tri) hit_tri[p] = (int32_t)h.tri;

    const AttrRec T = read_attrs(M.attrs, h.tri);
    const float u = h.u, v = h.v;
    const float w = 1.0f - u - v;
    const float tu = w * T.t0[0] + u * T.t1[0] + v * T.t2[0];
    const float tv = w * T.t0[1] + u * T.t1[1] + v * T.t2[1];
    F3 n = unit(add(add(scale(f3(T.n0), w), scale(f3(T.n1), u)), scale(f3(T.n2), v)));
    n = unit(xform_vector<3>(M.nrm, n));                                         // ray.c:39
    const float su = tu - floorf(tu), sv = tv - floorf(tv);                      // mesh.c:138-149
    const int tex_w = M.tex_w, tex_h = M.tex_h;
    const int txl = (int)(su * (float)(tex_w - 1));
    const int tyl = (int)(sv * (float)(tex_h - 1));
    const uint32_t texel = read_word(M.texels, (size_t)(tyl * tex_w + txl));
    const F3 color = f3((float)(texel & 0xFFu) / 255.0f, (float)((texel >> 8) & 0xFFu) / 255.0f, (float)((texel >> 16) & 0xFFu) / 255.0f);
    const F3 wd = camera_dir(A, px, py);
    const F3 so = add(add(f3(A.cam_pos), scale(wd, closest)), scale(n, 0.001f));  // scene.c:118-122
    BinPixel out;
    out.so[0] = so.x; out.so[1] = so.y; out.so[2] = so.z;
    out.lit = fmaxf(0.2f, dot(n, f3(A.light_dir)));                              // scene.c:115,143-144
    const F3 c = mul(color, f3(A.light_color));                                  // scene.c:148
    out.color[0] = c.x; out.color[1] = c.y; out.color[2] = c.z; out.pad = 0;
    B.pixels[p] = out;
    B.dark[p] = 0;
}

// ---- shadow rays (scene.c:124-139): any instance blocks ------------------------------------------------------------------
__device__ __forceinline__ F3 shadow_origin(const MeshArgs& M, const BinPixel* P) {
    const F3 so = f3(P->so[0], P->so[1], P->so[2]);
    return M.w_mode == 2 ? xform_point_w1(M.inv, so) : xform_point(M.inv, M.wrow, so);             // ray.c:19
}

__global__ void __launch_bounds__(64) binned_shadow_roots(const FrameArgs A, const InstancedArgs I, const BinArgs B) {
    const uint32_t lane = threadIdx.x;
    const BinItem item = read_uniform(&B.items[B.n_items + blockIdx.x]);
    const uint32_t inst = item.inst;
    const MeshArgs M = instance_constants(&I.inst[inst]);
    const BinRect R = read_uniform(&B.rects[inst]);
    const F3 sinv = f3(M.lit_inv);
    for (uint32_t k = item.first; k < item.first + item.count; k++) {
        int tx, ty;
        tile_of(R, true, k, tx, ty);
        const int px = tx * 8 + (int)(lane & 7), py = ty * 8 + (int)(lane >> 3);
        bool pass = false;
        if (px < A.width && py < A.height) {
            const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
            if (B.dark[p] == 0) {
                float entry;
                pass = slab_t(shadow_origin(M, &B.pixels[p]), sinv, M.root_min, M.root_max, entry);   // bvh.c:82 at the root
            }
        }
        const uint32_t at = reserve(&B.counts[2u * B.n_inst + inst], pass);
        if (pass) B.sbucket[(size_t)R.sslot0 + at] = (uint32_t)px | ((uint32_t)py << 16);
    }
}

__global__ void __launch_bounds__(64, 8) binned_shadow_walk(const FrameArgs A, const InstancedArgs I, const SceneArgs S, const BinArgs B) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    const BinItem item = read_uniform(&B.items[B.n_items + blockIdx.x]);
    const uint32_t inst = item.inst;
    const uint32_t n = read_uniform(&B.counts[2u * B.n_inst + inst]);
    if (item.first * 64u >= n) return;
    LdsStack<64> stack; stack.init(&lds[lane]);
    GlobalNodes nodes; nodes.all = S.pairs;
    const MeshArgs M = instance_constants(&I.inst[inst]);
    const BinRect R = read_uniform(&B.rects[inst]);
    const F3 sd = f3(M.lit_d), sinv = f3(M.lit_inv);
    Work work; work.slabs = 0; work.tris = 0;
    for (uint32_t b = item.first; b < item.first + item.count && b * 64u < n; b++) {
        const uint32_t e = b * 64u + lane;
        if (e >= n) continue;
        const uint32_t pixel = B.sbucket[(size_t)R.sslot0 + e];
        const size_t p = (size_t)(pixel >> 16) * (size_t)A.width + (size_t)(pixel & 0xFFFFu);
        if (B.dark[p] != 0) continue;                        // another instance has blocked this pixel's light meanwhile
        if (Solo().template any<false, false>(M, nodes, shadow_origin(M, &B.pixels[p]), sd, sinv, stack, work)) B.dark[p] = 1;
    }
}

// ---- colours (scene.c:141-154) -------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) binned_final(const FrameArgs A, const BinArgs B, const OutputPlanes O) {
    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    if (px >= A.width || py >= A.height) return;
    const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
    const uint8_t dark = B.dark[p];
    if (dark == 2) return;                                                       // no hit: written by binned_shade
    const BinPixel P = B.pixels[p];
    const float diffuse = dark ? 0.2f : P.lit;
    const F3 c = scale(f3(P.color), diffuse);                                    // scene.c:149
    uint8_t* rgb = O.rgb + p * 3;
    rgb[0] = (uint8_t)(fminf(c.x * 255.0f, 255.0f));                             // scene.c:152-154 (truncation)
    rgb[1] = (uint8_t)(fminf(c.y * 255.0f, 255.0f));
    rgb[2] = (uint8_t)(fminf(c.z * 255.0f, 255.0f));
    if (O.shadow) O.shadow[p] = dark;
}

hipError_t launch_instanced_binned(const FrameArgs& A, const InstancedArgs& I, const SceneArgs& S, const OutputPlanes& O, int32_t* hit_tri,
                                   const BinArgs& B, hipStream_t stream) {
    if (A.width <= 0 || A.height <= 0 || A.width > 65535 || A.height > 65535 || !O.rgb) return hipErrorInvalidValue;
    if (S.stack_entries < 2 || S.stack_entries > (uint32_t)kStackDepth) return hipErrorInvalidValue;
    if (!B.keys || !B.nearest || !B.pixels || !B.dark || !B.counts) return hipErrorInvalidValue;
    if ((B.n_items || B.n_shadow_items) && (!B.items || !B.rects || !B.hits || !B.bucket || !B.sbucket || !I.inst)) return hipErrorInvalidValue;
    const size_t pixels = (size_t)A.width * (size_t)A.height;
    const size_t clear_n = pixels > 3 * (size_t)B.n_inst ? pixels : 3 * (size_t)B.n_inst;
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    const size_t lds = (size_t)S.stack_entries * 64 * sizeof(uint32_t);
    hipLaunchKernelGGL(binned_clear, dim3((unsigned)((clear_n + 255) / 256)), dim3(256), 0, stream, B.keys, B.nearest, pixels, B.counts, 3u * B.n_inst);
    if (B.n_items) {
        hipLaunchKernelGGL((binned_camera_roots<false>), dim3(B.n_items), dim3(64), 0, stream, A, I, B);
        hipLaunchKernelGGL((binned_camera_roots<true>), dim3(B.n_items), dim3(64), 0, stream, A, I, B);
        hipLaunchKernelGGL((binned_camera_walk<false>), dim3(B.n_items), dim3(64), lds, stream, A, I, S, B);
        hipLaunchKernelGGL((binned_camera_walk<true>), dim3(B.n_items), dim3(64), lds, stream, A, I, S, B);
    }
    hipLaunchKernelGGL(binned_shade, dim3((tiles + 3) / 4), dim3(256), 0, stream, A, I, B, O, hit_tri);
    if (B.n_shadow_items) {
        hipLaunchKernelGGL(binned_shadow_roots, dim3(B.n_shadow_items), dim3(64), 0, stream, A, I, B);
        hipLaunchKernelGGL(binned_shadow_walk, dim3(B.n_shadow_items), dim3(64), lds, stream, A, I, S, B);
    }
    hipLaunchKernelGGL(binned_final, dim3((tiles + 3) / 4), dim3(256), 0, stream, A, B, O);
    return hipGetLastError();
}

}  // namespace rt
