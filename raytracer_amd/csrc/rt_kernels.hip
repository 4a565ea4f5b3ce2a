// rt_kernels.hip -- gfx950 kernels of the frame loop and their launchers.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (float results are compared
// bit-for-bit with a CPU build that has no FMA contraction).
//
// Variants (LaunchConfig::variant).  0 and 3 are the product; 1, 2 and 4..9 are experiments that lost against them (measurements in
// the repository's design notes) and are compiled only with -DRT_EXPERIMENTAL_VARIANTS (make EXPERIMENTAL=1) -- without it their
// ids run variant 0.
//   0  render_direct   one thread per pixel, one 8x8 tile per wave, nodes read through the
//                      vector cache.  Works for any scene the packer accepts.
//   1  render_resident persistent workgroups, one per CU, 16 waves each.  The PairNodes of the
//                      whole scene are staged into LDS once per workgroup, next to the per-lane
//                      traversal stacks; waves pull 8x8 tiles from an atomic counter until the
//                      frame is done.  Used when nodes + stacks fit in the CU's 160 KB of LDS.
//   2  render_resident with static tile assignment (wave w of W takes tiles w, w+W, ...).
//   3  render_tiles    variant 0 plus cooperative rendering of the screen area of big meshes:
//                      there a wave renders 8 pixels, each by 8 lanes that share the ray's
//                      traversal stack, and those waves are launched first.
//   4  render_pass1/2  two passes: one lane per pixel with a budget of node visits, then the pixels
//                      that ran out of budget cooperatively (8 lanes per ray).
//   5..9 render_lean   variant 0 with a short stack / other leaf batch sizes / occupancy hints.
// rt_render_frames uses render_batch (several frames per launch), see below.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rt_kernels.h"
#include "rt_pixel.h"
#include "rt_instanced.h"
#include "rt_lds_stack.h"
#include "rt_instanced_wave.h"
#include "rt_sampled.h"

namespace rt {

// PairNodes staged in LDS.
struct LdsNodes {
    const PairNode* all;
    __device__ __forceinline__ const PairNode& operator()(const MeshArgs& M, uint32_t ref) const { return all[M.pair_base + ref]; }
    __device__ __forceinline__ TriRec tri(const MeshArgs& M, uint32_t index) const { return read_tri(M.tris, index); }
};

// Pixel of (tile, lane): a wave owns an 8x8 tile, lanes row-major inside it, so the rays of a
// wave stay close in the image and take similar paths through the BVH.
__device__ __forceinline__ void tile_pixel(int tile, int lane, int tiles_x, int& px, int& py) {
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    px = tx * 8 + (lane & 7);
    py = ty * 8 + (lane >> 3);
}

__device__ __forceinline__ void store_pixel(const OutputPlanes& O, int width, int px, int py, const PixelOut& r) {
    const size_t p = (size_t)py * (size_t)width + (size_t)px;
    uint8_t* rgb = O.rgb + p * 3;
    rgb[0] = r.r; rgb[1] = r.g; rgb[2] = r.b;
    if (O.hit_id) O.hit_id[p] = r.hit_id;
    if (O.hit_t) O.hit_t[p] = r.hit_t;
    if (O.shadow) O.shadow[p] = r.shadow;
}

// Wave-level reduction of the work counters, one atomic per wave and counter.
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct LaneTotals {
    unsigned long long pixels, hits, darks, slabs, tris, idle;
};

// What a self-checking launch (RT_FLAG_VERIFY_SKIPS) found, summed over the wave; one atomic per wave and counter.
__device__ __forceinline__ void flush_checks(DeviceCounters* c, const Work& w) {
    const unsigned long long checks = wave_sum(w.checks), vis = wave_sum(w.bad_vis), svis = wave_sum(w.bad_shadow_vis);
    const unsigned long long entry = wave_sum(w.bad_entry), cull = wave_sum(w.bad_cull), inst = wave_sum(w.bad_inst);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&c->checks, checks);
        if (vis) atomicAdd(&c->bad_vis, vis);
        if (svis) atomicAdd(&c->bad_shadow_vis, svis);
        if (entry) atomicAdd(&c->bad_entry, entry);
        if (cull) atomicAdd(&c->bad_cull, cull);
        if (inst) atomicAdd(&c->bad_inst, inst);
    }
}

__device__ __forceinline__ void flush_counters(DeviceCounters* c, const LaneTotals& t) {
    const unsigned long long prim = wave_sum(t.pixels), hit = wave_sum(t.hits), dark = wave_sum(t.darks);
    const unsigned long long slabs = wave_sum(t.slabs), tris = wave_sum(t.tris), idle = wave_sum(t.idle);
    if ((threadIdx.x & 63) == 0) {
        if (idle) atomicAdd(&c->idle_shadow, idle);
        atomicAdd(&c->primary_rays, prim);
        atomicAdd(&c->shadow_rays, hit);
        atomicAdd(&c->hit_pixels, hit);
        atomicAdd(&c->shadowed_pixels, dark);
        atomicAdd(&c->slab_tests, slabs);
        atomicAdd(&c->tri_tests, tris);
    }
}

// ------------------------------------------------------------------------------------------
// Variant 0: one thread per pixel, scene read through the vector cache.
// ------------------------------------------------------------------------------------------
// VERIFY: the self-checking party walks every ray (SoloVerify, rt_pixel.h); the frame is the one the plain kernel renders.
template <int BLOCK, bool COUNT, bool FULL, bool VERIFY = false>
__global__ void __launch_bounds__(BLOCK) render_direct(const FrameArgs A, const SceneArgs S, const OutputPlanes O) {
    __shared__ uint32_t lds_stack[kStackDepth * BLOCK];
    LdsStack<BLOCK> stack;
    stack.init(&lds_stack[threadIdx.x]);
    GlobalNodes nodes; nodes.all = S.pairs;

    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    int px, py;
    tile_pixel(tile, threadIdx.x & 63, tiles_x, px, py);
    const bool active = px < A.width && py < A.height;

    LaneTotals tot = {0, 0, 0, 0, 0, 0};
    const unsigned long long t_start = O.tile_clock ? wall_clock64() : 0ull;
    Work work;
    if (active) {
        PixelOut r;
        if (VERIFY) r = shade_pixel<COUNT, FULL, false>(A, nodes, SoloVerify(), px, py, stack, work);
        else r = shade_pixel<COUNT, FULL, false>(A, nodes, Solo(), px, py, stack, work, O.shadow == nullptr);
        store_pixel(O, A.width, px, py, r);
        tot.pixels = 1; tot.hits = r.hit; tot.darks = r.dark; tot.slabs = work.slabs; tot.tris = work.tris; tot.idle = work.idle;
    }
    if (O.tile_clock && (threadIdx.x & 63) == 0 && tile < tiles_x * ((A.height + 7) >> 3)) {
        O.tile_clock[2 * (size_t)tile] = t_start;
        O.tile_clock[2 * (size_t)tile + 1] = wall_clock64();
    }
    if (COUNT || VERIFY) flush_counters(O.counters, tot);
    if (VERIFY) flush_checks(O.counters, work);
}

template <int BLOCK>
static hipError_t launch_direct(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, bool count, bool full, bool verify, hipStream_t stream) {
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    const int waves_per_block = BLOCK / 64;
    const int blocks = (tiles + waves_per_block - 1) / waves_per_block;
    if (verify)         hipLaunchKernelGGL((render_direct<BLOCK, false, false, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else if (count && full)  hipLaunchKernelGGL((render_direct<BLOCK, true, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else if (count)     hipLaunchKernelGGL((render_direct<BLOCK, true, false>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else if (full)      hipLaunchKernelGGL((render_direct<BLOCK, false, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else                hipLaunchKernelGGL((render_direct<BLOCK, false, false>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Sampled frames (rt_sampled.h): several camera rays per pixel, several lights.  One thread per
// pixel loops over its samples, so the lanes of a wave stay on neighbouring rays throughout.
// ------------------------------------------------------------------------------------------
// Seven waves per SIMD (71 registers, nothing spilled): 7.9 ms instead of 8.3 at five waves for the 4K / 16-sample / 2-light frame
// (mean of frames 0 and 45).
template <int BLOCK, bool COUNT, bool VERIFY = false>
__global__ void __launch_bounds__(BLOCK, VERIFY ? 1 : 7) render_sampled(const FrameArgs A, const SampleArgs X, const SceneArgs S, const OutputPlanes O) {
    extern __shared__ uint32_t lds_stack[];                      // S.stack_entries x BLOCK words
    LdsStack<BLOCK> stack;
    stack.init(&lds_stack[threadIdx.x]);
    GlobalNodes nodes; nodes.all = S.pairs;
    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    int px, py;
    tile_pixel(tile, threadIdx.x & 63, tiles_x, px, py);
    LaneTotals tot = {0, 0, 0, 0, 0, 0};
    Work work;
    if (px < A.width && py < A.height) {
        SampledOut r;
        if (VERIFY) r = shade_sampled_pixel<COUNT>(A, X, nodes, SoloVerify(), px, py, stack, work);
        else r = shade_sampled_pixel<COUNT>(A, X, nodes, Solo(), px, py, stack, work);
        const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
        uint8_t* rgb = O.rgb + p * 3;
        rgb[0] = r.r; rgb[1] = r.g; rgb[2] = r.b;
        if (O.hit_id) O.hit_id[p] = r.hit_id;
        tot.pixels = (unsigned long long)X.spp; tot.hits = r.hits; tot.darks = r.darks; tot.slabs = work.slabs; tot.tris = work.tris; tot.idle = work.idle;
    }
    if (VERIFY) flush_checks(O.counters, work);
    if (COUNT || VERIFY) {
        // shadow rays = hit samples x lights
        const unsigned long long prim = wave_sum(tot.pixels), hit = wave_sum(tot.hits), dark = wave_sum(tot.darks);
        const unsigned long long slabs = wave_sum(tot.slabs), tris = wave_sum(tot.tris), idle = wave_sum(tot.idle);
        if ((threadIdx.x & 63) == 0) {
            if (idle) atomicAdd(&O.counters->idle_shadow, idle);
            atomicAdd(&O.counters->primary_rays, prim);
            atomicAdd(&O.counters->shadow_rays, hit * (unsigned long long)X.n_lights);
            atomicAdd(&O.counters->hit_pixels, hit);
            atomicAdd(&O.counters->shadowed_pixels, dark);
            atomicAdd(&O.counters->slab_tests, slabs);
            atomicAdd(&O.counters->tri_tests, tris);
        }
    }
}

// The same frame with the samples of a tile spread over the four waves of a workgroup: wave w traces
// samples w, w + 4, ... of the tile's 64 pixels and leaves their values in LDS; wave 0 adds them up in
// sample order (the order is part of the result) and writes the pixel.  The longest dependent chain of a
// launch drops from spp samples to spp / 4, which matters when a launch has few tiles (a share of a
// frame split over GPUs) and at the end of every launch.
constexpr int kSplitWaves = 4;

template <bool COUNT>
__global__ void __launch_bounds__(64 * kSplitWaves) render_sampled_split(const FrameArgs A, const SampleArgs X, const SceneArgs S, const OutputPlanes O) {
    constexpr int BLOCK = 64 * kSplitWaves;
    extern __shared__ float lds_values[];                       // [spp][3][64], then S.stack_entries x BLOCK stack words
    uint32_t* lds_stack = reinterpret_cast<uint32_t*>(lds_values + (size_t)X.spp * 192);
    __shared__ uint32_t lds_counts[kSplitWaves][2][64];
    __shared__ int32_t lds_hit_id[64];
    LdsStack<BLOCK> stack;
    stack.init(&lds_stack[threadIdx.x]);
    GlobalNodes nodes; nodes.all = S.pairs;
    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int px, py;
    tile_pixel(tile, lane, tiles_x, px, py);
    const bool inside = px < A.width && py < A.height;
    LaneTotals tot = {0, 0, 0, 0, 0, 0};
    SampledOut out;
    out.hits = 0; out.darks = 0; out.hit_id = -1;
    Work work; work.slabs = 0; work.tris = 0;
    if (inside) {
        const int row = frame_row(A, py);
        for (int s = wave; s < X.spp; s += kSplitWaves) {
            const F3 v = sample_value<COUNT>(A, X, nodes, Solo(), px, row, s, stack, work, out);
            float* slot = lds_values + (size_t)s * 192 + lane;
            slot[0] = v.x; slot[64] = v.y; slot[128] = v.z;
        }
    }
    lds_counts[wave][0][lane] = out.hits; lds_counts[wave][1][lane] = out.darks;
    if (wave == 0) lds_hit_id[lane] = out.hit_id;
    __syncthreads();
    if (wave == 0 && inside) {
        F3 acc = f3(0.0f, 0.0f, 0.0f);
        for (int s = 0; s < X.spp; s++) {
            const float* slot = lds_values + (size_t)s * 192 + lane;
            const F3 v = f3(slot[0], slot[64], slot[128]);
            acc = s == 0 ? v : add(acc, v);
        }
        finish_sampled(X, acc, out);
        const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
        uint8_t* rgb = O.rgb + p * 3;
        rgb[0] = out.r; rgb[1] = out.g; rgb[2] = out.b;
        if (O.hit_id) O.hit_id[p] = lds_hit_id[lane];
        uint32_t hits = 0, darks = 0;
        for (int w = 0; w < kSplitWaves; w++) { hits += lds_counts[w][0][lane]; darks += lds_counts[w][1][lane]; }
        tot.pixels = (unsigned long long)X.spp; tot.hits = hits; tot.darks = darks;
    }
    if (COUNT) {
        tot.slabs = work.slabs; tot.tris = work.tris; tot.idle = work.idle;
        const unsigned long long prim = wave_sum(tot.pixels), hit = wave_sum(tot.hits), dark = wave_sum(tot.darks);
        const unsigned long long slabs = wave_sum(tot.slabs), tris = wave_sum(tot.tris), idle = wave_sum(tot.idle);
        if (lane == 0) {
            if (idle) atomicAdd(&O.counters->idle_shadow, idle);
            if (wave == 0) {
                atomicAdd(&O.counters->primary_rays, prim);
                atomicAdd(&O.counters->shadow_rays, hit * (unsigned long long)X.n_lights);
                atomicAdd(&O.counters->hit_pixels, hit);
                atomicAdd(&O.counters->shadowed_pixels, dark);
            }
            atomicAdd(&O.counters->slab_tests, slabs);
            atomicAdd(&O.counters->tri_tests, tris);
        }
    }
}

hipError_t launch_sampled(const FrameArgs& A, const SampleArgs& X, const SceneArgs& S, const OutputPlanes& O, int block, int split, hipStream_t stream) {
    if (A.width <= 0 || A.height <= 0 || X.spp < 1 || X.spp > kMaxSamples || X.n_lights < 1 || X.n_lights > kMaxLights || !O.rgb)
        return hipErrorInvalidValue;
    const bool verify = (A.flags & RT_FLAG_VERIFY_SKIPS) != 0;
    const bool count = (A.flags & RT_FLAG_COUNT_WORK) != 0 && !verify;
    if ((count || verify) && !O.counters) return hipErrorInvalidValue;
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    if (S.stack_entries < 2 || S.stack_entries > (uint32_t)kStackDepth) return hipErrorInvalidValue;
    const size_t stack_bytes = (size_t)S.stack_entries * 64 * sizeof(uint32_t);     // per wave
    if (verify) {
        hipLaunchKernelGGL((render_sampled<64, false, true>), dim3(tiles), dim3(64), stack_bytes, stream, A, X, S, O);
        return hipGetLastError();
    }
    const size_t dyn = (size_t)X.spp * 192 * sizeof(float) + stack_bytes * kSplitWaves;
    const size_t fixed = 4096;
    // split = 1: when the launch has too few tiles to hide a tile's sample loop behind other tiles (measured at
    // 16 samples: 65 k tiles 4.7 ms split / 6.9 ms looped, 130 k tiles 9.4 / 8.3); 2: always.
    const bool want = split == 2 || (split == 1 && tiles <= 100000);
    if (want && X.spp >= kSplitWaves && dyn + fixed <= 64 * 1024) {      // 64 KB of LDS per workgroup without opting in to more
        if (count) hipLaunchKernelGGL((render_sampled_split<true>), dim3(tiles), dim3(64 * kSplitWaves), dyn, stream, A, X, S, O);
        else       hipLaunchKernelGGL((render_sampled_split<false>), dim3(tiles), dim3(64 * kSplitWaves), dyn, stream, A, X, S, O);
    } else if (block == 256) {
        const int blocks = (tiles + 3) / 4;
        if (count) hipLaunchKernelGGL((render_sampled<256, true>), dim3(blocks), dim3(256), stack_bytes * 4, stream, A, X, S, O);
        else       hipLaunchKernelGGL((render_sampled<256, false>), dim3(blocks), dim3(256), stack_bytes * 4, stream, A, X, S, O);
    } else {
        if (count) hipLaunchKernelGGL((render_sampled<64, true>), dim3(tiles), dim3(64), stack_bytes, stream, A, X, S, O);
        else       hipLaunchKernelGGL((render_sampled<64, false>), dim3(tiles), dim3(64), stack_bytes, stream, A, X, S, O);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Instanced scenes (rt_instanced.h): per-instance MeshArgs in a device array, a hierarchy over
// the instances' world boxes.  Two per-lane stacks in LDS: one for the walk over the hierarchy,
// one for the walk inside an instance's BVH.
// ------------------------------------------------------------------------------------------
// Five waves per SIMD (90 registers, nothing spilled): 1.25 against 1.31 ms at four; six and seven waves spill and lose.
template <int BLOCK, bool COUNT, bool VERIFY = false>
__global__ void __launch_bounds__(BLOCK, VERIFY ? 1 : 5) render_instanced(const FrameArgs A, const InstancedArgs I, const SceneArgs S, const OutputPlanes O,
                                                          int32_t* __restrict__ hit_tri, const int tlas_entries) {
    // both stacks sized for the trees at hand (entries x BLOCK words each): LDS per wave decides how many
    // waves a CU holds here
    extern __shared__ uint32_t lds_stacks[];
    LdsStack<BLOCK> stack, tstack;
    tstack.init(&lds_stacks[threadIdx.x]);
    stack.init(&lds_stacks[(size_t)tlas_entries * BLOCK + threadIdx.x]);
    GlobalNodes nodes; nodes.all = S.pairs;
    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    int px, py;
    tile_pixel(tile, threadIdx.x & 63, tiles_x, px, py);
    LaneTotals tot = {0, 0, 0, 0, 0, 0};
    Work work;
    if (px < A.width && py < A.height) {
        InstancedOut r;
        if (VERIFY) r = shade_instanced_pixel<COUNT>(A, I, nodes, SoloVerify(), px, py, stack, tstack, work);
        else r = shade_instanced_pixel<COUNT>(A, I, nodes, Solo(), px, py, stack, tstack, work, O.shadow == nullptr);
        const size_t p = (size_t)py * (size_t)A.width + (size_t)px;
        uint8_t* rgb = O.rgb + p * 3;
        rgb[0] = r.r; rgb[1] = r.g; rgb[2] = r.b;
        if (O.hit_id) O.hit_id[p] = r.inst;
        if (hit_tri) hit_tri[p] = r.tri;
        if (O.hit_t) O.hit_t[p] = r.hit_t;
        if (O.shadow) O.shadow[p] = r.shadow;
        tot.pixels = 1; tot.hits = r.hit; tot.darks = r.dark; tot.slabs = work.slabs; tot.tris = work.tris; tot.idle = work.idle;
    }
    if (COUNT || VERIFY) flush_counters(O.counters, tot);
    if (VERIFY) flush_checks(O.counters, work);
}

// ------------------------------------------------------------------------------------------
// Instanced scenes, scheduled per wavefront (rt_instanced_wave.h).  One wave per workgroup; a wave owns TILES
// consecutive 8x8 tiles and keeps its 64 lanes supplied with their pixels.  Each round the wave counts its lanes
// by state and executes the step most of them wait for; box steps, which dominate, run in short bursts.
// ------------------------------------------------------------------------------------------
template <int TILES>
__global__ void __launch_bounds__(64) render_instanced_wave(const FrameArgs A, const InstancedArgs I, const SceneArgs S, const OutputPlanes O,
                                                            int32_t* __restrict__ hit_tri, const int n_tiles) {
    extern __shared__ uint32_t lds_stacks[];                     // entries x 64 words: one stack per lane for both levels
    const int lane = threadIdx.x;
    const int tiles_x = (A.width + 7) >> 3;
    const int first_tile = blockIdx.x * TILES;
    const int n_pix = (n_tiles - first_tile < TILES ? n_tiles - first_tile : TILES) * 64;
    const unsigned long long below = (1ull << lane) - 1ull;
    WaveLane<64> L;
    L.stack.init(&lds_stacks[lane]);
    L.pixel = -1; L.ref = kRefEnd; L.shadow_mode = true; L.in_inst = false; L.who = -2;
    int next = 0;                                                // wave-uniform: first pixel of the strip not handed out yet

    // Hands the lanes in `want` their next pixels.  A pixel outside the frame (ragged right / bottom tiles) becomes a
    // pixel that is already through (who = -2): its lane asks again in the next store step.
    auto refill = [&](bool want) {
        const unsigned long long votes = __ballot(want);
        const int idx = next + __popcll(votes & below);
        next += __popcll(votes);
        if (!want) return;
        if (idx >= n_pix) { L.pixel = -1; return; }
        L.pixel = idx;
        tile_pixel(first_tile + (idx >> 6), idx & 63, tiles_x, L.px, L.py);
        if (L.px < A.width && L.py < A.height) start_pixel(L, A, I);
        else { L.shadow_mode = true; L.ref = kRefEnd; L.who = -2; }
    };
    refill(true);

    for (;;) {
        const int st = lane_state(L);
        const int n_box = __popcll(__ballot(st == W_BOX)), n_leaf = __popcll(__ballot(st == W_LEAF));
        const int n_switch = __popcll(__ballot(st == W_SWITCH)), n_shade = __popcll(__ballot(st == W_SHADE));
        const int n_store = __popcll(__ballot(st == W_STORE));
        if (n_box + n_leaf + n_switch + n_shade + n_store == 0) break;          // every lane is done: the strip is rendered
        int others = n_leaf > n_switch ? n_leaf : n_switch;
        others = others > n_shade ? others : n_shade;
        others = others > n_store ? others : n_store;
        if (n_box >= others) {
            // a burst of box steps: while at least as many lanes want one as want anything else
            for (int burst = 0; burst < 4; burst++) {
                const bool mine = L.pixel >= 0 && !(L.ref & kLeafBit);
                if ((int)__popcll(__ballot(mine)) < (others > 1 ? others : 1)) break;
                if (mine) step_box(L);
            }
        } else if (n_leaf == others) {
            if (st == W_LEAF) {
                step_leaf(L);
                if (L.dark) L.ref = kRefEnd;                      // a blocked shadow ray is through
            }
        } else if (n_switch == others) {
            if (st == W_SWITCH) step_switch(L, I, S);
        } else if (n_shade == others) {
            if (st == W_SHADE) step_shade(L, A, I);
        } else {
            const bool mine = st == W_STORE;
            if (mine && L.who != -2) {
                const size_t p = (size_t)L.py * (size_t)A.width + (size_t)L.px;
                uint8_t* rgb = O.rgb + p * 3;
                if (L.who < 0) {
                    rgb[0] = rgb[1] = rgb[2] = 50;                                               // scene.c:156
                } else {
                    const F3 c = scale(L.color, L.dark ? 0.2f : L.lit);                          // scene.c:115,148-149
                    rgb[0] = colour_byte(c.x);                             // scene.c:152-154 (truncation)
                    rgb[1] = colour_byte(c.y);
                    rgb[2] = colour_byte(c.z);
                }
                if (O.hit_id) O.hit_id[p] = L.who < 0 ? -1 : L.who;
                if (hit_tri) hit_tri[p] = L.who < 0 ? -1 : (int32_t)L.btri;
                if (O.hit_t) O.hit_t[p] = L.closest;
                if (O.shadow) O.shadow[p] = L.who < 0 ? 255 : (L.dark ? 1 : 0);
            }
            refill(mine);
        }
    }
}

hipError_t launch_instanced(const FrameArgs& A, const InstancedArgs& I, const SceneArgs& S, const OutputPlanes& O, int32_t* hit_tri,
                            int tlas_entries, int block, int wave_tiles, hipStream_t stream) {
    if (A.width <= 0 || A.height <= 0 || !O.rgb || I.n_inst < 0 || (I.n_inst > 0 && !I.inst)) return hipErrorInvalidValue;
    if (tlas_entries < 2 || tlas_entries > kTlasStackDepth || S.stack_entries < 2 || S.stack_entries > (uint32_t)kStackDepth) return hipErrorInvalidValue;
    const bool verify = (A.flags & RT_FLAG_VERIFY_SKIPS) != 0;
    const bool count = (A.flags & RT_FLAG_COUNT_WORK) != 0 && !verify;
    if ((count || verify) && !O.counters) return hipErrorInvalidValue;
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    if (verify) {
        const size_t vlds = (size_t)(tlas_entries + (int)S.stack_entries) * 64 * sizeof(uint32_t);
        hipLaunchKernelGGL((render_instanced<64, false, true>), dim3(tiles), dim3(64), vlds, stream, A, I, S, O, hit_tri, tlas_entries);
        return hipGetLastError();
    }
    if (wave_tiles > 0 && !count) {
        // one stack for both levels: the hierarchy's entries, the end mark, what is left of a leaf of several instances, the
        // mark below an instance's nodes, the instance's entries
        const size_t wlds = (size_t)(tlas_entries + (int)S.stack_entries + 3) * 64 * sizeof(uint32_t);
        if (wave_tiles >= 8) hipLaunchKernelGGL((render_instanced_wave<8>), dim3((tiles + 7) / 8), dim3(64), wlds, stream, A, I, S, O, hit_tri, tiles);
        else if (wave_tiles >= 4) hipLaunchKernelGGL((render_instanced_wave<4>), dim3((tiles + 3) / 4), dim3(64), wlds, stream, A, I, S, O, hit_tri, tiles);
        else if (wave_tiles >= 2) hipLaunchKernelGGL((render_instanced_wave<2>), dim3((tiles + 1) / 2), dim3(64), wlds, stream, A, I, S, O, hit_tri, tiles);
        else hipLaunchKernelGGL((render_instanced_wave<1>), dim3(tiles), dim3(64), wlds, stream, A, I, S, O, hit_tri, tiles);
        return hipGetLastError();
    }
    const int threads = block == 256 ? 256 : 64;
    const size_t lds = (size_t)(tlas_entries + (int)S.stack_entries) * (size_t)threads * sizeof(uint32_t);
    if (block == 256) {
        const int blocks = (tiles + 3) / 4;
        if (count) hipLaunchKernelGGL((render_instanced<256, true>), dim3(blocks), dim3(256), lds, stream, A, I, S, O, hit_tri, tlas_entries);
        else       hipLaunchKernelGGL((render_instanced<256, false>), dim3(blocks), dim3(256), lds, stream, A, I, S, O, hit_tri, tlas_entries);
    } else {
        if (count) hipLaunchKernelGGL((render_instanced<64, true>), dim3(tiles), dim3(64), lds, stream, A, I, S, O, hit_tri, tlas_entries);
        else       hipLaunchKernelGGL((render_instanced<64, false>), dim3(tiles), dim3(64), lds, stream, A, I, S, O, hit_tri, tlas_entries);
    }
    return hipGetLastError();
}

#if defined(RT_EXPERIMENTAL_VARIANTS)
// ------------------------------------------------------------------------------------------
// Variants 5..9: variant 0 tuned for occupancy.  Frames are independent, so a caller that renders
// an animation can keep several frames in flight on different streams; the frame rate is then
// set by how many waves the chip holds and how long each one runs, not by the slowest tile.
// DEPTH = stack entries per lane (16 covers trees up to depth 14: 4 KB of LDS per wave instead
// of 8), LB = triangles fetched per leaf batch (12 VGPRs each), MINW = waves per SIMD the
// register allocation has to admit.
// ------------------------------------------------------------------------------------------
// Geometry source of the lean kernels: meshes marked resident are read from the workgroup's LDS
// copy, the others through the vector cache.  The mesh is the same for all lanes of a wave at any
// time, so the choice never diverges.
struct HybridNodes {
    const PairNode* global_all;
    const uint8_t* lds;          // the resident blob in LDS
    __device__ __forceinline__ PairNode operator()(const MeshArgs& M, uint32_t ref) const {
        if (M.res_pairs != kNotResident) return *reinterpret_cast<const PairNode*>(lds + M.res_pairs + (size_t)ref * sizeof(PairNode));
        return global_all[M.pair_base + ref];
    }
    __device__ __forceinline__ TriRec tri(const MeshArgs& M, uint32_t index) const {
        if (M.res_tris != kNotResident) return *reinterpret_cast<const TriRec*>(lds + M.res_tris + (size_t)index * sizeof(TriRec));
        return read_tri(M.tris, index);
    }
};

template <int BLOCK, int DEPTH, uint32_t LB, int MINW>
__global__ void __launch_bounds__(BLOCK, MINW) render_lean(const FrameArgs A, const SceneArgs S, const OutputPlanes O) {
    // [ stacks: DEPTH x BLOCK dwords ][ resident blob: S.resident_bytes ]
    extern __shared__ uint4 lds_dyn[];
    uint32_t* lds_stack = reinterpret_cast<uint32_t*>(lds_dyn);
    uint4* lds_blob = lds_dyn + (DEPTH * BLOCK) / 4;
    {
        const uint4* src = reinterpret_cast<const uint4*>(S.resident);
        const int n16 = (int)(S.resident_bytes / 16);
        for (int i = threadIdx.x; i < n16; i += BLOCK) lds_blob[i] = src[i];
        __syncthreads();
    }
    LdsStack<BLOCK> stack;
    stack.init(&lds_stack[threadIdx.x]);
    HybridNodes nodes; nodes.global_all = S.pairs; nodes.lds = reinterpret_cast<const uint8_t*>(lds_blob);
    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    int px, py;
    tile_pixel(tile, threadIdx.x & 63, tiles_x, px, py);
    const unsigned long long t_start = O.tile_clock ? wall_clock64() : 0ull;
    if (px < A.width && py < A.height) {
        Work work; work.slabs = 0; work.tris = 0;
        const PixelOut r = shade_pixel<false, false, false>(A, nodes, SoloT<LB>(), px, py, stack, work, O.shadow == nullptr);
        store_pixel(O, A.width, px, py, r);
    }
    if (O.tile_clock && (threadIdx.x & 63) == 0 && tile < tiles_x * ((A.height + 7) >> 3)) {
        O.tile_clock[2 * (size_t)tile] = t_start;
        O.tile_clock[2 * (size_t)tile + 1] = wall_clock64();
    }
}

template <int BLOCK, int DEPTH, uint32_t LB, int MINW>
static hipError_t launch_lean(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, hipStream_t stream) {
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    const int waves_per_block = BLOCK / 64;
    const int blocks = (tiles + waves_per_block - 1) / waves_per_block;
    const size_t lds = (size_t)DEPTH * BLOCK * sizeof(uint32_t) + S.resident_bytes;
    if (lds > 64 * 1024) return hipErrorInvalidValue;       // kResidentBudget keeps it below the default limit
    hipLaunchKernelGGL((render_lean<BLOCK, DEPTH, LB, MINW>), dim3(blocks), dim3(BLOCK), lds, stream, A, S, O);
    return hipGetLastError();
}

#endif  // RT_EXPERIMENTAL_VARIANTS

// ------------------------------------------------------------------------------------------
// Batched frames: one launch renders n_frames frames of the same size into consecutive images.
//
// A frame has a few tiles whose rays run several hundred microseconds (they cross a big mesh)
// while the bulk of the frame is done in far less; rendered alone, the frame lasts as long as
// those tiles.  Frames are independent, so a batch puts the long tiles of ALL its frames at the
// front of one grid (rows are scanned from first_row, the top of the big meshes' screen area,
// and wrap around) and the cheap tiles of all frames fill the machine behind them.  The grid
// takes `chunk` consecutive tiles of frame 0, the same tiles of frame 1, ... and then moves on to
// the next chunk, so that the waves resident on a CU at any time belong to few frames (their
// per-frame constants share the scalar cache).  The per-frame constants come from an array
// in device memory; the frame index is uniform per wave, so they are read with scalar loads.
// ------------------------------------------------------------------------------------------
// Waves per SIMD the register allocation has to admit.  Measured per launch of 8 frames: with two triangles fetched per leaf
// batch (71 registers) six waves 1.023 ms, seven 0.979, eight 0.970 with 28 bytes of scratch; with one triangle per batch (59
// registers, nothing spilled) eight waves 0.935 ms.  Since the walk can enter a mesh's sheared tree (rt_pixel.h) eight waves
// need 40 bytes of scratch per lane: eight 0.628 ms, seven (72 registers, nothing spilled) 0.607, six 0.614.
constexpr int kBatchWaves = 7;
template <int BLOCK, int DEPTH, uint32_t LB>
__global__ void __launch_bounds__(BLOCK, DEPTH <= 16 ? kBatchWaves : 5) render_batch(const FrameArgs* __restrict__ frames, const int n_frames, const SceneArgs S,
                                                      uint8_t* __restrict__ rgb_base, const unsigned long long frame_stride,
                                                      const int first_row, const int chunk, const int xcd_swizzle) {
    __shared__ uint32_t lds_stack[DEPTH * BLOCK];
    LdsStack<BLOCK> stack;
    stack.init(&lds_stack[threadIdx.x]);
    BatchNodes nodes; nodes.all = S.pairs;

    int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)));
    // Workgroups go to the 8 XCDs round robin.  With one wave per workgroup, transposing every run
    // of 64 waves as an 8x8 block gives each XCD 8 neighbouring tiles of a row instead of every
    // eighth tile, so the pixels of one cache line (five tiles wide) are written through one L2.
    if (BLOCK == 64 && xcd_swizzle) wave = (wave & ~63) | ((wave & 7) << 3) | ((wave >> 3) & 7);
    // `chunk` consecutive tiles of one frame, then the same tiles of the next frame, ...
    const int group = wave / (chunk * n_frames);
    const int within = wave - group * (chunk * n_frames);
    const int f = within / chunk;
    const int slot = group * chunk + (within - f * chunk);
    const FrameArgs& A = frames[f];
    const int tiles_x = (A.width + 7) >> 3, tiles_y = (A.height + 7) >> 3;
    if (slot >= tiles_x * tiles_y) return;
    int row = slot / tiles_x;
    const int col = slot - row * tiles_x;
    row += first_row >= 0 ? first_row : A.first_tile_row;      // one starting row for the batch, or each frame's own
    if (row >= tiles_y) row -= tiles_y;
    const int lane = threadIdx.x & 63;
    const int px = col * 8 + (lane & 7), py = row * 8 + (lane >> 3);
    if (px < A.width && py < A.height) {
        Work work; work.slabs = 0; work.tris = 0;
        const PixelOut r = shade_pixel<false, false, false>(A, nodes, SoloT<LB>(), px, py, stack, work, true);
        uint8_t* rgb = rgb_base + (size_t)f * frame_stride + ((size_t)py * (size_t)A.width + (size_t)px) * 3;
        rgb[0] = r.r; rgb[1] = r.g; rgb[2] = r.b;
    }
}

template <int BLOCK, int DEPTH>
static hipError_t launch_batch_t(const FrameArgs* d_frames, int n_frames, int width, int height, const SceneArgs& S, uint8_t* rgb,
                                 size_t frame_stride, int first_row, int chunk, int swizzle, hipStream_t stream) {
    const long long tiles = (long long)((width + 7) / 8) * ((height + 7) / 8);
    long long waves = (tiles + chunk - 1) / chunk * chunk * n_frames;
    waves = (waves + 63) / 64 * 64;
    const int waves_per_block = BLOCK / 64;
    const long long blocks = (waves + waves_per_block - 1) / waves_per_block;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((render_batch<BLOCK, DEPTH, kLeafBatch>), dim3((unsigned)blocks), dim3(BLOCK), 0, stream, d_frames, n_frames, S, rgb,
                       (unsigned long long)frame_stride, first_row, chunk, swizzle);
    return hipGetLastError();
}

hipError_t launch_batch(const FrameArgs* d_frames, int n_frames, int width, int height, const SceneArgs& S, uint8_t* rgb, size_t frame_stride,
                        int first_row, int chunk, int swizzle, int block, hipStream_t stream) {
    if (n_frames <= 0 || width <= 0 || height <= 0 || !d_frames || !rgb) return hipErrorInvalidValue;
    const int tiles_y = (height + 7) / 8;
    if (first_row >= tiles_y) first_row = 0;                   // < 0: every frame starts at its own FrameArgs::first_tile_row
    if (chunk < 1) chunk = 1;
    if (S.stack_entries <= 16) {
        if (block == 64) return launch_batch_t<64, 16>(d_frames, n_frames, width, height, S, rgb, frame_stride, first_row, chunk, swizzle, stream);
        return launch_batch_t<256, 16>(d_frames, n_frames, width, height, S, rgb, frame_stride, first_row, chunk, swizzle, stream);
    }
    if (block == 64) return launch_batch_t<64, kStackDepth>(d_frames, n_frames, width, height, S, rgb, frame_stride, first_row, chunk, swizzle, stream);
    return launch_batch_t<256, kStackDepth>(d_frames, n_frames, width, height, S, rgb, frame_stride, first_row, chunk, swizzle, stream);
}

#if !defined(RT_FAST)
// ------------------------------------------------------------------------------------------
// Batched frames for callers that cannot allocate (rt_render_frames_noalloc): the per-frame constants
// that rt_render_frames stages in a device array of the context go through this file-scope array
// instead.  Every GPU holds its own copy of the array at its own address until the process ends; its
// address on a device is looked up once, with that device current (batch_scratch_init, called by
// rt_ctx_create), and from then on host code uses that address only -- the array's name is not a
// device address on the host side.  All launches on one GPU share the array: calls that overlap on
// one GPU (different streams, or different contexts on the same device) race.
// ------------------------------------------------------------------------------------------
constexpr size_t kBatchScratchBytes = (size_t)RT_MAX_BATCH * sizeof(FrameArgs);
static __device__ __align__(256) unsigned char scratch[kBatchScratchBytes];

constexpr int kMaxScratchDevices = 64;
static void* g_scratch_ptr[kMaxScratchDevices] = {};
static std::mutex g_scratch_lock;

hipError_t batch_scratch_init(int device) {
    if (device < 0 || device >= kMaxScratchDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_scratch_lock);
    if (g_scratch_ptr[device]) return hipSuccess;
    int current = -1;
    hipError_t e = hipGetDevice(&current);
    if (e != hipSuccess) return e;
    if (current != device) return hipErrorInvalidDevice;       // the address differs from GPU to GPU
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

size_t batch_scratch_bytes() { return kBatchScratchBytes; }

hipError_t launch_batch_noalloc(int device, const FrameArgs* h_frames, int n_frames, int width, int height, const SceneArgs& S, uint8_t* rgb,
                                size_t frame_stride, int first_row, int chunk, int swizzle, int block, hipStream_t stream) {
    if (device < 0 || device >= kMaxScratchDevices || !h_frames) return hipErrorInvalidValue;
    if (n_frames < 1 || (size_t)n_frames * sizeof(FrameArgs) > kBatchScratchBytes) return hipErrorInvalidValue;
    void* ptr;
    {
        std::lock_guard<std::mutex> guard(g_scratch_lock);
        ptr = g_scratch_ptr[device];
    }
    if (!ptr) return hipErrorNotInitialized;
    // every byte the kernel reads is written here first: nothing of an earlier call is left to clear
    hipError_t e = hipMemcpyAsync(ptr, h_frames, (size_t)n_frames * sizeof(FrameArgs), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    return launch_batch(static_cast<const FrameArgs*>(ptr), n_frames, width, height, S, rgb, frame_stride, first_row, chunk, swizzle, block, stream);
}

#endif  // !RT_FAST

// ------------------------------------------------------------------------------------------
// Variant 3: tiles, with cooperative groups where rays are long.
//
// A ray through a big mesh visits hundreds of nodes (the reference's traversal has no distance
// culling to speak of on such meshes), one after the other when a single lane walks it: the few
// tiles that cover such a mesh then run longer than the rest of the frame takes.  In those tiles
// a ray is given to kGroupLanes = 8 adjacent lanes.  They keep ONE stack of pending nodes in LDS:
// each round every lane pops a different node, tests it, and the survivors are pushed back with
// offsets from a ballot, so the walk is 8 nodes wide and nothing is visited twice.  PairNodes and
// leaves wait on separate stacks so that a round is all box tests or all triangle tests.  The set
// of visited nodes is the one a single lane would visit (up to culling, which never changes the
// result), and the winner is chosen by (t, priority), so the outcome does not depend on the order.
// ------------------------------------------------------------------------------------------
constexpr int kCoopPairCap = 7 * 32;     // entries of the group's 256-entry LDS area used for PairNodes
static_assert(kCoopPairCap + 32 == kStackDepth * kGroupLanes, "the other 32 entries hold pending leaves (at most 23)");
constexpr int kCoopNarrowAt = kCoopPairCap - 2 * kGroupLanes - kStackDepth;   // see pair round

template <int BLOCK>
struct CoopGroup {
    uint32_t lane;          // lane in the wave; the group is lanes [lane & ~7, lane | 7]
    uint32_t* area;         // LDS: this group's 8 columns of the workgroup's stack area

    // The group's area is rows of 8 adjacent dwords, BLOCK dwords apart: entry i -> row i/8, column i%8.
    __device__ __forceinline__ uint32_t& pair_slot(int i) const { return area[(i >> 3) * BLOCK + (i & 7)]; }
    __device__ __forceinline__ uint32_t& leaf_slot(int i) const { return area[((kCoopPairCap + i) >> 3) * BLOCK + (i & 7)]; }

    // Bits of the group's 8 lanes in a wave-wide ballot.
    __device__ __forceinline__ uint32_t group_bits(unsigned long long votes) const { return (uint32_t)(votes >> (lane & ~7u)) & 0xFFu; }

    template <bool ANY, class Nodes>
    __device__ __forceinline__ bool run(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, float bound, Hit& out) const {
        constexpr bool CULL = !ANY;
        const uint32_t j = lane & 7u;
        const uint32_t below = (1u << j) - 1u;
        Work unused; unused.slabs = 0; unused.tris = 0;
        Query q;
        // All lanes of the group hold the same ray, so this is uniform over the group.
        const bool start = begin_query<ANY, false, false>(M, o, inv, bound, q, unused);
        int n_pair = 0, n_leaf = 0;
        if (start) {
            const uint32_t root = entry_ref<false>(M, d);
            if (root & kLeafBit) { n_leaf = 1; if (j == 0) leaf_slot(0) = root; }
            else { n_pair = 1; if (j == 0) pair_slot(0) = root; }
        }
        bool done = false;      // any-hit: the group has its answer
        for (;;) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // stack writes of the previous round
            __builtin_amdgcn_wave_barrier();
            const bool leaf_round = __any(n_leaf >= kGroupLanes) || !__any(n_pair > 0);
            if (leaf_round) {
                if (!__any(n_leaf > 0)) break;                          // nothing pending in any group of the wave
                const int take = n_leaf < kGroupLanes ? n_leaf : kGroupLanes;
                bool stop = false;
                if ((int)j < take) stop = test_leaf<ANY, false, false, kLeafBatch>(M, nodes, leaf_slot(n_leaf - 1 - (int)j), o, d, q, unused);
                n_leaf -= take;
                if (ANY) {
                    if (group_bits(__ballot(stop)) != 0) { done = true; n_leaf = 0; n_pair = 0; }
                } else {
                    // share the closest distance found so far inside the group: it only tightens culling
                    float lim = q.limit;
                    lim = fminf(lim, __shfl_xor(lim, 1, 64));
                    lim = fminf(lim, __shfl_xor(lim, 2, 64));
                    lim = fminf(lim, __shfl_xor(lim, 4, 64));
                    q.limit = lim;
                }
            } else {
                // Near the capacity of the pair stack only one lane pops: a round then adds at most
                // one entry and goes one level down, so at most kStackDepth more can accumulate.
                const bool narrow = __any(n_pair > kCoopNarrowAt);
                const int width = narrow ? 1 : kGroupLanes;
                const int take = n_pair < width ? n_pair : width;
                bool pl = false, pr = false, ll = false, lr = false;
                uint32_t lref = 0, rref = 0;
                if ((int)j < take) {
                    const PairNode N = nodes(M, pair_slot(n_pair - 1 - (int)j));
                    float tl, tr;
                    bool hl = slab_t(o, inv, N.lmin, N.lmax, tl);
                    bool hr = slab_t(o, inv, N.rmin, N.rmax, tr);
                    if (CULL) { hl = hl && !(tl > q.limit); hr = hr && !(tr > q.limit); }
                    lref = N.lref; rref = N.rref;
                    pl = hl && !(lref & kLeafBit); ll = hl && (lref & kLeafBit);
                    pr = hr && !(rref & kLeafBit); lr = hr && (rref & kLeafBit);
                }
                n_pair -= take;
                const uint32_t bpl = group_bits(__ballot(pl)), bpr = group_bits(__ballot(pr));
                const uint32_t bll = group_bits(__ballot(ll)), blr = group_bits(__ballot(lr));
                if (pl) pair_slot(n_pair + __popc(bpl & below)) = lref;
                if (pr) pair_slot(n_pair + __popc(bpl) + __popc(bpr & below)) = rref;
                n_pair += __popc(bpl) + __popc(bpr);
                if (ll) leaf_slot(n_leaf + __popc(bll & below)) = lref;
                if (lr) leaf_slot(n_leaf + __popc(bll) + __popc(blr & below)) = rref;
                n_leaf += __popc(bll) + __popc(blr);
            }
        }
        if (ANY) return done;
        // butterfly over the 8 lanes: afterwards every lane holds the group's winner
        bool found = q.found;
        Hit h = q.hit;
        for (int off = 1; off < kGroupLanes; off <<= 1) {
            Hit other;
            other.t = __shfl_xor(h.t, off, 64);
            other.u = __shfl_xor(h.u, off, 64);
            other.v = __shfl_xor(h.v, off, 64);
            other.tri = (uint32_t)__shfl_xor((int)h.tri, off, 64);
            other.prio = (uint32_t)__shfl_xor((int)h.prio, off, 64);
            const bool other_found = __shfl_xor((int)found, off, 64) != 0;
            if (beats(other_found, other, found, h)) { h = other; found = true; }
        }
        out = h;
        return found;
    }

    template <bool COUNT, bool FULL, class Stack, class Nodes>
    __device__ __forceinline__ bool closest(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, float bound,
                                            Hit& h, Stack&, Work&) const {
        return run<false>(M, nodes, o, d, inv, bound, h);
    }
    template <bool COUNT, bool FULL, class Stack, class Nodes>
    __device__ __forceinline__ bool any(const MeshArgs& M, const Nodes& nodes, F3 o, F3 d, F3 inv, Stack&, Work&) const {
        Hit h = no_hit();
        return run<true>(M, nodes, o, d, inv, 1e30f, h);
    }
    __device__ __forceinline__ bool exhausted() const { return false; }
    static constexpr bool kVerify = false;
};

__device__ __forceinline__ bool in_rect(const int32_t* r, int tx, int ty) { return tx >= r[0] && tx < r[2] && ty >= r[1] && ty < r[3]; }

template <int BLOCK, bool COUNT, bool FULL>
__global__ void __launch_bounds__(BLOCK) render_tiles(const FrameArgs A, const SceneArgs S, const OutputPlanes O) {
    __shared__ uint32_t lds_stack[kStackDepth * BLOCK];
    GlobalNodes nodes; nodes.all = S.pairs;

    const int lane = threadIdx.x & 63;
    const int tiles_x = (A.width + 7) >> 3, tiles_y = (A.height + 7) >> 3;
    int wave = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const unsigned long long t_start = O.tile_clock ? wall_clock64() : 0ull;
    LaneTotals tot = {0, 0, 0, 0, 0, 0};
    int tile = -1;

    // Wave index space: [rect 0: 8 waves per tile][rect 1] ... [all tiles, one wave each].
    // A tile that lies in several rectangles is rendered by the first; the others' waves, and
    // the one-wave-per-tile wave of any cooperative tile, have nothing to do.
    bool coop = false;
    int tx = 0, ty = 0, sub = 0;
    for (int r = 0; r < A.n_rects; r++) {
        const int w = A.rect[r][2] - A.rect[r][0], h = A.rect[r][3] - A.rect[r][1];
        const int n = w * h * kGroupLanes;
        if (wave < n) {
            const int t = wave >> 3;
            sub = wave & 7;
            ty = A.rect[r][1] + t / w; tx = A.rect[r][0] + t % w;
            coop = true;
            for (int e = 0; e < r; e++) if (in_rect(A.rect[e], tx, ty)) coop = false;
            if (!coop) return;
            break;
        }
        wave -= n;
    }
    if (coop) {
        const int p = lane >> 3;                       // pixel of this wave: a 4 wide, 2 high block of the tile
        const int px = tx * 8 + (sub & 1) * 4 + (p & 3);
        const int py = ty * 8 + (sub >> 1) * 2 + (p >> 2);
        if (sub == 0) tile = ty * tiles_x + tx;
        if (px < A.width && py < A.height) {
            CoopGroup<BLOCK> party;
            party.lane = (uint32_t)lane;
            party.area = &lds_stack[threadIdx.x & ~7u];
            LdsStack<BLOCK> unused_stack; unused_stack.init(nullptr);
            Work work; work.slabs = 0; work.tris = 0;
            const PixelOut r = shade_pixel<false, false, false>(A, nodes, party, px, py, unused_stack, work, O.shadow == nullptr);
            if ((lane & 7) == 0) store_pixel(O, A.width, px, py, r);
        }
    } else {
        if (wave >= tiles_x * tiles_y) return;
        ty = wave / tiles_x; tx = wave - ty * tiles_x;
        for (int e = 0; e < A.n_rects; e++) if (in_rect(A.rect[e], tx, ty)) return;
        tile = wave;
        const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
        if (px < A.width && py < A.height) {
            LdsStack<BLOCK> stack;
            stack.init(&lds_stack[threadIdx.x]);
            Work work; work.slabs = 0; work.tris = 0;
            const PixelOut r = shade_pixel<COUNT, FULL, false>(A, nodes, Solo(), px, py, stack, work, O.shadow == nullptr);
            store_pixel(O, A.width, px, py, r);
            tot.pixels = 1; tot.hits = r.hit; tot.darks = r.dark; tot.slabs = work.slabs; tot.tris = work.tris; tot.idle = work.idle;
        }
    }
    if (O.tile_clock && lane == 0 && tile >= 0) {
        O.tile_clock[2 * (size_t)tile] = t_start;
        O.tile_clock[2 * (size_t)tile + 1] = wall_clock64();
    }
    if (COUNT) flush_counters(O.counters, tot);
}

template <int BLOCK>
static hipError_t launch_tiles(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, bool count, bool full, hipStream_t stream) {
    long long waves = (long long)((A.width + 7) / 8) * ((A.height + 7) / 8);
    if ((count || full) && A.n_rects != 0) return hipErrorInvalidValue;       // cooperative tiles do not count work
    for (int r = 0; r < A.n_rects; r++) {
        const long long w = A.rect[r][2] - A.rect[r][0], h = A.rect[r][3] - A.rect[r][1];
        if (w <= 0 || h <= 0) return hipErrorInvalidValue;
        waves += w * h * kGroupLanes;
    }
    const int waves_per_block = BLOCK / 64;
    const int blocks = (int)((waves + waves_per_block - 1) / waves_per_block);
    if (count && full)  hipLaunchKernelGGL((render_tiles<BLOCK, true, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else if (count)     hipLaunchKernelGGL((render_tiles<BLOCK, true, false>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else if (full)      hipLaunchKernelGGL((render_tiles<BLOCK, false, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    else                hipLaunchKernelGGL((render_tiles<BLOCK, false, false>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O);
    return hipGetLastError();
}

#if defined(RT_EXPERIMENTAL_VARIANTS)
// ------------------------------------------------------------------------------------------
// Variant 4: two passes.  Pass 1 renders every tile with one lane per pixel, except that a pixel
// whose rays together need more than FrameArgs::defer_budget node visits is abandoned and appended to a list
// (ray compaction: ballot + popcount offsets inside the wave, one LDS add per wave, one global
// add per workgroup).  Pass 2 walks the list with cooperative groups, eight pixels per wave.
// Only the pixels that need the wide traversal pay for it, and none of them holds up a tile.
// ------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) render_pass1(const FrameArgs A, const SceneArgs S, const OutputPlanes O,
                                                      DeferQueue* dq, uint32_t* items, const uint32_t capacity, const int parity) {
    __shared__ uint32_t lds_stack[kStackDepth * BLOCK];
    __shared__ uint32_t block_count, block_base;
    if (threadIdx.x == 0) block_count = 0;
    __syncthreads();

    LdsStack<BLOCK> stack;
    stack.init(&lds_stack[threadIdx.x]);
    GlobalNodes nodes; nodes.all = S.pairs;

    const int lane = threadIdx.x & 63;
    const int tiles_x = (A.width + 7) >> 3, tiles_y = (A.height + 7) >> 3;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const unsigned long long t_start = O.tile_clock ? wall_clock64() : 0ull;
    bool defer = false;
    int px = 0, py = 0;
    if (tile < tiles_x * tiles_y) {
        tile_pixel(tile, lane, tiles_x, px, py);
        if (px < A.width && py < A.height) {
            Work work; work.slabs = 0; work.tris = 0;
            SoloBudget party; party.steps = A.defer_budget;
            const PixelOut r = shade_pixel<false, false, true>(A, nodes, party, px, py, stack, work);
            defer = r.deferred != 0;
            if (!defer) store_pixel(O, A.width, px, py, r);
        }
        if (O.tile_clock && lane == 0) {
            O.tile_clock[2 * (size_t)tile] = t_start;
            O.tile_clock[2 * (size_t)tile + 1] = wall_clock64();
        }
    }
    // compaction of the deferred pixels of this workgroup into the list
    const unsigned long long votes = __ballot(defer);
    const uint32_t n_wave = (uint32_t)__popcll(votes);
    uint32_t wave_off = 0;
    if (lane == 0 && n_wave) wave_off = atomicAdd(&block_count, n_wave);
    wave_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_off);
    __syncthreads();
    if (threadIdx.x == 0 && block_count) block_base = atomicAdd(&dq->count[parity], block_count);
    __syncthreads();
    if (defer) {
        const uint32_t at = block_base + wave_off + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
        if (at < capacity) items[at] = (uint32_t)py * (uint32_t)A.width + (uint32_t)px;
    }
}

constexpr int kPass2Block = 256;

__global__ void __launch_bounds__(kPass2Block) render_pass2(const FrameArgs A, const SceneArgs S, const OutputPlanes O,
                                                            DeferQueue* dq, const uint32_t* items, const uint32_t capacity, const int parity) {
    __shared__ uint32_t lds_stack[kStackDepth * kPass2Block];
    GlobalNodes nodes; nodes.all = S.pairs;
    const int lane = threadIdx.x & 63;
    uint32_t n = dq->count[parity];
    if (n > capacity) n = capacity;
    const uint32_t n_batches = (n + kGroupLanes - 1) / kGroupLanes;
    const uint32_t wave = blockIdx.x * (kPass2Block / 64) + (threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * (kPass2Block / 64);
    if (wave == 0 && lane == 0) dq->count[parity ^ 1] = 0;           // the next frame's counter
    CoopGroup<kPass2Block> party;
    party.lane = (uint32_t)lane;
    party.area = &lds_stack[threadIdx.x & ~7u];
    LdsStack<kPass2Block> unused_stack; unused_stack.init(nullptr);
    for (uint32_t b = wave; b < n_batches; b += total_waves) {     // b is uniform over the wave
        const uint32_t at = b * kGroupLanes + (uint32_t)(lane >> 3);
        if (at < n) {
            const uint32_t id = items[at];
            const int py = (int)(id / (uint32_t)A.width), px = (int)(id - (uint32_t)py * (uint32_t)A.width);
            Work work; work.slabs = 0; work.tris = 0;
            const PixelOut r = shade_pixel<false, false, false>(A, nodes, party, px, py, unused_stack, work, O.shadow == nullptr);
            if ((lane & 7) == 0) store_pixel(O, A.width, px, py, r);
        }
    }
}

template <int BLOCK>
static hipError_t launch_two_pass(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, const LaunchConfig& cfg, hipStream_t stream) {
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    const int waves_per_block = BLOCK / 64;
    const int blocks = (tiles + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL((render_pass1<BLOCK>), dim3(blocks), dim3(BLOCK), 0, stream, A, S, O, cfg.defer, cfg.defer_items,
                       cfg.defer_capacity, cfg.parity);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // LDS (32 KB per workgroup) admits five workgroups of pass 2 per CU.
    const int grid2 = (cfg.n_cus > 0 ? cfg.n_cus : 256) * 5;
    hipLaunchKernelGGL(render_pass2, dim3(grid2), dim3(kPass2Block), 0, stream, A, S, O, cfg.defer, cfg.defer_items,
                       cfg.defer_capacity, cfg.parity);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Variants 1 and 2: persistent workgroups with the scene's nodes resident in LDS.
// ------------------------------------------------------------------------------------------
constexpr int kResidentBlock = 1024;     // 16 waves: one workgroup per CU
constexpr int kTileChunk = 4;            // tiles handed out per atomic

// SCHED 0: tiles handed out dynamically, kTileChunk at a time, from one atomic counter; the next
//          chunk is requested before the current one is processed.
// SCHED 1: static assignment, wave w of W takes tiles w, w+W, w+2W, ...
template <bool COUNT, bool FULL, int SCHED>
__global__ void __launch_bounds__(kResidentBlock) render_resident(const FrameArgs A, const SceneArgs S, const OutputPlanes O,
                                                                  TileQueue* queue, const int n_tiles) {
    extern __shared__ uint4 lds_raw[];
    // [ PairNode x n_pairs ][ stack: entries x 1024 dwords ]
    PairNode* lds_nodes = reinterpret_cast<PairNode*>(lds_raw);
    uint32_t* lds_stack = reinterpret_cast<uint32_t*>(lds_raw + (size_t)S.n_pairs * (sizeof(PairNode) / sizeof(uint4)));

    const uint4* src = reinterpret_cast<const uint4*>(S.pairs);
    const int n16 = (int)S.n_pairs * (int)(sizeof(PairNode) / sizeof(uint4));
    for (int i = threadIdx.x; i < n16; i += kResidentBlock) lds_raw[i] = src[i];
    __syncthreads();

    LdsStack<kResidentBlock> stack;
    stack.init(&lds_stack[threadIdx.x]);
    LdsNodes nodes; nodes.all = lds_nodes;

    const int lane = threadIdx.x & 63;
    const int tiles_x = (A.width + 7) >> 3;
    LaneTotals tot = {0, 0, 0, 0, 0, 0};

    auto do_tile = [&](int tile) {
        int px, py;
        tile_pixel(tile, lane, tiles_x, px, py);
        const unsigned long long t_start = O.tile_clock ? wall_clock64() : 0ull;
        if (px < A.width && py < A.height) {
            Work work; work.slabs = 0; work.tris = 0;
            const PixelOut r = shade_pixel<COUNT, FULL, false>(A, nodes, Solo(), px, py, stack, work, O.shadow == nullptr);
            store_pixel(O, A.width, px, py, r);
            if (COUNT) { tot.pixels += 1; tot.hits += r.hit; tot.darks += r.dark; tot.slabs += work.slabs; tot.tris += work.tris; tot.idle += work.idle; }
        }
        if (O.tile_clock && lane == 0) {
            O.tile_clock[2 * (size_t)tile] = t_start;
            O.tile_clock[2 * (size_t)tile + 1] = wall_clock64();
        }
    };

    if (SCHED == 1) {
        const int total_waves = (int)gridDim.x * (kResidentBlock / 64);
        for (int tile = (int)blockIdx.x * (kResidentBlock / 64) + (int)(threadIdx.x >> 6); tile < n_tiles; tile += total_waves) do_tile(tile);
    } else {
        unsigned int ticket = 0;
        if (lane == 0) ticket = atomicAdd(&queue->next, (unsigned int)kTileChunk);
        for (;;) {
            const int base = (int)__builtin_amdgcn_readfirstlane(ticket);
            if (base >= n_tiles) break;             // every wave reaches this: the counter only grows
            if (lane == 0) ticket = atomicAdd(&queue->next, (unsigned int)kTileChunk);
            for (int k = 0; k < kTileChunk; k++)
                if (base + k < n_tiles) do_tile(base + k);
        }
        // The last wave of the grid to get here resets the queue for the next launch.  All waves
        // have stopped taking tiles by then, and the stores are visible at the kernel boundary.
        if (lane == 0) {
            const unsigned int total_waves = gridDim.x * (kResidentBlock / 64);
            const unsigned int d = atomicAdd(&queue->done, 1u);
            if (d == total_waves - 1) {
                atomicExch(&queue->next, 0u);
                atomicExch(&queue->done, 0u);
            }
        }
    }
    if (COUNT) flush_counters(O.counters, tot);
}

size_t resident_lds_bytes(const SceneArgs& S) {
    return (size_t)S.n_pairs * sizeof(PairNode) + (size_t)S.stack_entries * kResidentBlock * sizeof(uint32_t);
}

template <int SCHED>
static hipError_t launch_resident(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, const LaunchConfig& cfg,
                                  bool count, bool full, hipStream_t stream) {
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    const size_t lds = resident_lds_bytes(S);
    int grid = cfg.n_cus > 0 ? cfg.n_cus : 256;
    const int waves = kResidentBlock / 64;
    if ((long long)grid * waves > tiles) grid = (tiles + waves - 1) / waves;
    void (*kern)(const FrameArgs, const SceneArgs, const OutputPlanes, TileQueue*, const int) =
        count ? (full ? render_resident<true, true, SCHED> : render_resident<true, false, SCHED>)
              : (full ? render_resident<false, true, SCHED> : render_resident<false, false, SCHED>);
    // Dynamic LDS beyond 64 KB has to be enabled per kernel function and device; done once each.
    static bool enabled[16][4] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int which = (count ? 2 : 0) + (full ? 1 : 0);
    if (dev < 0 || dev >= 16 || !enabled[dev][which]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 16) enabled[dev][which] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kResidentBlock), lds, stream, A, S, O, cfg.queue, tiles);
    return hipGetLastError();
}

bool resident_fits(const SceneArgs& S) { return resident_lds_bytes(S) <= kMaxLdsBytes; }

#else
bool resident_fits(const SceneArgs&) { return false; }
#endif  // RT_EXPERIMENTAL_VARIANTS

hipError_t launch_render(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, const LaunchConfig& cfg, hipStream_t stream) {
    if (A.width <= 0 || A.height <= 0) return hipErrorInvalidValue;
    const bool verify = (A.flags & RT_FLAG_VERIFY_SKIPS) != 0;
    const bool count = (A.flags & RT_FLAG_COUNT_WORK) != 0 && !verify;
    const bool full = (A.flags & RT_FLAG_REFERENCE_ORDER) != 0 && !verify;
    if ((count || verify) && !O.counters) return hipErrorInvalidValue;
    if (verify) return A.n_rects != 0 ? hipErrorInvalidValue : launch_direct<64>(A, S, O, false, false, true, stream);   // one lane per ray
    int variant = cfg.variant;
#if !defined(RT_EXPERIMENTAL_VARIANTS)
    if (variant != 3) variant = 0;                             // the experiments are not in this build
#else
    if ((variant == 1 || variant == 2) && (!cfg.queue || !resident_fits(S))) variant = 0;     // scene too large for LDS
    if (variant != 3 && A.n_rects != 0) return hipErrorInvalidValue;
    // Work counts and the reference's visiting order are defined for one lane per ray.
    if (variant == 4 && (count || full)) variant = 0;
    // The lean variants exist for the plain frame only, and their short stack for shallow trees only.
    if (variant >= 5 && (count || full || S.stack_entries > 16)) variant = 0;
    if (variant >= 5) {
        const int b = cfg.block;
        switch (variant) {
            case 5: return b == 64 ? launch_lean<64, 16, 2, 1>(A, S, O, stream) : launch_lean<256, 16, 2, 1>(A, S, O, stream);
            case 6: return b == 64 ? launch_lean<64, 16, 1, 1>(A, S, O, stream) : launch_lean<256, 16, 1, 1>(A, S, O, stream);
            case 7: return b == 64 ? launch_lean<64, 16, 2, 2>(A, S, O, stream) : launch_lean<256, 16, 2, 8>(A, S, O, stream);
            case 8: return b == 64 ? launch_lean<64, 16, 4, 1>(A, S, O, stream) : launch_lean<256, 16, 4, 1>(A, S, O, stream);
            default: return b == 64 ? launch_lean<64, 16, 1, 2>(A, S, O, stream) : launch_lean<256, 16, 1, 8>(A, S, O, stream);
        }
    }
    if (variant == 4 && (!cfg.defer || !cfg.defer_items || (long long)cfg.defer_capacity < (long long)A.width * A.height)) return hipErrorInvalidValue;
#endif
    switch (variant) {
#if defined(RT_EXPERIMENTAL_VARIANTS)
        case 4:
            if (cfg.block == 64) return launch_two_pass<64>(A, S, O, cfg, stream);
            if (cfg.block == 128) return launch_two_pass<128>(A, S, O, cfg, stream);
            return launch_two_pass<256>(A, S, O, cfg, stream);
        case 1:
            return launch_resident<0>(A, S, O, cfg, count, full, stream);
        case 2:
            return launch_resident<1>(A, S, O, cfg, count, full, stream);
#endif
        case 3:
            if (cfg.block == 64) return launch_tiles<64>(A, S, O, count, full, stream);
            if (cfg.block == 128) return launch_tiles<128>(A, S, O, count, full, stream);
            return launch_tiles<256>(A, S, O, count, full, stream);
        case 0:
        default:
            if (cfg.block == 64) return launch_direct<64>(A, S, O, count, full, false, stream);
            if (cfg.block == 128) return launch_direct<128>(A, S, O, count, full, false, stream);
            return launch_direct<256>(A, S, O, count, full, false, stream);
    }
}

}  // namespace rt

#if defined(RT_FAST)
// The fast build of this file (rt_kernels_fast.hip: fused multiply-adds, approximate division and square root, the
// slab test above; `rt` is renamed for the whole translation unit) is reached through these three entry points.  The
// argument structures are the ones of the exact build, layout for layout.
extern "C" hipError_t rt_fast_launch_render(const void* frame_args, const void* scene_args, const void* planes, const void* launch_config,
                                            hipStream_t stream) {
    return rt::launch_render(*static_cast<const rt::FrameArgs*>(frame_args), *static_cast<const rt::SceneArgs*>(scene_args),
                             *static_cast<const rt::OutputPlanes*>(planes), *static_cast<const rt::LaunchConfig*>(launch_config), stream);
}
extern "C" hipError_t rt_fast_launch_instanced(const void* frame_args, const void* instanced_args, const void* scene_args, const void* planes, int32_t* hit_tri,
                                               int tlas_stack_entries, int block, int wave_tiles, hipStream_t stream) {
    return rt::launch_instanced(*static_cast<const rt::FrameArgs*>(frame_args), *static_cast<const rt::InstancedArgs*>(instanced_args),
                                *static_cast<const rt::SceneArgs*>(scene_args), *static_cast<const rt::OutputPlanes*>(planes), hit_tri, tlas_stack_entries,
                                block, wave_tiles, stream);
}
extern "C" hipError_t rt_fast_launch_batch(const void* d_frames, int n_frames, int width, int height, const void* scene_args, uint8_t* rgb,
                                           size_t frame_stride, int first_row, int chunk, int swizzle, int block, hipStream_t stream) {
    return rt::launch_batch(static_cast<const rt::FrameArgs*>(d_frames), n_frames, width, height, *static_cast<const rt::SceneArgs*>(scene_args), rgb,
                            frame_stride, first_row, chunk, swizzle, block, stream);
}
#endif
