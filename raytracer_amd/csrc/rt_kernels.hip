// rt_kernels.hip -- gfx950 kernels of the frame loop and their launchers.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (float results are compared
// bit-for-bit with a CPU build that has no FMA contraction).
#include <hip/hip_runtime.h>

#include "rt_kernels.h"
#include "rt_pixel.h"

namespace rt {

// Per-lane traversal stack in LDS, laid out [entry][thread] so that the 64 lanes of a wave
// touch 64 consecutive dwords on every push/pop (no bank conflicts).
template <int BLOCK>
struct LdsStack {
    uint32_t* base;      // &lds[threadIdx.x]
    int sp;
    __device__ __forceinline__ void clear() { sp = 0; }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
    __device__ __forceinline__ void push(uint32_t v) { base[sp * BLOCK] = v; sp++; }
    __device__ __forceinline__ uint32_t pop() { sp--; return base[sp * BLOCK]; }
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

__device__ __forceinline__ void flush_counters(DeviceCounters* c, bool active, const PixelOut& r, const Work& w) {
    const unsigned long long prim = wave_sum(active ? 1ull : 0ull);
    const unsigned long long hit = wave_sum(active ? (unsigned long long)r.hit : 0ull);
    const unsigned long long dark = wave_sum(active ? (unsigned long long)r.dark : 0ull);
    const unsigned long long slabs = wave_sum(active ? (unsigned long long)w.slabs : 0ull);
    const unsigned long long tris = wave_sum(active ? (unsigned long long)w.tris : 0ull);
    if ((threadIdx.x & 63) == 0) {
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
template <int BLOCK, bool COUNT, bool FULL>
__global__ void __launch_bounds__(BLOCK) render_direct(const FrameArgs A, const OutputPlanes O) {
    __shared__ uint32_t lds_stack[kStackDepth * BLOCK];
    LdsStack<BLOCK> stack;
    stack.base = &lds_stack[threadIdx.x];
    stack.sp = 0;

    const int tiles_x = (A.width + 7) >> 3;
    const int tile = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    int px, py;
    tile_pixel(tile, threadIdx.x & 63, tiles_x, px, py);
    const bool active = px < A.width && py < A.height;

    Work work; work.slabs = 0; work.tris = 0;
    PixelOut r; r.hit = 0; r.dark = 0;
    if (active) {
        r = shade_pixel<COUNT, FULL>(A, px, py, stack, work);
        store_pixel(O, A.width, px, py, r);
    }
    if (COUNT) flush_counters(O.counters, active, r, work);
}

template <int BLOCK>
static hipError_t launch_direct(const FrameArgs& A, const OutputPlanes& O, bool count, bool full, hipStream_t stream) {
    const int tiles = ((A.width + 7) / 8) * ((A.height + 7) / 8);
    const int waves_per_block = BLOCK / 64;
    const int blocks = (tiles + waves_per_block - 1) / waves_per_block;
    if (count && full)  hipLaunchKernelGGL((render_direct<BLOCK, true, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, O);
    else if (count)     hipLaunchKernelGGL((render_direct<BLOCK, true, false>), dim3(blocks), dim3(BLOCK), 0, stream, A, O);
    else if (full)      hipLaunchKernelGGL((render_direct<BLOCK, false, true>), dim3(blocks), dim3(BLOCK), 0, stream, A, O);
    else                hipLaunchKernelGGL((render_direct<BLOCK, false, false>), dim3(blocks), dim3(BLOCK), 0, stream, A, O);
    return hipGetLastError();
}

hipError_t launch_render(const FrameArgs& A, const OutputPlanes& O, const LaunchConfig& cfg, hipStream_t stream) {
    if (A.width <= 0 || A.height <= 0) return hipErrorInvalidValue;
    const bool count = (A.flags & RT_FLAG_COUNT_WORK) != 0;
    const bool full = (A.flags & RT_FLAG_REFERENCE_ORDER) != 0;
    if (count && !O.counters) return hipErrorInvalidValue;
    switch (cfg.variant) {
        case 0:
        default:
            if (cfg.block == 64) return launch_direct<64>(A, O, count, full, stream);
            if (cfg.block == 128) return launch_direct<128>(A, O, count, full, stream);
            return launch_direct<256>(A, O, count, full, stream);
    }
}

}  // namespace rt
