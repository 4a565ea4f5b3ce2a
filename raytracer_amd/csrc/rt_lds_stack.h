// rt_lds_stack.h -- the per-lane traversal stack of the kernels (device code).
#pragma once
#include <stdint.h>

namespace rt {

// Per-lane traversal stack in LDS, laid out [entry][thread] so that the 64 lanes of a wave
// touch 64 consecutive dwords on every push/pop (no bank conflicts).
template <int BLOCK>
struct LdsStack {
    uint32_t* base;      // &stack_area[threadIdx.x]
    uint32_t* top;       // next free entry: base + entries * BLOCK
    __device__ __forceinline__ void init(uint32_t* b) { base = b; top = b; }
    __device__ __forceinline__ void clear() { top = base; }
    __device__ __forceinline__ bool empty() const { return top == base; }
    __device__ __forceinline__ void push(uint32_t v) { *top = v; top += BLOCK; }
    __device__ __forceinline__ void push_if(bool c, uint32_t v) { if (c) *top = v; top += c ? BLOCK : 0; }
    __device__ __forceinline__ uint32_t pop() { top -= BLOCK; return *top; }
};

}  // namespace rt
