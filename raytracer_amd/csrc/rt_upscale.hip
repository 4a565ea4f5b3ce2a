// rt_upscale.hip -- gfx950 kernel of the encoder's bicubic upscale (reference scene.c:188-202,
// utils/image.c:4-78): RGB8 frames in, ARGB32 frames out.
//
// A wave owns 64 adjacent output columns and ROWS consecutive output rows; each lane walks down its
// column (rt_bicubic.h: upscale_strip), so the 64 lanes of a store cover 256 contiguous bytes and the
// source rows a wave touches are contiguous runs of about 64*3 bytes.  blockIdx.z is the frame.
#include <hip/hip_runtime.h>

#include "rt_bicubic.h"
#include "rt_kernels.h"

namespace rt {

template <int ROWS>
__global__ void __launch_bounds__(64) upscale_frames(const uint8_t* __restrict__ rgb, const unsigned long long frame_stride,
                                                     const int width, const int height, const int scaled_width,
                                                     const int scaled_height, uint32_t* __restrict__ argb,
                                                     const unsigned long long out_stride) {
    const int x = (int)(blockIdx.x * 64u + threadIdx.x);
    const int y0 = (int)blockIdx.y * ROWS;
    const uint8_t* frame = rgb + (size_t)blockIdx.z * frame_stride;
    uint32_t* out = argb + (size_t)blockIdx.z * out_stride;
    const bool live = x < scaled_width;
    // lanes past the right edge repeat the last column so that the wave stays uniform; they store nothing
    const int xc = live ? x : scaled_width - 1;
    upscale_strip(frame, width, height, scaled_width, scaled_height, xc, y0, ROWS,
                  [&](int y, uint32_t v) { if (live) out[(size_t)y * (size_t)scaled_width + (size_t)x] = v; });
}

hipError_t launch_upscale(const uint8_t* rgb, size_t frame_stride, int n_frames, int width, int height, int scaled_width,
                          int scaled_height, uint32_t* argb, size_t out_stride, hipStream_t stream) {
    if (!rgb || !argb || n_frames <= 0 || width <= 0 || height <= 0 || scaled_width < 2 || scaled_height < 2) return hipErrorInvalidValue;
    if (n_frames > 65535) return hipErrorInvalidValue;
    constexpr int ROWS = 8;
    const unsigned gx = (unsigned)((scaled_width + 63) / 64), gy = (unsigned)((scaled_height + ROWS - 1) / ROWS);
    if (gy > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL((upscale_frames<ROWS>), dim3(gx, gy, (unsigned)n_frames), dim3(64), 0, stream, rgb,
                       (unsigned long long)frame_stride, width, height, scaled_width, scaled_height, argb, (unsigned long long)out_stride);
    return hipGetLastError();
}

}  // namespace rt
