// rt_kernels.h -- launch interface between the host library and the HIP kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include "../../include/rt_mi355.h"
#include "rt_layout.h"

namespace rt {

constexpr int kNumVariants = 10;     // kernel variant ids accepted by LaunchConfig::variant (1, 2, 4..9: experiments, see rt_kernels.hip)
#if defined(RT_EXPERIMENTAL_VARIANTS)
constexpr int kExperimentalVariants = 1;
#else
constexpr int kExperimentalVariants = 0;
#endif
constexpr int kDefaultVariant = 3;
constexpr size_t kMaxLdsBytes = 160 * 1024;   // LDS of one gfx950 CU

struct LaunchConfig {
    int variant = kDefaultVariant;   // kernel variant id (see rt_kernels.hip)
    int block = 64;                  // threads per workgroup: 64, 128 or 256
    int n_cus = 256;                 // compute units of the device (grid of the persistent variant)
    TileQueue* queue = nullptr;      // zero-initialised device words owned by the context
    // two-pass variant: list of deferred pixels (y*width+x), its counters, and which counter this frame uses
    DeferQueue* defer = nullptr;
    uint32_t* defer_items = nullptr;
    uint32_t defer_capacity = 0;
    int parity = 0;
};

// True when the scene's nodes and the traversal stacks of one workgroup fit in LDS.
bool resident_fits(const SceneArgs& S);

// Enqueues one frame on `stream`.  Shapes are validated by the caller (rt_host.cpp):
// S.pairs holds S.n_pairs records and every MeshArgs range lies inside it, tris/attrs/texels are
// live device allocations sized for their mesh, A.n_mesh <= kMaxMeshes, O.rgb holds
// width*height*3 bytes and the optional planes width*height elements.  Launches that share a
// TileQueue must be ordered one after another (same stream, or synchronised by the caller).
hipError_t launch_render(const FrameArgs& A, const SceneArgs& S, const OutputPlanes& O, const LaunchConfig& cfg, hipStream_t stream);

// Enqueues n_frames frames of width x height in one launch.  d_frames is a device array of n_frames
// FrameArgs (same size, flags 0), rgb receives frame f at rgb + f*frame_stride; tile rows are
// scanned starting at first_row; `chunk` consecutive tiles of a frame are followed by the same tiles
// of the next frame.  block: 64 or 256.
hipError_t launch_batch(const FrameArgs* d_frames, int n_frames, int width, int height, const SceneArgs& S, uint8_t* rgb, size_t frame_stride,
                        int first_row, int chunk, int swizzle, int block, hipStream_t stream);

// The same launch with the per-frame constants staged in a file-scope device array of the library instead of a
// buffer of the caller's context: h_frames (page-locked host memory, n_frames <= RT_MAX_BATCH records) is copied there
// on `stream`, then the kernel is enqueued.  batch_scratch_init(device) must have succeeded once with `device`
// current; neither call allocates.  All launches on one GPU share the array (order them on one stream).
hipError_t batch_scratch_init(int device);
size_t batch_scratch_bytes();
hipError_t launch_batch_noalloc(int device, const FrameArgs* h_frames, int n_frames, int width, int height, const SceneArgs& S, uint8_t* rgb,
                                size_t frame_stride, int first_row, int chunk, int swizzle, int block, hipStream_t stream);

// Enqueues one sampled frame (X.spp camera rays per pixel, X.n_lights lights).  O.rgb is required,
// O.hit_id optional (sample 0), the other planes are not written.  block: 64 or 256.
hipError_t launch_sampled(const FrameArgs& A, const SampleArgs& X, const SceneArgs& S, const OutputPlanes& O, int block, int split, hipStream_t stream);

// Enqueues one frame of an instanced scene.  I.inst holds one MeshArgs per instance (pair ranges inside
// S.pairs, live tris/attrs/texels), I.tlas the hierarchy over the non-empty ones (leaves name their instance: write_direct in rt_pack.h), its depth at
// most kTlasStackDepth - 2.  O.hit_id receives the instance index, hit_tri (optional) the triangle.
// wave_tiles > 0 (and no work counting): the per-wavefront scheduler of rt_instanced_wave.h, a wave per strip of that many tiles.
hipError_t launch_instanced(const FrameArgs& A, const InstancedArgs& I, const SceneArgs& S, const OutputPlanes& O, int32_t* hit_tri,
                            int tlas_entries, int block, int wave_tiles, hipStream_t stream);

// Enqueues the bicubic upscale of n_frames RGB8 frames (width x height, frame f at rgb + f*frame_stride
// bytes) to ARGB32 frames (scaled_width x scaled_height, frame f at argb + f*out_stride pixels).
// scaled sizes must be at least 2 (the source coordinate divides by scaled-1).
hipError_t launch_upscale(const uint8_t* rgb, size_t frame_stride, int n_frames, int width, int height, int scaled_width,
                          int scaled_height, uint32_t* argb, size_t out_stride, hipStream_t stream);

}  // namespace rt
