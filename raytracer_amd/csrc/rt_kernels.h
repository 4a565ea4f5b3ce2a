// rt_kernels.h -- launch interface between the host library and the HIP kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include "../../include/rt_mi355.h"
#include "rt_layout.h"

namespace rt {

constexpr int kNumVariants = 1;      // kernel variants selectable through LaunchConfig::variant
constexpr int kDefaultVariant = 0;

struct LaunchConfig {
    int variant = kDefaultVariant;   // kernel variant id (see rt_kernels.hip)
    int block = 256;     // threads per workgroup
};

// Enqueues one frame on `stream`.  Shapes are validated by the caller (rt_host.cpp):
// every MeshArgs pointer is a live device allocation sized for its BVH, A.n_mesh <= kMaxMeshes,
// O.rgb holds width*height*3 bytes and the optional planes width*height elements.
hipError_t launch_render(const FrameArgs& A, const OutputPlanes& O, const LaunchConfig& cfg, hipStream_t stream);

}  // namespace rt
