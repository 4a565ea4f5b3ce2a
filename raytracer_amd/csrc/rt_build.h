// rt_build.h -- device build of the reference's BVH (accel/bvh.c:4-59); see rt_build.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <string>

#include "../../include/rt_mi355.h"

namespace rt {

constexpr int kMaxBuildLevels = 256;      // depth at which the device build gives up (the traversal stack allows far less)

struct BuildStats {
    float device_ms = 0.0f;    // first kernel to last kernel, without the copies from and to the host
    int levels = 0;            // depth of the tree, counted in levels of nodes
    int launches = 0;
    int nodes = 0;
};

// Same contract as the host build: reorders h_tris in place exactly as create_bvh does and writes the
// flat node array in creation order.  n > 0.  Returns an rt_status; err gets the reason.
int build_bvh_on_device(rt_triangle* h_tris, int32_t n, rt_flat_node* h_nodes, int32_t cap, int32_t* n_nodes, hipStream_t stream,
                        std::string& err, BuildStats* stats);

}  // namespace rt
