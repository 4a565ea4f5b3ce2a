// rt_build.h -- device build of the reference's BVH (accel/bvh.c:4-59); see rt_build.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <string>

#include <vector>

#include "../../include/rt_mi355.h"
#include "rt_layout.h"

namespace rt {

constexpr int kMaxBuildLevels = 256;      // depth at which the device build gives up (the traversal stack allows far less)

struct BuildStats {
    float device_ms = 0.0f;    // first kernel to last kernel, without the copies from and to the host
    int levels = 0;            // depth of the tree, counted in levels of nodes
    int launches = 0;
    int nodes = 0;
};

// What the packing step leaves: triangle and attribute records on the device (owned by the caller, hipFree),
// the PairNodes on the host, and the root's box / reference / chain depth as rt_pack.h's PackedMesh has them.
struct PackedOnDevice {
    TriRec* d_tris = nullptr;
    AttrRec* d_attrs = nullptr;
    std::vector<PairNode> pairs;
    float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
    uint32_t root_ref = kLeafBit;
    int depth = 0;
};

// A BVH built on the device and still there: download() gives the host form, pack() the records the
// traversal kernels read without a round trip through the host.
class DeviceBvh {
public:
    DeviceBvh() = default;
    DeviceBvh(const DeviceBvh&) = delete;
    DeviceBvh& operator=(const DeviceBvh&) = delete;
    ~DeviceBvh();
    int build(const rt_triangle* h_tris, int32_t n, hipStream_t stream, std::string& err);        // n > 0
    int download(rt_triangle* h_tris, rt_flat_node* h_nodes, int32_t cap, int32_t* n_nodes, hipStream_t stream, std::string& err);
    int triangles_to_host(rt_triangle* h_tris, hipStream_t stream, std::string& err);
    int pack(PackedOnDevice& out, hipStream_t stream, std::string& err);
    BuildStats stats;
private:
    struct Impl;
    Impl* impl = nullptr;
};

// Same contract as the host build: reorders h_tris in place exactly as create_bvh does and writes the
// flat node array in creation order.  n > 0.  Returns an rt_status; err gets the reason.
int build_bvh_on_device(rt_triangle* h_tris, int32_t n, rt_flat_node* h_nodes, int32_t cap, int32_t* n_nodes, hipStream_t stream,
                        std::string& err, BuildStats* stats);

}  // namespace rt
