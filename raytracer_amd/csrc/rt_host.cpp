// rt_host.cpp -- the C ABI declared in include/rt_mi355.h: context, scene upload, per-frame
// constant preparation and kernel launch.  There is no CPU rendering path in this library:
// without a GPU every rendering entry point fails with RT_ERR_NO_DEVICE / RT_ERR_DEVICE.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>
#include <utility>
#include <vector>

#include "rt_build.h"
#include "rt_kernels.h"
#include "rt_pack.h"

using namespace rt;

// the fast build of the kernels (rt_kernels_fast.hip)
extern "C" hipError_t rt_fast_launch_render(const void* frame_args, const void* scene_args, const void* planes, const void* launch_config, hipStream_t stream);
extern "C" hipError_t rt_fast_launch_instanced(const void* frame_args, const void* instanced_args, const void* scene_args, const void* planes, int32_t* hit_tri,
                                               int tlas_stack_entries, int block, int wave_tiles, hipStream_t stream);
extern "C" hipError_t rt_fast_launch_batch(const void* d_frames, int n_frames, int width, int height, const void* scene_args, uint8_t* rgb,
                                           size_t frame_stride, int first_row, int chunk, int swizzle, int block, hipStream_t stream);

namespace {

constexpr int kBatchSlots = 4;
constexpr int kInstSlots = 2;

struct DeviceMesh {
    std::vector<PairNode> pairs;   // host copy; the device copy lives in the scene's combined array
    std::vector<TriRec> tris;      // host copy, for the LDS-resident blob (empty for a mesh packed on the device)
    bool host_tris = true;
    int depth = 0;
    TriRec* d_tris = nullptr;
    AttrRec* d_attrs = nullptr;
    uint32_t* d_texels = nullptr;
    uint32_t* d_grid = nullptr;    // flat meshes: entry references per cell (Packer::build_grid); sheared tree: its header, families and records
    std::vector<uint32_t> shear_blob;   // host copy of the latter, for the per-frame part of its gate (shear_frame_masks)
    MeshArgs proto;          // static part filled at upload; placement filled per frame
    int32_t n_tris = 0, n_pairs = 0;
    // the tree over the triangles themselves that the fast build walks (Packer::build_loose) and its own copy of the triangle
    // records; the ordinary entry and no copy where there is none
    uint32_t loose_root = kLeafBit, loose_pairs = 0;
    TriRec* d_loose_tris = nullptr;
    size_t n_loose_tris = 0;
};

}  // namespace

namespace {
// What rt_render_frame_instanced computed for list entry i last time (see there).
struct InstanceCache {
    bool valid = false;
    rt_transform transform;
    float root_min[3], root_max[3];
    MeshArgs placed;
    InstanceBox box;
};

void copy_placement(const MeshArgs& from, MeshArgs& to) {
    memcpy(to.inv, from.inv, sizeof(to.inv)); memcpy(to.wrow, from.wrow, sizeof(to.wrow)); memcpy(to.nrm, from.nrm, sizeof(to.nrm));
    memcpy(to.cam_o, from.cam_o, sizeof(to.cam_o)); memcpy(to.lit_d, from.lit_d, sizeof(to.lit_d)); memcpy(to.lit_inv, from.lit_inv, sizeof(to.lit_inv));
    to.w_mode = from.w_mode;
}
}  // namespace

constexpr int kRingSlots = 8;      // most frames rt_render_frame_host keeps in flight

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool ev_pending = false;
    std::vector<DeviceMesh> meshes;
    PairNode* d_all_pairs = nullptr;         // PairNodes of all meshes, rebuilt when the mesh list changes
    uint32_t n_all_pairs = 0;
    uint8_t* d_resident = nullptr;           // nodes + triangles of the small meshes, as the lean kernels stage them in LDS
    uint32_t resident_bytes = 0;
    uint32_t stack_entries = 2;
    bool scene_dirty = true;
    TileQueue* d_queue = nullptr;
    DeferQueue* d_defer = nullptr;           // two-pass variant: counters ...
    uint32_t* d_defer_items = nullptr;       // ... and the list of deferred pixels (one slot per pixel of the largest frame)
    size_t defer_capacity = 0;
    unsigned two_pass_frames = 0;            // selects the counter of the next two-pass frame
    void* d_tile_clock = nullptr;           // caller-owned trace buffer (rt_set_trace_buffer)
    DeviceCounters* d_counters = nullptr;
    bool counters_pending = false;
    // planes used by rt_render_frame_host
    uint8_t* d_rgb = nullptr; int32_t* d_id = nullptr; float* d_t = nullptr; uint8_t* d_shadow = nullptr;
    size_t plane_pixels = 0;
    // buffers of rt_upscale_frame_host
    uint8_t* d_up_in = nullptr; uint32_t* d_up_out = nullptr;
    size_t up_in_bytes = 0, up_out_pixels = 0;
    LaunchConfig cfg;
    int timing = 1;
    int budget = 48;               // two-pass variant: node visits per pixel before it is deferred
    int resident_budget = (int)kResidentBudget;   // bytes of small meshes staged in LDS by the lean kernels
    int xcd_swizzle = 1;           // rt_render_frames: give each XCD runs of 8 neighbouring tiles
    int batch_rows = 1;            // rt_render_frames: 1 = each frame scans its tile rows from its own first heavy row, 0 = from the batch's
    int batch_chunk = 0;           // rt_render_frames: consecutive tiles of one frame before the next frame's; 0 = four tile rows
    int coop = 1;                  // cooperative rendering of the big meshes' screen area (variant 3)
    // rt_render_frames: per-frame constants travel through a ring of staging slots (pinned host
    // copy -> device copy on the caller's stream); a slot is reused once its event has completed.
    // rt_render_frame_instanced: per call [MeshArgs x n][PairNode x (2n + 2)] staged through one of
    // two slots (pinned host copy -> device copy on the caller's stream)
    uint8_t* h_inst[kInstSlots] = {};
    uint8_t* d_inst[kInstSlots] = {};
    size_t inst_capacity[kInstSlots] = {};     // instances a slot holds
    hipEvent_t inst_done[kInstSlots] = {};
    bool inst_used[kInstSlots] = {};
    unsigned inst_seq = 0;
    FrameArgs* h_batch = nullptr;
    FrameArgs* d_batch = nullptr;
    hipEvent_t batch_done[kBatchSlots] = {};
    bool batch_used[kBatchSlots] = {};
    unsigned batch_seq = 0;
    int regroup = 1;               // meshes uploaded from now on get a second, surface-area-split tree over the reference's leaves
    int device_build_min = 20000;  // triangle count from which rt_mesh_upload_soup / rt_mesh_rebuild build on the device
    int sampled_split = 1;         // sampled frames: the samples of a tile spread over four waves
    // rt_render_frame_host with option "frames_in_flight" > 1: each call takes the next slot of a ring (its own
    // stream and device image), queues kernel + copy to the caller's buffer and returns; rt_sync waits for all.
    int frames_in_flight = 1;
    hipStream_t ring_stream[kRingSlots] = {};
    hipEvent_t ring_done[kRingSlots] = {};
    uint8_t* ring_rgb[kRingSlots] = {};
    size_t ring_pixels[kRingSlots] = {};
    bool ring_used[kRingSlots] = {};
    unsigned ring_seq = 0;
    // rt_render_frames_sharded: this context's frames wait here (two slots of up to RT_MAX_BATCH images) for their copy to the
    // gathering GPU
    uint8_t* shard_buf[2] = {};
    size_t shard_bytes[2] = {};
    hipEvent_t shard_done[2] = {};
    bool shard_used[2] = {};
    std::vector<InstanceCache> inst_cache;
    StandingTlas standing;         // rt_render_frame_instanced: the hierarchy over the instances, kept between calls
    int keep_hierarchy = 1;
    float box_allowance = 0.0f;    // what the instances' world boxes were widened by last time (scene_box_allowance)
    int screen_cull = 1;           // meshes are skipped for camera rays of tiles their root box cannot reach (mesh_screen_rect)
    int fast = 0;                  // 1: rt_render_frame / rt_render_frames / strips run the fast build of the kernels (not bit-exact)
    int instanced_wave = 0;        // rt_render_frame_instanced: 0 = one lane per pixel, nested loops (faster); 1..8 = the per-wavefront scheduler, that many tiles per wave
    float inst_cache_cam[3] = {0, 0, 0}, inst_cache_light[3] = {0, 0, 0};
    float build_ms = 0.0f;         // last rt_bvh_build_device
    int build_levels = 0, build_launches = 0;
    std::string err;
};

namespace {

int fail(rt_ctx* ctx, int status, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return status;
}

int hip_fail(rt_ctx* ctx, hipError_t e, const char* what) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();   // clear the sticky-less error slot
    return fail(ctx, RT_ERR_DEVICE, msg);
}

#define RT_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

void free_mesh(DeviceMesh& m) {
    if (m.d_tris) (void)hipFree(m.d_tris);
    if (m.d_attrs) (void)hipFree(m.d_attrs);
    if (m.d_texels) (void)hipFree(m.d_texels);
    if (m.d_grid) (void)hipFree(m.d_grid);
    if (m.d_loose_tris) (void)hipFree(m.d_loose_tris);
    m = DeviceMesh();
}

template <class T>
int upload_array(rt_ctx* ctx, const std::vector<T>& src, T** dst) {
    *dst = nullptr;
    // Always allocate at least one record so kernels never see a null base.
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    RT_HIP(ctx, hipMalloc((void**)dst, bytes));
    if (!src.empty()) RT_HIP(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    else RT_HIP(ctx, hipMemset(*dst, 0, bytes));
    return RT_OK;
}

// Puts a packed mesh's triangles, attributes and nodes into dm.  Device arrays are reused when the
// triangle count did not change (a mesh rebuilt every frame); nobody may be reading them.
int set_geometry(rt_ctx* ctx, DeviceMesh& dm, const PackedMesh& pk, int32_t n_tris) {
    const bool reuse = dm.d_tris && dm.d_attrs && dm.n_tris == n_tris && n_tris > 0;
    if (reuse) {
        RT_HIP(ctx, hipMemcpy(dm.d_tris, pk.tris.data(), pk.tris.size() * sizeof(TriRec), hipMemcpyHostToDevice));
        RT_HIP(ctx, hipMemcpy(dm.d_attrs, pk.attrs.data(), pk.attrs.size() * sizeof(AttrRec), hipMemcpyHostToDevice));
    } else {
        TriRec* t = nullptr; AttrRec* a = nullptr;
        int st = upload_array(ctx, pk.tris, &t);
        if (st == RT_OK) st = upload_array(ctx, pk.attrs, &a);
        if (st != RT_OK) { if (t) (void)hipFree(t); if (a) (void)hipFree(a); return st; }
        if (dm.d_tris) (void)hipFree(dm.d_tris);
        if (dm.d_attrs) (void)hipFree(dm.d_attrs);
        dm.d_tris = t; dm.d_attrs = a;
    }
    if (dm.d_grid) { (void)hipFree(dm.d_grid); dm.d_grid = nullptr; }
    if (pk.grid_axis != 3) {
        const int gst = upload_array(ctx, pk.grid, &dm.d_grid);
        if (gst != RT_OK) return gst;
    }
    set_grid_args(pk, dm.d_grid, dm.proto);
    dm.shear_blob.clear();
    if (pk.grid_axis >= 4) dm.shear_blob = pk.grid;
    dm.pairs = pk.pairs; dm.tris = pk.tris; dm.depth = pk.depth; dm.host_tris = true;
    dm.proto.tris = dm.d_tris; dm.proto.attrs = dm.d_attrs;
    memcpy(dm.proto.root_min, pk.root_min, 12);
    memcpy(dm.proto.root_max, pk.root_max, 12);
    dm.proto.root_ref = pk.root_ref; dm.proto.ref_root = pk.ref_root;
    dm.proto.empty = pk.empty ? 1u : 0u;
    dm.proto.wide = pk.wide ? 1u : 0u;
    dm.n_tris = n_tris; dm.n_pairs = (int32_t)pk.tree_pairs;
    dm.loose_root = pk.root_ref; dm.loose_pairs = pk.tree_pairs;
    if (pk.loose_tris.empty()) {
        if (dm.d_loose_tris) { (void)hipFree(dm.d_loose_tris); dm.d_loose_tris = nullptr; }
        dm.n_loose_tris = 0;
    } else {
        if (dm.d_loose_tris && dm.n_loose_tris == pk.loose_tris.size()) {      // a mesh rebuilt every frame keeps its array
            RT_HIP(ctx, hipMemcpy(dm.d_loose_tris, pk.loose_tris.data(), pk.loose_tris.size() * sizeof(TriRec), hipMemcpyHostToDevice));
        } else {
            TriRec* fresh = nullptr;
            const int lst = upload_array(ctx, pk.loose_tris, &fresh);
            if (lst != RT_OK) { if (fresh) (void)hipFree(fresh); return lst; }
            if (dm.d_loose_tris) (void)hipFree(dm.d_loose_tris);
            dm.d_loose_tris = fresh;
            dm.n_loose_tris = pk.loose_tris.size();
        }
        dm.loose_root = pk.loose_root; dm.loose_pairs = pk.loose_pairs;
    }
    return RT_OK;
}

// Triangles in file order -> BVH and packed records in dm.  Large soups are built and packed on the device
// (rt_build.hip) and only the nodes come back; small ones, and trees the device packer leaves to the host
// (a leaf too large for one leaf reference), go through rt_bvh_build + Packer.  tris_out, if given, receives
// the triangles in post-build order (the order triangle ids refer to).
int build_into(rt_ctx* ctx, DeviceMesh& dm, const rt_triangle* tris, int32_t n, rt_triangle* tris_out) {
    std::string why;
    if (n >= ctx->device_build_min && n > 0) {
        DeviceBvh bvh;
        int st = bvh.build(tris, n, ctx->stream, why);
        if (st != RT_OK) return fail(ctx, st, why);
        ctx->build_ms = bvh.stats.device_ms; ctx->build_levels = bvh.stats.levels; ctx->build_launches = bvh.stats.launches;
        PackedOnDevice pk;
        st = bvh.pack(pk, ctx->stream, why);
        if (st == RT_OK) {
            if (tris_out) { st = bvh.triangles_to_host(tris_out, ctx->stream, why); if (st != RT_OK) { (void)hipFree(pk.d_tris); (void)hipFree(pk.d_attrs); return fail(ctx, st, why); } }
            if (pk.pairs.size() >= (size_t)kLeafBit) { (void)hipFree(pk.d_tris); (void)hipFree(pk.d_attrs); return fail(ctx, RT_ERR_UNSUPPORTED, "too many internal nodes"); }
            if (dm.d_tris) (void)hipFree(dm.d_tris);
            if (dm.d_attrs) (void)hipFree(dm.d_attrs);
            dm.d_tris = pk.d_tris; dm.d_attrs = pk.d_attrs;
            if (dm.d_grid) { (void)hipFree(dm.d_grid); dm.d_grid = nullptr; }
            dm.proto.grid = nullptr; dm.proto.grid_axis = 3u; dm.proto.grid_n = 0;        // meshes packed on the device have no grid
            dm.pairs.swap(pk.pairs); dm.tris.clear(); dm.tris.shrink_to_fit(); dm.host_tris = false; dm.depth = pk.depth;
            dm.proto.tris = dm.d_tris; dm.proto.attrs = dm.d_attrs;
            memcpy(dm.proto.root_min, pk.root_min, 12);
            memcpy(dm.proto.root_max, pk.root_max, 12);
            dm.proto.root_ref = pk.root_ref; dm.proto.ref_root = pk.root_ref;
            dm.proto.empty = 0u;
            dm.proto.wide = any_wide_triangle(tris, n) ? 1u : 0u;
            dm.n_tris = n; dm.n_pairs = (int32_t)dm.pairs.size();
            if (ctx->regroup) {
                // the nodes are on the host anyway: the surface-area-split tree over the same leaves is built here (0.4 s for
                // ten million triangles) and travels to the device with the reference's tree when the scene is committed
                PackedMesh both;
                both.pairs.swap(dm.pairs);
                both.root_ref = both.ref_root = pk.root_ref;
                both.tree_pairs = (uint32_t)both.pairs.size();
                both.depth = pk.depth;
                (void)Packer::regroup_packed(both);
                dm.pairs.swap(both.pairs);
                dm.proto.root_ref = both.root_ref; dm.proto.ref_root = both.ref_root;
                dm.depth = both.depth; dm.n_pairs = (int32_t)both.tree_pairs;
            }
            if (dm.d_loose_tris) { (void)hipFree(dm.d_loose_tris); dm.d_loose_tris = nullptr; }
            dm.n_loose_tris = 0;
            dm.loose_root = dm.proto.root_ref; dm.loose_pairs = (uint32_t)dm.n_pairs;
            return RT_OK;
        }
        if (st != RT_ERR_UNSUPPORTED) return fail(ctx, st, why);
        // fall through: the host packer splits oversize leaves, or says why the tree cannot be used
    }
    std::vector<rt_triangle> post(tris, tris + (n > 0 ? n : 0));
    std::vector<rt_flat_node> nodes((size_t)2 * (size_t)(n > 0 ? n : 1) + 1);
    int32_t used = 0;
    int st = rt_bvh_build(post.data(), n, nodes.data(), (int32_t)nodes.size(), &used);
    if (st != RT_OK) return fail(ctx, st, "BVH build failed");
    PackedMesh pk;
    st = Packer(post.data(), n, nodes.data(), used, pk, ctx->regroup != 0).run(why);
    if (st != RT_OK) return fail(ctx, st, why);
    st = set_geometry(ctx, dm, pk, n);
    if (st == RT_OK && tris_out && n > 0) memcpy(tris_out, post.data(), (size_t)n * sizeof(rt_triangle));
    return st;
}

// Rebuilds the combined node array after the mesh list changed.
int commit_scene(rt_ctx* ctx) {
    if (!ctx->scene_dirty) return RT_OK;
    RT_HIP(ctx, hipDeviceSynchronize());              // no launch may still read the old array
    if (ctx->d_all_pairs) (void)hipFree(ctx->d_all_pairs);
    ctx->d_all_pairs = nullptr;
    std::vector<PairNode> all;
    int depth = 0;
    for (auto& m : ctx->meshes) {
        m.proto.pair_base = (uint32_t)all.size();
        m.proto.pair_count = (uint32_t)m.n_pairs;       // nodes of the tree rays enter (the heuristics' measure of a mesh)
        all.insert(all.end(), m.pairs.begin(), m.pairs.end());
        if (m.depth > depth) depth = m.depth;
    }
    int st = upload_array(ctx, all, &ctx->d_all_pairs);
    if (st != RT_OK) return st;
    ctx->n_all_pairs = (uint32_t)all.size();
    // LDS-resident blob: the smallest meshes first, while they fit in kResidentBudget bytes.
    if (ctx->d_resident) (void)hipFree(ctx->d_resident);
    ctx->d_resident = nullptr;
    std::vector<uint8_t> blob;
    std::vector<size_t> order(ctx->meshes.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    auto bytes_of = [&](size_t i) { return ctx->meshes[i].pairs.size() * sizeof(PairNode) + ((ctx->meshes[i].tris.size() * sizeof(TriRec) + 63) & ~(size_t)63); };
    for (size_t i = 1; i < order.size(); i++)
        for (size_t j = i; j > 0 && bytes_of(order[j]) < bytes_of(order[j - 1]); j--) std::swap(order[j], order[j - 1]);
    for (auto& m : ctx->meshes) { m.proto.res_pairs = kNotResident; m.proto.res_tris = kNotResident; }
    for (size_t k : order) {
        DeviceMesh& m = ctx->meshes[k];
        if (m.proto.empty || !m.host_tris || blob.size() + bytes_of(k) > (size_t)ctx->resident_budget) continue;
        m.proto.res_pairs = (uint32_t)blob.size();
        blob.insert(blob.end(), (const uint8_t*)m.pairs.data(), (const uint8_t*)(m.pairs.data() + m.pairs.size()));
        m.proto.res_tris = (uint32_t)blob.size();
        blob.insert(blob.end(), (const uint8_t*)m.tris.data(), (const uint8_t*)(m.tris.data() + m.tris.size()));
        blob.resize((blob.size() + 63) & ~(size_t)63, 0);
    }
    st = upload_array(ctx, blob, &ctx->d_resident);
    if (st != RT_OK) return st;
    ctx->resident_bytes = (uint32_t)blob.size();
    ctx->stack_entries = (uint32_t)depth + 2;        // a chain of `depth` PairNodes holds at most depth+1 entries
    ctx->scene_dirty = false;
    return RT_OK;
}

SceneArgs scene_args(const rt_ctx* ctx) {
    SceneArgs S;
    S.pairs = ctx->d_all_pairs; S.n_pairs = ctx->n_all_pairs; S.stack_entries = ctx->stack_entries;
    S.resident = ctx->d_resident; S.resident_bytes = ctx->resident_bytes;
    return S;
}

// Shadow-ray rectangles of all meshes of a frame (mesh_shadow_rect): they need the bounds of the whole scene first.
void frame_shadow_rects(const rt_transform* transforms, int32_t n_transforms, FrameArgs& A) {
    double slo[3] = {1e300, 1e300, 1e300}, shi[3] = {-1e300, -1e300, -1e300};
    bool any = false;
    for (int32_t m = 0; m < n_transforms; m++) {
        if (A.mesh[m].empty) continue;
        const WorldBox w = world_box(transforms[m], A.mesh[m].root_min, A.mesh[m].root_max);
        for (int i = 0; i < 3; i++) { slo[i] = w.lo[i] < slo[i] ? w.lo[i] : slo[i]; shi[i] = w.hi[i] > shi[i] ? w.hi[i] : shi[i]; }
        any = true;
    }
    if (!any) return;
    for (int32_t m = 0; m < n_transforms; m++)
        mesh_shadow_rect(transforms[m], A.mesh[m].root_min, A.mesh[m].root_max, slo, shi, A, A.vis_shadow[m]);
}

// Whether a call with these flags runs the fast build of the kernels (option "fast"): the self-checking mode compares with the
// reference's way, and RT_FLAG_REFERENCE_ORDER asks for the reference's own walk; both stay with the exact build.
bool runs_fast(const rt_ctx* ctx, uint32_t flags) { return ctx->fast && !(flags & (RT_FLAG_VERIFY_SKIPS | RT_FLAG_REFERENCE_ORDER)); }

// The fast build enters a mesh through the tree over its triangles themselves, where the mesh has one, and reads that
// tree's copy of the triangle records.  Every walk of such a frame goes through it: the leaves of the mesh's other trees
// index the records in the reference's order.
void enter_loose_tree(const DeviceMesh& dm, MeshArgs& M) {
    if (!dm.d_loose_tris) return;
    if (M.grid_axis >= 4u) return;          // a mesh with a tree in a sheared frame keeps it: that tree prunes more than the loose one
    M.root_ref = M.ref_root = dm.loose_root;
    M.pair_count = dm.loose_pairs;
    M.tris = dm.d_loose_tris;
    M.res_pairs = M.res_tris = kNotResident;
}

// Per-frame constants: camera basis, light, and each mesh's placement (inverse matrices, camera
// origin and light direction in mesh space).
void fill_frame(const rt_ctx* ctx, const rt_camera& camera, const rt_light& light, const rt_transform* transforms, int32_t n_transforms,
                int32_t width, int32_t height, uint32_t flags, FrameArgs& A, bool fast_kernels = false) {
    memset(&A, 0, sizeof(A));
    fill_camera(camera, light, width, height, A);
    A.n_mesh = n_transforms;
    A.flags = flags;
    for (int32_t m = 0; m < n_transforms; m++) {
        A.mesh[m] = ctx->meshes[(size_t)m].proto;
        if (fast_kernels) enter_loose_tree(ctx->meshes[(size_t)m], A.mesh[m]);
        fill_placement(transforms[m], A, A.mesh[m]);
        if (A.mesh[m].grid_axis >= 4u) shear_frame_masks(ctx->meshes[(size_t)m].shear_blob.data(), A.mesh[m]);
        if (ctx->screen_cull) mesh_screen_rect(transforms[m], A.mesh[m].root_min, A.mesh[m].root_max, A, A.vis[m]);
        else { A.vis[m][0] = 0; A.vis[m][1] = 0; A.vis[m][2] = (width + 7) / 8; A.vis[m][3] = (height + 7) / 8; }
        for (int i = 0; i < 4; i++) A.vis_shadow[m][i] = i < 2 ? 0 : (i == 2 ? (width + 7) / 8 : (height + 7) / 8);
    }
    if (ctx->screen_cull && n_transforms > 0 && n_transforms <= kMaxMeshes) frame_shadow_rects(transforms, n_transforms, A);
}

// First tile row (8-pixel rows) of the screen area in which a ray of this frame, primary or
// shadow, can meet a big mesh; -1 when there is no big mesh or the area cannot be bounded.
int first_heavy_row(const rt_transform* transforms, int32_t n_transforms, const FrameArgs& A) {
    if (n_transforms > kMaxMeshes) return -1;
    WorldBox boxes[kMaxMeshes];                 // no heap: this runs inside the entry point that allocates nothing
    double slo[3] = {1e300, 1e300, 1e300}, shi[3] = {-1e300, -1e300, -1e300};
    for (int32_t m = 0; m < n_transforms; m++) {
        boxes[(size_t)m] = world_box(transforms[m], A.mesh[m].root_min, A.mesh[m].root_max);
        if (A.mesh[m].empty) continue;
        for (int i = 0; i < 3; i++) {
            slo[i] = boxes[(size_t)m].lo[i] < slo[i] ? boxes[(size_t)m].lo[i] : slo[i];
            shi[i] = boxes[(size_t)m].hi[i] > shi[i] ? boxes[(size_t)m].hi[i] : shi[i];
        }
    }
    int row = -1;
    for (int32_t m = 0; m < n_transforms; m++) {
        if (A.mesh[m].pair_count < kBigMeshPairs) continue;
        int r[4];
        if (!coop_rect(boxes[(size_t)m], slo, shi, A, true, r)) return -1;
        if (r[2] > r[0] && r[3] > r[1] && (row < 0 || r[1] < row)) row = r[1];
    }
    return row;
}

int ensure_planes(rt_ctx* ctx, size_t pixels) {
    if (pixels <= ctx->plane_pixels) return RT_OK;
    if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
    if (ctx->d_id) (void)hipFree(ctx->d_id);
    if (ctx->d_t) (void)hipFree(ctx->d_t);
    if (ctx->d_shadow) (void)hipFree(ctx->d_shadow);
    ctx->d_rgb = nullptr; ctx->d_id = nullptr; ctx->d_t = nullptr; ctx->d_shadow = nullptr;
    ctx->plane_pixels = 0;
    RT_HIP(ctx, hipMalloc((void**)&ctx->d_rgb, pixels * 3));
    RT_HIP(ctx, hipMalloc((void**)&ctx->d_id, pixels * sizeof(int32_t)));
    RT_HIP(ctx, hipMalloc((void**)&ctx->d_t, pixels * sizeof(float)));
    RT_HIP(ctx, hipMalloc((void**)&ctx->d_shadow, pixels));
    ctx->plane_pixels = pixels;
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_status_string(int status) {
    switch (status) {
        case RT_OK: return "ok";
        case RT_ERR_INVALID: return "invalid argument";
        case RT_ERR_DEVICE: return "device error";
        case RT_ERR_NO_DEVICE: return "no usable GPU";
        case RT_ERR_UNSUPPORTED: return "input outside the supported envelope";
        case RT_ERR_CAPACITY: return "output array too small";
        default: return "unknown status";
    }
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rt_ctx_create(int device, rt_ctx** out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return RT_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) return RT_ERR_NO_DEVICE;
    rt_ctx* ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return RT_ERR_DEVICE;
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_end);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_counters, sizeof(DeviceCounters));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_queue, sizeof(TileQueue));
    if (e == hipSuccess) e = hipMemset(ctx->d_queue, 0, sizeof(TileQueue));
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_defer, sizeof(DeferQueue));
    if (e == hipSuccess) e = hipMemset(ctx->d_defer, 0, sizeof(DeferQueue));
    const size_t batch_bytes = (size_t)kBatchSlots * RT_MAX_BATCH * sizeof(FrameArgs);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_batch, batch_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_batch, batch_bytes);
    for (int i = 0; i < kBatchSlots && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->batch_done[i], hipEventDisableTiming);
    for (int i = 0; i < kInstSlots && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->inst_done[i], hipEventDisableTiming);
    int cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e == hipSuccess) { ctx->cfg.queue = ctx->d_queue; ctx->cfg.n_cus = cus > 0 ? cus : 256; }
    // address of the library's file-scope staging array on this GPU (rt_render_frames_noalloc); `device` is current
    if (e == hipSuccess) e = batch_scratch_init(device);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        rt_ctx_destroy(ctx);
        return RT_ERR_DEVICE;
    }
    *out = ctx;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& m : ctx->meshes) free_mesh(m);
    if (ctx->d_all_pairs) (void)hipFree(ctx->d_all_pairs);
    if (ctx->d_resident) (void)hipFree(ctx->d_resident);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_defer) (void)hipFree(ctx->d_defer);
    if (ctx->d_defer_items) (void)hipFree(ctx->d_defer_items);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    for (int i = 0; i < kBatchSlots; i++) {
        if (!ctx->batch_done[i]) continue;
        if (ctx->batch_used[i]) (void)hipEventSynchronize(ctx->batch_done[i]);
        (void)hipEventDestroy(ctx->batch_done[i]);
    }
    for (int i = 0; i < kInstSlots; i++) {
        if (ctx->inst_done[i]) {
            if (ctx->inst_used[i]) (void)hipEventSynchronize(ctx->inst_done[i]);
            (void)hipEventDestroy(ctx->inst_done[i]);
        }
        if (ctx->h_inst[i]) (void)hipHostFree(ctx->h_inst[i]);
        if (ctx->d_inst[i]) (void)hipFree(ctx->d_inst[i]);
    }
    for (int i = 0; i < kRingSlots; i++) {
        if (ctx->ring_stream[i]) { (void)hipStreamSynchronize(ctx->ring_stream[i]); (void)hipStreamDestroy(ctx->ring_stream[i]); }
        if (ctx->ring_done[i]) (void)hipEventDestroy(ctx->ring_done[i]);
        if (ctx->ring_rgb[i]) (void)hipFree(ctx->ring_rgb[i]);
    }
    for (int i = 0; i < 2; i++) {
        if (ctx->shard_done[i]) { if (ctx->shard_used[i]) (void)hipEventSynchronize(ctx->shard_done[i]); (void)hipEventDestroy(ctx->shard_done[i]); }
        if (ctx->shard_buf[i]) (void)hipFree(ctx->shard_buf[i]);
    }
    if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    if (ctx->d_rgb) (void)hipFree(ctx->d_rgb);
    if (ctx->d_id) (void)hipFree(ctx->d_id);
    if (ctx->d_t) (void)hipFree(ctx->d_t);
    if (ctx->d_shadow) (void)hipFree(ctx->d_shadow);
    if (ctx->d_up_in) (void)hipFree(ctx->d_up_in);
    if (ctx->d_up_out) (void)hipFree(ctx->d_up_out);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

int rt_bvh_build(rt_triangle* tris, int32_t n, rt_flat_node* nodes_out, int32_t cap, int32_t* n_nodes) {
    if (n < 0 || (n > 0 && !tris) || !nodes_out || !n_nodes || cap < 1) return RT_ERR_INVALID;
    BvhBuilder b(tris, nodes_out, cap);
    if (b.build(0, n) < 0) return RT_ERR_CAPACITY;
    *n_nodes = b.used();
    return RT_OK;
}

int rt_bvh_build_device(rt_ctx* ctx, rt_triangle* tris, int32_t n, rt_flat_node* nodes_out, int32_t cap, int32_t* n_nodes) {
    if (!ctx) return RT_ERR_INVALID;
    if (n < 0 || (n > 0 && !tris) || !nodes_out || !n_nodes || cap < 1) return fail(ctx, RT_ERR_INVALID, "null or negative build arguments");
    if (n > RT_MAX_MESH_TRIS) return fail(ctx, RT_ERR_UNSUPPORTED, "too many triangles in one mesh");
    if (n == 0) return rt_bvh_build(tris, 0, nodes_out, cap, n_nodes);          // one empty node, nothing for the GPU to do
    RT_HIP(ctx, hipSetDevice(ctx->device));
    BuildStats stats;
    std::string err;
    const int rc = build_bvh_on_device(tris, n, nodes_out, cap, n_nodes, ctx->stream, err, &stats);
    if (rc != RT_OK) return fail(ctx, rc, err);
    ctx->build_ms = stats.device_ms; ctx->build_levels = stats.levels; ctx->build_launches = stats.launches;
    return RT_OK;
}

int rt_bvh_build_stats(const rt_ctx* ctx, float* device_ms, int32_t* levels, int32_t* launches) {
    if (!ctx) return RT_ERR_INVALID;
    if (device_ms) *device_ms = ctx->build_ms;
    if (levels) *levels = ctx->build_levels;
    if (launches) *launches = ctx->build_launches;
    return RT_OK;
}

int rt_bvh_flatten_tree(const rt_tree_node* root, rt_flat_node* nodes_out, int32_t cap, int32_t* n_nodes) {
    if (!root || !nodes_out || !n_nodes || cap < 1) return RT_ERR_INVALID;
    int32_t used = 0;
    if (flatten_tree(root, nodes_out, cap, used, 0) < 0) return RT_ERR_CAPACITY;
    *n_nodes = used;
    return RT_OK;
}

int rt_mesh_upload(rt_ctx* ctx, const rt_triangle* tris, int32_t n_tris, const rt_flat_node* nodes, int32_t n_nodes,
                   const uint8_t* h_rgba, int32_t tex_width, int32_t tex_height, int32_t* mesh_id) {
    if (!ctx) return RT_ERR_INVALID;
    if (!h_rgba || tex_width <= 0 || tex_height <= 0) return fail(ctx, RT_ERR_INVALID, "texture missing or empty");
    if ((int64_t)tex_width * tex_height > (int64_t)1 << 28) return fail(ctx, RT_ERR_UNSUPPORTED, "texture too large");
    if ((int)ctx->meshes.size() >= RT_MAX_GEOMETRIES) return fail(ctx, RT_ERR_UNSUPPORTED, "too many meshes uploaded");
    PackedMesh pk;
    std::string why;
    int st = Packer(tris, n_tris, nodes, n_nodes, pk, ctx->regroup != 0).run(why);
    if (st != RT_OK) return fail(ctx, st, why);

    RT_HIP(ctx, hipSetDevice(ctx->device));
    DeviceMesh dm;
    memset(&dm.proto, 0, sizeof(dm.proto));
    st = set_geometry(ctx, dm, pk, n_tris);
    if (st == RT_OK) {
        const size_t bytes = (size_t)tex_width * (size_t)tex_height * 4;
        hipError_t e = hipMalloc((void**)&dm.d_texels, bytes);
        if (e == hipSuccess) e = hipMemcpy(dm.d_texels, h_rgba, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = hip_fail(ctx, e, "texture upload");
    }
    if (st != RT_OK) { free_mesh(dm); return st; }
    dm.proto.texels = dm.d_texels;
    dm.proto.tex_w = tex_width; dm.proto.tex_h = tex_height;
    ctx->meshes.push_back(std::move(dm));
    ctx->scene_dirty = true;
    if (mesh_id) *mesh_id = (int32_t)ctx->meshes.size() - 1;
    return RT_OK;
}

int rt_mesh_replace(rt_ctx* ctx, int32_t mesh_id, const rt_triangle* tris, int32_t n_tris, const rt_flat_node* nodes, int32_t n_nodes) {
    if (!ctx) return RT_ERR_INVALID;
    if (mesh_id < 0 || mesh_id >= (int32_t)ctx->meshes.size()) return fail(ctx, RT_ERR_INVALID, "no such mesh");
    PackedMesh pk;
    std::string why;
    int st = Packer(tris, n_tris, nodes, n_nodes, pk, ctx->regroup != 0).run(why);
    if (st != RT_OK) return fail(ctx, st, why);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipDeviceSynchronize());                 // frames in flight still read the old arrays
    st = set_geometry(ctx, ctx->meshes[(size_t)mesh_id], pk, n_tris);
    ctx->scene_dirty = true;
    return st;
}

int rt_mesh_upload_soup(rt_ctx* ctx, const rt_triangle* tris, int32_t n_tris, const uint8_t* h_rgba, int32_t tex_width, int32_t tex_height,
                        int32_t* mesh_id, rt_triangle* tris_out) {
    if (!ctx) return RT_ERR_INVALID;
    if (n_tris < 0 || (n_tris > 0 && !tris)) return fail(ctx, RT_ERR_INVALID, "null or negative triangle array");
    if (n_tris > RT_MAX_MESH_TRIS) return fail(ctx, RT_ERR_UNSUPPORTED, "too many triangles in one mesh");
    if (!h_rgba || tex_width <= 0 || tex_height <= 0) return fail(ctx, RT_ERR_INVALID, "texture missing or empty");
    if ((int64_t)tex_width * tex_height > (int64_t)1 << 28) return fail(ctx, RT_ERR_UNSUPPORTED, "texture too large");
    if ((int)ctx->meshes.size() >= RT_MAX_GEOMETRIES) return fail(ctx, RT_ERR_UNSUPPORTED, "too many meshes uploaded");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    DeviceMesh dm;
    memset(&dm.proto, 0, sizeof(dm.proto));
    int st = build_into(ctx, dm, tris, n_tris, tris_out);
    if (st == RT_OK) {
        const size_t bytes = (size_t)tex_width * (size_t)tex_height * 4;
        hipError_t e = hipMalloc((void**)&dm.d_texels, bytes);
        if (e == hipSuccess) e = hipMemcpy(dm.d_texels, h_rgba, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = hip_fail(ctx, e, "texture upload");
    }
    if (st != RT_OK) { free_mesh(dm); return st; }
    dm.proto.texels = dm.d_texels;
    dm.proto.tex_w = tex_width; dm.proto.tex_h = tex_height;
    ctx->meshes.push_back(std::move(dm));
    ctx->scene_dirty = true;
    if (mesh_id) *mesh_id = (int32_t)ctx->meshes.size() - 1;
    return RT_OK;
}

int rt_mesh_rebuild(rt_ctx* ctx, int32_t mesh_id, const rt_triangle* tris, int32_t n_tris, rt_triangle* tris_out) {
    if (!ctx) return RT_ERR_INVALID;
    if (mesh_id < 0 || mesh_id >= (int32_t)ctx->meshes.size()) return fail(ctx, RT_ERR_INVALID, "no such mesh");
    if (n_tris < 0 || (n_tris > 0 && !tris)) return fail(ctx, RT_ERR_INVALID, "null or negative triangle array");
    if (n_tris > RT_MAX_MESH_TRIS) return fail(ctx, RT_ERR_UNSUPPORTED, "too many triangles in one mesh");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipDeviceSynchronize());                 // frames in flight still read the old arrays
    const int st = build_into(ctx, ctx->meshes[(size_t)mesh_id], tris, n_tris, tris_out);
    ctx->scene_dirty = true;
    return st;
}

int rt_mesh_upload_tree(rt_ctx* ctx, const rt_triangle* tris, int32_t n_tris, const rt_tree_node* root,
                        const uint8_t* h_rgba, int32_t tex_width, int32_t tex_height, int32_t* mesh_id) {
    if (!ctx) return RT_ERR_INVALID;
    if (!root) return fail(ctx, RT_ERR_INVALID, "null BVH root");
    std::vector<rt_flat_node> flat((size_t)(n_tris > 0 ? 2 * (size_t)n_tris : 1) + 1);
    int32_t used = 0;
    if (flatten_tree(root, flat.data(), (int32_t)flat.size(), used, 0) < 0) return fail(ctx, RT_ERR_UNSUPPORTED, "BVH has more nodes than 2*triangles+1");
    return rt_mesh_upload(ctx, tris, n_tris, flat.data(), used, h_rgba, tex_width, tex_height, mesh_id);
}

int rt_scene_clear(rt_ctx* ctx) {
    if (!ctx) return RT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& m : ctx->meshes) free_mesh(m);
    ctx->meshes.clear();
    ctx->scene_dirty = true;
    return RT_OK;
}

int rt_mesh_count(const rt_ctx* ctx) { return ctx ? (int)ctx->meshes.size() : RT_ERR_INVALID; }

}  // extern "C"

namespace {

// Rows of a frame `height` rows tall that belong to strips offset, offset + stride, ... (strips of eight rows).
int32_t strip_rows(int32_t height, int32_t stride, int32_t offset) {
    const int32_t total = (height + 7) / 8;
    if (offset >= total) return 0;
    const int32_t k = (total - offset + stride - 1) / stride;
    int32_t rows = 8 * k;
    if (offset + (k - 1) * stride == total - 1 && height % 8) rows -= 8 - height % 8;
    return rows;
}

// Turns A (filled for the whole frame) into the launch of one share of its strips.  false: nothing to render.
bool take_strips(FrameArgs& A, int32_t stride, int32_t offset) {
    const int32_t rows = strip_rows(A.height, stride, offset);
    A.full_height = A.height; A.strip_stride = stride; A.strip_offset = offset;
    A.height = rows;
    return rows > 0;
}

int render_frame_impl(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms,
                      int32_t n_transforms, int32_t width, int32_t height, int32_t stride, int32_t offset, uint32_t flags, void* d_rgb,
                      void* d_hit_id, void* d_hit_t, void* d_shadow, void* stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!camera || !light || !d_rgb) return fail(ctx, RT_ERR_INVALID, "camera, light and d_rgb are required");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    if (ctx->meshes.size() > (size_t)kMaxMeshes) return fail(ctx, RT_ERR_UNSUPPORTED, "more than RT_MAX_MESHES meshes: use rt_render_frame_instanced");
    if (n_transforms != (int32_t)ctx->meshes.size() || (n_transforms > 0 && !transforms))
        return fail(ctx, RT_ERR_INVALID, "one transform per uploaded mesh is required");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int cst = commit_scene(ctx);
    if (cst != RT_OK) return cst;
    const SceneArgs S = scene_args(ctx);

    FrameArgs A;
    fill_frame(ctx, *camera, *light, transforms, n_transforms, width, height, flags, A, runs_fast(ctx, flags));
    const bool share = stride != 1 || offset != 0;        // some of the frame's strips: plain tiles in launch order
    if (share && !take_strips(A, stride, offset)) return RT_OK;
    LaunchConfig cfg = ctx->cfg;
    if (share) cfg.variant = 0;
    // Work layout: screen rectangles where rays can meet a big mesh are rendered cooperatively and
    // launched first.  Not used when the caller asks for the reference's visiting order or for work
    // counts, when a rectangle cannot be bounded, or when they cover more than a quarter of the frame.
    if (cfg.variant == 3 && (flags & (RT_FLAG_COUNT_WORK | RT_FLAG_REFERENCE_ORDER | RT_FLAG_VERIFY_SKIPS)) == 0 && ctx->coop) {
        std::vector<WorldBox> boxes((size_t)n_transforms);
        double slo[3] = {1e300, 1e300, 1e300}, shi[3] = {-1e300, -1e300, -1e300};
        for (int32_t m = 0; m < n_transforms; m++) {
            boxes[(size_t)m] = world_box(transforms[m], A.mesh[m].root_min, A.mesh[m].root_max);
            if (A.mesh[m].empty) continue;
            for (int i = 0; i < 3; i++) {
                slo[i] = boxes[(size_t)m].lo[i] < slo[i] ? boxes[(size_t)m].lo[i] : slo[i];
                shi[i] = boxes[(size_t)m].hi[i] > shi[i] ? boxes[(size_t)m].hi[i] : shi[i];
            }
        }
        bool ok = true;
        long long area = 0;
        int n = 0;
        for (int32_t m = 0; m < n_transforms && ok; m++) {
            if (A.mesh[m].pair_count < kBigMeshPairs) continue;
            int r[4];
            ok = n < kMaxCoopRects && coop_rect(boxes[(size_t)m], slo, shi, A, ctx->coop == 1, r);
            if (ok && r[2] > r[0] && r[3] > r[1]) {
                for (int i = 0; i < 4; i++) A.rect[n][i] = r[i];
                area += (long long)(r[2] - r[0]) * (r[3] - r[1]);
                n++;
            }
        }
        const long long tiles = (long long)((width + 7) / 8) * ((height + 7) / 8);
        A.n_rects = (ok && area * 4 <= tiles) ? n : 0;
    }
    OutputPlanes O;
    O.rgb = (uint8_t*)d_rgb; O.hit_id = (int32_t*)d_hit_id; O.hit_t = (float*)d_hit_t; O.shadow = (uint8_t*)d_shadow;
    O.counters = ctx->d_counters;
    O.tile_clock = (unsigned long long*)ctx->d_tile_clock;
    if (flags & (RT_FLAG_COUNT_WORK | RT_FLAG_VERIFY_SKIPS)) {
        RT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(DeviceCounters), s));
        ctx->counters_pending = true;
    }
    if (ctx->timing) RT_HIP(ctx, hipEventRecord(ctx->ev_begin, s));
    if (kExperimentalVariants && cfg.variant == 4 && (flags & (RT_FLAG_COUNT_WORK | RT_FLAG_REFERENCE_ORDER | RT_FLAG_VERIFY_SKIPS)) == 0) {
        const size_t pixels = (size_t)width * (size_t)height;
        if (pixels > ctx->defer_capacity) {
            RT_HIP(ctx, hipDeviceSynchronize());          // an earlier frame may still be reading the old list
            if (ctx->d_defer_items) (void)hipFree(ctx->d_defer_items);
            ctx->d_defer_items = nullptr; ctx->defer_capacity = 0;
            RT_HIP(ctx, hipMalloc((void**)&ctx->d_defer_items, pixels * sizeof(uint32_t)));
            ctx->defer_capacity = pixels;
        }
        cfg.defer = ctx->d_defer;
        cfg.defer_items = ctx->d_defer_items;
        cfg.defer_capacity = (uint32_t)ctx->defer_capacity;
        cfg.parity = (int)(ctx->two_pass_frames++ & 1u);
        A.defer_budget = ctx->budget;
    }
    // the self-check compares against the reference's own arithmetic: it always runs the exact build
    hipError_t le = runs_fast(ctx, flags) ? rt_fast_launch_render(&A, &S, &O, &cfg, s) : launch_render(A, S, O, cfg, s);
    if (le != hipSuccess) {
        // leave the work-distribution words of the persistent / two-pass variants in their idle state
        (void)hipDeviceSynchronize();
        (void)hipMemset(ctx->d_queue, 0, sizeof(TileQueue));
        (void)hipMemset(ctx->d_defer, 0, sizeof(DeferQueue));
        return hip_fail(ctx, le, "kernel launch");
    }
    if (ctx->timing) { RT_HIP(ctx, hipEventRecord(ctx->ev_end, s)); ctx->ev_pending = true; }
    return RT_OK;
}

// One frame through the next slot of the ring: kernel and copy are queued on the slot's stream and the call
// returns.  Frames in flight overlap on the GPU, so the plain tile kernel is used: the tiles that run long in
// one frame are covered by the next frame's tiles, as in a batch.  The variants that keep work lists in the
// context would have to run one after another.
int render_frame_in_flight(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms, int32_t n_transforms,
                           int32_t width, int32_t height, uint32_t flags, uint8_t* h_rgb) {
    const size_t pixels = (size_t)width * (size_t)height;
    const int slots = ctx->frames_in_flight < kRingSlots ? ctx->frames_in_flight : kRingSlots;
    const int slot = (int)(ctx->ring_seq++ % (unsigned)slots);
    if (!ctx->ring_stream[slot]) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->ring_stream[slot], hipStreamNonBlocking));
    if (!ctx->ring_done[slot]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ring_done[slot], hipEventDisableTiming));
    if (ctx->ring_used[slot]) { RT_HIP(ctx, hipEventSynchronize(ctx->ring_done[slot])); ctx->ring_used[slot] = false; }
    if (pixels > ctx->ring_pixels[slot]) {
        if (ctx->ring_rgb[slot]) (void)hipFree(ctx->ring_rgb[slot]);
        ctx->ring_rgb[slot] = nullptr; ctx->ring_pixels[slot] = 0;
        RT_HIP(ctx, hipMalloc((void**)&ctx->ring_rgb[slot], pixels * 3));
        ctx->ring_pixels[slot] = pixels;
    }
    const int variant = ctx->cfg.variant, timing = ctx->timing;
    ctx->cfg.variant = 0; ctx->timing = 0;             // no per-frame event pair either: its times would overlap
    const int st = render_frame_impl(ctx, camera, light, transforms, n_transforms, width, height, 1, 0, flags, ctx->ring_rgb[slot], nullptr, nullptr,
                                     nullptr, ctx->ring_stream[slot]);
    ctx->cfg.variant = variant; ctx->timing = timing;
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpyAsync(h_rgb, ctx->ring_rgb[slot], pixels * 3, hipMemcpyDeviceToHost, ctx->ring_stream[slot]));
    RT_HIP(ctx, hipEventRecord(ctx->ring_done[slot], ctx->ring_stream[slot]));
    ctx->ring_used[slot] = true;
    return RT_OK;
}

int bad_strips(rt_ctx* ctx, int32_t stride, int32_t offset) {
    if (!ctx) return RT_ERR_INVALID;
    if (stride < 1 || offset < 0 || offset >= stride) return fail(ctx, RT_ERR_INVALID, "strips: need stride >= 1 and 0 <= offset < stride");
    return RT_OK;
}

}  // namespace

namespace {
// rt_render_frames and rt_render_frames_noalloc.  The second form touches no allocator: the combined node
// array must already be on the device (any earlier rendering call put it there), the per-frame constants go
// from the context's page-locked staging slots into the library's file-scope device array, and the staging
// slots, their events and the stream exist since rt_ctx_create.
int render_frames_impl(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms, int32_t n_frames,
                       int32_t n_transforms, int32_t width, int32_t height, void* d_rgb, size_t frame_stride, void* stream, bool noalloc) {
    if (!ctx) return RT_ERR_INVALID;
    if (!camera || !light || !d_rgb) return fail(ctx, RT_ERR_INVALID, "camera, light and d_rgb are required");
    if (n_frames < 1) return fail(ctx, RT_ERR_INVALID, "n_frames must be 1..RT_MAX_BATCH");
    if (n_frames > RT_MAX_BATCH)
        return fail(ctx, noalloc ? RT_ERR_CAPACITY : RT_ERR_INVALID,
                    noalloc ? "n_frames exceeds RT_MAX_BATCH, the size of the library's static staging array" : "n_frames must be 1..RT_MAX_BATCH");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    if (frame_stride < (size_t)width * (size_t)height * 3) return fail(ctx, RT_ERR_INVALID, "frame_stride smaller than one image");
    if (ctx->meshes.size() > (size_t)kMaxMeshes) return fail(ctx, RT_ERR_UNSUPPORTED, "more than RT_MAX_MESHES meshes: use rt_render_frame_instanced");
    if (n_transforms != (int32_t)ctx->meshes.size() || (n_transforms > 0 && !transforms))
        return fail(ctx, RT_ERR_INVALID, "one transform per uploaded mesh and frame is required");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (noalloc) {
        if (ctx->scene_dirty)
            return fail(ctx, RT_ERR_UNSUPPORTED, "the mesh list changed since the last rendering call: its node array is not on the device yet "
                                                 "(render once through an allocating entry point, e.g. rt_render_frames, first)");
        if ((size_t)n_frames * sizeof(FrameArgs) > batch_scratch_bytes()) return fail(ctx, RT_ERR_CAPACITY, "batch larger than the static staging array");
        if (ctx->fast) return fail(ctx, RT_ERR_UNSUPPORTED, "rt_render_frames_noalloc renders rt_render_frames' exact bytes: switch option \"fast\" off");
    } else {
        int cst = commit_scene(ctx);
        if (cst != RT_OK) return cst;
    }
    const SceneArgs S = scene_args(ctx);

    const int slot = (int)(ctx->batch_seq++ % (unsigned)kBatchSlots);
    if (ctx->batch_used[slot]) RT_HIP(ctx, hipEventSynchronize(ctx->batch_done[slot]));
    FrameArgs* h = ctx->h_batch + (size_t)slot * RT_MAX_BATCH;
    FrameArgs* d = ctx->d_batch + (size_t)slot * RT_MAX_BATCH;
    int first_row = -1;
    bool bounded = true;
    for (int32_t f = 0; f < n_frames; f++) {
        const rt_transform* tf = transforms ? transforms + (size_t)f * (size_t)n_transforms : nullptr;
        fill_frame(ctx, *camera, *light, tf, n_transforms, width, height, 0u, h[f], ctx->fast != 0);
        const int row = first_heavy_row(tf, n_transforms, h[f]);
        h[f].first_tile_row = row > 0 && row < (height + 7) / 8 ? row : 0;
        if (row < 0) bounded = false;
        else if (first_row < 0 || row < first_row) first_row = row;
    }
    if (!bounded || first_row < 0) first_row = 0;
    if (ctx->batch_rows) first_row = -1;               // every frame of the batch starts at its own heavy rows
    const int chunk = ctx->batch_chunk > 0 ? ctx->batch_chunk : (4 * ((width + 7) / 8) + 63) / 64 * 64;
    hipError_t le;
    if (noalloc) {
        if (ctx->timing) RT_HIP(ctx, hipEventRecord(ctx->ev_begin, s));
        le = launch_batch_noalloc(ctx->device, h, n_frames, width, height, S, (uint8_t*)d_rgb, frame_stride, first_row, chunk, ctx->xcd_swizzle,
                                  ctx->cfg.block, s);
    } else {
        RT_HIP(ctx, hipMemcpyAsync(d, h, (size_t)n_frames * sizeof(FrameArgs), hipMemcpyHostToDevice, s));
        if (ctx->timing) RT_HIP(ctx, hipEventRecord(ctx->ev_begin, s));
        le = ctx->fast ? rt_fast_launch_batch(d, n_frames, width, height, &S, (uint8_t*)d_rgb, frame_stride, first_row, chunk, ctx->xcd_swizzle, ctx->cfg.block, s)
                       : launch_batch(d, n_frames, width, height, S, (uint8_t*)d_rgb, frame_stride, first_row, chunk, ctx->xcd_swizzle, ctx->cfg.block, s);
    }
    if (le != hipSuccess) return hip_fail(ctx, le, "kernel launch");
    if (ctx->timing) { RT_HIP(ctx, hipEventRecord(ctx->ev_end, s)); ctx->ev_pending = true; }
    RT_HIP(ctx, hipEventRecord(ctx->batch_done[slot], s));
    ctx->batch_used[slot] = true;
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_render_frame(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms,
                    int32_t n_transforms, int32_t width, int32_t height, uint32_t flags, void* d_rgb, void* d_hit_id,
                    void* d_hit_t, void* d_shadow, void* stream) {
    return render_frame_impl(ctx, camera, light, transforms, n_transforms, width, height, 1, 0, flags, d_rgb, d_hit_id, d_hit_t, d_shadow, stream);
}

int rt_strip_rows(int32_t height, int32_t strip_stride, int32_t strip_offset) {
    if (height <= 0 || strip_stride < 1 || strip_offset < 0 || strip_offset >= strip_stride) return RT_ERR_INVALID;
    return strip_rows(height, strip_stride, strip_offset);
}

int rt_render_frame_strips(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms,
                           int32_t n_transforms, int32_t width, int32_t height, int32_t strip_stride, int32_t strip_offset, uint32_t flags,
                           void* d_rgb, void* d_hit_id, void* d_hit_t, void* d_shadow, void* stream) {
    const int st = bad_strips(ctx, strip_stride, strip_offset);
    if (st != RT_OK) return st;
    return render_frame_impl(ctx, camera, light, transforms, n_transforms, width, height, strip_stride, strip_offset, flags, d_rgb, d_hit_id,
                             d_hit_t, d_shadow, stream);
}

int rt_render_frames(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms, int32_t n_frames,
                     int32_t n_transforms, int32_t width, int32_t height, void* d_rgb, size_t frame_stride, void* stream) {
    return render_frames_impl(ctx, camera, light, transforms, n_frames, n_transforms, width, height, d_rgb, frame_stride, stream, false);
}

int rt_render_frames_noalloc(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms, int32_t n_frames,
                             int32_t n_transforms, int32_t width, int32_t height, void* d_rgb, size_t frame_stride, void* stream) {
    return render_frames_impl(ctx, camera, light, transforms, n_frames, n_transforms, width, height, d_rgb, frame_stride, stream, true);
}

}  // extern "C"

namespace {
int render_sampled_impl(rt_ctx* ctx, const rt_camera* camera, const rt_light* lights, int32_t n_lights, const rt_transform* transforms,
                        int32_t n_transforms, int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed, int32_t stride,
                        int32_t offset, uint32_t flags, void* d_rgb, void* d_hit_id, void* stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!camera || !lights || !d_rgb) return fail(ctx, RT_ERR_INVALID, "camera, lights and d_rgb are required");
    if (n_lights < 1 || n_lights > RT_MAX_LIGHTS) return fail(ctx, RT_ERR_UNSUPPORTED, "n_lights must be 1..RT_MAX_LIGHTS");
    if (samples_per_pixel < 1 || samples_per_pixel > RT_MAX_SAMPLES) return fail(ctx, RT_ERR_UNSUPPORTED, "samples_per_pixel must be 1..RT_MAX_SAMPLES");
    if (flags & RT_FLAG_REFERENCE_ORDER) return fail(ctx, RT_ERR_UNSUPPORTED, "RT_FLAG_REFERENCE_ORDER is not available for sampled frames");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    if (ctx->meshes.size() > (size_t)kMaxMeshes) return fail(ctx, RT_ERR_UNSUPPORTED, "more than RT_MAX_MESHES meshes: use rt_render_frame_instanced");
    if (n_transforms != (int32_t)ctx->meshes.size() || (n_transforms > 0 && !transforms))
        return fail(ctx, RT_ERR_INVALID, "one transform per uploaded mesh is required");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int cst = commit_scene(ctx);
    if (cst != RT_OK) return cst;
    const SceneArgs S = scene_args(ctx);
    FrameArgs A;
    fill_frame(ctx, *camera, lights[0], transforms, n_transforms, width, height, flags, A);
    SampleArgs X;
    fill_samples(lights, n_lights, samples_per_pixel, seed, A, X);
    if ((stride != 1 || offset != 0) && !take_strips(A, stride, offset)) return RT_OK;
    OutputPlanes O;
    O.rgb = (uint8_t*)d_rgb; O.hit_id = (int32_t*)d_hit_id; O.hit_t = nullptr; O.shadow = nullptr;
    O.counters = ctx->d_counters; O.tile_clock = nullptr;
    if (flags & (RT_FLAG_COUNT_WORK | RT_FLAG_VERIFY_SKIPS)) {
        RT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(DeviceCounters), s));
        ctx->counters_pending = true;
    }
    if (ctx->timing) RT_HIP(ctx, hipEventRecord(ctx->ev_begin, s));
    hipError_t le = launch_sampled(A, X, S, O, ctx->cfg.block, ctx->sampled_split, s);
    if (le != hipSuccess) return hip_fail(ctx, le, "kernel launch");
    if (ctx->timing) { RT_HIP(ctx, hipEventRecord(ctx->ev_end, s)); ctx->ev_pending = true; }
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_render_frame_sampled(rt_ctx* ctx, const rt_camera* camera, const rt_light* lights, int32_t n_lights, const rt_transform* transforms,
                            int32_t n_transforms, int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed, uint32_t flags,
                            void* d_rgb, void* d_hit_id, void* stream) {
    return render_sampled_impl(ctx, camera, lights, n_lights, transforms, n_transforms, width, height, samples_per_pixel, seed, 1, 0, flags,
                               d_rgb, d_hit_id, stream);
}

int rt_render_frame_sampled_strips(rt_ctx* ctx, const rt_camera* camera, const rt_light* lights, int32_t n_lights, const rt_transform* transforms,
                                   int32_t n_transforms, int32_t width, int32_t height, int32_t samples_per_pixel, uint32_t seed,
                                   int32_t strip_stride, int32_t strip_offset, uint32_t flags, void* d_rgb, void* d_hit_id, void* stream) {
    const int st = bad_strips(ctx, strip_stride, strip_offset);
    if (st != RT_OK) return st;
    return render_sampled_impl(ctx, camera, lights, n_lights, transforms, n_transforms, width, height, samples_per_pixel, seed, strip_stride,
                               strip_offset, flags, d_rgb, d_hit_id, stream);
}

int rt_render_frame_instanced(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_instance* instances, int32_t n_instances,
                              int32_t width, int32_t height, uint32_t flags, void* d_rgb, void* d_hit_instance, void* d_hit_tri,
                              void* d_hit_t, void* d_shadow, void* stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!camera || !light || !d_rgb) return fail(ctx, RT_ERR_INVALID, "camera, light and d_rgb are required");
    if (n_instances < 0 || (n_instances > 0 && !instances)) return fail(ctx, RT_ERR_INVALID, "bad instance list");
    if (n_instances > RT_MAX_INSTANCES) return fail(ctx, RT_ERR_UNSUPPORTED, "more than RT_MAX_INSTANCES instances");
    if (flags & RT_FLAG_REFERENCE_ORDER) return fail(ctx, RT_ERR_UNSUPPORTED, "RT_FLAG_REFERENCE_ORDER is not available for instanced scenes");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    for (int32_t i = 0; i < n_instances; i++)
        if (instances[i].mesh < 0 || instances[i].mesh >= (int32_t)ctx->meshes.size()) return fail(ctx, RT_ERR_INVALID, "instance of a mesh that was not uploaded");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int cst = commit_scene(ctx);
    if (cst != RT_OK) return cst;
    const SceneArgs S = scene_args(ctx);

    // staging slot, grown to the instance count
    const int slot = (int)(ctx->inst_seq++ % (unsigned)kInstSlots);
    if (ctx->inst_used[slot]) RT_HIP(ctx, hipEventSynchronize(ctx->inst_done[slot]));
    const size_t n = (size_t)(n_instances > 0 ? n_instances : 1);
    // a kept hierarchy that was patched holds its old leaves too: room for twice the instances
    const size_t off_pairs = (n * sizeof(MeshArgs) + 63) & ~(size_t)63, total = off_pairs + (2 * n + 2) * sizeof(PairNode);
    if (n > ctx->inst_capacity[slot]) {
        if (ctx->h_inst[slot]) (void)hipHostFree(ctx->h_inst[slot]);
        if (ctx->d_inst[slot]) (void)hipFree(ctx->d_inst[slot]);
        ctx->h_inst[slot] = nullptr; ctx->d_inst[slot] = nullptr; ctx->inst_capacity[slot] = 0;
        RT_HIP(ctx, hipHostMalloc((void**)&ctx->h_inst[slot], total, hipHostMallocDefault));
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_inst[slot], total));
        ctx->inst_capacity[slot] = n;
    }
    uint8_t* h = ctx->h_inst[slot];
    uint8_t* d = ctx->d_inst[slot];

    FrameArgs A;
    fill_frame(ctx, *camera, *light, nullptr, 0, width, height, flags, A);
    MeshArgs* inst = reinterpret_cast<MeshArgs*>(h);
    std::vector<InstanceBox> boxes;
    std::vector<uint32_t> ids;
    boxes.reserve((size_t)n_instances); ids.reserve((size_t)n_instances);
    // Placement constants and world boxes are kept from the previous call: an entry whose transform (and the
    // camera position and light direction its constants depend on) did not change is copied, not recomputed --
    // in a scene of thousands of placed meshes most stand still from frame to frame.
    if (memcmp(ctx->inst_cache_cam, A.cam_pos, 12) != 0 || memcmp(ctx->inst_cache_light, A.light_dir, 12) != 0) {
        ctx->inst_cache.clear();
        memcpy(ctx->inst_cache_cam, A.cam_pos, 12); memcpy(ctx->inst_cache_light, A.light_dir, 12);
    }
    ctx->inst_cache.resize((size_t)n_instances);
    bool any_wide = false;
    for (int32_t i = 0; i < n_instances; i++) {
        inst[i] = ctx->meshes[(size_t)instances[i].mesh].proto;
        if (runs_fast(ctx, flags)) enter_loose_tree(ctx->meshes[(size_t)instances[i].mesh], inst[i]);
        InstanceCache& c = ctx->inst_cache[(size_t)i];
        const bool same = c.valid && memcmp(&c.transform, &instances[i].transform, sizeof(rt_transform)) == 0 &&
                          memcmp(c.root_min, inst[i].root_min, 12) == 0 && memcmp(c.root_max, inst[i].root_max, 12) == 0;
        if (same) {
            copy_placement(c.placed, inst[i]);
        } else {
            fill_placement(instances[i].transform, A, inst[i]);
            c.valid = true; c.transform = instances[i].transform; c.placed = inst[i];
            memcpy(c.root_min, inst[i].root_min, 12); memcpy(c.root_max, inst[i].root_max, 12);
            c.box = instance_box(instances[i].transform, inst[i].root_min, inst[i].root_max);
        }
        if (!inst[i].empty) {
            boxes.push_back(c.box);
            if (inst[i].wide) { reach_camera(boxes.back(), A.cam_pos); any_wide = true; }
            ids.push_back((uint32_t)i);
        }
    }
    ctx->box_allowance = settled_allowance(ctx->box_allowance, scene_box_allowance(boxes, A.cam_pos));
    widen_boxes(boxes, ctx->box_allowance);
    Tlas T;
    if (ctx->keep_hierarchy) (void)ctx->standing.update(boxes, ids, (size_t)n_instances, T);
    else TlasBuilder(boxes, ids, T).run();
    if (T.depth + 2 > kTlasStackDepth) return fail(ctx, RT_ERR_UNSUPPORTED, "instance hierarchy deeper than its stack");
    if (T.pairs.size() > 2 * n + 2) return fail(ctx, RT_ERR_DEVICE, "instance hierarchy larger than its staging area");
    write_direct(T, reinterpret_cast<PairNode*>(h + off_pairs));
    RT_HIP(ctx, hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, s));

    InstancedArgs I;
    I.inst = reinterpret_cast<const MeshArgs*>(d);
    I.n_inst = (int32_t)ids.size();
    I.n_total = n_instances;
    I.tlas_root = direct_ref(T, T.root_ref);
    I.tlas = reinterpret_cast<const PairNode*>(d + off_pairs);
    memcpy(I.root_min, T.root_min, 12); memcpy(I.root_max, T.root_max, 12);
    OutputPlanes O;
    O.rgb = (uint8_t*)d_rgb; O.hit_id = (int32_t*)d_hit_instance; O.hit_t = (float*)d_hit_t; O.shadow = (uint8_t*)d_shadow;
    O.counters = ctx->d_counters; O.tile_clock = nullptr;
    if (flags & (RT_FLAG_COUNT_WORK | RT_FLAG_VERIFY_SKIPS)) {
        RT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(DeviceCounters), s));
        ctx->counters_pending = true;
    }
    if (ctx->timing) RT_HIP(ctx, hipEventRecord(ctx->ev_begin, s));
    // The per-wavefront scheduler culls by distance inside every instance; a list that places a mesh which must be walked without
    // that cull (MeshArgs::wide) is rendered by the default kernel whatever the option says.
    const int wave_tiles = any_wide ? 0 : ctx->instanced_wave;
    hipError_t le = runs_fast(ctx, flags) ? rt_fast_launch_instanced(&A, &I, &S, &O, (int32_t*)d_hit_tri, T.depth + 2, ctx->cfg.block, wave_tiles, s)
                                          : launch_instanced(A, I, S, O, (int32_t*)d_hit_tri, T.depth + 2, ctx->cfg.block, wave_tiles, s);
    if (le != hipSuccess) return hip_fail(ctx, le, "kernel launch");
    if (ctx->timing) { RT_HIP(ctx, hipEventRecord(ctx->ev_end, s)); ctx->ev_pending = true; }
    RT_HIP(ctx, hipEventRecord(ctx->inst_done[slot], s));
    ctx->inst_used[slot] = true;
    return RT_OK;
}

// ---- the frame loop over several GPUs of one process -------------------------------------------------------------
int rt_render_frames_sharded(rt_ctx* const* ctxs, int32_t n_ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms,
                             int32_t n_frames, int32_t n_transforms, int32_t width, int32_t height, void* d_rgb, size_t frame_stride) {
    if (!ctxs || n_ctx < 1 || n_ctx > 64) return RT_ERR_INVALID;
    for (int32_t r = 0; r < n_ctx; r++) if (!ctxs[r]) return RT_ERR_INVALID;
    rt_ctx* root = ctxs[0];
    if (!camera || !light || !d_rgb) return fail(root, RT_ERR_INVALID, "camera, light and d_rgb are required");
    if (n_frames < 0) return fail(root, RT_ERR_INVALID, "negative frame count");
    int callers_device = -1;                                         // put back before returning
    if (hipGetDevice(&callers_device) != hipSuccess) { (void)hipGetLastError(); callers_device = -1; }
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(root, RT_ERR_INVALID, "bad frame size");
    const size_t image = (size_t)width * (size_t)height * 3;
    if (frame_stride < image) return fail(root, RT_ERR_INVALID, "frame_stride smaller than one image");
    if (n_frames == 0) return RT_OK;
    for (int32_t r = 0; r < n_ctx; r++) {
        if ((int32_t)ctxs[r]->meshes.size() != n_transforms) return fail(root, RT_ERR_INVALID, "every context needs the scene's meshes uploaded (one transform per mesh)");
        for (int32_t q = 0; q < r; q++) if (ctxs[q] == ctxs[r]) return fail(root, RT_ERR_INVALID, "the same context listed twice");
    }
    if (n_transforms > 0 && !transforms && n_frames > 0) return fail(root, RT_ERR_INVALID, "transforms missing");
    uint8_t* out = (uint8_t*)d_rgb;
    std::vector<rt_transform> tf;
    // direct copies between the GPUs where the hardware allows them (otherwise the runtime stages the copies)
    for (int32_t r = 1; r < n_ctx; r++) {
        if (ctxs[r]->device == root->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctxs[r]->device, root->device) == hipSuccess && can && hipSetDevice(ctxs[r]->device) == hipSuccess)
            (void)hipDeviceEnablePeerAccess(root->device, 0);
        (void)hipGetLastError();                                    // "already enabled" is not an error here
    }
    // Frame f belongs to context f mod n_ctx.  Round after round every context gets its next batch: the kernel on its
    // own stream, then (contexts other than the first) one copy per frame from its staging slot to the frame's place on
    // the gathering GPU, queued behind the kernel on the same stream, so that the copies of one batch run over the
    // GPU's link to the gathering GPU while the next batch renders.  The first context renders in place.
    const int32_t share_max = (n_frames + n_ctx - 1) / n_ctx;
    int status = RT_OK;
    for (int32_t k0 = 0, round = 0; k0 < share_max && status == RT_OK; k0 += RT_MAX_BATCH, round++) {
        for (int32_t r = 0; r < n_ctx && status == RT_OK; r++) {
            rt_ctx* c = ctxs[r];
            const int32_t first = r + k0 * n_ctx;                    // first frame of this batch
            if (first >= n_frames) continue;
            int32_t nb = (n_frames - first + n_ctx - 1) / n_ctx;     // frames first, first + n_ctx, ... below n_frames
            if (nb > RT_MAX_BATCH) nb = RT_MAX_BATCH;
            tf.resize((size_t)nb * (size_t)(n_transforms > 0 ? n_transforms : 0));
            for (int32_t j = 0; j < nb; j++)
                for (int32_t m = 0; m < n_transforms; m++)
                    tf[(size_t)j * n_transforms + m] = transforms[(size_t)(first + j * n_ctx) * n_transforms + m];
            if (r == 0) {
                status = rt_render_frames(c, camera, light, tf.data(), nb, n_transforms, width, height, out + (size_t)first * frame_stride,
                                          frame_stride * (size_t)n_ctx, nullptr);
                continue;
            }
            const int slot = round & 1;
            hipError_t e = hipSetDevice(c->device);
            if (e == hipSuccess && c->shard_used[slot]) { e = hipEventSynchronize(c->shard_done[slot]); c->shard_used[slot] = false; }
            if (e == hipSuccess && !c->shard_done[slot]) e = hipEventCreateWithFlags(&c->shard_done[slot], hipEventDisableTiming);
            const size_t need = image * (size_t)nb;
            if (e == hipSuccess && need > c->shard_bytes[slot]) {
                if (c->shard_buf[slot]) (void)hipFree(c->shard_buf[slot]);
                c->shard_buf[slot] = nullptr; c->shard_bytes[slot] = 0;
                e = hipMalloc((void**)&c->shard_buf[slot], image * (size_t)RT_MAX_BATCH);
                if (e == hipSuccess) c->shard_bytes[slot] = image * (size_t)RT_MAX_BATCH;
            }
            if (e != hipSuccess) { status = hip_fail(root, e, "staging buffer of a sharded frame loop"); break; }
            status = rt_render_frames(c, camera, light, tf.data(), nb, n_transforms, width, height, c->shard_buf[slot], image, nullptr);
            if (status != RT_OK) { root->err = c->err; break; }
            for (int32_t j = 0; j < nb && e == hipSuccess; j++)
                e = hipMemcpyPeerAsync(out + (size_t)(first + j * n_ctx) * frame_stride, root->device, c->shard_buf[slot] + (size_t)j * image, c->device,
                                       image, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->shard_done[slot], c->stream);
            if (e != hipSuccess) { status = hip_fail(root, e, "copy of a frame to the gathering GPU"); break; }
            c->shard_used[slot] = true;
        }
    }
    // the frames are in d_rgb when the call returns (also after a failure nothing is left in flight)
    for (int32_t r = 0; r < n_ctx; r++) {
        hipError_t e = hipSetDevice(ctxs[r]->device);
        if (e == hipSuccess) e = hipStreamSynchronize(ctxs[r]->stream);
        ctxs[r]->shard_used[0] = ctxs[r]->shard_used[1] = false;
        if (e != hipSuccess && status == RT_OK) status = hip_fail(root, e, "waiting for a GPU of the sharded frame loop");
    }
    (void)hipSetDevice(callers_device >= 0 ? callers_device : root->device);
    return status;
}

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int rt_render_frame_host(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_transform* transforms,
                         int32_t n_transforms, int32_t width, int32_t height, uint32_t flags, uint8_t* h_rgb,
                         int32_t* h_hit_id, float* h_hit_t, uint8_t* h_shadow) {
    if (!ctx) return RT_ERR_INVALID;
    if (!h_rgb) return fail(ctx, RT_ERR_INVALID, "h_rgb is required");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pixels = (size_t)width * (size_t)height;
    if (ctx->frames_in_flight > 1 && !h_hit_id && !h_hit_t && !h_shadow && (flags & (RT_FLAG_COUNT_WORK | RT_FLAG_VERIFY_SKIPS)) == 0)
        return render_frame_in_flight(ctx, camera, light, transforms, n_transforms, width, height, flags, h_rgb);
    int st = ensure_planes(ctx, pixels);
    if (st != RT_OK) return st;
    st = rt_render_frame(ctx, camera, light, transforms, n_transforms, width, height, flags, ctx->d_rgb,
                         h_hit_id ? ctx->d_id : nullptr, h_hit_t ? ctx->d_t : nullptr, h_shadow ? ctx->d_shadow : nullptr, nullptr);
    if (st != RT_OK) return st;
    // copies queued behind the kernel, one wait: with page-locked destinations (rt_pinned_alloc) they run at bus speed
    RT_HIP(ctx, hipMemcpyAsync(h_rgb, ctx->d_rgb, pixels * 3, hipMemcpyDeviceToHost, ctx->stream));
    if (h_hit_id) RT_HIP(ctx, hipMemcpyAsync(h_hit_id, ctx->d_id, pixels * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (h_hit_t) RT_HIP(ctx, hipMemcpyAsync(h_hit_t, ctx->d_t, pixels * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (h_shadow) RT_HIP(ctx, hipMemcpyAsync(h_shadow, ctx->d_shadow, pixels, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_pinned_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return RT_ERR_NO_DEVICE; }
    if (hipHostMalloc(out, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return RT_ERR_DEVICE; }
    return RT_OK;
}

int rt_pinned_free(void* p) {
    if (!p) return RT_OK;
    if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return RT_ERR_DEVICE; }
    return RT_OK;
}

int rt_render_frame_instanced_host(rt_ctx* ctx, const rt_camera* camera, const rt_light* light, const rt_instance* instances,
                                   int32_t n_instances, int32_t width, int32_t height, uint32_t flags, uint8_t* h_rgb) {
    if (!ctx) return RT_ERR_INVALID;
    if (!h_rgb) return fail(ctx, RT_ERR_INVALID, "h_rgb is required");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t pixels = (size_t)width * (size_t)height;
    int st = ensure_planes(ctx, pixels);
    if (st != RT_OK) return st;
    st = rt_render_frame_instanced(ctx, camera, light, instances, n_instances, width, height, flags, ctx->d_rgb, nullptr, nullptr, nullptr,
                                   nullptr, nullptr);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpyAsync(h_rgb, ctx->d_rgb, pixels * 3, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_scaled_size(int32_t width, int32_t height, float scale_factor, int32_t* scaled_width, int32_t* scaled_height) {
    if (!scaled_width || !scaled_height || width <= 0 || height <= 0 || !(scale_factor > 0.0f)) return RT_ERR_INVALID;
    *scaled_width = (int32_t)((float)width / scale_factor + 0.5f);
    *scaled_height = (int32_t)((float)height / scale_factor + 0.5f);
    return RT_OK;
}

int rt_upscale_frames(rt_ctx* ctx, const void* d_rgb, size_t frame_stride, int32_t n_frames, int32_t width, int32_t height,
                      int32_t scaled_width, int32_t scaled_height, void* d_argb, size_t out_stride, void* stream) {
    if (!ctx) return RT_ERR_INVALID;
    if (!d_rgb || !d_argb) return fail(ctx, RT_ERR_INVALID, "d_rgb and d_argb are required");
    if (n_frames < 1 || n_frames > 65535) return fail(ctx, RT_ERR_INVALID, "n_frames must be 1..65535");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    if (scaled_width < 2 || scaled_height < 2) return fail(ctx, RT_ERR_UNSUPPORTED, "scaled size must be at least 2x2");
    if ((int64_t)scaled_width * scaled_height > (int64_t)1 << 30 || scaled_height > 65535 * 8) return fail(ctx, RT_ERR_UNSUPPORTED, "scaled size too large");
    if (frame_stride < (size_t)width * (size_t)height * 3) return fail(ctx, RT_ERR_INVALID, "frame_stride smaller than one image");
    if (out_stride < (size_t)scaled_width * (size_t)scaled_height) return fail(ctx, RT_ERR_INVALID, "out_stride smaller than one image");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (ctx->timing) RT_HIP(ctx, hipEventRecord(ctx->ev_begin, s));
    hipError_t le = launch_upscale((const uint8_t*)d_rgb, frame_stride, n_frames, width, height, scaled_width, scaled_height,
                                   (uint32_t*)d_argb, out_stride, s);
    if (le != hipSuccess) return hip_fail(ctx, le, "kernel launch");
    if (ctx->timing) { RT_HIP(ctx, hipEventRecord(ctx->ev_end, s)); ctx->ev_pending = true; }
    return RT_OK;
}

int rt_upscale_frame_host(rt_ctx* ctx, const uint8_t* h_rgb, int32_t width, int32_t height, int32_t scaled_width,
                          int32_t scaled_height, uint32_t* h_argb) {
    if (!ctx) return RT_ERR_INVALID;
    if (!h_rgb || !h_argb) return fail(ctx, RT_ERR_INVALID, "h_rgb and h_argb are required");
    if (width <= 0 || height <= 0 || (int64_t)width * height > (int64_t)1 << 30) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    if (scaled_width < 2 || scaled_height < 2) return fail(ctx, RT_ERR_UNSUPPORTED, "scaled size must be at least 2x2");
    if ((int64_t)scaled_width * scaled_height > (int64_t)1 << 30) return fail(ctx, RT_ERR_UNSUPPORTED, "scaled size too large");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)width * (size_t)height * 3, out_pixels = (size_t)scaled_width * (size_t)scaled_height;
    if (in_bytes > ctx->up_in_bytes || out_pixels > ctx->up_out_pixels) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_up_in) (void)hipFree(ctx->d_up_in);
        if (ctx->d_up_out) (void)hipFree(ctx->d_up_out);
        ctx->d_up_in = nullptr; ctx->d_up_out = nullptr; ctx->up_in_bytes = 0; ctx->up_out_pixels = 0;
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_up_in, in_bytes));
        RT_HIP(ctx, hipMalloc((void**)&ctx->d_up_out, out_pixels * sizeof(uint32_t)));
        ctx->up_in_bytes = in_bytes; ctx->up_out_pixels = out_pixels;
    }
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_up_in, h_rgb, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    int st = rt_upscale_frames(ctx, ctx->d_up_in, in_bytes, 1, width, height, scaled_width, scaled_height, ctx->d_up_out, out_pixels, nullptr);
    if (st != RT_OK) return st;
    RT_HIP(ctx, hipMemcpyAsync(h_argb, ctx->d_up_out, out_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_sync(rt_ctx* ctx, void* stream) {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    if (!stream)                                      // and the frames rt_render_frame_host left in flight
        for (int i = 0; i < kRingSlots; i++)
            if (ctx->ring_used[i]) { RT_HIP(ctx, hipEventSynchronize(ctx->ring_done[i])); ctx->ring_used[i] = false; }
    return RT_OK;
}

int rt_last_kernel_ms(rt_ctx* ctx, float* ms) {
    if (!ctx || !ms) return RT_ERR_INVALID;
    if (!ctx->ev_pending) return fail(ctx, RT_ERR_INVALID, "no timed launch yet (option \"timing\" off?)");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipEventSynchronize(ctx->ev_end));
    RT_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_begin, ctx->ev_end));
    return RT_OK;
}

int rt_last_counters(rt_ctx* ctx, rt_counters* out) {
    if (!ctx || !out) return RT_ERR_INVALID;
    if (!ctx->counters_pending) return fail(ctx, RT_ERR_INVALID, "no RT_FLAG_COUNT_WORK launch yet");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipDeviceSynchronize());
    DeviceCounters c;
    RT_HIP(ctx, hipMemcpy(&c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    out->primary_rays = c.primary_rays; out->shadow_rays = c.shadow_rays;
    out->slab_tests = c.slab_tests; out->tri_tests = c.tri_tests;
    out->hit_pixels = c.hit_pixels; out->shadowed_pixels = c.shadowed_pixels;
    out->skip_checks = c.checks; out->idle_shadow_rays = c.idle_shadow;
    out->vis_disagreements = c.bad_vis; out->shadow_vis_disagreements = c.bad_shadow_vis;
    out->entry_disagreements = c.bad_entry; out->cull_disagreements = c.bad_cull; out->instance_disagreements = c.bad_inst;
    return RT_OK;
}

int rt_set_trace_buffer(rt_ctx* ctx, void* d_tile_clock) {
    if (!ctx) return RT_ERR_INVALID;
    ctx->d_tile_clock = d_tile_clock;
    return RT_OK;
}

int rt_set_option(rt_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return RT_ERR_INVALID;
    if (!strcmp(name, "variant")) {
        if (value < 0 || value >= kNumVariants) return fail(ctx, RT_ERR_INVALID, "unknown kernel variant");
        ctx->cfg.variant = value; return RT_OK;
    }
    if (!strcmp(name, "block")) {
        if (value != 64 && value != 128 && value != 256) return fail(ctx, RT_ERR_INVALID, "block must be 64, 128 or 256");
        ctx->cfg.block = value; return RT_OK;
    }
    if (!strcmp(name, "timing")) { ctx->timing = value ? 1 : 0; return RT_OK; }
    if (!strcmp(name, "coop")) { ctx->coop = value; return RT_OK; }
    if (!strcmp(name, "xcd_swizzle")) { ctx->xcd_swizzle = value ? 1 : 0; return RT_OK; }
    if (!strcmp(name, "sampled_split")) { ctx->sampled_split = value < 0 ? 0 : (value > 2 ? 2 : value); return RT_OK; }
    if (!strcmp(name, "device_build_min")) { ctx->device_build_min = value < 1 ? 1 : value; return RT_OK; }
    if (!strcmp(name, "regroup")) { ctx->regroup = value ? 1 : 0; return RT_OK; }
    if (!strcmp(name, "keep_hierarchy")) { ctx->keep_hierarchy = value ? 1 : 0; ctx->standing.clear(); return RT_OK; }
    if (!strcmp(name, "fast")) { ctx->fast = value ? 1 : 0; return RT_OK; }
    if (!strcmp(name, "screen_cull")) { ctx->screen_cull = value ? 1 : 0; return RT_OK; }
    if (!strcmp(name, "instanced_wave")) { ctx->instanced_wave = value < 0 ? 0 : (value > 8 ? 8 : value); return RT_OK; }
    if (!strcmp(name, "frames_in_flight")) {
        if (value < 1 || value > kRingSlots) return fail(ctx, RT_ERR_INVALID, "frames_in_flight must be 1..8");
        const int st = rt_sync(ctx, nullptr);            // the ring is empty whenever its length changes
        if (st != RT_OK) return st;
        ctx->frames_in_flight = value; ctx->ring_seq = 0; return RT_OK;
    }
    if (!strcmp(name, "batch_rows")) { ctx->batch_rows = value ? 1 : 0; return RT_OK; }
    if (!strcmp(name, "batch_chunk")) {
        if (value < 0 || value > (1 << 20)) return fail(ctx, RT_ERR_INVALID, "batch_chunk out of range");
        ctx->batch_chunk = value; return RT_OK;
    }
    if (!strcmp(name, "resident_budget")) {
        if (value < 0 || value > (int)kResidentBudget) return fail(ctx, RT_ERR_INVALID, "resident_budget out of range");
        ctx->resident_budget = value; ctx->scene_dirty = true; return RT_OK;
    }
    if (!strcmp(name, "budget")) {
        if (value < 1) return fail(ctx, RT_ERR_INVALID, "budget must be positive");
        ctx->budget = value; return RT_OK;
    }
    return fail(ctx, RT_ERR_INVALID, std::string("unknown option ") + name);
}

int rt_get_option(const rt_ctx* ctx, const char* name, int* value) {
    if (!ctx || !name || !value) return RT_ERR_INVALID;
    if (!strcmp(name, "variant")) { *value = ctx->cfg.variant; return RT_OK; }
    if (!strcmp(name, "block")) { *value = ctx->cfg.block; return RT_OK; }
    if (!strcmp(name, "timing")) { *value = ctx->timing; return RT_OK; }
    if (!strcmp(name, "sampled_split")) { *value = ctx->sampled_split; return RT_OK; }
    if (!strcmp(name, "device_build_min")) { *value = ctx->device_build_min; return RT_OK; }
    if (!strcmp(name, "regroup")) { *value = ctx->regroup; return RT_OK; }
    if (!strcmp(name, "frames_in_flight")) { *value = ctx->frames_in_flight; return RT_OK; }
    if (!strcmp(name, "keep_hierarchy")) { *value = ctx->keep_hierarchy; return RT_OK; }
    if (!strcmp(name, "instanced_wave")) { *value = ctx->instanced_wave; return RT_OK; }
    if (!strcmp(name, "fast")) { *value = ctx->fast; return RT_OK; }
    if (!strcmp(name, "screen_cull")) { *value = ctx->screen_cull; return RT_OK; }
    if (!strcmp(name, "coop")) { *value = ctx->coop; return RT_OK; }
    if (!strcmp(name, "batch_rows")) { *value = ctx->batch_rows; return RT_OK; }
    if (!strcmp(name, "budget")) { *value = ctx->budget; return RT_OK; }
    if (!strcmp(name, "n_variants")) { *value = kNumVariants; return RT_OK; }
    if (!strcmp(name, "experimental_variants")) { *value = kExperimentalVariants; return RT_OK; }
    if (!strcmp(name, "default_variant")) { *value = kDefaultVariant; return RT_OK; }
    return RT_ERR_INVALID;
}

}  // extern "C"
