// rt_layout.h -- data layouts shared by the host packer and the HIP kernels.
//
// HBM layout of a mesh (all arrays 64-byte aligned, one allocation each per mesh):
//   PairNode[]  one 64-byte record per INTERNAL node of the reference BVH: the boxes of both
//               children plus a reference to each.  Visiting an internal node therefore costs one
//               aligned 64-byte fetch and yields both child slab tests; the node's own box was
//               tested when its parent was visited (the root box sits in MeshArgs).
//   TriRec[]    48 bytes per triangle, post-build order: v0, e1 = v1-v0, e2 = v2-v0 (the two
//               subtractions Moeller-Trumbore starts with, done once on the host, same rounding)
//               and the triangle's tie priority.
//   AttrRec[]   64 bytes per triangle: texture coordinates and vertex normals, read once per hit.
//   texture     RGBA8 rows exactly as decoded.
#pragma once
#include <stdint.h>

namespace rt {

constexpr int kMaxMeshes = 8;
constexpr int kStackDepth = 32;            // entries per lane; covers trees up to depth 30

// Child / stack reference.  bit31 set: leaf, bits 30..24 = triangle count (1..127),
// bits 23..0 = first triangle.  bit31 clear: index of a PairNode.
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr int kMaxLeafTris = 127;
inline constexpr uint32_t leaf_ref(uint32_t start, uint32_t count) { return kLeafBit | (count << 24) | start; }

// Lanes that share one ray in the cooperative traversal (rt_kernels.hip).
constexpr int kGroupLanes = 8;
// A mesh with at least this many PairNodes counts as "big": single rays can visit hundreds of its
// nodes, so the screen area it covers (and shadows) is rendered cooperatively.
constexpr uint32_t kBigMeshPairs = 64;
constexpr int kMaxCoopRects = 4;

struct alignas(64) PairNode {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    uint32_t lref, rref;
    uint32_t pad[2];
};
static_assert(sizeof(PairNode) == 64, "PairNode must be one 64-byte record");

struct alignas(16) TriRec {
    float v0[3];
    uint32_t prio;        // winner among equal-t hits = highest prio (see host packer)
    float e1[3];
    float pad1;           // bits: the triangle's index in the mesh's own triangle order (what a hit reports); read by the fast build,
                          // whose tree keeps its records in another order -- the exact build takes the record's position
    float e2[3];
    float pad2;
};
static_assert(sizeof(TriRec) == 48, "TriRec is three float4");

struct alignas(64) AttrRec {
    float t0[2], t1[2], t2[2];
    float n0[3], n1[3], n2[3];
    float pad;
};
static_assert(sizeof(AttrRec) == 64, "AttrRec must be one 64-byte record");

// Per-mesh kernel arguments: static geometry handles + the per-frame constants that the
// reference re-derives for every ray (ray.c:5-16, 27-37), computed once per frame on the host
// with the same float operations.
struct MeshArgs {
    uint32_t pair_base;          // first PairNode of this mesh in the scene's combined node array
    uint32_t pair_count;
    const TriRec* tris;
    const AttrRec* attrs;
    const uint32_t* texels;      // RGBA8 packed little-endian: R in bits 0..7
    float root_min[3], root_max[3];
    uint32_t root_ref;           // leaf_ref(...) or PairNode index 0
    int32_t tex_w, tex_h;
    uint32_t empty;              // 1: mesh has no triangles (root box never leads to a hit)
    uint32_t wide;               // 1: the mesh holds a triangle large enough for ray.c:50's cut-off |a| < 1e-7 to admit rays that run almost
                                 // inside its plane, where ray.c:63's distance can come out far in front of the surface: nothing
                                 // of such a mesh is skipped for being farther away than the closest hit (Packer: kWideTriangle)
    float inv[3][4];             // rows 0..2 of inverse(T*Rz*Ry*Rx)
    float wrow[4];               // row 3 of the same matrix (mat4_transform_point divides by it)
    float nrm[3][3];             // upper 3x3 of transpose(inverse(Rz*Ry*Rx))
    float cam_o[3];              // camera position in mesh space (mat4_transform_point, hoisted)
    float lit_d[3];              // normalised light direction in mesh space
    float lit_inv[3];            // 1.0f / lit_d
    uint32_t w_mode;             // 0: general w; 1: w == wrow[3] for every finite point; 2: w == 1.0f
    // Byte offsets of this mesh's PairNodes / TriRecs inside the scene's LDS-resident blob
    // (SceneArgs::resident), or kNotResident when the mesh is only in HBM.
    uint32_t res_pairs, res_tris;
    // Where the reference's own tree starts when root_ref enters a regrouped one (rt_pack.h, Packer::regroup);
    // equal to root_ref otherwise.  Same index space as root_ref (relative to pair_base).
    uint32_t ref_root;
    // grid_axis = 4 + k: a further tree over the reference's leaves, in the same node array, whose boxes live in a sheared frame
    // (axis k kept; of the other two, a = (k + 1) % 3 and b = (k + 2) % 3, the frame's coordinates are p_a + p_b and p_a - p_b),
    // where beams that run along the diagonals of a and b get tight boxes.  Its leaves are records like the flat grid's lists
    // (kGridListBit); `grid` then holds a header, the directions a ray must keep away from to enter this tree, and the records
    // (ShearHeader, Packer::build_shear).
    uint32_t shear_root;
    // Flat meshes (root box of zero thickness along grid_axis, e.g. a ground plane): a grid over the other two axes whose
    // cells name the node of the tree behind root_ref at which a ray that meets the mesh's plane inside the cell may start
    // its walk -- everything above that node has, for such a ray, exactly one child whose box it can pass (FlatGrid in
    // rt_pack.h).  grid_axis = 3: no grid.
    const uint32_t* grid;
    uint32_t grid_axis;
    int32_t grid_n;              // cells per side
    float grid_lo[2], grid_inv[2];      // per in-plane axis ((axis+1)%3, (axis+2)%3): origin, cells per unit length
    float grid_slack[2];         // how far a ray's plane point may be off for the cell's entry to hold
    float grid_bmax[2];          // largest |coordinate| of the mesh on that axis
};
// A grid cell of a flat mesh holds leaf_ref(0, 0) (nothing to test), a PairNode index (the walk starts there), or, with this
// bit, a list of leaves: bits 27..24 = count (1..kGridListMax), bits 23..0 = first record; the records (eight words each: leaf
// box min, max, leaf reference, mask of the leaf's triangles to test) follow the grid_n * grid_n cell words, rounded up to a
// multiple of eight words.
static_assert(sizeof(MeshArgs) == 280, "MeshArgs: the staging arrays of per-frame constants are sized by it");
constexpr uint32_t kGridListBit = 0x40000000u;
// MeshArgs::grid of a mesh with a sheared tree, in words: this header (16 words); MeshArgs::grid_n families of the mesh's
// triangles, eight words each (unit normal, how large |d . n| must at least be for the tree's boxes to hold for the family's
// triangles, first and number of the family's members, two unused); the members, two words each (a leaf record's number and the
// mask of the family's triangles in that leaf); from lists_at on, one record per leaf of the reference's tree (as the flat
// grid's list records: box min, max, leaf, mask).
struct ShearHeader {
    float reach;             // a ray enters the sheared tree only if |o| <= reach ...
    float origin_factor;     // ... and o lies outside [hull_min, hull_max] widened by origin_factor * (|o| + radius)
    float radius;            // largest |vertex|
    float hull_min[3], hull_max[3];
    uint32_t lists_at;       // first word of the leaf records (a multiple of eight)
    uint32_t members_at;     // first word of the members
    uint32_t near_root;      // a second tree over the same records for rays that start inside the widened hull: every box wider by how
                             // far such an origin can lie behind a plane and still have its triangle accepted
    float near_reach;        // ... for origins with |o| <= near_reach
    uint32_t pad[3];
};
static_assert(sizeof(ShearHeader) == 64, "ShearHeader is sixteen words");
constexpr uint32_t kGridListMax = 8;
constexpr uint32_t kNotResident = 0xFFFFFFFFu;
// Small meshes (nodes + triangles) are copied into LDS by every workgroup of the lean kernels, the
// smallest first, as long as they fit in this many bytes.
constexpr uint32_t kResidentBudget = 32 * 1024;

struct FrameArgs {
    // height = rows this launch renders (and the rows of its output planes).  A launch may render every
    // strip_stride-th strip of eight rows of a frame full_height rows tall, starting with strip
    // strip_offset (one rank's share of a frame split over several GPUs): row py of the launch is row
    // frame_row(py) of the frame.  A whole frame is full_height = height, stride 1, offset 0.
    int32_t width, height;
    int32_t full_height, strip_stride, strip_offset;
    int32_t first_tile_row;      // batched frames: tile row this frame's waves start scanning at (its heavy tiles first), 0 otherwise
    int32_t n_mesh;
    uint32_t flags;
    float cam_pos[3];
    float cam_right[3], cam_up[3], cam_fwd[3];
    float aspect, scale;
    float light_dir[3], light_color[3];
    // Tiles (8x8 pixels) inside any of these rectangles {x0, y0, x1, y1} (tile units, half open) are
    // rendered cooperatively: eight waves per tile, eight pixels per wave, every pixel's rays
    // traversed by kGroupLanes lanes together; those waves are launched first.  The rectangles
    // cover the screen area where a ray, primary or shadow, can meet a big mesh.
    int32_t n_rects;
    int32_t rect[kMaxCoopRects][4];
    // Two-pass variant: node visits one lane may spend on a pixel before handing it to the
    // cooperative pass.
    int32_t defer_budget;
    // Per mesh, the 8x8-pixel tiles of the FRAME {x0, y0, x1, y1} (half open) outside of which no camera ray can pass the
    // mesh's root box, not even with the rounding of the float test (see mesh_screen_rect, rt_pack.h): there the
    // reference's root test (bvh.c:82) fails for certain, and the mesh is skipped without transforming the ray.
    int32_t vis[kMaxMeshes][4];
    // The same for shadow rays: tiles outside of which no surface point of the scene seen in the tile can send a shadow ray
    // (towards light_dir) through the mesh's root box (mesh_shadow_rect).
    int32_t vis_shadow[kMaxMeshes][4];
    MeshArgs mesh[kMaxMeshes];
};

// Sampled rendering (several jittered camera rays per pixel, several directional lights): what it
// needs beyond FrameArgs.  Light 0 of a sampled frame is also the light in FrameArgs.
constexpr int kMaxLights = 4;
constexpr int kMaxSamples = 64;
struct SampleArgs {
    int32_t spp;                 // camera rays per pixel; 1 = the pixel centre
    uint32_t seed;
    int32_t n_lights;
    int32_t pad;
    float light_dir[kMaxLights][3], light_color[kMaxLights][3];
    float lit_d[kMaxMeshes][kMaxLights][3];      // light direction in mesh space, normalised
    float lit_inv[kMaxMeshes][kMaxLights][3];    // its reciprocal
};

// Instanced scenes: any number of placed meshes (the reference's Scene.meshes may hold the same
// triangles under many transforms, scene.c:31-35).  Every instance has a full MeshArgs in a device
// array, in the scene's mesh order; a two-level hierarchy over the instances' world-space boxes
// (same PairNode records; a leaf is leaf_ref(instance index, 1), or leaf_ref(0, 0) for none) only decides which instances a ray needs
// to be tested against -- the test itself is the reference's, in the instance's own space.
constexpr int kTlasStackDepth = 32;
struct InstancedArgs {
    const MeshArgs* inst;        // one record per entry of the scene's mesh list (n_total), list order
    int32_t n_inst;              // entries that have triangles (the hierarchy's leaves)
    uint32_t tlas_root;          // leaf ref or index of the root PairNode in `tlas`
    const PairNode* tlas;
    float root_min[3], root_max[3];
    int32_t n_total;             // entries of the list
};

struct DeviceCounters {          // accumulated with atomics when counting is requested
    unsigned long long primary_rays, shadow_rays, slab_tests, tri_tests, hit_pixels, shadowed_pixels;
    // self-checking launches (RT_FLAG_VERIFY_SKIPS): see Work in rt_pixel.h
    unsigned long long checks, bad_vis, bad_shadow_vis, bad_entry, bad_cull, bad_inst;
    unsigned long long idle_shadow;  // shadow rays (part of shadow_rays) whose result cannot change the colour: see Work::idle
};

// Scene-wide device data (valid for every frame until the meshes change).
struct SceneArgs {
    const PairNode* pairs;       // PairNodes of all meshes, mesh after mesh
    uint32_t n_pairs;
    uint32_t stack_entries;      // traversal stack entries per lane that the deepest mesh needs
    const uint8_t* resident;     // blob of the LDS-resident meshes: per mesh its PairNodes, then its TriRecs
    uint32_t resident_bytes;     // multiple of 64
};

// Work distribution state of the persistent kernels: tiles are handed out by an atomic counter;
// the last wave to finish puts both words back to zero for the next launch.
struct TileQueue {
    unsigned int next;
    unsigned int done;
};

// Two-pass rendering: pixels whose rays enter a big mesh are put on a list by the first pass and
// rendered cooperatively by the second.  Frames alternate between the two counters; the second
// pass of a frame clears the counter the next frame will use.
struct DeferQueue {
    unsigned int count[2];
};

struct OutputPlanes {
    uint8_t* rgb;
    int32_t* hit_id;
    float* hit_t;
    uint8_t* shadow;
    DeviceCounters* counters;
    unsigned long long* tile_clock;   // optional trace: per 8x8 tile {start, end} of the constant 100 MHz clock
};

}  // namespace rt
