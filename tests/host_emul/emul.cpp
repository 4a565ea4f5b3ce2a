// emul.cpp -- TEST-ONLY host build of the product's pixel pipeline (raytracer_amd/csrc/rt_pixel.h)
// over the product's packed scene layout (rt_pack.h).  It lets the CPU test-suite check the
// packer, the stack traversal and the tie rules against the oracle without a GPU.  It is not
// part of librt_mi355.so and nothing in the product loads it.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../raytracer_amd/csrc/rt_bicubic.h"
#include "../../raytracer_amd/csrc/rt_pack.h"
#include "../../raytracer_amd/csrc/rt_pixel.h"
#include "../../raytracer_amd/csrc/rt_instanced.h"
#include "../../raytracer_amd/csrc/rt_sampled.h"

using namespace rt;

namespace {
struct ArrayStack {
    uint32_t data[kStackDepth];
    int sp = 0;
    int high = 0;
    void clear() { sp = 0; }
    bool empty() const { return sp == 0; }
    void push(uint32_t v) { data[sp++] = v; if (sp > high) high = sp; }
    void push_if(bool c, uint32_t v) { if (c) push(v); }
    uint32_t pop() { return data[--sp]; }
};

struct EmulMesh {
    PackedMesh pk;
    std::vector<uint32_t> texels;
    int tw = 0, th = 0;
};
struct EmulScene {
    std::vector<EmulMesh> meshes;
    std::vector<PairNode> all_pairs;
    std::string err;
    int stack_high = 0;
    bool regroup = false;
    bool keep_hierarchy = false;       // instanced frames: the product's hierarchy kept from call to call
    float box_allowance = 0.0f;        // what the world boxes were widened by last time
    uint32_t* work_plane = nullptr;    // instanced frames: per pixel {box tests, triangle tests} of both rays, when set
    bool loose = false;                // rays enter through the tree of single triangles (Packer::build_loose), as in the kernels' fast build
    int vis_shrink = 0;                // tests of the self-check: screen and shadow rectangles made this many tiles too small on every side
    StandingTlas standing;
    int hierarchy_reused = 0;          // calls that reused the kept tree
};
struct CheckTotals {
    unsigned long long checks = 0, vis = 0, shadow_vis = 0, entry = 0, cull = 0, inst = 0, idle = 0;
    void add(const Work& w) { idle += w.idle; checks += w.checks; vis += w.bad_vis; shadow_vis += w.bad_shadow_vis; entry += w.bad_entry; cull += w.bad_cull; inst += w.bad_inst; }
    void merge(const CheckTotals& o) { idle += o.idle; checks += o.checks; vis += o.vis; shadow_vis += o.shadow_vis; entry += o.entry; cull += o.cull; inst += o.inst; }
    void store(rt_counters* c) const {
        c->skip_checks = checks; c->vis_disagreements = vis; c->shadow_vis_disagreements = shadow_vis;
        c->entry_disagreements = entry; c->cull_disagreements = cull; c->instance_disagreements = inst;
        c->idle_shadow_rays = idle;
    }
};
}  // namespace

extern "C" {

void* emul_scene_create() { return new EmulScene(); }
void emul_scene_destroy(void* s) { delete (EmulScene*)s; }
const char* emul_last_error(void* s) { return ((EmulScene*)s)->err.c_str(); }
int emul_stack_high(void* s) { return ((EmulScene*)s)->stack_high; }
void emul_keep_hierarchy(void* s, int on) { ((EmulScene*)s)->keep_hierarchy = on != 0; ((EmulScene*)s)->standing.clear(); ((EmulScene*)s)->hierarchy_reused = 0; }
int emul_hierarchy_reused(void* s) { return ((EmulScene*)s)->hierarchy_reused; }
// Deliberate damage, for the tests that the self-checking party notices a wrong shortcut:
// rectangles too small; a flat mesh's grid whose every cell names the first leaf only; the second tree's root with its left box moved out of every ray's way.
void emul_set_loose(void* s, int on) { ((EmulScene*)s)->loose = on != 0; }
void emul_set_vis_shrink(void* s, int tiles) { ((EmulScene*)s)->vis_shrink = tiles; }
int emul_damage_grid(void* sp, int mesh) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return RT_ERR_INVALID;
    PackedMesh& pk = s->meshes[(size_t)mesh].pk;
    if (pk.grid_axis >= 3 || pk.grid.empty()) return RT_ERR_UNSUPPORTED;
    uint32_t leaf = 0;
    for (const PairNode& n : pk.pairs) { if (n.lref & kLeafBit) { leaf = n.lref; break; } if (n.rref & kLeafBit) { leaf = n.rref; break; } }
    if (!leaf) return RT_ERR_UNSUPPORTED;
    const size_t cells = (size_t)pk.grid_n * (size_t)pk.grid_n;
    for (size_t i = 0; i < cells && i < pk.grid.size(); i++) pk.grid[i] = leaf;
    return RT_OK;
}
// The same grid with its triangle selection broken: every listed leaf keeps only its first triangle, and every cell that
// names a single triangle names the one after it.
int emul_damage_grid_masks(void* sp, int mesh) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return RT_ERR_INVALID;
    PackedMesh& pk = s->meshes[(size_t)mesh].pk;
    if (pk.grid_axis >= 3 || pk.grid.empty()) return RT_ERR_UNSUPPORTED;
    const size_t cells = (size_t)pk.grid_n * (size_t)pk.grid_n, list_base = (cells + 7) & ~(size_t)7;
    int changed = 0;
    for (size_t i = 0; i < cells; i++)
        if ((pk.grid[i] & kLeafBit) && ((pk.grid[i] >> 24) & 0x7Fu) == 1u) { pk.grid[i] = leaf_ref(((pk.grid[i] & 0x00FFFFFFu) + 1u) % (uint32_t)pk.tris.size(), 1u); changed++; }
    for (size_t r = list_base; r + 8 <= pk.grid.size(); r += 8) { if (pk.grid[r + 7] != 1u) changed++; pk.grid[r + 7] = 1u; }
    return changed > 0 ? RT_OK : RT_ERR_UNSUPPORTED;
}
int emul_damage_second_tree(void* sp, int mesh) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return RT_ERR_INVALID;
    const PackedMesh& pk = s->meshes[(size_t)mesh].pk;
    if (pk.root_ref == pk.ref_root || (pk.root_ref & kLeafBit)) return RT_ERR_UNSUPPORTED;      // no second tree
    size_t base = 0;
    for (int m = 0; m < mesh; m++) base += s->meshes[(size_t)m].pk.pairs.size();
    PairNode& n = s->all_pairs[base + pk.root_ref];
    for (int c = 0; c < 3; c++) { n.lmin[c] = 1.0e9f; n.lmax[c] = 1.0e9f + 64.0f; }         // a box no ray of the scene meets
    return RT_OK;
}
void emul_set_regroup(void* s, int on) { ((EmulScene*)s)->regroup = on != 0; }   // for meshes added afterwards

// Packs a mesh twice -- with the second tree built by run(), and with only the reference's tree followed by
// Packer::regroup_packed (the route of meshes packed on the device) -- and returns 0 when both give the same records, root
// references and depth; 1 when they differ; -1 when the mesh cannot be packed; 2 when neither route built a second tree.
int emul_regroup_routes_agree(const rt_triangle* tris, int32_t n_tris, const rt_flat_node* nodes, int32_t n_nodes) {
    std::string err;
    PackedMesh a, b;
    if (Packer(tris, n_tris, nodes, n_nodes, a, true).run(err) != RT_OK) return -1;
    if (Packer(tris, n_tris, nodes, n_nodes, b, false).run(err) != RT_OK) return -1;
    const bool done = Packer::regroup_packed(b);
    // run() may have appended the tree over the triangles themselves (Packer::build_loose), which the other route does not build
    // ... and the tree in a sheared frame (Packer::build_shear) behind it
    const size_t a_pairs = a.pairs.size() - (a.loose_tris.empty() ? 0 : a.loose_pairs) - a.shear_pairs;
    const bool had = a.root_ref != a.ref_root || a.tree_pairs != (uint32_t)a_pairs;
    if (!done && !had) return 2;
    const bool same_depth = a.depth == b.depth || (!a.loose_tris.empty() && a.depth == a.loose_depth && a.depth > b.depth);
    if (a_pairs != b.pairs.size() || a.root_ref != b.root_ref || a.ref_root != b.ref_root || a.tree_pairs != b.tree_pairs || !same_depth) return 1;
    return memcmp(a.pairs.data(), b.pairs.data(), a_pairs * sizeof(PairNode)) == 0 ? 0 : 1;
}

int emul_mesh_add(void* sp, const rt_triangle* tris, int32_t n_tris, const rt_flat_node* nodes, int32_t n_nodes,
                  const uint8_t* rgba, int32_t tw, int32_t th) {
    EmulScene* s = (EmulScene*)sp;
    EmulMesh m;
    int st = Packer(tris, n_tris, nodes, n_nodes, m.pk, s->regroup).run(s->err);
    if (st != RT_OK) return st;
    m.texels.resize((size_t)tw * th);
    memcpy(m.texels.data(), rgba, (size_t)tw * th * 4);
    m.tw = tw; m.th = th;
    s->all_pairs.insert(s->all_pairs.end(), m.pk.pairs.begin(), m.pk.pairs.end());
    s->meshes.push_back(std::move(m));
    return RT_OK;
}

// 1 if the packer found a triangle with |e1| |e2| >= kWideTriangle (the mesh is then walked without the distance cull), else 0.
int emul_mesh_wide(void* sp, int mesh) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return -1;
    return s->meshes[(size_t)mesh].pk.wide ? 1 : 0;
}
// MeshArgs::grid_axis of the mesh: 0..2 a flat mesh's grid, 3 nothing, 4 + k a tree in the sheared frame that keeps axis k.
int emul_mesh_grid_axis(void* sp, int mesh) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return -1;
    return (int)s->meshes[(size_t)mesh].pk.grid_axis;
}
// Deliberate damage: every box of the sheared tree's root moved out of every ray's way.
int emul_damage_shear(void* sp, int mesh) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return RT_ERR_INVALID;
    const PackedMesh& pk = s->meshes[(size_t)mesh].pk;
    if (pk.grid_axis < 4 || (pk.shear_root & (kLeafBit | kGridListBit))) return RT_ERR_UNSUPPORTED;
    size_t base = 0;
    for (int m = 0; m < mesh; m++) base += s->meshes[(size_t)m].pk.pairs.size();
    PairNode& root = s->all_pairs[base + pk.shear_root];
    for (int a = 0; a < 3; a++) { root.lmin[a] = 1.0e30f; root.lmax[a] = 1.1e30f; }
    return RT_OK;
}
// The per-frame part of a sheared tree's gate against the gate itself: for every camera ray of the frame that passes the mesh's
// root box, and for the frame's shadow direction, the families the gate finds close when it looks at all of them must be among
// those it finds with the host's masks.  Returns the number of rays for which they are not (-1: the mesh has no such tree);
// *looked gets the number of rays compared, *spared the families the masks let the camera rays skip, summed over those rays.
long long emul_shear_masks_hold(void* sp, int mesh, const rt_camera* cam, const rt_light* light, const rt_transform* tf, int32_t width, int32_t height,
                                long long* looked, long long* spared) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return -1;
    const EmulMesh& E = s->meshes[(size_t)mesh];
    if (E.pk.grid_axis < 4) return -1;
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, *light, width, height, A);
    MeshArgs M; memset(&M, 0, sizeof M);
    memcpy(M.root_min, E.pk.root_min, 12); memcpy(M.root_max, E.pk.root_max, 12);
    set_grid_args(E.pk, E.pk.grid.data(), M);
    fill_placement(*tf, A, M);
    MeshArgs plain = M;
    shear_frame_masks(E.pk.grid.data(), M);
    long long bad = 0, n = 0, skipped = 0;
    const uint64_t all = M.grid_n >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << M.grid_n) - 1);
    uint32_t valid; memcpy(&valid, &M.grid_slack[0], 4);
    uint32_t w0, w1; memcpy(&w0, &M.grid_lo[0], 4); memcpy(&w1, &M.grid_lo[1], 4);
    const uint64_t check = (valid & 1u) ? ((uint64_t)w0 | ((uint64_t)w1 << 32)) : all;
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const float fx = ((float)x + 0.5f) / (float)A.width, fy = ((float)y + 0.5f) / (float)A.full_height;
            const float rx = (2.0f * fx - 1.0f) * A.aspect * A.scale, ry = (1.0f - 2.0f * fy) * A.scale;
            const F3 wd = unit(add(add(scale(f3(A.cam_right), rx), scale(f3(A.cam_up), ry)), f3(A.cam_fwd)));
            const F3 ld = unit(xform_vector<4>(M.inv, wd));
            float entry;
            if (!slab_t(f3(M.cam_o), recip(ld), M.root_min, M.root_max, entry)) continue;
            uint64_t full = 0, masked = 0;
            uint32_t root_a, root_b;
            const bool a = shear_gate<false>(plain, f3(M.cam_o), ld, full, root_a), b = shear_gate<true>(M, f3(M.cam_o), ld, masked, root_b);
            if (a != b) { bad++; continue; }
            if (!a) continue;
            n++;
            skipped += bits_set(all & ~check);
            if (full & ~masked) bad++;
        }
    // the shadow direction, from an origin the gate admits
    const F3 o = f3(M.cam_o);
    uint64_t full = 0, masked = 0;
    uint32_t root_a, root_b;
    const bool a = shear_gate<false>(plain, o, f3(M.lit_d), full, root_a), b = shear_gate<true>(M, o, f3(M.lit_d), masked, root_b);
    if (a != b || (a && (full & ~masked))) bad++;
    if (a) n++;
    *looked = n; *spared = skipped;
    return bad;
}
int emul_mesh_info(void* sp, int mesh, int32_t* n_pairs, int32_t* depth, uint32_t* root_ref) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return RT_ERR_INVALID;
    *n_pairs = (int32_t)s->meshes[mesh].pk.pairs.size();
    *depth = s->meshes[mesh].pk.depth;
    *root_ref = s->meshes[mesh].pk.root_ref;
    return RT_OK;
}

int emul_mesh_prios(void* sp, int mesh, uint32_t* out) {
    EmulScene* s = (EmulScene*)sp;
    if (mesh < 0 || mesh >= (int)s->meshes.size()) return RT_ERR_INVALID;
    const auto& t = s->meshes[mesh].pk.tris;
    for (size_t i = 0; i < t.size(); i++) out[i] = t[i].prio;
    return RT_OK;
}

// The tiles outside of which the product skips a mesh for camera rays (mesh_screen_rect).
void emul_screen_rect(const rt_transform* tf, const float* bmin, const float* bmax, const rt_camera* cam, const rt_light* light,
                      int32_t width, int32_t height, int32_t* rect4) {
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, *light, width, height, A);
    mesh_screen_rect(*tf, bmin, bmax, A, rect4);
}

// The tiles outside of which the product skips a mesh for shadow rays (mesh_shadow_rect); scene bounds as 3 + 3 doubles.
void emul_shadow_rect(const rt_transform* tf, const float* bmin, const float* bmax, const double* scene_lo, const double* scene_hi,
                      const rt_camera* cam, const rt_light* light, int32_t width, int32_t height, int32_t* rect4) {
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, *light, width, height, A);
    mesh_shadow_rect(*tf, bmin, bmax, scene_lo, scene_hi, A, rect4);
}

// Per-frame constants as the product computes them: inv rows 0..3 (16 floats) and nrm (9 floats).
void emul_placement(const rt_transform* tf, const rt_camera* cam, const rt_light* light, float* inv16, float* nrm9,
                    float* cam_o3, float* lit_d3, uint32_t* w_mode) {
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, *light, 16, 16, A);
    MeshArgs M; memset(&M, 0, sizeof M);
    fill_placement(*tf, A, M);
    memcpy(inv16, M.inv, 48); memcpy(inv16 + 12, M.wrow, 16);
    memcpy(nrm9, M.nrm, 36);
    memcpy(cam_o3, M.cam_o, 12); memcpy(lit_d3, M.lit_d, 12);
    *w_mode = M.w_mode;
}

int emul_render(void* sp, const rt_camera* cam, const rt_light* light, const rt_transform* tfs, int32_t n_tf,
                int32_t width, int32_t height, uint8_t* rgb, int32_t* hit_id, float* hit_t, uint8_t* shadow,
                rt_counters* counters, int32_t full, uint32_t* px_slabs, uint32_t* px_tris) {
    EmulScene* s = (EmulScene*)sp;
    if (n_tf != (int32_t)s->meshes.size() || n_tf > kMaxMeshes) { s->err = "transform count"; return RT_ERR_INVALID; }
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, *light, width, height, A);
    A.n_mesh = n_tf;
    uint32_t base = 0;
    for (int m = 0; m < n_tf; m++) {
        MeshArgs& M = A.mesh[m];
        const EmulMesh& E = s->meshes[(size_t)m];
        M.pair_base = base; M.pair_count = E.pk.tree_pairs; base += (uint32_t)E.pk.pairs.size();
        M.tris = E.pk.tris.data(); M.attrs = E.pk.attrs.data(); M.texels = E.texels.data();
        memcpy(M.root_min, E.pk.root_min, 12); memcpy(M.root_max, E.pk.root_max, 12);
        M.root_ref = E.pk.root_ref; M.ref_root = E.pk.ref_root;
        if (s->loose && !E.pk.loose_tris.empty() && E.pk.grid_axis < 4) { M.root_ref = M.ref_root = E.pk.loose_root; M.tris = E.pk.loose_tris.data(); }
        set_grid_args(E.pk, E.pk.grid.data(), M); M.tex_w = E.tw; M.tex_h = E.th; M.empty = E.pk.empty ? 1u : 0u; M.wide = E.pk.wide ? 1u : 0u;
        fill_placement(tfs[m], A, M);
        shear_frame_masks(E.pk.grid.data(), M);
        mesh_screen_rect(tfs[m], M.root_min, M.root_max, A, A.vis[m]);
    }
    {   // shadow-ray rectangles, as the product's fill_frame computes them
        double slo[3] = {1e300, 1e300, 1e300}, shi[3] = {-1e300, -1e300, -1e300};
        bool any = false;
        for (int m = 0; m < n_tf; m++) {
            const int32_t whole[4] = {0, 0, (width + 7) / 8, (height + 7) / 8};
            memcpy(A.vis_shadow[m], whole, sizeof whole);
            if (A.mesh[m].empty) continue;
            const WorldBox wb = world_box(tfs[m], A.mesh[m].root_min, A.mesh[m].root_max);
            for (int i = 0; i < 3; i++) { slo[i] = wb.lo[i] < slo[i] ? wb.lo[i] : slo[i]; shi[i] = wb.hi[i] > shi[i] ? wb.hi[i] : shi[i]; }
            any = true;
        }
        for (int m = 0; any && m < n_tf; m++) mesh_shadow_rect(tfs[m], A.mesh[m].root_min, A.mesh[m].root_max, slo, shi, A, A.vis_shadow[m]);
    }
    for (int m = 0; m < n_tf && s->vis_shrink; m++)
        for (int32_t* r : {A.vis[m], A.vis_shadow[m]}) { r[0] += s->vis_shrink; r[1] += s->vis_shrink; r[2] -= s->vis_shrink; r[3] -= s->vis_shrink; }
    GlobalNodes nodes; nodes.all = s->all_pairs.data();
    BatchNodes batch_nodes; batch_nodes.all = s->all_pairs.data();       // rt_render_frames' kernel: with the per-frame part of the gate
    unsigned long long c_slab = 0, c_tri = 0, c_hit = 0, c_dark = 0;
    int high = 0;
    CheckTotals checks;
    #pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_slab, c_tri, c_hit, c_dark) reduction(max : high)
    for (int y = 0; y < height; y++) {
        CheckTotals row;
        for (int x = 0; x < width; x++) {
            ArrayStack st;
            Work w;
            // full: 1 = reference visiting order; 0 = nearer child first with distance culling; 2 = the same without work
            // counts, as the kernels run it: meshes are skipped for tiles outside their screen rectangle; 3 = as 2 with the
            // self-checking party (RT_FLAG_VERIFY_SKIPS); 4 = as 2 for a caller that stores the colour only (rt_render_frames' kernel)
            PixelOut r = full == 1 ? shade_pixel<true, true, false>(A, nodes, Solo(), x, y, st, w)
                       : full == 2 ? shade_pixel<false, false, false>(A, nodes, Solo(), x, y, st, w)
                       : full == 3 ? shade_pixel<false, false, false>(A, nodes, SoloVerify(), x, y, st, w)
                       : full == 4 ? shade_pixel<false, false, false>(A, batch_nodes, Solo(), x, y, st, w, true)
                                   : shade_pixel<true, false, false>(A, nodes, Solo(), x, y, st, w);
            row.add(w);
            const size_t p = (size_t)y * width + x;
            rgb[p * 3] = r.r; rgb[p * 3 + 1] = r.g; rgb[p * 3 + 2] = r.b;
            if (hit_id) hit_id[p] = r.hit_id;
            if (hit_t) hit_t[p] = r.hit_t;
            if (shadow) shadow[p] = r.shadow;
            if (px_slabs) px_slabs[p] = w.slabs;
            if (px_tris) px_tris[p] = w.tris;
            c_slab += w.slabs; c_tri += w.tris; c_hit += r.hit; c_dark += r.dark;
            if (st.high > high) high = st.high;
        }
        #pragma omp critical
        checks.merge(row);
    }
    s->stack_high = high;
    if (counters) {
        counters->primary_rays = (uint64_t)width * height; counters->shadow_rays = c_hit;
        counters->slab_tests = c_slab; counters->tri_tests = c_tri;
        counters->hit_pixels = c_hit; counters->shadowed_pixels = c_dark;
        checks.store(counters);
    }
    return RT_OK;
}

void emul_sample_offset(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float* jx, float* jy) { sample_offset(x, y, s, seed, *jx, *jy); }

// The product's sampled pixel (rt_sampled.h) over a frame.
int emul_render_sampled(void* sp, const rt_camera* cam, const rt_light* lights, int32_t n_lights, const rt_transform* tfs, int32_t n_tf,
                        int32_t spp, uint32_t seed, int32_t width, int32_t height, uint8_t* rgb, int32_t* hit_id, rt_counters* counters,
                        int32_t verify) {
    EmulScene* s = (EmulScene*)sp;
    if (n_tf != (int32_t)s->meshes.size() || n_tf > kMaxMeshes || n_lights < 1 || n_lights > kMaxLights) { s->err = "counts"; return RT_ERR_INVALID; }
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, lights[0], width, height, A);
    A.n_mesh = n_tf;
    uint32_t base = 0;
    for (int m = 0; m < n_tf; m++) {
        MeshArgs& M = A.mesh[m];
        const EmulMesh& E = s->meshes[(size_t)m];
        M.pair_base = base; M.pair_count = E.pk.tree_pairs; base += (uint32_t)E.pk.pairs.size();
        M.tris = E.pk.tris.data(); M.attrs = E.pk.attrs.data(); M.texels = E.texels.data();
        memcpy(M.root_min, E.pk.root_min, 12); memcpy(M.root_max, E.pk.root_max, 12);
        M.root_ref = E.pk.root_ref; M.ref_root = E.pk.ref_root; set_grid_args(E.pk, E.pk.grid.data(), M); M.tex_w = E.tw; M.tex_h = E.th; M.empty = E.pk.empty ? 1u : 0u; M.wide = E.pk.wide ? 1u : 0u;
        fill_placement(tfs[m], A, M);
        shear_frame_masks(E.pk.grid.data(), M);
        mesh_screen_rect(tfs[m], M.root_min, M.root_max, A, A.vis[m]);
    }
    SampleArgs X;
    fill_samples(lights, n_lights, spp, seed, A, X);
    GlobalNodes nodes; nodes.all = s->all_pairs.data();
    unsigned long long c_slab = 0, c_tri = 0, c_hit = 0, c_dark = 0;
    CheckTotals checks;
    #pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_slab, c_tri, c_hit, c_dark)
    for (int y = 0; y < height; y++) {
        CheckTotals row;
        for (int x = 0; x < width; x++) {
            ArrayStack st;
            Work w;
            const SampledOut r = verify ? shade_sampled_pixel<false>(A, X, nodes, SoloVerify(), x, y, st, w)
                                        : shade_sampled_pixel<true>(A, X, nodes, Solo(), x, y, st, w);
            const size_t p = (size_t)y * width + x;
            rgb[p * 3] = r.r; rgb[p * 3 + 1] = r.g; rgb[p * 3 + 2] = r.b;
            if (hit_id) hit_id[p] = r.hit_id;
            c_slab += w.slabs; c_tri += w.tris; c_hit += r.hits; c_dark += r.darks;
            row.add(w);
        }
        #pragma omp critical
        checks.merge(row);
    }
    if (counters) {
        counters->primary_rays = (uint64_t)width * height * (uint64_t)spp; counters->shadow_rays = c_hit * (uint64_t)n_lights;
        counters->slab_tests = c_slab; counters->tri_tests = c_tri;
        counters->hit_pixels = c_hit; counters->shadowed_pixels = c_dark;
        checks.store(counters);
    }
    return RT_OK;
}

// Instanced frames rendered after this call also record, per pixel, the box and triangle tests of its two rays
// ({boxes, triangles} as uint32 pairs, row-major); a null pointer stops the recording.
void emul_work_plane(void* sp, uint32_t* per_pixel) { ((EmulScene*)sp)->work_plane = per_pixel; }

// The product's instanced pixel (rt_instanced.h) with the product's instance hierarchy (rt_pack.h).
int emul_render_instanced(void* sp, const rt_camera* cam, const rt_light* light, const int32_t* mesh_ids, const rt_transform* tfs, int32_t n_inst,
                          int32_t width, int32_t height, uint8_t* rgb, int32_t* hit_inst, int32_t* hit_tri, float* hit_t, uint8_t* shadow,
                          rt_counters* counters, int32_t* tlas_pairs, int32_t* tlas_depth, int32_t verify) {
    EmulScene* s = (EmulScene*)sp;
    FrameArgs A; memset(&A, 0, sizeof A);
    fill_camera(*cam, *light, width, height, A);
    A.n_mesh = 0;
    std::vector<uint32_t> base(s->meshes.size() + 1, 0);
    for (size_t m = 0; m < s->meshes.size(); m++) base[m + 1] = base[m] + (uint32_t)s->meshes[m].pk.pairs.size();
    std::vector<MeshArgs> inst((size_t)n_inst);
    std::vector<InstanceBox> boxes;
    std::vector<uint32_t> ids;
    for (int32_t i = 0; i < n_inst; i++) {
        if (mesh_ids[i] < 0 || mesh_ids[i] >= (int32_t)s->meshes.size()) { s->err = "mesh id"; return RT_ERR_INVALID; }
        const EmulMesh& E = s->meshes[(size_t)mesh_ids[i]];
        MeshArgs& M = inst[(size_t)i];
        memset(&M, 0, sizeof M);
        M.pair_base = base[(size_t)mesh_ids[i]]; M.pair_count = E.pk.tree_pairs;
        M.tris = E.pk.tris.data(); M.attrs = E.pk.attrs.data(); M.texels = E.texels.data();
        memcpy(M.root_min, E.pk.root_min, 12); memcpy(M.root_max, E.pk.root_max, 12);
        M.root_ref = E.pk.root_ref; M.ref_root = E.pk.ref_root;
        if (s->loose && !E.pk.loose_tris.empty()) { M.root_ref = M.ref_root = E.pk.loose_root; M.tris = E.pk.loose_tris.data(); }
        set_grid_args(E.pk, E.pk.grid.data(), M); M.tex_w = E.tw; M.tex_h = E.th; M.empty = E.pk.empty ? 1u : 0u; M.wide = E.pk.wide ? 1u : 0u;
        fill_placement(tfs[i], A, M);
        if (!M.empty) {
            boxes.push_back(instance_box(tfs[i], M.root_min, M.root_max));
            if (M.wide) reach_camera(boxes.back(), A.cam_pos);
            ids.push_back((uint32_t)i);
        }
    }
    s->box_allowance = settled_allowance(s->box_allowance, scene_box_allowance(boxes, A.cam_pos));
    widen_boxes(boxes, s->box_allowance);
    Tlas T;
    if (s->keep_hierarchy) s->hierarchy_reused += s->standing.update(boxes, ids, (size_t)n_inst, T) ? 1 : 0;
    else TlasBuilder(boxes, ids, T).run();
    if (tlas_pairs) *tlas_pairs = (int32_t)T.pairs.size();
    if (tlas_depth) *tlas_depth = T.depth;
    InstancedArgs I;
    std::vector<PairNode> direct(T.pairs.size());
    write_direct(T, direct.data());
    I.inst = inst.data(); I.n_inst = (int32_t)ids.size(); I.n_total = n_inst; I.tlas_root = direct_ref(T, T.root_ref); I.tlas = direct.data();
    memcpy(I.root_min, T.root_min, 12); memcpy(I.root_max, T.root_max, 12);
    GlobalNodes nodes; nodes.all = s->all_pairs.data();
    unsigned long long c_slab = 0, c_tri = 0, c_hit = 0, c_dark = 0;
    CheckTotals checks;
    #pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_slab, c_tri, c_hit, c_dark)
    for (int y = 0; y < height; y++) {
        CheckTotals row;
        for (int x = 0; x < width; x++) {
            ArrayStack st, ts;
            Work w;
            const InstancedOut r = verify ? shade_instanced_pixel<false>(A, I, nodes, SoloVerify(), x, y, st, ts, w)
                                          : shade_instanced_pixel<true>(A, I, nodes, Solo(), x, y, st, ts, w);
            row.add(w);
            const size_t p = (size_t)y * width + x;
            rgb[p * 3] = r.r; rgb[p * 3 + 1] = r.g; rgb[p * 3 + 2] = r.b;
            if (hit_inst) hit_inst[p] = r.inst;
            if (hit_tri) hit_tri[p] = r.tri;
            if (hit_t) hit_t[p] = r.hit_t;
            if (shadow) shadow[p] = r.shadow;
            c_slab += w.slabs; c_tri += w.tris; c_hit += r.hit; c_dark += r.dark;
            if (s->work_plane) { s->work_plane[p * 2] = w.slabs; s->work_plane[p * 2 + 1] = w.tris; }
        }
        #pragma omp critical
        checks.merge(row);
    }
    if (counters) {
        counters->primary_rays = (uint64_t)width * height; counters->shadow_rays = c_hit;
        counters->slab_tests = c_slab; counters->tri_tests = c_tri;
        counters->hit_pixels = c_hit; counters->shadowed_pixels = c_dark;
        checks.store(counters);
    }
    return RT_OK;
}

// The product's upscale strips (rt_bicubic.h) over a whole frame: columns x strips of `rows` output rows,
// as the kernel divides the work.
void emul_upscale(const uint8_t* frame, int32_t width, int32_t height, int32_t scaled_width, int32_t scaled_height,
                  int32_t rows, uint32_t* argb) {
    #pragma omp parallel for schedule(static)
    for (int y0 = 0; y0 < scaled_height; y0 += rows)
        for (int x = 0; x < scaled_width; x++)
            upscale_strip(frame, width, height, scaled_width, scaled_height, x, y0, rows,
                          [&](int y, uint32_t v) { argb[(size_t)y * scaled_width + x] = v; });
}

}  // extern "C"
