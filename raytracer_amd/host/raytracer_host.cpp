// raytracer_host.cpp -- the reference's host API over the MI355X rendering core.
//
// Scene bookkeeping, OBJ/WebP loading and the BVH build stay on the host (they run once at
// load time in the reference too); render_scene() hands the frame to librt_mi355.so.
#include "raytracer_host.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_mi355.h"
#include "../csrc/rt_exact.h"
#include "../csrc/rt_pack.h"

static_assert(sizeof(Triangle) == sizeof(rt_triangle) && sizeof(Transform) == sizeof(rt_transform), "layout");
static_assert(sizeof(Camera) == sizeof(rt_camera) && sizeof(DirectionalLight) == sizeof(rt_light), "layout");
static_assert(sizeof(BVHNode) == sizeof(rt_tree_node) && sizeof(BVHNode) == 48, "layout");

namespace {

// ------------------------------------------------------------------ device side-table
// The reference's Scene has no room for a device handle and is passed around by value, so the
// GPU context lives in a table keyed by the scene's frame-pointer array (unique while alive).
// `stamp` tells two trees apart that were built one after the other at the same address: malloc readily hands a
// freed root's address to the next create_bvh, and a rebuilt tree means changed triangles.
struct MeshKey {
    const void* tris; const void* tex; const void* root; size_t n; int tw, th; uint64_t stamp;
    bool operator==(const MeshKey& o) const {
        return tris == o.tris && tex == o.tex && root == o.root && n == o.n && tw == o.tw && th == o.th && stamp == o.stamp;
    }
};
std::mutex g_stamp_lock;
std::map<const void*, uint64_t> g_root_stamp;     // root of every tree built by this library -> its build number
uint64_t g_builds = 0;
void stamp_root(const void* root) {
    if (!root) return;
    std::lock_guard<std::mutex> g(g_stamp_lock);
    g_root_stamp[root] = ++g_builds;
}
void forget_root(const void* root) {
    std::lock_guard<std::mutex> g(g_stamp_lock);
    g_root_stamp.erase(root);
}
uint64_t stamp_of(const void* root) {
    std::lock_guard<std::mutex> g(g_stamp_lock);
    auto it = g_root_stamp.find(root);
    return it == g_root_stamp.end() ? 0 : it->second;       // 0: a tree the caller linked together itself
}
// One GPU that renders frames of a scene.  A scene normally has one; after rt_host_set_gpus it has one per listed
// device and frame f goes to entry f mod n (frames are independent, raytracer.c:40-57).
struct GpuState {
    rt_ctx* ctx = nullptr;
    int device = 0;
    int in_flight = 1;                    // the context's "frames_in_flight" as last set
    std::vector<MeshKey> geometries;      // distinct (triangles, BVH, texture) uploaded, in upload order
};
struct SceneState {
    std::vector<GpuState> gpus;
    int last_status = 0;
};
std::mutex g_lock;
std::map<const void*, SceneState> g_states;
std::map<const void*, void*> g_pinned_frames;    // Scene.frames array -> the page-locked block its frames live in
int g_device = -1;
int g_in_flight = 1;                              // rt_host_set_frames_in_flight
std::vector<int> g_gpu_list;                      // rt_host_set_gpus; empty = the one device of pick_device()

int pick_device() {
    if (g_device >= 0) return g_device;
    const char* e = getenv("RT_DEVICE");
    return e ? atoi(e) : 0;
}

SceneState* state_of(const Scene* scene, bool create) {
    auto it = g_states.find((const void*)scene->frames);
    if (it != g_states.end()) return &it->second;
    if (!create) return nullptr;
    SceneState& st = g_states[(const void*)scene->frames];
    if (g_gpu_list.empty()) { st.gpus.resize(1); st.gpus[0].device = pick_device(); }
    else { st.gpus.resize(g_gpu_list.size()); for (size_t i = 0; i < st.gpus.size(); i++) st.gpus[i].device = g_gpu_list[i]; }
    return &st;
}

// Brings the GPU's geometry list in line with the scene's meshes.  Scene.meshes may repeat a geometry
// (the same triangle array, BVH and texture under another transform): each distinct one is uploaded once
// and geometry_of[i] says which one mesh i uses.
int sync_scene(Scene* scene, GpuState& st, std::vector<int32_t>& geometry_of) {
    if (!st.ctx) {
        int rc = rt_ctx_create(st.device, &st.ctx);
        if (rc != RT_OK) { fprintf(stderr, "render_scene: cannot open GPU %d: %s\n", st.device, rt_status_string(rc)); return rc; }
    }
    std::vector<MeshKey> now;
    geometry_of.assign(scene->mesh_count, -1);
    for (size_t i = 0; i < scene->mesh_count; i++) {
        const Mesh& m = scene->meshes[i];
        const MeshKey k{m.triangles, m.texture_data, m.bvh.root, m.triangle_count, m.texture_width, m.texture_height, stamp_of(m.bvh.root)};
        size_t g = 0;
        while (g < now.size() && !(now[g] == k)) g++;
        if (g == now.size()) now.push_back(k);
        geometry_of[i] = (int32_t)g;
    }
    if (now == st.geometries) return RT_OK;
    int rc = rt_scene_clear(st.ctx);
    st.geometries.clear();
    for (size_t g = 0; rc == RT_OK && g < now.size(); g++) {
        const MeshKey& k = now[g];
        if (!k.root || !k.tex) { fprintf(stderr, "render_scene: a mesh of the scene was not loaded\n"); return RT_ERR_INVALID; }
        rc = rt_mesh_upload_tree(st.ctx, (const rt_triangle*)k.tris, (int32_t)k.n, (const rt_tree_node*)k.root, (const uint8_t*)k.tex, k.tw, k.th, nullptr);
    }
    if (rc != RT_OK) { fprintf(stderr, "render_scene: scene upload failed: %s (%s)\n", rt_status_string(rc), rt_last_error(st.ctx)); return rc; }
    st.geometries = now;
    return RT_OK;
}

// Waits for the frames render_scene left in flight (g_lock held by the caller).
int wait_frames(SceneState* st) {
    if (!st) return RT_OK;
    int rc = RT_OK;
    for (GpuState& g : st->gpus) {
        if (!g.ctx || g.in_flight <= 1) continue;
        const int r = rt_sync(g.ctx, nullptr);
        if (r != RT_OK) { fprintf(stderr, "render_scene: %s (%s)\n", rt_status_string(r), rt_last_error(g.ctx)); st->last_status = r; rc = r; }
    }
    return rc;
}

int render_into(Scene* scene, uint8_t* rgb, int32_t* hit_id, float* hit_t, uint8_t* shadow) {
    std::lock_guard<std::mutex> g(g_lock);
    SceneState* state = state_of(scene, true);
    // the GPU of this frame: frame f on entry f mod n of the scene's GPU list
    const size_t turn = scene->current_frame > 0 ? (size_t)scene->current_frame % state->gpus.size() : 0;
    GpuState* st = &state->gpus[turn];
    std::vector<int32_t> geometry_of;
    int rc = sync_scene(scene, *st, geometry_of);
    const int in_flight = (hit_id || hit_t || shadow) ? 1 : g_in_flight;
    if (rc == RT_OK && st->in_flight != in_flight) {
        rc = rt_set_option(st->ctx, "frames_in_flight", in_flight);      // waits for the frames queued so far
        if (rc == RT_OK) st->in_flight = in_flight;
    }
    if (rc == RT_OK) {
        // one mesh per geometry, in order, and few enough: the plain frame; otherwise the mesh list is a
        // list of placed geometries
        bool plain = scene->mesh_count == st->geometries.size() && scene->mesh_count <= (size_t)RT_MAX_MESHES;
        if (plain) {
            std::vector<rt_transform> tfs(scene->mesh_count);
            for (size_t i = 0; i < scene->mesh_count; i++) memcpy(&tfs[i], &scene->meshes[i].transform, sizeof(rt_transform));
            rc = rt_render_frame_host(st->ctx, (const rt_camera*)&scene->camera, (const rt_light*)&scene->light, tfs.data(),
                                      (int32_t)tfs.size(), scene->width, scene->height, 0, rgb, hit_id, hit_t, shadow);
        } else if (hit_id || hit_t || shadow) {
            rc = RT_ERR_UNSUPPORTED;
            fprintf(stderr, "render_scene: debug planes are not available for scenes with repeated or more than %d meshes\n", RT_MAX_MESHES);
        } else {
            std::vector<rt_instance> inst(scene->mesh_count);
            for (size_t i = 0; i < scene->mesh_count; i++) {
                inst[i].mesh = geometry_of[i];
                memcpy(&inst[i].transform, &scene->meshes[i].transform, sizeof(rt_transform));
            }
            rc = rt_render_frame_instanced_host(st->ctx, (const rt_camera*)&scene->camera, (const rt_light*)&scene->light, inst.data(),
                                                (int32_t)inst.size(), scene->width, scene->height, 0, rgb);
        }
        if (rc != RT_OK) fprintf(stderr, "render_scene: %s (%s)\n", rt_status_string(rc), rt_last_error(st->ctx));
    }
    state->last_status = rc;
    return rc;
}

// ------------------------------------------------------------------ libwebp at run time
// The image ships libwebp.so.7 but no headers; only the two decode entry points the reference
// uses (mesh.c:105,131) are needed, so they are looked up by name.
typedef uint8_t* (*webp_decode_fn)(const uint8_t*, size_t, int*, int*);
typedef void (*webp_free_fn)(void*);
webp_decode_fn g_webp_decode = nullptr;
webp_free_fn g_webp_free = nullptr;

bool load_webp() {
    if (g_webp_decode) return true;
    static const char* names[] = {"libwebp.so.7", "libwebp.so", "libwebp.so.6", nullptr};
    for (int i = 0; names[i]; i++) {
        void* h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!h) continue;
        g_webp_decode = (webp_decode_fn)dlsym(h, "WebPDecodeRGBA");
        g_webp_free = (webp_free_fn)dlsym(h, "WebPFree");
        if (g_webp_decode) return true;
    }
    return false;
}

// ------------------------------------------------------------------ OBJ reading
// Same grammar the reference accepts (mesh.c:41-76): "v x y z", "vt u v", "vn x y z" and
// "f a/b/c a/b/c a/b/c" with 1-based indices; everything else is skipped.  Storage grows as
// needed instead of the reference's fixed million-element arrays.
const char* skip_ws(const char* p) { while (*p == ' ' || *p == '\t') p++; return p; }

int read_floats(const char* p, float* out, int n) {
    int got = 0;
    for (; got < n; got++) {
        char* end = nullptr;
        float v = strtof(p, &end);
        if (end == p) break;
        out[got] = v; p = end;
    }
    return got;
}

// One face corner: "v", "v/vt", "v//vn" or "v/vt/vn".  Missing parts are reported as 0 (OBJ indices
// start at 1); negative indices count back from the end of their list.
bool read_corner(const char*& p, long idx[3], size_t n_pos, size_t n_uv, size_t n_nor) {
    idx[0] = idx[1] = idx[2] = 0;
    const size_t counts[3] = {n_pos, n_uv, n_nor};
    for (int k = 0; k < 3; k++) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end != p) { idx[k] = v < 0 ? (long)counts[k] + v + 1 : v; p = end; }
        else if (k == 0) return false;
        if (*p != '/') break;
        p++;
    }
    return true;
}

float g_save_quality = 75.0f;   // WebPConfigInit's default (scene.c:174)
int g_save_threads = 0;
int g_obj_mode = 0;      // 0: the reference's reader; 1: tolerant reader (rt_host_set_obj_mode)

}  // namespace

extern "C" {

// ------------------------------------------------------------------ small constructors
Camera create_camera(Vec3 position, Vec3 look_at, Vec3 up, float fov) {            // camera.c:8-10
    Camera c; c.position = position; c.look_at = look_at; c.up = up; c.fov = fov; return c;
}

DirectionalLight create_directional_light(Vec3 direction, Vec3 color) {            // light.c:3-5
    const rt::F3 d = rt::unit(rt::f3(direction.x, direction.y, direction.z));
    DirectionalLight l; l.direction = Vec3{d.x, d.y, d.z}; l.color = color; return l;
}

// ------------------------------------------------------------------ BVH (host build)
BVHNode* create_bvh_node(Triangle* triangles, int start, int count) {              // bvh.c:4-59
    if (count < 0) return nullptr;
    std::vector<rt_flat_node> flat((size_t)2 * (size_t)(count > 0 ? count : 1) + 1);
    rt::BvhBuilder b((rt_triangle*)triangles, flat.data(), (int32_t)flat.size());
    if (b.build(start, count) < 0) return nullptr;
    std::vector<BVHNode*> heap((size_t)b.used(), nullptr);
    for (int32_t i = b.used() - 1; i >= 0; i--) {            // children have larger indices than their parent
        BVHNode* n = (BVHNode*)malloc(sizeof(BVHNode));
        if (!n) return nullptr;
        memcpy(&n->bounds.min, &flat[(size_t)i].bounds_min, 12);
        memcpy(&n->bounds.max, &flat[(size_t)i].bounds_max, 12);
        n->start_idx = flat[(size_t)i].start_idx; n->triangle_count = flat[(size_t)i].triangle_count;
        n->left = flat[(size_t)i].left >= 0 ? heap[(size_t)flat[(size_t)i].left] : nullptr;
        n->right = flat[(size_t)i].right >= 0 ? heap[(size_t)flat[(size_t)i].right] : nullptr;
        heap[(size_t)i] = n;
    }
    stamp_root(heap[0]);
    return heap[0];
}

BVH create_bvh(Triangle* triangles, size_t count) {                                 // bvh.c:61-67
    BVH b; b.triangles = triangles; b.triangle_count = count;
    b.root = create_bvh_node(triangles, 0, (int)count);
    return b;
}

static void free_subtree(BVHNode* node) {
    if (!node) return;
    free_subtree(node->left);
    free_subtree(node->right);
    free(node);
}

void destroy_bvh_node(BVHNode* node) {                                              // bvh.c:69-73
    if (!node) return;
    forget_root(node);
    free_subtree(node);
}

void destroy_bvh(BVH* bvh) {                                                        // bvh.c:75-78
    if (!bvh) return;
    destroy_bvh_node(bvh->root);
    bvh->root = nullptr;
}

// ------------------------------------------------------------------ meshes
Mesh create_mesh(const char* obj_filename, const char* texture_filename) {          // mesh.c:6-119
    Mesh mesh;
    memset(&mesh, 0, sizeof(mesh));
    FILE* f = fopen(obj_filename, "r");
    if (!f) { fprintf(stderr, "Failed to open %s\n", obj_filename); return mesh; }
    std::vector<Vec3> pos, nor;
    std::vector<Vec2> uv;
    std::vector<Triangle> tris;
    char* line = nullptr;        // getline: no limit on the length of a line (the reference's is 255 characters)
    size_t cap = 0;
    while (getline(&line, &cap, f) >= 0) {
        const char* p = line;
        if (p[0] == 'v' && p[1] == ' ') {
            float v[3] = {0, 0, 0}; read_floats(p + 2, v, 3); pos.push_back(Vec3{v[0], v[1], v[2]});
        } else if (p[0] == 'v' && p[1] == 't') {
            float v[2] = {0, 0}; read_floats(p + 3, v, 2); uv.push_back(Vec2{v[0], v[1]});
        } else if (p[0] == 'v' && p[1] == 'n') {
            float v[3] = {0, 0, 0}; read_floats(p + 3, v, 3); nor.push_back(Vec3{v[0], v[1], v[2]});
        } else if (p[0] == 'f') {
            // corners of the face
            std::vector<long> c;
            const char* q = p + 1;
            for (;;) {
                q = skip_ws(q);
                long idx[3];
                if (!read_corner(q, idx, pos.size(), uv.size(), nor.size())) break;
                c.insert(c.end(), idx, idx + 3);
            }
            const size_t corners = c.size() / 3;
            bool ok = corners >= 3;
            for (size_t k = 0; k < corners && ok; k++) {
                ok = c[3 * k] >= 1 && (size_t)c[3 * k] <= pos.size() && c[3 * k + 1] >= 0 && (size_t)c[3 * k + 1] <= uv.size() &&
                     c[3 * k + 2] >= 0 && (size_t)c[3 * k + 2] <= nor.size();
                // the reference reads exactly "v/vt/vn" three times (mesh.c:59-63): every part must be there
                if (g_obj_mode == 0 && k < 3 && (c[3 * k + 1] == 0 || c[3 * k + 2] == 0)) ok = false;
            }
            if (!ok) { fprintf(stderr, "%s: skipping face the reader cannot use: %s", obj_filename, line); continue; }
            // reference reader: the first three corners are the triangle, further corners are ignored;
            // tolerant reader: a fan over all corners
            const size_t fan = g_obj_mode == 0 ? 1 : corners - 2;
            for (size_t k = 0; k < fan; k++) {
                const size_t i0 = 0, i1 = k + 1, i2 = k + 2;
                Triangle t;
                t.v0 = pos[(size_t)c[3 * i0] - 1]; t.v1 = pos[(size_t)c[3 * i1] - 1]; t.v2 = pos[(size_t)c[3 * i2] - 1];
                const Vec2 no_uv = {0.0f, 0.0f};
                t.t0 = c[3 * i0 + 1] ? uv[(size_t)c[3 * i0 + 1] - 1] : no_uv;
                t.t1 = c[3 * i1 + 1] ? uv[(size_t)c[3 * i1 + 1] - 1] : no_uv;
                t.t2 = c[3 * i2 + 1] ? uv[(size_t)c[3 * i2 + 1] - 1] : no_uv;
                // a corner without a normal gets the triangle's geometric normal
                const rt::F3 a0 = rt::f3(t.v0.x, t.v0.y, t.v0.z), a1 = rt::f3(t.v1.x, t.v1.y, t.v1.z), a2 = rt::f3(t.v2.x, t.v2.y, t.v2.z);
                const rt::F3 g = rt::unit(rt::cross(rt::sub(a1, a0), rt::sub(a2, a0)));
                const Vec3 face_n = {g.x, g.y, g.z};
                t.n0 = c[3 * i0 + 2] ? nor[(size_t)c[3 * i0 + 2] - 1] : face_n;
                t.n1 = c[3 * i1 + 2] ? nor[(size_t)c[3 * i1 + 2] - 1] : face_n;
                t.n2 = c[3 * i2 + 2] ? nor[(size_t)c[3 * i2 + 2] - 1] : face_n;
                tris.push_back(t);
            }
        }
    }
    free(line);
    fclose(f);
    mesh.triangles = (Triangle*)malloc(sizeof(Triangle) * (tris.empty() ? 1 : tris.size()));
    if (!mesh.triangles) return mesh;
    if (!tris.empty()) memcpy(mesh.triangles, tris.data(), sizeof(Triangle) * tris.size());
    mesh.triangle_count = tris.size();

    FILE* tf = fopen(texture_filename, "rb");
    if (!tf) { fprintf(stderr, "Failed to open texture %s\n", texture_filename); return mesh; }
    fseek(tf, 0, SEEK_END);
    long size = ftell(tf);
    fseek(tf, 0, SEEK_SET);
    std::vector<uint8_t> bytes((size_t)(size > 0 ? size : 0));
    const bool read_ok = size > 0 && fread(bytes.data(), 1, (size_t)size, tf) == (size_t)size;
    fclose(tf);
    if (!read_ok) return mesh;
    if (!load_webp()) { fprintf(stderr, "create_mesh: libwebp is not available, cannot decode %s\n", texture_filename); return mesh; }
    mesh.texture_data = g_webp_decode(bytes.data(), bytes.size(), &mesh.texture_width, &mesh.texture_height);   // mesh.c:105
    if (!mesh.texture_data) { fprintf(stderr, "create_mesh: %s is not a WebP image\n", texture_filename); return mesh; }

    mesh.bvh = create_bvh(mesh.triangles, mesh.triangle_count);                                                  // mesh.c:110
    printf("Loaded %zu vertices, %zu texcoords, %zu normals, %zu triangles\n", pos.size(), uv.size(), nor.size(), tris.size());
    return mesh;
}

void set_mesh_position(Mesh* mesh, Vec3 position) { mesh->transform.position = position; }     // mesh.c:121
void set_mesh_rotation(Mesh* mesh, Vec3 rotation) { mesh->transform.rotation = rotation; }     // mesh.c:125

void destroy_mesh(Mesh* mesh) {                                                                 // mesh.c:129-136
    if (!mesh) return;
    free(mesh->triangles);
    if (mesh->texture_data) { if (g_webp_free) g_webp_free(mesh->texture_data); else free(mesh->texture_data); }
    destroy_bvh(&mesh->bvh);
    mesh->triangles = nullptr; mesh->texture_data = nullptr; mesh->triangle_count = 0;
}

Vec3 sample_mesh_texture(const Mesh* mesh, float u, float v) {                                  // mesh.c:138-149
    u = u - floorf(u); v = v - floorf(v);
    const int x = (int)(u * (float)(mesh->texture_width - 1)), y = (int)(v * (float)(mesh->texture_height - 1));
    const unsigned char* t = mesh->texture_data + ((size_t)y * (size_t)mesh->texture_width + (size_t)x) * 4;
    return Vec3{t[0] / 255.0f, t[1] / 255.0f, t[2] / 255.0f};
}

// ------------------------------------------------------------------ scene
Scene create_scene(int width, int height, int duration_ms, int fps, float scale_factor) {      // scene.c:8-29
    Scene s;
    memset(&s, 0, sizeof(s));
    s.frame_count = (duration_ms * fps) / 1000;
    s.width = (int)(width * scale_factor);
    s.height = (int)(height * scale_factor);
    s.scale_factor = scale_factor;
    s.duration_ms = duration_ms; s.fps = fps;
    // The reference sizes every buffer for the unscaled image (scene.c:25); take the larger of the
    // two so that scale factors above 1 do not overrun it.
    size_t px = (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0);
    const size_t spx = (size_t)(s.width > 0 ? s.width : 0) * (size_t)(s.height > 0 ? s.height : 0);
    if (spx > px) px = spx;
    const int n = s.frame_count > 0 ? s.frame_count : 0;
    s.frames = (unsigned char**)calloc((size_t)n + 1, sizeof(unsigned char*));
    // With a GPU: one page-locked block for all frames (render_scene's copy back runs at bus speed and
    // touches no fresh pages).  Otherwise, and if that fails, one malloc per frame like the reference.
    const size_t each = (px * 3 + 1 + 63) & ~(size_t)63;
    void* block = nullptr;
    if (n > 0 && rt_pinned_alloc(each * (size_t)n, &block) == RT_OK) {
        for (int i = 0; i < n; i++) s.frames[i] = (unsigned char*)block + each * (size_t)i;
        std::lock_guard<std::mutex> g(g_lock);
        g_pinned_frames[(const void*)s.frames] = block;
    } else {
        for (int i = 0; i < n; i++) s.frames[i] = (unsigned char*)malloc(px * 3 + 1);
    }
    return s;
}

void add_mesh_to_scene(Scene* scene, Mesh mesh) {                                              // scene.c:31-35
    Mesh* grown = (Mesh*)realloc(scene->meshes, (scene->mesh_count + 1) * sizeof(Mesh));
    if (!grown) return;
    scene->meshes = grown;
    scene->meshes[scene->mesh_count++] = mesh;
}

void set_scene_camera(Scene* scene, Vec3 position, Vec3 look_at, Vec3 up, float fov) {         // scene.c:37
    scene->camera = create_camera(position, look_at, up, fov);
}

void set_scene_light(Scene* scene, Vec3 direction, Vec3 color) {                               // scene.c:41
    scene->light = create_directional_light(direction, color);
}

void next_frame(Scene* scene) {                                                                 // scene.c:45-50
    if (++scene->current_frame >= scene->frame_count) scene->current_frame = scene->frame_count - 1;
}

void render_scene(Scene* scene) {                                                               // scene.c:52-160
    if (!scene || !scene->frames || scene->current_frame < 0 || scene->current_frame >= scene->frame_count) return;
    render_into(scene, scene->frames[scene->current_frame], nullptr, nullptr, nullptr);
}

void save_scene(Scene* scene, const char* filename) {                                           // scene.c:162-225
    // Upscale on the GPU (scene.c:188-202), encode on host threads while the next frame is upscaled,
    // frame times as the reference gives them to its encoder (scene.c:204, 209).  No status is
    // returned, like the reference; problems are reported on stderr.
    if (!scene || !filename || !scene->frames || scene->frame_count <= 0) return;
    if (rt_host_wait(scene) != RT_OK) return;
    int sw = 0, sh = 0;
    if (rt_host_scaled_size(scene, &sw, &sh) != RT_OK) { fprintf(stderr, "save_scene: bad scene size\n"); return; }
    rt_host_animation* anim = nullptr;
    if (rt_host_animation_open(filename, sw, sh, g_save_quality, g_save_threads, &anim) != RT_OK) {
        fprintf(stderr, "save_scene: cannot write %s\n", filename);
        return;
    }
    std::vector<uint32_t> argb((size_t)sw * (size_t)sh);
    const int step = scene->duration_ms / scene->frame_count;
    int rc = RT_OK;
    for (int frame = 0; frame < scene->frame_count && rc == RT_OK; frame++) {
        rc = rt_host_upscale_frame(scene, frame, argb.data());
        // the encoder shows a frame from its timestamp to the next one's; the last one ends at duration_ms
        const int shown = frame + 1 < scene->frame_count ? step : scene->duration_ms - frame * step;
        if (rc == RT_OK) rc = rt_host_animation_add(anim, argb.data(), 1, shown);
    }
    const int rc_close = rt_host_animation_close(anim);
    if (rc != RT_OK || rc_close != RT_OK) fprintf(stderr, "save_scene: %s was not written completely (%s)\n", filename, rt_status_string(rc != RT_OK ? rc : rc_close));
}

void destroy_scene(Scene* scene) {                                                              // scene.c:227-239
    if (!scene) return;
    {
        std::lock_guard<std::mutex> g(g_lock);
        auto it = g_states.find((const void*)scene->frames);
        if (it != g_states.end()) {
            (void)wait_frames(&it->second);                    // copies into scene->frames may still be queued
            for (GpuState& gpu : it->second.gpus) if (gpu.ctx) rt_ctx_destroy(gpu.ctx);
            g_states.erase(it);
        }
    }
    free(scene->meshes);
    if (scene->frames) {
        void* block = nullptr;
        {
            std::lock_guard<std::mutex> g(g_lock);
            auto it = g_pinned_frames.find((const void*)scene->frames);
            if (it != g_pinned_frames.end()) { block = it->second; g_pinned_frames.erase(it); }
        }
        if (block) rt_pinned_free(block);
        else for (int i = 0; i < scene->frame_count; i++) free(scene->frames[i]);
        free(scene->frames);
    }
    scene->meshes = nullptr; scene->frames = nullptr; scene->mesh_count = 0;
}

// ------------------------------------------------------------------ additions
void rt_host_set_device(int device) { std::lock_guard<std::mutex> g(g_lock); g_device = device; }

int rt_host_last_status(const Scene* scene) {
    std::lock_guard<std::mutex> g(g_lock);
    SceneState* st = scene ? state_of(scene, false) : nullptr;
    return st ? st->last_status : RT_ERR_INVALID;
}

int rt_host_render_debug(Scene* scene, int32_t* hit_id, float* hit_t, uint8_t* shadow) {
    if (!scene || !scene->frames || scene->current_frame < 0 || scene->current_frame >= scene->frame_count) return RT_ERR_INVALID;
    return render_into(scene, scene->frames[scene->current_frame], hit_id, hit_t, shadow);
}

void rt_host_demo_animation(int frame, Transform* drone, Transform* treasure) {
    // raytracer.c:41: frame * (2.0f * M_PI / 120.0f) is a double product, stored to a float.
    const float t = (float)(frame * ((double)2.0f * 3.14159265358979323846 / (double)120.0f));
    if (drone) {
        drone->position = Vec3{2.0f * cosf(t), 1.0f + 0.2f * sinf(2 * t), 2.0f * sinf(t)};     // raytracer.c:44-45
        drone->rotation = Vec3{0.1f * sinf(t), t, 0.1f * cosf(t)};                             // raytracer.c:46-47
    }
    if (treasure) {
        treasure->position = Vec3{1.0f, 0.5f + 0.1f * sinf(t), 1.0f};                          // raytracer.c:50-51
        treasure->rotation = Vec3{0, t * 0.5f, 0};                                             // raytracer.c:52-53
    }
}

int rt_host_scaled_size(const Scene* scene, int* scaled_width, int* scaled_height) {            // scene.c:164-165
    if (!scene) return RT_ERR_INVALID;
    return rt_scaled_size(scene->width, scene->height, scene->scale_factor, scaled_width, scaled_height);
}

int rt_host_upscale_frame(Scene* scene, int index, uint32_t* argb) {                             // scene.c:188-202
    if (!scene || !argb || !scene->frames || index < 0 || index >= scene->frame_count) return RT_ERR_INVALID;
    int sw = 0, sh = 0;
    int rc = rt_host_scaled_size(scene, &sw, &sh);
    if (rc != RT_OK) return rc;
    std::lock_guard<std::mutex> g(g_lock);
    SceneState* st = state_of(scene, true);
    GpuState& first = st->gpus[0];                   // the encoder's side: the first GPU of the scene's list
    if (!first.ctx) {
        rc = rt_ctx_create(first.device, &first.ctx);
        if (rc != RT_OK) { fprintf(stderr, "rt_host_upscale_frame: cannot open GPU %d: %s\n", first.device, rt_status_string(rc)); st->last_status = rc; return rc; }
    }
    rc = wait_frames(st);
    if (rc != RT_OK) return rc;
    rc = rt_upscale_frame_host(first.ctx, scene->frames[index], scene->width, scene->height, sw, sh, argb);
    if (rc != RT_OK) fprintf(stderr, "rt_host_upscale_frame: %s (%s)\n", rt_status_string(rc), rt_last_error(first.ctx));
    st->last_status = rc;
    return rc;
}

int rt_host_set_gpus(const int* devices, int n) {
    if (n < 0 || n > 64 || (n > 0 && !devices)) return RT_ERR_INVALID;
    const int have = rt_device_count();
    for (int i = 0; i < n; i++) if (devices[i] < 0 || devices[i] >= have) return RT_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> g(g_lock);
    g_gpu_list.assign(devices, devices + n);
    return RT_OK;
}

void rt_host_set_frames_in_flight(int n) { std::lock_guard<std::mutex> g(g_lock); g_in_flight = n < 1 ? 1 : (n > 8 ? 8 : n); }

int rt_host_wait(Scene* scene) {
    if (!scene) return RT_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_lock);
    return wait_frames(state_of(scene, false));
}

void rt_host_set_obj_mode(int tolerant) { g_obj_mode = tolerant ? 1 : 0; }

void rt_host_invalidate_scene(Scene* scene) {
    if (!scene) return;
    std::lock_guard<std::mutex> g(g_lock);
    SceneState* st = state_of(scene, false);
    if (st) for (GpuState& gpu : st->gpus) gpu.geometries.clear();      // the next render_scene uploads every mesh again
}

void rt_host_set_save_options(float quality, int threads) { g_save_quality = quality > 100.0f ? 100.0f : quality; g_save_threads = threads; }

int rt_host_write_ppm(const Scene* scene, int index, const char* filename) {
    if (!scene || !filename || index < 0 || index >= scene->frame_count) return RT_ERR_INVALID;
    const int rc = rt_host_wait(const_cast<Scene*>(scene));
    if (rc != RT_OK) return rc;
    FILE* f = fopen(filename, "wb");
    if (!f) return RT_ERR_INVALID;
    fprintf(f, "P6\n%d %d\n255\n", scene->width, scene->height);
    fwrite(scene->frames[index], 1, (size_t)scene->width * (size_t)scene->height * 3, f);
    fclose(f);
    return RT_OK;
}

}  // extern "C"
