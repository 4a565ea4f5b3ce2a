/*
 * ref_compose.c -- drives the COMPILED REFERENCE functions over a frame.
 *
 * TEST INFRASTRUCTURE ONLY.  Built only where /root/reference exists, into
 * oracle/_ref/libref.so together with the reference's own translation units
 * (math/vec3.c math/mat4.c math/ray.c geometry/aabb.c accel/bvh.c render/camera.c
 * render/light.c utils/image.c), which compile as they are.  scene.c and
 * geometry/mesh.c cannot be compiled in this image (they include <webp/...> headers
 * that are not installed), so the pixel loop around the reference calls
 * (scene.c:58-158) and the texel fetch (mesh.c:138-149) are written out here; every
 * geometric step -- camera ray, ray/normal transforms, BVH build and traversal --
 * is the reference's own object code.
 */
#include "accel/bvh.h"
#include "render/camera.h"
#include "render/light.h"
#include "utils/image.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    Triangle* tris;
    int n;
    BVH bvh;
    const uint8_t* rgba;      /* borrowed */
    int tw, th;
} RefMesh;

typedef struct {              /* same layout as ONode in rt_oracle.h */
    AABB bounds;
    int32_t left, right, start_idx, triangle_count;
} FlatNode;

RefMesh* refc_mesh_create(const Triangle* tris, int n, const uint8_t* rgba, int tw, int th) {
    RefMesh* m = (RefMesh*)calloc(1, sizeof(RefMesh));
    m->tris = (Triangle*)malloc(sizeof(Triangle) * (size_t)(n > 0 ? n : 1));
    memcpy(m->tris, tris, sizeof(Triangle) * (size_t)n);
    m->n = n;
    m->bvh = create_bvh(m->tris, (size_t)n);          /* reference builder: permutes m->tris */
    m->rgba = rgba; m->tw = tw; m->th = th;
    return m;
}

void refc_mesh_destroy(RefMesh* m) {
    if (!m) return;
    destroy_bvh(&m->bvh);
    free(m->tris);
    free(m);
}

void refc_mesh_triangles(const RefMesh* m, Triangle* out) { memcpy(out, m->tris, sizeof(Triangle) * (size_t)m->n); }

static int32_t flatten(const BVHNode* n, FlatNode* out, int32_t cap, int32_t* used) {
    if (*used >= cap) return -1;
    int32_t me = (*used)++;
    out[me].bounds = n->bounds;
    out[me].start_idx = n->start_idx;
    out[me].triangle_count = n->triangle_count;
    out[me].left = out[me].right = -1;
    if (n->left) { int32_t l = flatten(n->left, out, cap, used); if (l < 0) return -1; out[me].left = l; }
    if (n->right) { int32_t r = flatten(n->right, out, cap, used); if (r < 0) return -1; out[me].right = r; }
    return me;
}

/* The pointer tree written out node, left subtree, right subtree.  Returns node count. */
int32_t refc_mesh_nodes(const RefMesh* m, FlatNode* out, int32_t cap) {
    int32_t used = 0;
    if (flatten(m->bvh.root, out, cap, &used) < 0) return -1;
    return used;
}

/* Thin by-value wrappers so that ctypes callers need not pass large structs by value. */
void refc_camera_ray(const Camera* cam, float x, float y, float aspect, Ray* out) { *out = get_camera_ray(cam, x, y, aspect); }
void refc_transform_ray(const Ray* ray, const Transform* tf, Ray* out) { *out = transform_ray(*ray, *tf); }
void refc_transform_normal(const Vec3* n, const Transform* tf, Vec3* out) { *out = transform_normal(*n, *tf); }
int  refc_ray_aabb(const Ray* ray, const AABB* box) { return ray_aabb_intersect(*ray, *box) ? 1 : 0; }
int  refc_ray_triangle(const Ray* ray, const Vec3* v0, const Vec3* v1, const Vec3* v2, float* t, float* u, float* v) {
    return ray_triangle_intersect(*ray, *v0, *v1, *v2, t, u, v) ? 1 : 0;
}
int  refc_intersect(const RefMesh* m, const Ray* ray, float* t, float* u, float* v, int32_t* idx) {
    int i = 0;
    int h = intersect_bvh(m->bvh.root, *ray, m->tris, t, u, v, &i) ? 1 : 0;
    if (h) *idx = i;
    return h;
}
void refc_make_light(const Vec3* dir, const Vec3* color, DirectionalLight* out) { *out = create_directional_light(*dir, *color); }
uint32_t refc_bicubic(const uint8_t* frame, float x, float y, int w, int h) {
    return bicubic_interpolate((unsigned char*)frame, x, y, w, h);
}

/* The upscale loop of save_scene (scene.c:188-202) around the reference's bicubic_interpolate. */
void refc_upscale_frame(const uint8_t* frame, int width, int height, int scaled_width, int scaled_height,
                        uint32_t* argb, int threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(static) num_threads(threads)
    for (int y = 0; y < scaled_height; y++) {
        for (int x = 0; x < scaled_width; x++) {
            float src_x = x * (width - 1.0f) / (scaled_width - 1.0f);
            float src_y = y * (height - 1.0f) / (scaled_height - 1.0f);
            argb[(size_t)y * scaled_width + x] = bicubic_interpolate((unsigned char*)frame, src_x, src_y, width, height);
        }
    }
}

void refc_render_frame(RefMesh* const* meshes, const Transform* tfs, int mesh_count, const Camera* cam,
                       const DirectionalLight* light, int width, int height, uint8_t* rgb,
                       int32_t* hit_id, float* hit_t, uint8_t* shadow, int threads) {
    float aspect = (float)width / height;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            Ray ray = get_camera_ray(cam, (x + 0.5f) / width, (y + 0.5f) / height, aspect);
            float best = 1e30f;
            int who = -1, tri = -1;
            Vec2 uv = {0, 0};
            Vec3 nrm = {0, 0, 0};
            for (int m = 0; m < mesh_count; m++) {
                float t = best, u, v;
                int idx;
                Ray lr = transform_ray(ray, tfs[m]);
                if (intersect_bvh(meshes[m]->bvh.root, lr, meshes[m]->tris, &t, &u, &v, &idx) && t < best) {
                    const Triangle* T = &meshes[m]->tris[idx];
                    float w = 1.0f - u - v;
                    best = t; who = m; tri = idx;
                    uv.u = w * T->t0.u + u * T->t1.u + v * T->t2.u;
                    uv.v = w * T->t0.v + u * T->t1.v + v * T->t2.v;
                    nrm = vec3_normalize(vec3_add(vec3_add(vec3_mul(T->n0, w), vec3_mul(T->n1, u)), vec3_mul(T->n2, v)));
                    nrm = transform_normal(nrm, tfs[m]);
                }
            }
            size_t px = (size_t)y * width + x;
            uint8_t dark = 255;
            if (who >= 0) {
                const RefMesh* M = meshes[who];
                float fu = uv.u - floorf(uv.u), fv = uv.v - floorf(uv.v);
                int tx = (int)(fu * (M->tw - 1)), ty = (int)(fv * (M->th - 1));
                const uint8_t* texel = M->rgba + (size_t)(ty * M->tw + tx) * 4;
                Vec3 color = { texel[0] / 255.0f, texel[1] / 255.0f, texel[2] / 255.0f };
                float diffuse = 0.2f;
                Vec3 p = vec3_add(ray.origin, vec3_mul(ray.direction, best));
                Ray sr = { vec3_add(p, vec3_mul(nrm, 0.001f)), light->direction };
                dark = 0;
                for (int m = 0; m < mesh_count && !dark; m++) {
                    float st = 1e30f, su, sv;
                    int si;
                    Ray lr = transform_ray(sr, tfs[m]);
                    if (intersect_bvh(meshes[m]->bvh.root, lr, meshes[m]->tris, &st, &su, &sv, &si)) dark = 1;
                }
                if (!dark) diffuse = fmaxf(diffuse, vec3_dot(nrm, light->direction));
                color = vec3_mul(vec3_mul_vec3(color, light->color), diffuse);
                rgb[px * 3 + 0] = (unsigned char)(fminf(color.x * 255.0f, 255.0f));
                rgb[px * 3 + 1] = (unsigned char)(fminf(color.y * 255.0f, 255.0f));
                rgb[px * 3 + 2] = (unsigned char)(fminf(color.z * 255.0f, 255.0f));
            } else {
                rgb[px * 3 + 0] = rgb[px * 3 + 1] = rgb[px * 3 + 2] = 50;
            }
            if (hit_id) hit_id[px] = who >= 0 ? ((who << 24) | tri) : -1;
            if (hit_t) hit_t[px] = best;
            if (shadow) shadow[px] = dark;
        }
    }
}

int refc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
