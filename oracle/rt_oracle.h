/*
 * rt_oracle.h -- CPU restatement of the reference ray tracer's per-pixel ray/scene loop.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker that the HIP path is compared
 * against; it is never linked, loaded or called by the product library.  Only tests,
 * the smoke check and the benchmark's cpu_baseline leg may use it.
 *
 * Every function restates one piece of /root/reference (file:line given at each
 * definition in rt_oracle.c) in plain C with IEEE float32 arithmetic and no FMA
 * contraction (the library must be compiled with -ffp-contract=off).
 *
 * Parity pin: the functions that the reference can be compiled for in isolation
 * (vec3/mat4/ray/aabb/bvh/camera/light/image) are checked bit-for-bit against the
 * compiled reference (oracle/_ref/libref.so) and against fixtures generated from it.
 * The per-pixel glue of render_scene (scene.c:52-160), sample_mesh_texture
 * (mesh.c:138-149) and the OBJ reader (mesh.c:41-76) live in translation units that
 * need <webp/...> headers this image does not have, so those three pieces are
 * restated here and pinned only through composition with the compiled reference
 * functions (oracle/ref_compose.c), not by a run of the reference's own loop.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } OVec3;                  /* math/vec3.h:6 */
typedef struct { float u, v; } OVec2;                     /* math/vec3.h:7 */
typedef struct { OVec3 origin, direction; } ORay;         /* math/ray.h:8-10 */
typedef struct { OVec3 position, rotation; } OTransform;  /* math/ray.h:12-15 */
typedef struct { OVec3 min, max; } OAabb;                 /* geometry/aabb.h:9-12 */
typedef struct { OVec3 position, look_at, up; float fov; } OCamera;   /* render/camera.h:7-12 */
typedef struct { OVec3 direction, color; } OLight;        /* render/light.h:6-9 */
typedef struct {                                          /* geometry/triangle.h:6-10 (96 B) */
    OVec3 v0, v1, v2;
    OVec2 t0, t1, t2;
    OVec3 n0, n1, n2;
} OTriangle;

/* Index-linked image of the reference's heap node (accel/bvh.h:8-14): children are
 * indices into the node array (-1 = NULL); nodes are stored in creation order, i.e.
 * a node, then its whole left subtree, then its whole right subtree. */
typedef struct {
    OAabb bounds;
    int32_t left, right;
    int32_t start_idx, triangle_count;
} ONode;

typedef struct {                                          /* geometry/mesh.h:11-19, flattened */
    const OTriangle* triangles;
    int32_t triangle_count;
    const ONode* nodes;                                    /* nodes[0] is the root */
    int32_t node_count;
    const uint8_t* texture_rgba;                           /* WebPDecodeRGBA layout, row-major RGBA8 */
    int32_t texture_width, texture_height;
    OTransform transform;
} OMesh;

typedef struct {                                           /* work counters (all rays of a frame) */
    uint64_t primary_rays, shadow_rays;
    uint64_t slab_tests, tri_tests;
    uint64_t hit_pixels, shadowed_pixels;
} OCounters;

/* ---- unit-level restatements ---- */
ORay   orc_camera_ray(const OCamera* cam, float x, float y, float aspect);
ORay   orc_transform_ray(ORay ray, OTransform tf);
OVec3  orc_transform_normal(OVec3 n, OTransform tf);
int    orc_ray_aabb(ORay ray, OAabb box);
int    orc_ray_triangle(ORay ray, OVec3 v0, OVec3 v1, OVec3 v2, float* t, float* u, float* v);
int    orc_intersect_bvh(const ONode* nodes, int32_t node, ORay ray, const OTriangle* tris,
                         float* t, float* u, float* v, int32_t* tri_idx);
OLight orc_make_light(OVec3 direction, OVec3 color);
void   orc_sample_texture(const OMesh* mesh, float u, float v, float rgb[3]);
uint32_t orc_bicubic(const uint8_t* frame, float x, float y, int width, int height);
void   orc_scaled_size(int width, int height, float scale_factor, int* scaled_width, int* scaled_height);
void   orc_upscale_frame(const uint8_t* frame, int width, int height, int scaled_width, int scaled_height,
                         uint32_t* argb, int32_t threads);

/* Per-frame constants of one mesh placement, exactly as transform_ray / transform_normal
 * derive them for every ray: inv = inverse(T*Rz*Ry*Rx), nrm = transpose(inverse(Rz*Ry*Rx)). */
void   orc_mesh_matrices(OTransform tf, float inv[16], float nrm[16]);

/* ---- builders / loaders ---- */
/* Reorders tris in place exactly as the reference builder does and writes the node array.
 * Returns the node count, or -1 if cap is too small. */
int32_t orc_build_bvh(OTriangle* tris, int32_t count, ONode* nodes_out, int32_t cap);
/* Reads v/vt/vn/f lines the way the reference does.  Returns triangle count (<0 on error). */
int32_t orc_load_obj(const char* path, OTriangle* tris_out, int32_t cap,
                     int32_t* n_v, int32_t* n_vt, int32_t* n_vn);

/* ---- the frame loop ---- */
/* rgb: width*height*3 bytes.  Optional per-pixel debug planes (may be NULL):
 *   hit_id : (mesh<<24)|triangle index of the primary hit, -1 for a miss
 *   hit_t  : closest_t (1e30f for a miss)
 *   shadow : 1 if the shadow ray was blocked, 0 if lit, 255 if no primary hit
 * counters may be NULL.  threads<=0 uses the OpenMP default. */
void   orc_render_frame(const OMesh* meshes, int32_t mesh_count, const OCamera* cam,
                        const OLight* light, int32_t width, int32_t height,
                        uint8_t* rgb, int32_t* hit_id, float* hit_t, uint8_t* shadow,
                        OCounters* counters, int32_t threads);

/* The same frame with the hit reported as two planes, list index of the mesh and triangle index (-1 = miss), for
 * mesh lists too long for the packed id. */
void   orc_render_frame_ids(const OMesh* meshes, int32_t mesh_count, const OCamera* cam,
                            const OLight* light, int32_t width, int32_t height,
                            uint8_t* rgb, int32_t* hit_mesh, int32_t* hit_tri, float* hit_t, uint8_t* shadow,
                            OCounters* counters, int32_t threads);

/* Animation of the reference's driver (raytracer.c:41-53): placements of the drone
 * (mesh 0) and the treasure (mesh 1) at a frame; the ground keeps the identity. */
void   orc_animation_transforms(int32_t frame, OTransform out[3]);

/* Sampled frames: several camera rays per pixel, several lights (no counterpart in the reference). */
void   orc_sample_offset(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float* jx, float* jy);
void   orc_render_frame_sampled(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* lights,
                                int32_t n_lights, int32_t spp, uint32_t seed, int32_t width, int32_t height,
                                uint8_t* rgb, int32_t* hit_id, OCounters* counters, int32_t threads);
void   orc_render_window_sampled(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* lights,
                                 int32_t n_lights, int32_t spp, uint32_t seed, int32_t width, int32_t height,
                                 int32_t x0, int32_t y0, int32_t win_w, int32_t win_h,
                                 uint8_t* rgb, int32_t* hit_id, OCounters* counters, int32_t threads);
int    orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
