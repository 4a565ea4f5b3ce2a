/*
 * raytracer_host.h -- the reference ray tracer's host API, kept name-for-name and
 * layout-for-layout, on top of the MI355X rendering core (include/rt_mi355.h).
 *
 * A program written against the reference's headers (scene.h, geometry/mesh.h, accel/bvh.h,
 * render/camera.h, render/light.h, utils/image.h) compiles against this one header and links
 * libraytracer_host.so instead of the reference's objects.  render_scene() runs the frame on
 * the GPU; there is no CPU rendering path behind it.
 *
 * Struct layouts equal the reference's (sizes on LP64 in brackets) because callers reach into
 * them directly, e.g. scene.meshes[0] in the reference driver (raytracer.c:44).
 */
#ifndef RAYTRACER_HOST_H
#define RAYTRACER_HOST_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } Vec3;                                   /* [12] math/vec3.h:6 */
typedef struct { float u, v; } Vec2;                                      /* [8]  math/vec3.h:7 */
typedef struct { float m[4][4]; } Mat4;                                   /* [64] math/mat4.h:6-8 */
typedef struct { Vec3 origin, direction; } Ray;                           /* [24] math/ray.h:8-10 */
typedef struct { Vec3 position, rotation; } Transform;                    /* [24] math/ray.h:12-15, radians about x,y,z */
typedef struct { Vec3 min, max; } AABB;                                   /* [24] geometry/aabb.h:9-12 */
typedef struct { Vec3 v0, v1, v2; Vec2 t0, t1, t2; Vec3 n0, n1, n2; } Triangle;   /* [96] geometry/triangle.h:6-10 */
typedef struct BVHNode {                                                   /* [48] accel/bvh.h:8-14 */
    AABB bounds;
    struct BVHNode* left;
    struct BVHNode* right;
    int start_idx, triangle_count;
} BVHNode;
typedef struct { BVHNode* root; Triangle* triangles; size_t triangle_count; } BVH;          /* accel/bvh.h:16-20 */
typedef struct {                                                           /* geometry/mesh.h:11-19 */
    Triangle* triangles;
    size_t triangle_count;
    unsigned char* texture_data;       /* RGBA8 rows as decoded from the WebP file */
    int texture_width, texture_height;
    BVH bvh;
    Transform transform;
} Mesh;
typedef struct { Vec3 position, look_at, up; float fov; } Camera;         /* [40] render/camera.h:7-12 */
typedef struct { Vec3 direction, color; } DirectionalLight;               /* [24] render/light.h:6-9 */
typedef struct {                                                           /* scene.h:13-26 */
    Mesh* meshes;
    size_t mesh_count;
    Camera camera;
    DirectionalLight light;
    unsigned char** frames;            /* frame_count buffers of RGB8, row-major */
    int frame_count, current_frame;
    int width, height;
    float scale_factor;
    int duration_ms, fps;
} Scene;

/* scene.h:29-36 */
Scene create_scene(int width, int height, int duration_ms, int fps, float scale_factor);
void add_mesh_to_scene(Scene* scene, Mesh mesh);
void set_scene_camera(Scene* scene, Vec3 position, Vec3 look_at, Vec3 up, float fov);
void set_scene_light(Scene* scene, Vec3 direction, Vec3 color);
void next_frame(Scene* scene);
void render_scene(Scene* scene);       /* GPU; on failure prints to stderr and leaves the frame untouched */
void save_scene(Scene* scene, const char* filename);
void destroy_scene(Scene* scene);

/* geometry/mesh.h:22-26 */
Mesh create_mesh(const char* obj_filename, const char* texture_filename);
void set_mesh_position(Mesh* mesh, Vec3 position);
void set_mesh_rotation(Mesh* mesh, Vec3 rotation);
void destroy_mesh(Mesh* mesh);
Vec3 sample_mesh_texture(const Mesh* mesh, float u, float v);

/* accel/bvh.h:23-28 */
BVHNode* create_bvh_node(Triangle* triangles, int start, int count);
BVH create_bvh(Triangle* triangles, size_t count);
void destroy_bvh_node(BVHNode* node);
void destroy_bvh(BVH* bvh);
/* One ray against one mesh on the HOST (the reference's CPU traversal, for programs that call it directly;
 * render_scene does not use it). */
bool intersect_bvh(BVHNode* node, Ray ray, const Triangle* triangles, float* t_out, float* u_out, float* v_out, int* tri_idx);

/* render/camera.h:15, render/light.h:12 */
Camera create_camera(Vec3 position, Vec3 look_at, Vec3 up, float fov);
DirectionalLight create_directional_light(Vec3 direction, Vec3 color);

/* Value-type helpers of math/vec3.h:10-18, math/mat4.h:11-20, math/ray.h:18-21, geometry/aabb.h:15-18,
 * render/camera.h:16 and utils/image.h:7-9, with the reference's float results.  Host utilities only:
 * the frame path does not use them. */
Vec3 vec3_add(Vec3 a, Vec3 b);
Vec3 vec3_sub(Vec3 a, Vec3 b);
Vec3 vec3_mul(Vec3 a, float t);
Vec3 vec3_div(Vec3 a, float t);
float vec3_dot(Vec3 a, Vec3 b);
Vec3 vec3_cross(Vec3 a, Vec3 b);
float vec3_length(Vec3 v);
Vec3 vec3_normalize(Vec3 v);
Vec3 vec3_mul_vec3(Vec3 a, Vec3 b);
Mat4 mat4_identity(void);
Mat4 mat4_multiply(Mat4 a, Mat4 b);
Mat4 mat4_translation(Vec3 t);
Mat4 mat4_rotation_x(float angle);
Mat4 mat4_rotation_y(float angle);
Mat4 mat4_rotation_z(float angle);
Vec3 mat4_transform_point(Mat4 m, Vec3 p);
Vec3 mat4_transform_vector(Mat4 m, Vec3 v);
Mat4 mat4_inverse(Mat4 m);
Mat4 mat4_transpose(Mat4 m);
Ray transform_ray(Ray ray, Transform transform);
Vec3 transform_normal(Vec3 normal, Transform transform);
bool ray_triangle_intersect(Ray ray, Vec3 v0, Vec3 v1, Vec3 v2, float* t, float* u_out, float* v_out);
AABB create_empty_aabb(void);
AABB expand_aabb(AABB box, Vec3 point);
AABB get_triangle_bounds(Triangle tri);
bool ray_aabb_intersect(Ray ray, AABB box);
Ray get_camera_ray(const Camera* camera, float x, float y, float aspect);
float cubic_hermite(float A, float B, float C, float D, float t);
uint32_t get_pixel_rgb(unsigned char* frame, int x, int y, int width, int height);
uint32_t bicubic_interpolate(unsigned char* frame, float x, float y, int width, int height);

/* ---- additions that the reference does not have ---- */
/* GPU used by scenes created afterwards on this thread's process (default 0, or $RT_DEVICE). */
void rt_host_set_device(int device);
/* Scenes created afterwards deal their frames over several GPUs: render_scene of frame f (scene->current_frame) runs on
 * devices[f mod n], each with its own copy of the scene's meshes, and writes scene->frames[f] as always.  Frames are
 * independent (raytracer.c:40-57), so with rt_host_set_frames_in_flight(k >= 2) the GPUs work side by side; save_scene
 * and rt_host_upscale_frame use devices[0].  A device may be listed more than once.  n = 0 returns to the single device
 * of rt_host_set_device.  Returns 0 or a negative rt_status (unknown device: RT_ERR_NO_DEVICE). */
int rt_host_set_gpus(const int* devices, int n);
/* Status of the most recent render_scene() on this scene: 0 or a negative rt_status. */
int rt_host_last_status(const Scene* scene);
/* Debug planes of the most recent render_scene(): any pointer may be NULL.
 * hit_id = (mesh<<24)|triangle or -1; hit_t = closest t; shadow = 1 blocked / 0 lit / 255 miss.
 * Re-renders the current frame with those planes enabled. */
int rt_host_render_debug(Scene* scene, int32_t* hit_id, float* hit_t, uint8_t* shadow);
/* Placements of the reference driver's animation at `frame` (raytracer.c:41-53): meshes[0]
 * (drone) and meshes[1] (treasure); computed with the same float/double mix as the driver. */
void rt_host_demo_animation(int frame, Transform* drone, Transform* treasure);
/* Writes frame `index` as a binary PPM (P6). */
int rt_host_write_ppm(const Scene* scene, int index, const char* filename);
/* OBJ reader of create_mesh.  0 (default): what the reference reads (mesh.c:41-76) -- "f v/vt/vn v/vt/vn v/vt/vn",
 * corners after the third ignored, faces it cannot use skipped with a message instead of read out of bounds.
 * 1: tolerant -- faces with more than three corners become a triangle fan, "v", "v/vt" and "v//vn" corners are
 * accepted (texture coordinate 0,0 / geometric normal where missing), negative indices count from the end.
 * In both modes lines and element counts are unbounded. */
void rt_host_set_obj_mode(int tolerant);
/* render_scene keeps the scene's meshes on the GPU and uploads a mesh again when its triangle array, texture or BVH
 * is another one (a BVH rebuilt by create_bvh counts as another one even at the same address).  A caller that
 * changes triangles or texels IN PLACE without rebuilding the BVH calls this before the next render_scene. */
void rt_host_invalidate_scene(Scene* scene);
/* The size and the pixels save_scene hands to its encoder for one frame (scene.c:164-165, 188-202):
 * argb receives scaled_width*scaled_height words, 0xFF000000 | r<<16 | g<<8 | b, upscaled on the GPU. */
int rt_host_scaled_size(const Scene* scene, int* scaled_width, int* scaled_height);
int rt_host_upscale_frame(Scene* scene, int index, uint32_t* argb);
/* Animation file writer behind save_scene (scene.c:164-224), usable on its own for frames that arrive
 * one by one (e.g. from other GPUs).  Frames are encoded as full-canvas key frames on `threads` host
 * threads (0: one per core, at most 16) while the caller produces the next ones, and appended in order.
 * quality in [0,100] is libwebp's lossy quality factor (WebPConfigInit's default is 75), negative = lossless.
 * add() copies the frame and returns; format 0 = RGB8 (the layout of Scene.frames), 1 = ARGB32 words
 * (the layout of WebPPicture.argb).  Without a WebP encoder in the process (rt_host_animation_has_webp() == 0)
 * the same calls write concatenated binary PPM images to `filename` + ".ppm". */
typedef struct rt_host_animation rt_host_animation;
int rt_host_animation_has_webp(void);
int rt_host_animation_open(const char* filename, int width, int height, float quality, int threads, rt_host_animation** out);
int rt_host_animation_add(rt_host_animation* a, const void* pixels, int format, int duration_ms);
int rt_host_animation_close(rt_host_animation* a);
/* Frames render_scene may leave in flight (1..8; default 1 = every call returns with its frame in
 * scene->frames[current_frame], like the reference).  With n > 1 render_scene queues the frame's kernel and
 * its copy into scene->frames[current_frame] and returns; the frame is complete after rt_host_wait(scene),
 * and save_scene, rt_host_upscale_frame, rt_host_write_ppm, rt_host_render_debug and destroy_scene wait by
 * themselves.  A driver that only renders and then saves (raytracer.c:36-63) needs no other change.
 * Applies to scenes of this process from their next render_scene on. */
void rt_host_set_frames_in_flight(int n);
/* Waits until every frame render_scene has queued for this scene is in scene->frames.  Returns 0 or a negative rt_status. */
int rt_host_wait(Scene* scene);
/* Encoder settings save_scene uses: quality as above (default 75), threads as above (default 0). */
void rt_host_set_save_options(float quality, int threads);

#ifdef __cplusplus
}
#endif
#endif
