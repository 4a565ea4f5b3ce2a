// raytracer_math.cpp -- the small value-type functions of the reference's headers (math/vec3.h,
// math/mat4.h, math/ray.h, geometry/aabb.h, render/camera.h, utils/image.h), so that host programs
// written against them keep linking.  They are built from the same float building blocks the
// device code uses (csrc/rt_exact.h, csrc/rt_bicubic.h; FMA contraction off), so their results are
// the ones the GPU path is tested against.  None of them is on the frame path: render_scene() does
// not call them; intersect_bvh is a one-ray utility for programs written against accel/bvh.h.
#include <math.h>
#include <string.h>

#include "../csrc/rt_bicubic.h"
#include "../csrc/rt_exact.h"
#include "raytracer_host.h"

namespace {
inline rt::F3 in(Vec3 v) { return rt::f3(v.x, v.y, v.z); }
inline Vec3 out(rt::F3 v) { Vec3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }
inline rt::M4 in(const Mat4& m) { rt::M4 r; memcpy(r.m, m.m, sizeof r.m); return r; }
inline Mat4 out(const rt::M4& m) { Mat4 r; memcpy(r.m, m.m, sizeof r.m); return r; }
inline rt::M4 rotation_of(Transform tf) {                                             // ray.c:7-9
    return rt::m4_mul(rt::m4_rotation(2, tf.rotation.z), rt::m4_mul(rt::m4_rotation(1, tf.rotation.y), rt::m4_rotation(0, tf.rotation.x)));
}

// Result of the reference's traversal below one node (accel/bvh.c:80-129).
struct Found {
    bool hit;
    float t, u, v;
    int idx;
};

// A leaf keeps the closest accepted triangle below `bound` (strict, so the lowest index wins equal distances);
// an inner node asks both children with a fresh bound of 1e30f and keeps the left answer only when it is
// strictly closer than the right one.  The caller's bound therefore matters in a root that is a leaf only.
Found walk(const BVHNode* node, rt::F3 o, rt::F3 d, rt::F3 inv, const Triangle* tris, float bound) {
    Found f = {false, 0.0f, 0.0f, 0.0f, -1};
    const float lo[3] = {node->bounds.min.x, node->bounds.min.y, node->bounds.min.z};
    const float hi[3] = {node->bounds.max.x, node->bounds.max.y, node->bounds.max.z};
    if (!rt::slab(o, inv, lo, hi)) return f;
    if (!node->left && !node->right) {
        for (int i = 0; i < node->triangle_count; i++) {
            const Triangle& T = tris[node->start_idx + i];
            const rt::F3 a = in(T.v0);
            float t, u, v;
            if (rt::moller_trumbore(o, d, a, rt::sub(in(T.v1), a), rt::sub(in(T.v2), a), t, u, v) && t < bound) {
                bound = t;
                f.hit = true; f.t = t; f.u = u; f.v = v; f.idx = node->start_idx + i;
            }
        }
        return f;
    }
    Found l = f, r = f;
    if (node->left) l = walk(node->left, o, d, inv, tris, 1e30f);
    if (node->right) r = walk(node->right, o, d, inv, tris, 1e30f);
    if (l.hit && (!r.hit || l.t < r.t)) return l;
    return r;
}
}  // namespace

extern "C" {

// ---- accel/bvh.h:27-28 ----
// One ray against one mesh's tree on the host, as accel/bvh.c:80-129 does it: every box whose ancestors' boxes
// the ray passes is visited, the reciprocal direction is the one ray_aabb_intersect derives per call.  Outputs
// are written on a hit only.
bool intersect_bvh(BVHNode* node, Ray ray, const Triangle* triangles, float* t_out, float* u_out, float* v_out, int* tri_idx) {
    if (!node || !t_out) return false;
    const rt::F3 d = in(ray.direction);
    const Found f = walk(node, in(ray.origin), d, rt::recip(d), triangles, *t_out);
    if (!f.hit) return false;
    *t_out = f.t;
    if (u_out) *u_out = f.u;
    if (v_out) *v_out = f.v;
    if (tri_idx) *tri_idx = f.idx;
    return true;
}

// ---- math/vec3.h:10-18 ----
Vec3 vec3_add(Vec3 a, Vec3 b) { return out(rt::add(in(a), in(b))); }
Vec3 vec3_sub(Vec3 a, Vec3 b) { return out(rt::sub(in(a), in(b))); }
Vec3 vec3_mul(Vec3 a, float t) { return out(rt::scale(in(a), t)); }
Vec3 vec3_div(Vec3 a, float t) { return out(rt::scale(in(a), 1.0f / t)); }            // vec3.c:15: a * (1/t)
float vec3_dot(Vec3 a, Vec3 b) { return rt::dot(in(a), in(b)); }
Vec3 vec3_cross(Vec3 a, Vec3 b) { return out(rt::cross(in(a), in(b))); }
float vec3_length(Vec3 v) { return sqrtf(rt::dot(in(v), in(v))); }
Vec3 vec3_normalize(Vec3 v) { return out(rt::unit(in(v))); }
Vec3 vec3_mul_vec3(Vec3 a, Vec3 b) { return out(rt::mul(in(a), in(b))); }

// ---- math/mat4.h:11-20 ----
Mat4 mat4_identity(void) { return out(rt::m4_identity()); }
Mat4 mat4_multiply(Mat4 a, Mat4 b) { return out(rt::m4_mul(in(a), in(b))); }
Mat4 mat4_translation(Vec3 t) { return out(rt::m4_translation(in(t))); }
Mat4 mat4_rotation_x(float angle) { return out(rt::m4_rotation(0, angle)); }
Mat4 mat4_rotation_y(float angle) { return out(rt::m4_rotation(1, angle)); }
Mat4 mat4_rotation_z(float angle) { return out(rt::m4_rotation(2, angle)); }
Vec3 mat4_transform_point(Mat4 m, Vec3 p) { return out(rt::xform_point(m.m, m.m[3], in(p))); }
Vec3 mat4_transform_vector(Mat4 m, Vec3 v) { return out(rt::xform_vector<4>(m.m, in(v))); }
Mat4 mat4_inverse(Mat4 m) { return out(rt::m4_inverse(in(m))); }
Mat4 mat4_transpose(Mat4 m) { return out(rt::m4_transpose(in(m))); }

// ---- math/ray.h:18-21 ----
Ray transform_ray(Ray ray, Transform transform) {                                     // ray.c:3-23
    const rt::M4 inv = rt::m4_inverse(rt::m4_mul(rt::m4_translation(in(transform.position)), rotation_of(transform)));
    Ray r;
    r.origin = out(rt::xform_point(inv.m, inv.m[3], in(ray.origin)));
    r.direction = out(rt::unit(rt::xform_vector<4>(inv.m, in(ray.direction))));
    return r;
}
Vec3 transform_normal(Vec3 normal, Transform transform) {                             // ray.c:25-40
    const rt::M4 nrm = rt::m4_transpose(rt::m4_inverse(rotation_of(transform)));
    return out(rt::unit(rt::xform_vector<4>(nrm.m, in(normal))));
}
bool ray_triangle_intersect(Ray ray, Vec3 v0, Vec3 v1, Vec3 v2, float* t, float* u_out, float* v_out) {   // ray.c:42-67
    float tt = 0.0f, uu = 0.0f, vv = 0.0f;
    const rt::F3 a = in(v0);
    const bool hit = rt::moller_trumbore(in(ray.origin), in(ray.direction), a, rt::sub(in(v1), a), rt::sub(in(v2), a), tt, uu, vv);
    if (hit) { *t = tt; *u_out = uu; *v_out = vv; }
    return hit;
}

// ---- geometry/aabb.h:15-18 ----
AABB create_empty_aabb(void) {                                                        // aabb.c:4-9
    AABB b;
    b.min.x = b.min.y = b.min.z = 1e30f;
    b.max.x = b.max.y = b.max.z = -1e30f;
    return b;
}
AABB expand_aabb(AABB box, Vec3 p) {                                                  // aabb.c:11-19
    box.min.x = rt::min_later(box.min.x, p.x); box.min.y = rt::min_later(box.min.y, p.y); box.min.z = rt::min_later(box.min.z, p.z);
    box.max.x = rt::max_later(box.max.x, p.x); box.max.y = rt::max_later(box.max.y, p.y); box.max.z = rt::max_later(box.max.z, p.z);
    return box;
}
AABB get_triangle_bounds(Triangle tri) {                                              // aabb.c:21-26
    return expand_aabb(expand_aabb(expand_aabb(create_empty_aabb(), tri.v0), tri.v1), tri.v2);
}
bool ray_aabb_intersect(Ray ray, AABB box) {                                          // aabb.c:28-56
    const float lo[3] = {box.min.x, box.min.y, box.min.z}, hi[3] = {box.max.x, box.max.y, box.max.z};
    return rt::slab(in(ray.origin), rt::recip(in(ray.direction)), lo, hi);
}

// ---- render/camera.h:16 ----
Ray get_camera_ray(const Camera* camera, float x, float y, float aspect) {            // camera.c:12-30
    const rt::F3 fwd = rt::unit(rt::sub(in(camera->look_at), in(camera->position)));
    const rt::F3 right = rt::unit(rt::cross(fwd, in(camera->up)));
    const rt::F3 up = rt::cross(right, fwd);
    const float scale = tanf((float)((double)(camera->fov * 0.5f) * 3.14159265358979323846 / (double)180.0f));   // camera.c:17
    const float rx = (2.0f * x - 1.0f) * aspect * scale;
    const float ry = (1.0f - 2.0f * y) * scale;
    Ray r;
    r.origin = camera->position;
    r.direction = out(rt::unit(rt::add(rt::add(rt::scale(right, rx), rt::scale(up, ry)), fwd)));
    return r;
}

// ---- utils/image.h:7-9 ----
float cubic_hermite(float A, float B, float C, float D, float t) { return rt::cubic_hermite(A, B, C, D, t); }   // image.c:4-11
uint32_t get_pixel_rgb(unsigned char* frame, int x, int y, int width, int height) {   // image.c:14-23
    const unsigned char* p = frame + ((size_t)rt::clamp_index(y, height) * (size_t)width + (size_t)rt::clamp_index(x, width)) * 3;
    return ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[2];
}
uint32_t bicubic_interpolate(unsigned char* frame, float x, float y, int width, int height) {   // image.c:25-78
    const int x1 = (int)x, y1 = (int)y;
    const float fx = x - (float)x1, fy = y - (float)y1;
    int off[4];
    for (int k = 0; k < 4; k++) off[k] = rt::clamp_index(x1 - 1 + k, width) * 3;
    rt::Rgb3 w[4];
    for (int k = 0; k < 4; k++) w[k] = rt::row_pass(frame + (size_t)rt::clamp_index(y1 - 1 + k, height) * (size_t)width * 3, off, fx);
    return rt::column_pass(w, fy);
}

}  // extern "C"
