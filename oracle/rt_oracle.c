/*
 * rt_oracle.c -- CPU restatement of the reference's per-pixel ray/scene loop.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Compile with -ffp-contract=off: the
 * canonical arithmetic is IEEE float32 with one rounding per operation, evaluated in
 * the same order as the reference's C expressions.  All citations are to files under
 * /root/reference.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vec3 (math/vec3.c:3-41) */
static inline OVec3 v3(float x, float y, float z) { OVec3 r = {x, y, z}; return r; }
static inline OVec3 v_add(OVec3 a, OVec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* vec3.c:3 */
static inline OVec3 v_sub(OVec3 a, OVec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* vec3.c:7 */
static inline OVec3 v_scale(OVec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }          /* vec3.c:11 */
static inline float v_dot(OVec3 a, OVec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }        /* vec3.c:19 */
static inline OVec3 v_cross(OVec3 a, OVec3 b) {                                                  /* vec3.c:23 */
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec3.c:15,31,35: normalise multiplies by the reciprocal of the length. */
static inline OVec3 v_unit(OVec3 a) { return v_scale(a, 1.0f / sqrtf(v_dot(a, a))); }
static inline OVec3 v_mul(OVec3 a, OVec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }      /* vec3.c:39 */

/* ------------------------------------------------------------------ mat4 (math/mat4.c) */
typedef struct { float m[4][4]; } M4;

static M4 m_identity(void) {                                   /* mat4.c:3-7 */
    M4 r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
    return r;
}

static M4 m_mul(M4 a, M4 b) {                                  /* mat4.c:9-19: acc starts at 0.0f, k ascending */
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc += a.m[i][k] * b.m[k][j];
            r.m[i][j] = acc;
        }
    return r;
}

/* mat4.c:29-60.  axis 0/1/2 = x/y/z; the sine sits at (p,q) negated and (q,p) plain
 * for x and z, and with the signs exchanged for y. */
static M4 m_rotation(int axis, float angle) {
    M4 r = m_identity();
    float c = cosf(angle), s = sinf(angle);
    if (axis == 0)      { r.m[1][1] = c; r.m[1][2] = -s; r.m[2][1] = s;  r.m[2][2] = c; }
    else if (axis == 1) { r.m[0][0] = c; r.m[0][2] = s;  r.m[2][0] = -s; r.m[2][2] = c; }
    else                { r.m[0][0] = c; r.m[0][1] = -s; r.m[1][0] = s;  r.m[1][1] = c; }
    return r;
}

static M4 m_translation(OVec3 t) {                             /* mat4.c:21-27 */
    M4 r = m_identity();
    r.m[0][3] = t.x; r.m[1][3] = t.y; r.m[2][3] = t.z;
    return r;
}

/* mat4.c:79-214.  The reference spells out all 16 cofactors; they follow one pattern.
 * For entry (i,j): R = the three rows != j and C = the three columns != i, ascending;
 * the six products, in this order and each evaluated (a*b)*c, are
 *   +R0C0 R1C1 R2C2  -R0C0 R1C2 R2C1  -R1C0 R0C1 R2C2  +R1C0 R0C2 R2C1  +R2C0 R0C1 R1C2  -R2C0 R0C2 R1C1
 * all multiplied by (-1)^(i+j) and summed left to right.  A leading unary minus in the
 * reference negates the first factor, which is exact, so sign*product reproduces it. */
static M4 m_inverse(M4 a) {
    static const int ra[6] = {0, 0, 1, 1, 2, 2};                                      /* row of 1st factor (col is C0) */
    static const int rb[6] = {1, 1, 0, 0, 0, 0}, rc[6] = {2, 2, 2, 2, 1, 1};          /* rows of 2nd, 3rd factor */
    static const int cb[6] = {1, 2, 1, 2, 1, 2}, cc[6] = {2, 1, 2, 1, 2, 1};          /* cols of 2nd, 3rd factor */
    static const float ps[6] = {1.0f, -1.0f, -1.0f, 1.0f, 1.0f, -1.0f};
    M4 cof;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int R[3], C[3], nr = 0, nc = 0;
            for (int k = 0; k < 4; k++) { if (k != j) R[nr++] = k; if (k != i) C[nc++] = k; }
            float sign = ((i + j) & 1) ? -1.0f : 1.0f;
            float acc = 0.0f;
            for (int k = 0; k < 6; k++) {
                float p = (a.m[R[ra[k]]][C[0]] * a.m[R[rb[k]]][C[cb[k]]]) * a.m[R[rc[k]]][C[cc[k]]];
                float term = (sign * ps[k]) * p;               /* exact: multiply by +-1 */
                acc = (k == 0) ? term : acc + term;
            }
            cof.m[i][j] = acc;
        }
    float det = a.m[0][0] * cof.m[0][0] + a.m[0][1] * cof.m[1][0] + a.m[0][2] * cof.m[2][0] + a.m[0][3] * cof.m[3][0];
    if (det == 0) return m_identity();                         /* mat4.c:199-202 */
    det = 1.0f / det;
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = cof.m[i][j] * det;
    return r;
}

static M4 m_transpose(M4 a) {                                  /* mat4.c:216-224 */
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
    return r;
}

static OVec3 m_point(M4 m, OVec3 p) {                          /* mat4.c:62-69: divides by w */
    float w = m.m[3][0] * p.x + m.m[3][1] * p.y + m.m[3][2] * p.z + m.m[3][3];
    return v3((m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3]) / w,
              (m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3]) / w,
              (m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3]) / w);
}

static OVec3 m_vector(M4 m, OVec3 v) {                         /* mat4.c:71-77 */
    return v3(m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z,
              m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
              m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z);
}

static M4 rotation_of(OTransform tf) {                         /* ray.c:5-7,12-13 / ray.c:27-33: Rz*(Ry*Rx) */
    return m_mul(m_rotation(2, tf.rotation.z), m_mul(m_rotation(1, tf.rotation.y), m_rotation(0, tf.rotation.x)));
}

void orc_mesh_matrices(OTransform tf, float inv[16], float nrm[16]) {
    M4 rot = rotation_of(tf);
    M4 to_local = m_inverse(m_mul(m_translation(tf.position), rot));   /* ray.c:11-16 */
    M4 n = m_transpose(m_inverse(rot));                                /* ray.c:37 */
    memcpy(inv, to_local.m, sizeof(float) * 16);
    memcpy(nrm, n.m, sizeof(float) * 16);
}

/* ------------------------------------------------------------------ rays */
ORay orc_transform_ray(ORay ray, OTransform tf) {              /* ray.c:3-23 */
    M4 to_local = m_inverse(m_mul(m_translation(tf.position), rotation_of(tf)));
    ORay r;
    r.origin = m_point(to_local, ray.origin);
    r.direction = v_unit(m_vector(to_local, ray.direction));
    return r;
}

OVec3 orc_transform_normal(OVec3 n, OTransform tf) {           /* ray.c:25-40 */
    return v_unit(m_vector(m_transpose(m_inverse(rotation_of(tf))), n));
}

int orc_ray_triangle(ORay ray, OVec3 v0, OVec3 v1, OVec3 v2, float* t, float* u_out, float* v_out) {   /* ray.c:42-67 */
    const float eps = 0.0000001f;
    OVec3 e1 = v_sub(v1, v0), e2 = v_sub(v2, v0);
    OVec3 h = v_cross(ray.direction, e2);
    float a = v_dot(e1, h);
    if (a > -eps && a < eps) return 0;
    float f = 1.0f / a;
    OVec3 s = v_sub(ray.origin, v0);
    float u = f * v_dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    OVec3 q = v_cross(s, e1);
    float v = f * v_dot(ray.direction, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    *t = f * v_dot(e2, q);                                     /* written even when rejected below (ray.c:63-66) */
    *u_out = u;
    *v_out = v;
    return *t > eps;
}

int orc_ray_aabb(ORay ray, OAabb box) {                        /* aabb.c:33-56 */
    float ix = 1.0f / ray.direction.x, iy = 1.0f / ray.direction.y, iz = 1.0f / ray.direction.z;
    float a = (box.min.x - ray.origin.x) * ix, b = (box.max.x - ray.origin.x) * ix;
    float tmin = fminf(a, b), tmax = fmaxf(a, b);
    a = (box.min.y - ray.origin.y) * iy; b = (box.max.y - ray.origin.y) * iy;
    tmin = fmaxf(tmin, fminf(a, b)); tmax = fminf(tmax, fmaxf(a, b));
    a = (box.min.z - ray.origin.z) * iz; b = (box.max.z - ray.origin.z) * iz;
    tmin = fmaxf(tmin, fminf(a, b)); tmax = fminf(tmax, fmaxf(a, b));
    return tmax >= tmin && tmax > 0;
}

ORay orc_camera_ray(const OCamera* cam, float x, float y, float aspect) {   /* camera.c:12-30 */
    OVec3 fwd = v_unit(v_sub(cam->look_at, cam->position));
    OVec3 right = v_unit(v_cross(fwd, cam->up));
    OVec3 up = v_cross(right, fwd);
    /* camera.c:17: (fov*0.5f) is float; *M_PI/180.0f is evaluated in double, then narrowed for tanf. */
    float scale = tanf((float)((double)(cam->fov * 0.5f) * 3.14159265358979323846 / (double)180.0f));
    float rx = (2.0f * x - 1.0f) * aspect * scale;
    float ry = (1.0f - 2.0f * y) * scale;
    ORay r;
    r.origin = cam->position;
    r.direction = v_unit(v_add(v_add(v_scale(right, rx), v_scale(up, ry)), fwd));
    return r;
}

OLight orc_make_light(OVec3 direction, OVec3 color) {          /* light.c:3-5 */
    OLight l; l.direction = v_unit(direction); l.color = color; return l;
}

/* ------------------------------------------------------------------ BVH traversal (bvh.c:80-129) */
typedef struct { uint64_t slabs, tris; } Work;

static int walk(const ONode* nodes, int32_t ni, ORay ray, const OTriangle* tris,
                float* t_out, float* u_out, float* v_out, int32_t* idx_out, Work* w) {
    const ONode* n = &nodes[ni];
    w->slabs++;
    if (!orc_ray_aabb(ray, n->bounds)) return 0;               /* bvh.c:82 */
    int hit = 0;
    float closest = *t_out;                                    /* bvh.c:85 */
    if (n->left < 0 && n->right < 0) {                         /* bvh.c:87-103: leaf, lowest index wins ties */
        for (int i = 0; i < n->triangle_count; i++) {
            const OTriangle* tr = &tris[n->start_idx + i];
            float t, u, v;
            w->tris++;
            if (orc_ray_triangle(ray, tr->v0, tr->v1, tr->v2, &t, &u, &v) && t < closest) {
                closest = t; *t_out = t; *u_out = u; *v_out = v; *idx_out = n->start_idx + i; hit = 1;
            }
        }
        return hit;
    }
    /* bvh.c:105-125: both children always visited with a fresh 1e30 bound; the right child wins ties. */
    float t1 = 1e30f, t2 = 1e30f, u1 = 0, v1 = 0, u2 = 0, v2 = 0;
    int32_t i1 = 0, i2 = 0;
    int h1 = n->left >= 0 ? walk(nodes, n->left, ray, tris, &t1, &u1, &v1, &i1, w) : 0;
    int h2 = n->right >= 0 ? walk(nodes, n->right, ray, tris, &t2, &u2, &v2, &i2, w) : 0;
    if (h1 && (!h2 || t1 < t2)) { *t_out = t1; *u_out = u1; *v_out = v1; *idx_out = i1; hit = 1; }
    else if (h2)                { *t_out = t2; *u_out = u2; *v_out = v2; *idx_out = i2; hit = 1; }
    return hit;
}

int orc_intersect_bvh(const ONode* nodes, int32_t node, ORay ray, const OTriangle* tris,
                      float* t, float* u, float* v, int32_t* tri_idx) {
    Work w = {0, 0};
    return walk(nodes, node, ray, tris, t, u, v, tri_idx, &w);
}

/* ------------------------------------------------------------------ BVH build (bvh.c:4-59) */
static float centroid_axis(const OTriangle* tr, int axis) {    /* bvh.c:30-31 */
    OVec3 c = v_scale(v_add(v_add(tr->v0, tr->v1), tr->v2), 1.0f / 3.0f);
    return axis == 0 ? c.x : (axis == 1 ? c.y : c.z);
}

static OAabb grow(OAabb b, OVec3 p) {                          /* aabb.c:10-23 */
    b.min = v3(fminf(b.min.x, p.x), fminf(b.min.y, p.y), fminf(b.min.z, p.z));
    b.max = v3(fmaxf(b.max.x, p.x), fmaxf(b.max.y, p.y), fmaxf(b.max.z, p.z));
    return b;
}

static int32_t build(OTriangle* tris, int32_t start, int32_t count, ONode* out, int32_t cap, int32_t* used) {
    if (*used >= cap) return -1;
    int32_t me = (*used)++;
    ONode* n = &out[me];
    n->start_idx = start; n->triangle_count = count; n->left = n->right = -1;
    OAabb b = { {1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f} };     /* aabb.c:3-8 */
    for (int32_t i = 0; i < count; i++) {
        b = grow(b, tris[start + i].v0); b = grow(b, tris[start + i].v1); b = grow(b, tris[start + i].v2);
    }
    n->bounds = b;
    if (count > 4) {                                            /* bvh.c:19 */
        OVec3 e = v_sub(b.max, b.min);
        int axis = 0;
        if (e.y > e.x) axis = 1;                                /* bvh.c:23 */
        if (e.z > e.x && e.z > e.y) axis = 2;                   /* bvh.c:24 */
        float split = 0.0f;
        for (int32_t i = 0; i < count; i++) split += centroid_axis(&tris[start + i], axis);
        split /= count;                                         /* bvh.c:33: float / int */
        int32_t mid = start;
        for (int32_t i = 0; i < count; i++) {                   /* bvh.c:37-48: swap-to-front partition */
            if (centroid_axis(&tris[start + i], axis) < split) {
                OTriangle tmp = tris[start + i]; tris[start + i] = tris[mid]; tris[mid] = tmp;
                mid++;
            }
        }
        int32_t nl = mid - start;
        if (nl > 0 && nl < count) {                             /* bvh.c:52-55 */
            int32_t l = build(tris, start, nl, out, cap, used);
            if (l < 0) return -1;
            int32_t r = build(tris, mid, count - nl, out, cap, used);
            if (r < 0) return -1;
            out[me].left = l; out[me].right = r;
        }
    }
    return me;
}

int32_t orc_build_bvh(OTriangle* tris, int32_t count, ONode* nodes_out, int32_t cap) {    /* bvh.c:61-67 */
    int32_t used = 0;
    if (build(tris, 0, count, nodes_out, cap, &used) < 0) return -1;
    return used;
}

/* ------------------------------------------------------------------ OBJ reader (mesh.c:25-78) */
int32_t orc_load_obj(const char* path, OTriangle* tris_out, int32_t cap, int32_t* n_v, int32_t* n_vt, int32_t* n_vn) {
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    enum { MAXE = 1000000 };                                    /* mesh.c:25-27 */
    OVec3* pos = (OVec3*)malloc(sizeof(OVec3) * MAXE);
    OVec2* uv = (OVec2*)malloc(sizeof(OVec2) * MAXE);
    OVec3* nor = (OVec3*)malloc(sizeof(OVec3) * MAXE);
    int32_t np = 0, nt = 0, nn = 0, ntri = 0;
    char line[256];                                             /* mesh.c:40 */
    while (fgets(line, sizeof line, f)) {
        if (line[0] == 'v' && line[1] == ' ') {
            if (np < MAXE) { sscanf(line + 2, "%f %f %f", &pos[np].x, &pos[np].y, &pos[np].z); np++; }
        } else if (line[0] == 'v' && line[1] == 't') {
            if (nt < MAXE) { sscanf(line + 3, "%f %f", &uv[nt].u, &uv[nt].v); nt++; }
        } else if (line[0] == 'v' && line[1] == 'n') {
            if (nn < MAXE) { sscanf(line + 3, "%f %f %f", &nor[nn].x, &nor[nn].y, &nor[nn].z); nn++; }
        } else if (line[0] == 'f') {                            /* mesh.c:59-75: 1-based v/vt/vn triples */
            int a[9];
            if (sscanf(line + 2, "%d/%d/%d %d/%d/%d %d/%d/%d", &a[0], &a[1], &a[2], &a[3], &a[4], &a[5], &a[6], &a[7], &a[8]) == 9
                && ntri < cap) {
                OTriangle* tr = &tris_out[ntri++];
                tr->v0 = pos[a[0] - 1]; tr->v1 = pos[a[3] - 1]; tr->v2 = pos[a[6] - 1];
                tr->t0 = uv[a[1] - 1];  tr->t1 = uv[a[4] - 1];  tr->t2 = uv[a[7] - 1];
                tr->n0 = nor[a[2] - 1]; tr->n1 = nor[a[5] - 1]; tr->n2 = nor[a[8] - 1];
            }
        }
    }
    fclose(f);
    free(pos); free(uv); free(nor);
    if (n_v) *n_v = np;
    if (n_vt) *n_vt = nt;
    if (n_vn) *n_vn = nn;
    return ntri;
}

/* ------------------------------------------------------------------ texture (mesh.c:138-149) */
void orc_sample_texture(const OMesh* mesh, float u, float v, float rgb[3]) {
    u = u - floorf(u);
    v = v - floorf(v);
    int x = (int)(u * (mesh->texture_width - 1));
    int y = (int)(v * (mesh->texture_height - 1));
    int idx = (y * mesh->texture_width + x) * 4;
    rgb[0] = mesh->texture_rgba[idx] / 255.0f;
    rgb[1] = mesh->texture_rgba[idx + 1] / 255.0f;
    rgb[2] = mesh->texture_rgba[idx + 2] / 255.0f;
}

/* ------------------------------------------------------------------ bicubic (utils/image.c:4-78) */
static float hermite(float A, float B, float C, float D, float t) {            /* image.c:4-11 */
    float a = -A / 2.0f + (3.0f * B) / 2.0f - (3.0f * C) / 2.0f + D / 2.0f;
    float b = A - (5.0f * B) / 2.0f + 2.0f * C - D / 2.0f;
    float c = -A / 2.0f + C / 2.0f;
    return a * t * t * t + b * t * t + c * t + B;
}

static uint32_t texel_clamped(const uint8_t* frame, int x, int y, int w, int h) {  /* image.c:14-22 */
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const uint8_t* p = frame + (size_t)(y * w + x) * 3;
    return ((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | p[2];
}

uint32_t orc_bicubic(const uint8_t* frame, float x, float y, int width, int height) {   /* image.c:25-78 */
    int x1 = (int)x, y1 = (int)y;
    float fx = x - x1, fy = y - y1;
    float col[3][4];
    for (int r = 0; r < 4; r++) {
        uint32_t p[4];
        for (int c = 0; c < 4; c++) p[c] = texel_clamped(frame, x1 + c - 1, y1 + r - 1, width, height);
        for (int ch = 0; ch < 3; ch++) {
            int sh = 16 - 8 * ch;
            col[ch][r] = hermite((float)((p[0] >> sh) & 0xFF), (float)((p[1] >> sh) & 0xFF),
                                 (float)((p[2] >> sh) & 0xFF), (float)((p[3] >> sh) & 0xFF), fx);
        }
    }
    uint32_t out = 0xFFu << 24;
    for (int ch = 0; ch < 3; ch++) {
        int q = (int)(hermite(col[ch][0], col[ch][1], col[ch][2], col[ch][3], fy) + 0.5f);
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        out |= (uint32_t)q << (16 - 8 * ch);
    }
    return out;
}

/* Output size of the encoder's upscale (scene.c:164-165). */
void orc_scaled_size(int width, int height, float scale_factor, int* scaled_width, int* scaled_height) {
    *scaled_width = (int)(width / scale_factor + 0.5f);
    *scaled_height = (int)(height / scale_factor + 0.5f);
}

/* One frame through the bicubic upscale loop (scene.c:188-202): ARGB, row-major. */
void orc_upscale_frame(const uint8_t* frame, int width, int height, int scaled_width, int scaled_height,
                       uint32_t* argb, int32_t threads) {
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(static) num_threads(threads)
    for (int y = 0; y < scaled_height; y++) {
        for (int x = 0; x < scaled_width; x++) {
            float src_x = x * (width - 1.0f) / (scaled_width - 1.0f);      /* scene.c:191 */
            float src_y = y * (height - 1.0f) / (scaled_height - 1.0f);    /* scene.c:192 */
            argb[(size_t)y * scaled_width + x] = orc_bicubic(frame, src_x, src_y, width, height);
        }
    }
}

/* ------------------------------------------------------------------ the frame loop (scene.c:52-160) */
static void render_frame_planes(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* light,
                                int32_t width, int32_t height, uint8_t* rgb, int32_t* hit_id, float* hit_t,
                                uint8_t* shadow, OCounters* counters, int32_t threads, int32_t* hit_mesh_out, int32_t* hit_tri_out) {
    float aspect = (float)width / height;                       /* scene.c:53 */
    uint64_t c_prim = 0, c_shad = 0, c_slab = 0, c_tri = 0, c_hit = 0, c_dark = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(dynamic, 4) num_threads(threads) \
        reduction(+ : c_prim, c_shad, c_slab, c_tri, c_hit, c_dark)
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            Work w = {0, 0};
            ORay ray = orc_camera_ray(cam, (x + 0.5f) / width, (y + 0.5f) / height, aspect);   /* scene.c:60-63 */
            c_prim++;
            float closest = 1e30f;
            int hit = 0, hit_mesh = -1, hit_tri = -1;
            OVec2 uv = {0, 0};
            OVec3 normal = {0, 0, 0};
            for (int m = 0; m < mesh_count; m++) {              /* scene.c:72-108 */
                const OMesh* mesh = &meshes[m];
                float t = closest, u = 0, v = 0;
                int32_t ti = 0;
                ORay local = orc_transform_ray(ray, mesh->transform);
                if (walk(mesh->nodes, 0, local, mesh->triangles, &t, &u, &v, &ti, &w) && t < closest) {
                    closest = t; hit = 1; hit_mesh = m; hit_tri = ti;
                    const OTriangle* tr = &mesh->triangles[ti];
                    float bw = 1.0f - u - v;
                    uv.u = bw * tr->t0.u + u * tr->t1.u + v * tr->t2.u;            /* scene.c:89-94 */
                    uv.v = bw * tr->t0.v + u * tr->t1.v + v * tr->t2.v;
                    normal = v_unit(v_add(v_add(v_scale(tr->n0, bw), v_scale(tr->n1, u)), v_scale(tr->n2, v)));
                    normal = orc_transform_normal(normal, mesh->transform);        /* scene.c:106 */
                }
            }
            size_t px = (size_t)y * width + x;
            uint8_t* out = rgb + px * 3;                         /* scene.c:110 */
            uint8_t sh = 255;
            if (hit) {
                float tex[3];
                orc_sample_texture(&meshes[hit_mesh], uv.u, uv.v, tex);            /* scene.c:112 */
                float diffuse = 0.2f;
                OVec3 p = v_add(ray.origin, v_scale(ray.direction, closest));      /* scene.c:118 */
                ORay sray;
                sray.origin = v_add(p, v_scale(normal, 0.001f));                   /* scene.c:119 */
                sray.direction = light->direction;
                c_shad++; c_hit++;
                int dark = 0;
                for (int m = 0; m < mesh_count && !dark; m++) {                    /* scene.c:124-139 */
                    float st = 1e30f, su = 0, sv = 0;
                    int32_t si = 0;
                    ORay local = orc_transform_ray(sray, meshes[m].transform);
                    if (walk(meshes[m].nodes, 0, local, meshes[m].triangles, &st, &su, &sv, &si, &w)) dark = 1;
                }
                if (!dark) diffuse = fmaxf(diffuse, v_dot(normal, light->direction));   /* scene.c:142-145 */
                else c_dark++;
                sh = (uint8_t)dark;
                OVec3 c = v_scale(v_mul(v3(tex[0], tex[1], tex[2]), light->color), diffuse);   /* scene.c:148-149 */
                out[0] = (unsigned char)(fminf(c.x * 255.0f, 255.0f));             /* scene.c:152-154: truncation */
                out[1] = (unsigned char)(fminf(c.y * 255.0f, 255.0f));
                out[2] = (unsigned char)(fminf(c.z * 255.0f, 255.0f));
            } else {
                out[0] = out[1] = out[2] = 50;                   /* scene.c:156 */
            }
            if (hit_id) hit_id[px] = hit ? ((hit_mesh << 24) | hit_tri) : -1;
            if (hit_mesh_out) hit_mesh_out[px] = hit ? hit_mesh : -1;
            if (hit_tri_out) hit_tri_out[px] = hit ? hit_tri : -1;
            if (hit_t) hit_t[px] = closest;
            if (shadow) shadow[px] = sh;
            c_slab += w.slabs; c_tri += w.tris;
        }
    }
    if (counters) {
        counters->primary_rays = c_prim; counters->shadow_rays = c_shad;
        counters->slab_tests = c_slab; counters->tri_tests = c_tri;
        counters->hit_pixels = c_hit; counters->shadowed_pixels = c_dark;
    }
}

void orc_render_frame(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* light,
                      int32_t width, int32_t height, uint8_t* rgb, int32_t* hit_id, float* hit_t,
                      uint8_t* shadow, OCounters* counters, int32_t threads) {
    render_frame_planes(meshes, mesh_count, cam, light, width, height, rgb, hit_id, hit_t, shadow, counters, threads, NULL, NULL);
}

void orc_render_frame_ids(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* light,
                          int32_t width, int32_t height, uint8_t* rgb, int32_t* hit_mesh, int32_t* hit_tri, float* hit_t,
                          uint8_t* shadow, OCounters* counters, int32_t threads) {
    render_frame_planes(meshes, mesh_count, cam, light, width, height, rgb, NULL, hit_t, shadow, counters, threads, hit_mesh, hit_tri);
}

/* ------------------------------------------------------------------ sampled frames (beyond the reference)
 * Several camera rays per pixel and several lights around the reference's per-ray operations.  There is no
 * reference for this: it is pinned only where it coincides with the reference (spp == 1, one light),
 * where it must equal orc_render_frame.  Definition: include/rt_mi355.h, rt_render_frame_sampled. */
static uint32_t mix32(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}

void orc_sample_offset(uint32_t x, uint32_t y, uint32_t s, uint32_t seed, float* jx, float* jy) {
    uint32_t h0 = mix32(x ^ mix32(y ^ mix32(s ^ mix32(seed))));
    uint32_t h1 = mix32(h0);
    *jx = (float)(h0 >> 8) * (1.0f / 16777216.0f);
    *jy = (float)(h1 >> 8) * (1.0f / 16777216.0f);
}

void orc_render_frame_sampled(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* lights,
                              int32_t n_lights, int32_t spp, uint32_t seed, int32_t width, int32_t height,
                              uint8_t* rgb, int32_t* hit_id, OCounters* counters, int32_t threads) {
    orc_render_window_sampled(meshes, mesh_count, cam, lights, n_lights, spp, seed, width, height, 0, 0, width, height,
                              rgb, hit_id, counters, threads);
}

/* The pixels [x0, x0 + win_w) x [y0, y0 + win_h) of the same frame (rays are those of the width x height frame), written
 * row-major into planes of the window's own size: a frame of 8 million pixels and 16 samples is hours of CPU work, a window
 * of it is seconds. */
void orc_render_window_sampled(const OMesh* meshes, int32_t mesh_count, const OCamera* cam, const OLight* lights,
                               int32_t n_lights, int32_t spp, uint32_t seed, int32_t width, int32_t height,
                               int32_t x0, int32_t y0, int32_t win_w, int32_t win_h,
                               uint8_t* rgb, int32_t* hit_id, OCounters* counters, int32_t threads) {
    float aspect = (float)width / height;
    uint64_t c_prim = 0, c_shad = 0, c_slab = 0, c_tri = 0, c_hit = 0, c_dark = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#endif
    #pragma omp parallel for schedule(dynamic, 4) num_threads(threads) \
        reduction(+ : c_prim, c_shad, c_slab, c_tri, c_hit, c_dark)
    for (int y = y0; y < y0 + win_h; y++) {
        for (int x = x0; x < x0 + win_w; x++) {
            Work w = {0, 0};
            float acc[3] = {0, 0, 0};
            int first_id = -1;
            for (int s = 0; s < spp; s++) {
                float jx = 0.5f, jy = 0.5f;
                if (spp > 1) orc_sample_offset((uint32_t)x, (uint32_t)y, (uint32_t)s, seed, &jx, &jy);
                ORay ray = orc_camera_ray(cam, (x + jx) / width, (y + jy) / height, aspect);
                c_prim++;
                float closest = 1e30f;
                int hit = 0, hit_mesh = -1, hit_tri = -1;
                OVec2 uv = {0, 0};
                OVec3 normal = {0, 0, 0};
                for (int m = 0; m < mesh_count; m++) {
                    const OMesh* mesh = &meshes[m];
                    float t = closest, u = 0, v = 0;
                    int32_t ti = 0;
                    ORay local = orc_transform_ray(ray, mesh->transform);
                    if (walk(mesh->nodes, 0, local, mesh->triangles, &t, &u, &v, &ti, &w) && t < closest) {
                        closest = t; hit = 1; hit_mesh = m; hit_tri = ti;
                        const OTriangle* tr = &mesh->triangles[ti];
                        float bw = 1.0f - u - v;
                        uv.u = bw * tr->t0.u + u * tr->t1.u + v * tr->t2.u;
                        uv.v = bw * tr->t0.v + u * tr->t1.v + v * tr->t2.v;
                        normal = v_unit(v_add(v_add(v_scale(tr->n0, bw), v_scale(tr->n1, u)), v_scale(tr->n2, v)));
                        normal = orc_transform_normal(normal, mesh->transform);
                    }
                }
                float value[3] = {50.0f, 50.0f, 50.0f};
                if (hit) {
                    c_hit++;
                    if (s == 0) first_id = (hit_mesh << 24) | hit_tri;
                    float tex[3];
                    orc_sample_texture(&meshes[hit_mesh], uv.u, uv.v, tex);
                    OVec3 p = v_add(ray.origin, v_scale(ray.direction, closest));
                    ORay sray;
                    sray.origin = v_add(p, v_scale(normal, 0.001f));
                    OVec3 color = {0, 0, 0};
                    for (int l = 0; l < n_lights; l++) {
                        sray.direction = lights[l].direction;
                        c_shad++;
                        int dark = 0;
                        for (int m = 0; m < mesh_count && !dark; m++) {
                            float st = 1e30f, su = 0, sv = 0;
                            int32_t si = 0;
                            ORay local = orc_transform_ray(sray, meshes[m].transform);
                            if (walk(meshes[m].nodes, 0, local, meshes[m].triangles, &st, &su, &sv, &si, &w)) dark = 1;
                        }
                        float diffuse = 0.2f;
                        if (!dark) diffuse = fmaxf(diffuse, v_dot(normal, lights[l].direction));
                        else c_dark++;
                        OVec3 term = v_scale(v_mul(v3(tex[0], tex[1], tex[2]), lights[l].color), diffuse);
                        color = l == 0 ? term : v_add(color, term);
                    }
                    value[0] = fminf(color.x * 255.0f, 255.0f);
                    value[1] = fminf(color.y * 255.0f, 255.0f);
                    value[2] = fminf(color.z * 255.0f, 255.0f);
                }
                for (int k = 0; k < 3; k++) acc[k] = s == 0 ? value[k] : acc[k] + value[k];
            }
            size_t px = (size_t)(y - y0) * win_w + (x - x0);
            for (int k = 0; k < 3; k++) {
                float v = spp > 1 ? acc[k] / (float)spp : acc[k];
                rgb[px * 3 + k] = (unsigned char)v;
            }
            if (hit_id) hit_id[px] = first_id;
            c_slab += w.slabs; c_tri += w.tris;
        }
    }
    if (counters) {
        counters->primary_rays = c_prim; counters->shadow_rays = c_shad;
        counters->slab_tests = c_slab; counters->tri_tests = c_tri;
        counters->hit_pixels = c_hit; counters->shadowed_pixels = c_dark;
    }
}

/* ------------------------------------------------------------------ driver animation (raytracer.c:41-53) */
void orc_animation_transforms(int32_t frame, OTransform out[3]) {
    /* raytracer.c:41: the factor 2.0f*M_PI/120.0f is a double; the product is narrowed to float. */
    float t = (float)(frame * ((double)2.0f * 3.14159265358979323846 / (double)120.0f));
    memset(out, 0, sizeof(OTransform) * 3);
    out[0].position = v3(2.0f * cosf(t), 1.0f + 0.2f * sinf(2 * t), 2.0f * sinf(t));    /* raytracer.c:44-45 */
    out[0].rotation = v3(0.1f * sinf(t), t, 0.1f * cosf(t));                           /* raytracer.c:46-47 */
    out[1].position = v3(1.0f, 0.5f + 0.1f * sinf(t), 1.0f);                           /* raytracer.c:50-51 */
    out[1].rotation = v3(0, t * 0.5f, 0);                                              /* raytracer.c:52-53 */
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
