// rt_exact.h -- float32 building blocks with the reference's operation order.
//
// Everything here must be compiled with FMA contraction OFF and correctly rounded
// division / square root: each C operator below is exactly one IEEE rounding, in the order
// the reference's expressions evaluate (citations are to files of the reference).
// Usable from host code (g++ / hipcc host pass) and from device code.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct F3 { float x, y, z; };

RT_HD F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD F3 f3(const float* p) { return f3(p[0], p[1], p[2]); }
RT_HD F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }            // vec3.c:3
RT_HD F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }            // vec3.c:7
RT_HD F3 scale(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }             // vec3.c:11
RT_HD float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           // vec3.c:19
RT_HD F3 cross(F3 a, F3 b) {                                                        // vec3.c:23
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// Reciprocal and reciprocal square root.  Exact build: the correctly rounded division (and square root) the reference's
// expressions ask for.  Fast build of the kernels (RT_FAST, device code only): the hardware approximations, one
// instruction each, about 1 ulp.
#if defined(RT_FAST) && defined(__HIP_DEVICE_COMPILE__)
RT_HD float rcp1(float x) { return __builtin_amdgcn_rcpf(x); }
RT_HD float rsqrt1(float x) { return __builtin_amdgcn_rsqf(x); }
#else
RT_HD float rcp1(float x) { return 1.0f / x; }
RT_HD float rsqrt1(float x) { return 1.0f / sqrtf(x); }
#endif
// vec3.c:15,31,35: v * (1.0f / sqrtf(v.v)) -- a reciprocal and three multiplies, not three divides.
RT_HD F3 unit(F3 a) { return scale(a, rsqrt1(dot(a, a))); }
RT_HD F3 mul(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }            // vec3.c:39
RT_HD F3 recip(F3 a) { return f3(rcp1(a.x), rcp1(a.y), rcp1(a.z)); }                // aabb.c:34-38

// scene.c:152-154 stores (unsigned char)fminf(c * 255.0f, 255.0f).  For a negative value (a light colour below zero) C leaves
// that conversion undefined; the reference as compiled for x86-64 converts to a 32-bit integer by truncation (anything below
// the integer range gives INT_MIN) and keeps the low byte.  Written out here, so that every compiler gives those bytes: left
// as the bare cast, device code merged the three byte stores of a pixel on the assumption that each value fits a byte, and a
// negative red channel spilled into green and blue.
RT_HD uint8_t byte_of(float v) { return (uint8_t)(uint32_t)(int32_t)fmaxf(v, -2147483648.0f); }
RT_HD uint8_t colour_byte(float c) { return byte_of(fminf(c * 255.0f, 255.0f)); }

// fminf/fmaxf as the reference's C library evaluates them where the result is stored (box bounds, aabb.c:10-23):
// of two equal operands -- +0 and -0 -- the second is returned, and a NaN second operand is ignored.
// Compilers that expand fminf inline pick the other zero.
RT_HD float min_later(float a, float b) { return (a < b || b != b) ? a : b; }
RT_HD float max_later(float a, float b) { return (a > b || b != b) ? a : b; }

// aabb.c:33-56 with the reciprocals passed in (same values the reference recomputes per call).
// fminf/fmaxf return the non-NaN operand, which the zero-extent boxes of flat meshes rely on
// (0 * inf = NaN inside the slab arithmetic).
RT_HD bool slab(F3 o, F3 inv, const float* bmin, const float* bmax) {
    float a = (bmin[0] - o.x) * inv.x, b = (bmax[0] - o.x) * inv.x;
    float tmin = fminf(a, b), tmax = fmaxf(a, b);
    a = (bmin[1] - o.y) * inv.y; b = (bmax[1] - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(a, b)); tmax = fminf(tmax, fmaxf(a, b));
    a = (bmin[2] - o.z) * inv.z; b = (bmax[2] - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(a, b)); tmax = fminf(tmax, fmaxf(a, b));
    return tmax >= tmin && tmax > 0;
}
// Same test, also reporting the entry distance (used only for distance-ordered traversal).
RT_HD bool slab_t(F3 o, F3 inv, const float* bmin, const float* bmax, float& entry) {
    float a = (bmin[0] - o.x) * inv.x, b = (bmax[0] - o.x) * inv.x;
    float tmin = fminf(a, b), tmax = fmaxf(a, b);
    a = (bmin[1] - o.y) * inv.y; b = (bmax[1] - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(a, b)); tmax = fminf(tmax, fmaxf(a, b));
    a = (bmin[2] - o.z) * inv.z; b = (bmax[2] - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(a, b)); tmax = fminf(tmax, fmaxf(a, b));
    entry = tmin;
    return tmax >= tmin && tmax > 0;
}

// ray.c:42-67 with edge1/edge2 precomputed (v1-v0, v2-v0: the same two subtractions).
// Returns true iff the reference would return true; t,u,v as the reference writes them.
RT_HD bool moller_trumbore(F3 o, F3 d, F3 v0, F3 e1, F3 e2, float& t, float& u, float& v) {
    const float eps = 0.0000001f;
    F3 h = cross(d, e2);
    float a = dot(e1, h);
    if (a > -eps && a < eps) return false;
    float f = rcp1(a);
    F3 s = sub(o, v0);
    u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    F3 q = cross(s, e1);
    v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot(e2, q);
    return t > eps;
}

// ---- 4x4 matrices (host side: per-frame constants) -------------------------------------
struct M4 { float m[4][4]; };

inline M4 m4_identity() {                                          // mat4.c:3-7
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return r;
}
inline M4 m4_mul(const M4& a, const M4& b) {                       // mat4.c:9-19
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.0f;                                      // the reference accumulates onto 0.0f
            for (int k = 0; k < 4; k++) acc += a.m[i][k] * b.m[k][j];
            r.m[i][j] = acc;
        }
    return r;
}
inline M4 m4_translation(F3 t) {                                   // mat4.c:21-27
    M4 r = m4_identity();
    r.m[0][3] = t.x; r.m[1][3] = t.y; r.m[2][3] = t.z;
    return r;
}
// mat4.c:29-60: rotation about axis 0/1/2 (x/y/z).  The two off-diagonal entries are (p,q)
// and (q,p) with p<q the other two axes; the minus sign is on (p,q) except for y.
inline M4 m4_rotation(int axis, float angle) {
    M4 r = m4_identity();
    const float c = cosf(angle), s = sinf(angle);
    const int p = axis == 0 ? 1 : 0, q = axis == 2 ? 1 : 2;
    r.m[p][p] = c; r.m[q][q] = c;
    r.m[p][q] = axis == 1 ? s : -s;
    r.m[q][p] = axis == 1 ? -s : s;
    return r;
}
// mat4.c:79-214.  Entry (i,j) of the adjugate: with R the rows != j and C the columns != i
// (ascending), the reference sums, left to right, the six products
//   +R0C0*R1C1*R2C2 -R0C0*R1C2*R2C1 -R1C0*R0C1*R2C2 +R1C0*R0C2*R2C1 +R2C0*R0C1*R1C2 -R2C0*R0C2*R1C1
// times (-1)^(i+j); each product is (a*b)*c.  Then det from row 0 / column 0, identity if det == 0,
// else every entry times 1.0f/det.
inline M4 m4_inverse(const M4& a) {
    static const int r1[6] = {0, 0, 1, 1, 2, 2}, r2[6] = {1, 1, 0, 0, 0, 0}, r3[6] = {2, 2, 2, 2, 1, 1};
    static const int c2[6] = {1, 2, 1, 2, 1, 2}, c3[6] = {2, 1, 2, 1, 2, 1};
    static const bool neg[6] = {false, true, true, false, false, true};
    M4 adj;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            int R[3], C[3];
            for (int k = 0, nr = 0, nc = 0; k < 4; k++) { if (k != j) R[nr++] = k; if (k != i) C[nc++] = k; }
            const bool flip = ((i + j) & 1) != 0;
            float acc = 0.0f;
            for (int k = 0; k < 6; k++) {
                float p = (a.m[R[r1[k]]][C[0]] * a.m[R[r2[k]]][C[c2[k]]]) * a.m[R[r3[k]]][C[c3[k]]];
                if (neg[k] != flip) p = -p;                        // negation is exact
                acc = k == 0 ? p : acc + p;
            }
            adj.m[i][j] = acc;
        }
    float det = a.m[0][0] * adj.m[0][0] + a.m[0][1] * adj.m[1][0] + a.m[0][2] * adj.m[2][0] + a.m[0][3] * adj.m[3][0];
    if (det == 0) return m4_identity();
    det = 1.0f / det;
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = adj.m[i][j] * det;
    return r;
}
inline M4 m4_transpose(const M4& a) {                              // mat4.c:216-224
    M4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
    return r;
}

// mat4.c:62-69 on rows given separately (rows 0..2 as [3][4], row 3 as [4]).
RT_HD F3 xform_point(const float (*m)[4], const float* wrow, F3 p) {
    float w = wrow[0] * p.x + wrow[1] * p.y + wrow[2] * p.z + wrow[3];
    return f3((m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3]) / w,
              (m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3]) / w,
              (m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]) / w);
}
// The same when the caller knows that w == 1.0f for every finite point (row 3 is 0 0 0 1): x / 1.0f is x.
RT_HD F3 xform_point_w1(const float (*m)[4], F3 p) {
    return f3(m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3],
              m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3],
              m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3]);
}
// mat4.c:71-77 on a [3][4] or [3][3] block.
template <int STRIDE>
RT_HD F3 xform_vector(const float (*m)[STRIDE], F3 v) {
    return f3(m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
              m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z);
}

}  // namespace rt
