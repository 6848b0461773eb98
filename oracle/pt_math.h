/*
 * oracle/pt_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * The arithmetic of the reference's hot path lives in a third-party
 * dependency that is ABSENT from /root/reference: cglm
 * (https://github.com/recp/cglm.git, .gitmodules:1-3; external/cglm/ is an
 * empty directory, no pinned commit is recoverable).  This header restates
 * the published algorithm of each cglm struct-API function the path calls,
 * under its own names, following SURVEY.md section 8(c)'s function table.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures
 * for this path, and it cannot be built here without writing stand-ins for
 * the missing cglm headers, so nothing the reference provides pins the exact
 * float formulation below.  The few known answers in SURVEY.md section 8(c)
 * are checked in tests/test_oracle.py.
 *
 * Each function names the cglm function it restates and the reference call
 * sites that use it.  Plain C, binary32 throughout, evaluation order written
 * out explicitly; build with -ffp-contract=off (a default x86-64 gcc build
 * of the reference has no FMA to contract into).
 */
#ifndef ORACLE_PT_MATH_H
#define ORACLE_PT_MATH_H

#include "path_tracer_api.h" /* vec3s: the scene's input format */

#include <float.h>
#include <math.h>
#include <stdbool.h>

static inline vec3s om_make(float x, float y, float z) {
    vec3s v;
    v.x = x;
    v.y = y;
    v.z = z;
    return v;
}

/* glms_vec3_zero / _one / _broadcast -- src/renderer.c:16,44,55,103 */
static inline vec3s om_zero(void) { return om_make(0.0f, 0.0f, 0.0f); }
static inline vec3s om_one(void) { return om_make(1.0f, 1.0f, 1.0f); }

/* glms_vec3_add / _sub / _mul -- src/renderer.c:16,41,49,65,73,106;
 * src/ray.c:29,31,49,51-52 */
static inline vec3s om_add(vec3s a, vec3s b) {
    return om_make(a.x + b.x, a.y + b.y, a.z + b.z);
}
static inline vec3s om_sub(vec3s a, vec3s b) {
    return om_make(a.x - b.x, a.y - b.y, a.z - b.z);
}
static inline vec3s om_mul(vec3s a, vec3s b) {
    return om_make(a.x * b.x, a.y * b.y, a.z * b.z);
}

/* glms_vec3_scale / _negate / _dot -- src/renderer.c:16,37,40,49,57,59,66,108;
 * src/ray.c:29,49,55-56.  dot is x*x' + y*y' + z*z' summed left to right. */
static inline vec3s om_scale(vec3s a, float s) {
    return om_make(a.x * s, a.y * s, a.z * s);
}
static inline vec3s om_negate(vec3s a) { return om_make(-a.x, -a.y, -a.z); }
static inline float om_dot(vec3s a, vec3s b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}

/* glm_vec3_cross: (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x) */
static inline vec3s om_cross(vec3s a, vec3s b) {
    return om_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z,
                   a.x * b.y - a.y * b.x);
}

/* glms_vec3_normalize -- src/renderer.c:17,41,47,64,81; src/ray.c:30.
 * norm = sqrtf(dot(a,a)); below FLT_EPSILON the result is the zero vector;
 * otherwise a * (1.0f / norm) (multiply by the reciprocal). */
static inline vec3s om_normalize(vec3s a) {
    float norm = sqrtf(om_dot(a, a));
    if (norm < FLT_EPSILON)
        return om_zero();
    return om_scale(a, 1.0f / norm);
}

/* glms_vec3_crossn -- src/ray.c:50 */
static inline vec3s om_crossn(vec3s a, vec3s b) {
    return om_normalize(om_cross(a, b));
}

/* glms_vec3_lerp -- src/renderer.c:72: from + (to - from) * t, unclamped */
static inline vec3s om_lerp(vec3s from, vec3s to, float t) {
    vec3s span = om_sub(to, from);
    return om_add(from, om_scale(span, t));
}

/* glm_clamp = glm_min(glm_max(v, lo), hi) with `a > b ? a : b` / `a < b ? a
 * : b` (a NaN therefore becomes lo) -- glms_vec3_clamp, src/renderer.c:88 */
static inline float om_clampf(float v, float lo, float hi) {
    float up = (v > lo) ? v : lo;
    return (up < hi) ? up : hi;
}
static inline vec3s om_clamp(vec3s v, float lo, float hi) {
    return om_make(om_clampf(v.x, lo, hi), om_clampf(v.y, lo, hi),
                   om_clampf(v.z, lo, hi));
}

/* glms_vec3_reflect -- src/renderer.c:47: v - n * (2 * dot(v, n)) */
static inline vec3s om_reflect(vec3s v, vec3s n) {
    float twice = 2.0f * om_dot(v, n);
    return om_sub(v, om_scale(n, twice));
}

/* glms_vec3_refract -- src/renderer.c:62-63.  ndi = dot(n, v);
 * eni = eta*ndi; k = 1 - eta*eta + eni*eni; k < 0: dest = 0, false;
 * else dest = v*eta - n*(eni + sqrtf(k)), true. */
static inline bool om_refract(vec3s v, vec3s n, float eta, vec3s *dest) {
    float ndi = om_dot(n, v);
    float eni = eta * ndi;
    float k = 1.0f - eta * eta + eni * eni;
    if (k < 0.0f) {
        *dest = om_zero();
        return false;
    }
    vec3s bent = om_scale(v, eta);
    float along = eni + sqrtf(k);
    *dest = om_sub(bent, om_scale(n, along));
    return true;
}

/* glms_ray_sphere -- src/ray.c:21.  Unit-length `dir` assumed.  The
 * numerically stabilised form (remedy term + the q / c/q root pair):
 *   dp = c - o; ddp = dot(dir, dp); dpp = dot(dp, dp); dp -= dir*ddp;
 *   dscr = r*r - dot(dp, dp); dscr < 0 -> false;
 *   q = ddp >= 0 ? ddp + sqrt(dscr) : ddp - sqrt(dscr);
 *   t1 = q; t2 = (dpp - r*r)/q; order so t1 <= t2; false iff both < 0. */
static inline bool om_ray_sphere(vec3s origin, vec3s dir, vec3s center,
                                 float radius, float *t1, float *t2) {
    vec3s dp = om_sub(center, origin);
    float ddp = om_dot(dir, dp);
    float dpp = om_dot(dp, dp);
    dp = om_sub(dp, om_scale(dir, ddp));
    float r2 = radius * radius;
    float dscr = r2 - om_dot(dp, dp);
    if (dscr < 0.0f)
        return false;
    dscr = sqrtf(dscr);
    float q = (ddp >= 0.0f) ? (ddp + dscr) : (ddp - dscr);
    float near_t = q;
    float far_t = (dpp - r2) / q;
    if (near_t > far_t) {
        float keep = near_t;
        near_t = far_t;
        far_t = keep;
    }
    *t1 = near_t;
    *t2 = far_t;
    if (near_t < 0.0f && far_t < 0.0f)
        return false;
    return true;
}

/* glms_ray_triangle -- src/ray.c:42-43.  Moeller-Trumbore, eps = 1e-6,
 * two-sided, u in [0,1], v >= 0, u+v <= 1, hit iff dist > eps. */
static inline bool om_ray_triangle(vec3s origin, vec3s dir, vec3s v0, vec3s v1,
                                   vec3s v2, float *dist) {
    const float eps = 0.000001f;
    vec3s edge1 = om_sub(v1, v0);
    vec3s edge2 = om_sub(v2, v0);
    vec3s p = om_cross(dir, edge2);
    float det = om_dot(edge1, p);
    if (det > -eps && det < eps)
        return false;
    float inv_det = 1.0f / det;
    vec3s t = om_sub(origin, v0);
    float u = inv_det * om_dot(t, p);
    if (u < 0.0f || u > 1.0f)
        return false;
    vec3s q = om_cross(t, edge1);
    float v = inv_det * om_dot(dir, q);
    if (v < 0.0f || u + v > 1.0f)
        return false;
    float d = inv_det * om_dot(edge2, q);
    *dist = d;
    return d > eps;
}

#endif /* ORACLE_PT_MATH_H */
