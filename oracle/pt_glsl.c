/*
 * oracle/pt_glsl.c -- TEST INFRASTRUCTURE: a CPU restatement of the reference's
 * OTHER renderer, the OpenGL fragment shader shaders/rtx_frag.glsl, as the
 * checker of the product's PTGPU_FLAG_GLSL_SEMANTICS mode.
 * Only tests/ may load this.
 *
 * PARITY UNPINNED, twice over: the reference holds no image or fixture of its
 * shader's output, and GLSL leaves the precision of its built-ins (normalize,
 * reflect, refract, sqrt, log, cos, division) to the implementation.  This
 * file fixes them as sequentially rounded binary32 operations (gcc,
 * -ffp-contract=off), log and cos as the two short kernels of
 * include/ptgpu_glsl_math.h (a shared specification, like ptgpu_rng.h); the
 * device side makes the same choices, so its frames can be held to this file
 * bit for bit.
 *
 * What the shader does differently from src/renderer.c, all
 * restated here with its line numbers:
 *   :64-75   PCG hash stream, one state per pixel carried across the samples
 *   :78-90   rand_unit_sphere = normalize(3 Box-Muller normals): 6 draws per hit
 *   :92-117  own quadratic (a = d.d, first root >= 0), no unit-direction assumption
 *   :121-156 Moeller-Trumbore that also rejects u, v, t < 1e-6
 *   :158-192 spheres first, then triangles; a miss may overwrite a hit at <= 0
 *   :194-272 explicit LIFO stack, throughput form: total += colour * Le;
 *            colour *= albedo; roughness not halved; reflect NOT renormalised,
 *            origin offset 1e-4; material refractive_index; total internal
 *            reflection drops the transmitted ray only; children weighted
 *            (1 - t) and t
 *   :274-295 rays through pixel CENTRES; no per-sample clamp; mean over samples
 *   GL_RGB8 store (src/main.c:297,328): clamp to [0,1], round to nearest
 */
#include "path_tracer_api.h"
#include "ptgpu_glsl_math.h"
#include "ptgpu_rng.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static v3 mk(float x, float y, float z) { v3 v = {x, y, z}; return v; }
static v3 from(vec3s a) { return mk(a.x, a.y, a.z); }
static v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 mul(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 cross3(v3 a, v3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* GLSL normalize: v * (1 / length(v)); no zero test (0 * inf = NaN, as undefined
 * as the shader leaves it) */
static v3 normalize3(v3 a) { return scale(a, 1.0f / sqrtf(dot3(a, a))); }
/* GLSL reflect: I - 2 * dot(N, I) * N */
static v3 reflect3(v3 i, v3 n) { return sub(i, scale(n, 2.0f * dot3(n, i))); }
/* GLSL refract: k = 1 - eta^2 (1 - dot(N,I)^2); k < 0 -> 0; else eta I - (eta dot(N,I) + sqrt k) N */
static v3 refract3(v3 i, v3 n, float eta) {
    float ndi = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f)
        return mk(0.0f, 0.0f, 0.0f);
    return sub(scale(i, eta), scale(n, eta * ndi + sqrtf(k)));
}

/* :78-90 */
static float normal_dist(uint32_t *state) {
    float theta = 2.0f * 3.1415926535897932384626433832795f * ptgpu_pcg_unit(state);
    float rho = sqrtf(-2.0f * ptgpu_glsl_logf(ptgpu_pcg_unit(state)));
    return rho * ptgpu_glsl_cosf(theta);
}
static v3 rand_unit_sphere(uint32_t *state) {
    float x = normal_dist(state);
    float y = normal_dist(state);
    float z = normal_dist(state);
    return normalize3(mk(x, y, z));
}

/* :92-117 */
static float ray_sphere_intersect(v3 o, v3 d, v3 center, float radius) {
    o = sub(o, center);
    float a = dot3(d, d);
    float b = 2.0f * dot3(o, d);
    float c = dot3(o, o) - radius * radius; /* pow(s.radius, 2.0) */
    float det = b * b - 4.0f * a * c;
    if (det < 0.0f)
        return -1.0f;
    float sqrt_det = sqrtf(det);
    float two_a = 2.0f * a;
    float t1 = (-b - sqrt_det) / two_a;
    if (t1 >= 0.0f)
        return t1;
    float t2 = (-b + sqrt_det) / two_a;
    if (t2 >= 0.0f)
        return t2;
    return -1.0f;
}

/* :121-156 */
static float ray_triangle_intersect(v3 o, v3 d, v3 v0, v3 v1, v3 v2) {
    const float epsilon = 1e-6f;
    v3 edge1 = sub(v1, v0), edge2 = sub(v2, v0);
    v3 ray_cross_e2 = cross3(d, edge2);
    float det = dot3(edge1, ray_cross_e2);
    if (fabsf(det) < epsilon)
        return -1.0f;
    float inv_det = 1.0f / det;
    v3 s = sub(o, v0);
    float u = inv_det * dot3(s, ray_cross_e2);
    if (u < epsilon || u > 1.0f)
        return -1.0f;
    v3 s_cross_e1 = cross3(s, edge1);
    float v = inv_det * dot3(d, s_cross_e1);
    if (v < epsilon || (u + v) > 1.0f)
        return -1.0f;
    float t = inv_det * dot3(edge2, s_cross_e1);
    if (t < epsilon)
        return -1.0f;
    return t;
}

typedef struct { float distance; v3 point, normal; size_t material; } hit_t;

/* :158-192 -- the GL driver uploads the scene's spheres, then its triangles,
 * each in scene order (src/main.c:80-135) */
static hit_t trace(const scene *w, v3 origin, v3 direction) {
    hit_t h;
    memset(&h, 0, sizeof h);
    h.distance = -1.0f;
    for (size_t i = 0; i < w->num_objects; i++) {
        const shape *o = &w->objects[i];
        if (o->tag != SPHERE)
            continue;
        float dist = ray_sphere_intersect(origin, direction, from(o->sphere.center), o->sphere.radius);
        if (h.distance > 0.0f && (dist < 0.0f || dist >= h.distance))
            continue;
        h.distance = dist;
        h.point = add(origin, scale(direction, dist));
        h.normal = normalize3(sub(h.point, from(o->sphere.center)));
        h.material = o->material;
    }
    for (size_t i = 0; i < w->num_objects; i++) {
        const shape *o = &w->objects[i];
        if (o->tag != TRIANGLE)
            continue;
        v3 v0 = from(o->triangle.v0), v1 = from(o->triangle.v1), v2 = from(o->triangle.v2);
        float dist = ray_triangle_intersect(origin, direction, v0, v1, v2);
        if (h.distance > 0.0f && (dist < 0.0f || dist >= h.distance))
            continue;
        h.distance = dist;
        h.point = add(origin, scale(direction, dist));
        v3 normal = normalize3(cross3(sub(v0, v1), sub(v0, v2)));
        float side = dot3(direction, normal) < 0.0f ? 1.0f : -1.0f; /* 2 * int(..) - 1 */
        h.normal = scale(normal, side);
        h.material = o->material;
    }
    return h;
}

typedef struct { v3 origin, direction; int bounces; v3 color; } item_t;

/* :194-272 */
static v3 incident_light(const scene *w, int max_bounces, item_t *stack, v3 origin, v3 direction,
                         uint32_t *rng, uint64_t *rays) {
    int n = 0;
    stack[n].origin = origin;
    stack[n].direction = direction;
    stack[n].bounces = 0;
    stack[n].color = mk(1.0f, 1.0f, 1.0f);
    n++;
    const v3 sky = from(w->sky_color);
    v3 total = mk(0.0f, 0.0f, 0.0f);
    while (n > 0) {
        item_t cur = stack[--n];
        if (cur.bounces > max_bounces) {
            total = add(total, mul(cur.color, sky));
            continue;
        }
        ++*rays;
        hit_t hit = trace(w, cur.origin, cur.direction);
        if (hit.distance < 0.0f) {
            total = add(total, mul(cur.color, sky));
            continue;
        }
        const shape_material *mat = &w->materials[hit.material];
        v3 le = scale(from(mat->emission_color), mat->emission_strength);
        total = add(total, mul(cur.color, le));
        cur.color = mul(cur.color, from(mat->albedo));
        v3 deviation = scale(rand_unit_sphere(rng), mat->roughness);
        v3 normal = normalize3(add(hit.normal, deviation));
        if (mat->transparency < 1.0f) {
            item_t next;
            next.direction = reflect3(cur.direction, normal);
            next.origin = add(hit.point, scale(next.direction, 0.0001f));
            next.bounces = cur.bounces + 1;
            next.color = scale(cur.color, 1.0f - mat->transparency);
            stack[n++] = next;
        }
        if (mat->transparency > 0.0f) {
            float facing = dot3(cur.direction, normal);
            float eta = facing < 0.0f ? 1.0f / mat->refractive_index : mat->refractive_index;
            v3 rn = facing < 0.0f ? normal : scale(normal, -1.0f);
            item_t next;
            next.direction = refract3(cur.direction, rn, eta);
            if (next.direction.x == 0.0f && next.direction.y == 0.0f && next.direction.z == 0.0f)
                continue;
            next.origin = add(hit.point, scale(next.direction, 0.0001f));
            next.bounces = cur.bounces + 1;
            next.color = scale(cur.color, mat->transparency);
            stack[n++] = next;
        }
    }
    return total;
}

/* the shared specification functions (include/ptgpu_rng.h, ptgpu_glsl_math.h), for the tests */
float oracle_glsl_logf(float x) { return ptgpu_glsl_logf(x); }
float oracle_glsl_cosf(float x) { return ptgpu_glsl_cosf(x); }
uint32_t oracle_pcg_next(uint32_t *state) { return ptgpu_pcg_next(state); }
float oracle_pcg_unit(uint32_t *state) { return ptgpu_pcg_unit(state); }

/* float -> GL_RGB8: clamp to [0,1] (a NaN to 0), times 255, to nearest */
static uint8_t unorm8(float v) {
    if (!(v > 0.0f))
        return 0;
    if (v > 1.0f)
        v = 1.0f;
    return (uint8_t)(v * 255.0f + 0.5f);
}

/* Rows [row_begin, row_end) of a w x h picture, row 0 = top.  `mean` (may be
 * NULL) receives the unclamped per-pixel means, 3 floats per pixel of the
 * full frame.  Returns the number of trace_ray calls. */
uint64_t oracle_glsl_render_rows(const scene *world, uint8_t *frame, float *mean, size_t w, size_t h,
                                 size_t row_begin, size_t row_end, uint32_t spp, uint32_t max_bounces,
                                 float fov_rad, uint32_t random_seed) {
    uint64_t rays = 0;
    item_t *stack = malloc(sizeof(item_t) * ((size_t)max_bounces + 3));
    const float tan_fov_2 = tanf(fov_rad / 2.0f);
    const float fw = (float)w, fh = (float)h;
    const float aspect = fw / fh; /* src/main.c:319 */
    for (size_t sy = row_begin; sy < row_end && sy < h; sy++) {
        for (size_t sx = 0; sx < w; sx++) {
            /* the quad's interpolated coordinates at the pixel centre; +y is up */
            const float cx = ((float)sx + 0.5f) / fw * 2.0f - 1.0f;
            const float cy = 1.0f - ((float)sy + 0.5f) / fh * 2.0f;
            /* :276: uint(float) of a negative value is undefined in GLSL; taken
             * here as truncation toward zero, then two's complement */
            const float fidx = (cy * fh) * fw + (cx * fw);
            uint32_t rng = (uint32_t)(int64_t)fidx + random_seed;
            const v3 dir = normalize3(mk(cx * tan_fov_2, cy * tan_fov_2 / aspect, 1.0f));
            v3 result = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t s = 0; s < spp; s++) {
                rng += s;
                result = add(result, incident_light(world, (int)max_bounces, stack,
                                                    mk(0.0f, 0.0f, 0.0f), dir, &rng, &rays));
            }
            const float n = (float)spp;
            result = mk(result.x / n, result.y / n, result.z / n);
            const size_t at = 3 * (sy * w + sx);
            if (mean != NULL) {
                mean[at + 0] = result.x;
                mean[at + 1] = result.y;
                mean[at + 2] = result.z;
            }
            frame[at + 0] = unorm8(result.x);
            frame[at + 1] = unorm8(result.y);
            frame[at + 2] = unorm8(result.z);
        }
    }
    free(stack);
    return rays;
}
